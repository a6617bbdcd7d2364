"""Flat-buffer Adam with bf16 shadow weights: the parameter-sized work of one optimisation step in a handful of
launches (include/glr.h: glr_sumsq_partial / glr_clip_coef / glr_adam_step).

The reference trains under Lightning's native AMP with `torch.optim.Adam(betas=(0.5, 0.999))` and
`gradient_clip_val 0.25` (/root/reference/gloria/builder.py:84-87, run.py:172-207).  Done with stock torch pieces that
is, per step and independent of the batch size: one fp32->bf16 cast per weight (autocast), one bf16->fp32 cast per
gradient, ~130 fills, a 70-launch gradient-norm reduction, the clip multiply and the optimizer - about 600 launches
and 7 ms on one MI355X, which is what bounds a data-parallel rank at 32 pairs per GPU.  Here:

  * every trainable parameter lives in ONE flat fp32 master buffer (plus flat exp_avg / exp_avg_sq);
  * parameters of Linear / Conv modules are bf16 VIEWS of a flat shadow buffer that the Adam kernel
    rewrites from the masters - the forward needs no weight cast; their gradients arrive as bf16 views of a flat
    gradient buffer (what autocast's backward produces before its cast) - no gradient cast;
  * normalisation parameters, embeddings and everything else stay fp32 (views of the master buffer itself);
  * the gradient norm of clip_grad_norm_ is two launches over the flat gradient buffers, its coefficient never
    leaves the device and is applied inside the Adam kernel;
  * zero_grad is one memset per flat buffer; the flat gradient buffers are the all-reduce buckets of the
    data-parallel reducer (gloria.dist.GradReducer.from_flat).

Numerics are those of the AMP recipe: bf16(master) is exactly what autocast's cast hands the forward, and the fp32
Adam update runs on the masters.
"""

import torch

from . import _native as N


def _storage_flat(t):
    """1-D view of a dense tensor's elements in MEMORY order (channels-last conv weights stay as they lie)."""
    return torch.as_strided(t, (t.numel(),), (1,), t.storage_offset())


CHUNK = 16384          # elements per workgroup of the pointer-table kernels


class _Group:
    def __init__(self, params, shadow, device, flat_grads):
        self.params, self.shadow, self.flat_grads = params, shadow, flat_grads
        self.offsets, off = [], 0
        for p in params:
            self.offsets.append(off)
            off += (p.numel() + 7) // 8 * 8
        self.n = off
        f32 = dict(dtype=torch.float32, device=device)
        self.master = torch.zeros(self.n, **f32)
        self.exp_avg = torch.zeros(self.n, **f32)
        self.exp_avg_sq = torch.zeros(self.n, **f32)
        self.gdt = torch.bfloat16 if shadow else torch.float32
        self.grad = torch.zeros(self.n, dtype=self.gdt, device=device) if flat_grads else None
        self.shadow_buf = torch.zeros(self.n, dtype=torch.bfloat16, device=device) if shadow else None
        for p, o in zip(params, self.offsets):
            k = p.numel()
            with torch.no_grad():
                self.master[o:o + k].copy_(_storage_flat(p.data).float())
                src = self.shadow_buf if shadow else self.master
                if shadow:
                    src[o:o + k].copy_(self.master[o:o + k])
                p.data = src.as_strided(p.shape, p.stride(), storage_offset=o)
                p.grad = None
        # static chunk table { param, count, offset in the parameter, offset in the flat buffers } + a pinned staging
        # buffer for this step's gradient addresses (gradients stay one tensor per parameter, as autograd leaves them)
        import numpy as np
        ent, self.chunk_start = [], []
        for i, (p, o) in enumerate(zip(params, self.offsets)):
            self.chunk_start.append(len(ent))
            for c0 in range(0, p.numel(), CHUNK):
                ent.append((i, min(CHUNK, p.numel() - c0), c0, o + c0))
        self.chunk_start.append(len(ent))
        tab = np.zeros(len(ent), dtype=np.dtype([("param", "<i4"), ("count", "<i4"), ("poff", "<i8"), ("foff", "<i8")]))
        for k, e in enumerate(ent):
            tab[k] = e
        self.n_chunks = len(ent)
        self.table = torch.from_numpy(tab.view(np.uint8).copy()).to(device)
        # pinned staging buffers, ONE PER CALL SITE (= per bucket of the data-parallel reducer, keyed by its first
        # parameter) + an event each: the host may not rewrite a buffer whose asynchronous upload has not run yet, and
        # with a buffer of its own a bucket only ever waits for its OWN upload of the previous step - long done.  (Two
        # buffers used alternately made the third bucket of a step wait for the first one's upload, i.e. for the GPU to
        # reach it: the host could not run ahead of the backward pass - 14 ms inside run_backward at 32 pairs per rank.)
        self.ptr_stage = {}
        self.ptr_dev = torch.zeros(len(params), dtype=torch.int64, device=device)
        self._keep = []

    def stage_grad_pointers(self, i0=0, i1=None):
        """addresses of this step's gradients of parameters [i0, i1) (dense, parameter strides, group dtype) -> device"""
        i1 = len(self.params) if i1 is None else i1
        st = self.ptr_stage.get(i0)
        if st is None:
            st = self.ptr_stage[i0] = [torch.zeros(len(self.params), dtype=torch.int64).pin_memory(), None]
        if st[1] is not None:
            st[1].synchronize()                       # the upload that last used this staging buffer has run
        host = st[0]
        keep, ptrs = [], []
        gdt = self.gdt
        for p in self.params[i0:i1]:
            g = p.grad
            if g is None:
                ptrs.append(0)
                continue
            if g.dtype != gdt or g.stride() != p.stride():
                g = torch.empty_strided(p.shape, p.stride(), dtype=gdt, device=p.device).copy_(g)
            keep.append(g)
            ptrs.append(g.data_ptr())
        host.numpy()[i0:i1] = ptrs        # one vectorised write (a tensor element store costs microseconds each)
        self._keep = keep                 # alive until the kernels that read them have been queued (same stream)
        self.ptr_dev[i0:i1].copy_(host[i0:i1], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st[1] = ev

    def gather(self, i0, i1):
        """data-parallel: the gradients of parameters [i0, i1) -> their slots of the flat gradient buffer (one launch),
        then released; returns the slice of the flat buffer that holds them (an all-reduce bucket)"""
        self.stage_grad_pointers(i0, i1)
        c0, c1 = self.chunk_start[i0], self.chunk_start[i1]
        rec = 24                          # bytes per chunk-table entry
        N.check(N.lib().glr_gather_mt(N.ptr(self.table[c0 * rec:]), c1 - c0, N.ptr(self.ptr_dev), N.dtype_code(self.gdt),
                                      N.ptr(self.grad), N.stream()), "glr_gather_mt")
        for i in range(i0, i1):
            self.params[i].grad = None
        end = self.offsets[i1] if i1 < len(self.params) else self.n
        return self.grad[self.offsets[i0]:end]

    def view(self, buf, i):
        p, o = self.params[i], self.offsets[i]
        return buf.as_strided(p.shape, p.stride(), storage_offset=o)


class ShadowAdam(torch.optim.Optimizer):
    """Adam (torch semantics) over flat buffers.  `shadow_ids`: ids of the parameters that become bf16 shadows."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=None, shadow_ids=(),
                 flat_grads=False, all_params=None):
        """Gradients always stay where autograd puts them (one tensor per parameter: making `.grad` a view of a flat
        buffer costs an accumulation launch per parameter).  flat_grads=False (single process): the norm / Adam kernels
        read them through a pointer table.  flat_grads=True (data parallel): the reducer gathers every bucket's
        gradients into a flat buffer with one launch when its last gradient arrives (`_Group.gather`), all-reduces
        that slice, and the kernels read the flat buffer."""
        params = [p for p in params if p.requires_grad]
        if not params or not all(p.is_cuda for p in params):
            raise RuntimeError("ShadowAdam runs on GPU parameters (the flat kernels are HIP)")
        # the group carries torch.optim.Adam's own keys (at their defaults) so that its state_dict() loads into a stock
        # Adam and back
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        # `all_params`: every trainable parameter of the model in registration order, INCLUDING the ones this optimizer
        # leaves out (builder.build_optimizer drops parameters the graph never reaches).  Checkpoints are written and
        # read in that order - the layout torch.optim.Adam(model.parameters()) and the reference's Lightning
        # checkpoints use - so the optimizer state maps BY PARAMETER, never by position in this optimizer's own list.
        self.all_params = [p for p in (all_params if all_params is not None else params) if p.requires_grad]
        own = {id(p) for p in params}
        if not own <= {id(p) for p in self.all_params}:
            raise ValueError("ShadowAdam: all_params must contain every optimised parameter")
        dev = params[0].device
        shadow_ids = set(shadow_ids)
        # reverse registration order: backward produces gradients roughly back to front, so consecutive ranges of a
        # flat buffer complete together (the reducer's buckets)
        rev = list(reversed(params))
        self.flat_grads = bool(flat_grads)
        self.groups = [_Group(pl, sh, dev, self.flat_grads)
                       for pl, sh in (([p for p in rev if id(p) in shadow_ids], True),
                                      ([p for p in rev if id(p) not in shadow_ids], False)) if pl]
        self.max_grad_norm = float(max_grad_norm) if max_grad_norm else 0.0
        self.t = 0
        L = N.lib()
        self._nblocks = [L.glr_sumsq_blocks(g.n) if self.flat_grads else g.n_chunks for g in self.groups]
        self._partial = torch.zeros(sum(self._nblocks), dtype=torch.float32, device=dev)
        self.clip_state = torch.zeros(2, dtype=torch.float32, device=dev)          # [norm, coefficient] of the last step
        for g in self.groups:                     # per-parameter views of the moments: the stock state_dict layout
            for i, p in enumerate(g.params):
                self.state[p] = {"step": torch.zeros((), dtype=torch.float32), "exp_avg": g.view(g.exp_avg, i),
                                 "exp_avg_sq": g.view(g.exp_avg_sq, i)}

    # ---------------------------------------------------------------- one step
    def zero_grad(self, set_to_none=False):
        for g in self.groups:
            if self.flat_grads:
                g.grad.zero_()            # slots of parameters without a gradient (and the padding) read as zero
            for p in g.params:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        L, st = N.lib(), N.stream()
        o = 0
        for g, nb in zip(self.groups, self._nblocks):
            if self.flat_grads:
                N.check(L.glr_sumsq_partial(N.ptr(g.grad), N.dtype_code(g.gdt), g.n, N.ptr(self._partial[o:]), st),
                        "glr_sumsq_partial")
            else:
                g.stage_grad_pointers()
                N.check(L.glr_sumsq_mt(N.ptr(g.table), g.n_chunks, N.ptr(g.ptr_dev), N.dtype_code(g.gdt),
                                       N.ptr(self._partial[o:]), st), "glr_sumsq_mt")
            o += nb
        N.check(L.glr_clip_coef(N.ptr(self._partial), o, self.max_grad_norm, N.ptr(self.clip_state), st), "glr_clip_coef")
        self.t += 1
        pg = self.param_groups[0]
        hyper = (float(pg["lr"]), float(pg["betas"][0]), float(pg["betas"][1]), float(pg["eps"]),
                 float(pg["weight_decay"]), self.t, N.ptr(self.clip_state), st)
        for g in self.groups:
            if self.flat_grads:
                N.check(L.glr_adam_step(N.ptr(g.master), N.ptr(g.exp_avg), N.ptr(g.exp_avg_sq), N.ptr(g.grad),
                                        N.dtype_code(g.gdt), N.ptr(g.shadow_buf), g.n, *hyper), "glr_adam_step")
            else:
                N.check(L.glr_adam_step_mt(N.ptr(g.table), g.n_chunks, N.ptr(g.ptr_dev), N.dtype_code(g.gdt),
                                           N.ptr(g.master), N.ptr(g.exp_avg), N.ptr(g.exp_avg_sq), N.ptr(g.shadow_buf),
                                           *hyper), "glr_adam_step_mt")
        for s in self.state.values():
            s["step"] = torch.tensor(float(self.t))

    # ---------------------------------------------------------------- checkpoints: fp32 masters, stock layout
    def master_of(self, p):
        for g in self.groups:
            for i, q in enumerate(g.params):
                if q is p:
                    return g.view(g.master, i)
        return None

    def load_masters(self, named_fp32):
        """named_fp32: iterable of (parameter, fp32 tensor): masters (and shadows) are rewritten from it."""
        with torch.no_grad():
            for p, t in named_fp32:
                m = self.master_of(p)
                if m is not None:
                    m.copy_(t.to(m.device, torch.float32))
                    if p.dtype != torch.float32:
                        p.data.copy_(m)

    def state_dict(self):
        """stock layout over `all_params` (the reference optimizer's parameter list): index = position of the parameter
        in the model's registration order; parameters this optimizer leaves out simply have no state entry"""
        own = {id(p) for grp in self.param_groups for p in grp["params"]}
        state = {i: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.state[p].items()}
                 for i, p in enumerate(self.all_params) if id(p) in own}
        grp = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        grp["params"] = list(range(len(self.all_params)))
        return {"state": state, "param_groups": [grp]}

    def load_state_dict(self, state_dict):
        """stock layout in, flat buffers out: the moments are copied INTO the flat views.  Accepted: a state dict over
        the model's full parameter list (this class's own state_dict(), torch.optim.Adam(model.parameters()), the
        reference's checkpoints) or over exactly this optimizer's parameters; anything else raises."""
        sd = state_dict["state"]
        ids = [i for grp in state_dict["param_groups"] for i in grp["params"]]
        own = [p for grp in self.param_groups for p in grp["params"]]
        if len(ids) == len(self.all_params):
            target = self.all_params
        elif len(ids) == len(own):
            target = own
        else:
            raise ValueError(f"optimizer state covers {len(ids)} parameters; expected {len(self.all_params)} (model order) "
                             f"or {len(own)} (this optimizer's parameters)")
        steps = set()
        with torch.no_grad():
            for i, p in zip(ids, target):
                if i not in sd or p not in self.state:
                    continue              # no state saved / a parameter this optimizer does not train
                for k in ("exp_avg", "exp_avg_sq"):
                    if tuple(sd[i][k].shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state {k} of parameter {i} has shape {tuple(sd[i][k].shape)}, "
                                         f"the parameter {tuple(p.shape)}")
                    self.state[p][k].copy_(sd[i][k])
                steps.add(int(float(sd[i]["step"])))
        if len(steps) > 1:
            raise ValueError(f"optimizer state holds different step counts {sorted(steps)}: the flat Adam keeps one")
        if steps:
            self.t = steps.pop()
        for grp, src in zip(self.param_groups, state_dict["param_groups"]):
            for k in ("lr", "betas", "eps", "weight_decay"):
                grp[k] = src[k]
        for s in self.state.values():
            s["step"] = torch.tensor(float(self.t))


def shadow_parameter_ids(model):
    """parameters autocast would cast to bf16 on every use: those of Linear / Conv modules (weights and biases).
    Normalisation layers, free-standing parameters and EMBEDDINGS stay fp32: autocast leaves F.embedding alone, and
    the position / token-type rows collect hundreds of gradient contributions per step, which a bf16 gradient buffer
    would sum at 8 bits."""
    ids = set()
    for m in model.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Conv2d)):
            ids.update(id(p) for p in m.parameters(recurse=False))
    return ids
