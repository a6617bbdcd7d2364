"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI;
"gloo" in the CPU tests).  The reference has no collective call sites at all; this is
new capability whose oracle is the single-process full-batch result.

Collectives on the path:
  * all-gather of text embeddings (word-level [B, D, L] and global [B, D]) with a reduce-scatter
    of their gradients in backward;
  * all-gather of the similarity block-rows (256 KB at B = 256) for the column cross entropy;
  * bucketed SUM all-reduce of parameter gradients: `GradReducer` keeps every `.grad` as a view into a
    flat bucket and launches a bucket's all-reduce from a post-accumulate hook as soon as its gradients
    are ready, overlapped with the rest of backward (`DistContext.allreduce_grads` is the simple
    after-backward form kept for the CPU tests).  xGMI is point-to-point, so buckets are large (64 MiB)
    to amortise per-collective latency and let RCCL use all 7 links.
The loss every rank differentiates is the GLOBAL-batch mean, so gradients are summed, not averaged.
"""

import os

import torch
import torch.distributed as dist

_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


class _AllGatherGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        world = dist.get_world_size(group)
        x = x.contiguous()
        out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        group = ctx.group
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        g = g.contiguous()
        n = g.shape[0] // world
        if dist.get_backend(group) == "gloo":          # gloo has no reduce_scatter
            dist.all_reduce(g, group=group)
            return g[rank * n:(rank + 1) * n].clone(), None
        out = torch.empty((n,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
        dist.reduce_scatter_tensor(out, g, group=group)
        return out, None


class DistContext:
    def __init__(self, group=None, cpu_group=None):
        self.group = group
        self.cpu_group = cpu_group        # gloo group for host-side integers (caption lengths): no device round trip
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        # GLR_FORCE_DIST=1 runs the sharded code path (all-gather / reduce-scatter / bucketed all-reduce)
        # even with a single rank: lets one GPU rehearse what the 8-GPU node will execute
        self.active = self.world_size > 1 or os.environ.get("GLR_FORCE_DIST", "0") == "1"

    # -- embeddings
    def all_gather_grad(self, x):
        return _AllGatherGrad.apply(x, self.group)

    def all_gather_nograd(self, x):
        x = x.detach().contiguous()
        out = torch.empty((self.world_size * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x, group=self.group)
        return out

    def all_gather_ints(self, values):
        """host int lists of equal length per rank -> concatenated host list (one small collective).  With a gloo
        side group the integers never touch the GPU: a device collective would need a D2H copy and a stream sync per
        step before the host can plan the word tiles, draining the queue the encoders were launched into."""
        if self.cpu_group is not None:
            t = torch.tensor(list(values), dtype=torch.int32)
            out = torch.empty(self.world_size * t.numel(), dtype=torch.int32)
            dist.all_gather_into_tensor(out, t, group=self.cpu_group)
            return out.tolist()
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor(list(values), dtype=torch.int32, device=dev)
        out = torch.empty(self.world_size * t.numel(), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(out, t, group=self.group)
        return out.cpu().tolist()

    def all_reduce_scalar(self, value, like=None):
        """SUM of a scalar (tensor or Python number) over the group, returned on the caller's device."""
        if torch.is_tensor(value):
            t = value.detach().float().reshape(1).clone()
        else:
            dev = like.device if torch.is_tensor(like) else (
                torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else "cpu")
            t = torch.full((1,), float(value), dtype=torch.float32, device=dev)
        dist.all_reduce(t, group=self.group)
        return t[0]

    # -- parameter gradients
    def allreduce_grads(self, params, bucket_bytes=64 << 20):
        """SUM all-reduce of .grad over the group in flat buckets; returns after all buckets landed."""
        buckets, cur, size = [], [], 0
        for p in params:
            if p.grad is None:
                continue
            cur.append(p)
            size += p.grad.numel() * p.grad.element_size()
            if size >= bucket_bytes:
                buckets.append(cur)
                cur, size = [], 0
        if cur:
            buckets.append(cur)
        pending = []
        for b in buckets:
            flat = torch.cat([p.grad.reshape(-1) for p in b])
            work = dist.all_reduce(flat, group=self.group, async_op=True)
            pending.append((work, flat, b))
        for work, flat, b in pending:
            work.wait()
            off = 0
            for p in b:
                n = p.grad.numel()
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n

    def barrier(self):
        dist.barrier(group=self.group)


class GradReducer:
    """Bucketed gradient all-reduce overlapped with backward.

    Parameters are grouped (in reverse registration order, the order backward produces them) into flat
    buckets of `bucket_bytes`; every `.grad` is a VIEW into its bucket, so nothing is copied in or out.
    A post-accumulate hook counts ready gradients; when a bucket is complete its SUM all-reduce is
    launched asynchronously on the collective stream while backward keeps running.  `finish()` waits for
    all buckets.  xGMI is point-to-point: few large buckets (64 MiB) keep RCCL on all links."""

    @classmethod
    def from_flat(cls, groups, ctx, bucket_bytes=64 << 20, overlap=True):
        """groups: the parameter groups of a flat optimizer (gloria.optim.ShadowAdam, flat_grads=True).  A bucket is a
        run of consecutive parameters of one group (>= bucket_bytes of gradient); when its last gradient arrives the
        group gathers the bucket's gradients into the flat buffer with ONE launch (`group.gather`) and the slice is
        all-reduced asynchronously while backward keeps running.
        overlap=False: no hooks at all - `finish()` gathers each group with one launch and all-reduces it whole.  At
        small per-rank batches the step is bound by HOST time, and ~360 Python hook calls + 5 bucket launches from
        autograd's device thread (2.7 ms per step at 32 pairs) cost more than hiding a ~1 ms all-reduce buys."""
        if not overlap:
            bucket_bytes = float("inf")
        self = cls.__new__(cls)
        self.ctx, self.buckets, self._views, self._fired = ctx, [], {}, set()
        self._hide_unused = False
        self._flat = []                   # per bucket: (group, first parameter, one past the last)
        for g in groups:
            esz = 2 if g.gdt == torch.bfloat16 else 4
            start = 0
            for i in range(len(g.params)):
                end_off = g.offsets[i + 1] if i + 1 < len(g.params) else g.n
                if (end_off - g.offsets[start]) * esz >= bucket_bytes or i + 1 == len(g.params):
                    idx = len(self.buckets)
                    plist = g.params[start:i + 1]
                    self.buckets.append((None, plist))
                    self._flat.append((g, start, i + 1))
                    if overlap:
                        for q in plist:
                            q.register_post_accumulate_grad_hook(lambda _p, k=idx: self._on_ready(k, _p))
                    start = i + 1
        self._pending = []
        self._ready = [0] * len(self.buckets)
        self._done = [False] * len(self.buckets)
        self._streams = {}
        self._open = False
        return self

    def _join_streams(self, i):
        """The gradients of bucket i may come from several HIP streams (the two encoders run on two streams, and
        autograd runs a hook under the stream its gradient was produced on).  The bucket's gather launch and its
        all-reduce are queued on the CURRENT stream only, so it first waits for an event recorded NOW on every other
        stream that contributed (their kernels were queued before this hook ran), and the foreign gradients are marked
        as used by this stream so the caching allocator does not hand their memory out early."""
        seen = self._streams.get(i)
        if not seen:
            return
        cur = torch.cuda.current_stream()
        for handle, (st, plist) in seen.items():
            if handle == cur.cuda_stream:
                continue
            ev = torch.cuda.Event()
            ev.record(st)
            cur.wait_event(ev)
            for p in plist:
                if p.grad is not None:
                    p.grad.record_stream(cur)
        seen.clear()

    def _reduce_bucket(self, i):
        self._join_streams(i)
        if getattr(self, "_flat", None):
            g, i0, i1 = self._flat[i]
            buf = g.gather(i0, i1)
            self._done[i] = True
        else:
            buf = self.buckets[i][0]
        self._pending.append(dist.all_reduce(buf, group=self.ctx.group, async_op=True))

    def __init__(self, params, ctx, bucket_bytes=64 << 20):
        self.ctx = ctx
        self._hide_unused = True
        self.buckets = []                 # (flat tensor, [params])
        self._views = {}                  # param -> its gradient view into the bucket
        self._fired = set()               # params whose gradient arrived in the current backward
        cur, size = [], 0
        for p in reversed([p for p in params if p.requires_grad]):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self._seal(cur)
                cur, size = [], 0
        if cur:
            self._seal(cur)
        self._pending = []
        self._ready = [0] * len(self.buckets)
        self._streams = {}                # bucket -> {stream handle: (stream, [params])} of the current cycle
        self._open = False                # a zero_grad() .. finish() cycle is in progress

    def _seal(self, plist):
        dev, dt = plist[0].device, plist[0].dtype
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=dt, device=dev)
        off, idx = 0, len(self.buckets)
        for p in plist:
            n = p.numel()
            # the view gets the parameter's own strides (channels-last conv weights, incl. the ambiguous 1x1 case):
            # fused Adam requires params and grads with identical strides, and autograd's layout contract
            # then accumulates in place without a copy.  Parameters are dense, so n elements cover the view.
            view = flat.as_strided(p.shape, p.stride(), storage_offset=off)
            assert view.stride() == p.stride() and view.numel() == n
            self._views[p] = view
            p.grad = view
            off += n
            p.register_post_accumulate_grad_hook(lambda _p, i=idx: self._on_ready(i, _p))
        self.buckets.append((flat, plist))

    def zero_grad(self):
        if getattr(self, "_flat", None):
            for g in {id(f[0]): f[0] for f in self._flat}.values():
                g.grad.zero_()
                for p in g.params:
                    p.grad = None
            self._done = [False] * len(self.buckets)
        else:
            for flat, plist in self.buckets:
                flat.zero_()
                for p in plist:
                    p.grad = self._views[p]       # re-attach: finish() hides the views of unused parameters
        self._ready = [0] * len(self.buckets)
        self._pending = []
        self._fired = set()
        self._streams = {}
        self._open = True

    def _on_ready(self, i, p):
        if not self._open or p in self._fired:
            # a bucket may only be reduced once per cycle: a second backward would all-reduce already reduced sums
            raise RuntimeError("GradReducer: one backward per zero_grad()/finish() cycle (no gradient accumulation)")
        self._fired.add(p)
        self._ready[i] += 1
        if p.is_cuda:
            # which stream produced this gradient (plain-int handle: ~0.3 us per hook)
            h = _raw_stream(p.device.index) if _raw_stream is not None else torch.cuda.current_stream().cuda_stream
            seen = self._streams.setdefault(i, {})
            ent = seen.get(h)
            if ent is None:
                ent = seen[h] = (torch.cuda.current_stream(), [])
            ent[1].append(p)
        if self._ready[i] == len(self.buckets[i][1]):
            self._reduce_bucket(i)

    def finish(self):
        # parameters that received no gradient this step leave their bucket incomplete: reduce it now
        for i, (flat, plist) in enumerate(self.buckets):
            if self._ready[i] != len(plist):
                self._reduce_bucket(i)
        for w in self._pending:
            w.wait()
        self._pending = []
        self._open = False
        # a parameter no rank used (e.g. the BERT pooler with last_n_layers > 1) must look to the optimizer as it
        # does in the single-process run - grad None, skipped - not as a zero gradient that weight decay acts on
        if not self._hide_unused:
            return
        for _, plist in self.buckets:
            for p in plist:
                if p not in self._fired:
                    p.grad = None


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("GLR_FORCE_DIST", "0") == "1"
    if world <= 1 and not force:
        return None
    if world <= 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29533")
    if not dist.is_initialized():
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend)
    cpu_group = None
    if dist.get_backend() == "nccl":
        try:
            cpu_group = dist.new_group(backend="gloo")
        except Exception:          # noqa: BLE001 - gloo unavailable: fall back to the device collective
            cpu_group = None
    return DistContext(cpu_group=cpu_group)
