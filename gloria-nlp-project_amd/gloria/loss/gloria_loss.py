"""MI355X-native `gloria.loss.gloria_loss`: same names, arguments, return tuples and error
behaviour as /root/reference/gloria/loss/gloria_loss.py, computed by the hand-written HIP
kernels of libglr.so (include/glr.h) instead of a Python loop over sentences.

    cosine_similarity  (ref :11-16)      attention_fn (ref :19-63)
    global_loss        (ref :66-88)      local_loss   (ref :99-201)

Differences that are deliberate and documented in DESIGN.md:
  * GPU only.  CPU tensors raise: there is no CPU or eager fallback for the forward.
  * fp32 inputs run the fp32-MFMA kernels (the 1e-4 parity mode); bf16 inputs (autocast) run the
    bf16-MFMA kernels with fp32 accumulation / softmax / log / exp.  `set_compute_dtype` can force
    bf16 operands for fp32 inputs.
  * `local_loss` forms the whole B x B similarity matrix in ONE launch; nothing of size
    O(B * |img_features|) is kept for backward (the reference keeps one transposed copy per
    sentence).
"""

from typing import List, Optional, Sequence

import numpy as np
import os

import torch

from .. import _native as N
from . import _recompute as R

_COMPUTE_DTYPE = None          # None = follow the input dtype
# bench.py sets this to {} to collect HIP events on the launch stream (the stream torch's events record on is the
# stream libglr launches on: N.stream()).  Keys -> lists of (event, event):
#   "k1_fwd"     the K1 forward launches alone               "k1_fwd_op"  packing + Gram + tiling + K1 forward
#   "k1_bwd"     the K1 backward launch alone                "k1_bwd_op"  K1 backward + gradient GEMMs + scatter
# and "k1_flops": algorithmic forward FLOPs of every forward call.
PROFILE = None
_DEBUG_HOOK = None          # diagnostics only: called with the K1 backward outputs (tools)
HANDOVER_A1 = os.environ.get("GLR_K1_A1", "1") != "0"     # forward -> backward hand-over of a1 (A/B switch)


class _Range:
    """event pair around a region of the current stream, recorded only while PROFILE is a dict"""

    def __init__(self, key):
        self.key, self.on = key, PROFILE is not None

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            PROFILE.setdefault(self.key, []).append((self.e0, self.e1))
        return False


def set_compute_dtype(dtype: Optional[torch.dtype]):
    """Force the MFMA operand dtype (torch.float32 / torch.bfloat16) or None to follow inputs."""
    global _COMPUTE_DTYPE
    if dtype not in (None, torch.float32, torch.bfloat16):
        raise TypeError("compute dtype must be float32, bfloat16 or None")
    _COMPUTE_DTYPE = dtype


def _op_code(*tensors):
    if _COMPUTE_DTYPE is not None:
        return N.dtype_code(_COMPUTE_DTYPE)
    return N.GLR_BF16 if any(t.dtype == torch.bfloat16 for t in tensors) else N.GLR_F32


def _as_supported(t):
    if t.dtype in (torch.float32, torch.bfloat16):
        return t.contiguous()
    return t.float().contiguous()      # fp16 / fp64 inputs are computed in fp32


# ------------------------------------------------------------------------------------------
# fused local similarity (K1) as an autograd function
# ------------------------------------------------------------------------------------------

class _Opts:
    def __init__(self, temp1, temp2, temp3, agg, eps, want_attn, img_offset, word_start, pair_only, want_wctx,
                 want_amean=False):
        self.temp1, self.temp2, self.temp3 = float(temp1), float(temp2), float(temp3)
        self.agg, self.eps = agg, float(eps)
        self.want_attn, self.img_offset, self.word_start = want_attn, int(img_offset), int(word_start)
        self.pair_only, self.want_wctx, self.want_amean = pair_only, want_wctx, want_amean


def _pack_operands(img_features, words, no_attn_vec, cap_lens, o):
    """Operand packing shared by forward and backward: vt, gram, tp, tnorm + the tile plan."""
    L = N.lib()
    dev = img_features.device
    B, D = img_features.shape[:2]
    S = img_features[0, 0].numel()
    shift = 0 if no_attn_vec is None else 1
    s_eff = S + shift
    s_pad = L.glr_region_pad(s_eff)
    if s_pad > N.MAX_SPAD:
        raise ValueError(f"{s_eff} regions exceed the kernel limit of {N.MAX_SPAD}")
    if D % 64 != 0:
        raise ValueError(f"embedding dim must be a multiple of 64, got {D}")
    code = _op_code(img_features, words)
    odt = N.torch_dtype(code)
    plan = N.TilePlan(cap_lens, dev, L.glr_tile_capacity(code), allow_pairs=(s_pad == N.MAX_SPAD and s_eff < s_pad))
    # channels-last feature maps already are [B, S, D] in memory: no transpose needed
    img = img_features
    layout = 0
    if img.dim() == 4 and not img.is_contiguous() and img.is_contiguous(memory_format=torch.channels_last):
        layout = 1
    else:
        img = img.contiguous()
    in_code = N.dtype_code(img.dtype)
    if words.dtype != img.dtype:
        words = words.to(img.dtype)
    words = words.contiguous()
    na = None if no_attn_vec is None else no_attn_vec.detach().to(img.dtype).contiguous()
    st = N.stream()
    # K1 streams the K-tiled copies (contiguous 1-KiB DMA pieces); the row-major vt / tp stay for the gradient GEMMs
    vt = torch.empty(B, s_pad, D, dtype=odt, device=dev)
    vt_t = torch.empty_like(vt)
    N.check(L.glr_pack_regions_tiled(N.ptr(img), in_code, layout, N.ptr(na), N.ptr(vt), N.ptr(vt_t), B, D, S, code, st),
            "glr_pack_regions_tiled")
    gram = torch.bmm(vt, vt.transpose(1, 2))            # plain batched GEMM (hipBLASLt): G[b] = V^T V
    gram_t = torch.empty_like(gram)
    # tiling + the ones row (a padded region) out of which the forward pair kernel reads Z_w = sum_r e2[w, r]
    # (include/glr.h, tile_rowflags); every kernel masks padded regions, so nothing else sees the row
    N.check(L.glr_tile_gram(N.ptr(gram), N.ptr(gram_t), s_pad, B, s_eff, code, st), "glr_tile_gram")
    tp = torch.empty(plan.n_slots, D, dtype=odt, device=dev)
    tnorm = torch.empty(plan.n_slots, dtype=torch.float32, device=dev)
    N.check(L.glr_pack_words(N.ptr(words), in_code, N.ptr(plan.sent_slot0), N.ptr(plan.cap_lens), N.ptr(tp),
                             N.ptr(tnorm), words.shape[0], D, words.shape[2], o.word_start, plan.n_slots,
                             plan.capacity, code, st), "glr_pack_words")
    tp_t = torch.empty_like(tp)
    N.check(L.glr_tile_k(N.ptr(tp), N.ptr(tp_t), N.TILE_WORDS, plan.n_tiles, D * vt.element_size(), st), "glr_tile_k")
    return plan, code, vt, vt_t, gram_t, tp, tp_t, tnorm, s_eff, s_pad, shift


def _k1_args(plan, vt, gram, tp, tnorm, B, D, s_eff, o, backward=False):
    head = (N.ptr(vt), N.ptr(gram), N.ptr(tp), N.ptr(tnorm), N.ptr(plan.sent_slot0), N.ptr(plan.cap_lens),
            N.ptr(plan.tile_first), N.ptr(plan.order), N.ptr(plan.tile_nsub))
    if backward == "all_single":
        # extra gradient inputs (regulariser rows / attention maps): every tile through the single-tile kernel
        items = (N.ptr(plan.all_tile), plan.n_all, None, 0, None)
    else:
        items = (N.ptr(plan.single_tile) if plan.n_single else None, plan.n_single,
                 N.ptr(plan.pair_tile) if plan.n_pair else None, plan.n_pair) \
            + (() if backward else (plan.n_long_pair,)) + (N.ptr(plan.pair_desc),)
    return head + items + (plan.n_tiles, plan.n_sent, B, D, s_eff, o.temp1, o.temp2, o.temp3, N.AGG[o.agg], o.eps)


class LocalSimFn(torch.autograd.Function):
    """sim[b, i] for local images x all sentences (+ diagonal attention maps, + word-mean attention rows).
    HIP forward (K1) and HIP backward (K1 bwd + three plain GEMMs), including the gradients that arrive
    through the diagonal attention maps (attention-supervision loss) and through the word-mean rows
    (regularisers).  Only the standalone attention_fn outputs (pair mode: maps + weighted context of B
    pairs) are differentiated through the torch restatement in loss/_recompute.py."""

    @staticmethod
    def forward(ctx, img_features, words_emb, no_attn_vec, cap_lens, opts):
        N.require_cuda(img_features, words_emb, no_attn_vec)
        L = N.lib()
        o = opts
        img = _as_supported(img_features.detach()) if img_features.dtype not in (torch.float32, torch.bfloat16) \
            else img_features.detach()
        words = _as_supported(words_emb.detach())
        op_range = _Range("k1_fwd_op")
        op_range.__enter__()
        plan, code, vt, vt_t, gram_t, tp, tp_t, tnorm, s_eff, s_pad, shift = _pack_operands(img, words, no_attn_vec, cap_lens, o)
        dev = img.device
        B, D = img.shape[:2]
        n_sent = plan.n_sent
        need_grad = any(t is not None and t.requires_grad for t in (img_features, words_emb, no_attn_vec)) \
            and not o.pair_only
        sim = (torch.zeros if o.pair_only else torch.empty)(B, n_sent, dtype=torch.float32, device=dev)
        lse = torch.empty(B, n_sent, s_pad, dtype=torch.float32, device=dev) if need_grad else None
        wstat = torch.empty(B, plan.n_slots, 4, dtype=torch.float32, device=dev) if need_grad else None
        attn = attn_off = None
        strip = 0 if o.want_wctx else shift
        if o.want_attn:
            attn_off, off_host = plan.attn_offsets(s_eff - strip, dev)
            attn = torch.zeros(int(off_host[-1]), dtype=torch.float32, device=dev)
        amean = torch.empty(B, n_sent, s_pad, dtype=torch.float32, device=dev) if o.want_amean else None
        # the forward pair kernel hands the word-softmax values to the backward pair kernel (98 KB per image x pair, fp16,
        # in the kernels' own register order): the backward then skips its score stream
        a1buf = None
        if need_grad and plan.n_pair and HANDOVER_A1 and not o.pair_only:
            a1buf = torch.empty(B * plan.n_pair * 24576, dtype=torch.int32, device=dev)
        with _Range("k1_fwd"):
            N.check(L.glr_local_attn_fwd(*_k1_args(plan, vt_t, gram_t, tp_t, tnorm, B, D, s_eff, o), N.ptr(sim), n_sent,
                                         N.ptr(lse), N.ptr(wstat), N.ptr(attn), N.ptr(attn_off), strip,
                                         1 if o.pair_only else 0, o.img_offset, N.ptr(amean), N.ptr(a1buf), code, N.stream()),
                    "glr_local_attn_fwd")
        op_range.__exit__()
        if PROFILE is not None:
            n_words = plan.n_words if not o.pair_only else plan.n_words // max(n_sent, 1)
            PROFILE.setdefault("k1_flops", []).append((4.0 * s_eff * D + 6.0 * D) * B * n_words)
        wctx = None
        if o.want_wctx:      # attention_fn's first output: V . a2^T for the B diagonal pairs (plain bmm)
            n = int(plan.cap_lens_host[0])
            a2 = attn.view(B, n, s_eff)
            wctx = torch.bmm(a2.to(vt.dtype), vt[:, :s_eff]).transpose(1, 2).float().contiguous()   # [B, D, n]
            attn = a2[:, :, shift:].contiguous().view(-1)
        ctx.save_for_backward(img_features, words_emb, no_attn_vec, vt, vt_t, gram_t, tp, tp_t, tnorm, sim, lse, wstat, a1buf)
        ctx.plan, ctx.opts, ctx.meta = plan, o, (code, s_eff, s_pad, shift)
        ctx.set_materialize_grads(False)       # unused outputs (maps, context) arrive as None
        if attn is None:
            attn = sim.new_zeros(0)
        if wctx is None:
            wctx = sim.new_zeros(0)
        if amean is None:
            amean = sim.new_zeros(0)
        return sim, attn, wctx, amean

    @staticmethod
    def backward(ctx, dsim, dattn, dwctx, damean):
        img_features, words_emb, no_attn_vec, vt, vt_t, gram_t, tp, tp_t, tnorm, sim, lse, wstat, a1buf = ctx.saved_tensors
        plan, o = ctx.plan, ctx.opts
        code, s_eff, s_pad, shift = ctx.meta
        L = N.lib()
        dev = img_features.device
        B, D = img_features.shape[:2]
        d_img = None              # allocated by whoever contributes first (the common case needs no fp32 staging at all)
        d_words = torch.zeros(words_emb.shape, dtype=torch.float32, device=dev)
        d_na = None if no_attn_vec is None else torch.zeros(D, dtype=torch.float32, device=dev)

        have_dam = o.want_amean and damean is not None and damean.numel() > 0
        need_attn = o.want_attn and dattn is not None and dattn.numel() > 0
        need_wctx = o.want_wctx and dwctx is not None and dwctx.numel() > 0
        # the gradient of the diagonal attention maps (attention-supervision loss) goes through K1 backward too;
        # only attention_fn's own outputs (pair mode / weighted context) keep the torch restatement below
        hip_attn = need_attn and not o.pair_only and not o.want_wctx
        if (dsim is not None or have_dam or hip_attn) and not o.pair_only:
            if lse is None:
                raise RuntimeError("local similarity was computed without gradient state")
            if dsim is None:
                dsim = torch.zeros_like(sim)
            dam = damean.float().contiguous() if have_dam else None
            dat = dattn.float().contiguous() if hip_attn else None
            strip = 0 if o.want_wctx else shift
            dat_off = plan.attn_offsets(s_eff - strip, dev)[0] if hip_attn else None
            odt = vt.dtype
            ns = plan.n_slots
            xout = torch.empty(ns, B, s_pad, dtype=odt, device=dev)
            aout = torch.empty(B, ns, s_pad, dtype=odt, device=dev)
            baout = torch.empty(B, ns, s_pad, dtype=odt, device=dev)
            gamma = torch.empty(B, ns, dtype=torch.float32, device=dev)
            beta = torch.empty(B, ns, dtype=torch.float32, device=dev)
            g = dsim.float().contiguous()
            bwd_range = _Range("k1_bwd_op")
            bwd_range.__enter__()
            with _Range("k1_bwd"):
                # the extra gradient inputs (regulariser rows / attention maps) ride on the pair kernel's a1 hand-over
                # variant; without the hand-over buffer every tile goes through the single-tile kernel
                mode = "all_single" if ((dam is not None or dat is not None) and a1buf is None) else True
                N.check(L.glr_local_attn_bwd(*_k1_args(plan, vt_t, gram_t, tp_t, tnorm, B, D, s_eff, o, mode), N.ptr(sim),
                                             N.ptr(g), plan.n_sent, N.ptr(lse), N.ptr(wstat), N.ptr(dam), N.ptr(dat),
                                             N.ptr(dat_off), strip, o.img_offset, N.ptr(xout), N.ptr(aout), N.ptr(baout),
                                             N.ptr(gamma), N.ptr(beta), N.ptr(a1buf), code, N.stream()), "glr_local_attn_bwd")
            if _DEBUG_HOOK is not None:
                _DEBUG_HOOK(xout, aout, baout, gamma, beta)
            # gradient GEMMs (plain library GEMMs on the kernel's outputs)
            x2d = xout.view(ns, B * s_pad)
            dtp = (x2d @ vt.view(B * s_pad, D)).float() - gamma.sum(0).unsqueeze(1) * tp.float()      # [ns, D]
            P = torch.bmm(baout.transpose(1, 2), aout)                                                 # [B, S, S], baout = beta a2
            # dvt = X^T tp - P vt: the second product accumulates onto the first inside the GEMM (one rounding of the
            # fp32 sum to the operand dtype instead of two rounded products subtracted by three elementwise passes)
            dvt = torch.baddbmm((x2d.t() @ tp).view(B, s_pad, D), P, vt, alpha=-1.0)                   # [B, S, D]
            si, wi, slot = plan.word_index(dev)
            d_words.permute(0, 2, 1)[si, wi + o.word_start] = dtp[slot]
            cl = (img_features.dim() == 4 and img_features.dtype == dvt.dtype and not img_features.is_contiguous()
                  and img_features.is_contiguous(memory_format=torch.channels_last))
            if cl and not (need_attn and not hip_attn) and not need_wctx:
                # channels-last features ARE [B, S, D] in memory: the packed-region gradient, minus the no-attention row
                # and the padding, is the feature gradient - a strided view, no transpose / zero-fill / cast pass
                H_, W_ = img_features.shape[2], img_features.shape[3]
                d_img = dvt[:, shift:s_eff].unflatten(1, (H_, W_)).permute(0, 3, 1, 2)
            else:
                d_img = dvt[:, shift:s_eff].transpose(1, 2).float()
            if d_na is not None:
                d_na += dvt[:, 0].float().sum(0)
            bwd_range.__exit__()

        need_attn = need_attn and not hip_attn
        if need_attn or need_wctx:
            # gradient through the attention maps of the B diagonal pairs (torch restatement, small)
            img = img_features.detach().float().reshape(B, D, -1)
            words = words_emb.detach().float().requires_grad_(True)
            na = None if no_attn_vec is None else no_attn_vec.detach().float().requires_grad_(True)
            with torch.enable_grad():
                vc = img.clone().requires_grad_(True)
                V = vc if na is None else torch.cat([na.view(1, D, 1).expand(B, D, 1), vc], 2)
                lens = plan.cap_lens_host[o.img_offset:o.img_offset + B]
                a2 = R.diag_attention(V, words[o.img_offset:o.img_offset + B], lens, o.word_start, o.temp1)
                loss = 0.0
                if need_attn:
                    flat = torch.cat([a2[b, :int(lens[b]), shift:].reshape(-1) for b in range(B)])
                    off0 = int((plan.cap_lens_host[:o.img_offset].astype(np.int64) * (s_eff - shift)).sum())
                    loss = loss + (flat * dattn[off0:off0 + flat.numel()].float()).sum()
                if need_wctx:
                    ctxv = torch.einsum("bdr,bwr->bdw", V, a2)
                    loss = loss + (ctxv * dwctx[:, :, :ctxv.shape[2]].float()).sum()
                loss.backward()
            d_img = vc.grad if d_img is None else d_img.float() + vc.grad
            d_words += words.grad
            if na is not None and na.grad is not None:
                d_na += na.grad
        if d_img is None:
            d_img = torch.zeros(img_features.shape, dtype=img_features.dtype, device=dev)
        return (d_img.reshape(img_features.shape).to(img_features.dtype), d_words.to(words_emb.dtype),
                None if d_na is None else d_na.to(no_attn_vec.dtype), None, None)


# ------------------------------------------------------------------------------------------
# dual cross entropy (K2) and global similarity (K3)
# ------------------------------------------------------------------------------------------

class DualCEFn(torch.autograd.Function):
    """(loss0, loss1) of a full B x B matrix; gradient only for this rank's block of rows."""

    @staticmethod
    def forward(ctx, sim_rows, sim_full, row0):
        N.require_cuda(sim_rows, sim_full)
        L = N.lib()
        full = sim_full.detach().float().contiguous()
        B = full.shape[0]
        if full.shape[1] != B:
            raise ValueError("similarity matrix must be square")
        lse = torch.empty(2, B, dtype=torch.float32, device=full.device)
        losses = torch.empty(2, dtype=torch.float32, device=full.device)
        N.check(L.glr_dual_ce_fwd(N.ptr(full), B, N.ptr(lse[0]), N.ptr(lse[1]), N.ptr(losses), N.stream()),
                "glr_dual_ce_fwd")
        ctx.save_for_backward(full, lse)
        ctx.row0, ctx.n_rows = int(row0), sim_rows.shape[0]
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g0, g1):
        full, lse = ctx.saved_tensors
        B = full.shape[0]
        g = torch.stack([g0.float() if g0 is not None else full.new_zeros(()),
                         g1.float() if g1 is not None else full.new_zeros(())]).contiguous()
        dsim = torch.empty(ctx.n_rows, B, dtype=torch.float32, device=full.device)
        N.check(N.lib().glr_dual_ce_bwd(N.ptr(full), B, N.ptr(lse[0]), N.ptr(lse[1]), N.ptr(g), ctx.row0,
                                        ctx.n_rows, N.ptr(dsim), N.stream()), "glr_dual_ce_bwd")
        return dsim, None, None


def dual_cross_entropy(sim_rows, sim_full=None, row0=0):
    """CE(sim, arange) and CE(sim^T, arange) (ref :86-87, :167-170).  `sim_rows` is this process's
    block of rows of the square matrix `sim_full` (default: the matrix itself)."""
    if sim_full is None:
        sim_full = sim_rows
    return DualCEFn.apply(sim_rows, sim_full, row0)


class GlobalSimFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, txt, temp3, eps):
        N.require_cuda(img, txt)
        L = N.lib()
        a, t = img.detach().float().contiguous(), txt.detach().float().contiguous()
        Bi, D = a.shape
        Bt = t.shape[0]
        sim = torch.empty(Bi, Bt, dtype=torch.float32, device=a.device)
        ni = torch.empty(Bi, dtype=torch.float32, device=a.device)
        nt = torch.empty(Bt, dtype=torch.float32, device=a.device)
        N.check(L.glr_global_sim_fwd(N.ptr(a), N.ptr(t), Bi, Bt, D, float(temp3), float(eps), N.ptr(sim), Bt,
                                     N.ptr(ni), N.ptr(nt), N.stream()), "glr_global_sim_fwd")
        ctx.save_for_backward(a, t, ni, nt, sim)
        ctx.temp3, ctx.eps, ctx.dt = float(temp3), float(eps), (img.dtype, txt.dtype)
        return sim

    @staticmethod
    def backward(ctx, dsim):
        a, t, ni, nt, sim = ctx.saved_tensors
        Bi, D = a.shape
        Bt = t.shape[0]
        d = dsim.float().contiguous()
        da, dt_ = torch.empty_like(a), torch.empty_like(t)
        N.check(N.lib().glr_global_sim_bwd(N.ptr(a), N.ptr(t), N.ptr(ni), N.ptr(nt), N.ptr(sim), N.ptr(d), Bt, Bi, Bt, D,
                                           ctx.temp3, ctx.eps, N.ptr(da), N.ptr(dt_), N.stream()),
                "glr_global_sim_bwd")
        return da.to(ctx.dt[0]), dt_.to(ctx.dt[1]), None, None


def global_similarity(cnn_code, rnn_code, eps=1e-8, temp3=10.0):
    """temp3 * cosine matrix [B_img, B_txt] (ref :75-80)."""
    return GlobalSimFn.apply(cnn_code, rnn_code, temp3, eps)


# ------------------------------------------------------------------------------------------
# the reference's public functions
# ------------------------------------------------------------------------------------------

class CosineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, eps):
        N.require_cuda(x1, x2)
        a, b = x1.detach().float().contiguous(), x2.detach().float().contiguous()
        rows, D = a.shape
        out = torch.empty(rows, dtype=torch.float32, device=a.device)
        stats = torch.empty(rows, 3, dtype=torch.float32, device=a.device)
        N.check(N.lib().glr_cosine_fwd(N.ptr(a), N.ptr(b), rows, D, float(eps), N.ptr(out), N.ptr(stats), N.stream()),
                "glr_cosine_fwd")
        ctx.save_for_backward(a, b, stats)
        ctx.eps, ctx.dt = float(eps), (x1.dtype, x2.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b, stats = ctx.saved_tensors
        rows, D = a.shape
        d1, d2 = torch.empty_like(a), torch.empty_like(b)
        N.check(N.lib().glr_cosine_bwd(N.ptr(a), N.ptr(b), N.ptr(stats), N.ptr(g.float().contiguous()), rows, D,
                                       ctx.eps, N.ptr(d1), N.ptr(d2), N.stream()), "glr_cosine_bwd")
        return d1.to(ctx.dt[0]), d2.to(ctx.dt[1]), None


def cosine_similarity(x1, x2, dim=1, eps=1e-8):
    """Returns cosine similarity between x1 and x2, computed along dim (ref :11-16: the clamp is
    applied to the product of the norms).  Inside local_loss the cosine is fused into K1; this
    standalone entry point (used by the retrieval / inference helpers) runs glr_cosine_fwd/bwd."""
    N.require_cuda(x1, x2)
    a, b = x1.movedim(dim, -1), x2.movedim(dim, -1)
    shape = a.shape[:-1]
    out = CosineFn.apply(a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1]), eps)
    return out.view(shape).to(x1.dtype).squeeze()


def attention_fn(query, context, temp1, no_attn_vec=None):
    """
    query: batch x ndf x queryL
    context: batch x ndf x ih x iw (sourceL=ihxiw)
    returns (weightedContext [batch, ndf, queryL], attn [batch, queryL, ih, iw])   (ref :19-63)
    One launch of K1 in pair mode: image b attends with query b.
    """
    B, D, n = query.shape
    ih, iw = context.size(2), context.size(3)
    opts = _Opts(temp1, 1.0, 1.0, "sum", 1e-8, True, 0, 0, True, True)
    _, attn, wctx, _ = LocalSimFn.apply(context, query, no_attn_vec, [n] * B, opts)
    return wctx[:, :, :n].to(query.dtype), attn.view(B, n, ih, iw).to(query.dtype)


def global_loss(cnn_code, rnn_code, eps=1e-8, temp3=10.0):
    """(loss0, loss1) of the global image-sentence InfoNCE (ref :66-88)."""
    if cnn_code.dim() == 3:
        cnn_code, rnn_code = cnn_code.squeeze(0), rnn_code.squeeze(0)
    sim = global_similarity(cnn_code, rnn_code, eps=eps, temp3=temp3)
    return dual_cross_entropy(sim)


def local_similarity(img_features, words_emb, cap_lens: Sequence[int], temp1=4.0, temp2=5.0, temp3=10.0,
                     agg="sum", no_attn_vec=None, eps=1e-8, want_attn=True, img_offset=0, word_start=0,
                     want_amean=False):
    """B_img x n_sent similarity matrix (already * temp3) + flat diagonal attention maps
    (+ with want_amean the word-mean attention rows [B_img, n_sent, S_pad] of ALL pairs, else None)."""
    opts = _Opts(temp1, temp2, temp3, agg, eps, want_attn, img_offset, word_start, False, False, want_amean)
    sim, attn, _, amean = LocalSimFn.apply(img_features, words_emb, no_attn_vec, [int(c) for c in cap_lens], opts)
    return sim, attn, (amean if want_amean else None)


class AttnRegFn(torch.autograd.Function):
    """Per-image sums of the attention regularisers (K6): out[b] = (sum_i entropy, sum_{i != d} sym-KL, no-attn score)."""

    @staticmethod
    def forward(ctx, amean, s_eff, shift, img_offset):
        N.require_cuda(amean)
        a = amean.detach().float().contiguous()
        B, n_sent, s_pad = a.shape
        out = torch.empty(B, 4, dtype=torch.float32, device=a.device)
        N.check(N.lib().glr_attn_reg_fwd(N.ptr(a), B, n_sent, s_pad, int(s_eff), int(shift), int(img_offset), N.ptr(out),
                                         N.stream()), "glr_attn_reg_fwd")
        ctx.save_for_backward(a)
        ctx.meta = (int(s_eff), int(shift), int(img_offset))
        return out[:, 0].sum(), out[:, 1].sum(), out[:, 2].sum()

    @staticmethod
    def backward(ctx, g_ent, g_kl, g_na):
        (a,) = ctx.saved_tensors
        s_eff, shift, img_offset = ctx.meta
        B, n_sent, s_pad = a.shape
        z = a.new_zeros(())
        coef = torch.stack([g if g is not None else z for g in (g_ent, g_kl, g_na)]).float().contiguous()
        dam = torch.empty_like(a)
        N.check(N.lib().glr_attn_reg_bwd(N.ptr(a), B, n_sent, s_pad, s_eff, shift, img_offset, N.ptr(coef), N.ptr(dam),
                                         N.stream()), "glr_attn_reg_bwd")
        return dam, None, None, None


def attention_regularisers(amean, s_eff, shift, img_offset, no_attn_loss_weight, attention_divergence_loss_weight,
                           attention_entropy_loss_weight):
    """(no_attn_loss, kl_loss, entropy_loss) of ref :172-199 from the word-mean attention rows of this
    process's images against ALL n_sent sentences.  The means run over the GLOBAL batch (n_sent), so the values
    of data-parallel ranks add up to the single-process value."""
    n = amean.shape[1]
    ent_sum, kl_sum, na_sum = AttnRegFn.apply(amean, s_eff, shift, img_offset)
    no_attn_loss = no_attn_loss_weight * (na_sum / n) if no_attn_loss_weight is not None else 0            # :173-177
    kl_loss = attention_divergence_loss_weight * (-(kl_sum / (n * (n - 1)))) \
        if attention_divergence_loss_weight is not None else 0                                              # :180-192
    entropy_loss = ent_sum / (n * n) if attention_entropy_loss_weight is not None else 0                  # :195-197 (weight unused)
    return no_attn_loss, kl_loss, entropy_loss


class AttentionMaps(list):
    """The reference's list of [1, n_i, ih, iw] attention maps, plus the flat buffer they are views
    of (so the attention-supervision kernel can consume them without re-packing)."""
    flat = None
    cap_lens = None
    first = 0
    hw = (0, 0)


class AttnSupFn(torch.autograd.Function):
    """mean_b -log sum(label_b * normalised nearest-upsampled mean attention)  (K4, ref gloria_model.py:143-147)."""

    @staticmethod
    def forward(ctx, attn_flat, labels, cap_lens, first, count, ih, iw):
        N.require_cuda(attn_flat, labels)
        dev = attn_flat.device
        cl = np.asarray(cap_lens, dtype=np.int64)
        off = np.zeros(len(cl) + 1, dtype=np.int64)
        np.cumsum(cl * ih * iw, out=off[1:])
        off_d = N.upload(off[:-1].copy(), dev)
        cl_d = N.upload(cl.astype(np.int32), dev)
        lab = labels.detach().to(torch.uint8).contiguous()
        a = attn_flat.detach().float().contiguous()
        loss_b = torch.empty(count, dtype=torch.float32, device=dev)
        dmap = torch.zeros_like(a)
        N.check(N.lib().glr_attn_sup_fwd(N.ptr(a), N.ptr(off_d), N.ptr(cl_d), int(first), N.ptr(lab), int(count),
                                         lab.shape[1], lab.shape[2], ih, iw, N.ptr(loss_b), N.ptr(dmap), N.stream()),
                "glr_attn_sup_fwd")
        ctx.save_for_backward(dmap)
        ctx.count = count
        return loss_b.mean()

    @staticmethod
    def backward(ctx, g):
        (dmap,) = ctx.saved_tensors
        return dmap * (g / ctx.count), None, None, None, None, None, None


def attention_supervision_loss(att_maps, segmentation_labels):
    """-log sum(label * U / sum U).mean() over the local diagonal pairs, U the nearest-upsampled
    word-mean attention map (ref gloria_model.py:143-147), from the flat map buffer (K4)."""
    ih, iw = att_maps.hw
    return AttnSupFn.apply(att_maps.flat, segmentation_labels, att_maps.cap_lens, att_maps.first, len(att_maps),
                           ih, iw)


def split_attention_maps(attn_flat, cap_lens, ih, iw, first=0, count=None) -> List[torch.Tensor]:
    """flat diagonal maps -> the reference's list of [1, n_i, ih, iw] tensors (ref :141-143)."""
    count = len(cap_lens) if count is None else count
    maps, off = AttentionMaps(), int(sum(int(c) for c in cap_lens[:first])) * ih * iw
    maps.flat, maps.cap_lens, maps.first, maps.hw = attn_flat, [int(c) for c in cap_lens], first, (ih, iw)
    for i in range(first, first + count):
        n = int(cap_lens[i])
        maps.append(attn_flat[off:off + n * ih * iw].view(1, n, ih, iw))
        off += n * ih * iw
    return maps


def local_loss(
    img_features, words_emb, cap_lens, temp1=4.0, temp2=5.0, temp3=10.0, agg="sum", no_attn_vec=None,
    no_attn_loss_weight=None, attention_divergence_loss_weight=None, attention_entropy_loss_weight=None
):
    """Local region x word InfoNCE (ref :99-201).  Returns
    (loss0, loss1, no_attn_loss, kl_loss, entropy_loss, att_maps)."""
    want_aux = (no_attn_loss_weight is not None or attention_divergence_loss_weight is not None
                or attention_entropy_loss_weight is not None)
    ih, iw = img_features.shape[2], img_features.shape[3]
    cap_lens = [int(c) for c in cap_lens]
    sim, attn, amean = local_similarity(img_features, words_emb, cap_lens, temp1, temp2, temp3, agg, no_attn_vec,
                                        want_amean=want_aux)
    loss0, loss1 = dual_cross_entropy(sim)
    att_maps = split_attention_maps(attn, cap_lens, ih, iw)
    no_attn_loss = kl_loss = entropy_loss = 0
    if want_aux:
        shift = 0 if no_attn_vec is None else 1
        no_attn_loss, kl_loss, entropy_loss = attention_regularisers(
            amean, ih * iw + shift, shift, 0, no_attn_loss_weight, attention_divergence_loss_weight,
            attention_entropy_loss_weight)
    return loss0, loss1, no_attn_loss, kl_loss, entropy_loss, att_maps
