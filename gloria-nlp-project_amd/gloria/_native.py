"""ctypes binding of libglr.so (C ABI: include/glr.h).

The product path has NO fallback: if the HIP library is missing or a call fails, a
RuntimeError is raised.  torch is used only for device memory and the current stream.
"""

import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_void_p, POINTER

import numpy as np
import torch

GLR_F32, GLR_BF16 = 0, 1
AGG = {"sum": 0, "mean": 1, "max": 2}
TILE_WORDS = 64
MAX_SPAD = 384

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "lib", "libglr.so"))

SYMBOLS = {
    # name: (restype, argtypes)
    "glr_version": (c_int, []),
    "glr_region_pad": (c_int, [c_int]),
    "glr_tile_capacity": (c_int, [c_int]),
    "glr_plan_tiles_bound": (c_int, [c_void_p, c_int, c_int]),
    "glr_plan_tiles": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "glr_plan_items": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "glr_plan_rowflags": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "glr_plan_pair_desc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    "glr_tile_k": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_longlong, c_int, c_void_p]),
    "glr_pack_regions": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "glr_pack_regions_tiled": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_void_p]),
    "glr_tile_gram": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_longlong, c_int, c_int, c_void_p]),
    "glr_pack_words": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                               c_int, c_int, c_int, c_void_p]),
    "glr_local_attn_fwd": (c_int, [c_void_p] * 10 + [c_int, c_void_p, c_int, c_int, c_void_p] + [c_int] * 5 + [c_float] * 3 + [c_int, c_float, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
                                   # ..., sim, ld, lse, wstat, attn, attn_off, strip, pair_only, img_offset, amean, a1buf, dtype, stream
    "glr_local_attn_bwd": (c_int, [c_void_p] * 10 + [c_int, c_void_p, c_int, c_void_p] + [c_int] * 5 + [c_float] * 3
                           + [c_int, c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
                                   # sim, dsim, ld, lse, wstat, damean, dattn, attn_off, strip, img_offset,
                                   # xout, aout, baout, gamma, beta, a1buf, dtype, stream
    "glr_sumsq_blocks": (c_int, [ctypes.c_longlong]),
    "glr_gather_mt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "glr_sumsq_mt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "glr_adam_step_mt": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float,
                                 c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "glr_sumsq_partial": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_void_p, c_void_p]),
    "glr_clip_coef": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "glr_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, ctypes.c_longlong, c_float,
                              c_float, c_float, c_float, c_float, c_int, c_void_p, c_void_p]),
    "glr_bn_workspace_floats": (c_int, [ctypes.c_longlong, c_int]),
    "glr_bn_act_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_float, c_float, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "glr_bn_act_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong,
                               c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "glr_ln_workspace_floats": (c_int, [ctypes.c_longlong, c_int]),
    "glr_drop_add_ln_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong, c_int, c_float, c_float,
                                    ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "glr_drop_add_ln_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_longlong,
                                    c_int, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "glr_embedding_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "glr_type_embedding_workspace_floats": (c_int, [ctypes.c_longlong, c_int]),
    "glr_type_embedding_bwd": (c_int, [c_void_p, c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_colsum_workspace_floats": (c_int, [ctypes.c_longlong, c_int]),
    "glr_colsum_bf16": (c_int, [c_void_p, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "glr_attn_max_tokens": (c_int, [c_int]),
    "glr_attn_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_float,
                             ctypes.c_ulonglong, ctypes.c_ulonglong, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "glr_attn_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                             c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    "glr_upsample_bilinear_cl": (c_int, [c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong,
                                         c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "glr_maxpool3s2_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_maxpool3s2_bwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "glr_cell_counts": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_kth_value": (c_int, [c_void_p, c_int, ctypes.c_longlong, ctypes.c_longlong, c_void_p, c_void_p]),
    "glr_topk_desc": (c_int, [c_void_p, c_int, ctypes.c_longlong, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_threshold_counts": (c_int, [c_void_p, c_void_p, c_void_p, c_int, ctypes.c_longlong, c_void_p, c_void_p]),
    "glr_attn_reg_fwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "glr_attn_reg_bwd": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_image_minmax": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "glr_collate_images": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_aug_geom": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "glr_aug_jitter": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "glr_u8_to_tensor": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "glr_selftest_quotient": (c_int, [c_int, c_int, c_void_p, c_void_p]),
    "glr_dual_ce_fwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "glr_dual_ce_bwd": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "glr_global_sim_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_int,
                                   c_void_p, c_void_p, c_void_p]),
    "glr_global_sim_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p]),
    "glr_wordpiece_segsum_fwd": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "glr_wordpiece_segsum_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                         c_int, c_void_p]),
    "glr_attn_sup_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "glr_cosine_fwd": (c_int, [c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p]),
    "glr_cosine_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p,
                               c_void_p]),
}

_lib = None


def lib():
    """Load libglr.so (once).  Raises if it has not been built: there is no CPU path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the GLoRIA hot path needs the HIP library "
                "(build it with `python -c 'import __graft_entry__ as g; g.build()'` or `make -C csrc`)")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(handle, name)          # AttributeError if the ABI is incomplete
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        raise RuntimeError(f"libglr: {what} failed with code {code}")


def dtype_code(dt):
    if dt == torch.float32:
        return GLR_F32
    if dt == torch.bfloat16:
        return GLR_BF16
    raise TypeError(f"libglr supports float32 and bfloat16 tensors, got {dt}")


def torch_dtype(code):
    return torch.float32 if code == GLR_F32 else torch.bfloat16


def ptr(t):
    """device address of a tensor as a plain int (ctypes converts it for a c_void_p parameter) or None"""
    return None if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """the current HIP stream of the current device as a plain int (ctypes converts it for a void* parameter); the raw
    accessor avoids building a torch.cuda.Stream object per launch (the 32-pair step is bound by host time)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def upload(array, device):
    """host numpy array -> device tensor through PINNED staging memory, asynchronously on the current stream.  A copy
    from pageable memory is staged by the runtime and can hold the host until the stream has drained - in the middle of
    a training step that parks the host behind the whole encoder forward, and the launches that follow then run with the
    GPU idle between them (measured as 0.4 - 0.8 ms of gaps inside the K1 forward op at 256 pairs).  torch's caching
    host allocator recycles the pinned block once the copy's event has passed."""
    t = torch.from_numpy(np.ascontiguousarray(array))
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("the GLoRIA hot path runs on the GPU only (HIP kernels); got a CPU tensor")


MAX_PAIR_SEG = 8        # sentences per tile pair (table rows of the pair kernels)


class TilePlan:
    """Sentence -> word-slot packing (host planning by glr_plan_tiles + device copies)."""

    def __init__(self, cap_lens, device, capacity=TILE_WORDS, allow_pairs=True):
        cl = np.ascontiguousarray(np.asarray(cap_lens, dtype=np.int32))
        n = int(cl.shape[0])
        L = lib()
        bound = L.glr_plan_tiles_bound(cl.ctypes.data_as(c_void_p), n, capacity)
        if bound <= 0:
            raise ValueError(f"cap_lens must be in [1, 512] (glr_plan_tiles_bound -> {bound})")
        slot0 = np.zeros(n, dtype=np.int32)
        tile_first = np.zeros(bound + 1, dtype=np.int32)
        order = np.zeros(bound, dtype=np.int32)
        nsub = np.zeros(bound, dtype=np.int32)
        pairing = bool(allow_pairs and capacity == TILE_WORDS)
        nt = L.glr_plan_tiles(cl.ctypes.data_as(c_void_p), n, capacity, MAX_PAIR_SEG if pairing else 0,
                              slot0.ctypes.data_as(c_void_p),
                              tile_first.ctypes.data_as(c_void_p), order.ctypes.data_as(c_void_p),
                              nsub.ctypes.data_as(c_void_p))
        if nt <= 0:
            raise ValueError(f"glr_plan_tiles failed ({nt})")
        singles = np.zeros(nt, dtype=np.int32)
        pairs = np.zeros(nt, dtype=np.int32)
        alls = np.zeros(nt, dtype=np.int32)
        counts = np.zeros(3, dtype=np.int32)
        rc = L.glr_plan_items(nsub.ctypes.data_as(c_void_p), tile_first.ctypes.data_as(c_void_p), nt,
                              1 if pairing else 0, MAX_PAIR_SEG,
                              singles.ctypes.data_as(c_void_p), pairs.ctypes.data_as(c_void_p),
                              alls.ctypes.data_as(c_void_p), counts.ctypes.data_as(c_void_p))
        if rc != 0:
            raise ValueError(f"glr_plan_items failed ({rc})")
        self.n_single, self.n_pair, self.n_all = (int(c) for c in counts)
        # pairs that are the two tiles of ONE 65..128-word sentence: a prefix of the pair list (multi-tile sentences
        # are planned first); they run the 8-wave pair kernel, the rest one workgroup per tile
        is_long = nsub[pairs[:self.n_pair]] == 2
        self.n_long_pair = int(is_long.sum())
        if not bool(is_long[:self.n_long_pair].all()):
            raise RuntimeError("tile plan: long pairs are expected to lead the pair list")
        # run boundaries per tile and lane half for the forward pair kernel (full-width tiles only)
        flags = np.zeros(nt * 8, dtype=np.uint32)
        if self.n_pair:
            rc = L.glr_plan_rowflags(cl.ctypes.data_as(c_void_p), slot0.ctypes.data_as(c_void_p),
                                     tile_first.ctypes.data_as(c_void_p), order.ctypes.data_as(c_void_p),
                                     nsub.ctypes.data_as(c_void_p), nt, capacity, flags.ctypes.data_as(c_void_p))
            if rc != 0:
                raise ValueError(f"glr_plan_rowflags failed ({rc})")
        self.rowflags_host = flags.reshape(nt, 8)
        # one 256-byte descriptor per forward pair (sentences + row flags): what a workgroup reads about its pair
        desc = np.zeros(max(self.n_pair, 1) * 64, dtype=np.int32)
        if self.n_pair:
            rc = L.glr_plan_pair_desc(cl.ctypes.data_as(c_void_p), slot0.ctypes.data_as(c_void_p),
                                      tile_first.ctypes.data_as(c_void_p), order.ctypes.data_as(c_void_p),
                                      nsub.ctypes.data_as(c_void_p), nt, capacity, pairs.ctypes.data_as(c_void_p),
                                      self.n_pair, desc.ctypes.data_as(c_void_p))
            if rc != 0:
                raise ValueError(f"glr_plan_pair_desc failed ({rc})")
        self.capacity = capacity
        self.cap_lens_host = cl
        self.n_sent, self.n_tiles, self.n_slots = n, nt, nt * TILE_WORDS
        self.sent_slot0_host = slot0
        self.n_words = int(cl.sum())
        head = n + n + (nt + 1) + bound + nt + self.n_single + self.n_pair + self.n_all
        pad = np.zeros((-head) % 64, dtype=np.int32)     # the descriptors start on a 256-byte boundary (16-byte LDS-DMA pieces)
        pack = np.concatenate([cl, slot0, tile_first[: nt + 1], order, nsub[:nt], singles[:self.n_single],
                               pairs[:self.n_pair], alls[:self.n_all], pad, desc[:self.n_pair * 64]]).astype(np.int32)
        dev = upload(pack, device)
        o = 0
        self.cap_lens = dev[o:o + n]; o += n
        self.sent_slot0 = dev[o:o + n]; o += n
        self.tile_first = dev[o:o + nt + 1]; o += nt + 1
        self.order = dev[o:o + bound]; o += bound
        self.tile_nsub = dev[o:o + nt]; o += nt
        self.single_tile = dev[o:o + self.n_single]; o += self.n_single
        self.pair_tile = dev[o:o + self.n_pair]; o += self.n_pair
        self.all_tile = dev[o:o + self.n_all]; o += self.n_all + len(pad)
        self.pair_desc = dev[o:o + self.n_pair * 64] if self.n_pair else None
        self._dev = dev
        self._word_index = None

    def word_index(self, device):
        """(sentence, word, slot) of every packed word, as device int64 tensors (for scattering the
        packed word gradient back to [B, D, L])."""
        if self._word_index is None:
            si = np.repeat(np.arange(self.n_sent), self.cap_lens_host)
            wi = np.concatenate([np.arange(n) for n in self.cap_lens_host])
            slot = self.sent_slot0_host[si].astype(np.int64) + (wi // self.capacity) * TILE_WORDS + wi % self.capacity
            self._word_index = tuple(upload(a.astype(np.int64), device) for a in (si, wi, slot))
        return self._word_index

    def attn_offsets(self, s_out, device):
        off = np.zeros(self.n_sent + 1, dtype=np.int64)
        np.cumsum(self.cap_lens_host.astype(np.int64) * s_out, out=off[1:])
        return upload(off[:-1].copy(), device), off
