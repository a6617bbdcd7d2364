"""Dropout keys of the fused kernels (fused_ln: Philox4x32-10 key / counter prefix; fused_attn: key of its counter hash).

Eager launches: `philox_args(dev)` draws (seed, offset) from torch's CUDA generator on the host and advances the offset by
4 - reproducible under torch.manual_seed, resumable with a restored RNG state.

Launches captured in a hipGraph cannot take the key as a kernel argument (a replay would repeat the capture-time mask):
while `capturing(cell)` is active every draw returns the device cell {seed, offset base} plus a per-site offset
0, 4, 8, ... (the kernels compute key = (cell[0], cell[1] + site offset), include/glr.h `rng_cell`), and the owner of the
graph calls `GraphRng.refresh()` before each replay: the same draw-and-advance on the host, written to the cell on the
replay's stream.  A graph replay therefore consumes the generator exactly like the eager launches it replaces."""

import contextlib

import numpy as np
import torch

from .. import _native as N

_CAPTURE = None          # GraphRng of the capture in progress (single-threaded: set by capturing())


def _generator(dev):
    return torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]


class GraphRng:
    def __init__(self, dev):
        self.dev = dev
        self.cell = torch.zeros(2, dtype=torch.int64, device=dev)
        self.slots = 0               # offsets handed out during the capture (4 per dropout site)

    def refresh(self):
        gen = _generator(self.dev)
        seed, off = gen.initial_seed() & 0xFFFFFFFFFFFFFFFF, gen.get_offset()
        gen.set_offset(off + max(self.slots, 4))
        host = np.array([seed, off], dtype=np.uint64).view(np.int64)
        self.cell.copy_(N.upload(host, self.dev), non_blocking=True)


@contextlib.contextmanager
def capturing(rng):
    global _CAPTURE
    prev, _CAPTURE = _CAPTURE, rng
    try:
        yield rng
    finally:
        _CAPTURE = prev


def philox_args(dev):
    """(seed, offset, rng_cell pointer or None) for one dropout site"""
    if _CAPTURE is not None and torch.cuda.is_current_stream_capturing():
        off = _CAPTURE.slots
        _CAPTURE.slots += 4
        return 0, off, N.ptr(_CAPTURE.cell)
    gen = _generator(dev)
    seed, off = gen.initial_seed(), gen.get_offset()
    gen.set_offset(off + 4)
    return seed & 0xFFFFFFFFFFFFFFFF, off, None
