"""MIOpen settings of the training step (the ResNet-50 convolutions stay MIOpen calls).

Find mode (`torch.backends.cudnn.benchmark = True`) picks the fastest solver per convolution by running every
applicable one once per process.  What made that take ~110 s on a fresh box (measured, profiles/r03_miopen_probe.txt) is
not the search as such but ONE family of candidates: the `ConvDirectNaiveConv*` reference solvers (double-precision
accumulation, 0.14 s forward / 0.40 s weight-gradient per call at 256 images, 568 calls = 108 s), which never win.
MIOpen's own switches take them out of the candidate list.

A user find-db shipped with the repo (rounds 1-2) does NOT avoid the search: MIOpen only trusts a find-db record whose
solvers already have invokers in the CURRENT process ("Find-db regenerating" in its log for every convolution of a new
process), so the record is rebuilt - by the same search - each time.  The db and its `miopen_find_db` claim are gone."""

import os

_NAIVE = ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
          "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW")


def activate():
    """Call before the first convolution.  Returns True if find (benchmark) mode should be used."""
    if os.environ.get("GLR_MIOPEN_NAIVE", "0") != "1":
        for k in _NAIVE:
            os.environ.setdefault(k, "0")
    return os.environ.get("GLR_MIOPEN_BENCHMARK", "1") != "0"
