"""MIOpen settings of the training step (the ResNet-50 convolutions stay MIOpen calls).

Find mode (`torch.backends.cudnn.benchmark = True`) picks the fastest solver per convolution by running every
applicable one in each new process.  Measured on fresh boxes (profiles/r03_miopen_probe.txt), first training step:

  * as shipped in rounds 1-2 (user find-db + perf-db, naive solvers on)     110 - 113 s
  * naive reference solvers off, NO perf-db                                   252 s
  * perf-db + naive reference solvers off (this module)                       see bench.py `first_step_s`

What the two measurements say.  (1) The user FIND-db never short-cuts the search: MIOpen trusts a find-db record only if
its solvers already have invokers in the current process, logs "Find-db regenerating" for every convolution of a new
process and runs the candidates again - rounds 1-2's `miopen_find_db: true` claim was wrong, and 108 of the 110 s were
568 launches of the `ConvDirectNaiveConv*` reference solvers (double accumulation, 0.14 - 0.40 s per call at 256 images),
which never win; MIOpen's own switches take them out of the candidate list.  (2) The user PERF-db does matter: torch
passes `exhaustiveSearch = benchmark`, so without tuned parameters for the two CK implicit-GEMM solvers MIOpen TUNES
them ("Starting search: ConvHipImplicitGemmGroup*Xdlops") - minutes.  The shipped `miopen_db/*.udb.txt` holds those
parameters for the bench's convolutions at 32 / 64 / 128 / 256 images per GPU in bf16 and - added at the end of round 3 -
for the fp32 configuration (BASELINE config 1: 64 images, fp32): without them its first training step tuned for 204 s
(180 s of CK backward-data / weight-gradient candidates in `gpurun_out/fp32prof`), with them it takes 5.7 s."""

import glob
import os
import shutil

_HERE = os.path.dirname(os.path.abspath(__file__))
DB_DIR = os.path.normpath(os.path.join(_HERE, "..", "miopen_db"))
_NAIVE = ("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
          "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW")


def activate():
    """Call before the first convolution.  Returns True if find (benchmark) mode should be used."""
    if os.environ.get("GLR_MIOPEN_NAIVE", "0") != "1":
        for k in _NAIVE:
            os.environ.setdefault(k, "0")
    if not os.environ.get("MIOPEN_USER_DB_PATH"):
        files = glob.glob(os.path.join(DB_DIR, "*.udb.txt"))
        if files:
            # MIOpen appends to its user db: work on a per-process copy so concurrent ranks never share a file
            work = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"glr_miopen_db_{os.getpid()}")
            os.makedirs(work, exist_ok=True)
            for f in glob.glob(os.path.join(DB_DIR, "*")):
                shutil.copy(f, work)
            os.environ["MIOPEN_USER_DB_PATH"] = work
    return os.environ.get("GLR_MIOPEN_BENCHMARK", "1") != "0"
