"""Tuned MIOpen find-db for the ResNet-50 convolutions of the bench shapes (per-GPU batch 256/128/64/32,
bf16 NHWC, gfx950, the image's MIOpen build).  With it, cudnn.benchmark (MIOpen find mode) resolves every
convolution from the db instead of searching for minutes.  `activate()` must run before the first conv."""

import glob
import os
import shutil

_HERE = os.path.dirname(os.path.abspath(__file__))
DB_DIR = os.path.normpath(os.path.join(_HERE, "..", "miopen_db"))


def activate():
    """Point MIOpen's user db at a private copy of the shipped db.  Returns True if benchmark mode should be used."""
    if os.environ.get("GLR_MIOPEN_BENCHMARK") == "0":
        return False
    if os.environ.get("MIOPEN_USER_DB_PATH"):                      # caller manages the db (e.g. db collection runs)
        return os.environ.get("GLR_MIOPEN_BENCHMARK") == "1"
    files = glob.glob(os.path.join(DB_DIR, "*.ufdb.txt"))
    if not files:
        return os.environ.get("GLR_MIOPEN_BENCHMARK") == "1"
    # MIOpen appends to its user db: work on a per-process copy so concurrent ranks never share a file
    work = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"glr_miopen_db_{os.getpid()}")
    os.makedirs(work, exist_ok=True)
    for f in glob.glob(os.path.join(DB_DIR, "*")):
        shutil.copy(f, work)
    os.environ["MIOPEN_USER_DB_PATH"] = work
    return True
