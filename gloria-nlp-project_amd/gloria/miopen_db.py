"""Tuned MIOpen find-db for the ResNet-50 convolutions of the bench shapes (bf16 NHWC, gfx950, the image's
MIOpen build).  With it, cudnn.benchmark (MIOpen find mode) resolves every convolution from the db instead
of searching for minutes.  `activate()` must run before the first conv."""

import glob
import os
import shutil

_HERE = os.path.dirname(os.path.abspath(__file__))
DB_DIR = os.path.normpath(os.path.join(_HERE, "..", "miopen_db"))


def covered_batches():
    """Per-GPU batch sizes the shipped find-db has entries for (field 8 of a find-db key is N)."""
    out = set()
    for f in glob.glob(os.path.join(DB_DIR, "*.ufdb.txt")):
        with open(f) as fh:
            for line in fh:
                key = line.split("=", 1)[0].split("-")
                if len(key) > 8 and key[7].isdigit():
                    out.add(int(key[7]))
    return out


def activate(batch=None):
    """Point MIOpen's user db at a private copy of the shipped db.  Returns True if benchmark (find) mode
    should be used: only when the db covers convolutions at per-GPU batch `batch` - a shape missing
    from the db would make find mode search for minutes inside the first steps."""
    if os.environ.get("GLR_MIOPEN_BENCHMARK") == "0":
        return False
    if os.environ.get("MIOPEN_USER_DB_PATH"):                      # caller manages the db (e.g. db collection runs)
        return os.environ.get("GLR_MIOPEN_BENCHMARK") == "1"
    files = glob.glob(os.path.join(DB_DIR, "*.ufdb.txt"))
    if not files or (batch is not None and int(batch) not in covered_batches()):
        return os.environ.get("GLR_MIOPEN_BENCHMARK") == "1"
    # MIOpen appends to its user db: work on a per-process copy so concurrent ranks never share a file
    work = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"glr_miopen_db_{os.getpid()}")
    os.makedirs(work, exist_ok=True)
    for f in glob.glob(os.path.join(DB_DIR, "*")):
        shutil.copy(f, work)
    os.environ["MIOPEN_USER_DB_PATH"] = work
    return True
