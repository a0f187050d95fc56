"""MIOpen solver selection for the conv stacks.

The ROCm image ships no gfx950 tuning database: in immediate mode (what ``torch.backends.cudnn.benchmark = False``
uses) MIOpen then picks a solver per convolution from heuristics, and a find pass over the ~200 distinct
convolutions of the train step costs 12 minutes of start-up on every fresh process.  ``ffwm_amd/miopen_db/`` holds
the result of ONE such find pass of ``python bench.py`` on an MI355X (tools/make_miopen_finddb.sh: MIOpen's own
text find-db / perf-db for gfx950 with 256 CUs, 245 KB) -- pure solver-selection data, no kernels.  With it
immediate mode selects the measured-fastest solver at no start-up cost: 63.7 -> 61.0 ms per train step.

``install()`` copies the database into a private writable directory (MIOpen appends to its user database) and
points ``MIOPEN_USER_DB_PATH`` at the copy; it does nothing when the user already set that variable or when
``FFWM_MIOPEN_DB=0``.  It must run before the first convolution of the process.
"""
import os
import shutil
import tempfile

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def install():
    if os.environ.get("FFWM_MIOPEN_DB", "1") == "0" or "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ.get("MIOPEN_USER_DB_PATH")
    if not os.path.isdir(DB_DIR):
        return None
    dst = os.path.join(tempfile.gettempdir(), "ffwm_amd_miopen_db_%d" % os.getuid())
    try:
        os.makedirs(dst, exist_ok=True)
        for name in os.listdir(DB_DIR):
            if not name.endswith(".txt"):
                continue
            src, out = os.path.join(DB_DIR, name), os.path.join(dst, name)
            if not os.path.exists(out) or os.path.getsize(out) < os.path.getsize(src):
                # (MIOpen only appends: a shorter copy is stale.)  Several ranks may race: write aside, rename atomically
                tmp = "%s.%d.tmp" % (out, os.getpid())
                shutil.copyfile(src, tmp)
                os.replace(tmp, out)
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
