"""A pinned MIOpen find-db for the colour branch's dense convolutions (ResNet34 / PSPNet 3x3 and 7x7: MIOpen's kernels).

With `torch.backends.cudnn.benchmark = True` MIOpen times its candidate solvers once per process and keeps the fastest.  For half of
this network's convolutions two solvers are within 2 % of each other (CK's grouped-convolution kernels and the `igemm_fwd_gtcx35_nhwc`
assembly kernels), so a fresh machine draws a different mix every time and the step moves by +-0.4 ms with it (profiles/r05_miopen_draw_probe.txt;
round 4 saw +-0.8 ms).  MIOpen stores what it found in a *user find-db* (text files under MIOPEN_USER_DB_PATH) and re-reads it instead
of searching again; `miopen_pin/` holds the find-db (solver ranking per problem) and perf-db (tuning parameters per solver) of ONE
search on an MI355X, taken with `scripts/miopen_draw_probe.sh` -- "find once, pin the solver ids".  `use()` points MIOpen at a private
copy of it (MIOpen appends problems it has not seen), unless the caller has chosen a database directory already.

Only a benchmark / deployment convenience: results do not depend on it beyond MIOpen's own algorithm-to-algorithm rounding."""
import atexit
import os
import re
import shutil
import tempfile
import warnings

PIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_pin")
TAG = "r05 search on MI355X (gfx950, 256 CUs), MIOpen 3.5.0: fp32 NHWC forward problems of the bs = 8, 480 x 640 colour branch"


FWD_SOLVER = "ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC"


def use(rank=0, only_solver=None):
    """Call before the first convolution of the process.  Returns a short description for bench records.
    only_solver: restrict MIOpen's search to one solver (MIOPEN_DEBUG_FIND_ONLY_SOLVER) -- forward-only processes; a fresh machine has no
    compiled kernels in its cache, so MIOpen discards the pinned find-db records and searches again: the ranking between near-tied
    solvers is then redrawn, while the pinned perf-db still fixes each solver's kernel variant."""
    if only_solver and not os.environ.get("MIOPEN_DEBUG_FIND_ONLY_SOLVER"):
        os.environ["MIOPEN_DEBUG_FIND_ONLY_SOLVER"] = only_solver
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return "MIOPEN_USER_DB_PATH set by the caller: " + os.environ["MIOPEN_USER_DB_PATH"]
    dst = os.path.join(tempfile.gettempdir(), f"ffb6d_miopen_pin_{os.getpid()}_{rank}")
    os.makedirs(dst, exist_ok=True)
    atexit.register(shutil.rmtree, dst, ignore_errors=True)          # the private copy goes with the process
    n = 0
    for f in os.listdir(PIN_DIR):
        if f.endswith(".txt"):
            shutil.copy(os.path.join(PIN_DIR, f), os.path.join(dst, f))
            n += 1
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    note = version_note()
    if note:
        warnings.warn(note)
    return f"pinned user find-db / perf-db ({n} files of ffb6d_amd/miopen_pin: {TAG})" + (f", solver {only_solver} only" if only_solver else "") + \
        (f"; {note}" if note else "")


def pinned_version():
    """(major, minor, patch) of the MIOpen build the pinned files were written by (it is part of their names), or None"""
    for f in os.listdir(PIN_DIR):
        m = re.search(r"\.HIP\.(\d+)_(\d+)_(\d+)_", f)
        if m:
            return tuple(int(g) for g in m.groups())
    return None


def version_note():
    """MIOpen reads user databases only under its own version's file name: with another MIOpen build the pin is silently ignored and every
    process searches again (the step then moves by +-0.4 ms with the draw, like `--miopen-db fresh`).  Returns a note in that case."""
    try:
        import torch
        v = int(torch.backends.cudnn.version() or 0)
    except Exception:      # pragma: no cover
        return ""
    have, want = (v // 1000000, v // 1000 % 1000, v % 1000), pinned_version()
    if want and v and have != want:
        return (f"installed MIOpen {have[0]}.{have[1]}.{have[2]} differs from the pinned database's {want[0]}.{want[1]}.{want[2]}: "
                "the pin is not read, MIOpen searches afresh")
    return ""
