"""A pinned MIOpen find-db for the colour branch's dense convolutions (ResNet34 / PSPNet 3x3 and 7x7: MIOpen's kernels).

With `torch.backends.cudnn.benchmark = True` MIOpen times its candidate solvers once per process and keeps the fastest.  For half of
this network's convolutions two solvers are within 2 % of each other (CK's grouped-convolution kernels and the `igemm_fwd_gtcx35_nhwc`
assembly kernels), so a fresh machine draws a different mix every time and the step moves by +-0.4 ms with it (profiles/r05_miopen_draw_probe.txt;
round 4 saw +-0.8 ms).  MIOpen stores what it found in a *user find-db* (text files under MIOPEN_USER_DB_PATH) and re-reads it instead
of searching again; `miopen_pin/` holds the find-db (solver ranking per problem) and perf-db (tuning parameters per solver) of ONE
search on an MI355X, taken with `scripts/miopen_draw_probe.sh` -- "find once, pin the solver ids".  `use()` points MIOpen at a private
copy of it (MIOpen appends problems it has not seen), unless the caller has chosen a database directory already.

Only a benchmark / deployment convenience: results do not depend on it beyond MIOpen's own algorithm-to-algorithm rounding."""
import os
import shutil
import tempfile

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
    n = 0
    for f in os.listdir(PIN_DIR):
        if f.endswith(".txt"):
            shutil.copy(os.path.join(PIN_DIR, f), os.path.join(dst, f))
            n += 1
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return f"pinned user find-db / perf-db ({n} files of ffb6d_amd/miopen_pin: {TAG})" + (f", solver {only_solver} only" if only_solver else "")
