"""Build recipe for the gfx950 shared library (hipcc, in-tree, no JIT cache).

    python -m ffb6d_amd.build            # compile ffb6d_amd/lib/libffb6d_amd.so

hipcc cross-compiles for gfx950 without a GPU; the built .so is git-ignored but travels
with the working tree to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libffb6d_amd.so")

SOURCES = ["errors.hip", "knn.hip", "knn_pruned.hip", "neighbour_ops.hip", "pose.hip",
           "knn_pick.hip", "mlp_pm.hip", "ops_pm.hip", "inputs.hip", "holefill.hip"]

# -ffp-contract=off: the KNN distance and the position encoding must round every product
# and sum separately to stay bit-identical with the reference's CPU arithmetic
# (nanoflann.hpp:323-348); kernels that want an FMA call fmaf() explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or put /opt/rocm/bin on PATH)")


def sources():
    extra = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip") and f not in SOURCES)
    return [os.path.join(CSRC, f) for f in SOURCES + extra]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    return any(os.path.getmtime(d) > t for d in deps)


def _stale(obj, src, headers):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src] + headers)


def build(force=False, verbose=True):
    """One object per source (recompiled only when the source or a header is newer), compiled in parallel, then linked."""
    if not force and not needs_build():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = os.path.join(LIB_DIR, "obj")
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", f) for f in os.listdir(os.path.join(ROOT, "include"))]
    headers.append(os.path.abspath(__file__))
    base = [hipcc_path()] + [f for f in FLAGS if f != "-shared"] + ["-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
    jobs = []
    for src in sources():
        obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, src, headers):
            jobs.append(base + ["-c", src, "-o", obj])
    if verbose and jobs:
        print("[ffb6d_amd.build] compiling", " ".join(os.path.basename(j[-3]) for j in jobs), flush=True)

    def run(cmd):
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    objs = [os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o") for src in sources()]
    subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB_PATH + ".tmp"], check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
