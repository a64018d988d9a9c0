"""Builds tests/hostsim/libhostsim.so: per-thread kernel bodies of ffb6d_amd/csrc/*_body.h compiled for the HOST
(`hipcc --cuda-host-only`), so CPU tests can execute the source the GPU runs.  Test infrastructure only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ffb6d_amd", "csrc")
SRC = os.path.join(HERE, "hostsim.hip")
LIB = os.path.join(HERE, "libhostsim.so")


def build():
    from ffb6d_amd.build import hipcc_path
    deps = [SRC] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith("_body.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    # -ffp-contract=off as the product build: the host must round every product and sum like the device does
    cmd = [hipcc_path(), "-x", "hip", "--cuda-host-only", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-I" + CSRC, SRC, "-o", LIB + ".tmp"]
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB
