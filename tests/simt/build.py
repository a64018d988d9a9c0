"""Builds tests/simt/libsimt_ffb6d.so: kernels of the product library (csrc/mlp_pm.hip: the point-major GEMM family and the fused
attentive pooling; csrc/upconv.hip and csrc/posenc.hip with their launchers) compiled for the HOST against the SIMT emulator (tests/simt/simt.*, fake/hip/hip_runtime.h).
The kernel sources are used as they are, except for mechanical substitutions made on a scratch copy:
  * the declaration of the dynamic shared array becomes a pointer to the emulator's buffer; `kernel<<<...>>>(...)` launches are
    rewritten into the hipLaunchKernelGGL macro (which the fake runtime maps to the emulator);
  * where a wave reads LDS data that OTHER lanes of the same wave wrote without any instruction in between that the emulator
    treats as a rendezvous (hardware runs a wave in lock step, fibers do not), a `simt::wave_sync()` is inserted: one place,
    between the epilogue of mlp_pm_stream_kernel (lanes deposit result rows in the wave's LDS image) and the whole-row stores.
Test infrastructure only."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "ffb6d_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libsimt_ffb6d.so")
KERNEL_SOURCES = ["errors.hip", "seg_sort.hip", "mlp_pm.hip", "mlp_pm_big.hip", "mlp_chain.hip", "lfa_pm.hip", "train_ops.hip", "train_rows.hip", "upconv.hip", "posenc.hip", "ops_pm.hip", "neighbour_ops.hip", "knn.hip", "knn_pruned.hip", "knn_pick.hip", "pose.hip", "inputs.hip", "holefill.hip"]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"

# statements after which a wave relies on lock-step execution for LDS traffic between its lanes
LOCKSTEP_AFTER = {"mlp_pm.hip": ["stream_epilogue<T, TM, LSM>(p, acc, img, OS, r0, l31, kh, bias_lds, HASY ? yreg : nullptr);"]}
DYN_SHARED = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(16\)\)\)\s+)?((?:unsigned )?\w+)\s+(\w+)\[\];")


def _split_top(text):
    parts, depth, cur = [], 0, ""
    for ch in text:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            parts.append(cur.strip())
            cur = ""
        else:
            cur += ch
    parts.append(cur.strip())
    return parts


def chevrons_to_launch_macro(src):
    """kernel<T><<<grid, block, shmem, stream>>>(args)  ->  hipLaunchKernelGGL((kernel<T>), dim3(grid), dim3(block), shmem, stream, args)"""
    out, pos = "", 0
    while True:
        i = src.find("<<<", pos)
        if i < 0:
            return out + src[pos:]
        j = src.index(">>>", i)
        k = i                                    # kernel expression: identifier [+ one template argument list] before <<<
        if src[k - 1] == ">":
            depth = 0
            while True:
                k -= 1
                depth += {">": 1, "<": -1}.get(src[k], 0)
                if depth == 0:
                    break
        while k > 0 and (src[k - 1].isalnum() or src[k - 1] in "_:"):
            k -= 1
        cfg = _split_top(src[i + 3:j]) + ["0", "0"]
        a = src.index("(", j)
        depth, e = 0, a
        while True:
            depth += {"(": 1, ")": -1}.get(src[e], 0)
            if depth == 0:
                break
            e += 1
        out += src[pos:k] + "hipLaunchKernelGGL((%s), dim3(%s), dim3(%s), %s, %s, %s)" % (
            src[k:i], cfg[0], cfg[1], cfg[2], cfg[3], src[a + 1:e])
        pos = e + 1


ASM_BLOCKS = {}      # (pattern, replacement) per source with inline assembly: none left since csrc/shared_mlp.hip was removed


def transformed(name):
    with open(os.path.join(CSRC, name)) as fh:
        src = fh.read()
    if name in ASM_BLOCKS:
        pat, repl = ASM_BLOCKS[name]
        src, n = pat.subn(repl, src)
        if n != 1:
            raise RuntimeError(f"{name}: inline-assembly block not found exactly once")
    src = chevrons_to_launch_macro(src)
    src = DYN_SHARED.sub(r"\1* \2 = reinterpret_cast<\1*>(simt::dyn_shared());", src)
    for anchor in LOCKSTEP_AFTER.get(name, []):
        if src.count(anchor) != 1:
            raise RuntimeError(f"{name}: lock-step anchor not found exactly once: {anchor}")
        src = src.replace(anchor, anchor + "\n        simt::wave_sync();      // inserted by tests/simt/build.py")
    dst = os.path.join(OUT, name.replace(".hip", ".simt.cpp"))
    with open(dst, "w") as fh:
        fh.write(src)
    return dst


def build():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, f) for f in ("simt.cpp", "simt.h", "build.py")] + \
        [os.path.join(HERE, "fake", "hip", "hip_runtime.h"), os.path.join(HERE, "fake", "rocprim", "rocprim.hpp")]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(OUT, exist_ok=True)
    srcs = [transformed(n) for n in KERNEL_SOURCES] + [os.path.join(HERE, "simt.cpp")]
    cmd = [CLANG, "-x", "c++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unused-value", "-Wno-psabi",
           "-Wno-unknown-attributes", "-I" + os.path.join(HERE, "fake"), "-I" + HERE, "-I" + os.path.join(ROOT, "include"),
           "-I" + CSRC] + srcs + ["-o", LIB + ".tmp"]
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build())
