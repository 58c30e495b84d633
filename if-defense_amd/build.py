"""Build libifd.so (HIP, gfx950 only) in-tree with hipcc.

    python if-defense_amd/build.py [--force]

The library is built into ``if-defense_amd/csrc/libifd.so`` so that it travels with the repo
snapshot to the GPU box (a JIT cache under ~/.cache would not).  hipcc cross-compiles gfx950
without a GPU, so this also runs in the CPU-only build container.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libifd.so")
# the same library with the optimiser kernels' repulsion terms on the IEEE sqrt / divide / expf expansions instead of the
# 1-ulp hardware instructions (-DIFD_EXACT_REP, knn_device.h rep_point2): built next to the default one so that a GPU test
# can hold the two against each other (tests/test_gpu_parity.py::test_exact_repulsion_build)
LIB_EXACT = os.path.join(CSRC, "libifd_exact.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I" + CSRC, "-I" + os.path.join(os.path.dirname(HERE), "include")] + os.environ.get("IFD_EXTRA_FLAGS", "").split()


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "ifd.h"))
    return hdr


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


# per-file flags.  optimize_bf.hip / onet_bf.hip (the split-precision optimiser kernels): no SLP vectorisation - it turns pairs of f32 operations into
# v_pk_*_f32 instructions, and a packed-f32 instruction whose consumer does not follow back to back occasionally delivers a wrong
# result while the other wave of its SIMD streams bf16 MFMAs (scripts/pk_mfma_coexec.hip, profiles/r05_pk_mfma_coexec.txt);
# tests/test_abi_cpu.py checks that the shipped kernels contain none.
FILE_FLAGS = {"optimize_bf.hip": ["-fno-slp-vectorize"], "onet_bf.hip": ["-fno-slp-vectorize"], "decode_bf.hip": ["-fno-slp-vectorize"]}


def _compile(src: str, force: bool, extra=(), suffix="") -> str:
    obj = os.path.splitext(src)[0] + suffix + ".o"
    if force or _stale(obj, [src] + _deps()):
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + list(extra) + ["-x", "hip", "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if r.stderr.strip():
            sys.stderr.write(r.stderr)
    return obj


STAMP = os.path.join(CSRC, ".build_flags")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.{hip,cpp} into libifd.so.  The flag set of the last build is kept in csrc/.build_flags: a
    library built with different flags (e.g. a diagnostic IFD_EXTRA_FLAGS="-DIFD_PROF" build) is always rebuilt."""
    srcs = sources()
    flags = " ".join(f for f in FLAGS if not f.startswith("-I"))      # location-independent: the tree travels to the GPU box
    if not os.path.exists(STAMP) or open(STAMP).read() != flags:
        force = True
    opt_src = os.path.join(CSRC, "optimize.hip")
    with ThreadPoolExecutor(max_workers=min(8, len(srcs) + 1)) as ex:
        exact = ex.submit(_compile, opt_src, force, ("-DIFD_EXACT_REP",), "_exact")
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
        exact_obj = exact.result()
    for lib, olist in ((LIB, objs), (LIB_EXACT, [exact_obj if o == os.path.splitext(opt_src)[0] + ".o" else o for o in objs])):
        if force or _stale(lib, olist):
            cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + olist
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(STAMP, "w") as f:
        f.write(flags)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
