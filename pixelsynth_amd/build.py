"""Build libpixelsynth_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).

    python -m pixelsynth_amd.build [--force]

One object per translation unit, linked into pixelsynth_amd/libpixelsynth_hip.so.  Both HIP units are built
with -ffp-contract=off: splat.hip because its index paths must be bit-exact against the oracle, the lmconv*.hip units so
that the post ops inlined into different kernels (whole-grid vs column step) round identically; the matrix
products are explicit MFMA intrinsics and are not affected.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("PS_HIP_LIB") or os.path.join(HERE, "libpixelsynth_hip.so")   # (PS_HIP_LIB: tuning builds, with PS_OBJ_SUFFIX)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

UNITS = [
    ("splat.hip", ["-ffp-contract=off"]),
    ("lmconv.hip", ["-ffp-contract=off"]),
    ("lmconv_grid.hip", ["-ffp-contract=off"]),
    ("lmconv_column.hip", ["-ffp-contract=off"]),
    ("lmconv_tp.hip", ["-ffp-contract=off"]),
    ("vq.hip", ["-ffp-contract=off"]),
    ("nets.hip", ["-ffp-contract=off"]),
    ("conv_f16x3.hip", ["-ffp-contract=off", "-Wno-inline-asm"]),
    ("conv_thin.hip", ["-ffp-contract=off"]),
    ("conv1x1.hip", ["-ffp-contract=off"]),
    ("vq_ends.hip", ["-ffp-contract=off"]),
    ("host_order.cpp", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
EXTRA = os.environ.get("PS_EXTRA_HIPCC_FLAGS", "").split()  # tuning builds, e.g. -DPS_TUNING_BUILD -DPS_CHAIN_TRACE_BUILD
COMMON += EXTRA
if EXTRA:   # ps_build_info() (csrc/host_order.cpp) names them: a number measured through such a library carries its provenance
    COMMON += ['-DPS_BUILD_EXTRA_FLAGS="%s"' % " ".join(EXTRA).replace('"', "'")]


def _deps():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [
        os.path.join(os.path.dirname(HERE), "include", "pixelsynth_hip.h"),
        os.path.join(os.path.dirname(HERE), "include", "pixelsynth_hip_debug.h")]


def build(force=False, verbose=True):
    objs, todo = [], []
    dep_m = max(os.path.getmtime(d) for d in _deps())
    for src, flags in UNITS:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + os.environ.get("PS_OBJ_SUFFIX", "") + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), dep_m):
            todo.append([HIPCC, "-x", "hip", "-c", sp, "-o", obj] + COMMON + flags)
        objs.append(obj)
    if todo:   # the units are independent: compiled side by side (a full build is the longest unit, not the sum)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(todo), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(run, todo))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
