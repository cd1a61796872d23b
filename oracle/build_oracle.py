"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE -- never shipped, never measured as product).

  oracle/libpixelsynth_oracle.so   <- oracle/pixelsynth_oracle.c   (our C restatement)
  oracle/_ref/get_custom_order.so  <- /root/reference/models/lmconv/get_custom_order.c
        the reference's OWN Cython-generated C (the shipped .so is cpython-37m and cannot be
        loaded by python 3.10).  Compiled from where it lies with plain gcc; nothing is copied.
        Only built when /root/reference exists (this container); the GPU box uses the prebuilt file.

Usage: python oracle/build_oracle.py
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_C = "/root/reference/models/lmconv/get_custom_order.c"


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build_oracle(force=False):
    src = os.path.join(HERE, "pixelsynth_oracle.c")
    dst = os.path.join(HERE, "libpixelsynth_oracle.so")
    if force or _newer(src, dst):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
               src, "-o", dst, "-lm"]
        subprocess.check_call(cmd)
    return dst


def build_ref(force=False):
    """Compile the reference's get_custom_order.c in place -> oracle/_ref/ (git-ignored)."""
    out_dir = os.path.join(HERE, "_ref")
    dst = os.path.join(out_dir, "get_custom_order.so")
    if not os.path.exists(REF_C):
        return dst if os.path.exists(dst) else None
    os.makedirs(out_dir, exist_ok=True)
    if force or _newer(REF_C, dst):
        import numpy
        cmd = ["gcc", "-O2", "-shared", "-fPIC", "-w",
               "-I" + sysconfig.get_paths()["include"], "-I" + numpy.get_include(),
               REF_C, "-o", dst]
        subprocess.check_call(cmd)
    return dst


if __name__ == "__main__":
    print(build_oracle(force="--force" in sys.argv))
    print(build_ref(force="--force" in sys.argv))
