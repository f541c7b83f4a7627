"""Build the C part of the oracle (gcc): oracle/_build/libvqoracle.so.  Checker only -- never shipped or
linked by the product path."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libvqoracle.so")


def build():
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(HERE, "vq_argmin.c")
    if os.path.exists(LIB) and os.path.getmtime(LIB) > os.path.getmtime(src):
        return LIB
    # -ffp-contract=off: every fmaf is explicit, nothing else may be fused
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-o", LIB, src, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build())
