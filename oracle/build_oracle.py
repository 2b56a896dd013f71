"""ORACLE (test infrastructure): compile oracle/csrc/*.c into oracle/_lib/liboracle_seq.so (git-ignored; it travels to
the GPU box with the snapshot like the product's own .so).  Called by __graft_entry__.build()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "seq_gemm.c")
LIB = os.path.join(HERE, "_lib", "liboracle_seq.so")


def build(force=False):
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", "-o", LIB, SRC])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
