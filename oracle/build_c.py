"""Compile the C restatement (oracle/ba_oracle.c) into oracle/_build/liboracle.so with gcc + OpenMP.
Test infrastructure: building the checker is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ba_oracle.c")
LIB = os.path.join(HERE, "_build", "liboracle.so")


def build(force=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    # no -ffast-math: the oracle's arithmetic must stay IEEE; -ffp-contract=off keeps a*b+c as written
    subprocess.run(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"],
                   check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
