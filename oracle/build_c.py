"""Compile the C restatement (oracle/ba_oracle.c) into oracle/_build/liboracle.so with gcc + OpenMP.
Test infrastructure: building the checker is not using it."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "ba_oracle.c")
LIB = os.path.join(HERE, "_build", "liboracle.so")


VARIANTS = {
    # same source, every camera's observation list walked backwards in the sums B_c, g_c and the rows of S: an equally
    # legitimate summation order - used to measure how far the oracle determines ITSELF on ill-conditioned objectives
    "reverse_sums": ("liboracle_rev.so", ["-DBAO_REVERSE_SUMS"]),
}


def build(force=False, variant=None):
    lib, extra = (LIB, []) if variant is None else (os.path.join(HERE, "_build", VARIANTS[variant][0]), VARIANTS[variant][1])
    if not force and os.path.exists(lib) and os.path.getmtime(lib) >= os.path.getmtime(SRC):
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    # no -ffast-math: the oracle's arithmetic must stay IEEE; -ffp-contract=off keeps a*b+c as written
    subprocess.run(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", "-fPIC", "-shared"] + extra +
                   ["-o", lib, SRC, "-lm"], check=True)
    return lib


if __name__ == "__main__":
    print(build(force=True))
    for v in VARIANTS:
        print(build(force=True, variant=v))
