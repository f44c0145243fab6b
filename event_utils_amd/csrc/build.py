"""Builds libevk.so (gfx950 only) in-tree with hipcc.  `python -m event_utils_amd.csrc.build [--force]`.
hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libevk.so")
# -ffp-contract=off : per-event values must round exactly like the reference's separate numpy/torch ops
# -munsafe-fp-atomics: hardware float atomics (outputs live in ordinary device memory)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-ldl"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def needs_build():
    if not os.path.isfile(LIB):
        return True
    deps = sources() + glob.glob(os.path.join(HERE, "*.h")) + \
        [os.path.join(HERE, "..", "..", "include", "evk.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def build(force=False, verbose=True, extra=()):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + list(extra) + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
