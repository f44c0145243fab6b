"""Builds libevk.so (gfx950 only) in-tree with hipcc.  `python -m event_utils_amd.csrc.build [--force]`.
hipcc cross-compiles without a GPU; the .so travels to the GPU box with the repo snapshot.

Every .hip file is compiled to its own object (in parallel, cached under _obj/ by the modification times of the file and
of every header) and the objects are linked: a change to one kernel file rebuilds that file only, and an A/B build
(`variant(...)`, tools/ab_build.sh) recompiles just the files its -D flags concern."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libevk.so")
OBJ = os.path.join(HERE, "_obj")
# -ffp-contract=off : per-event values must round exactly like the reference's separate numpy/torch ops
# -munsafe-fp-atomics: hardware float atomics (outputs live in ordinary device memory)
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "-Wall",
          "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-ldl"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "*.hip")))


def _headers():
    return glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(HERE, "..", "..", "include", "evk.h"), os.path.abspath(__file__)]


def needs_build():
    if not os.path.isfile(LIB):
        return True
    deps = sources() + _headers()
    return any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps)


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _compile(src, obj, extra, verbose):
    cmd = [_hipcc()] + CFLAGS + list(extra) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return obj


def _objects(tag, extra, only, force, verbose):
    """Objects of every source, compiled with `extra` flags where `only` (None = everywhere) names the file; cached."""
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())
    jobs, objs = [], []
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        flagged = bool(extra) and (only is None or base in only)
        obj = os.path.join(OBJ, "%s%s.o" % (base, ("." + tag) if flagged else ""))
        objs.append(obj)
        stale = force or not os.path.isfile(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header)
        if stale or flagged:
            jobs.append((src, obj, extra if flagged else ()))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(lambda j: _compile(j[0], j[1], j[2], verbose), jobs))
    return objs


def _link(objs, out, verbose):
    cmd = [_hipcc()] + LDFLAGS + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=HERE)
    return out


def build(force=False, verbose=True, extra=()):
    if not force and not needs_build():
        return LIB
    return _link(_objects("x", tuple(extra), None, force, verbose), LIB, verbose)


def variant(name, flags, only=None, out_dir=None, verbose=False):
    """An A/B build: libevk_<name>.so with `flags` (a list of -D...) applied to the files named in `only` (base names without
    .hip; None = all files).  The other objects are the product's cached ones."""
    out_dir = out_dir or os.path.join(HERE, "..", "..", "tools", "exp")
    os.makedirs(out_dir, exist_ok=True)
    return _link(_objects(name, tuple(flags), only, False, verbose), os.path.join(out_dir, "libevk_%s.so" % name), verbose)


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
