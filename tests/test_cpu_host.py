"""CPU (no GPU): the C-ABI library loads and exports every symbol include/evk.h declares; host-side logic; the
product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "evk.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(evk_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from event_utils_amd.csrc import build
    build.build(verbose=False)
    from event_utils_amd import _lib
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libevk.so does not export %s" % n
    bound = set(_lib.SIGNATURES) | set(_lib._SPECIAL)
    assert bound == set(names), "ctypes binding and header disagree: %s" % (bound ^ set(names))
    assert _lib.lib().evk_version() == 100
    assert _lib.lib().evk_error_string(-1) == b"invalid argument"


def test_python_flag_constants_equal_the_header():
    """event_utils_amd/_lib.py restates the flags of include/evk.h: every EVK_* integer constant there must be the header's
    #define of the same name (the ctypes binding is the reference-side stub of INTEGRATION.md: it may not drift)."""
    import re
    from event_utils_amd import _lib
    text = open(os.path.join(ROOT, "include", "evk.h")).read()
    defines = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"^#define\s+(EVK_[A-Z0-9_]+)\s+(-?(?:0x[0-9a-fA-F]+|\d+))[uU]?\b", text, re.M)}
    names = [k for k, v in vars(_lib).items() if k.startswith("EVK_") and isinstance(v, int)]
    assert len(names) >= 15
    for k in names:
        assert k in defines and defines[k] == getattr(_lib, k), (k, getattr(_lib, k), defines.get(k))


def test_argument_errors_need_no_gpu():
    from event_utils_amd import _lib
    L = _lib.lib()
    # n < 0 and null output are rejected before anything touches the device
    assert L.evk_voxel_f32(None, None, None, None, -1, 0.0, 1.0, 5, 4, 4, None, None, None) == -1
    assert L.evk_image_nearest_i32(None, None, None, 0, 4, 4, None, None, None) == -1
    assert L.evk_variance_f32(None, 10, None, None, 0, None) == -1
    with pytest.raises(_lib.EvkError):
        _lib.check(-2, "x")


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_product_fails_loudly_without_gpu():
    import event_utils_amd as E
    from event_utils_amd._lib import EvkError
    x = np.array([1, 2]); p = np.array([1, 1])
    with pytest.raises(EvkError):
        E.events_to_image(x, x, p)
    with pytest.raises(EvkError):
        E.events_to_voxel_torch(torch.ones(3), torch.ones(3), torch.arange(3.), torch.ones(3), 2)
    with pytest.raises(EvkError):
        E.get_iwe((0., 0.), x * 1.0, x * 1.0, x * 1.0, p * 1.0, E.linvel_warp(), (180, 240))


def test_product_never_imports_oracle():
    import subprocess
    import sys
    code = "import sys, event_utils_amd, event_utils_amd.lib; assert not any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules), 'oracle imported'"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "event_utils_amd")):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in s and "from oracle" not in s, f


def test_reference_api_surface():
    """Names, positional order and defaults of the reference's public functions."""
    import inspect
    import event_utils_amd as E
    from event_utils_amd.contrast_max import events_cmax

    def sig(f):
        return [(p.name, p.default) for p in inspect.signature(f).parameters.values()]
    E_ = inspect.Parameter.empty
    assert sig(E.events_to_image) == [("xs", E_), ("ys", E_), ("ps", E_), ("sensor_size", (180, 240)),
                                      ("interpolation", None), ("padding", False), ("meanval", False), ("default", 0)]
    assert sig(E.events_to_image_torch) == [("xs", E_), ("ys", E_), ("ps", E_), ("device", None),
                                            ("sensor_size", (180, 240)), ("clip_out_of_range", True),
                                            ("interpolation", None), ("padding", True), ("default", 0)]
    assert sig(E.events_to_voxel) == [("xs", E_), ("ys", E_), ("ts", E_), ("ps", E_), ("B", E_),
                                      ("sensor_size", (180, 240)), ("temporal_bilinear", True)]
    assert sig(E.events_to_voxel_torch) == [("xs", E_), ("ys", E_), ("ts", E_), ("ps", E_), ("B", E_), ("device", None),
                                            ("sensor_size", (180, 240)), ("temporal_bilinear", True)]
    assert sig(E.get_iwe)[:11] == [("params", E_), ("xs", E_), ("ys", E_), ("ts", E_), ("ps", E_), ("warpfunc", E_),
                                   ("img_size", E_), ("compute_gradient", False), ("use_polarity", True),
                                   ("return_events", False), ("return_per_event_contrast", False)]
    assert sig(E.optimize) == [("xs", E_), ("ys", E_), ("ts", E_), ("ps", E_), ("warp", E_), ("obj", E_),
                               ("numeric_grads", True), ("img_size", (180, 240))]
    s = sig(events_cmax.optimize_contrast)
    assert [n for n, _ in s] == ["xs", "ys", "ts", "ps", "warp_function", "objective", "optimizer", "x0",
                                 "numeric_grads", "blur_sigma", "img_size", "grid_search_init", "minimum_events"]
    assert sig(E.linvel_warp.warp)[1:] == [("xs", E_), ("ys", E_), ("ts", E_), ("ps", E_), ("t0", E_), ("params", E_),
                                           ("compute_grad", False)]
    w, o = E.linvel_warp(), E.variance_objective()
    assert (w.name, w.dims) == ("linvel_warp", 2)
    assert (o.name, o.use_polarity, o.has_derivative, o.default_blur, o.adaptive_lifespan, o.pixel_crossings,
            o.minimum_events) == ("variance", True, True, 1.0, False, 5, 10000)
    o.iter_update(np.array([3.0, 4.0]))
    assert o.lifespan == 1.0 and o.recompute_lifespan
    o.iter_update(np.array([0, 0]))
    assert o.lifespan == 5


def test_gaussian_kernel_matches_oracle_and_scipy():
    from event_utils_amd.contrast_max.objectives import gaussian_kernel1d
    from oracle.reference_np import gaussian_kernel1d as ok
    for s in (0.5, 1.0, 2.0, 3.3):
        w, r = gaussian_kernel1d(s)
        w2, r2 = ok(s)
        assert r == r2 and np.array_equal(w, w2)


def test_f32_lossless_policy():
    from event_utils_amd.events import _f32_lossless
    assert _f32_lossless(np.arange(10))
    assert _f32_lossless(np.array([0.5, 0.25, 3.0]))
    assert not _f32_lossless(np.array([0.1]))
    assert _f32_lossless(np.array([0.1], dtype=np.float32))


def test_numeric_gradient_formula_equals_scipy_internal_forward_differences(golden):
    """evaluate_numeric_gradient must reproduce what fmin_bfgs(..., epsilon=1) estimates internally (the reference's
    default path, events_cmax.py:343).  No GPU: the product objective's evaluate_function is replaced by the oracle's."""
    import warnings
    import scipy.optimize as opt
    import event_utils_amd as E
    from oracle import reference_np as R
    g = golden("f8_objective")
    x, y, t, p = (np.asarray(g[k], dtype=np.float64) for k in ("xs", "ys", "ts", "ps"))
    robj, rw = R.variance_objective(), R.linvel_warp()
    obj = E.variance_objective()
    obj.evaluate_function = lambda prm, *a, **k: robj.evaluate_function(prm, x, y, t, p, rw, (180, 240), 1.0)
    args = (None, None, None, None, object(), (180, 240), 1.0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = opt.fmin_bfgs(obj.evaluate_function, np.array([0, 0]), args=args, epsilon=1, disp=False)
        b = opt.fmin_bfgs(obj.evaluate_function, np.array([0, 0]), fprime=obj.evaluate_numeric_gradient, args=args,
                          disp=False)
    assert np.allclose(a, g_numeric_argmax(golden), atol=1e-9)
    assert np.allclose(a, b, atol=1e-9)
    # the value that comes with the gradient is f(x) itself (one of the three forward-difference evaluations)
    x0 = np.array([12.0, -7.0])
    fv, gv = obj.evaluate_function_and_numeric_gradient(x0, *args)
    assert fv == obj.evaluate_function(x0, *args) and np.array_equal(gv, obj.evaluate_numeric_gradient(x0, *args))


def g_numeric_argmax(golden):
    return golden("f9_optimize_trace")["numeric_argmax"]


class _QuadraticObjective:
    """Duck-typed plugin objective (no GPU): f(v) = |v - v*|^2 - 1e4, minimum at v* = (40, -25)."""
    has_derivative = False

    def evaluate_function(self, params=None, xs=None, ys=None, ts=None, ps=None, warpfunc=None, img_size=None,
                          blur_sigma=None, showimg=False, iwe=None):
        return float((params[0] - 40.0) ** 2 + (params[1] + 25.0) ** 2 - 1e4)


class _PluginWarp:
    name, dims = "plugin", 2


def test_parameter_space_samplers_host_logic(golden):
    """grid_search_initial / find_new_range / grid_search_optimisation / objective_landscape / cut_events_to_lifespan /
    segmentation_mask_from_d_iwe: sample positions, orders and selections pinned to what the reference produced
    (tests/golden/f14_search.npz); objective values come from a plugin objective so no GPU is needed."""
    import inspect
    from event_utils_amd.contrast_max import events_cmax as C
    from event_utils_amd.contrast_max.objectives import cut_events_to_lifespan
    g, g8 = golden("f14_search"), golden("f8_objective")
    obj, w = _QuadraticObjective(), _PluginWarp()
    r = C.grid_search_initial(None, None, None, None, w, obj, (180, 240))
    assert np.array_equal(np.array(r["search_axes"]), g["gsi_log5_axes"])
    assert np.array_equal(np.array(r["params"]), g["gsi_log5_params"])
    assert r["min_params"] == (15.0, -15.0) and r["min_func_eval"] == obj.evaluate_function((15.0, -15.0))
    r = C.grid_search_initial(None, None, None, None, w, obj, (180, 240), log_scale=False, num_samples_per_param=7,
                              param_ranges=[[-60, 60], [-90, 30]])
    assert np.array_equal(np.array(r["search_axes"]), g["gsi_lin7_axes"])
    assert np.array_equal(np.array(r["params"]), g["gsi_lin7_params"])
    for q, rng in zip(g["fnr_params"], g["fnr_ranges"]):
        assert np.array_equal(np.array(C.find_new_range(g["fnr_axes"], q)), rng)
    r = C.grid_search_optimisation(None, None, None, None, w, obj, (180, 240), log_scale=False)
    assert np.allclose(r["min_params"], (40.0, -25.0), atol=0.5)
    assert C.recursive_search is C.grid_search_optimisation

    class NoMinimum(_QuadraticObjective):
        def evaluate_function(self, params=None, **k):
            return 1.0
    assert C.grid_search_initial(None, None, None, None, w, NoMinimum(), (180, 240))["min_params"] is None

    a = g["landscape_args"]
    img = C.objective_landscape(None, None, None, None, obj, w, x_range=(a[0], a[1]), y_range=(a[2], a[3]),
                                resolution=a[4])
    assert img.shape == g["landscape"].shape and img.min() == 0.0 and abs(img.max() - 1.0) < 1e-6
    raw = np.array([[-obj.evaluate_function((x * 20 - 100, y * 20 - 80)) for x in range(10)] for y in range(7)])
    assert np.allclose(img, (raw - raw.min()) / (raw.max() - raw.min() + 1e-6), rtol=0, atol=1e-12)

    x, y, t, p = (np.asarray(g8[k], dtype=np.float64) for k in ("xs", "ys", "ts", "ps"))
    for i, prm in enumerate(([400., -250.], [4000., -2500.])):
        cut = cut_events_to_lifespan(x, y, t, p, np.array(prm), 5, minimum_events=5000)
        assert len(cut[0]) == g["cut_len"][i] and cut[2][0] == g["cut_first_t"][i]
    assert np.array_equal(C.segmentation_mask_from_d_iwe(g["seg_d_iwe"]), g["seg_mask"])
    assert np.array_equal(C.segmentation_mask_from_d_iwe(g["seg_d_iwe"], th=0.05), g["seg_mask_th"])

    names = lambda f: list(inspect.signature(f).parameters)
    assert names(C.grid_search_initial) == ["xs", "ys", "ts", "ps", "warp_function", "objective_function", "img_size",
                                            "param_ranges", "log_scale", "num_samples_per_param"]
    assert names(C.grid_search_optimisation) == names(C.grid_search_initial) + ["depth", "th0", "max_iters"]
    assert names(C.draw_objective_function) == ["xs", "ys", "ts", "ps", "objective", "warpfunc", "x_range", "y_range",
                                                "gt", "show_gt", "resolution", "img_size", "show_axes", "norm_min",
                                                "norm_max", "show"]
    assert names(C.optimize_r2) == names(C.optimize)
    assert names(C.grid_cmax) == ["xs", "ys", "ts", "ps", "roi_size", "step", "warp", "obj", "min_events"]


def test_header_is_plain_c_and_the_library_is_usable_from_c(tmp_path):
    """include/evk.h must compile as C99 on its own (no C++, no HIP, no torch types) and a C program must be able to
    bind the library through it: the drop-in boundary is a C ABI, not a Python extension."""
    import shutil
    import subprocess
    from event_utils_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "use_evk.c"
    src.write_text(r"""
#include <dlfcn.h>
#include <stdio.h>
#include "evk.h"
int main(int argc, char **argv) {
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    int (*version)(void) = (int (*)(void))dlsym(h, "evk_version");
    const char *(*errstr)(int) = (const char *(*)(int))dlsym(h, "evk_error_string");
    int (*ntiles)(int, int, int, int) = (int (*)(int, int, int, int))dlsym(h, "evk_bucket_num_tiles");
    if (!version || !errstr || !ntiles) return 3;
    /* the prototypes of the header and the symbols must agree: take the address through the declared type */
    int (*declared)(int, int, int, int) = evk_bucket_num_tiles; (void)declared;
    printf("%d|%s|%d\n", version(), errstr(EVK_EINVAL), ntiles(480, 640, 5, 4));
    return 0;
}
""")
    exe = tmp_path / "use_evk"
    lib = _lib.lib_path() if hasattr(_lib, "lib_path") else os.path.join(root, "event_utils_amd", "csrc", "libevk.so")
    inc = os.path.join(root, "include")
    hdr_only = tmp_path / "hdr.c"
    hdr_only.write_text('#include "evk.h"\nint evk_header_is_c(void) { return EVK_OK; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(hdr_only), "-o",
                    str(tmp_path / "hdr.o")], check=True, capture_output=True)
    # (dlsym's void* -> function pointer conversion is what -pedantic would object to in the program itself)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe), "-ldl",
           "-Wl,--unresolved-symbols=ignore-all"]
    subprocess.run(cmd, check=True, capture_output=True)
    out = subprocess.run([str(exe), lib], check=True, capture_output=True, text=True).stdout.strip().split("|")
    assert int(out[0]) >= 100 and out[1] and int(out[2]) == 20 * 30


def test_bench_with_more_gpus_than_the_box_has_fails_loudly():
    """`python bench.py --gpus 2` launches its own two ranks (torch.distributed.run); on a box without two GPUs it must
    die with a non-zero exit status instead of printing a one-GPU number under a two-GPU label."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--no-cmax"], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert '"metric"' not in r.stdout
    # the rank / world bookkeeping: a WORLD_SIZE that disagrees with --gpus is refused before any GPU work
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and "must agree" in (r.stdout + r.stderr)


def test_voxel_kernels_compile_without_register_spills(tmp_path):
    """The kernels the default voxel dispatch can reach -- k_part_sorted<1024, 8 | 12, 4 | 8, Src*> and k_voxel_tiles2<512, 2, *, *,
    4 | 8> -- must not spill: scratch accesses share the load / store counter (gfx9 has ONE) and stall the software pipelines.
    Cross-compiles evk_voxel2.hip for gfx950 (no GPU needed) and reads the code object's metadata."""
    import re
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.isfile(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    from event_utils_amd.csrc import build as B
    src = os.path.join(B.HERE, "evk_voxel2.hip")
    flags = list(B.CFLAGS)
    subprocess.run([hipcc] + flags + ["-c", src, "-o", str(tmp_path / "v2.o"), "-save-temps=obj"], check=True, cwd=B.HERE,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert asm, os.listdir(tmp_path)
    text = open(tmp_path / asm[0]).read()
    kernels = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    seen = {n: (int(v), int(sp)) for n, v, sp in kernels if "k_part_sorted" in n or "k_voxel_tiles2" in n or "k_voxel_live" in n}
    assert len(seen) >= 14, sorted(seen)
    spilled = {n: vs for n, vs in seen.items() if vs[1]}
    assert not spilled, spilled
    assert "Folded Spill" not in "".join(l for l in text.splitlines(True) if "scratch_" in l)
    # 8-byte records: three workgroups of 8 waves per CU; 4-byte records (HBM-resident calls, two workgroups per CU by their
    # LDS): 128 registers, three table entries per lane and batch (evk_voxel2.hip, k_voxel_tiles2)
    assert all(v <= 80 for n, (v, _) in seen.items() if "k_voxel_tiles2" in n and "ELi8EEEv" in n)
    assert all(v <= 128 for n, (v, _) in seen.items() if "k_voxel_tiles2" in n)
    assert all(v <= 128 for n, (v, _) in seen.items() if "k_part_sorted" in n)
    # the live pair must fit one CU together: the partition's 4 waves per SIMD at <= 80 registers leave 192 for the consumer's 2
    assert all(v <= 80 for n, (v, _) in seen.items() if "k_part_sorted" in n and "Lb1EEEvT2_" in n)   # (LIVE is the last template argument)
    assert all(v <= 96 for n, (v, _) in seen.items() if "k_voxel_live" in n)


def test_evk_bfgs_line_search_logic_on_a_known_function():
    """events_cmax.evk_bfgs needs no GPU: on an objective object that evaluates a tilted, badly scaled quadratic bowl plus a
    quartic term it must reach the minimiser from (0, 0) with few (value + gradient) evaluations and three-point passes, stop
    by itself, call the callback once per accepted point, and never evaluate anything after a failed line search."""
    from event_utils_amd.contrast_max.events_cmax import evk_bfgs
    A = np.array([[3.0, 0.3], [0.3, 0.05]])      # positive definite, condition number ~150
    xm = np.array([40.0, -25.0])

    class Bowl:
        def __init__(self):
            self.fg_calls, self.batch_calls = 0, 0

        def f(self, q):
            d = np.asarray(q, dtype=np.float64) - xm
            return 0.5 * d.dot(A).dot(d) * 1e-3 + 1e-7 * np.sum(d ** 4) - 2.0

        def evaluate_function_and_gradient(self, q, *a):
            self.fg_calls += 1
            d = np.asarray(q, dtype=np.float64) - xm
            return np.float32(self.f(q)), (A.dot(d) * 1e-3 + 4e-7 * d ** 3).astype(np.float32)

        def evaluate_function_and_numeric_gradient(self, q, *a):
            self.fg_calls += 1
            q = np.asarray(q, dtype=np.float64)
            f0 = self.f(q)
            return np.float32(f0), np.array([self.f(q + [1, 0]) - f0, self.f(q + [0, 1]) - f0], dtype=np.float32)

        def evaluate_function_batch(self, pts, *a):
            self.batch_calls += 1
            assert len(pts) == 3
            return [np.float32(self.f(q)) for q in pts]
    for numeric in (False, True):
        o, seen, tr = Bowl(), [], []
        x = evk_bfgs(o, np.array([0, 0]), (), numeric_grads=numeric, callback=lambda q: seen.append(np.array(q)), trace=tr)
        if numeric:     # forward differences with epsilon = 1 are a biased gradient: the search ends where no step along it helps
            assert o.f(x) - o.f(xm) < 1e-3 * (o.f(np.zeros(2)) - o.f(xm)), (x, o.f(x))
        else:           # (float32 values: the flat axis resolves ~0.1)
            assert np.linalg.norm(x - xm) < 0.2, x
        assert len(seen) == len(tr) - 1 and o.fg_calls == len(tr)
        assert o.fg_calls <= 25 and o.batch_calls <= 40
        assert all(tr[k + 1][1] <= tr[k][1] for k in range(len(tr) - 1))       # monotone: every accepted point improves f


def test_library_bfgs_loop_equals_the_python_loop_on_a_known_function():
    """evk_bfgs2_minimize (evk_optim.hip: the quasi-Newton loop of events_cmax.evk_bfgs in C, here on caller-supplied
    evaluators; the GPU entry evk_cmax_bfgs_variance_tiled_f32 is the same template on the library's own evaluation calls) is a
    host function: on the bowl of the test above, driven through ctypes callbacks, it must visit the points of the Python loop
    BIT FOR BIT -- analytic and numeric gradients, unit step first or not -- and hand a declined point and an evaluator's
    error code back."""
    import ctypes
    from event_utils_amd import _lib
    from event_utils_amd.contrast_max.events_cmax import evk_bfgs
    A = np.array([[3.0, 0.3], [0.3, 0.05]])
    xm = np.array([40.0, -25.0])

    def f(q):
        d = np.asarray(q, dtype=np.float64) - xm
        return float(np.float32(0.5 * d.dot(A).dot(d) * 1e-3 + 1e-7 * np.sum(d ** 4) - 2.0))

    def fg(q):
        d = np.asarray(q, dtype=np.float64) - xm
        return f(q), [float(v) for v in (A.dot(d) * 1e-3 + 4e-7 * d ** 3).astype(np.float32)]

    def f3(pts):
        return [f(q) for q in pts]

    class Bound:                                   # what evk_bfgs asks of an objective that can bind itself to its events
        evaluate_function_and_gradient = evaluate_function_and_numeric_gradient = None     # (never reached: the closures are used)

        def bind_fast(self, *a):
            return fg, f3
    FG = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_double))
    F3 = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double))
    limit = [None]

    def c_fg(user, xp, fp, gp):
        if limit[0] is not None and abs(xp[0]) > limit[0]:
            return 1
        fv, gv = fg([xp[0], xp[1]])
        fp[0], gp[0], gp[1] = fv, gv[0], gv[1]
        return 0

    def c_f3(user, pp, fsp):
        if limit[0] is not None and max(abs(pp[0]), abs(pp[2]), abs(pp[4])) > limit[0]:
            return 1
        for k, v in enumerate(f3([[pp[2 * k], pp[2 * k + 1]] for k in range(3)])):
            fsp[k] = v
        return 0
    cb_fg, cb_f3 = FG(c_fg), F3(c_f3)
    fn = _lib.lib().evk_bfgs2_minimize
    ptr = lambda a: ctypes.c_void_p(a.ctypes.data)  # noqa: E731

    def run_c(numeric, unit_first, fg_cb=cb_fg, f3_cb=cb_f3):
        x0 = np.zeros(2)
        opts = np.array([1e-3, 1e-5, 1e-6, 100.0, float(numeric), float(unit_first)])
        res = np.zeros(6 + 5 * 101)
        rc = fn(ctypes.cast(fg_cb, ctypes.c_void_p) if fg_cb is not None else None, ctypes.cast(f3_cb, ctypes.c_void_p), None,
                ptr(x0), ptr(opts), ptr(res), 101)
        return rc, res
    for numeric in (False, True):
        for unit_first in (True, False):
            tr = []
            xp = evk_bfgs(Bound(), np.zeros(2), (), numeric_grads=numeric, trace=tr, unit_first=unit_first, native=False)
            rc, res = run_c(numeric, unit_first, None if numeric else cb_fg)
            assert rc == 0 and res[5] == 0.0 and int(res[3]) == len(tr) and np.array_equal(res[:2], xp), (numeric, unit_first)
            rows = res[6:6 + 5 * len(tr)].reshape(-1, 5)
            for row, (q, fv, gv) in zip(rows, tr):
                assert np.array_equal(row[:2], q) and row[2] == fv and np.array_equal(row[3:], gv)
            if numeric:     # (forward differences with epsilon = 1: a biased gradient along the flat axis, as in the test above)
                assert f(xp) - f(xm) < 1e-3 * (f(np.zeros(2)) - f(xm))
            else:
                assert np.linalg.norm(xp - xm) < 0.2
    # a declined point ends the run with status 1 at the last accepted point; an evaluator's error code comes back as it is
    limit[0] = 20.0
    rc, res = run_c(False, True)
    assert rc == 0 and res[5] == 1.0 and abs(res[0]) <= 20.0 and res[4] >= 1
    limit[0] = None
    bad = FG(lambda user, xp, fp, gp: -7)
    assert run_c(False, True, bad)[0] == -7
    assert run_c(False, True, None)[0] == -1          # EVK_EINVAL: analytic gradients need fg


def test_resident_events_and_objectives_copy_without_their_call_caches():
    """Marshalled library calls cached on a DeviceEvents / an objective hold ctypes pointers (which copy.deepcopy and pickle
    refuse) into per-stream scratch: they are transient, a copy carries the state only and registers itself for
    release_scratch like any other DeviceEvents."""
    import copy
    import ctypes
    import pickle
    import torch
    import event_utils_amd as E
    from event_utils_amd.events import _LIVE
    x = torch.arange(4, dtype=torch.float32)
    ev = E.DeviceEvents(x, x.clone(), x.clone(), x.clone())
    ev.many_evaluations = True
    ev.__dict__["_cmax_calls"] = {"k": {"args": [ctypes.c_void_p(3)]}}
    ev.__dict__["_cmax_last_single"] = ev.__dict__["_cmax_calls"]["k"]
    for clone in (copy.deepcopy(ev), pickle.loads(pickle.dumps(ev))):
        assert not any(k.startswith("_cmax") for k in clone.__dict__) and clone in _LIVE
        assert clone.many_evaluations and torch.equal(clone.x, ev.x) and len(clone) == 4
    o = E.variance_objective(adaptive_lifespan=True)
    o.sensor_size = (12, 16)
    o.__dict__["_fast_memo"] = (("key",), None, {"args": [ctypes.c_void_p(5)]})
    o2 = copy.deepcopy(o)
    assert "_fast_memo" not in o2.__dict__
    assert o2.sensor_size == (12, 16) and o2.adaptive_lifespan and "_fast_memo" in o.__dict__


def test_bench_extras_guard_prints_the_line_it_has_and_leaves():
    """bench.py's watchdog of the N > 1 extra legs: when a leg does not return (a collective one rank never enters), rank 0
    prints the line as it stands -- the headline measured before the extras -- with `extras_timed_out`, and the process
    leaves with exit code 0; when the legs do return in time nothing is printed."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "g = bench._ExtrasGuard({'metric': 'm', 'value': 1.5}, 0, float(sys.argv[1])); g.leg = 'breakdown'\n"
            "time.sleep(float(sys.argv[2])); g.done(); time.sleep(0.3); print('finished')\n" % root)
    r = subprocess.run([sys.executable, "-c", code, "0.2", "30"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.5 and line["extras_timed_out"]["leg"] == "breakdown"
    r = subprocess.run([sys.executable, "-c", code, "20", "0.1"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "finished", (r.stdout, r.stderr)


def test_image_kernels_compile_without_register_spills(tmp_path):
    """The same for evk_image2.hip: every k_part_sorted instantiation of the event-image / timestamp / indexed / derivative column
    sources and their tile kernels.  (The sources that load 24 words per four events -- int64 pixels -- have no 12-event
    instantiation: it spilled 60-81 registers.)"""
    import re
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.isfile(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    from event_utils_amd.csrc import build as B
    src = os.path.join(B.HERE, "evk_image2.hip")
    subprocess.run([hipcc] + list(B.CFLAGS) + ["-c", src, "-o", str(tmp_path / "i2.o"), "-save-temps=obj"], check=True, cwd=B.HERE,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = [f for f in os.listdir(tmp_path) if f.endswith("gfx950.s")]
    assert asm, os.listdir(tmp_path)
    text = open(tmp_path / asm[0]).read()
    kernels = re.findall(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text)
    seen = {n: (int(v), int(sp)) for n, v, sp in kernels if "k_part_sorted" in n or "k_image_tiles" in n}
    assert len(seen) >= 17, sorted(seen)
    assert not {n: vs for n, vs in seen.items() if vs[1]}
    assert all(v <= 128 for v, _ in seen.values())
    assert not any(("SrcIdx" in n or "SrcDrv" in n) and "ELi12E" in n for n in seen)
