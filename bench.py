#!/usr/bin/env python
"""
bench.py -- headline benchmark (BASELINE.json): Mevents/s of events_to_voxel_torch, 5 temporal bins, 640x480, 10 M
synthetic events per GPU resident in HBM (configs[1]); plus contrast-maximisation evaluations/s (configs[2] shape) in
the `cmax` object, one GPU's share of configs[4] (`c5_share`: 50 M events, 1280x720, beyond the Infinity Cache), the event
images on the same design (`image_10m`, `image_c1`) and the reference's CPU paths timed on this host (`cpu_baseline`).

The timed loop ROTATES over COLUMN_SETS distinct 10 M-event streams (640 MB of columns, more than the 256 MB Infinity
Cache holds), so that every step reads its events from HBM; `infinity_cache_resident` reports the same call re-run on ONE
set (160 MB of columns + 80 MB of records that stay in the cache between steps) beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
N = 1: a "step" is ONE call of the public drop-in signature events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=...) on
device tensors -- grid allocation, ts[0] / ts[-1], the out-of-range check and every kernel of the call included; only
the host<->device copies of the events are excluded.  `internal_step_ms` times the same kernels through the internal
entry point (resident grid, no check) beside it.
N > 1: `python bench.py --gpus N` re-launches itself under torch.distributed.run (one rank per GPU, RCCL; the driver's
own torchrun command line works as well).  Every rank voxelises ITS shard of the stream (weak scaling: 10 M events per
rank, disjoint time-sorted slices of one N*10 M stream) and the (B, H, W) grids are all-reduced (the path's only
exchange step); value = events of all ranks / max-over-ranks time; the reduced grid is checked (`checked`).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, B = 480, 640, 5
N_PER_GPU = 10_000_000
HBM_PEAK_GBS = 8000.0
COLUMN_SETS = 4       # distinct event streams the timed loop rotates over: 4 x 160 MB > the 256 MB Infinity Cache


def synth(seed, n, t_lo, t_hi, real_xy=False):
    rng = np.random.default_rng(seed)
    if real_xy:
        x = rng.uniform(1, W - 1, n).astype(np.float32)
        y = rng.uniform(1, H - 1, n).astype(np.float32)
    else:
        x = rng.integers(0, W, n).astype(np.float32)
        y = rng.integers(0, H, n).astype(np.float32)
    t = np.sort(rng.uniform(t_lo, t_hi, n)).astype(np.float32)
    p = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    return x, y, t, p


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 outside torch.distributed.run: re-exec under it (one rank per GPU, RCCL),
    forwarding every argument.  On a box with fewer than N GPUs rank >= n_visible fails loudly at set_device."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=7, help="repetitions of the K-step timed loop; the median is reported")
    ap.add_argument("--events", type=int, default=N_PER_GPU)
    ap.add_argument("--impl", default=None, help="kernel variant: direct | tiled | auto")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-cmax", action="store_true")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    # EVK_BENCH_DRY_RUN_ONE_GPU=1: a dry run of the N > 1 control flow where only ONE GPU exists -- every rank on cuda:0, the
    # collectives over gloo (RCCL refuses two ranks on one device).  Its numbers mean nothing and the line says so ("dry_run");
    # it exists so that rank-dependent branches, barriers and collectives have run with world_size > 1 before the first
    # multi-GPU node sees this file.
    dry = os.environ.get("EVK_BENCH_DRY_RUN_ONE_GPU") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)          # fails loudly when this rank has no GPU
    dev = torch.device("cuda", local_rank)
    dist = None
    # EVK_BENCH_FORCE_DIST=1 (with --gpus 1) drives the N > 1 code path -- process group, async all-reduce, barriers,
    # the self-check -- on a single GPU: a smoke test of the scaling harness where only one GPU is available
    use_dist = world > 1 or os.environ.get("EVK_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)

    import event_utils_amd as E
    from event_utils_amd import tiled
    from event_utils_amd.events import DeviceEvents
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device

    n = args.events
    # rank r holds the r-th time slice of one stream spanning [0, 0.1 s)
    span = 0.1 / world
    x, y, t, p = synth(1 + rank, n, rank * span, (rank + 1) * span)
    xd, yd, td, pd = (torch.from_numpy(a).to(dev) for a in (x, y, t, p))
    # the streams the timed loop rotates over: set 0 is the one above, the others differ in their seed only
    sets = [(xd, yd, td, pd)]
    t_lo_all, t_hi_all = float(t[0]), float(t[-1])
    for k in range(1, COLUMN_SETS):
        cols_k = synth(1000 * k + 1 + rank, n, rank * span, (rank + 1) * span)
        t_lo_all, t_hi_all = min(t_lo_all, float(cols_k[2][0])), max(t_hi_all, float(cols_k[2][-1]))
        sets.append(tuple(torch.from_numpy(a).to(dev) for a in cols_k))
    if use_dist:   # global ts[0] / ts[-1]: two scalars, agreed once outside the timed region
        lo = torch.tensor([t_lo_all], device=dev)
        hi = torch.tensor([t_hi_all], device=dev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        t_first, t_last = float(lo.item()), float(hi.item())
    else:
        t_first, t_last = float(t[0]), float(t[-1])

    impl = args.impl or tiled.default_impl()
    # N > 1: the all-reduce of step i (RCCL, its own stream) overlaps the kernels of step i+1 (double-buffered grids), or the
    # two run one after the other on one stream; every grid is fully reduced before the clock stops.  Overlapping pays when
    # the exchange takes about as long as the kernels (xGMI) and costs when it does not (its two cross-stream dependencies
    # per step are ~10 us of queue latency each), so -- unless EVK_BENCH_SYNC_ALLREDUCE=1 / 0 forces one -- both are timed
    # in the warm-up and the faster one (max over ranks, the same decision on every rank) is what the K steps run.
    forced = os.environ.get("EVK_BENCH_SYNC_ALLREDUCE")
    overlap = use_dist and forced != "1"
    outs = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(2 if use_dist else 1)]
    works = [None] * len(outs)
    out = outs[0]

    def step_sharded(i):
        # one complete voxelisation of this rank's shard into a resident grid (global ts[0] / ts[-1] agreed once), then
        # the grid is summed over the ranks: the path's only exchange step
        k = i % len(outs)
        if works[k] is not None:
            works[k].wait()          # stream-level: this buffer's previous all-reduce has finished
            works[k] = None
        _voxel_f32_device(*sets[i % COLUMN_SETS], B, (H, W), t_first, t_last, out=outs[k], check=False, impl=impl, fresh=True)
        if overlap:
            works[k] = dist.all_reduce(outs[k], op=dist.ReduceOp.SUM, async_op=True)
        else:
            dist.all_reduce(outs[k], op=dist.ReduceOp.SUM)

    keep = [None]

    def step_public(i):
        # N = 1: the reference's own call on device tensors -- events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=...)
        # (voxel_grid.py:114): allocates the grid, reads ts[0] / ts[-1] on the device, counts out-of-range events
        keep[0] = E.events_to_voxel_torch(*sets[i % COLUMN_SETS], B, sensor_size=(H, W))

    def step_resident(i):   # the same call on ONE stream: its 160 MB of columns stay in the Infinity Cache between steps
        keep[0] = E.events_to_voxel_torch(xd, yd, td, pd, B, sensor_size=(H, W))

    def step_internal(i):
        _voxel_f32_device(xd, yd, td, pd, B, (H, W), t_first, t_last, out=out, check=False, impl=impl, fresh=True)

    step = step_sharded if use_dist else step_public

    def drain():
        for k, wk in enumerate(works):
            if wk is not None:
                wk.wait()
                works[k] = None

    def timed(fn, steps, warm):
        # device wake-up (clocks, first-touch of every buffer, RCCL channel setup), then the W warm-up steps asked for:
        # with a small W the first timed steps would otherwise still be ramping
        for i in range(300):     # ~25 ms: 30 steps (2.5 ms) left the first timed loop 5 % slower than the later ones
            fn(i)
        drain()
        for i in range(warm):
            fn(i)
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            fn(i)
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    exchange_choice = None
    if use_dist and forced not in ("0", "1"):
        # (overlapped: the kernels leave LDS on every CU for the collective's workgroups, share_cu -- the library's default
        # in a multi-rank job; serial: nothing runs beside them, so they take the single-GPU geometry)
        trial = {}
        for mode in (True, False):
            overlap = mode
            tiled.FORCE["share_cu"] = mode
            trial[mode] = timed(step, max(args.steps, 20), args.warmup)     # identical on every rank (max over ranks)
        overlap = trial[True] <= trial[False]
        tiled.FORCE["share_cu"] = overlap
        exchange_choice = {"overlapped_ms": round(trial[True] / max(args.steps, 20) * 1e3, 4),
                           "serial_ms": round(trial[False] / max(args.steps, 20) * 1e3, 4),
                           "chosen": "overlapped" if overlap else "serial", "share_cu": overlap}
    # Round 6: the K-step region is ~1.5 ms, and one such region scatters by more than the changes a round makes (boxes differ by
    # +-6 %, a single region by +-3 %): the K steps are timed REPS times -- each repetition exactly K steps between a barrier +
    # synchronize on both sides, max over ranks, after its own W warm-up steps -- and the line reports the MEDIAN repetition
    # (`ms_per_step`, `value`), with every repetition and the spread beside it (`repetitions`).
    reps_ms = []
    for _ in range(max(1, args.reps)):
        reps_ms.append(timed(step, args.steps, args.warmup) / args.steps * 1e3)
    E.check_errors()                     # deferred out-of-range reports of the timed calls (none expected)
    ms_per_step = float(np.median(reps_ms))
    elapsed = ms_per_step * args.steps * 1e-3
    value = n * world / (elapsed / args.steps) / 1e6
    # device-side time per step, outside the timed region (an event pair around every step of the timed loop would
    # itself cost ~8 us per step): one pair of HIP events around K more steps
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(args.steps):
        step(i)
    drain()
    e1.record()
    torch.cuda.synchronize()
    dev_ms = e0.elapsed_time(e1) / args.steps

    # ---- roofline of the dominant kernel(s): HIP-event timing of the voxel call alone (no memset, no collective) ----
    kinfo = tiled.time_voxel_kernels(sets, t_first, t_last, B, H, W, impl=impl, reps=max(5, args.steps))
    alg_bytes = 16.0 * n + out.numel() * 4.0
    roofline = roofline_block(kinfo, alg_bytes, n, "c2")

    result = {
        "metric": "Mevents/s (voxel 5-bin 640x480)", "value": round(value, 1), "unit": "Mevents/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: 10M events/GPU, 640x480, events_to_voxel_torch 5 temporal bins "
                               "(temporal-bilinear, nearest pixel), uniform-random events with +-1 polarities; the timed loop "
                               "rotates over %d distinct event streams (%d MB of float32 columns in HBM, more than the 256 MB "
                               "Infinity Cache), every step reads its events from HBM" % (COLUMN_SETS, COLUMN_SETS * 16 * n // 10**6),
                   "events_per_gpu": n, "sensor": [H, W], "bins": B, "impl": kinfo["impl"],
                   "timed_call": ("_voxel_f32_device on this rank's shard (global ts[0]/ts[-1] agreed once, resident "
                                  "output grids) + all_reduce(SUM) of the grid" if use_dist else
                                  "the public events_to_voxel_torch(xs, ys, ts, ps, B, sensor_size=(480, 640)) on device "
                                  "tensors: allocates the grid, reads ts[0]/ts[-1] on the device, counts out-of-range "
                                  "events (EVK_ERRORS=%s)" % E.error_mode()),
                   "parallelism": ("event-sharded x%d, RCCL all-reduce of the (B,H,W) grid per step%s"
                                   % (world, ", overlapped with the next step's kernels" if overlap else ""))
                   if use_dist else "single GPU"},
        "device_ms_per_step": round(dev_ms, 4),
        "repetitions": {"count": len(reps_ms), "steps_each": args.steps, "ms_per_step": [round(v, 4) for v in reps_ms],
                        "median": round(ms_per_step, 4), "min": round(min(reps_ms), 4), "max": round(max(reps_ms), 4),
                        "spread_pct": round(100.0 * (max(reps_ms) - min(reps_ms)) / ms_per_step, 2),
                        "note": "ms_per_step / value are the MEDIAN repetition; every repetition is exactly `steps` steps between "
                                "barrier + synchronize, after `warmup` warm-up steps"},
        "roofline": roofline,
    }
    if dry:
        result["dry_run"] = "EVK_BENCH_DRY_RUN_ONE_GPU=1: %d ranks share cuda:0, collectives over gloo -- control flow only, NOT a measurement" % world
    guard = None
    if use_dist:
        # From here on everything is EXTRA information beside a headline that has been measured.  These legs have run on ONE
        # GPU only (EVK_BENCH_FORCE_DIST, the gloo dry run): should one of them hang on the first multi-GPU node -- a collective
        # that one rank never enters --, rank 0 still prints the line it has after EVK_BENCH_EXTRAS_TIMEOUT seconds
        # (`extras_timed_out` says which leg it was in) and every rank leaves.  A WRONG result is different: the self-checks
        # below end the run without a line (SystemExit), as before.
        guard = _ExtrasGuard(result, rank, float(os.environ.get("EVK_BENCH_EXTRAS_TIMEOUT", "420")))
        guard.leg = "check_sharded"
        result["rccl_ranks"] = 0 if dry else dist.get_world_size()       # (the dry run's collectives are gloo's)
        result.update(check_sharded(dist, dev, outs[(args.steps - 1) % 2], sets[(args.steps - 1) % COLUMN_SETS][3], n, world))
        if exchange_choice:
            result["exchange"] = exchange_choice
        guard.leg = "oracle_self_check"
        result["oracle_self_check"] = oracle_self_check(dist, dev, _voxel_f32_device, sets[0], t_first, t_last, impl)
        guard.leg = "breakdown"
        # ---- what the step is made of, so that a 1 -> N curve can be read without re-running: the kernels alone, the grid
        #      collective alone (same buffer shape, both forms), the two back to back, and the N = 1 entry point on this rank
        from event_utils_amd import distributed as DD
        zgrid = torch.zeros((B, H, W), dtype=torch.float32, device=dev)     # zeros stay zeros under repeated sums

        def compute_only(i):
            _voxel_f32_device(*sets[i % COLUMN_SETS], B, (H, W), t_first, t_last, out=outs[0], check=False, impl=impl, fresh=True)

        def allreduce_only(i):
            dist.all_reduce(zgrid, op=dist.ReduceOp.SUM)

        def rsag_only(i):
            DD.reduce_scatter_all_gather_sum_(zgrid)

        def serial(i):
            compute_only(i)
            dist.all_reduce(outs[0], op=dist.ReduceOp.SUM)

        def serial_rsag(i):
            compute_only(i)
            DD.reduce_scatter_all_gather_sum_(outs[0])

        def banded(K):
            # ONE call with its exchange overlapped INSIDE it: one partition, the tile kernel in K row bands, band k all-reduced
            # (asynchronously, RCCL's stream) while band k + 1 accumulates, the reduced bands assembled at the end
            def run(i):
                c = sets[i % COLUMN_SETS]
                DD.banded_exchange(tiled.voxel2_bands(c, n, t_first, t_last, B, H, W, K), outs[0])
            return run
        ms = lambda fn: round(timed(fn, args.steps, args.warmup) / args.steps * 1e3, 4)   # noqa: E731

        def with_share_cu(flag, fn):      # the kernels in the geometry that leaves LDS for a collective's workgroups, or not
            prev = tiled.FORCE["share_cu"]
            tiled.FORCE["share_cu"] = flag
            try:
                return ms(fn)
            finally:
                tiled.FORCE["share_cu"] = prev
        breakdown = lambda: {   # noqa: E731
            "headline_ms": round(ms_per_step, 4), "compute_ms": ms(compute_only),
            "compute_share_cu_on_ms": with_share_cu(True, compute_only), "compute_share_cu_off_ms": with_share_cu(False, compute_only),
            "allreduce_ms": ms(allreduce_only),
            "reduce_scatter_all_gather_ms": ms(rsag_only), "serial_ms": ms(serial), "serial_rsag_ms": ms(serial_rsag),
            "banded_ms": {str(K): ms(banded(K)) for K in (2, 4)},
            "n1_equivalent_ms": ms(step_public), "grid_bytes": int(zgrid.numel() * 4),
            "note": "max over ranks, barrier + synchronize on both sides like ms_per_step.  headline_ms = the timed step "
                    "(`exchange`: all-reduce of step i overlapped with the kernels of step i+1, or serial -- the faster of "
                    "the two in the warm-up); compute_ms = this rank's kernels with "
                    "no collective (internal entry, resident grid); allreduce_ms / reduce_scatter_all_gather_ms = the grid "
                    "exchange alone; serial_* = kernels then exchange, not overlapped; banded_ms[K] = ONE call whose tile "
                    "kernel runs in K row bands, band k all-reduced while band k + 1 accumulates (EVK_VOXEL_COLLECTIVE=bandsK: "
                    "the exchange overlapped inside a single call); n1_equivalent_ms = the public "
                    "events_to_voxel_torch call BENCH's N = 1 `value` times, here on every rank at once without a "
                    "collective (it allocates the grid and reads ts[0]/ts[-1] on the device: ~1 % above compute_ms); "
                    "compute_share_cu_on / _off_ms = compute_ms with the partition geometry that leaves LDS for a collective's "
                    "workgroups forced on / off (tiled.FORCE['share_cu']; default: on in a multi-rank job)"}
        try:   # extra information only: it must never cost the scaling run its JSON line
            result["breakdown"] = breakdown()
        except Exception as e:  # noqa: BLE001
            result["breakdown"] = {"error": repr(e)}
    else:
        # the same work through the internal entry point (resident output, host-supplied ts[0]/ts[-1], no out-of-range
        # check) and through the public call with per-call synchronous error reporting
        el_int = timed(step_internal, args.steps, args.warmup)
        el_res = timed(step_resident, args.steps, args.warmup)
        result["infinity_cache_resident"] = {
            "ms_per_step": round(el_res / args.steps * 1e3, 4), "Mevents_per_s": round(n / (el_res / args.steps) / 1e6, 1),
            "whole_call_frac": round(alg_bytes / (el_res / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "the headline call re-run on ONE event stream: its 160 MB of columns and 80 MB of records stay in the 256 MB "
                    "Infinity Cache between steps (what rounds 1-3 reported as the headline)"}
        # float32 weights that are not +-1 / 0: two float64 LDS atomics per event in the tile kernel instead of the integer
        # counting mode (DESIGN.md section 3, K2').  (a) weights a record carries exactly in its 21 polarity bits (small
        # integers, halves, quarters: p * {0.25 .. 3}); (b) arbitrary float32 weights, which go through the side array.
        keepg = [None]
        for key, mult in (("voxel_general_polarity_ms", torch.tensor([0.25, 0.5, 1.0, 1.5, 2.0, 3.0], device=dev)[torch.arange(n, device=dev) % 6]),
                          ("voxel_wide_polarity_ms", torch.linspace(0.5, 1.5, n, device=dev))):
            pgen = (pd * mult).contiguous()

            def step_general(i, pgen=pgen):
                c = sets[i % COLUMN_SETS]
                keepg[0] = E.events_to_voxel_torch(c[0], c[1], c[2], pgen, B, sensor_size=(H, W))
            result[key] = round(timed(step_general, args.steps, args.warmup) / args.steps * 1e3, 4)
            del pgen, mult
        # EVK_VOXEL2_LIVE (round 5, opt-in): the same public call with its tiles accumulated WHILE the partition sorts, by a
        # consumer kernel on the library's second stream -- measured slower than the two launches it overlaps (DESIGN.md
        # section 3, profiles/r05_live_ab.txt); reported here so that every bench run repeats the A/B
        tiled.FORCE["live"] = True
        try:
            result["voxel_live_ms"] = round(timed(step_public, args.steps, args.warmup) / args.steps * 1e3, 4)
        finally:
            tiled.FORCE["live"] = None
        # the headline runs in the DEFAULT error mode (strict since round 5: the reference's synchronous IndexError, the
        # call waits for its partition kernel's report); the opt-in deferred mode beside it
        prev_mode = os.environ.get("EVK_ERRORS")
        os.environ["EVK_ERRORS"] = "deferred"
        el_deferred = timed(step_public, args.steps, args.warmup)
        E.check_errors()
        if prev_mode is None:
            os.environ.pop("EVK_ERRORS")
        else:
            os.environ["EVK_ERRORS"] = prev_mode
        result["public_api_ms"] = round(ms_per_step, 4)
        result["internal_step_ms"] = round(el_int / args.steps * 1e3, 4)
        result["error_mode"] = E.error_mode()
        result["public_api_deferred_errors_ms"] = round(el_deferred / args.steps * 1e3, 4)

    if use_dist and not args.no_cmax:
        guard.leg = "c5"
        try:   # extra information only: it must never cost the scaling run its JSON line
            c5 = bench_c5(E, DeviceEvents, dist, rank, world, dev, impl)
        except Exception as e:  # noqa: BLE001
            c5 = {"error": repr(e)}
        result["c5"] = c5
    if rank == 0 and world == 1 and not use_dist:
        try:
            result["c5_share"] = bench_c5_share(tiled, dev, impl)
        except Exception as e:  # noqa: BLE001
            result["c5_share"] = {"error": repr(e)}
        try:
            result["prebucketed"] = bench_prebucketed(tiled, sets, n, t_first, t_last, dev, max(8, args.steps))
        except Exception as e:  # noqa: BLE001
            result["prebucketed"] = {"error": repr(e)}
        result["native_dtypes"] = bench_native(DeviceEvents, _voxel_f32_device, x, y, t, p, B, H, W, impl,
                                               max(5, args.steps))
        try:
            result["from_host_arrays"] = bench_from_host(E, x, y, t, p, B, H, W)
        except Exception as e:  # noqa: BLE001
            result["from_host_arrays"] = {"error": repr(e)}
        for key, fn in (("voxel_structured", bench_structured), ("image_10m", bench_image_10m), ("image_c1", bench_image_c1)):
            try:
                result[key] = fn(E, tiled, dev, impl)
            except Exception as e:  # noqa: BLE001
                result[key] = {"error": repr(e)}
    if rank == 0 and world == 1 and not use_dist and not args.no_cmax:
        result["cmax"] = bench_cmax(E, DeviceEvents, dev, impl)
    if rank == 0 and world == 1 and not use_dist and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(x, y, t, p)
    if use_dist:
        guard.leg = "final barrier"
        dist.barrier()
        guard.done()
        dist.destroy_process_group()
    # RCCL's version banner sits in the C stdio buffer (stdout is a pipe) and would come out at exit, AFTER the JSON:
    # push everything out first so that the JSON is the last line of stdout
    import ctypes
    sys.stdout.flush()
    ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(json.dumps(result), flush=True)


class _ExtrasGuard:
    """Watchdog of the N > 1 line's extra legs (see main): after `seconds` rank 0 prints the line as it stands and every
    rank leaves with os._exit -- a rank blocked inside a collective cannot be unwound any other way."""

    def __init__(self, result, rank, seconds):
        import threading
        self.result, self.rank, self.seconds, self.leg = result, rank, seconds, "?"
        self._finished = threading.Event()
        self._thread = threading.Thread(target=self._watch, daemon=True)
        self._thread.start()

    def done(self):
        self._finished.set()

    def _watch(self):
        if self._finished.wait(self.seconds):
            return
        try:
            if self.rank == 0:
                line = dict(self.result)
                line["extras_timed_out"] = {"leg": self.leg, "after_seconds": self.seconds,
                                            "note": "the headline above was complete; an extra leg did not return"}
                sys.stdout.flush()
                print(json.dumps(line), flush=True)
        finally:
            if self.rank != 0:
                time.sleep(5.0)     # rank 0's line first: the launcher ends the job when any rank has left
            os._exit(0)


def roofline_block(kinfo, alg_bytes, n, tag):
    """`frac` / `achieved` are the WHOLE CALL's: algorithmic bytes / the sum of its kernels' HIP-event durations (the
    partition is part of the single-shot call); the dominant kernel's own figure is `dominant_kernel_frac`."""
    dom_ms, call_ms = kinfo["dominant_ms"], sum(kinfo["kernels_ms_exact"].values()) if "kernels_ms_exact" in kinfo else kinfo["total_ms"]
    achieved = alg_bytes / (call_ms * 1e-3) / 1e9
    traffic, traffic_src, call_traffic = pmc_traffic(kinfo["dominant"], tag)
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": call_traffic, "traffic_source": traffic_src,
         "kernel": " + ".join(kinfo["kernels_ms"]) + " (the whole call)", "kernel_ms": round(call_ms, 4),
         "algorithmic_bytes": alg_bytes,
         "dominant_kernel": kinfo["dominant"], "dominant_kernel_ms": round(dom_ms, 4),
         "dominant_kernel_frac": round(alg_bytes / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "dominant_kernel_traffic": traffic,
         "back_to_back_call_ms": round(kinfo["total_ms"], 4),
         "kernels_ms": kinfo["kernels_ms"]}
    return r


def check_sharded(dist, dev, grid, pd, n, world):
    """Self-check of the N > 1 line: (a) every event puts its polarity, split over two bins, into the grid, so the
    reduced grid must sum to the polarity sum of ALL ranks' events; (b) every rank must hold the same reduced grid."""
    torch.cuda.synchronize()
    psum = pd.double().sum().reshape(1)
    dist.all_reduce(psum, op=dist.ReduceOp.SUM)
    gsum = float(grid.double().sum().item())
    w = torch.arange(grid.numel(), device=dev, dtype=torch.float64).reshape(grid.shape) % 251.0
    cs = (grid.double() * w).sum().reshape(1)
    cmin, cmax = cs.clone(), cs.clone()
    dist.all_reduce(cmin, op=dist.ReduceOp.MIN)
    dist.all_reduce(cmax, op=dist.ReduceOp.MAX)
    tol = 1e-3 * float(np.sqrt(n * world))
    ok = abs(gsum - float(psum.item())) <= tol and float(cmin.item()) == float(cmax.item())
    if not ok:
        raise SystemExit("sharded voxel grid failed its self-check: grid sum %r vs polarity sum %r (tol %g), checksum "
                         "min %r max %r over ranks" % (gsum, float(psum.item()), tol, float(cmin.item()), float(cmax.item())))
    return {"checked": True, "check": {"grid_sum": gsum, "polarity_sum_all_ranks": float(psum.item()), "tolerance": tol,
                                        "grid_checksum_identical_on_all_ranks": True}}


def oracle_self_check(dist, dev, voxel, cols, t_first, t_last, impl, m=200_000):
    """Per-rank self-check of the N > 1 line against the ORACLE (the checker, outside every timed region): the first m events of
    this rank's shard voxelised by the product path -- with the GLOBAL ts[0] / ts[-1], as the sharded step does -- against
    oracle/reference_np.events_to_voxel_torch(t_range=...) on the host; bar 1e-5 of the grid's maximum (north_star).  Every rank
    checks its own shard; the line carries the worst rank's error and fails loudly if any rank is off."""
    from oracle import reference_np as R
    x, y, t, p = (c[:m].contiguous() for c in cols)
    out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    voxel(x, y, t, p, B, (H, W), t_first, t_last, out=out, check=False, impl=impl, fresh=True)
    ref = R.events_to_voxel_torch(*(c.cpu().numpy() for c in (x, y, t, p)), B, sensor_size=(H, W), accum="f64",
                                  t_range=(t_first, t_last))
    err = float(np.abs(out.cpu().numpy().astype(np.float64) - ref).max())
    scale = float(np.abs(ref).max())
    rel = torch.tensor([err / max(scale, 1e-30)], device=dev, dtype=torch.float64)
    dist.all_reduce(rel, op=dist.ReduceOp.MAX)
    worst = float(rel.item())
    if not worst <= 1e-5:
        raise SystemExit("sharded voxel path failed its oracle self-check: worst rank's max error %.3e of the grid maximum" % worst)
    return {"events_per_rank": int(x.numel()), "max_err_over_grid_max_worst_rank": worst, "bar": 1e-5, "ok": True,
            "oracle": "oracle/reference_np.events_to_voxel_torch (float64 accumulation, global ts[0]/ts[-1])"}


def bench_c5_share(tiled, dev, impl):
    """One GPU's share of configs[4]: 50 M events, 1280x720, 5 bins -- 800 MB of columns, beyond the 256 MB Infinity
    Cache, i.e. the HBM-resident size of the voxel call (the 10 M-event headline re-reads 160 MB that stay in the MALL)."""
    H5, W5, n5 = 720, 1280, 50_000_000
    rng = np.random.default_rng(40)
    x = rng.integers(0, W5, n5).astype(np.float32)
    y = rng.integers(0, H5, n5).astype(np.float32)
    t = np.sort(rng.uniform(0.0, 0.1, n5)).astype(np.float32)
    p = (rng.integers(0, 2, n5) * 2 - 1).astype(np.float32)
    cols = [torch.from_numpy(a).to(dev) for a in (x, y, t, p)]
    kinfo = tiled.time_voxel_kernels([cols], float(t[0]), float(t[-1]), B, H5, W5, impl=impl, reps=10)
    alg = 16.0 * n5 + B * H5 * W5 * 4.0
    res = {"workload": "one rank's share of configs[4]: 50M events, 1280x720, 5 bins, single GPU, HBM-resident (800 MB)",
           "ms_per_call": round(kinfo["total_ms"], 4), "Mevents_per_s": round(n5 / kinfo["total_ms"] / 1e3, 1),
           "whole_call_frac": round(alg / (kinfo["total_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
           "roofline": roofline_block(kinfo, alg, n5, "c5_share")}
    del cols
    torch.cuda.empty_cache()
    return res


def bench_prebucketed(tiled, sets, n, t_first, t_last, dev, reps):
    """SURVEY.md 8(d) "with and without the bucketing pre-pass": BASELINE.json's north_star defines its kernel on "events
    pre-sorted/bucketed by output tile".  The streams of the headline loop are bucketed ONCE (evk_bucket_events_f32: 16-byte
    records (x, y, t, p), tile-contiguous; not timed as part of the kernel, reported beside it), then evk_voxel_tiled_f32 --
    LDS tile accumulate, exclusive plain-store flush -- is timed with HIP events on the launch stream, rotating over the
    bucketed streams (640 MB of records: every launch reads its records from HBM).  Algorithmic bytes = 16 B/event + the grid,
    exactly what this kernel moves (`traffic`: the committed PMC pass of the same workload)."""
    from event_utils_amd import _lib, _device as D
    L = _lib.lib()
    tw, th = 4, 4        # 16 x 16 pixel tiles: 1200 tiles at 640x480 -- 4.7 per CU (600 tiles of 32 x 16 are 3 workgroups on 88 CUs
                         # and 2 on the rest: 41.7 against 35.6 us, tools/prebucketed_sweep.py)
    bks = [tiled.bucket_events(*c, 0, H, W, tw, th) for c in sets]
    nbytes = int(L.evk_voxel_tiled_staging_bytes(bks[0].ntiles, n, B, tw, th))
    staging = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
    out = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    it = [0]

    def kernel():
        it[0] += 1
        bk = bks[it[0] % len(bks)]
        _lib.call("evk_voxel_tiled_f32", D.ptr(bk.records), D.ptr(bk.bucket_start), n, H, W, tw, th, t_first, t_last, B,
                  _lib.EVK_VOXEL_OVERWRITE, D.ptr(out), D.ptr(staging), nbytes, D.stream())

    def bucket():
        it[0] += 1
        k = it[0] % len(bks)
        tiled.bucket_events(*sets[k], 0, H, W, tw, th, into=bks[k])
    reps = max(reps, len(bks)) // len(bks) * len(bks)
    k_ms = tiled._time_ms(kernel, reps)
    # self-check of what was timed: the grid of stream 1 against the one-pass call's (both accumulate float64 in LDS)
    it[0] = 0
    kernel()
    ref = torch.empty_like(out)
    tiled.voxel_f32(*sets[1 % len(sets)], t_first, t_last, B, H, W, ref, None, impl="tiled", fresh=True)
    err = float((out - ref).abs().max().item())
    scale = float(ref.abs().max().item())
    b_ms = tiled._time_ms(bucket, reps)
    alg = 16.0 * n + B * H * W * 4.0
    traffic, src, _ = pmc_traffic("k_voxel_tiled", "prebucketed")
    return {"workload": "configs[1] with the events ALREADY bucketed by output tile (north_star's own assumption): 10M events, "
                        "640x480, 5 bins; 16-byte (x, y, t, p) records, tile-contiguous, %d streams rotating (HBM-resident)" % len(bks),
            "kernel": "k_voxel_tiled (evk_voxel_tiled_f32): one workgroup per 16x16 tile, float64 LDS accumulators, plain-store flush",
            "kernel_ms": round(k_ms, 4), "Mevents_per_s": round(n / k_ms / 1e3, 1),
            "roofline": {"bound": "hbm", "achieved": round(alg / (k_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": src,
                         "algorithmic_bytes": alg},
            "bucketing_ms": round(b_ms, 4),
            "bucketing_note": "evk_bucket_events_f32 (histogram + scan + write-combined scatter), paid once per stream; with it "
                              "the pair takes kernel_ms + bucketing_ms -- the headline's one-pass call does both jobs in ms_per_step",
            "max_abs_diff_vs_one_pass_call": err, "grid_max": scale, "checked": bool(err <= 1e-5 * max(scale, 1.0))}


def bench_native(DeviceEvents, voxel, x, y, t, p, B, H, W, impl, reps):
    """The same workload held in the reference's on-disk dtypes (int16 x, y; float64 epoch-second t; uint8 {0,1} p:
    13 B/event): (a) bucketed straight from those columns, (b) widened to four float32 columns
    by evk_native_to_columns_f32 first (what a loader's casts amount to, done on the device) and then voxelised."""
    ev = DeviceEvents.from_native(x.astype(np.int16), y.astype(np.int16), 1.6e9 + t.astype(np.float64),
                                  ((p + 1) / 2).astype(np.uint8))
    nat = ev.native
    out = torch.empty((B, H, W), dtype=torch.float32, device=nat.t.device)
    t_first, t_last = ev.t_at(0), ev.t_at(-1)

    def direct_from_native():
        voxel(None, None, None, None, B, (H, W), t_first, t_last, out=out, check=False, impl=impl, fresh=True, native=nat)

    def widen_then_voxel():
        cols = nat.widen()
        voxel(*cols, B, (H, W), t_first, t_last, out=out, check=False, impl=impl, fresh=True)
    res = {"bytes_per_event": 13, "layout": "x,y int16 + t float64 + p uint8 columns"}
    for name, fn in (("from_native_ms", direct_from_native), ("widen_then_f32_path_ms", widen_then_voxel)):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name] = round(a.elapsed_time(b) / reps, 4)
    res["from_native_Mevents_s"] = round(len(x) / res["from_native_ms"] / 1e3, 1)
    return res


def bench_structured(E, tiled, dev, impl):
    """The headline call on scenes that are not uniform noise (real event data is edges and blobs): the moving-edge scene
    of configs[3] at 10 M events / 640x480 and 50 M / 1280x720, and a blob holding half of the events in 100x100 pixels.
    A launch of the tile kernel lasts as long as its busiest CU; hot tiles are cut into pieces (evk_voxel2.hip)."""
    res = {"note": "events_to_voxel_torch, 5 bins, kernels of one call; ratio = total_ms / the uniform-random call of the same size"}
    for (Hs, Ws, ns, reps) in ((H, W, N_PER_GPU, 10), (720, 1280, 50_000_000, 5)):
        tag = "%dx%d_%dM" % (Ws, Hs, ns // 1_000_000)
        base = None
        for scene in ("uniform", "moving_edges", "hot_blob"):
            if scene == "hot_blob" and ns > N_PER_GPU:
                continue
            rng = np.random.default_rng(2)
            if scene == "moving_edges":
                xs, ys, ts, ps = structured_scene(3, ns, Hs, Ws)
                xs, ys = np.floor(xs), np.floor(ys)
            else:
                xs = rng.integers(0, Ws, ns).astype(np.float32)
                ys = rng.integers(0, Hs, ns).astype(np.float32)
                if scene == "hot_blob":
                    hot = rng.random(ns) < 0.5
                    xs[hot] = (Ws // 3 + rng.integers(0, 100, int(hot.sum()))).astype(np.float32)
                    ys[hot] = (Hs // 3 + rng.integers(0, 100, int(hot.sum()))).astype(np.float32)
                ts = np.sort(rng.uniform(0.0, 0.1, ns)).astype(np.float32)
                ps = (rng.integers(0, 2, ns) * 2 - 1).astype(np.float32)
            cols = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (xs, ys, ts, ps)]
            k = tiled.time_voxel_kernels([cols], float(ts[0]), float(ts[-1]), B, Hs, Ws, impl=impl, reps=reps)
            base = k["total_ms"] if scene == "uniform" else base
            res["%s_%s" % (tag, scene)] = {"total_ms": round(k["total_ms"], 4), "kernels_ms": k["kernels_ms"],
                                          "Mevents_per_s": round(ns / k["total_ms"] / 1e3, 1),
                                          "ratio_to_uniform": round(k["total_ms"] / base, 3)}
            del cols
        torch.cuda.empty_cache()
    return res


def _image_kernel_times(tiled, kind, cols, n, Hc, Wc, dtype, reps):
    """HIP-event times of one event-image call through the one-pass path (evk_image2.hip): whole call, partition, tiles."""
    from event_utils_amd import _lib
    inf = float("inf")
    img = torch.zeros((Hc, Wc), dtype=dtype, device=cols[0].device)
    call = lambda stage=0: tiled.image2(kind, *cols, n, Hc, Wc, inf, inf, img, None, fresh=(kind != "bilinear"), stage=stage)  # noqa: E731
    if not call():
        return None
    return {"call_ms": tiled._time_ms(call, reps), "k_part_sorted": tiled._time_ms(lambda: call(_lib.EVK_VOXEL2_PARTITION_ONLY), reps),
            "k_image_tiles": tiled._time_ms(lambda: call(_lib.EVK_VOXEL2_TILES_ONLY), reps)}


def bench_image_10m(E, tiled, dev, impl):
    """The event images at the headline's size: 10 M events on the 640x480 sensor through the one-pass partition + LDS tiles
    (evk_image2.hip), device-resident columns, 12 B/event algorithmic (x, y, weight) + the image.  events_to_image (numpy
    path: int32 columns / canvas, bit-exact) is timed at its C entry point (the public call is numpy in / numpy out: PCIe
    both ways); events_to_image_torch through the public call on device tensors."""
    from event_utils_amd import _lib, _device as D
    n = N_PER_GPU
    rng = np.random.default_rng(5)
    x = rng.uniform(0, W - 1, n).astype(np.float32); y = rng.uniform(0, H - 1, n).astype(np.float32)
    pu = (rng.integers(0, 2, n) * 2 - 1).astype(np.float32)
    xd, yd, pud = (torch.from_numpy(a).to(dev) for a in (x, y, pu))
    pfd = (pud * torch.linspace(0.5, 1.5, n, device=dev)).contiguous()
    xi, yi, pi = xd.int(), yd.int(), pud.int()
    alg = 12.0 * n + H * W * 4.0
    res = {"workload": "10M events, 640x480, uniform-random; 12 B/event (x, y, weight) + the image = %.1f MB algorithmic" % (alg / 1e6),
           "impl": "one-pass partition (k_part_sorted, 4-byte nearest / 8-byte bilinear records, exact float weights as a side run where needed) + k_image_tiles_n / _b; tiles %dx%d"
                   % tiled.voxel2_shape(H, W, 1)}

    def block(k, public_ms=None, tag=None):
        if k is None:
            return {"error": "no tiling"}
        b = {"call_ms": round(k["call_ms"], 4), "Mevents_per_s": round(n / k["call_ms"] / 1e3, 1),
             "roofline_frac": round(alg / (k["call_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
             "kernels_ms": {"k_part_sorted": round(k["k_part_sorted"], 4), "k_image_tiles": round(k["k_image_tiles"], 4)}}
        if public_ms is not None:
            b["public_call_ms"] = round(public_ms, 4)
            b["public_Mevents_per_s"] = round(n / public_ms / 1e3, 1)
        if tag is not None:     # HBM bytes of one call from the committed rocprofv3 PMC passes of this same workload
            _, src, call_traffic = pmc_traffic("k_part_sorted", tag)
            if call_traffic:
                b["traffic"], b["traffic_source"] = call_traffic, src
        return b

    def public(**kw):
        keep = [None]

        def fn():
            keep[0] = E.events_to_image_torch(xd, yd, kw.pop("_p", pud), sensor_size=(H, W), **kw)
        return fn
    res["events_to_image_int32"] = block(_image_kernel_times(tiled, "i32", (xi, yi, pi), n, H, W, torch.int32, 20), tag="img_nearest")
    for name, kind, pcol, kw in (("events_to_image_torch_nearest", "f32", pud, dict(interpolation=None, padding=False)),
                                 ("events_to_image_torch_nearest_float_weights", "f32", pfd, dict(interpolation=None, padding=False)),
                                 ("events_to_image_torch_bilinear", "bilinear", pud, dict(interpolation='bilinear', padding=False)),
                                 ("events_to_image_torch_bilinear_float_weights", "bilinear", pfd,
                                  dict(interpolation='bilinear', padding=False))):
        keep = [None]

        def fn(pcol=pcol, kw=kw):
            keep[0] = E.events_to_image_torch(xd, yd, pcol, sensor_size=(H, W), **kw)
        res[name] = block(_image_kernel_times(tiled, kind, (xd, yd, pcol), n, H, W, torch.float32, 20), tiled._time_ms(fn, 20),
                          tag="img_bilinear" if name == "events_to_image_torch_bilinear" else None)
    E.check_errors()
    # the global-atomic kernels these calls ran on until round 4 (EVK_IMPL=direct), for the record
    canvas = torch.zeros((H, W), dtype=torch.int32, device=dev)
    img = torch.zeros((H, W), dtype=torch.float32, device=dev)
    inf = float("inf")
    res["direct_kernels_ms"] = {
        "k_image_nearest_int": round(tiled._time_ms(lambda: _lib.call("evk_image_nearest_i32", D.ptr(xi), D.ptr(yi), D.ptr(pi), n, H, W,
                                                                      D.ptr(canvas), None, D.stream()), 3), 4),
        "k_image_bilinear_f32": round(tiled._time_ms(lambda: _lib.call("evk_image_bilinear_f32", D.ptr(xd), D.ptr(yd), D.ptr(pud), n, H,
                                                                       W, inf, inf, D.ptr(img), None, D.stream()), 3), 4)}
    # the average-timestamp images (events_to_timestamp_image_torch, image.py:285-353) on the same design
    # since round 6: 16 B/event + four (H+1, W+1) planes; eight global atomics per event before
    try:
        td_ = torch.sort(torch.rand(n, device=dev) * 0.1).values.contiguous()
        planes = torch.zeros((4, H + 1, W + 1), dtype=torch.float32, device=dev)
        ts_args = (n, H + 1, W + 1, float(W), float(H), 0, 0.0, 0.1)
        ts_call = lambda stage=0: tiled.timestamp_images2(xd, yd, td_, pud, *ts_args, planes, None, stage=stage)  # noqa: E731
        alg_ts = 16.0 * n + 4 * (H + 1) * (W + 1) * 4.0
        keep = [None]

        def ts_public():
            keep[0] = E.events_to_timestamp_image_torch(xd, yd, td_, pud, sensor_size=(H, W))
        if ts_call():
            call_ms = tiled._time_ms(ts_call, 20)
            res["events_to_timestamp_image_torch"] = {
                "call_ms": round(call_ms, 4), "Mevents_per_s": round(n / call_ms / 1e3, 1),
                "roofline_frac": round(alg_ts / (call_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "kernels_ms": {"k_part_sorted": round(tiled._time_ms(lambda: ts_call(_lib.EVK_VOXEL2_PARTITION_ONLY), 20), 4),
                               "k_image_tiles_ts": round(tiled._time_ms(lambda: ts_call(_lib.EVK_VOXEL2_TILES_ONLY), 20), 4)},
                "public_call_ms": round(tiled._time_ms(ts_public, 20), 4),
                "direct_kernel_ms": round(tiled._time_ms(lambda: _lib.call(
                    "evk_timestamp_images_f32", D.ptr(xd), D.ptr(yd), D.ptr(td_), D.ptr(pud), n, H + 1, W + 1, float(W), float(H), 0, 0.0,
                    0.1, D.ptr(planes), None, D.stream()), 3), 4),
                "algorithmic_bytes": alg_ts,
                **({"traffic": pmc_traffic("k_part_sorted", "img_timestamp")[2],
                    "traffic_source": pmc_traffic("k_part_sorted", "img_timestamp")[1]}
                   if pmc_traffic("k_part_sorted", "img_timestamp")[2] else {}),
                "note": "16 B/event + four planes; the tile kernel is bound by its eight 64-bit fixed-point LDS atomics per event (events "
                        "whose normalised time lies outside [-1, 1] add float64 values to a window of their own); direct_kernel_ms = "
                        "evk_timestamp_images_f32, eight global atomics per event"}
        E.check_errors()
    except Exception as e:  # noqa: BLE001
        res["events_to_timestamp_image_torch"] = {"error": repr(e)}
    return res


def bench_image_c1(E, tiled, dev, impl):
    """configs[0]: 1 M events, 240x180, events_to_image (nearest pixel, integer count) -- the plumbing / bit-exactness
    configuration (12 B/event at int32).  (a) the kernels on device-resident int32 columns, against the
    12 B/event HBM roofline: the one-pass path the call takes, and the direct kernel; (b) the public
    numpy-in / numpy-out call (host arrays: PCIe both ways, never a roofline figure)."""
    from event_utils_amd import _lib, _device as D
    n1, H1, W1 = 1_000_000, 180, 240
    rng = np.random.default_rng(0)
    xi = rng.integers(0, W1, n1).astype(np.int64)
    yi = rng.integers(0, H1, n1).astype(np.int64)
    pi = (rng.integers(0, 2, n1) * 2 - 1).astype(np.int64)
    xd, yd, pd = (torch.from_numpy(a.astype(np.int32)).to(dev) for a in (xi, yi, pi))
    canvas = torch.zeros((H1 + 1, W1 + 1), dtype=torch.int32, device=dev)
    oob = torch.zeros(1, dtype=torch.int32, device=dev)

    def kernel():
        _lib.call("evk_image_nearest_i32", D.ptr(xd), D.ptr(yd), D.ptr(pd), n1, H1 + 1, W1 + 1, D.ptr(canvas), D.ptr(oob),
                  D.stream())
    d_ms = tiled._time_ms(kernel, 50)
    k = _image_kernel_times(tiled, "i32", (xd, yd, pd), n1, H1 + 1, W1 + 1, torch.int32, 50)
    k_ms = k["call_ms"] if k else d_ms
    E.events_to_image(xi, yi, pi, sensor_size=(H1, W1))
    t0 = time.perf_counter()
    for _ in range(5):
        img = E.events_to_image(xi, yi, pi, sensor_size=(H1, W1))
    pub_ms = (time.perf_counter() - t0) / 5 * 1e3
    alg = 12.0 * n1 + (H1 + 1) * (W1 + 1) * 4.0
    return {"workload": "configs[0]: 1M events, 240x180, events_to_image nearest (int32 accumulate, bit-exact)",
            "kernel": "k_part_sorted + k_image_tiles_n (one-pass partition + LDS tiles, int32 cells)", "kernel_ms": round(k_ms, 4),
            "kernels_ms": {a: round(b, 4) for a, b in k.items()} if k else None,
            "Mevents_per_s": round(n1 / k_ms / 1e3, 1), "algorithmic_bytes": alg,
            "roofline_frac": round(alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "direct_kernel_ms": round(d_ms, 4),
            "bound": "two launches of fixed cost (~10 us each) at this size; the direct kernel (one global int32 atomic per event, "
                     "~21 G/s: profiles/r01_direct_atomics_probe.json) takes direct_kernel_ms",
            "public_numpy_call_ms": round(pub_ms, 3), "public_numpy_call_note": "host int64 arrays in, float64 image out",
            "checksum_ok": bool(int(img.sum()) == int(pi.sum()))}


def pmc_traffic(kernel, tag):
    """HBM bytes per launch of `kernel` (and of the whole call) from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json, newest round first: separate --pmc passes for reads and writes of this same workload, gfx950 corrections
    applied as MI355X_MICROARCH.md prescribes; tools/profile_round.sh).  PMC counters cannot be collected from inside the
    timed process, so this is the recorded measurement of the workload `tag`; None when the profile is absent."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % rnd)
        if not os.path.isfile(path):
            continue
        try:
            prof = json.load(open(path)).get(tag, {})
            for name, v in prof.get("kernels", {}).items():
                if kernel.split("(")[0] in name:
                    return (v["hbm_bytes_per_launch_corrected"],
                            "profiles/%s_pmc_traffic.json[%s] (rocprofv3 --pmc, per launch)" % (rnd, tag), prof.get("whole_call_bytes"))
        except Exception:
            pass
    return None, None, None


def _time_evals(obj, w, ev, prm, size, reps=10):
    res = {}
    for name, fn in (("f", obj.evaluate_function), ("grad", obj.evaluate_gradient)):
        for _ in range(2):
            fn(prm, ev, None, None, None, w, size, 1.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(prm, ev, None, None, None, w, size, 1.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        n = len(ev)
        res[name + "_evals_per_s"] = round(1.0 / dt, 2)
        res[name + "_ms"] = round(dt * 1e3, 4)
        res[name + "_Mevents_per_s"] = round(n / dt / 1e6, 1)
        res[name + "_hbm_frac"] = round((16.0 * n) / dt / 1e9 / HBM_PEAK_GBS, 4)
        # device time of one evaluation: the same calls enqueued back to back without reading the scalars back (the host
        # then never waits), between ONE pair of HIP events -- the difference to *_ms is the host's share (first-launch
        # latency out of an idle queue, the poll of the pinned result, Python)
        obj.enqueue_only = True
        try:
            fn(prm, ev, None, None, None, w, size, 1.0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4 * reps):
                fn(prm, ev, None, None, None, w, size, 1.0)
            e1.record()
            torch.cuda.synchronize()
            res[name + "_device_ms"] = round(e0.elapsed_time(e1) / (4 * reps), 4)
        finally:
            obj.enqueue_only = False
    return res


def structured_scene(seed, n, Hs, Ws, flow=(40.0, -25.0), t_hi=0.1):
    """Moving vertical (+1) / horizontal (-1) edges: a scene whose contrast is maximised at `flow`."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.uniform(0.0, t_hi, n))
    vert = rng.random(n) < 0.5
    ex = rng.choice(np.arange(40, Ws - 40, 24), n) + rng.normal(0, 0.3, n)
    ey = rng.choice(np.arange(40, Hs - 40, 24), n) + rng.normal(0, 0.3, n)
    x0 = np.where(vert, ex, rng.uniform(30, Ws - 30, n))
    y0 = np.where(vert, rng.uniform(30, Hs - 30, n), ey)
    x = (x0 + (t - t[-1]) * flow[0]).astype(np.float32)
    y = (y0 + (t - t[-1]) * flow[1]).astype(np.float32)
    return x, y, t.astype(np.float32), np.where(vert, 1.0, -1.0).astype(np.float32)


def bench_cmax(E, DeviceEvents, dev, impl):
    """configs[2]: 10 M events, 640x480, warp + IWE + variance (and + analytic gradient) evaluations/s.
    configs[3]: 50 M events, 1280x720: evaluations/s and a full optimize() loop (BFGS iterations/s)."""
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast
    out = {}
    x, y, t, p = synth(2, N_PER_GPU, 0.0, 0.1, real_xy=True)
    ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    obj, w = E.variance_objective(), E.linvel_warp()
    obj.sensor_size, obj.impl = (H, W), impl
    c3 = {"workload": "configs[2]: 10M events, 640x480, get_iwe(linvel)+blur+variance; params (30,-20) px/s"}
    c3.update(_time_evals(obj, w, ev, np.array([30.0, -20.0]), (H, W)))
    out.update(c3)
    del ev
    try:    # the same evaluation the way the reference composes it on the CPU, on a bounded sample
        out["cpu_baseline"] = cmax_cpu_baseline(x, y, t, p)
    except Exception as e:  # noqa: BLE001
        out["cpu_baseline"] = {"error": repr(e)}
    # ---- configs[3]: 50 M events, 1280x720, full optimize() ----
    H4, W4, n4 = 720, 1280, 50_000_000
    x, y, t, p = structured_scene(3, n4, H4, W4)
    ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
    obj = E.variance_objective()
    obj.sensor_size, obj.impl = (H4, W4), impl
    c4 = {"workload": "configs[3]: 50M events, 1280x720, moving-edge scene (true flow (40,-25) px/s)"}
    c4.update(_time_evals(obj, w, ev, np.array([30.0, -20.0]), (H4, W4), reps=5))
    bks = list(ev._buckets.values())
    c4["plan"] = {"records": "compact 8 B" if bks and bks[0].iwe_flag else "full 16 B (sub-pixel coordinates)",
                  "structured_scene": bool(bks and bks[0].structured),
                  "accumulators": "64-bit fixed point, odd LDS pitches"}
    # the same scene as an event camera delivers it (integer pixel coordinates): the bucketed records compact to 8 bytes
    evp = DeviceEvents.from_arrays(np.floor(x), np.floor(y), t, p, precision="f32")
    px = _time_evals(obj, w, evp, np.array([30.0, -20.0]), (H4, W4), reps=5)
    bks = list(evp._buckets.values())
    c4["sensor_pixel_events"] = {"f_ms": px["f_ms"], "grad_ms": px["grad_ms"],
                                 "records": "compact 8 B" if bks and bks[0].iwe_flag else "full 16 B"}
    del evp
    for mode, numeric, exact in (("bfgs_numeric_grads(reference default)", True, True),
                                 ("bfgs_analytic_consistent_grad", False, False)):
        o = E.variance_objective()
        o.sensor_size, o.impl, o.reference_exact = (H4, W4), impl, exact
        cnt = {"f": 0, "g": 0, "fg": 0, "it": 0}
        f0, g0, it0, fg0 = o.evaluate_function, o.evaluate_gradient, o.iter_update, o.evaluate_function_and_gradient
        fn0 = o.evaluate_function_and_numeric_gradient

        def fgw(*a, **k):
            cnt["fg"] += 1
            return fg0(*a, **k)

        def fnw(*a, **k):
            cnt["fg"] += 1
            return fn0(*a, **k)
        o.evaluate_function_and_numeric_gradient = fnw

        def fw(*a, **k):
            cnt["f"] += 1
            return f0(*a, **k)

        def gw(*a, **k):
            cnt["g"] += 1
            return g0(*a, **k)

        def iw(*a, **k):
            cnt["it"] += 1
            return it0(*a, **k)
        o.evaluate_function, o.evaluate_gradient, o.iter_update, o.evaluate_function_and_gradient = fw, gw, iw, fgw
        import warnings
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            argmax = optimize_contrast(ev, None, None, None, w, o, numeric_grads=numeric, blur_sigma=1.0, img_size=(H4, W4))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        iters = max(cnt["it"] - 1, 1)      # the first iter_update is the explicit call before the optimiser
        c4[mode] = {"argmax": [round(float(a), 3) for a in np.asarray(argmax, dtype=float)], "seconds": round(dt, 4),
                    "bfgs_iters": iters, "f_evals": cnt["f"], "grad_evals": cnt["g"],
                    "value_and_grad_evals(one pass each)": cnt["fg"],
                    "iters_per_s": round(iters / dt, 2),
                    "event_passes_per_s": round((cnt["f"] + cnt["g"] + cnt["fg"]) / dt, 2)}
        # the same optimisation COLD: a fresh DeviceEvents on the same device columns -- the bucketing of the events by tile
        # (k_tile_hist / scan / scatter) and, for streams beyond the Infinity Cache, the compaction of the records are inside
        # the timed region (`seconds` above starts with the events already bucketed by the evaluations timed before it)
        cold = ev.fresh_view()
        o2 = E.variance_objective()
        o2.sensor_size, o2.impl, o2.reference_exact = (H4, W4), impl, exact
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            argmax_cold = optimize_contrast(cold, None, None, None, w, o2, numeric_grads=numeric, blur_sigma=1.0, img_size=(H4, W4))
        torch.cuda.synchronize()
        c4[mode]["optimize_cold_s"] = round(time.perf_counter() - t0, 4)
        c4[mode]["optimize_cold_same_argmax"] = bool(np.allclose(np.asarray(argmax_cold, dtype=float), np.asarray(argmax, dtype=float),
                                                                 atol=1e-6))
        del cold
        # the same optimisation with optimizer='evk_bfgs' (events_cmax.evk_bfgs: three step lengths per pass, two passes per
        # iteration, instead of scipy's strong-Wolfe search)
        o3 = E.variance_objective()
        o3.sensor_size, o3.impl, o3.reference_exact = (H4, W4), impl, exact
        best = None
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            a3 = optimize_contrast(ev, None, None, None, w, o3, optimizer="evk_bfgs", numeric_grads=numeric, blur_sigma=1.0,
                                   img_size=(H4, W4))
            torch.cuda.synchronize()
            dt3 = time.perf_counter() - t0
            best = dt3 if best is None else min(best, dt3)
        c4[mode]["evk_bfgs"] = {"seconds": round(best, 4), "argmax": [round(float(v), 3) for v in np.asarray(a3, dtype=float)],
                                "speedup_vs_fmin_bfgs": round(dt / best, 2)}
    out["c4"] = c4
    del ev
    torch.cuda.empty_cache()
    # ---- optimizer='evk_bfgs' WARM and COLD at both cmax configurations (round 6): every real optimize() call
    #      (events_cmax.py:348-368 on a fresh window of events) is cold -- the bucketing of the events by IWE tile is inside it
    try:
        out["evk_bfgs"] = bench_evk_bfgs(E, DeviceEvents, impl)
    except Exception as e:  # noqa: BLE001
        out["evk_bfgs"] = {"error": repr(e)}
    return out


def bench_from_host(E, x, y, t, p, B, H, W):
    """The PCIe-inclusive rate of the headline call (never `value`): events_to_voxel_torch on HOST tensors -- what a caller
    who keeps the reference's CPU tensors pays -- 160 MB of columns up, the 6 MB grid back, per call; pageable memory (what
    torch.from_numpy gives) and pinned memory.  Median of 5 calls after 2 warm-up calls."""
    res = {"workload": "configs[1] columns as CPU float32 tensors in, CPU grid out (160 MB up + 6.1 MB down per call)"}
    cols = [torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)) for a in (x, y, t, p)]
    for kind in ("pageable", "pinned"):
        c = cols if kind == "pageable" else [a.pin_memory() for a in cols]
        ts = []
        for k in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            g = E.events_to_voxel_torch(c[0], c[1], c[2], c[3], B, sensor_size=(H, W))
            torch.cuda.synchronize()
            if k >= 2:
                ts.append(time.perf_counter() - t0)
        assert g.device.type == "cpu"
        ms = float(np.median(ts)) * 1e3
        res[kind] = {"ms_per_call": round(ms, 3), "Mevents_per_s": round(len(x) / ms / 1e3, 1),
                     "host_link_GB_per_s": round((16.0 * len(x) + 4.0 * B * H * W) / ms / 1e6, 1)}
        del c
    return res


def bench_evk_bfgs(E, DeviceEvents, impl):
    """optimize_contrast(..., optimizer='evk_bfgs') with the consistent analytic gradient on the moving-edge scene at configs[2]
    size (10 M events, 640x480), configs[3] size (50 M, 1280x720) and on a 1 M-event window: `warm_seconds` = best of 3 on a
    DeviceEvents whose events are already bucketed (what rounds 4-5 reported; `warm_seconds_python_loop`: the same iteration
    driven from Python, rounds 4-6's form), `cold_seconds` = median of 3 runs each on a FRESH DeviceEvents over the
    same device columns (bucketing, record compaction and every first-use allocation inside the timed region)."""
    import warnings
    from event_utils_amd.contrast_max.events_cmax import optimize_contrast
    res = {}
    w = E.linvel_warp()
    from event_utils_amd import tiled
    for tag, (n, Hs, Ws) in (("c3_10M_640x480", (N_PER_GPU, H, W)), ("c4_50M_1280x720", (50_000_000, 720, 1280)),
                             ("window_1M_640x480", (1_000_000, H, W))):
        x, y, t, p = structured_scene(3, n, Hs, Ws)
        ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")

        def run(e):
            o = E.variance_objective()
            o.sensor_size, o.impl, o.reference_exact = (Hs, Ws), impl, False
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                a = optimize_contrast(e, None, None, None, w, o, optimizer="evk_bfgs", numeric_grads=False, blur_sigma=1.0,
                                      img_size=(Hs, Ws))
            torch.cuda.synchronize()
            passes[0] = getattr(o, "native_passes", None)
            return time.perf_counter() - t0, np.asarray(a, dtype=float)
        passes = [None]
        run(ev)                                   # first use of every buffer / code object
        warm = min(run(ev)[0] for _ in range(3))
        tiled.FORCE["native_bfgs"] = False        # A/B: the same iteration as a Python loop over the bound evaluation calls
        try:
            run(ev)
            warm_py = min(run(ev)[0] for _ in range(3))
        finally:
            tiled.FORCE["native_bfgs"] = True
        cold, arg = [], None
        for _ in range(3):
            fresh = ev.fresh_view()
            dt, arg = run(fresh)
            cold.append(dt)
            del fresh
        res[tag] = {"loop": "inside the library (evk_cmax_bfgs_variance_tiled_f32)", "warm_seconds": round(warm, 5),
                    "event_passes": passes[0],
                    "warm_seconds_python_loop": round(warm_py, 5), "cold_seconds": round(float(np.median(cold)), 5),
                    "cold_runs": [round(v, 5) for v in cold], "cold_minus_warm_ms": round((float(np.median(cold)) - warm) * 1e3, 3),
                    "argmax_cold": [round(float(v), 3) for v in arg], "true_flow": [40.0, -25.0]}
        del ev
        torch.cuda.empty_cache()
    return res


def bench_c5(E, DeviceEvents, dist, rank, world, dev, impl):
    """configs[4]: 50 M events per GPU (400 M on 8), 1280x720, event-sharded: every rank holds one time slice of the
    stream; voxel grid (5 bins) and IWE (+dIWE) are all-reduced over RCCL, the blur / variance / gradient run replicated.
    Timed like the main metric: barrier + synchronize on both sides, max over ranks."""
    from event_utils_amd.distributed import shard_objective
    from event_utils_amd.representations.voxel_grid import _voxel_f32_device
    H5, W5, n5, B5 = 720, 1280, 50_000_000, 5
    span = 0.1 / world
    ok = torch.ones(1, device=dev)
    try:    # the part that can fail on one rank only (memory): agree on it before any further collective
        rng = np.random.default_rng(40 + rank)
        x = rng.uniform(1, W5 - 1, n5).astype(np.float32)
        y = rng.uniform(1, H5 - 1, n5).astype(np.float32)
        t = np.sort(rng.uniform(rank * span, (rank + 1) * span, n5)).astype(np.float32)
        p = (rng.integers(0, 2, n5) * 2 - 1).astype(np.float32)
        ev = DeviceEvents.from_arrays(x, y, t, p, precision="f32")
        grid = torch.empty((B5, H5, W5), dtype=torch.float32, device=dev)
        xi, yi = ev.x.floor(), ev.y.floor()           # voxelisation takes pixel coordinates
        ends = (float(t[0]), float(t[-1]))
        del x, y, t, p
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        ok.zero_()
        err = repr(e)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if ok.item() == 0:
        return {"error": "setup failed on at least one rank" + (": " + err if "err" in locals() else "")}
    lo = torch.tensor([ends[0]], device=dev)
    hi = torch.tensor([ends[1]], device=dev)
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    t_first, t_last = float(lo.item()), float(hi.item())
    obj, w = shard_objective(E.variance_objective(), t_last), E.linvel_warp()
    obj.sensor_size, obj.impl = (H5, W5), impl
    prm = np.array([30.0, -20.0])

    def voxel():
        _voxel_f32_device(xi, yi, ev.t, ev.p, B5, (H5, W5), t_first, t_last, out=grid, check=False, impl=impl, fresh=True)
        dist.all_reduce(grid, op=dist.ReduceOp.SUM)

    res = {"workload": "configs[4]: %d x 50M events, 1280x720, event-sharded, RCCL all-reduce of the grids" % world}

    def rows(fn):   # the same evaluation with the row-sharded post-pass (all-to-all of row blocks + 8-double all-reduce)
        def run():
            os.environ["EVK_SHARDED_POST"] = "rows"
            try:
                return fn()
            finally:
                os.environ["EVK_SHARDED_POST"] = "replicated"
        return run
    f_eval = lambda: obj.evaluate_function(prm, ev, None, None, None, w, (H5, W5), 1.0)      # noqa: E731
    g_eval = lambda: obj.evaluate_gradient(prm, ev, None, None, None, w, (H5, W5), 1.0)      # noqa: E731
    # the parts of each line: this rank's kernels alone (same global reference time, no collective) and the collective
    # alone (a zero buffer of the exchanged shape), so that a loss of linearity can be attributed to one of them
    from event_utils_amd import distributed as DD
    lobj = E.variance_objective()
    lobj.sensor_size, lobj.impl, lobj.t_ref = (H5, W5), impl, t_last
    zg = torch.zeros((B5, H5, W5), dtype=torch.float32, device=dev)
    z1 = torch.zeros((1, H5 + 1, W5 + 1), dtype=torch.float32, device=dev)
    z3 = torch.zeros((3, H5 + 1, W5 + 1), dtype=torch.float32, device=dev)

    def voxel_rsag():
        _voxel_f32_device(xi, yi, ev.t, ev.p, B5, (H5, W5), t_first, t_last, out=grid, check=False, impl=impl, fresh=True)
        DD.reduce_scatter_all_gather_sum_(grid)
    parts = (("voxel_rsag", voxel_rsag, 10),
             ("voxel_compute", lambda: _voxel_f32_device(xi, yi, ev.t, ev.p, B5, (H5, W5), t_first, t_last, out=grid,
                                                         check=False, impl=impl, fresh=True), 10),
             ("voxel_allreduce", lambda: dist.all_reduce(zg, op=dist.ReduceOp.SUM), 10),
             ("voxel_reduce_scatter_all_gather", lambda: DD.reduce_scatter_all_gather_sum_(zg), 10),
             ("f_compute", lambda: lobj.evaluate_function(prm, ev, None, None, None, w, (H5, W5), 1.0), 10),
             ("f_allreduce", lambda: dist.all_reduce(z1, op=dist.ReduceOp.SUM), 10),
             ("grad_compute", lambda: lobj.evaluate_gradient(prm, ev, None, None, None, w, (H5, W5), 1.0), 10),
             ("grad_allreduce", lambda: dist.all_reduce(z3, op=dist.ReduceOp.SUM), 10))
    for name, fn, reps in (("voxel", voxel, 10), ("f", f_eval, 10), ("grad", g_eval, 10),
                           ("f_rows_post", rows(f_eval), 10), ("grad_rows_post", rows(g_eval), 10)) + parts:
        for _ in range(2):
            fn()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        dt = torch.tensor([(time.perf_counter() - t0) / reps], device=dev, dtype=torch.float64)
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        dt = float(dt.item())
        res[name + "_ms"] = round(dt * 1e3, 4)
        if name in ("voxel", "f", "grad", "f_rows_post", "grad_rows_post", "voxel_rsag"):
            res[name + "_Mevents_per_s"] = round(n5 * world / dt / 1e6, 1)
        if name in ("f", "grad", "f_rows_post", "grad_rows_post"):
            res[name + "_evals_per_s"] = round(1.0 / dt, 2)
    res["parts"] = ("*_compute_ms: this rank's kernels alone (replicated post-pass included for f / grad, which then also "
                    "return their scalars to the host); *_allreduce_ms / voxel_reduce_scatter_all_gather_ms: the collective "
                    "alone on a zero buffer of the exchanged shape (grid %d B, IWE %d B, IWE+dIWE %d B); voxel_rsag = the "
                    "voxel line with EVK_VOXEL_COLLECTIVE=rsag" % (zg.numel() * 4, z1.numel() * 4, z3.numel() * 4))
    return res


def cmax_cpu_baseline(x, y, t, p, m=2_000_000):
    """variance objective / gradient the way the reference composes them (numpy warp + mask, torch index_put_ splat,
    scipy gaussian_filter; oracle/reference_torch_cpu.py, pinned to the golden vectors), all host threads, on the first
    m events of the configs[2] stream."""
    from oracle import reference_torch_cpu as T
    xs, ys, ts, ps = (a[:m].astype(np.float64) for a in (x, y, t, p))
    prm = np.array([30.0, -20.0])
    res = {"unit": "Mevents/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": "first %d events of the configs[2] stream, 640x480, median of 5 runs" % m}
    for name, fn in (("f", T.variance_f), ("grad", T.variance_grad)):
        fn(prm, xs, ys, ts, ps, (H, W), (H, W), 1.0)
        ts_ = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn(prm, xs, ys, ts, ps, (H, W), (H, W), 1.0)
            ts_.append(time.perf_counter() - t0)
        res[name + "_Mevents_per_s"] = round(m / float(np.median(ts_)) / 1e6, 2)
    return res


def cpu_baseline(x, y, t, p):
    """The reference's numpy CPU path (events_to_voxel, voxel_grid.py:184-217) as restated in oracle/reference_np.py,
    timed on this box's host, one thread, on the same 10 M-event workload (bounded: 1 warm-up on 1 M, 5 timed reps)."""
    from oracle import reference_np as R
    xi, yi = x.astype(np.int64), y.astype(np.int64)
    t64, p64 = t.astype(np.float64), p.astype(np.float64)
    m = 1_000_000
    R.events_to_voxel(xi[:m], yi[:m], t64[:m], p64[:m], B, sensor_size=(H, W))
    reps, best = 5, []
    for _ in range(reps):
        t0 = time.perf_counter()
        R.events_to_voxel(xi, yi, t64, p64, B, sensor_size=(H, W))
        best.append(time.perf_counter() - t0)
    dt = float(np.median(best))
    res = {"value": round(len(x) / dt / 1e6, 2), "unit": "Mevents/s", "cores": 1, "kind": "port",
           "sample": "oracle numpy events_to_voxel (reference numpy path), %d events, 640x480x5, median of %d runs, "
                     "%.2f s each; host has %d logical cores" % (len(x), reps, dt, os.cpu_count())}
    try:    # best-effort CPU: the reference's torch path (B index_put_ passes) with every host thread torch will use
        from oracle import reference_torch_cpu as T
        cols = [torch.from_numpy(a) for a in (x, y, t, p)]
        T.events_to_voxel_torch(*(c[:m] for c in cols), B, sensor_size=(H, W))
        best = []
        for _ in range(reps):
            t0 = time.perf_counter()
            T.events_to_voxel_torch(*cols, B, sensor_size=(H, W))
            best.append(time.perf_counter() - t0)
        dtm = float(np.median(best))
        res["multithreaded"] = {"value": round(len(x) / dtm / 1e6, 2), "unit": "Mevents/s", "cores": torch.get_num_threads(),
                                "kind": "port", "sample": "oracle torch-CPU events_to_voxel_torch (reference torch path, "
                                "index_put_ accumulate), same %d events, median of %d runs, %.2f s each" % (len(x), reps, dtm)}
    except Exception as e:  # noqa: BLE001
        res["multithreaded"] = {"error": repr(e)}
    return res


if __name__ == "__main__":
    main()
