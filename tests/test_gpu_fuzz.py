"""GPU (-m gpu): a short, seeded slice of tools/fuzz_parity.py -- the public calls against the oracle on random sensor sizes,
event counts around the boundaries of the one-pass paths, hot pixels, every kind of polarity / time-stamp column, every
EVK_IMPL, error semantics.  (The long runs are recorded in profiles/r04_fuzz_parity.txt; what they found is pinned by
dedicated tests: test_deterministic_mode_at_small_event_counts_views_and_constant_time_stamps,
test_rms_objective_of_a_handful_of_events, the float64 image in test_f11_gather_contrast_and_timestamp_images.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


@pytest.mark.parametrize("kind,seed0,cases", [("voxel", 100, 60), ("image", 200, 60), ("native", 300, 30), ("iwe", 400, 40),
                                              ("objective", 500, 40), ("windows", 600, 40), ("misc", 700, 60),
                                              ("errors", 800, 80), ("prims", 900, 60), ("search", 1000, 8)])
def test_random_cases_agree_with_the_oracle(kind, seed0, cases):
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        import fuzz_parity as F
    finally:
        sys.argv = argv
    fn = {"voxel": F.case_voxel, "image": F.case_image, "native": F.case_native, "iwe": F.case_iwe, "objective": F.case_objective,
          "windows": F.case_windows, "misc": F.case_misc, "errors": F.case_errors, "prims": F.case_prims,
          "search": F.case_search}[kind]
    failed = []
    for seed in range(seed0, seed0 + cases):
        desc, err = fn(np.random.default_rng(910_000 + seed))
        if err is not None:
            failed.append("seed %d: %s -> %s" % (seed, desc, err))
    assert not failed, "\n".join(failed)
