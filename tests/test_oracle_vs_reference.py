"""CPU, build container only: the committed fixtures tests/golden/*.npz are what the REAL reference computes.
Every generator script under oracle/ (make_golden*.py) is re-run against /root/reference -- imported through
oracle/ref_loader.py -- with its `save` redirected into memory, and every array of every fixture must come out bit for bit
as committed.  Together with tests/test_oracle_golden.py (oracle == fixtures) this pins the oracle to the reference.
Skipped where /root/reference does not exist (the GPU box): nothing under tests/ reads the reference at run time there."""
import importlib
import os

import numpy as np
import pytest

REFERENCE = "/root/reference"
GENERATORS = ("make_golden", "make_golden_native", "make_golden_search", "make_golden_windows", "make_golden_classes")

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "lib")),
                                reason="the reference checkout is not on this machine")


@pytest.fixture(scope="module")
def regenerated():
    import warnings
    import torch
    threads = torch.get_num_threads()
    made = {}

    def capture(name, **arrs):
        made[name] = {k: np.asarray(v) for k, v in arrs.items()}
    import oracle.make_golden as base
    mods = [importlib.import_module("oracle." + g) for g in GENERATORS]
    saved = [(m, m.save) for m in mods]
    try:
        for m in mods:
            m.save = capture
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for m in mods:
                m.main()
    finally:
        for m, fn in saved:
            m.save = fn
        torch.set_num_threads(threads)
    assert base.save is saved[0][1]
    return made


def test_every_committed_fixture_is_regenerated(regenerated):
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    committed = sorted(f[:-4] for f in os.listdir(golden) if f.endswith(".npz"))
    assert committed == sorted(regenerated), (committed, sorted(regenerated))


def test_fixtures_reproduce_bit_for_bit_from_the_reference(regenerated):
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    checked = 0
    for name, arrs in sorted(regenerated.items()):
        g = np.load(os.path.join(golden, name + ".npz"))
        assert sorted(g.files) == sorted(arrs), name
        for k in g.files:
            a, b = g[k], arrs[k]
            assert a.dtype == b.dtype and a.shape == b.shape, (name, k, a.dtype, b.dtype, a.shape, b.shape)
            same = np.array_equal(a, b, equal_nan=True) if a.dtype.kind in "fc" else np.array_equal(a, b)
            assert same, (name, k)
            checked += 1
    assert checked > 250


def test_timestamp_image_with_a_nan_pixel_matches_the_reference():
    """Round 6: scipy.stats.rankdata (image.py:371) propagates a NaN pixel to every rank; the oracle's dense rank (np.unique)
    is pinned to that behaviour directly against the real class (no fixture: two 12-pixel images)."""
    from oracle import ref_loader, reference_np as R
    ref = ref_loader.load()
    for nanpix in (True, False):
        a, b = ref.image.TimestampImage((3, 4)), R.TimestampImage((3, 4))
        for o in (a, b):
            o.image[0, 0] = 5.0
            o.image[2, 1] = -1.0
            if nanpix:
                o.image[1, 2] = np.nan
        ga, gb = a.get_image(), b.get_image()
        assert np.array_equal(ga, gb, equal_nan=True) and bool(np.isnan(ga).all()) == nanpix
