"""The device's octree sampler (sampler.hip) finds cells and picks points with a cell directory and bracketed 8-ary
searches; the host sampler and the oracle use plain binary searches.  The two families live in one header
(csrc/fit_shared.h, OctView) and must return the same positions -- a lower bound and the r-th set bit are unique.
The library's host-only self-check compares them on synthetic Morton orders (no GPU needed); the end-to-end octree
parity tests (test_parity_gpu.py) then compare whole runs with the oracle, draw counts included."""
import ctypes as C

import pytest

import ransac_jl_amd as R
from ransac_jl_amd import _lib as L


@pytest.mark.parametrize("n,seed", [(1, 1), (63, 2), (64, 3), (65, 4), (1000, 5), (4096, 6), (50_000, 7), (200_003, 8)])
def test_directory_and_bracketed_searches_equal_the_plain_ones(n, seed):
    bad = C.c_int64(-1)
    L.check(L.lib("diag").rh_dbg_oct_search_selftest(n, seed, 4000, C.byref(bad)))
    assert bad.value == 0


def test_selftest_rejects_bad_arguments():
    bad = C.c_int64(0)
    with pytest.raises(R.RansacHipError):
        L.check(L.lib("diag").rh_dbg_oct_search_selftest(0, 1, 10, C.byref(bad)))
