"""The reference's own octree through the C ABI (rh_octree_*: buildoctree / OctreeRefinery / iswithinrectangle /
octreedepth / getnthcell, src/octree.jl:11-22, 158-244; findleaf as fitting.jl:397 uses it): held against the reference's
known answers (test/octree.jl, transcribed in tests/golden/reference_known_answers.json) and, cell by cell, against the
oracle's restatement.  The build and the queries are host code: no GPU needed; the enabled-cell gather is the GPU test."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import ransac_jl_amd as R
from oracle import oracle as orc
from ransac_jl_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")))


def test_iswithinrectangle_reference_cases(golden):   # test/octree.jl:10-114
    g = golden["iswithinrectangle"]
    assert len(g["cases"]) == 38
    for p, exp in g["cases"]:
        assert R.iswithinrectangle((g["origin"], g["widths"]), p) == bool(exp), p


def test_octree_grid_known_answers(golden):            # test/octree.jl:116-140
    g = golden["octree_grid"]
    n, d = g["grid_n"], g["divide_by"]
    ps = np.array([[i / d, j / d, k / d] for i in range(n) for j in range(n) for k in range(n)], dtype=np.float64)
    root = R.buildoctree(ps)
    assert R.octreedepth(root) == g["expected_octree_depth"]
    leaf = R.findleaf(root, ps[g["query_point_1based"] - 1])
    assert leaf.data.depth == g["expected_leaf_depth"] and leaf.isleaf()
    # getnthcell: the cell itself at its own depth, its ancestors above, the root at 1, nothing below or at n < 1
    assert R.getnthcell(leaf, leaf.data.depth) == leaf
    assert R.getnthcell(leaf, 1) == root and R.getnthcell(leaf, 2) == leaf.parent
    for lv in g["getnthcell_nothing_levels"]:      # test/octree.jl:133-136: levels below 1 and beyond the leaf's depth give nothing
        assert R.getnthcell(leaf, lv) is None
    assert R.getnthcell(leaf, -3) is None
    assert np.array_equal(root.data.incellpoints, np.arange(1, len(ps) + 1))     # the root holds every index (octree.jl:240)
    assert root.parent is None and len(root.children) == 8


@pytest.mark.parametrize("seed,n,shift", [(1, 3000, 0.0), (2, 20000, 7.5), (3, 500, -3.0)])
def test_octree_equals_the_oracles_cell_by_cell(seed, n, shift):
    """random clouds (with a non-zero minimum corner: the root then spans [minV, minV + maxV], Q2): every cell along the
    path of 200 query points has the oracle's depth and point list; the trees have the same depth"""
    rng = np.random.default_rng(seed)
    ps = rng.uniform(0, 10, size=(n, 3)) + shift
    ps[: n // 10] = np.round(ps[: n // 10], 1)          # points on cell faces
    root = R.buildoctree(ps)
    ot = orc.Octree(ps)
    assert R.octreedepth(root) == ot.depth()
    for q in rng.integers(0, n, 200):
        d, path = ot.findleaf(ps[q])
        leaf = R.findleaf(root, ps[q])
        assert leaf.data.depth == d
        cell = leaf
        for node in reversed(path):
            assert np.array_equal(cell.data.incellpoints, ot.node_points(node))
            cell = cell.parent
        assert cell is None
    # Q3: a child keeps vmin < p <= vmax -- the points on the minimum faces of the root fall out of every child
    kept = np.concatenate([c.data.incellpoints for c in root.children])
    o, w = root.boundary
    on_min_face = np.where((ps <= o).any(axis=1))[0] + 1
    assert len(on_min_face) >= 1 and not np.isin(on_min_face, kept).any()


@pytest.mark.parametrize("seed,n,shift", [(4, 4000, 0.0), (5, 25000, 1000.0), (6, 800, -3.0)])
def test_float32_octree_equals_the_oracles_binary32_tree(seed, n, shift):
    """buildoctree on Float32 vertices (a Float32 cloud, octree.jl:102-109): corners, divisions, child widths and the
    vmin < p <= vmax tests are binary32 operations -- cell by cell the oracle's binary32 tree, and (with the large shift,
    where a binary32 division rounds visibly) NOT the binary64 tree of the same points widened"""
    rng = np.random.default_rng(seed)
    ps = (rng.uniform(0, 10, size=(n, 3)) * 1.2345 + shift * 1.0001).astype(np.float32)
    ps[: n // 10] = np.round(ps[: n // 10], 1)
    root = R.buildoctree(ps)
    ot = orc.Octree(ps)
    assert R.octreedepth(root) == ot.depth()
    for q in rng.integers(0, n, 200):
        d, path = ot.findleaf(ps[q].astype(np.float64))
        leaf = R.findleaf(root, ps[q].astype(np.float64))
        assert leaf.data.depth == d
        cell = leaf
        for node in reversed(path):
            assert np.array_equal(cell.data.incellpoints, ot.node_points(node))
            cell = cell.parent
        assert cell is None
    o, w = root.boundary
    assert np.array_equal(np.asarray(o), ps.min(axis=0).astype(np.float64)) and np.array_equal(np.asarray(w), ps.max(axis=0).astype(np.float64))
    if shift == 1000.0:   # the binary64 tree of the same points widened: its divisions carry bits a binary32 sum rounds away
        wide = R.buildoctree(ps.astype(np.float64))
        ndiff = 0
        for q in rng.integers(0, n, 50):
            a, b = R.findleaf(root, ps[q].astype(np.float64)), R.findleaf(wide, ps[q].astype(np.float64))
            (oa, wa), (ob, wb) = a.boundary, b.boundary
            ndiff += not (np.array_equal(np.asarray(oa), np.asarray(ob)) and np.array_equal(np.asarray(wa), np.asarray(wb)))
        assert ndiff > 0


def test_float32_cloud_carries_the_binary32_tree():
    """(no device needed for the tree itself; the cloud is only built where there is one)"""
    import ransac_jl_amd.api as api
    ps = np.random.default_rng(2).uniform(0, 1, size=(300, 3)).astype(np.float32)
    t32, t64 = api._Octree(ps), api._Octree(ps.astype(np.float64))
    assert t32.vertices.dtype == np.float32 and t64.vertices.dtype == np.float64
    assert R.lib().rh_octree_build_f32(None, 5, C.byref(C.c_void_p())) != 0


def test_octree_pc_property_and_bad_arguments():
    import ransac_jl_amd.api as api
    ps = np.random.default_rng(0).uniform(0, 1, size=(100, 3))
    root = R.buildoctree(ps)
    out = C.c_int32()
    assert R.lib().rh_octree_getnthcell(root._t._h, 10 ** 6, 1, C.byref(out)) != 0
    assert R.lib().rh_octree_build(None, 5, C.byref(C.c_void_p())) != 0
    assert hasattr(api.RANSACCloud, "octree")


@pytest.mark.gpu
def test_cell_enabled_gather_on_the_device():
    """enabled_inds = cell.data.incellpoints[pc.isenabled[cell.data.incellpoints]] (fitting.jl:405-407)"""
    from ransac_jl_amd import synth
    xyz, nrm, truth = synth.make_cloud(30_000, ["plane", "sphere"], 0.3, seed=3)
    subs = synth.make_subsets(30_000, 2, seed=3)
    pc = R.RANSACCloud(xyz, nrm, subs)
    rng = np.random.default_rng(1)
    en = rng.random(30_000) < 0.6
    pc.set_enabled(en)
    root = pc.octree
    assert R.octreedepth(pc) == orc.Octree(xyz).depth()
    cells = [root] + root.children + [R.findleaf(root, xyz[i]) for i in rng.integers(0, 30_000, 20)]
    for cell in cells:
        pts = cell.data.incellpoints
        assert np.array_equal(R.cell_enabled_points(pc, cell), pts[en[pts - 1]])
