"""Pin the CPU oracle (oracle/pdbeda_oracle.c) and the product's host-side header math
against golden vectors produced by the reference itself (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import VOXEL_CASES, load_case, blobs_from_record, crs_set
from oracle import oracle as ora

RTOL = 1e-12   # oracle sums run in a different order than CPython's set iteration


@pytest.fixture(scope="module", params=VOXEL_CASES)
def case(request):
    z, header, grid = load_case(request.param)
    return request.param, z, header, grid, ora.Oracle(header, grid)


def test_header_fields_bit_exact(case):
    """DensityHeader derived fields (ref ccp4.py:225-286) -- bit for bit."""
    _, z, h, _, _ = case
    assert np.array_equal(np.asarray(h.origin, dtype=np.float64), z["h_origin"])
    assert np.array_equal(np.asarray(h.orthoMat, dtype=np.float64), z["h_ortho"])
    assert np.array_equal(np.asarray(h.deOrthoMat, dtype=np.float64), z["h_deortho"])
    assert np.float64(h.unitVolume) == z["h_unit_volume"]
    assert np.array_equal(np.asarray(h.gridLength, dtype=np.float64), z["h_grid_length"])
    ints = list(h.ncrs) + list(h.crsStart) + list(h.xyzInterval) + list(h.map2xyz) + list(h.map2crs) + list(h.crsInterval) + list(h.uniqueNcrs)
    assert np.array_equal(np.array(ints, dtype=np.int64), z["h_ints"])


def test_mean_std_numpy_tree(case):
    """meanDensity / stdDensity (ref ccp4.py:343-363): the restated numpy summation tree gives the reference's values
    bit for bit, and numpy's own on sizes that exercise full 8192-blocks plus an irregular tail."""
    _, z, _, grid, o = case
    assert o.mean_std() == (float(z["mean"]), float(z["std"]))
    rng = np.random.default_rng(int(grid.size))
    for n in (7, 129, 8191, 8192 * 3 + 5, 100003):
        a = (rng.standard_normal(n) * 10 ** rng.uniform(-2, 2, n)).astype(np.float32)

        class H(object):
            pass
        m, s = ora.Oracle.__new__(ora.Oracle), None
        m.L, m.density = ora.lib(), a
        assert m.mean_std() == (float(np.mean(a.astype(np.float64))), float(np.std(a.astype(np.float64)))), n


def test_point_density_wrap_contract(case):
    """getPointDensityFromCrs / testValidCrs (ref cutils.pyx:125-167)."""
    _, z, _, _, o = case
    got = np.array([o.point_density(p) for p in z["pt_crs"]])
    assert np.array_equal(got, z["pt_density"])
    valid = np.array([o.valid_crs(p) for p in z["pt_crs"]], dtype=np.uint8)
    assert np.array_equal(valid, z["pt_valid"])


def test_crs2xyz_bit_exact(case):
    """crs2xyzCoord (ref ccp4.py:304-316): unfused mul+add / np.dot accumulation order."""
    _, z, _, _, o = case
    got = np.array([o.crs2xyz(p) for p in z["pt_crs"]])
    assert np.array_equal(got, z["pt_xyz"])


def test_xyz2crs_round_half_even(case):
    """xyz2crsCoord on float32 atom coordinates (ref ccp4.py:288-302, Q5)."""
    _, z, _, _, o = case
    got = np.array([o.xyz2crs(p.astype(np.float64)) for p in z["x2c_xyz"]])
    assert np.array_equal(got, z["x2c_crs"])


def test_sum_of_abs(case):
    _, z, _, _, o = case
    for cut, want in zip(z["soa_cut"], z["soa"]):
        assert o.sum_of_abs(cut) == pytest.approx(want, rel=1e-12)


@pytest.mark.parametrize("tag", ["p30", "n30", "p15", "n20"])
def test_full_map_blobs(case, tag):
    """createFullCrsList + createCrsLists + fromCrsList (ref cutils.pyx:185-203, 44-70; ccp4.py:522-545)."""
    _, z, _, _, o = case
    cut = float(z["full_%s_cut" % tag])
    lst = o.full_crs_list(cut)
    assert np.array_equal(lst, z["full_%s_list" % tag])          # same voxels, same c-major order
    want = blobs_from_record(z, "full_" + tag)
    got = o.blob_list(lst)
    assert len(got) == len(want)
    for g, w in zip(got, want):                                   # same emission order
        assert crs_set(g["crs"]) == crs_set(w["crs"])
        assert g["totalDensity"] == pytest.approx(w["totalDensity"], rel=RTOL)
        assert np.allclose(g["centroid"], w["centroid"], rtol=1e-10, atol=1e-12)
        assert np.allclose(g["coordCenter"], w["coordCenter"], rtol=1e-10, atol=1e-12)
        assert g["volume"] == pytest.approx(w["volume"], rel=RTOL)
    # the O(N) composite the bench's cpu_baseline leg times gives the same blobs
    fb = o.full_blobs(cut, labels=True)
    assert len(fb["n"]) == len(want)
    for i, w in enumerate(want):
        assert fb["n"][i] == len(w["crs"])
        assert fb["totalDensity"][i] == pytest.approx(w["totalDensity"], rel=RTOL)
        assert np.allclose(fb["centroid"][i], w["centroid"], rtol=1e-10, atol=1e-12)
        c, r, s = w["crs"][:, 0], w["crs"][:, 1], w["crs"][:, 2]
        assert (fb["labels"][s, r, c] == i).all()
    assert (fb["labels"] >= 0).sum() == sum(len(w["crs"]) for w in want)


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_sphere_lists_and_blobs(case, ci):
    """getSphereCrsFromXyz (ref cutils.pyx:220-248): box (Q4), strict filter (Q2), fp64 distance."""
    _, z, _, _, o = case
    cut = float(z["sph_cut"][ci])
    off = z["sph%d_off" % ci]
    boff = np.concatenate([[0], np.cumsum(z["sphb%d_nblobs" % ci])])
    want_blobs = blobs_from_record(z, "sphb%d" % ci)
    for a in range(len(z["sph_xyz"])):
        xyz = z["sph_xyz"][a].astype(np.float64)
        got = o.sphere_crs(xyz, z["sph_radius"][a], cut)
        assert np.array_equal(got, z["sph%d_crs" % ci][off[a]:off[a + 1]])   # same voxels in the same order
        wb = want_blobs[boff[a]:boff[a + 1]]
        gb = o.blob_list(got) if len(got) else []
        assert len(gb) == len(wb)
        for g, w in zip(gb, wb):
            assert crs_set(g["crs"]) == crs_set(w["crs"])
            if abs(w["totalDensity"]) > 1e-9:
                assert g["totalDensity"] == pytest.approx(w["totalDensity"], rel=1e-10)


def test_sphere_valid(case):
    _, z, _, _, o = case
    got = np.array([o.valid_xyz(x.astype(np.float64), r) for x, r in zip(z["sph_xyz"], z["sph_radius"])], dtype=np.uint8)
    assert np.array_equal(got, z["sph_valid"])


def test_sphere_unions(case):
    """getSphereCrsFromXyzList (ref cutils.pyx:250-271): set union on raw crs."""
    _, z, _, _, o = case
    for gi in range(6):
        idx = list(range(4 * gi, 4 * gi + 4))
        xyz = z["sph_xyz"][idx].astype(np.float64)
        for tag, rad in (("s", np.full(4, 1.9)), ("l", z["sph_radius"][idx])):
            for ci in range(3):
                want = z["uni_%s%d_g%d" % (tag, ci, gi)]
                got = o.sphere_crs_list(xyz, rad, float(z["sph_cut"][ci]))
                assert np.array_equal(got, want)


def test_overlap(case):
    _, z, _, _, o = case
    blobs = blobs_from_record(z, "full_p15")[:12]
    for (i, j), want in zip(z["ovl_pairs"], z["ovl_res"]):
        assert ora.test_overlap(blobs[i]["crs"], blobs[j]["crs"]) == bool(want)


def test_symmetry_atoms(case):
    """createSymmetryAtoms (ref cutils.pyx:73-103)."""
    _, z, h, _, _ = case
    idx, sym, xyz = ora.symmetry_atoms(z["sph_xyz"].astype(np.float64), z["sym_rot"], np.asarray(h.orthoMat, dtype=np.float64),
                                       z["sym_box"][0], z["sym_box"][1])
    assert np.array_equal(idx, z["sym_atom"])
    assert np.array_equal(sym, z["sym_sym"])
    assert np.allclose(xyz, z["sym_xyz"], rtol=0, atol=1e-12)
