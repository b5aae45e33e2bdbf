"""Parity tests proper: the HIP path (through the C-ABI) against the golden vectors produced
by the reference and against the CPU oracle, on the same inputs.

Bars (BASELINE.json north_star): voxel membership, counts, blob order bit exact; float
sums within 1e-5 relative -- the assertions below hold the tighter REL = 1e-9.
"""
import io
import os

import numpy as np
import pytest

from conftest import VOXEL_CASES, GOLDEN, ROOT, load_case, blobs_from_record, crs_set

pytestmark = pytest.mark.gpu

REL = 1e-9


def device_blobs(bl):
    st = bl.stats()
    crs, off = bl.voxels()
    return [{"crs": crs[off[i]:off[i + 1]], "n": int(st["n"][i]), "totalDensity": st["totalDensity"][i], "centroid": st["centroid"][i],
             "coordCenter": st["coordCenter"][i], "volume": st["volume"][i], "firstKey": int(st["firstKey"][i]), "group": int(st["group"][i])}
            for i in range(len(st["n"]))]


def assert_blob_equal(g, w, rel=REL):
    assert crs_set(g["crs"]) == crs_set(w["crs"])
    assert g["n"] == len(w["crs"])
    if abs(w["totalDensity"]) > 1e-9:
        assert g["totalDensity"] == pytest.approx(w["totalDensity"], rel=rel)
        assert np.allclose(g["centroid"], w["centroid"], rtol=rel, atol=1e-9)
    assert np.allclose(g["coordCenter"], w["coordCenter"], rtol=rel, atol=1e-9)
    assert g["volume"] == pytest.approx(w["volume"], rel=rel)


@pytest.fixture(scope="module", params=VOXEL_CASES)
def case(request, gpu_ctx):
    from pdb_eda_amd import ccp4
    z, header, grid = load_case(request.param)
    dm = ccp4.parse(io.BytesIO(z["ccp4_bytes"].tobytes()), request.param, ctx=gpu_ctx)
    return request.param, z, dm


def test_stats_and_sum_of_abs(case):
    _, z, dm = case
    # == , not approx: the device reproduces numpy's summation tree (the default cutoffs are float32(mean + k std))
    assert dm.meanDensity == float(z["mean"])
    assert dm.stdDensity == float(z["std"])
    for cut, want in zip(z["soa_cut"], z["soa"]):
        assert dm.getTotalAbsDensity(float(cut)) == pytest.approx(want, rel=1e-12)


def test_point_density_and_geometry(case):
    _, z, dm = case
    m = dm._map
    assert np.array_equal(m.point_density(z["pt_crs"]), z["pt_density"])
    assert np.array_equal(m.valid_crs(z["pt_crs"]).astype(np.uint8), z["pt_valid"])
    assert np.array_equal(m.crs2xyz(z["pt_crs"]), z["pt_xyz"])                       # bit exact device geometry
    assert np.array_equal(m.xyz2crs(z["x2c_xyz"].astype(np.float64)), z["x2c_crs"])
    assert dm.getPointDensityFromCrs([int(x) for x in z["pt_crs"][7]]) == z["pt_density"][7]


@pytest.mark.parametrize("tag", ["p30", "n30", "p15", "n20"])
def test_full_map_blobs_vs_reference(case, tag):
    _, z, dm = case
    cut = float(z["full_%s_cut" % tag])
    bl = dm._map.full_blobs(cut)
    got = device_blobs(bl)
    want = blobs_from_record(z, "full_" + tag)
    assert len(got) == len(want)
    for g, w in zip(got, want):          # the reference's emission order
        assert_blob_equal(g, w)
    # DensityBlob surface
    blobs = dm.createFullBlobList(cut)
    assert [b.numVoxels for b in blobs] == [len(w["crs"]) for w in want]
    if blobs:
        assert blobs[0].crsList == crs_set(want[0]["crs"])


def test_full_map_fused_and_labels(case):
    from oracle import oracle as ora
    name, z, dm = case
    cp, cn = float(z["full_p30_cut"]), float(z["full_n30_cut"])
    green, red = dm._map.full_blobs_pm(cp, cn, labels=True)
    for bl, tag in ((green, "p30"), (red, "n30")):
        got, want = device_blobs(bl), blobs_from_record(z, "full_" + tag)
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert_blob_equal(g, w)
    _, header, grid = load_case(name)
    o = ora.Oracle(header, grid)
    for bl, cut in ((green, cp), (red, cn)):
        lab = bl.labels(dm._map.unique_shape)
        assert np.array_equal(lab, o.full_blobs(cut, labels=True)["labels"])
    # lazily computed labels of a non-fused list agree too
    single = dm._map.full_blobs(cp)
    assert np.array_equal(single.labels(dm._map.unique_shape), green.labels(dm._map.unique_shape))
    # ... and those of a NEGATIVE list on its own (one volume, signed -1: its labels must not lose the volume-0 count that the
    # red list of a fused job sheds), eager and lazy
    red_labels = red.labels(dm._map.unique_shape)
    for with_labels in (True, False):
        lone = dm._map.full_blobs(cn, labels=with_labels)
        assert np.array_equal(lone.labels(dm._map.unique_shape), red_labels)
        assert np.array_equal(lone.stats()["n"], red.stats()["n"])
    assert dm.createFullBlobList(0.0) is None


@pytest.mark.parametrize("ci", [0, 1, 2, 3])
def test_sphere_blobs_per_atom(case, ci):
    """findAberrantBlobs for single coordinates, batched: one group per atom."""
    _, z, dm = case
    cut = float(z["sph_cut"][ci])
    n = len(z["sph_xyz"])
    bl = dm._map.sphere_blobs(z["sph_xyz"].astype(np.float64), z["sph_radius"].astype(np.float32), np.arange(n + 1), cut)
    got = device_blobs(bl)
    off = z["sph%d_off" % ci]
    boff = np.concatenate([[0], np.cumsum(z["sphb%d_nblobs" % ci])])
    want_blobs = blobs_from_record(z, "sphb%d" % ci)
    for a in range(n):
        ga = [g for g in got if g["group"] == a]
        want_set = crs_set(z["sph%d_crs" % ci][off[a]:off[a + 1]])
        assert set().union(*[crs_set(g["crs"]) for g in ga]) == want_set if ga else want_set == set()
        wa = want_blobs[boff[a]:boff[a + 1]]
        assert len(ga) == len(wa)
        for g, w in zip(ga, wa):      # single-atom lists keep the reference's emission order
            assert_blob_equal(g, w)


def test_sphere_unions_and_region_sums(case):
    _, z, dm = case
    xyz = z["sph_xyz"].astype(np.float64)
    for tag, radii in (("s", np.full(len(xyz), 1.9, dtype=np.float32)), ("l", z["sph_radius"].astype(np.float32))):
        goff = np.arange(0, len(xyz) + 1, 4)
        for ci in range(3):
            cut = float(z["sph_cut"][ci])
            got = device_blobs(dm._map.sphere_blobs(xyz, radii, goff, cut))
            for gi in range(6):
                want = crs_set(z["uni_%s%d_g%d" % (tag, ci, gi)])
                gg = [g for g in got if g["group"] == gi]
                have = set().union(*[crs_set(g["crs"]) for g in gg]) if gg else set()
                assert have == want
                key = "unib_%s%d_g%d" % (tag, ci, gi)
                if want:
                    wb = blobs_from_record(z, key)
                    assert len(gg) == len(wb)
                    for g, w in zip(sorted(gg, key=lambda b: min(crs_set(b["crs"]))), wb):
                        assert_blob_equal(g, w)
        # regional sums: identity sum(blob.totalDensity) == masked sum over the sphere union
        cut = float(z["sph_cut"][1])
        pos, neg, cnt, valid = dm._map.region_sums(xyz, radii, goff, cut)
        for gi in range(6):
            assert cnt[gi] == len(z["uni_%s0_g%d" % (tag, gi)])
            wp = z["unib_%s1_g%d_total" % (tag, gi)].sum() if ("unib_%s1_g%d_total" % (tag, gi)) in z.files else 0.0
            wn = z["unib_%s2_g%d_total" % (tag, gi)].sum() if ("unib_%s2_g%d_total" % (tag, gi)) in z.files else 0.0
            assert pos[gi] == pytest.approx(wp, rel=REL, abs=1e-12)
            assert neg[gi] == pytest.approx(wn, rel=REL, abs=1e-12)
            if tag == "s":
                assert bool(valid[gi]) == bool(z["uni_valid_g%d" % gi])
    # per-atom validity (testValidXyz)
    n = len(xyz)
    _, _, cnt, valid = dm._map.region_sums(xyz, z["sph_radius"].astype(np.float32), np.arange(n + 1), 0.1)
    assert np.array_equal(valid.astype(np.uint8), z["sph_valid"])
    off0 = z["sph0_off"]
    assert np.array_equal(cnt, np.diff(off0))


def test_list_blobs_merge_and_overlap(case):
    _, z, dm = case
    want = blobs_from_record(z, "full_p15")[:12]
    # createBlobList on an explicit voxel list reproduces the clustering
    allcrs = np.concatenate([w["crs"] for w in want]) if want else np.zeros((0, 3), np.int32)
    got = device_blobs(dm._map.list_blobs(allcrs))
    assert sorted(map(lambda g: tuple(sorted(crs_set(g["crs"]))), got)) == sorted(tuple(sorted(crs_set(w["crs"]))) for w in want)
    if len(want) >= 2:
        from pdb_eda_amd.ccp4 import DensityBlob
        a = DensityBlob.fromCrsList(want[0]["crs"], dm)
        b = DensityBlob.fromCrsList(want[1]["crs"], dm)
        assert a.totalDensity == pytest.approx(want[0]["totalDensity"], rel=REL)
        assert np.allclose(a.centroid, want[0]["centroid"], rtol=REL)
        a.merge(b)
        assert len(a.crsList) == len(want[0]["crs"]) + len(want[1]["crs"])
        assert a.totalDensity == pytest.approx(want[0]["totalDensity"] + want[1]["totalDensity"], rel=REL)
    blobs = want
    sets = [w["crs"] for w in blobs]
    off = np.concatenate([[0], np.cumsum([len(s) for s in sets])]).astype(np.int64)
    if len(z["ovl_pairs"]):
        res = dm._ctx.test_overlap(np.concatenate(sets), off, z["ovl_pairs"][:, 0], z["ovl_pairs"][:, 1])
        assert np.array_equal(res.astype(np.uint8), z["ovl_res"])


def test_symmetry_and_nearest_atom(case, gpu_ctx):
    _, z, dm = case
    idx, sym, xyz = gpu_ctx.symmetry_atoms(z["sph_xyz"].astype(np.float64), z["sym_rot"], np.asarray(dm.header.orthoMat, dtype=np.float64),
                                           z["sym_box"][0], z["sym_box"][1])
    assert np.array_equal(idx, z["sym_atom"])
    assert np.array_equal(sym, z["sym_sym"])
    assert np.allclose(xyz, z["sym_xyz"], rtol=0, atol=1e-12)
    from scipy.spatial.distance import cdist
    cen = blobs_from_record(z, "full_p15")
    if cen:
        c = np.array([b["centroid"] for b in cen])
        d = cdist(c, z["sym_xyz"])
        gi, gd = gpu_ctx.nearest_atom(c, z["sym_xyz"])
        assert np.array_equal(gi, np.argmin(d, axis=1))
        assert np.allclose(gd, d.min(axis=1), rtol=1e-15, atol=0)


# ---- edge cases the reference's test design calls for (painted cubes with analytic answers) -----

def _dm(grid, gpu_ctx, **kw):
    from pdb_eda_amd import ccp4, synthetic
    ns, nr, nc = grid.shape
    spec = synthetic.MapSpec(ncrs=(nc, nr, ns), **kw)
    return ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, grid)), "edge", ctx=gpu_ctx)


def test_painted_cubes(gpu_ctx):
    """The reference's own KAT idea (tests/test_ccp4.py:75-108): cubes of +-1 with analytic blobs."""
    g = np.zeros((40, 40, 40), dtype=np.float32)
    g[11:13, 11:13, 11:13] = 1
    g[7:10, 11:14, 7:10] = 1
    g[11:13, 7:9, 7:9] = -1
    g[7:10, 7:10, 11:14] = -1
    dm = _dm(g, gpu_ctx, spacing=0.5)
    centre = dm.header.crs2xyzCoord([10, 10, 10])
    green = dm.findAberrantBlobs(centre, 5, 0.5)
    assert sorted(b.numVoxels for b in green) == [8, 27]
    big = [b for b in green if b.numVoxels == 27][0]
    assert np.allclose(big.centroid, dm.header.crs2xyzCoord([8, 12, 8]))
    assert big.totalDensity == 27 and big.volume == pytest.approx(27 * dm.header.unitVolume)
    red = dm.findAberrantBlobs(centre, 5, -0.5)
    assert sorted(b.numVoxels for b in red) == [8, 27]
    assert sorted(b.totalDensity for b in red) == [-27, -8]
    full_g, full_r = dm.createFullBlobLists(0.5)
    assert sorted(b.numVoxels for b in full_g) == [8, 27] and sorted(b.numVoxels for b in full_r) == [8, 27]
    # merge KAT (tests/test_ccp4.py:111-131): 8 + 8 + 1 voxels touching through (10,10,10)
    g2 = np.zeros((40, 40, 40), dtype=np.float32)
    g2[11:13, 11:13, 11:13] = 1
    dm2 = _dm(g2, gpu_ctx, spacing=0.5)
    b1 = dm2.findAberrantBlobs(centre, 5, 0.5)[0]
    g3 = np.zeros((40, 40, 40), dtype=np.float32)
    g3[8:10, 8:10, 8:10] = 1
    g3[10, 10, 10] = 1
    dm3 = _dm(g3, gpu_ctx, spacing=0.5)
    b2 = dm3.findAberrantBlobs(centre, 5, 0.5)[0]
    assert b2.numVoxels == 9
    assert b1.testOverlap(b2)


def test_empty_full_and_degenerate(gpu_ctx):
    from oracle import oracle as ora
    g = np.zeros((5, 6, 70), dtype=np.float32)
    dm = _dm(g, gpu_ctx)
    assert dm.createFullBlobList(0.5) == []
    g[:] = 1.0
    dm = _dm(g, gpu_ctx)
    blobs = dm.createFullBlobList(0.5)
    assert len(blobs) == 1 and blobs[0].numVoxels == g.size
    assert dm.createFullBlobList(1.0)[0].numVoxels == g.size      # inclusive threshold (Q2)
    assert dm.createFullBlobList(np.nextafter(np.float32(1), np.float32(2))) == []
    one = _dm(np.ones((1, 1, 1), dtype=np.float32), gpu_ctx)
    b = one.createFullBlobList(0.5)
    assert len(b) == 1 and b[0].crsList == {(0, 0, 0)}
    # checkerboard in c (maximum number of runs per word), isolated in r/s -> every voxel its own blob
    cb = np.zeros((9, 9, 130), dtype=np.float32)
    cb[::2, ::2, ::2] = 1.0
    dm = _dm(cb, gpu_ctx)
    blobs = dm.createFullBlobList(0.5)
    assert len(blobs) == int(cb.sum()) and all(b.numVoxels == 1 for b in blobs)
    keys = [b.firstKey for b in blobs]
    assert keys == sorted(keys)
    # dense checkerboard in 3-D is ONE 26-connected blob
    cb2 = np.zeros((8, 8, 130), dtype=np.float32)
    idx = np.indices(cb2.shape).sum(axis=0)
    cb2[idx % 2 == 0] = 1.0
    dm = _dm(cb2, gpu_ctx)
    assert [b.numVoxels for b in dm.createFullBlobList(0.5)] == [int(cb2.sum())]
    # a long diagonal snake: deep union-find chains, runs crossing word boundaries
    sn = np.zeros((3, 200, 200), dtype=np.float32)
    for i in range(200):
        sn[i % 3, i, i] = 2.0
        sn[(i + 1) % 3, i, 199 - i] = 2.0
    dm = _dm(sn, gpu_ctx)
    o = ora.Oracle(dm.header, sn)
    want = o.full_blobs(1.0, labels=True)
    bl = dm._map.full_blobs(1.0)
    assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"])
    assert np.array_equal(bl.stats()["n"], want["n"])


def test_face_words_with_many_runs(gpu_ctx):
    """Tile faces crossed by words of more runs than their component record holds (7 + the run at the last bit): the
    cross-tile merge looks those up by run id.  Combs (every other voxel) against solid / shifted combs / other signs, across
    the r, s and c faces and their diagonals, in tiles that still fit LDS."""
    from oracle import oracle as ora
    g = np.zeros((24, 24, 600), dtype=np.float32)          # [s][r][c]: 3 x 3 x 3 tiles (256 x 8 x 8)
    comb = np.zeros(64, dtype=np.float32); comb[::2] = 1.0
    g[3, 7, 64:128] = comb;  g[3, 8, 64:128] = 1.0          # r face: 32 runs under one solid run
    g[5, 15, 0:64] = 1.0;    g[5, 16, 0:64] = comb          # r face, comb on the later side, word 0
    g[7, 2, 128:192] = comb; g[8, 3, 129:193] = comb        # s face, diagonal in r, combs shifted by one: a zigzag
    g[15, 20, 192:256] = comb; g[16, 20, 192:256] = np.roll(comb, 1)   # s face: comb ends at bit 254 / shifted comb at 255
    g[16, 21, 256:300] = 1.0                                # ... and the c face behind the shifted comb's last voxel
    g[10, 12, 449:512:2] = -1.0; g[10, 13, 512:520] = -1.0  # c face of the second tile column, other sign, diagonal in r
    g[11, 12, 448:512] = -comb;  g[11, 11, 512] = -1.0
    g[20, 7, 300:364] = comb; g[20, 7, 364:420] = 0.0; g[20, 8, 301:365:2] = 1.0   # comb against comb shifted across two words
    dm = _dm(g, gpu_ctx)
    o = ora.Oracle(dm.header, g)
    green, red = dm._map.full_blobs_pm(0.5, -0.5, labels=True)
    for bl, c in ((green, 0.5), (red, -0.5)):
        want = o.full_blobs(c, labels=True)
        st = bl.stats()
        assert np.array_equal(st["n"], want["n"])
        assert np.array_equal(st["firstKey"], want["firstKey"])
        assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"])
        assert np.allclose(st["totalDensity"], want["totalDensity"], rtol=REL)
        c = bl.counters()
        assert c["unit_tiles_runs"] == 0 and c["unit_tiles_comps"] == 0   # (the LDS path and its face records, not the fallback)


@pytest.mark.parametrize("shape,seed,nsd", [((128, 128, 128), 5, 1.5), ((96, 100, 200), 6, 3.0), ((61, 67, 130), 8, 1.0),
                                            ((40, 48, 256), 9, 0.3),     # dense: most tiles overflow LDS -> unit-tile fallback
                                            ((24, 40, 600), 10, 0.8)])   # rows wider than one tile (c tiles) + dense
def test_random_maps_vs_oracle(gpu_ctx, shape, seed, nsd):
    """Seeded smooth-noise maps at sizes the reference cannot cluster (O(N^2)) but the oracle can."""
    from oracle import oracle as ora
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise(shape, seed, 1.5)
    dm = _dm(g, gpu_ctx)
    o = ora.Oracle(dm.header, g)
    mean, std = dm.meanDensity, dm.stdDensity
    assert mean == pytest.approx(float(np.mean(g, dtype=np.float64)), rel=1e-10, abs=1e-14)
    assert std == pytest.approx(float(np.std(g.astype(np.float64))), rel=1e-12)
    cut = mean + nsd * std
    green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
    for bl, c in ((green, cut), (red, -cut)):
        want = o.full_blobs(c, labels=True)
        st = bl.stats()
        assert np.array_equal(st["n"], want["n"])
        assert np.array_equal(st["firstKey"], want["firstKey"])
        assert np.allclose(st["totalDensity"], want["totalDensity"], rtol=REL)
        assert np.allclose(st["centroid"], want["centroid"], rtol=REL, atol=1e-9)
        assert np.allclose(st["coordCenter"], want["coordCenter"], rtol=REL, atol=1e-9)
        assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"])
        crs, off = bl.voxels()
        assert off[-1] == st["n"].sum()
        lab = want["labels"]
        assert np.array_equal(lab[crs[:, 2], crs[:, 1], crs[:, 0]], np.repeat(np.arange(len(st["n"])), st["n"]))


@pytest.mark.parametrize("n_rows", [1, 2, 3, 5, 9, 16, 17, 18, 32, 33, 127, 128, 129, 511, 576, 577, 1023, 1024, 1025, 1031, 2047, 2049, 3100])
def test_mean_std_are_numpys_to_the_bit_on_every_tail_shape(gpu_ctx, n_rows):
    """np.mean / np.std of the voxels (ccp4.py:343-363) depend on numpy's summation TREE: 8192-element calls added in order, each a pairwise
    sum over 128-element leaves with eight accumulators, the array's tail by the general recursion n2 = (n / 2) & ~7.  k_np_final builds the
    tail's tree as a heap in LDS (round 5): grids of 1 x n_rows x 8 (+ 3) voxels walk the tail through leaves shorter than 8, one leaf, a
    first split, lengths that are no multiple of 8, tails just below and above a full call, and several calls + a tail.  Equality, not closeness."""
    rng = np.random.default_rng(1000 + n_rows)
    for nc in (8, 11):
        g = (rng.standard_normal((1, n_rows, nc)) * 3.0 + 0.7).astype(np.float32)
        dm = _dm(g, gpu_ctx)
        flat = g.reshape(-1).astype(np.float64)
        assert dm.meanDensity == float(np.mean(flat)), (n_rows, nc, g.size)
        assert dm.stdDensity == float(np.std(flat)), (n_rows, nc, g.size)


def test_recycled_arenas_and_edge_overflow(monkeypatch):
    """Every device arena poisoned with 0xFF when handed out (a kernel that trusts recycled memory to be zero
    shows up at once) and a cross-tile pair buffer far too small (shards overflow: their tail is united on the
    spot) -- same answers, including the dense fallbacks and the grids wider than one tile."""
    from oracle import oracle as ora
    from pdb_eda_amd import _native, synthetic
    monkeypatch.setenv("PDBEDA_DEBUG_POISON", "1")
    monkeypatch.setenv("PDBEDA_DEBUG_EDGE_CAP", "4096")
    ctx = _native.Context(0)
    for shape, seed, nsd in (((24, 40, 600), 10, 0.8), ((40, 48, 256), 9, 1.5), ((17, 23, 70), 3, 1.0), ((24, 40, 600), 11, 2.0)):
        g = synthetic.smooth_noise(shape, seed, 1.5)
        dm = _dm(g, ctx)
        o = ora.Oracle(dm.header, g)
        cut = dm.meanDensity + nsd * dm.stdDensity
        for rep in range(2):   # the second pass runs in the arena the first one gave back
            green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
            for bl, c in ((green, cut), (red, -cut)):
                want = o.full_blobs(c, labels=True)
                st = bl.stats()
                assert np.array_equal(st["n"], want["n"])
                assert np.array_equal(st["firstKey"], want["firstKey"])
                assert np.allclose(st["totalDensity"], want["totalDensity"], rtol=REL)
                assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"])
            green.free(); red.free()
        # the grouped (sphere / list) engine on poisoned arenas
        xyz = np.array([[3.0, 4.0, 5.0], [6.5, 7.25, 3.0]])
        got = dm._map.sphere_blobs(xyz, np.array([1.5, 2.0]), np.array([0, 1, 2]), cut)
        st = got.stats()
        for k in range(2):
            assert int(st["n"][st["group"] == k].sum()) == len(o.sphere_crs(xyz[k], [1.5, 2.0][k], cut))
        got.free()


def test_sparse_maps_write_every_label_line(monkeypatch):
    """Label volumes of SPARSE maps (pdb_eda's own 2.5 - 4 sigma: most 128-byte lines of the volume hold no label) on 0xFF-poisoned
    arenas: a line that no kernel wrote shows at once.  Widths that are whole lines, whole 4-voxel groups only, and neither, partial
    tiles on every axis, a map without a significant voxel, both launch forms (fused, one sign).  (Written for round 6's experiment
    that zeroed the empty lines of sparse tiles in k_tile_label -- measured, slower, not merged: DESIGN section 4 -- and kept: the
    sparse regime had no poisoned-arena case.)"""
    from oracle import oracle as ora
    from pdb_eda_amd import _native, synthetic
    monkeypatch.setenv("PDBEDA_DEBUG_POISON", "1")
    ctx = _native.Context(0)
    cases = [((40, 48, 256), 21, 3.0), ((17, 23, 200), 22, 2.5), ((9, 30, 36), 23, 2.0), ((33, 20, 100), 24, 3.0), ((12, 9, 130), 25, 2.5),
             ((24, 17, 600), 26, 3.0), ((24, 17, 516), 27, 3.5), ((10, 10, 67), 28, 2.5), ((70, 64, 128), 29, 4.0), ((16, 16, 256), 30, 50.0)]
    for shape, seed, nsd in cases:
        g = synthetic.smooth_noise(shape, seed, 1.5)
        dm = _dm(g, ctx)
        o = ora.Oracle(dm.header, g)
        cut = dm.meanDensity + nsd * dm.stdDensity
        for rep in range(2):   # (the second pass runs in the arena the first one gave back, poisoned again)
            lists = dm._map.full_blobs_pm(cut, -cut, labels=True) if rep == 0 else (dm._map.full_blobs(cut, labels=True), dm._map.full_blobs(-cut, labels=True))
            for bl, c in zip(lists, (cut, -cut)):
                want = o.full_blobs(c, labels=True)
                assert np.array_equal(bl.stats()["n"], want["n"]), (shape, nsd)
                assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"]), (shape, nsd, rep)
            for bl in lists:
                bl.free()


@pytest.mark.parametrize("nsd", [1.5, 0.5, 3.0])
def test_protein_like_map_vs_oracle(gpu_ctx, nsd):
    """A chain of Gaussian atoms: ONE blob spans the map at 1.5 sigma (hot roots in the global union-find and in the
    record fold), its interior tiles are too dense to park their values in LDS (re-read path) and a one-sign job uses
    the run slots of both signs."""
    from oracle import oracle as ora
    from pdb_eda_amd import ccp4, synthetic
    spec = synthetic.MapSpec(ncrs=(256, 72, 64), spacing=0.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([249, 65, 57]))
    st = synthetic.chain_structure(700, 3, lo, hi)
    g = synthetic.gaussian_sum_grid(header, st, synthetic.synthetic_params()["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=3)
    dm = _dm(g, gpu_ctx, spacing=0.5)
    o = ora.Oracle(dm.header, g)
    cut = dm.meanDensity + nsd * dm.stdDensity
    want = o.full_blobs(cut, labels=True)
    assert want["n"].max() > 10000                     # the chain is one big blob
    for rep in range(2):
        bl = dm._map.full_blobs(cut, labels=True)
        st_ = bl.stats()
        assert np.array_equal(st_["n"], want["n"])
        assert np.array_equal(st_["firstKey"], want["firstKey"])
        assert np.allclose(st_["totalDensity"], want["totalDensity"], rtol=REL)
        assert np.allclose(st_["centroid"], want["centroid"], rtol=REL, atol=1e-9)
        assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"])
        assert bl.counters()["unit_tiles_runs"] == 0 and bl.counters()["unit_tiles_comps"] == 0   # stays on the LDS path
    green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
    assert np.array_equal(green.stats()["n"], want["n"])
    wr = o.full_blobs(-cut)
    assert np.array_equal(red.stats()["n"], wr["n"])


def test_shapes_fuzz(gpu_ctx):
    """Seeded sweep over grid shapes and cutoffs: partial tiles on every axis, 1 .. 5 mask words per row (one, two and
    three tiles along c of equal and unequal width), one-sign and fused jobs, labels on -- all against the oracle."""
    from oracle import oracle as ora
    from pdb_eda_amd import synthetic
    rng = np.random.default_rng(20240607)
    widths = [1, 7, 63, 64, 65, 100, 129, 192, 200, 256, 257, 300, 320, 321, 400, 513]
    for k in range(36):
        nc = int(widths[k % len(widths)])
        nr = int(rng.integers(1, 41))
        ns = int(rng.integers(1, 41 if nc < 300 else 21))
        nsd = float(rng.choice([0.6, 1.0, 1.5, 2.5]))
        g = synthetic.smooth_noise((ns, nr, nc), 100 + k, float(rng.choice([0.8, 1.5, 2.5])))
        dm = _dm(g, gpu_ctx)
        o = ora.Oracle(dm.header, g)
        cut = dm.meanDensity + nsd * dm.stdDensity
        if not np.isfinite(cut) or cut <= 0:
            continue
        lists = dm._map.full_blobs_pm(cut, -cut, labels=True) if k % 3 else (dm._map.full_blobs(cut, labels=True), dm._map.full_blobs(-cut, labels=True))
        for bl, c in zip(lists, (cut, -cut)):
            want = o.full_blobs(c, labels=True)
            st = bl.stats()
            assert np.array_equal(st["n"], want["n"]), (k, (ns, nr, nc), nsd)
            assert np.array_equal(st["firstKey"], want["firstKey"]), (k, (ns, nr, nc), nsd)
            assert np.allclose(st["totalDensity"], want["totalDensity"], rtol=REL), (k, (ns, nr, nc), nsd)
            assert np.allclose(st["centroid"], want["centroid"], rtol=REL, atol=1e-9), (k, (ns, nr, nc), nsd)
            assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"]), (k, (ns, nr, nc), nsd)
        for bl in lists:
            bl.free()


def test_map_combine_download_and_order_statistics(gpu_ctx):
    """pdbeda_map_combine / pdbeda_map_download / pdbeda_abs_select_hist against numpy, and their argument checks."""
    from pdb_eda_amd import _native, synthetic
    a = synthetic.smooth_noise((21, 19, 70), 41, 1.2)
    b = synthetic.smooth_noise((21, 19, 70), 42, 1.0)
    da, db = _dm(a, gpu_ctx), _dm(b, gpu_ctx)
    c = _native.DeviceMap.combine(da._map, db._map, -2.0)
    want = (a.astype(np.float64) - 2.0 * b.astype(np.float64))
    assert np.array_equal(c.download(), want.astype(np.float32))
    assert np.array_equal(da._map.download(), a)
    # order statistics of |a| and |a - 2 b| over the voxels where both are below their cut (here: the whole grid is the unique box)
    cut_a, cut_b = float(np.abs(a).mean() * 1.3), float(np.abs(want).mean() * 1.1)
    keep = (np.abs(a.astype(np.float64)) < cut_a) & (np.abs(want) < cut_b)
    n = int(keep.sum())
    assert da._map.abs_order_statistics(db._map, -2.0, cut_a, cut_b, 0) == n and n > 100
    sa, sc = np.sort(np.abs(a.astype(np.float64))[keep]), np.sort(np.abs(want)[keep])
    ranks = [0, 1, n // 3, n // 2, n - 1]
    assert da._map.abs_order_statistics(db._map, -2.0, cut_a, cut_b, 0, ranks) == [float(sa[r]) for r in ranks]
    assert da._map.abs_order_statistics(db._map, -2.0, cut_a, cut_b, 1, ranks) == [float(sc[r]) for r in ranks]
    # |a| alone (no second map)
    n1 = int((np.abs(a.astype(np.float64)) < cut_a).sum())
    assert da._map.abs_order_statistics(None, 0.0, cut_a, 0.0, 0) == n1
    # argument checks: shapes must agree, which = 1 needs the second map
    small = _dm(synthetic.smooth_noise((5, 6, 7), 43, 1.0), gpu_ctx)
    with pytest.raises(_native.PdbedaError):
        _native.DeviceMap.combine(da._map, small._map, 1.0)
    with pytest.raises(_native.PdbedaError):
        da._map.abs_order_statistics(None, 0.0, cut_a, 0.0, 1)


def _full_size_case(g, gpu_ctx, nsd, **spec_kw):
    """Size-independent properties of a fused green/red labelling + oracle equality of BOTH lists (labels, order, sums)."""
    from oracle import oracle as ora
    dm = _dm(g, gpu_ctx, **spec_kw)
    o = ora.Oracle(dm.header, g)
    mean, std = o.mean_std()
    assert (dm.meanDensity, dm.stdDensity) == (mean, std)                     # numpy's tree, also at 16.8 M voxels
    cut = dm.meanDensity + nsd * dm.stdDensity
    us, ur, uc = dm._map.unique_shape
    gu = g[:us, :ur, :uc]
    green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
    for bl, sign in ((green, 1), (red, -1)):
        st = bl.stats()
        lab = bl.labels((us, ur, uc))
        mask = (gu >= np.float32(cut)) if sign > 0 else (gu <= np.float32(-cut))
        assert np.array_equal(lab >= 0, mask)                                 # every significant voxel labelled, nothing else
        assert st["n"].sum() == mask.sum()
        assert np.array_equal(np.bincount(lab[mask], minlength=len(st["n"])), st["n"])
        assert st["totalDensity"].sum() == pytest.approx(float(gu[mask].astype(np.float64).sum()), rel=1e-10)
        assert (np.diff(st["firstKey"]) > 0).all()                            # reference emission order
        # idempotence: a one-sign job gives the same partition
        again = dm._map.full_blobs(cut if sign > 0 else -cut)
        assert np.array_equal(again.stats()["n"], st["n"])
        want = o.full_blobs(cut if sign > 0 else -cut, labels=True)
        assert np.array_equal(lab, want["labels"])
        assert np.array_equal(st["n"], want["n"]) and np.array_equal(st["firstKey"], want["firstKey"])
        assert np.allclose(st["totalDensity"], want["totalDensity"], rtol=REL)
        assert np.allclose(st["centroid"], want["centroid"], rtol=REL, atol=1e-9)
        assert np.allclose(st["coordCenter"], want["coordCenter"], rtol=REL, atol=1e-9)
    assert green.counters()["unit_tiles_runs"] + green.counters()["unit_tiles_comps"] == 0   # the LDS fast path, not the fallback
    return len(green), len(red)


@pytest.mark.parametrize("nsd", [1.5, 3.0])
def test_full_size_properties(gpu_ctx, nsd):
    """BASELINE configs[1] at its full size (256^3): blue-style 1.5 sigma (~13 % of the voxels set) and the
    sparse green/red regime at 3 sigma (~0.27 %, almost every tile empty): properties + oracle equality, both signs."""
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise((256, 256, 256), 7, 1.5)
    n_green, n_red = _full_size_case(g, gpu_ctx, nsd, spacing=0.4)
    assert n_green > 1000 and n_red > 1000


@pytest.mark.parametrize("nsd", [1.5, 3.0])
def test_full_size_non_orthogonal(gpu_ctx, nsd):
    """SURVEY 8d-2's second variant at full size: gamma = 120 degrees, axis order (2, 1, 3), crsStart != 0 and
    interval > ncrs (so crs2xyz takes the matrix path, centroids go through the skewed basis)."""
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise((256, 256, 256), 11, 1.5)
    _full_size_case(g, gpu_ctx, nsd, interval=(272, 264, 280), crs_start=(-9, 5, 17), axis_order=(2, 1, 3),
                    cell=(105.6, 108.8, 112.0), angles=(90.0, 90.0, 120.0))


def test_full_size_repeating_map(gpu_ctx):
    """ncrs > interval along two axes (the stored box repeats the cell): only the unique box is labelled (ccp4.py:262-269)."""
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise((200, 256, 256), 13, 1.5)
    _full_size_case(g, gpu_ctx, 1.5, interval=(232, 256, 180), cell=(92.8, 102.4, 72.0))


@pytest.mark.timeout(600)
def test_six_times_full_size(gpu_ctx):
    """A 512 x 384 x 512 map (6x BASELINE configs[1]; two c tiles per row, 6144 tiles: more than one round of resident
    workgroups, the tile faces along c in play): the same properties + oracle equality of both lists."""
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise((512, 384, 512), 21, 1.5)
    n_green, n_red = _full_size_case(g, gpu_ctx, 1.5, spacing=0.4)
    assert n_green > 50000 and n_red > 50000


def _equal_to_oracle(dm, grid, cut):
    from oracle import oracle as ora
    o = ora.Oracle(dm.header, grid)
    green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
    for bl, c in ((green, cut), (red, -cut)):
        want = o.full_blobs(c, labels=True)
        st = bl.stats()
        assert np.array_equal(st["n"], want["n"]) and np.array_equal(st["firstKey"], want["firstKey"])
        assert np.allclose(st["totalDensity"], want["totalDensity"], rtol=REL)
        assert np.array_equal(bl.labels(dm._map.unique_shape), want["labels"])
    return green, red


def test_typical_size_arena_and_its_two_overflows(gpu_ctx):
    """Whole-map jobs are carved for what maps need in practice, not for the worst case (round 4): a 256^3 job at +-1.5 sigma
    holds well under 0.6 GB (2.9 GB worst case) and runs once.  A map whose unit tiles need more ids than the job has, and a
    map with more blobs than the job has table rows, raise the device flag, stay inside their arena and are run again in a
    worst-case arena by the first accessor -- same answers as the oracle, `reruns` = 1."""
    from pdb_eda_amd import synthetic
    # (1) the bench's configuration: one run, a small arena
    g = synthetic.smooth_noise((256, 256, 256), 7, 1.5)
    dm = _dm(g, gpu_ctx, spacing=0.4)
    cut = dm.meanDensity + 1.5 * dm.stdDensity
    green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
    c = green.counters()
    assert c["reruns"] == 0 and c["arena_bytes"] < 600e6, c
    assert len(green) > 1000 and len(red) > 1000
    green.free(); red.free()
    # (2) every tile a unit tile, more word-runs than the typical arena has ids for (64 tiles' worth above the tiles' own ranges)
    g = synthetic.smooth_noise((128, 128, 256), 3, 0.7)
    dm = _dm(g, gpu_ctx)
    cut = dm.meanDensity + 0.1 * dm.stdDensity
    green, red = _equal_to_oracle(dm, g, cut)
    c = green.counters()
    assert c["reruns"] == 1 and c["unit_tiles_runs"] + c["unit_tiles_comps"] > 64, c
    assert red.counters()["reruns"] == 1
    green.free(); red.free()
    # (3) a blob per 2 x 2 x 2 cell: more blobs than one table row per 32 keys
    g = np.full((32, 40, 64), -1.0, dtype=np.float32)
    g[::2, ::2, ::2] = 1.0
    g[1::2, 1::2, 1::2] = -3.0
    dm = _dm(g, gpu_ctx)
    green, red = _equal_to_oracle(dm, g, 0.5)
    assert len(green) == 16 * 20 * 32 and green.counters()["reruns"] == 1
    green.free(); red.free()
    # ... and the debug hook that carves the worst case at once gives the same lists in one run
    import os
    from pdb_eda_amd import _native
    os.environ["PDBEDA_DEBUG_WORST_CASE_ARENA"] = "1"
    try:
        ctx = _native.Context(0)
        dm2 = _dm(g, ctx)
        green2, red2 = _equal_to_oracle(dm2, g, 0.5)
        assert green2.counters()["reruns"] == 0
        green2.free(); red2.free()
        ctx.close()
    finally:
        del os.environ["PDBEDA_DEBUG_WORST_CASE_ARENA"]


def test_many_contexts_under_a_small_pool_cap(monkeypatch):
    """Eight contexts on one device, mixed 200^3 - 256^3 whole-map jobs, with the per-context arena pool capped far below what
    the jobs park (PDBEDA_POOL_CAP_MB): freed arenas go back to the driver instead of piling up, every job still answers, and
    the parked total stays bounded."""
    from pdb_eda_amd import _native, ccp4, synthetic
    monkeypatch.setenv("PDBEDA_POOL_CAP_MB", "96")
    ctxs = [_native.Context(0) for _ in range(8)]
    try:
        grids = {n: synthetic.smooth_noise((n, n, n), 100 + n, 1.5) for n in (200, 224, 256)}
        counts = {}
        for rep in range(2):
            for k, ctx in enumerate(ctxs):
                n = (200, 224, 256)[(k + rep) % 3]
                dm = _dm(grids[n], ctx, spacing=0.4)
                cut = dm.meanDensity + 1.5 * dm.stdDensity
                green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
                got = (len(green), len(red), int(green.stats()["n"].sum()))
                assert counts.setdefault(n, got) == got and got[0] > 100
                assert green.counters()["arena_bytes"] < 600e6
                green.free(); red.free()
                dm._map.free()
    finally:
        for ctx in ctxs:
            ctx.close()


def test_wide_tiles_beside_normal_and_unit_tiles(gpu_ctx):
    """A tile with more components than the LDS accumulators hold (256) but few enough word-runs keeps its in-LDS unions (round
    4: a WIDE tile, tile_mode 2 -- its components take ids above the tiles' own ranges and its sums are made 256 components at
    a time); before, it was relabelled run by run.  Here wide tiles (a lattice of isolated voxels: 1 024 components a tile),
    unit tiles (columns on a checkerboard, both signs: more word-runs than LDS holds) and ordinary tiles (smooth noise) share faces, and lines
    along c, r, s and a diagonal tie blobs together across all three kinds -- against the oracle, labels included, both signs."""
    from pdb_eda_amd import synthetic
    ns, nr, nc = 24, 24, 512
    g = (0.3 * synthetic.smooth_noise((ns, nr, nc), 17, 1.5)).astype(np.float32)
    g = np.clip(g, -0.45, 0.45)                                     # nothing significant yet
    noise = synthetic.smooth_noise((ns, nr, nc), 18, 1.5)
    g[:8, :, :] = np.where(np.abs(noise[:8]) > 1.2, np.sign(noise[:8]), g[:8]).astype(np.float32)      # ordinary tiles (s < 8)
    g[8:16, :, :] = 0.0
    g[8:13:2, ::2, ::4] = 1.0                                       # wide tiles: 3 x 4 x 64 isolated voxels a tile
    g[9:12:2, 1::2, 2::8] = -1.0                                    # ... and a red lattice between them (both signs share the run slots)
    g[16:, :, :256] = 0.0
    g[16:, ::2, 0:256:2] = 1.0                                      # unit tiles: 8 x 4 x 128 word-runs a tile and sign (more than the
    g[16:, 1::2, 1:256:2] = -1.0                                    # 4 096 run slots even a dense tile has in LDS)
    g[16::2, 1::2, 256::4] = 1.0                                    # wide again, beside the unit tiles along c
    g[:, 11, 100] = 1.0                                             # a line along s through all three kinds
    g[12, :, 301] = 1.0                                             # ... along r inside the wide band
    g[10, 5, :] = 1.0                                               # ... along c across the c face
    for k in range(24):
        g[k, k, 200 + k] = 1.0                                      # a diagonal: corner contacts across faces
        g[k, 23 - k, 280 + 2 * k] = -1.0
    dm = _dm(g, gpu_ctx)
    green, red = _equal_to_oracle(dm, g, 0.5)
    c = green.counters()
    assert c["wide_tiles"] >= 8 and c["unit_tiles_runs"] >= 3, c      # (a map this small may need the second, worst-case arena for its unit tiles)
    one = dm._map.full_blobs(0.5, labels=True)                      # the one-sign job takes the same paths
    assert np.array_equal(one.stats()["n"], green.stats()["n"]) and one.counters()["wide_tiles"] >= 8
    green.free(); red.free(); one.free()


@pytest.mark.parametrize("nsd", [1.0, 0.6, 0.2])
def test_dense_noise_stays_in_lds(gpu_ctx, nsd):
    """Smooth noise below ~1.1 sigma has 1 500 - 2 700 word-runs a tile: more than the parent table proper (1 408), and its
    sections are far too dense to park their values -- such a tile uses the parked values' LDS as parent slots (up to 4 096
    word-runs) and reads its values from L2 (round 4; a unit tile before: 1.6 ms a step at 1 sigma against 0.14 now).  Against the
    oracle, labels included; no tile may leave the LDS path."""
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise((40, 48, 512), 31, 1.5)
    dm = _dm(g, gpu_ctx)
    cut = dm.meanDensity + nsd * dm.stdDensity
    green, red = _equal_to_oracle(dm, g, cut)
    c = green.counters()
    assert c["unit_tiles_runs"] == 0 and c["unit_tiles_comps"] == 0 and c["reruns"] == 0, c
    assert c["run_ids"] > 0
    green.free(); red.free()


@pytest.mark.parametrize("shape", [(264, 256, 256), (344, 330, 336)])      # groups of 32 and of 64 rank counters
def test_map_beyond_the_one_trip_rank_table(gpu_ctx, shape):
    """A fused job of more than 2^25 keys (more than 256^3 voxels a sign) has rank-counter groups of more than 16 counters: the
    groups' totals are summed once (k_group_counts) and every ranking workgroup reads those instead of the whole counter array
    (round 4: 512^3 took 0.94 ms a step without them, 0.62 with).  Counts, keys, totals and the label volume against the oracle."""
    from pdb_eda_amd import synthetic
    g = synthetic.smooth_noise(shape, 41, 1.5)
    dm = _dm(g, gpu_ctx)
    cut = dm.meanDensity + 2.0 * dm.stdDensity
    green, red = _equal_to_oracle(dm, g, cut)
    assert len(green) > 5000 and len(red) > 5000 and green.counters()["reruns"] == 0
    g2, r2 = dm._map.full_blobs_pm(cut, -cut)           # no labels: k_emit_tiles ranks through the same totals
    assert np.array_equal(g2.stats()["firstKey"], green.stats()["firstKey"]) and np.array_equal(r2.stats()["n"], red.stats()["n"])
    green.free(); red.free(); g2.free(); r2.free()


BORROWED_WORKER = r'''
import io, sys
import numpy as np
import torch                                    # (first: torch brings its own HIP runtime, which wants to be the one that finds the GPU)
torch.zeros(1, device="cuda")
sys.path.insert(0, %(root)r)
from pdb_eda_amd import _native, ccp4, synthetic
ctx = _native.Context(0)
g = synthetic.smooth_noise((40, 44, 72), 31, 1.5)
spec = synthetic.MapSpec(ncrs=(72, 44, 40))
dm = lambda grid: ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, grid)), "b", ctx=ctx)
uploaded = dm(g)
geom = uploaded.header.geometry()
rng = np.random.default_rng(3)
xyz = np.asarray([uploaded.header.crs2xyzCoord([int(c), int(r), int(s)]) for c, r, s in zip(rng.integers(4, 68, 60), rng.integers(4, 40, 60), rng.integers(4, 36, 60))], dtype=np.float64)
radii = np.full(60, 1.4, dtype=np.float32)
off = np.arange(61, dtype=np.int64)
cut = float(uploaded.meanDensity + 1.0 * uploaded.stdDensity)
def spheres(m, c):
    st = m.sphere_blobs(xyz, radii, off, c).stats()
    return st["n"].tolist(), st["totalDensity"].tolist(), st["group"].tolist()
want = spheres(uploaded._map, cut)
t = torch.from_numpy(g).to("cuda")
torch.cuda.synchronize()
borrowed = _native.DeviceMap(ctx, t, geom, device_ptr=t.data_ptr())
assert spheres(borrowed, cut) == want, "first call of a borrowed map"
assert borrowed.stats() == (uploaded.meanDensity, uploaded.stdDensity)
for x, y in zip(borrowed.full_blobs_pm(cut, -cut, labels=True), uploaded._map.full_blobs_pm(cut, -cut, labels=True)):
    sx, sy = x.stats(), y.stats()
    assert np.array_equal(sx["n"], sy["n"]) and np.array_equal(sx["totalDensity"], sy["totalDensity"])
    assert np.array_equal(x.labels(borrowed.unique_shape), y.labels(uploaded._map.unique_shape))
t.mul_(20.0)                                    # rewritten in place: the cached quantum of the blob sums would be too fine for it
torch.cuda.synchronize()
borrowed.invalidate()
scaled = dm((g * np.float32(20.0)).astype(np.float32))
cut20 = float(scaled.meanDensity + 1.0 * scaled.stdDensity)
assert spheres(borrowed, cut20) == spheres(scaled._map, cut20), "after invalidate"
borrowed.free()
print("borrowed ok")
'''


@pytest.mark.timeout(300)
def test_borrowed_device_buffer(tmp_path):
    """pdbeda_map_from_device (the caller's own HBM buffer: how bench.py hands its torch tensor over) and pdbeda_map_invalidate: a borrowed
    map gives what an uploaded one gives -- also when its FIRST call is a per-atom sphere batch (nothing is known about the map yet: the
    quantum of the blob sums is computed behind a wait of its own, while the batch's inputs sit staged in the pinned block) -- and a map
    rewritten in place gives the new contents' results after invalidate().  In a child process: torch must initialise the GPU first."""
    import subprocess
    import sys
    script = tmp_path / "borrowed.py"
    script.write_text(BORROWED_WORKER % {"root": os.path.dirname(os.path.dirname(os.path.abspath(__file__)))})
    proc = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=280)
    assert proc.returncode == 0 and "borrowed ok" in proc.stdout, proc.stderr[-3000:]


ATOM_ENGINE_WORKER = r'''
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
ctx = _native.Context(0)
out = []
for shape, seed, spacing, radii, nsd in (((40, 44, 48), 31, 0.5, (0.8, 1.6, 3.4), 0.3), ((36, 40, 52), 32, 0.35, (1.0, 2.4, 5.2), 1.0), ((30, 30, 30), 33, 0.5, (3.5,), 0.0)):
    g = synthetic.smooth_noise(shape, seed, 1.0)
    spec = synthetic.MapSpec(ncrs=shape[::-1], spacing=spacing)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    m = _native.DeviceMap(ctx, g, header.geometry())
    mean, std = m.stats()
    rng = np.random.default_rng(seed)
    lo, hi = np.array(header.crs2xyzCoord([2, 2, 2])), np.array(header.crs2xyzCoord([shape[2] - 3, shape[1] - 3, shape[0] - 3]))
    xyz = lo + rng.random((40, 3)) * (hi - lo)
    rad = np.array([radii[k %% len(radii)] for k in range(40)], dtype=np.float32)
    bl = m.sphere_blobs(xyz, rad, np.arange(41), mean + nsd * std if nsd else 0.0)
    st = bl.stats()
    crs, off = bl.voxels()
    lists = [sorted(map(tuple, crs[a:b].tolist())) for a, b in zip(off[:-1], off[1:])]
    out.append({"n": st["n"].tolist(), "key": st["firstKey"].tolist(), "group": st["group"].tolist(), "total": [float.hex(float(x)) for x in st["totalDensity"]],
                "centroid": [float.hex(float(x)) for x in st["centroid"].ravel()], "voxels": lists})
    bl.free(); m.free()
json.dump(out, open(%(out)r, "w"))
'''


@pytest.mark.timeout(300)
def test_atom_engine_paths_agree(tmp_path):
    """Round 6: a per-atom sphere batch is labelled by one launch (k_atom_engine).  Its three ways through a volume -- the LDS tables, the
    kernel's own global-table path for a volume beyond them (forced here by shrinking the tables' use: PDBEDA_DEBUG_ATOM_CAPS), and the five
    generic kernels (boxes beyond one word a row or 512 rows, or PDBEDA_ATOM_ENGINE=0) -- give the same blobs to the last bit: counts, first
    keys, groups, sums, centroids, voxel sets.  Radii from 0.8 to 5.2 A at 0.35 / 0.5 A spacing (rows of 6 to 32 voxels, boxes of 36 to
    1 024 rows), cutoffs from 0 (every voxel of the sphere) to 1 sigma of noise (hundreds of runs, dozens of blobs in a box)."""
    import json
    import subprocess
    import sys
    outs = []
    for k, env_extra in enumerate(({}, {"PDBEDA_DEBUG_ATOM_CAPS": "0,0"}, {"PDBEDA_DEBUG_ATOM_CAPS": "24,3"}, {"PDBEDA_ATOM_ENGINE": "0"})):
        out, script = tmp_path / ("out%d.json" % k), tmp_path / ("worker%d.py" % k)
        script.write_text(ATOM_ENGINE_WORKER % {"root": ROOT, "out": str(out)})
        proc = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=280)
        assert proc.returncode == 0, proc.stderr[-3000:]
        outs.append(json.loads(out.read_text()))
    assert sum(len(c["n"]) for c in outs[0]) > 200 and max(max(c["n"]) for c in outs[0]) > 1000
    for k in (1, 2, 3):
        assert outs[k] == outs[0], k
