"""densityAnalysis rows (aggregateCloud, region density / discrepancy, blob statistics) on the
MI355X path against golden vectors produced by the reference's DensityAnalysis.
Tolerance: 1e-5 relative (BASELINE.json north_star) -- asserted at 1e-8; counts are exact."""
import io
import json

import numpy as np
import pytest

from conftest import ANALYSIS_CASES, load_analysis_case

pytestmark = pytest.mark.gpu
REL = 1e-8


@pytest.fixture(scope="module", params=ANALYSIS_CASES)
def analysis(request, gpu_ctx):
    from pdb_eda_amd import ccp4, synthetic, densityAnalysis
    z, spec, st, pdb, params = load_analysis_case(request.param)
    densityAnalysis.setGlobals(params)
    dens = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["dens"])), request.param, ctx=gpu_ctx)
    diff = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["diff"])), request.param, ctx=gpu_ctx)
    densityAnalysis._attachCutoffs(dens, diff)
    an = densityAnalysis.DensityAnalysis(request.param, dens, diff, st, pdb)
    return z, an


def test_aggregate_cloud_totals(analysis):
    z, an = analysis
    assert an.densityElectronRatio == pytest.approx(float(z["ratio"]), rel=REL)
    assert an.numVoxelsAggregated == int(z["num_voxels"])
    assert an.totalAggregatedElectrons == pytest.approx(float(z["total_electrons"]), rel=1e-12)
    assert an.totalAggregatedDensity == pytest.approx(float(z["total_density"]), rel=REL)


def test_atom_cloud_descriptions(analysis):
    z, an = analysis
    atoms = an.atomCloudDescriptions
    assert len(atoms) == len(z["acd_chain"])
    for f in ("chain", "residue_number", "residue_name", "atom_name", "atom_type", "num_voxels", "electrons"):
        assert np.array_equal(np.asarray(atoms[f]), z["acd_" + f]), f
    for f in ("density_electron_ratio", "bfactor", "centroid_distance", "centroid_xyz", "adj_density_electron_ratio", "domain_fraction",
              "corrected_fraction", "corrected_density_electron_ratio", "volume"):
        assert np.allclose(np.asarray(atoms[f]), z["acd_" + f], rtol=1e-8, atol=1e-9), f


def test_residue_and_domain_clouds(analysis):
    z, an = analysis
    got = np.array([[r[1]] + [r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.residueCloudDescriptions], dtype=np.float64).reshape(-1, 8)
    want = z["res_rows"]
    assert got.shape == want.shape
    # the reference's order inside a residue depends on CPython set order: compare as sorted multisets
    key = lambda a: a[np.lexsort((a[:, 7], a[:, 2], a[:, 0]))]
    assert np.allclose(key(got), key(want), rtol=1e-8, atol=1e-9)
    got = np.array([[r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.domainCloudDescriptions], dtype=np.float64).reshape(-1, 7)
    want = z["dom_rows"]
    assert got.shape == want.shape
    assert np.allclose(got[np.argsort(got[:, 0])], want[np.argsort(want[:, 0])], rtol=1e-8, atol=1e-9)


def test_medians_and_overlap_counts(analysis):
    z, an = analysis
    want = json.loads(str(z["medians"]))
    assert set(an.medians) == set(want)
    for col, d in want.items():
        for t, v in d.items():
            assert float(an.medians[col][t]) == pytest.approx(v, rel=1e-8, abs=1e-10), (col, t)
    assert dict(an.atomTypeOverlapCompleteness) == json.loads(str(z["overlap_complete"]))
    assert dict(an.atomTypeOverlapIncompleteness) == json.loads(str(z["overlap_incomplete"]))


def test_region_discrepancy_and_density(analysis):
    z, an = analysis
    got = np.array([r[6:] for r in an.calculateAtomRegionDiscrepancies(3.5, 3.0)], dtype=np.float64)
    assert np.allclose(got, z["atom_discrepancy"], rtol=REL, atol=1e-12)
    got = np.array([r[6:] for r in an.calculateAtomRegionDiscrepancies(2.0, 2.5, type="CA")], dtype=np.float64)
    assert np.allclose(got, z["atom_discrepancy_r2"], rtol=REL, atol=1e-12)
    got = np.array([r[5:] for r in an.calculateResidueRegionDiscrepancies(3.5, 3.0)], dtype=np.float64)
    assert np.allclose(got, z["residue_discrepancy"], rtol=REL, atol=1e-12)
    got = np.array([r[6:] for r in an.calculateAtomRegionDensity(1.0, 1.5, useOptimizedRadii=True)], dtype=np.float64)
    assert np.allclose(got, z["atom_density"], rtol=REL, atol=1e-12)
    got = np.array([r[5:] for r in an.calculateResidueRegionDensity(1.2, 1.5)], dtype=np.float64)
    assert np.allclose(got, z["residue_density"], rtol=REL, atol=1e-12)


def test_symmetry_atoms_and_validity(analysis):
    z, an = analysis
    sym = an.symmetryAtoms
    assert len(sym) == int(z["sym_count"])
    for i, tag, xyz, row, ok in zip(z["sym_pick"], z["sym_tags"], z["sym_coords"], z["sym_discrepancy"], z["sym_valid"]):
        assert tuple(sym[i].symmetry) == tuple(int(v) for v in tag)
        assert np.allclose(np.asarray(sym[i].coord, dtype=np.float64), xyz, rtol=0, atol=1e-10)
        res, valid = an.calculateRegionDiscrepancy([sym[i].coord], 3.5, 3.0, testValidCrs=True)
        assert valid == bool(ok)
        assert np.allclose(res, row, rtol=REL, atol=1e-12)
    rows = an.calculateSymmetryAtomRegionDiscrepancies(3.5, 3.0, type="CB")
    assert all(len(r) == len(an.symmetryAtomRegionDiscrepancyHeader) for r in rows)


def test_blob_statistics(analysis):
    z, an = analysis
    for tag, blobs in (("green", an.greenBlobList), ("red", an.redBlobList)):
        stats = an.calculateAtomSpecificBlobStatistics(blobs)
        want = z["blob_%s_num" % tag]
        assert len(stats) == len(want)
        got = np.array([[s[0], s[2], s[3], s[4]] for s in stats], dtype=np.float64).reshape(-1, 4)
        assert np.allclose(got, want, rtol=1e-8, atol=1e-10)
        assert [s[1] for s in stats] == list(z["blob_%s_sign" % tag])
        assert ["%s|%s|%s|%s" % (s[6], s[7], s[8], tuple(int(v) for v in s[9])) for s in stats] == list(z["blob_%s_atom" % tag])
        assert np.allclose(np.array([list(s[11]) for s in stats]).reshape(-1, 3), z["blob_%s_centroid" % tag], rtol=1e-8, atol=1e-9)


def test_blob_lists_are_lazy_sequences(analysis):
    """createFullBlobList & co. return the blobs as a sequence that makes its DensityBlob objects on first access
    (ccp4.DeviceBlobs): the statistics table straight from the device list's columns must be the table built from the objects,
    `+` / `==` / indexing behave as a list's, and the objects of `a + b` are those of `a` and `b`."""
    from pdb_eda_amd import ccp4
    z, an = analysis

    def same(a, b):          # tables of rows whose cells are numbers, strings, tuples and coordinate arrays
        return len(a) == len(b) and all(len(r) == len(q) and all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in zip(r, q)) for r, q in zip(a, b))
    cut = an.diffDensityObj.diffDensityCutoff
    green, red = an.diffDensityObj.createFullBlobLists(cut)
    assert isinstance(green, ccp4.DeviceBlobs) and green.columns() is not None
    both = green + red
    assert isinstance(both, ccp4.DeviceBlobs) and len(both) == len(green) + len(red)
    from_columns = an.calculateAtomSpecificBlobStatistics(both)
    assert both.columns() is not None                       # ... and still no object was made
    objects = list(both)                                    # now they are
    assert green.columns() is None and both.columns() is None
    assert objects[0] is green[0] and objects[len(green)] is red[0] and both[-1] is red[len(red) - 1]
    from_objects = an.calculateAtomSpecificBlobStatistics(both)
    assert same(from_objects, from_columns)
    assert same(an.calculateAtomSpecificBlobStatistics(objects), from_columns)       # a plain list of the same blobs
    assert green == list(green) and list(green) == green and green[:2] == [green[0], green[1]]
    assert green + [] == list(green) and [] + green == list(green)
    assert [b.numVoxels for b in both] == [row[3] for row in from_columns]
    empty = an.diffDensityObj.createFullBlobList(1e30)
    assert empty == [] and len(empty) == 0 and not empty and an.calculateAtomSpecificBlobStatistics(empty) == []


def test_rscc_rsr_metrics(analysis):
    """RSCC / RSR per residue and per atom, the Fo / Fc scale check and the Fc map quirk (densityAnalysis.py:426-435, 783-882)."""
    z, an = analysis
    an.biopdbObj.header = {"resolution": 2.0}
    rm = an.residueMetrics()
    assert ["%s|%s|%s" % (r[0], r[1], r[2]) for r in rm] == list(z["residue_metrics_id"])
    assert np.allclose(np.array([[r[3], r[4], r[5], r[6]] for r in rm]), z["residue_metrics"], rtol=1e-8, atol=1e-12)
    am = an.atomMetrics()
    assert ["%s|%s|%s|%s" % (r[0], r[1], r[2], r[3]) for r in am] == list(z["atom_metrics_id"])
    assert np.allclose(np.array([[r[6], r[7], r[8], r[9]] for r in am]), z["atom_metrics"], rtol=1e-8, atol=1e-12)
    assert np.allclose(an.medianAbsFoFc(), z["median_abs_fo_fc"], rtol=1e-9)
    fc = an.fc
    assert np.allclose([fc.meanDensity, fc.stdDensity], z["fc_mean_std"], rtol=1e-9)    # the reference's Fc keeps the Fo statistics
    want_fc = (z["dens"].astype(np.float64) - 2 * z["diff"].astype(np.float64)).astype(np.float32)
    assert np.array_equal(fc.density, want_fc)                                            # computed on the device, bit for bit
    crs0 = [[3, 4, 5], [10, 2, 7]]
    assert np.array_equal(fc._map.point_density(crs0), [float(want_fc[5, 4, 3]), float(want_fc[7, 2, 10])])
    # one explicit voxel set through the single-set entry point == the batched path
    atom = an.asymmetryAtoms[3]
    crs = an.densityObj.getSphereCrsFromXyz(atom.coord, an._metricsRadius(), 0.0)
    one = an.calculateRsccRsrMetrics(crs)
    assert one[0] == pytest.approx(am[3][6], rel=1e-9) and one[1] == pytest.approx(am[3][7], rel=1e-9)


def test_silent_failure_contract(gpu_ctx):
    """Q7: below the electrons minimum everything stays None and users of the ratio raise."""
    from pdb_eda_amd import ccp4, synthetic, densityAnalysis
    z, spec, st, pdb, params = load_analysis_case("orth")
    densityAnalysis.setGlobals(params)
    dens = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["dens"])), "x", ctx=gpu_ctx)
    diff = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["diff"])), "x", ctx=gpu_ctx)
    densityAnalysis._attachCutoffs(dens, diff)
    an = densityAnalysis.DensityAnalysis("x", dens, diff, st, pdb)
    an.aggregateCloud(minTotalElectrons=1e9)
    assert an._densityElectronRatio is None and an._medians is None
    an2 = densityAnalysis.DensityAnalysis("x", dens, diff, st, pdb)
    an2.aggregateCloud = lambda *a, **k: None
    with pytest.raises(RuntimeError):
        an2.calculateRegionDiscrepancy([st.child_list[0].child_list[0].child_list[0].child_list[0].coord], 3.5)


def test_structure_without_symmetry_operators(gpu_ctx):
    """A PDB file without REMARK 290 has NO operators (tests/golden/pdbheader.json): the reference then builds empty symmetry
    atom lists (densityAnalysis.py:896-912, the loop over rotationMats runs zero times) and its blob statistics fail with
    cdist's ValueError on the empty coordinate array (:933) -- an ordinary exception that drops the entry, never a library
    error that stops a pool."""
    import copy
    from pdb_eda_amd import ccp4, synthetic, densityAnalysis
    z, spec, st, pdb, params = load_analysis_case(ANALYSIS_CASES[0])
    densityAnalysis.setGlobals(params)
    dens = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["dens"])), "nosym", ctx=gpu_ctx)
    diff = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["diff"])), "nosym", ctx=gpu_ctx)
    densityAnalysis._attachCutoffs(dens, diff)
    pdb = copy.deepcopy(pdb)
    pdb.header.rotationMats = []
    an = densityAnalysis.DensityAnalysis("nosym", dens, diff, st, pdb)
    assert len(an.symmetryAtoms) == 0 and len(an.symmetryOnlyAtoms) == 0 and len(an.asymmetryAtoms) == 0
    assert np.asarray(an.symmetryAtomCoords).shape[0] == 0
    assert an.calculateAtomSpecificBlobStatistics([]) == []
    blobs = an.greenBlobList
    assert blobs
    with pytest.raises(ValueError):
        an.calculateAtomSpecificBlobStatistics(blobs)
    # the raw C entry point with zero operators is an empty result, not an argument error
    idx, sym, xyz = gpu_ctx.symmetry_atoms(np.zeros((3, 3)), np.zeros((0, 12)), np.eye(3), np.zeros(3), np.ones(3))
    assert len(idx) == 0 and len(sym) == 0 and len(xyz) == 0


def test_host_sized_sphere_batches_are_checked_on_the_device(monkeypatch):
    """Per-atom sphere batches are sized by the HOST (an atom's box size follows from its radius) and nobody waits for the
    device's totals; k_make_vols holds its own totals against the host's and, should they ever be larger, empties the volumes
    before anything is painted and fails the call -- no write out of bounds.  The debug hook halves the host's totals."""
    from pdb_eda_amd import _native, ccp4, synthetic
    import io
    spec = synthetic.MapSpec(ncrs=(48, 40, 36), spacing=0.5)
    g = synthetic.smooth_noise((36, 40, 48), 5, 1.5)
    xyz = np.array([[8.0, 9.0, 7.0], [12.0, 10.5, 9.0], [15.0, 6.0, 11.0]])
    radii = np.array([1.5, 2.0, 1.5], dtype=np.float32)
    good = _native.Context(0)
    dm = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, g)), "ok", ctx=good)
    want = dm._map.sphere_blobs(xyz, radii, np.arange(4), 0.0).stats()["n"]
    assert len(want) >= 3
    monkeypatch.setenv("PDBEDA_DEBUG_SHRINK_TOTALS", "1")
    bad = _native.Context(0)                                   # (debug hooks are read when a context is made)
    monkeypatch.delenv("PDBEDA_DEBUG_SHRINK_TOTALS")
    dm2 = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, g)), "shrunk", ctx=bad)
    with pytest.raises(_native.PdbedaError):
        dm2._map.sphere_blobs(xyz, radii, np.arange(4), 0.0).stats()
    with pytest.raises(_native.PdbedaError):
        dm2._map.region_sums(xyz, radii, np.arange(4), 0.1)
    # groups of several atoms wait for the device's totals as before: not affected by the hook, and the context is still good
    assert np.array_equal(dm2._map.sphere_blobs(xyz, radii, np.array([0, 3]), 0.0).stats()["n"], dm._map.sphere_blobs(xyz, radii, np.array([0, 3]), 0.0).stats()["n"])
    assert np.array_equal(dm._map.sphere_blobs(xyz, radii, np.arange(4), 0.0).stats()["n"], want)
