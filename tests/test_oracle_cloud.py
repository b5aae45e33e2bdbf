"""CPU: the oracle's aggregateCloud composite (oracle/pdbeda_oracle.c: ora_cloud_begin / ora_cloud_finish -- a restatement of
densityAnalysis.py:571-731 on the flattened structure) is PINNED on the reference's own results: it stands in for the device
behind the product's host-side table code (flattening before, statistics tail after -- the same code the GPU path runs), and every
table must equal what the REFERENCE's DensityAnalysis produced on the same entry -- the two small analysis goldens and the three
entries at the BASELINE sizes (1 000 / 2 000 / 500 atoms).  That makes it the checker for entries of thousands of atoms, which the
reference itself needs minutes for (tests/test_gpu_analysis_big.py::test_many_atoms_against_the_oracle_composite)."""
import json
import os

import numpy as np
import pytest

from conftest import ANALYSIS_CASES, load_analysis_case
from oracle import oracle as ora

HERE = os.path.dirname(os.path.abspath(__file__))


def _analyse(header, grid, st, params):
    from oracle import cpu_entry
    return cpu_entry.analyse(header, grid, st, params)


def _check(an, z):
    assert an.densityElectronRatio == pytest.approx(float(z["ratio"]), rel=1e-9)
    assert an.numVoxelsAggregated == int(z["num_voxels"])
    assert an.totalAggregatedElectrons == pytest.approx(float(z["total_electrons"]), rel=1e-12)
    assert an.totalAggregatedDensity == pytest.approx(float(z["total_density"]), rel=1e-9)
    atoms = an.atomCloudDescriptions
    assert len(atoms) == len(z["acd_chain"])
    for f in ("chain", "residue_number", "residue_name", "atom_name", "atom_type", "num_voxels", "electrons"):
        assert np.array_equal(np.asarray(atoms[f]), z["acd_" + f]), f
    for f in ("density_electron_ratio", "bfactor", "centroid_distance", "centroid_xyz", "adj_density_electron_ratio", "domain_fraction",
              "corrected_fraction", "corrected_density_electron_ratio", "volume"):
        assert np.allclose(np.asarray(atoms[f]), z["acd_" + f], rtol=1e-8, atol=1e-10), f
    got = np.array([[r[1]] + [r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.residueCloudDescriptions], dtype=np.float64).reshape(-1, 8)
    key = lambda a: a[np.lexsort((a[:, 7], a[:, 2], a[:, 0]))]
    assert got.shape == z["res_rows"].shape and np.allclose(key(got), key(z["res_rows"]), rtol=1e-8, atol=1e-10)
    got = np.array([[r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.domainCloudDescriptions], dtype=np.float64).reshape(-1, 7)
    want = z["dom_rows"]
    assert got.shape == want.shape and np.allclose(got[np.argsort(got[:, 0])], want[np.argsort(want[:, 0])], rtol=1e-8, atol=1e-10)
    assert dict(an.atomTypeOverlapCompleteness) == json.loads(str(z["overlap_complete"]))
    assert dict(an.atomTypeOverlapIncompleteness) == json.loads(str(z["overlap_incomplete"]))
    want = json.loads(str(z["medians"]))
    for col, d in want.items():
        for t, v in d.items():
            assert float(an.medians[col][t]) == pytest.approx(v, rel=1e-8, abs=1e-10), (col, t)


@pytest.mark.parametrize("name", ANALYSIS_CASES)
def test_composite_equals_reference_small(name):
    from pdb_eda_amd import ccp4, synthetic
    z, spec, st, pdb, params = load_analysis_case(name)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    _check(_analyse(header, z["dens"], st, params), z)


@pytest.mark.parametrize("name", ["c0_1stp_like", "c3_multiple_entry", "c2_bench_entry", "c5_hex_perm"])
def test_composite_equals_reference_at_baseline_sizes(name):
    from pdb_eda_amd import synthetic
    z = np.load(os.path.join(HERE, "golden", "analysis_big_%s.npz" % name), allow_pickle=False)
    ncrs, n_res, seed, spacing = synthetic.BIG_CASES[name]
    spec, header, st, params, dens, diff, rot = synthetic.cube_entry(ncrs, n_res, seed, spacing, synthetic.BIG_CASE_SPECS.get(name))
    assert float(np.sum(dens, dtype=np.float64)) == float(z["dens_checksum"])
    _check(_analyse(header, dens, st, params), z)
