"""CPU: the oracle's sphere / blob primitives reproduce the reference's regional statistics
(calculateRegionDiscrepancy / calculateRegionDensity, densityAnalysis.py:1037-1211) on the analysis
golden vectors -- including the identity the device path relies on:
sum(blob.totalDensity over findAberrantBlobs) == masked sum over the de-duplicated sphere union."""
import numpy as np
import pytest

from conftest import ANALYSIS_CASES, load_analysis_case
from oracle import oracle as ora


@pytest.mark.parametrize("name", ANALYSIS_CASES)
def test_region_discrepancy_rows(name):
    from pdb_eda_amd import ccp4, synthetic
    z, spec, st, pdb, params = load_analysis_case(name)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    diff = z["diff"]
    o = ora.Oracle(header, diff)
    ratio = float(z["ratio"])
    mean, std = float(np.mean(diff.astype(np.float64))), float(np.std(diff.astype(np.float64)))
    cut = mean + 3.0 * std
    atoms = list(st.get_atoms())
    total_abs = o.sum_of_abs(cut)
    for ai in range(0, len(atoms), 7):
        xyz = [atoms[ai].coord.astype(np.float64)]
        green = o.find_aberrant_blobs(xyz, [3.5], cut)
        red = o.find_aberrant_blobs(xyz, [3.5], -cut)
        pos = sum(b["totalDensity"] for b in green)
        neg = sum(b["totalDensity"] for b in red)
        count = len(o.sphere_crs(xyz[0], 3.5, 0.0))
        expected = total_abs / diff.size * count
        want = z["atom_discrepancy"][ai]
        got = [abs(pos) + abs(neg), (abs(pos) + abs(neg)) / ratio, expected, expected / ratio, pos + neg, (pos + neg) / ratio, pos, pos / ratio, neg, neg / ratio]
        assert np.allclose(got, want, rtol=1e-9, atol=1e-12)
        # identity: clustered blob totals == plain masked sums over the sphere voxels
        vox = o.sphere_crs(xyz[0], 3.5, 0.0)
        d = np.array([o.point_density(v) for v in vox])
        assert d[d > np.float32(cut)].sum() == pytest.approx(pos, rel=1e-9, abs=1e-12)
        assert d[d < np.float32(-cut)].sum() == pytest.approx(neg, rel=1e-9, abs=1e-12)
