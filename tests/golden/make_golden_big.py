"""Reference goldens at the BASELINE analysis sizes: the REFERENCE DensityAnalysis (Cython cutils path) on
  c0  100 x 108 x 96, 1 000 atoms   (configs[0] stand-in, "1stp-shaped")
  c2  128^3, 2 000 atoms            (bench.py's analysis entry)
  c3  200^3, 500 atoms              (one entry of configs[3])
Only the case table (sizes + seeds: pdb_eda_amd.synthetic.BIG_CASES regenerates the inputs bit for bit) and the reference's
numeric outputs are kept -- no grids in git.  Build container only, one core, several minutes:
    python tests/golden/make_golden_big.py [case ...]        -> tests/golden/analysis_big_<case>.npz
"""
import io
import json
import os
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402
from pdb_eda_amd import synthetic, structure  # noqa: E402


def run_case(name, ccp4, da):
    ncrs, n_res, seed, spacing = synthetic.BIG_CASES[name]
    spec, header, st, params, dens, diff, rot = synthetic.cube_entry(ncrs, n_res, seed, spacing, synthetic.BIG_CASE_SPECS.get(name))
    da.setGlobals(params)
    t0 = time.perf_counter()
    densityObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), name)
    diffObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), name)
    densityObj.densityCutoff = densityObj.meanDensity + 1.5 * densityObj.stdDensity
    diffObj.diffDensityCutoff = diffObj.meanDensity + 3 * diffObj.stdDensity
    pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid=name, resolution=2.0, spaceGroup="P_1", rotationMats=rot))
    an = da.DensityAnalysis(name, densityObj, diffObj, st, pdbObj)
    out = {"case": np.array(json.dumps({"ncrs": list(ncrs), "residues": n_res, "seed": seed, "spacing": spacing})),
           "dens_checksum": np.float64(np.sum(dens, dtype=np.float64)), "diff_checksum": np.float64(np.sum(diff, dtype=np.float64)),
           "mean_std": np.array([densityObj.meanDensity, densityObj.stdDensity, diffObj.meanDensity, diffObj.stdDensity])}
    timing = {"parse": time.perf_counter() - t0}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        an.aggregateCloud()
        timing["aggregateCloud"] = time.perf_counter() - t0
        assert an.densityElectronRatio, "synthetic entry failed the reference's own minimum-electrons gate"
        out["ratio"] = np.float64(an.densityElectronRatio)
        out["num_voxels"] = np.int64(an.numVoxelsAggregated)
        out["total_electrons"] = np.float64(an.totalAggregatedElectrons)
        out["total_density"] = np.float64(an.totalAggregatedDensity)
        atoms = an.atomCloudDescriptions
        for f in atoms.dtype.names:
            out["acd_" + f] = np.asarray(atoms[f])
        out["res_rows"] = np.array([[r[1]] + [r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.residueCloudDescriptions], dtype=np.float64).reshape(-1, 8)
        out["dom_rows"] = np.array([[r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.domainCloudDescriptions], dtype=np.float64).reshape(-1, 7)
        out["medians"] = np.array(json.dumps({k: {t: float(v) for t, v in d.items()} for k, d in an.medians.items()}))
        out["overlap_complete"] = np.array(json.dumps(dict(an.atomTypeOverlapCompleteness)))
        out["overlap_incomplete"] = np.array(json.dumps(dict(an.atomTypeOverlapIncompleteness)))
        t0 = time.perf_counter()
        out["atom_discrepancy"] = np.array([r[6:] for r in an.calculateAtomRegionDiscrepancies(3.5, 3.0)], dtype=np.float64)
        timing["atomRegionDiscrepancies"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        out["residue_discrepancy"] = np.array([r[5:] for r in an.calculateResidueRegionDiscrepancies(3.5, 3.0)], dtype=np.float64)
        timing["residueRegionDiscrepancies"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for tag, blobs in (("green", an.greenBlobList), ("red", an.redBlobList)):
            stats = an.calculateAtomSpecificBlobStatistics(blobs)
            out["blob_%s_num" % tag] = np.array([[s[0], s[2], s[3], s[4]] for s in stats], dtype=np.float64).reshape(-1, 4)
            out["blob_%s_atom" % tag] = np.array(["%s|%s|%s|%s" % (s[6], s[7], s[8], tuple(int(v) for v in s[9])) for s in stats])
            out["blob_%s_centroid" % tag] = np.array([list(s[11]) for s in stats], dtype=np.float64).reshape(-1, 3)
        timing["blobStatistics"] = time.perf_counter() - t0
    out["reference_seconds"] = np.array(json.dumps({k: round(v, 2) for k, v in timing.items()}))
    path = os.path.join(HERE, "analysis_big_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", timing, flush=True)


if __name__ == "__main__":
    ccp4, da = refload.load()
    for name in sys.argv[1:] or list(synthetic.BIG_CASES):
        run_case(name, ccp4, da)
