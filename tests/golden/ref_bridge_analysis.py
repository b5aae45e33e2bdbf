"""The reference's densityAnalysis path timed HERE (build container) on the synthetic "~2 A entry" of bench.py's analysis leg
(same generator and seeds): aggregateCloud, calculateAtomRegionDiscrepancies, calculateResidueRegionDiscrepancies, green / red
blob statistics.  Build container only:  python tests/golden/ref_bridge_analysis.py [n_residues ...]"""
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
from pdb_eda_amd import ccp4 as my_ccp4  # noqa: E402

ccp4, da = refload.load()
rows = []
for n_res in [int(v) for v in sys.argv[1:]] or [100, 400]:
    edge = 128
    spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
    header = my_ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([edge - 7] * 3))
    st = synthetic.chain_structure(n_res, 5, lo, hi, hetero_every=9, zero_occupancy_every=37)
    params = synthetic.synthetic_params()
    da.setGlobals(params)
    dens = synthetic.gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=5)
    diff = (synthetic.noise_grid(spec, 105, 1.2) * 0.12).astype(np.float32)
    rot = [np.hstack([np.eye(3), np.zeros((3, 1))]), np.array([[-1.0, 0, 0, 0.5 * header.xlength], [0, -1.0, 0, 0], [0, 0, 1.0, 0.5 * header.zlength]])]
    pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="synth", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
    t = {}
    t0 = time.perf_counter()
    densityObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), "synth")
    diffObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), "synth")
    densityObj.densityCutoff = densityObj.meanDensity + 1.5 * densityObj.stdDensity
    diffObj.diffDensityCutoff = diffObj.meanDensity + 3 * diffObj.stdDensity
    an = da.DensityAnalysis("synth", densityObj, diffObj, st, pdbObj)
    t["parse"] = time.perf_counter() - t0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter(); an.aggregateCloud(); t["aggregateCloud"] = time.perf_counter() - t0
        t0 = time.perf_counter(); an.calculateAtomRegionDiscrepancies(3.5, 3.0, ""); t["atomRegionDiscrepancies"] = time.perf_counter() - t0
        t0 = time.perf_counter(); an.calculateResidueRegionDiscrepancies(3.5, 3.0, ""); t["residueRegionDiscrepancies"] = time.perf_counter() - t0
        t0 = time.perf_counter(); an.calculateAtomSpecificBlobStatistics(an.greenBlobList + an.redBlobList); t["blobStatistics"] = time.perf_counter() - t0
    rows.append({"residues": n_res, "atoms": len(list(st.get_atoms())), "grid": edge, "seconds": {k: round(v, 3) for k, v in t.items()},
                 "total_s": round(sum(t.values()), 3), "density_electron_ratio": float(an.densityElectronRatio)})
    print(rows[-1], flush=True)
    with open(os.path.join(ROOT, "profiles", "r01_reference_analysis_cpu.json"), "w") as fh:
        json.dump({"note": "reference pdb_eda (Cython cutils, -O3) in the build container, one core, on the synthetic entry of bench.py's analysis leg "
                           "(400 residues = the bench entry)", "rows": rows}, fh, indent=1)
        fh.write("\n")
