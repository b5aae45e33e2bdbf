"""Bridge for the entries/min CPU baseline: the REFERENCE's per-entry work of `pdb_eda multiple` (parse the
2Fo-Fc map, aggregateCloud: multipleStructures.py:320-356) against the CPU restatement bench.py times on the GPU box
(oracle/cpu_entry.multiple_entry), HERE, on the same configs[3] entry (200^3, 500 atoms), one core each -- plus the restatement
over a multiprocessing.Pool on all cores of this container.  Build container only:
    python tests/golden/ref_bridge_multiple.py   -> profiles/r03_reference_multiple_cpu.json"""
import io
import json
import os
import sys
import tempfile
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402
from pdb_eda_amd import synthetic, structure  # noqa: E402
from oracle import cpu_entry  # noqa: E402


def main():
    ccp4, da = refload.load()
    ncrs, n_res, seed, spacing = synthetic.BIG_CASES["c3_multiple_entry"]
    spec, header, st, params, dens, diff, rot = synthetic.cube_entry(ncrs, n_res, seed, spacing)
    da.setGlobals(params)
    blob = synthetic.ccp4_bytes(spec, dens)
    t0 = time.perf_counter()
    densityObj = ccp4.parse(io.BytesIO(blob), "c3")
    densityObj.densityCutoff = densityObj.meanDensity + 1.5 * densityObj.stdDensity
    t_parse = time.perf_counter() - t0
    pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="c3", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
    an = da.DensityAnalysis("c3", densityObj, None, st, pdbObj)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t0 = time.perf_counter()
        an.aggregateCloud()
        t_cloud = time.perf_counter() - t0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c3.ccp4")
        with open(path, "wb") as fh:
            fh.write(blob)
        task = (path, n_res, seed, ncrs[0], spacing)
        cpu_entry.multiple_entry(task)
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            rec = cpu_entry.multiple_entry(task)
        t_port = (time.perf_counter() - t0) / reps
        cores = len(os.sched_getaffinity(0))
        pool = cpu_entry.multiple_baseline([task] * (2 * cores), cores, seconds=10.0)
    assert abs(rec["ratio"] - an.densityElectronRatio) <= 1e-9 * abs(an.densityElectronRatio) and rec["num_voxels"] == an.numVoxelsAggregated
    out = {"note": "one configs[3] entry (200^3 grid, 500 atoms, pdb_eda_amd.synthetic.BIG_CASES['c3_multiple_entry']) in the build container: the reference "
                   "(Cython cutils, -O3; parse of the 2Fo-Fc map + aggregateCloud, what analyzePDBID spends its time on) vs the CPU restatement bench.py "
                   "times on the GPU box (oracle/cpu_entry.multiple_entry); same densityElectronRatio and voxel count",
           "reference_one_core_s": {"parse_2FoFc": round(t_parse, 3), "aggregateCloud": round(t_cloud, 3), "total": round(t_parse + t_cloud, 3)},
           "reference_entries_per_min_one_core": round(60.0 / (t_parse + t_cloud), 2),
           "port_one_core_s": round(t_port, 4), "port_entries_per_min_one_core": round(60.0 / t_port, 1),
           "reference_over_port": round((t_parse + t_cloud) / t_port, 1),
           "port_pool": {"cores": cores, "entries_per_min": round(pool["entries_per_min"], 1), "seconds": round(pool["seconds"], 2)},
           "density_electron_ratio": float(an.densityElectronRatio)}
    with open(os.path.join(ROOT, "profiles", "r03_reference_multiple_cpu.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
