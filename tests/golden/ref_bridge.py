"""Bridge for the CPU baseline: the reference's Cython path and the CPU restatement (oracle) timed HERE, in the
build container, on the same grids.  The reference never travels to the GPU box, so bench.py times the restatement there
("port") and this ratio, measured where both run, links the two.  Build container only:  python tests/golden/ref_bridge.py [tag]
(-> profiles/<tag>_reference_vs_restatement_cpu.json; default r04)"""
import io
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refload  # noqa: E402
from pdb_eda_amd import synthetic  # noqa: E402
from pdb_eda_amd import ccp4 as my_ccp4  # noqa: E402
from oracle import oracle as ora  # noqa: E402

ccp4, _ = refload.load(with_density_analysis=False)
rows = []
for edge, nsd in ((64, 3.0), (100, 3.0), (128, 3.0), (200, 3.0), (48, 1.5), (64, 1.5)):
    spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.4)
    grid = synthetic.smooth_noise((edge, edge, edge), seed=7, sigma_voxels=1.5)
    blob = synthetic.ccp4_bytes(spec, grid)
    dm = ccp4.parse(io.BytesIO(blob), "bridge")
    cut = dm.meanDensity + nsd * dm.stdDensity
    t0 = time.perf_counter()
    g = dm.createFullBlobList(cut)
    r = dm.createFullBlobList(-cut)
    t_ref = time.perf_counter() - t0
    header = my_ccp4.DensityHeader.fromFileHeader(blob[:1024])
    o = ora.Oracle(header, grid)
    reps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.5:
        a = o.full_blobs(cut)
        b = o.full_blobs(-cut)
        reps += 1
    t_ora = (time.perf_counter() - t0) / reps
    assert len(a["n"]) == len(g) and len(b["n"]) == len(r)
    n = edge ** 3
    rows.append({"grid": edge, "cutoff_sigma": nsd, "significant_voxels": int(a["n"].sum() + b["n"].sum()), "reference_s": t_ref,
                 "reference_Mvoxels_per_s": n / t_ref / 1e6, "restatement_s": t_ora, "restatement_Mvoxels_per_s": n / t_ora / 1e6, "ratio": t_ref / t_ora})
    print(rows[-1], flush=True)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
with open(os.path.join(ROOT, "profiles", "%s_reference_vs_restatement_cpu.json" % tag), "w") as fh:
    json.dump({"note": "build container (8 vCPU Xeon 2.1 GHz), one core, threshold + clustering + blob statistics of both signs; reference = pdb_eda Cython path "
                       "(cutils built -O3), restatement = oracle/pdbeda_oracle.c ora_full_blobs", "rows": rows}, fh, indent=1)
    fh.write("\n")
