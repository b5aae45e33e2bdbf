# Timing driver: runs the whole-map labelling step a few times and prints per-kernel HIP-event times.
# Used under rocprofv3 (kernel trace / PMC passes) and for A/B builds via PDBEDA_LIB.
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
n = 256
spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
grid = synthetic.smooth_noise((n, n, n), seed=7, sigma_voxels=1.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, grid, header.geometry())
mean, std = dmap.stats()
nsd = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5      # (python tools/profile_step.py 3.0: pdb_eda's own cutoff)
cut = mean + nsd * std
for _ in range(3):
    k = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
ctx.profile_begin()
for _ in range(10):
    k = dmap.full_blobs_pm(cut, -cut, labels=True)
prof = ctx.profile_end()
print(k[0].counters())
print(os.environ.get("PDBEDA_LIB", "main").split("/")[-1], "prezero=%s" % os.environ.get("PDBEDA_PREZERO", "1"), nsd, {a: round(ms / c * 1e3, 1) for a, (c, ms) in sorted(prof.items())})
