# Runs the 256^3 labelling step a few times and synchronises WITHOUT reading any result: the driver of the phase-count variants
# (tools/exp/phase_counts.sh), whose truncated kernels leave no valid results behind.
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
n = 256
spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
grid = synthetic.smooth_noise((n, n, n), seed=7, sigma_voxels=1.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, grid, header.geometry())
mean, std = dmap.stats()
cut = mean + 1.5 * std
keep = []
for _ in range(6):
    keep.append(dmap.full_blobs_pm(cut, -cut, labels=True))
ctx.synchronize()
print("done")
# (freeing the lists reads nothing from the device)
