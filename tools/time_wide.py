# Whole-map labelling of a grid wider than one tile (rows of 320 voxels): the c-tiled path.
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
nc, nr, ns = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (320, 256, 205)
spec = synthetic.MapSpec(ncrs=(nc, nr, ns), spacing=0.4)
grid = synthetic.smooth_noise((ns, nr, nc), seed=7, sigma_voxels=1.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, grid, header.geometry())
mean, std = dmap.stats()
cut = mean + 1.5 * std
for _ in range(3):
    k = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
ctx.profile_begin()
for _ in range(10):
    k = dmap.full_blobs_pm(cut, -cut, labels=True)
prof = ctx.profile_end()
tot = sum(ms / c for c, ms in prof.values())
print("%dx%dx%d: %.0f us/step, %.1f Gvoxel/s" % (nc, nr, ns, tot * 1e3, nc * nr * ns / tot / 1e6), {a: round(ms / c * 1e3, 1) for a, (c, ms) in sorted(prof.items())})
