# Worst case for the cross-tile union-find: ONE blob that visits every tile of a 256^3 map in tile order (a serpentine line
# along r in every 8-section slab, joined at alternating ends): 1024 tile components in a chain.
#   python tools/time_snake.py
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
edge = 256
g = np.zeros((edge, edge, edge), dtype=np.float32)           # [s][r][c]
for j in range(edge // 8):
    s = 8 * j + 4
    g[s, :, 10] = 1.0                                         # a line along r through all 32 r-tiles of this slab
    if j + 1 < edge // 8:
        r_end = edge - 1 if j % 2 == 0 else 0
        g[s:s + 9, r_end, 10] = 1.0                           # join to the next slab at alternating ends
spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, g, header.geometry())
for _ in range(3):
    k = dmap.full_blobs(0.5, labels=True)
ctx.synchronize()
ctx.profile_begin()
for _ in range(10):
    k = dmap.full_blobs(0.5, labels=True)
prof = ctx.profile_end()
st = k.stats()
print("blobs %d, voxels %d (expected 1 blob of %d)" % (len(st["n"]), int(st["n"].sum()), int((g > 0.5).sum())))
print("total %.0f us" % (1e3 * sum(ms / c for c, ms in prof.values())), {a: round(ms / c * 1e3, 1) for a, (c, ms) in sorted(prof.items())})
