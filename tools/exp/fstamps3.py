import sys, os, ctypes as C
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
for _ in range(3):
    g, r = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
lib = _native.lib()
out = np.zeros((1024, 8), dtype=np.uint64)
lib.pdbeda_bloblist_stamps.restype = C.c_int
lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), 1024) == 0
t = out.astype(np.int64)
for name, a, b in (("list loads + LDS writes", 0, 1), ("merge loop (wave 0)", 1, 2), ("hash inserts (wave 0)", 2, 3)):
    d = t[:, b] - t[:, a]
    print("%-28s median %5d p90 %5d max %5d ticks" % (name, np.median(d), np.percentile(d, 90), d.max()))
print("thread 0: pairs median %d, runs in its two lists median %d" % (np.median(t[:, 4]), np.median(t[:, 5])))
