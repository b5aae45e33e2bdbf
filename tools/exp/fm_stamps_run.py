import sys, os, ctypes as C
sys.path.insert(0, os.getcwd())
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
out = np.zeros((2048, 8), dtype=np.uint64)
lib.pdbeda_bloblist_stamps.restype = C.c_int
lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), out.size) == 0
t = out.astype(np.int64)[128:128 + 1024]
t0 = t[:, 0].min()
for j, name in enumerate(["entry", "loads back + set cleared", "wave 0 has its pairs", "all pairs in the set", "compacted", "united"]):
    d = (t[:, j] - t0) / 100.0
    print("%-16s median %5.1f  p90 %5.1f  max %5.1f us" % (name, np.median(d), np.percentile(d, 90), d.max()))
