# Reads the stamps of a libSTAMP.so build: median / max over tiles of the time between consecutive barriers of k_tile_label.
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
nt = 1024
out = np.zeros((nt, 32), dtype=np.uint64)
lib.pdbeda_bloblist_stamps.restype = C.c_int
lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), nt) == 0
t = out.astype(np.int64)
t0 = t[:, 0].min()
print("kernel span (10 ns ticks): start spread %d, last end %d" % (t[:, 0].max() - t0, t.max() - t0))
print("hook rounds per tile: median %d  p90 %d  max %d" % (np.median(t[:, 20]), np.percentile(t[:, 20], 90), t[:, 20].max()))
print("B1 merge of thread 0 done %d ticks after the batch-top barrier (median; stamp 21 - stamp 7)" % np.median(t[:, 21] - t[:, 7]))
prev = t[:, 0]
for k in range(1, 20):
    cur = t[:, k]
    ok = cur > 0
    if not ok.any():
        continue
    d = (cur - prev)[ok]
    print("stamp %2d: tiles %4d  dt median %5d  p90 %5d  max %5d   (at median %5d)" % (k, ok.sum(), np.median(d), np.percentile(d, 90), d.max(), np.median(cur[ok] - t0)))
    prev = np.where(ok, cur, prev)
