# stamps of k_face_merge (libFSTAMP.so): per tile [start, after set clear, before list loads, merge done (thread 0), after barrier, compacted, unions done, n_pairs]
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
t0 = t[:, 0].min()
print("tiles start spread %d ticks (10 ns); last end %d; pairs per tile median %d max %d" % (t[:, 0].max() - t0, t[:, 6].max() - t0, np.median(t[:, 7]), t[:, 7].max()))
names = ["set clear + barrier", "task setup (face_rows load)", "list loads + merge + inserts (thread 0)", "barrier (all merges)", "compaction", "unions"]
for k in range(1, 7):
    d = t[:, k] - t[:, k - 1]
    print("%-42s median %5d  p90 %5d  max %5d   (at median %5d)" % (names[k - 1], np.median(d), np.percentile(d, 90), d.max(), np.median(t[:, k] - t0)))
