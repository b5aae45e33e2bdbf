# Diagnostic: per-phase cycle shares of k_tile_label from in-kernel s_memtime stamps.
# Needs a library built with -DPDBEDA_STAMPS (PDBEDA_LIB=build/abl/libSTAMP.so).
import sys, os, ctypes as C
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
cut = mean + 1.5 * std
for _ in range(3):
    g, r = dmap.full_blobs_pm(cut, -cut, labels=True)
nt = 1024
out = np.zeros((nt, 16), dtype=np.uint64)
lib = _native.lib()
lib.pdbeda_bloblist_stamps.restype = C.c_int
lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), nt) == 0
o = out.astype(np.int64)
print("2->8 setup", np.median(o[:,8]-o[:,2]), " 8->9 barrier", np.median(o[:,9]-o[:,8]), " 9->3 enumerate", np.median(o[:,3]-o[:,9]))
d = np.diff(o[:, :8], axis=1)
names = ["P1 stream+ballot+compact", "A2/A3 index+sums", "B1 enumerate", "B2 rounds", "C1 number", "C2 fold+publish", "flush"]
tot = (out[:, 7].astype(np.int64) - out[:, 0].astype(np.int64))
print("per-tile cycles (s_memtime ticks, 100 MHz?): median total", np.median(tot))
for k, nm in enumerate(names):
    print("%-28s median %8.0f  mean %8.0f  share %5.1f%%" % (nm, np.median(d[:, k]), d[:, k].mean(), 100 * d[:, k].sum() / tot.sum()))
print("kernel span (ticks):", int(out[:, 7].max() - out[:, 0].min()))
