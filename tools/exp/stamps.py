# Reads the stamps of a libSTAMP.so build (tools/exp/mkstamp.py): where the time of k_tile_label goes, per tile and per wave.
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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
nt = ((n + 255) // 256) * ((n + 7) // 8) ** 2
out = np.zeros((nt, 64), dtype=np.uint64)
lib.pdbeda_bloblist_stamps.restype = C.c_int
lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), nt) == 0
t = out.astype(np.int64)
t0 = t[:, 0].min()
us = lambda x: x / 100.0   # s_memrealtime ticks at 100 MHz
print("tiles %d; kernel span: starts spread %.1f us, last end %.1f us" % (nt, us(t[:, 0].max() - t0), us(t[:, 5].max() - t0)))
names = {1: "barrier 1 (stream done)", 2: "barrier 2 (unions done)", 3: "barrier 3 (numbered)", 4: "barrier 4 (folded)", 5: "end (flushed)"}
prev = t[:, 0]
for k in range(1, 6):
    d = t[:, k] - prev
    print("%-26s dt median %5.1f  p90 %5.1f  max %5.1f us   (reached at median %5.1f, max %5.1f us after the first start)" % (
        names[k], us(np.median(d)), us(np.percentile(d, 90)), us(d.max()), us(np.median(t[:, k] - t0)), us((t[:, k] - t0).max())))
    prev = t[:, k]
for name, lo, ref in (("wave: stream end - tile start", 8, t[:, 0:1]), ("wave: numbering (A2) after barrier 1", 16, t[:, 1:2]), ("wave: unions (B) after A2", 24, None),
                      ("wave: folds (C2) after barrier 3", 32, t[:, 3:4])):
    w = t[:, lo:lo + 8]
    d = w - (t[:, 16:24] if ref is None else ref)
    print("%-40s median %5.1f  p90 %5.1f  max %5.1f us; slowest wave of a tile: median %5.1f us" % (name, us(np.median(d)), us(np.percentile(d, 90)), us(d.max()), us(np.median(d.max(axis=1)))))
# where the late tiles are: by XCD group (workgroup id mod 8), by position along s (tile id // 32 at 256^3) and by CU slot order
end = us(t[:, 5] - t0)
b1 = us(t[:, 1] - t0)
ids = np.arange(nt)
print("end of tile by (id mod 8):", [round(float(np.median(end[ids % 8 == k])), 1) for k in range(8)], " max:", [round(float(end[ids % 8 == k].max()), 1) for k in range(8)])
print("barrier 1 by (id mod 8):  ", [round(float(np.median(b1[ids % 8 == k])), 1) for k in range(8)])
q = max(1, nt // 8)
print("end of tile by id range (eighths):", [round(float(np.median(end[k * q:(k + 1) * q])), 1) for k in range(8)], " max:", [round(float(end[k * q:(k + 1) * q].max()), 1) for k in range(8)])
print("barrier 1 by id range (eighths):  ", [round(float(np.median(b1[k * q:(k + 1) * q])), 1) for k in range(8)])
late = np.argsort(end)[-16:]
print("the 16 latest tiles:", sorted(late.tolist()), "ends", np.round(np.sort(end)[-16:], 1).tolist())
print("corr(end, barrier1) = %.2f   corr(end - barrier1, barrier1) = %.2f" % (np.corrcoef(end, b1)[0, 1], np.corrcoef(end - b1, b1)[0, 1]))
