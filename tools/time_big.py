# A 512^3 map (8x BASELINE configs[1]): correctness against the oracle (counts, keys, totals) and throughput.
#   python tools/time_big.py [edge [nsd]]
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
from oracle import oracle as ora
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.4)
g = synthetic.smooth_noise((edge, edge, edge), seed=3, sigma_voxels=1.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, g, header.geometry())
mean, std = dmap.stats()
cut = mean + (float(sys.argv[2]) if len(sys.argv) > 2 else 1.5) * std
for _ in range(2):
    green, red = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    green, red = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
dt = (time.perf_counter() - t0) / n
print("%d^3: %.3f ms/step = %.1f Gvoxel/s; blobs %d / %d" % (edge, 1e3 * dt, edge ** 3 / dt / 1e9, len(green), len(red)), flush=True)
ctx.profile_begin()
green, red = dmap.full_blobs_pm(cut, -cut, labels=True)
len(green)
prof = ctx.profile_end()
print("kernels (us):", {k: round(1e3 * ms, 1) for k, (_, ms) in sorted(prof.items())}, green.counters(), flush=True)
o = ora.Oracle(header, g)
t0 = time.perf_counter()
want = o.full_blobs(cut, labels=True)
print("oracle %.1f s" % (time.perf_counter() - t0), flush=True)
st = green.stats()
ok = (np.array_equal(st["n"], want["n"]) and np.array_equal(st["firstKey"], want["firstKey"]) and np.allclose(st["totalDensity"], want["totalDensity"], rtol=1e-9)
      and np.array_equal(green.labels(dmap.unique_shape), want["labels"]))
print("green list == oracle:", ok)
sys.exit(0 if ok else 1)
