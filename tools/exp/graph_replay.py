# Experiment: does replaying the whole-map step from a hipGraph (stream capture of the library's own launches) shorten the step?
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch's)
from pdb_eda_amd import _native, ccp4, synthetic
n = 256
spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
grid = synthetic.smooth_noise((n, n, n), seed=7, sigma_voxels=1.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, grid, header.geometry())
mean, std = dmap.stats()
cut = mean + 1.5 * std
for _ in range(5):
    keep = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
K = 200
t0 = time.perf_counter()
for _ in range(K):
    keep = dmap.full_blobs_pm(cut, -cut, labels=True)
ctx.synchronize()
print("plain launches: %.2f us/step" % (1e6 * (time.perf_counter() - t0) / K))
import glob
hip = None
for line in open("/proc/self/maps"):
    if "libamdhip64" in line:
        hip = C.CDLL(line.split()[-1])
        break
stream = C.c_void_p(ctx.stream)
graph, gexec = C.c_void_p(), C.c_void_p()
assert hip.hipStreamBeginCapture(stream, 2) == 0          # hipStreamCaptureModeRelaxed
held = dmap.full_blobs_pm(cut, -cut, labels=True)
rc = hip.hipStreamEndCapture(stream, C.byref(graph))
assert rc == 0, rc
assert hip.hipGraphInstantiate(C.byref(gexec), graph, None, None, 0) == 0
for _ in range(5):
    assert hip.hipGraphLaunch(gexec, stream) == 0
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    hip.hipGraphLaunch(gexec, stream)
ctx.synchronize()
print("graph replay:   %.2f us/step" % (1e6 * (time.perf_counter() - t0) / K))
print("blobs", len(held[0]), len(held[1]), "vs", len(keep[0]), len(keep[1]))
