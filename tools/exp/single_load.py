# One CCP4 grid (200^3 = 32 MB, warm page cache) -> HBM by itself: median call time, and the upload engine's own trace of the last calls.
#   PDBEDA_FILE_READERS=3 PDBEDA_FILE_CHUNK_KB=8192 python tools/exp/single_load.py [edge]
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 200
_native.pin_to_device(0)
ctx = _native.Context(0)
tmp = tempfile.mkdtemp(prefix="pdbeda_single_")
try:
    spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
    dens = np.random.default_rng(1).standard_normal((edge, edge, edge)).astype(np.float32)
    path = os.path.join(tmp, "m.ccp4")
    with open(path, "wb") as f:
        f.write(synthetic.ccp4_bytes(spec, dens))
    geom = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec)).geometry()
    times = []
    for k in range(40):
        t0 = time.perf_counter()
        m = _native.DeviceMap.from_file(ctx, path, 1024, False, geom)
        times.append(time.perf_counter() - t0)
        m.free()
    t = 1e3 * np.array(times[8:])
    print("readers %s chunk %s KB: median %.3f ms = %.1f GB/s, best %.3f ms" % (os.environ.get("PDBEDA_FILE_READERS", "3"), os.environ.get("PDBEDA_FILE_CHUNK_KB", "8192"),
          np.median(t), 4e-6 * edge ** 3 / np.median(t), t.min()), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
