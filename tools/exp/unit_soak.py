# Unit tiles under load: two PROCESSES x eight streams label dense maps (every tile beyond LDS) over and over and compare with the oracle.
#   python tools/exp/unit_soak.py [rounds]      (the parent starts one child; both run the same loop)
import os, sys, subprocess, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
rounds = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "child" else 20
child = None
if "child" not in sys.argv:
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "child", str(rounds)])
else:
    rounds = int(sys.argv[2])
from oracle import oracle as ora
from pdb_eda_amd import ccp4, synthetic, multipleStructures
shape = (24, 40, 600)
g = synthetic.smooth_noise(shape, 9, 0.5)
blob = synthetic.ccp4_bytes(synthetic.MapSpec(ncrs=shape[::-1]), g)
header = ccp4.DensityHeader.fromFileHeader(blob[:1024])
cut = float(np.mean(g, dtype=np.float64)) + 0.3 * float(np.std(g.astype(np.float64)))
want = ora.Oracle(header, g).full_blobs(cut, labels=True)
def work(k, ctx):
    dm = ccp4.parse(io.BytesIO(blob), "d%d" % k, ctx=ctx)
    green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
    st = green.stats()
    return bool(np.array_equal(st["n"], want["n"]) and np.array_equal(st["firstKey"], want["firstKey"]) and np.array_equal(green.labels(dm._map.unique_shape), want["labels"]))
pool = multipleStructures.StreamPool(device=0, n_streams=8)
ok = 0
for r in range(rounds):
    res = pool.map(work, list(range(16)))
    assert all(res), (r, res)
    ok += 1
print("%s: %d / %d rounds of 16 jobs on 8 streams equal to the oracle" % ("child" if child is None else "parent", ok, rounds), flush=True)
if child is not None:
    sys.exit(child.wait())
