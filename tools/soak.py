# Soak test: many random shapes / cutoffs / repetitions against the oracle (bit-exact counts, keys, labels).
import sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
from oracle import oracle as ora
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
ctx = _native.Context(0)
t0 = time.time()
bad = 0
for k in range(n_cases):
    nc = int(rng.choice([rng.integers(1, 80), rng.integers(60, 330), rng.integers(250, 700)]))
    nr, ns = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    if nc * nr * ns > 6_000_000:
        ns = max(1, 6_000_000 // (nc * nr))
    nsd = float(rng.choice([0.2, 0.6, 1.0, 1.5, 2.5]))
    g = synthetic.smooth_noise((ns, nr, nc), 5000 + k, float(rng.choice([0.7, 1.5, 3.0])))
    spec = synthetic.MapSpec(ncrs=(nc, nr, ns))
    dm = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, g)), "soak", ctx=ctx)
    o = ora.Oracle(dm.header, g)
    cut = dm.meanDensity + nsd * dm.stdDensity
    if not np.isfinite(cut) or cut <= 0:
        continue
    want = [o.full_blobs(cut, labels=True), o.full_blobs(-cut, labels=True)]
    for rep in range(3):
        lists = dm._map.full_blobs_pm(cut, -cut, labels=True)
        for bl, w in zip(lists, want):
            st = bl.stats()
            ok = (np.array_equal(st["n"], w["n"]) and np.array_equal(st["firstKey"], w["firstKey"]) and
                  np.allclose(st["totalDensity"], w["totalDensity"], rtol=1e-9) and np.array_equal(bl.labels(dm._map.unique_shape), w["labels"]))
            if not ok:
                bad += 1
                print("MISMATCH case", k, (ns, nr, nc), nsd, "rep", rep, flush=True)
        for bl in lists:
            bl.free()
    if k % 25 == 0:
        print("case", k, (ns, nr, nc), nsd, "%.0fs" % (time.time() - t0), flush=True)
print("done: %d cases, %d mismatches, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
