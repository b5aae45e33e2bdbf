import sys, os, io
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from conftest import load_case
from pdb_eda_amd import _native, ccp4
from oracle import oracle as ora
z, header, grid = load_case("orth")
ctx = _native.Context(0)
dm = ccp4.parse(io.BytesIO(z["ccp4_bytes"].tobytes()), "orth", ctx=ctx)
print("shape", grid.shape)
o = ora.Oracle(dm.header, grid)
for tag in ["p30", "n30", "p15", "n20"]:
    cut = float(z["full_%s_cut" % tag])
    for rep in range(3):
        bl = dm._map.full_blobs(cut)
        st = bl.stats()
        want = o.full_blobs(cut)
        bad = np.where(~np.isclose(st["totalDensity"], want["totalDensity"], rtol=1e-9))[0]
        print(tag, rep, len(st["n"]), "bad", bad[:10], st["totalDensity"][bad[:3]], want["totalDensity"][bad[:3]], st["n"][bad[:3]], bl.counters())
