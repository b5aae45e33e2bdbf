# Whole-map labelling of a protein-like 2Fo-Fc map (one giant chain blob at 1.5 sigma): robustness of the merge kernels.
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_res = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([edge - 7] * 3))
st = synthetic.chain_structure(n_res, 5, lo, hi)
params = synthetic.synthetic_params()
t0 = time.perf_counter()
dens = synthetic.gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=5)
print("synth %.1fs" % (time.perf_counter() - t0), flush=True)
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, dens, header.geometry())
mean, std = dmap.stats()
for nsd in (1.5, 1.0, 0.5, 3.0):
    cut = mean + nsd * std
    for _ in range(2):
        k = dmap.full_blobs(cut, labels=True)
    ctx.synchronize()
    ctx.profile_begin()
    for _ in range(5):
        k = dmap.full_blobs(cut, labels=True)
    prof = ctx.profile_end()
    st_ = k.stats()
    tot = sum(ms / c for c, ms in prof.values())
    print("nsd %.1f blobs %d biggest %d sig %.2f%% total %.0f us" % (nsd, len(st_["n"]), st_["n"].max(), 100.0 * st_["n"].sum() / dens.size, tot * 1e3),
          {a: round(ms / c * 1e3, 1) for a, (c, ms) in sorted(prof.items())}, k.counters(), flush=True)
