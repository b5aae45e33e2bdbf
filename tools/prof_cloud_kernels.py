# Device time per kernel of DensityAnalysis.aggregateCloud on the synthetic ~2 A entry (HIP events inside the library).
import sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, structure, densityAnalysis as da
edge, n_res = 128, 400
spec = synthetic.MapSpec(ncrs=(edge,) * 3, spacing=0.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([edge - 7] * 3))
st = synthetic.chain_structure(n_res, 5, lo, hi, hetero_every=9, zero_occupancy_every=37)
params = synthetic.synthetic_params(); da.setGlobals(params)
dens = synthetic.gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=5)
diff = (synthetic.noise_grid(spec, 105, 1.2) * 0.12).astype(np.float32)
ctx = _native.Context(0)
rot = [np.hstack([np.eye(3), np.zeros((3, 1))])]
pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="t", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
for rep in range(2):
    densityObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), "t", ctx=ctx)
    diffObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), "t", ctx=ctx)
    da._attachCutoffs(densityObj, diffObj)
    an = da.DensityAnalysis("t", densityObj, diffObj, st, pdbObj)
    ctx.profile_begin()
    t0 = time.perf_counter()
    an.aggregateCloud()
    wall = time.perf_counter() - t0
    prof = ctx.profile_end()
print("aggregateCloud wall %.1f ms, device %.2f ms" % (wall * 1e3, sum(ms for _, ms in prof.values())))
for k, (c, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print("  %-18s calls %2d  %.3f ms" % (k, c, ms))
