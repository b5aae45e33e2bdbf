"""Device-side split of the library calls of one analysis entry: python tools/prof_cloud.py"""
import sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from pdb_eda_amd import _native, ccp4, synthetic, structure, densityAnalysis as da

ctx = _native.Context(0)
edge, n_res = 128, 400
spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([edge - 7] * 3))
st = synthetic.chain_structure(n_res, 5, lo, hi, hetero_every=9, zero_occupancy_every=37)
params = synthetic.synthetic_params()
da.setGlobals(params)
dens = synthetic.gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=5)
diff = (synthetic.noise_grid(spec, 105, 1.2) * 0.12).astype(np.float32)
files = synthetic.ccp4_bytes(spec, dens), synthetic.ccp4_bytes(spec, diff)
rot = [np.hstack([np.eye(3), np.zeros((3, 1))]), np.array([[-1.0, 0, 0, 0.5 * header.xlength], [0, -1.0, 0, 0], [0, 0, 1.0, 0.5 * header.zlength]])]
pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="synth", resolution=2.0, spaceGroup="P_1", rotationMats=rot))

calls = {}
for cls, names in ((_native.DeviceMap, ["aggregate_cloud", "region_sums", "full_blobs_pm", "stats", "sum_of_abs"]), (_native.Context, ["symmetry_atoms", "nearest_atom"]),
                   (_native.BlobList, ["stats"])):
    for name in names:
        def wrap(orig, label):
            def f(*a, **k):
                t0 = time.perf_counter()
                try:
                    return orig(*a, **k)
                finally:
                    calls.setdefault(label, []).append(time.perf_counter() - t0)
            return f
        setattr(cls, name, wrap(getattr(cls, name), cls.__name__ + "." + name))

for rep in range(4):
    calls.clear()
    st.__dict__.pop("_pdbeda_columns", None)
    d = ccp4.parse(io.BytesIO(files[0]), "synth", ctx=ctx)
    f = ccp4.parse(io.BytesIO(files[1]), "synth", ctx=ctx)
    da._attachCutoffs(d, f)
    an = da.DensityAnalysis("synth", d, f, st, pdbObj)
    if rep == 3:
        ctx.profile_begin()
    an.aggregateCloud()
    an.calculateAtomRegionDiscrepancies(3.5, 3.0, "")
    an.calculateResidueRegionDiscrepancies(3.5, 3.0, "")
    an.calculateAtomSpecificBlobStatistics(an.greenBlobList + an.redBlobList)
prof = ctx.profile_end()
print("library calls (ms, last repetition):")
for k, v in calls.items():
    print("  %-28s %d x  %.3f" % (k, len(v), 1e3 * sum(v)))
print("kernels (count, total ms):")
for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s %3d  %.4f" % (k, n, ms))
print("kernel total ms %.3f" % sum(ms for _, ms in prof.values()))
