"""cProfile of the analysis entry of bench.py (host side), one table per phase: python tools/prof_analysis.py [reps]"""
import sys, os, io, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process: torch first)
from pdb_eda_amd import _native, ccp4, synthetic, structure, densityAnalysis as da

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
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

phases = ["parse_upload", "aggregateCloud", "region_discrepancies", "blob_statistics"]
profs = {p: cProfile.Profile() for p in phases}
wall = {p: [] for p in phases}


def once(profile):
    def run(name, fn):
        t0 = time.perf_counter()
        if profile:
            profs[name].enable()
        out = fn()
        if profile:
            profs[name].disable()
        else:
            wall[name].append(time.perf_counter() - t0)
        return out

    st.__dict__.pop("_pdbeda_columns", None)          # every repetition is a fresh entry

    def parse():
        d = ccp4.parse(io.BytesIO(files[0]), "synth", ctx=ctx)
        f = ccp4.parse(io.BytesIO(files[1]), "synth", ctx=ctx)
        da._attachCutoffs(d, f)
        return da.DensityAnalysis("synth", d, f, st, pdbObj)
    an = run("parse_upload", parse)
    run("aggregateCloud", an.aggregateCloud)
    run("region_discrepancies", lambda: (an.calculateAtomRegionDiscrepancies(3.5, 3.0, ""), an.calculateResidueRegionDiscrepancies(3.5, 3.0, "")))
    run("blob_statistics", lambda: an.calculateAtomSpecificBlobStatistics(an.greenBlobList + an.redBlobList))


once(False)
for _ in range(reps):
    once(False)
print({p: round(1e3 * min(v[1:]), 2) for p, v in wall.items()}, "ms (best of %d, unprofiled)" % reps)
for _ in range(reps):
    once(True)
for p in phases:
    print("=" * 30, p, "(cumulative over %d reps)" % reps)
    s = pstats.Stats(profs[p]); s.sort_stats("tottime").print_stats(14)
