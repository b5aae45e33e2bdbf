import sys, os, io, pickle
sys.path.insert(0, os.getcwd())
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, structure, densityAnalysis as da
ctx = _native.Context(0)
ncrs, n_res, seed, spacing = synthetic.BIG_CASES["c2_bench_entry"]
spec, header, st, params, dens, diff, rot = synthetic.cube_entry(ncrs, n_res, seed, spacing)
da.setGlobals(params)
files = synthetic.ccp4_bytes(spec, dens), synthetic.ccp4_bytes(spec, diff)
pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="synth", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
d = ccp4.parse(io.BytesIO(files[0]), "synth", ctx=ctx); f = ccp4.parse(io.BytesIO(files[1]), "synth", ctx=ctx)
da._attachCutoffs(d, f)
an = da.DensityAnalysis("synth", d, f, st, pdbObj)
keep = {}
orig = da.DensityAnalysis._cloudStatistics
def spy(inp, res, ratio, unitVolume, typeMap):
    keep.update(inp={k: v for k, v in inp.items() if k != "cols"}, res=res, ratio=ratio, unitVolume=unitVolume)
    return orig(inp, res, ratio, unitVolume, typeMap)
da.DensityAnalysis._cloudStatistics = staticmethod(spy)
an.aggregateCloud()
g = an.greenBlobList; r = an.redBlobList
keep["green_stats"] = g[0]._list.stats(); keep["red_stats"] = r[0]._list.stats()
an._calculateSymmetryAtoms()
sa = an._symmetryAtoms
keep["sym"] = {k: getattr(sa, k) for k in sa.__dict__ if isinstance(getattr(sa, k), np.ndarray)}
keep["symCoords"] = an._symmetryAtomCoords
cen = np.array([b.centroid for b in g + r]); 
keep["nearest"] = ctx.nearest_atom(cen, np.asarray(an.symmetryAtomCoords, dtype=np.float64))
pickle.dump(keep, open("gpurun_out/c2_dump.pkl", "wb"))
print("ok", len(g), len(r))
