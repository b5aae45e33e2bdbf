# Times the densityAnalysis path (aggregateCloud, region discrepancies, blob statistics) on a synthetic "~2 A entry"
# (SURVEY 8d config 3 stand-in): n_res residues of a poly-ALA random walk in a 0.5 A grid.
import sys, os, io, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, structure, densityAnalysis as da

n_res = int(sys.argv[1]) if len(sys.argv) > 1 else 400
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 128
spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
lo = np.array(header.crs2xyzCoord([6, 6, 6])); hi = np.array(header.crs2xyzCoord([edge - 7] * 3))
st = synthetic.chain_structure(n_res, 5, lo, hi, hetero_every=9, zero_occupancy_every=37)
params = synthetic.synthetic_params()
da.setGlobals(params)
t0 = time.perf_counter()
dens = synthetic.gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=5)
diff = (synthetic.noise_grid(spec, 105, 1.2) * 0.12).astype(np.float32)
print("synth %.2fs, atoms %d" % (time.perf_counter() - t0, len(list(st.get_atoms()))), flush=True)
ctx = _native.Context(0)
rot = [np.hstack([np.eye(3), np.zeros((3, 1))]), np.array([[-1.0, 0, 0, 0.5 * header.xlength], [0, -1.0, 0, 0], [0, 0, 1.0, 0.5 * header.zlength]])]
pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="t", resolution=2.0, spaceGroup="P_1", rotationMats=rot))

def run():
    t = {}
    t0 = time.perf_counter()
    densityObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), "t", ctx=ctx)
    diffObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), "t", ctx=ctx)
    da._attachCutoffs(densityObj, diffObj)
    an = da.DensityAnalysis("t", densityObj, diffObj, st, pdbObj)
    t["parse+upload"] = time.perf_counter() - t0; t0 = time.perf_counter()
    an.aggregateCloud()
    t["aggregateCloud"] = time.perf_counter() - t0; t0 = time.perf_counter()
    r = an.calculateAtomRegionDiscrepancies(3.5, 3.0, "")
    t["atomRegionDiscrepancies"] = time.perf_counter() - t0; t0 = time.perf_counter()
    r2 = an.calculateResidueRegionDiscrepancies(3.5, 3.0, "")
    t["residueRegionDiscrepancies"] = time.perf_counter() - t0; t0 = time.perf_counter()
    b = an.calculateAtomSpecificBlobStatistics(an.greenBlobList + an.redBlobList)
    t["blobStatistics"] = time.perf_counter() - t0
    return an, t

an, t = run()
an, t = run()
print({k: round(v * 1e3, 1) for k, v in t.items()}, "ms; ratio", an.densityElectronRatio, "atoms analysed", len(an.atomCloudDescriptions))
if len(sys.argv) > 3:
    pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
