# Where the wall time of the analysis entry's region tables and blob statistics goes, call by call (wrapped functions, inclusive times; bench.py's entry):
#   python tools/exp/time_tables.py
import io, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, structure
from pdb_eda_amd import densityAnalysis as da

spent, calls = collections.Counter(), collections.Counter()


def wrap(owner, name, label=None):
    fn = getattr(owner, name)
    label = label or name

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            spent[label] += time.perf_counter() - t0
            calls[label] += 1
    setattr(owner, name, timed)


ctx = _native.Context(0)
spec, header, st, params, dens, diff, rot = synthetic.cube_entry((128, 128, 128), 400, 11, 0.5)
da.setGlobals(params)
files = [synthetic.ccp4_bytes(spec, dens), synthetic.ccp4_bytes(spec, diff)]
pdb = structure.PDBEntry(structure.PDBHeader(pdbid="t", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
for name in ("region_sums", "sum_of_abs", "full_blobs_pm", "crs2xyz"):
    wrap(_native.DeviceMap, name)
for name in ("nearest_atom", "symmetry_atoms"):
    wrap(_native.Context, name)
wrap(_native.BlobList, "stats", "BlobList.stats")
wrap(da.DensityAnalysis, "_calculateSymmetryAtoms")
wrap(da.DensityAnalysis, "_atomPick")
wrap(da.DensityAnalysis, "_residuePick")
wrap(da.DensityAnalysis, "_discrepancyColumns")
wrap(da._SymAtomList, "columns", "_SymAtomList.columns")
wrap(ccp4.DeviceBlobs, "columns", "DeviceBlobs.columns")
da.DensityAnalysis._rows = staticmethod(da.DensityAnalysis._rows)
raw_rows = da.DensityAnalysis._rows


def rows(*c):
    t0 = time.perf_counter()
    try:
        return raw_rows(*c)
    finally:
        spent["_rows"] += time.perf_counter() - t0
        calls["_rows"] += 1
da.DensityAnalysis._rows = staticmethod(rows)

reps, total = 12, collections.Counter()
for rep in range(reps + 2):
    st.__dict__.pop("_pdbeda_columns", None)
    d = ccp4.parse(io.BytesIO(files[0]), "t", ctx=ctx)
    f = ccp4.parse(io.BytesIO(files[1]), "t", ctx=ctx)
    da._attachCutoffs(d, f)
    an = da.DensityAnalysis("t", d, f, st, pdb)
    an.aggregateCloud()
    if rep == 2:
        spent.clear(); calls.clear(); total.clear()
    t0 = time.perf_counter()
    an.calculateAtomRegionDiscrepancies(3.5, 3.0, "")
    an.calculateResidueRegionDiscrepancies(3.5, 3.0, "")
    t1 = time.perf_counter()
    an.calculateAtomSpecificBlobStatistics(an.greenBlobList + an.redBlobList)
    t2 = time.perf_counter()
    total["region tables"] += t1 - t0
    total["blob statistics"] += t2 - t1
print("per entry (ms): " + ", ".join("%s %.3f" % (k, 1e3 * v / reps) for k, v in total.items()))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print("  %-28s %6.3f ms  (%d calls)" % (k, 1e3 * v / reps, calls[k] // reps))
