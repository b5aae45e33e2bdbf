# Where the wall time of DensityAnalysis.aggregateCloud goes on bench.py's analysis entry, call by call (wrapped functions, inclusive times):
#   python tools/exp/time_cloud.py
import io, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, structure
from pdb_eda_amd import densityAnalysis as da

spent, calls = collections.Counter(), collections.Counter()


def wrap(owner, name, label=None, static=False):
    fn = getattr(owner, name)
    label = label or name

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            spent[label] += time.perf_counter() - t0
            calls[label] += 1
    setattr(owner, name, staticmethod(timed) if static else timed)


ctx = _native.Context(0)
spec, header, st, params, dens, diff, rot = synthetic.cube_entry((128, 128, 128), 400, 11, 0.5)
da.setGlobals(params)
files = [synthetic.ccp4_bytes(spec, dens), synthetic.ccp4_bytes(spec, diff)]
pdb = structure.PDBEntry(structure.PDBHeader(pdbid="t", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
wrap(_native.DeviceMap, "aggregate_cloud")
wrap(da.DensityAnalysis, "_cloudInputs")
wrap(da.DensityAnalysis, "_cloudInputsFixed", static=True)
wrap(da.DensityAnalysis, "_cloudStatistics", static=True)
wrap(structure, "columns", "structure.columns")
lib = ctx._lib
raw = lib.pdbeda_aggregate_cloud


def lib_call(*a):
    t0 = time.perf_counter()
    try:
        return raw(*a)
    finally:
        spent["  pdbeda_aggregate_cloud (C)"] += time.perf_counter() - t0
        calls["  pdbeda_aggregate_cloud (C)"] += 1
lib.pdbeda_aggregate_cloud = lib_call
reps, total = 12, 0.0
for rep in range(reps + 2):
    st.__dict__.pop("_pdbeda_columns", None)
    d = ccp4.parse(io.BytesIO(files[0]), "t", ctx=ctx)
    f = ccp4.parse(io.BytesIO(files[1]), "t", ctx=ctx)
    da._attachCutoffs(d, f)
    an = da.DensityAnalysis("t", d, f, st, pdb)
    if rep == 2:
        spent.clear(); calls.clear(); total = 0.0
    t0 = time.perf_counter()
    an.aggregateCloud()
    total += time.perf_counter() - t0
print("aggregateCloud per entry: %.3f ms" % (1e3 * total / reps))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print("  %-32s %6.3f ms  (%d calls)" % (k, 1e3 * v / reps, calls[k] // reps))
