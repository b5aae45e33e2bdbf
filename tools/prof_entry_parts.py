"""Wall time of the host-side parts of one multiple-structure entry (default 200^3 maps, ~500 atoms), without a profiler's overhead:
python tools/prof_entry_parts.py [reps [edge [residues]]]     (128 400 = the entry of bench.py's analysis_entry leg)"""
import sys, os, time, tempfile, shutil, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures as ms, densityAnalysis as da, structure, _native, ccp4

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
edge = int(sys.argv[2]) if len(sys.argv) > 2 else 200
n_res = int(sys.argv[3]) if len(sys.argv) > 3 else 100
_native.pin_to_device(0)
da.setGlobals(synthetic.synthetic_params())
acc = collections.defaultdict(float)


def timed(owner, name, label=None):
    fn = getattr(owner, name)
    raw = fn.__func__ if isinstance(fn, staticmethod) else fn
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return raw(*a, **k)
        finally:
            acc[label or name] += time.perf_counter() - t0
    setattr(owner, name, staticmethod(wrapper) if isinstance(owner.__dict__.get(name), staticmethod) else wrapper)


timed(structure.Columns, "__init__", "Columns.__init__ (walk of the structure)")
timed(da.DensityAnalysis, "_cloudInputsFixed", "_cloudInputsFixed")
timed(da.DensityAnalysis, "_cloudInputs", "_cloudInputs (incl. the two above)")
timed(da.DensityAnalysis, "_cloudStatistics", "_cloudStatistics")
timed(da.DensityAnalysis, "aggregateCloud", "aggregateCloud (all of it)")
timed(_native.DeviceMap, "aggregate_cloud", "pdbeda_aggregate_cloud")
timed(_native.DeviceMap, "stats", "pdbeda_map_stats")
timed(ms, "loadEntry", "loadEntry")
timed(_native.DeviceMap, "region_sums", "pdbeda_region_sums (two calls)")
timed(_native.Context, "symmetry_atoms", "pdbeda_symmetry_atoms")
timed(_native.Context, "nearest_atom", "pdbeda_nearest_atom")
timed(_native.BlobList, "stats", "pdbeda_bloblist_stats (two calls)")
timed(_native.DeviceMap, "full_blobs_pm", "pdbeda_full_blobs_pm")
timed(da.DensityAnalysis, "_calculateSymmetryAtoms", "_calculateSymmetryAtoms (incl. the library call)")
timed(da.DensityAnalysis, "_discrepancyColumns", "_discrepancyColumns (two calls)")
timed(da.DensityAnalysis, "calculateAtomRegionDiscrepancies", "calculateAtomRegionDiscrepancies (all of it)")
timed(da.DensityAnalysis, "calculateResidueRegionDiscrepancies", "calculateResidueRegionDiscrepancies (all of it)")
timed(da.DensityAnalysis, "calculateAtomSpecificBlobStatistics", "calculateAtomSpecificBlobStatistics (all of it)")
timed(ccp4.DensityBlob, "listFromDevice", "DensityBlob.listFromDevice (two calls)")
timed(structure, "read_pdb", "read_pdb")
tmp = tempfile.mkdtemp(prefix="pdbeda_parts_")
try:
    loaders = [synthetic.write_entry_files(tmp, "e%d" % k, edge, n_res, k, as_paths=True) for k in range(4)]
    ctx = _native.Context(0)
    for k in range(4):
        ms.analyzeEntry(ms.Entry("w%d" % k, loaders[k]), ctx, {}, True)
    acc.clear()
    t0 = time.perf_counter()
    for i in range(reps):
        assert ms.analyzeEntry(ms.Entry("e%d" % i, loaders[i % 4]), ctx, {}, True)
    total = (time.perf_counter() - t0) / reps
    print("analyzeEntry: %.2f ms per entry" % (1e3 * total))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-45s %.3f ms" % (k, 1e3 * v / reps))
    # the same wrappers around bench.py's analysis_entry leg (its entry, all four phases; 6 passes)
    import bench
    acc.clear()
    leg = bench.analysis_leg(ctx)
    print("analysis_entry leg: %s = %.2f ms" % (leg["ms"], leg["ms_per_entry"]))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
        print("  %-55s %.3f ms" % (k, 1e3 * v / 7))
finally:
    shutil.rmtree(tmp)
