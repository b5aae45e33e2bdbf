"""Wall time of the host-side parts of one multiple-structure entry (200^3 maps, ~500 atoms), without a profiler's overhead:
python tools/prof_entry_parts.py [reps]"""
import sys, os, time, tempfile, shutil, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures as ms, densityAnalysis as da, structure, _native

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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
tmp = tempfile.mkdtemp(prefix="pdbeda_parts_")
try:
    loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(4)]
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
finally:
    shutil.rmtree(tmp)
