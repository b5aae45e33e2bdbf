# Which library calls does one analyzeEntry of the multiple-structure mode make (resident entry)?  Counts by name, and each call's time alone.
import os, sys, time, tempfile, shutil, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pdb_eda_amd import _native, synthetic, multipleStructures, densityAnalysis
_native.pin_to_device(0)
densityAnalysis.setGlobals(synthetic.synthetic_params())
tmp = tempfile.mkdtemp(prefix="pdbeda_calls_")
try:
    loader = synthetic.write_entry_files(tmp, "e0", 200, 100, 0, as_paths=True)
    ctx = _native.Context(0)
    os.environ["PDBEDA_EAGER_DIFF_MAP"] = sys.argv[1] if len(sys.argv) > 1 else "0"
    entry = multipleStructures.Entry("e0", loader)
    loaded = multipleStructures.loadEntry(entry, ctx)
    for _ in range(5):
        multipleStructures.analyzeEntry(entry, ctx, {}, True, loaded=loaded)
    calls, spent = collections.Counter(), collections.Counter()
    check = _native.Context.check
    last = [time.perf_counter()]

    def counted(self, rc, what="call"):
        calls[what] += 1
        return check(self, rc, what)
    _native.Context.check = counted
    t0 = time.perf_counter()
    multipleStructures.analyzeEntry(entry, ctx, {}, True, loaded=loaded)
    dt = time.perf_counter() - t0
    _native.Context.check = check
    print("one analyzeEntry: %.2f ms, %d checked library calls" % (1e3 * dt, sum(calls.values())))
    for name, n in calls.most_common():
        print("  %3d  %s" % (n, name))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
