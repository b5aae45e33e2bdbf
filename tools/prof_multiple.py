"""Per-entry cost of the multiple-structure worker, in-process (configs[3] entry: 200^3 maps, ~500 atoms): python tools/prof_multiple.py"""
import sys, os, time, tempfile, shutil, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PROF_TORCH", "1") == "1":
    import torch  # noqa: F401
from pdb_eda_amd import _native, synthetic, multipleStructures, densityAnalysis

densityAnalysis.setGlobals(synthetic.synthetic_params())
tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
try:
    loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(2)]
    entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 2], cost_hint=0.0) for i in range(24)]
    ctx = _native.Context(0)
    failures = {}
    for e in entries[:4]:
        multipleStructures.analyzeEntry(e, ctx, failures, True)
    ts = []
    for e in entries:
        t0 = time.perf_counter()
        r = multipleStructures.analyzeEntry(e, ctx, failures, True)
        ts.append(time.perf_counter() - t0)
    print("per entry ms: min %.2f median %.2f max %.2f; failures %s" % (1e3 * min(ts), 1e3 * sorted(ts)[len(ts) // 2], 1e3 * max(ts), failures))
    pr = cProfile.Profile(); pr.enable()
    for e in entries[:8]:
        multipleStructures.analyzeEntry(e, ctx, failures, True)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30); pstats.Stats(pr).sort_stats("cumtime").print_stats(45)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
