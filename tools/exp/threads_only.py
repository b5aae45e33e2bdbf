# StreamPool (threads in ONE process, a context = stream each; an entry is loaded and analysed by the thread that took it) by lane count:
#   python tools/exp/threads_only.py 4 6 8 [--entries 256]
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pdb_eda_amd import _native, synthetic, multipleStructures, densityAnalysis
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_entries = int(sys.argv[sys.argv.index("--entries") + 1]) if "--entries" in sys.argv else 256
if "--entries" in sys.argv: args.remove(sys.argv[sys.argv.index("--entries") + 1])
_native.pin_to_device(0)
densityAnalysis.setGlobals(synthetic.synthetic_params())
if "--switch" in sys.argv:
    sys.setswitchinterval(float(sys.argv[sys.argv.index("--switch") + 1])); args.remove(sys.argv[sys.argv.index("--switch") + 1])
tmp = tempfile.mkdtemp(prefix="pdbeda_thr_")
try:
    loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(16)]
    entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 16], cost_hint=0.0) for i in range(n_entries)]
    fn = lambda e, ctx: multipleStructures.analyzeEntry(e, ctx, {}, True)
    for w in [int(a) for a in args]:
        pool = multipleStructures.StreamPool(0, w, silent=True)
        pool.map(fn, entries[:2 * w])
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            recs = pool.map(fn, entries)
            best = min(best, time.perf_counter() - t0)
        print("threads %d: %.2f ms/entry (%d ok) eager=%s" % (w, 1e3 * best / len(entries), sum(1 for r in recs if r), os.environ.get("PDBEDA_EAGER_DIFF_MAP", "0")), flush=True)
        pool.close()
finally:
    shutil.rmtree(tmp, ignore_errors=True)
