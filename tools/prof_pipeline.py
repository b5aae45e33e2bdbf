"""Inside one worker: load (other thread, other context) and analysis times of the two-stage pipeline: python tools/prof_pipeline.py"""
import sys, os, time, tempfile, shutil, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures as ms, densityAnalysis, _native

print("pinned", _native.pin_to_device(0))
sys.setswitchinterval(float(os.environ.get("PROF_SWITCH", "0.005")))
densityAnalysis.setGlobals(synthetic.synthetic_params())
tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
try:
    loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(4)]
    entries = [ms.Entry("e%04d" % i, loaders[i % 4]) for i in range(48)]
    ctx = [_native.Context(0), _native.Context(0)]
    if os.environ.get("PROF_TIMEOUT"):
        for c in ctx:
            c.set_timeout(float(os.environ["PROF_TIMEOUT"]))
    for e in entries[:4]:
        ms.analyzeEntry(e, ctx[0], {}, True); ms.analyzeEntry(e, ctx[1], {}, True)
    # sequential
    tl = ta = 0.0
    t0 = time.perf_counter()
    for i, e in enumerate(entries):
        t1 = time.perf_counter(); loaded = ms.loadEntry(e, ctx[i % 2]); t2 = time.perf_counter()
        ms.analyzeEntry(e, ctx[i % 2], {}, True, loaded=loaded); t3 = time.perf_counter()
        tl += t2 - t1; ta += t3 - t2
    n = len(entries)
    print("sequential: %.2f ms/entry (load %.2f, analyse %.2f)" % (1e3 * (time.perf_counter() - t0) / n, 1e3 * tl / n, 1e3 * ta / n))
    # pipelined
    loads, waits, anas = [], [], []
    def load(i, box):
        t1 = time.perf_counter()
        box.append(ms.loadEntry(entries[i], ctx[i % 2]))
        loads.append(time.perf_counter() - t1)
    def start(i):
        box = []; th = threading.Thread(target=load, args=(i, box)); th.start(); return th, box
    t0 = time.perf_counter()
    pending = start(0)
    for i, e in enumerate(entries):
        th, box = pending
        t1 = time.perf_counter(); th.join(); t2 = time.perf_counter()
        pending = start(i + 1) if i + 1 < n else None
        ms.analyzeEntry(e, ctx[i % 2], {}, True, loaded=box[0]); t3 = time.perf_counter()
        waits.append(t2 - t1); anas.append(t3 - t2)
    print("pipelined: %.2f ms/entry (load in thread %.2f, wait for it %.2f, analyse %.2f)" %
          (1e3 * (time.perf_counter() - t0) / n, 1e3 * sum(loads) / n, 1e3 * sum(waits) / n, 1e3 * sum(anas) / n))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
