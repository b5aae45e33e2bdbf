"""Thread pool vs process pool on configs[3] entries handed over as file paths: python tools/prof_pool3.py"""
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures, densityAnalysis, _native

if __name__ == "__main__":
    print("main pinned to", _native.pin_to_device(0), "cpus")
    tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
    try:
        params = synthetic.synthetic_params()
        densityAnalysis.setGlobals(params)
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(4)]
        entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 4], cost_hint=0.0) for i in range(96)]
        for workers in (1, 2, 4, 6):
            pool = multipleStructures.StreamPool(0, workers, silent=True)
            fn = lambda e, ctx: multipleStructures.analyzeEntry(e, ctx, {}, True)
            pool.map(fn, entries[:2 * workers])
            for rep in range(2):
                t0 = time.perf_counter()
                recs = pool.map(fn, entries)
                dt = time.perf_counter() - t0
                print("threads %d rep %d: %.2f ms/entry (%d ok)" % (workers, rep, 1e3 * dt / len(entries), sum(1 for r in recs if r)), flush=True)
            pool.close()
        for workers in (2, 4, 5):
            pool = multipleStructures.ProcessPool(0, workers, params=params, silent=True)
            try:
                pool.warm()
                pool.map(entries[:2 * workers])
                for rep in range(2):
                    t0 = time.perf_counter()
                    recs = pool.map(entries)
                    dt = time.perf_counter() - t0
                    print("processes %d rep %d: %.2f ms/entry (%d ok)" % (workers, rep, 1e3 * dt / len(entries), sum(1 for r in recs if r)), flush=True)
            finally:
                pool.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
