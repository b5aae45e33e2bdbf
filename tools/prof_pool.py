"""One-worker and four-worker process pools on configs[3] entries, repeated: python tools/prof_pool.py"""
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures

if __name__ == "__main__":
    if os.environ.get("PROF_PIN", "1") == "1":
        from pdb_eda_amd import _native
        print("main pinned to", _native.pin_to_device(0), "cpus")
    tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
    try:
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k) for k in range(4)]
        for workers, as_paths in ((4, False), (4, True), (1, False), (1, True), (4, False), (4, True)):
            for ld in loaders:
                ld.as_paths = as_paths
            entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 4], cost_hint=0.0) for i in range(64)]
            print("as_paths", as_paths)
            pool = multipleStructures.ProcessPool(0, workers, params=synthetic.synthetic_params(), silent=True)
            try:
                pool.warm()
                pool.map(entries[:2 * workers])
                for rep in range(3):
                    t0 = time.perf_counter()
                    recs = pool.map(entries)
                    dt = time.perf_counter() - t0
                    print("workers %d rep %d: %.2f ms/entry (%d ok)" % (workers, rep, 1e3 * dt / len(entries), sum(1 for r in recs if r)), flush=True)
            finally:
                pool.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
