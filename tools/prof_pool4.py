"""What bounds the process pool: entries with maps of different sizes (same 500-atom model): python tools/prof_pool4.py"""
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures, _native

if __name__ == "__main__":
    print("main pinned to", _native.pin_to_device(0), "cpus")
    tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
    try:
        params = synthetic.synthetic_params()
        for edge in (200, 128, 96):
            loaders = [synthetic.write_entry_files(tmp, "e%d_%d" % (edge, k), edge, 100, k, spacing=0.5 * 200 / edge if False else 0.5, as_paths=True) for k in range(4)]
            entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 4], cost_hint=0.0) for i in range(96)]
            for workers in (1, 4):
                pool = multipleStructures.ProcessPool(0, workers, params=params, silent=True)
                try:
                    pool.warm()
                    pool.map(entries[:2 * workers])
                    best = 1e9
                    for rep in range(3):
                        t0 = time.perf_counter()
                        recs = pool.map(entries)
                        best = min(best, time.perf_counter() - t0)
                    print("edge %d (%.0f MB per entry), workers %d: %.2f ms/entry (%d ok)" % (edge, 8e-6 * edge ** 3, workers, 1e3 * best / len(entries), sum(1 for r in recs if r)), flush=True)
                finally:
                    pool.close()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
