# Process pool of multiple-structure mode: entries per task (the worker pipelines load and analysis inside a task).  python tools/exp/pool_chunks.py
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pdb_eda_amd import synthetic, multipleStructures, densityAnalysis, _native
if __name__ == "__main__":
    _native.pin_to_device(0)
    params = synthetic.synthetic_params()
    tmp = tempfile.mkdtemp(prefix="pdbeda_ct_")
    try:
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(8)]
        entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 8], cost_hint=0.0) for i in range(256)]
        pool = multipleStructures.ProcessPool(0, 4, params=params, silent=True)
        pool.warm(); pool.map(entries[:16])
        for chunk in (4, 8, 16, 32):
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); recs = pool.map(entries, chunk=chunk); best = min(best, time.perf_counter() - t0)
            print("chunk %d: %.2f ms/entry (%d ok)" % (chunk, 1e3 * best / len(entries), sum(1 for r in recs if r)), flush=True)
        pool.close()
    finally:
        shutil.rmtree(tmp)
