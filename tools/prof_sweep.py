"""Optimise-mode iteration cost (BASELINE configs[4] shape): resident entries re-analysed under a parameter table, threads vs streams.
   python tools/prof_sweep.py"""
import sys, os, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from pdb_eda_amd import synthetic, multipleStructures, optimizeSweep, densityAnalysis, _native

if __name__ == "__main__":
    print("pinned", _native.pin_to_device(0))
    tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
    try:
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(4)]
        entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 4], cost_hint=0.0) for i in range(32)]
        sets = synthetic.sweep_param_sets()
        for streams in (1, 2, 4):
            t0 = time.perf_counter()
            sw = optimizeSweep.Sweep(entries, 0, streams)
            t_load = time.perf_counter() - t0
            sw.iteration(sets[0])
            ts = []
            for rep in range(2):
                for p in sets[:3]:
                    t0 = time.perf_counter()
                    sw.iteration(p)
                    ts.append(time.perf_counter() - t0)
            sw.close()
            print("streams %d: load %.1f ms/entry; iteration %.2f ms/entry (best of %d iterations over %d entries)" %
                  (streams, 1e3 * t_load / len(entries), 1e3 * min(ts) / len(entries), len(ts), len(entries)), flush=True)
        for workers in (2, 4):
            t0 = time.perf_counter()
            sw = optimizeSweep.ProcessSweep(entries, 0, workers)
            t_load = time.perf_counter() - t0
            sw.iteration(sets[0])
            ts = []
            for rep in range(2):
                for p in sets[:3]:
                    t0 = time.perf_counter()
                    sw.iteration(p)
                    ts.append(time.perf_counter() - t0)
            sw.close()
            print("processes %d: load %.1f ms/entry (incl. spawn); iteration %.2f ms/entry (best of %d iterations over %d entries)" %
                  (workers, 1e3 * t_load / len(entries), 1e3 * min(ts) / len(entries), len(ts), len(entries)), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
