"""Where a one-worker process pool spends its time per entry: python tools/prof_pool2.py"""
import sys, os, time, tempfile, shutil, multiprocessing
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures


def timed(entry):
    t0 = time.perf_counter()
    c0 = time.process_time()
    dens, diff, st, pdb = entry.loader()
    t1 = time.perf_counter()
    rec = multipleStructures.analyzeEntry(entry, multipleStructures._worker_context(), {}, True)
    t2 = time.perf_counter()
    return (t1 - t0, t2 - t1, time.process_time() - c0, os.sched_getaffinity(0).__len__())


if __name__ == "__main__":
    tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
    try:
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k) for k in range(2)]
        entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 2], cost_hint=0.0) for i in range(32)]
        for workers in (1, 4):
            pool = multiprocessing.get_context("spawn").Pool(workers, multipleStructures._worker_init, (0, synthetic.synthetic_params(), 0.0, True))
            pool.map(timed, entries[:8], chunksize=1)
            t0 = time.perf_counter()
            out = pool.map(timed, entries, chunksize=1)
            dt = time.perf_counter() - t0
            n = len(out)
            print("workers %d: wall %.2f ms/entry; inside worker: loader %.2f, analyzeEntry(incl. a second load) %.2f ms, cpu %.2f ms; affinity %d cpus" %
                  (workers, 1e3 * dt / n, 1e3 * sum(o[0] for o in out) / n, 1e3 * sum(o[1] for o in out) / n, 1e3 * sum(o[2] for o in out) / n, out[0][3]), flush=True)
            pool.close(); pool.join()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
