"""Multiple-structure mode on one GPU: where the time of an entry goes, by pool shape.  One script, four questions:

  python tools/prof_pool.py pools   [--edge 200] [--entries 64] [--workers 1,4] [--bytes]   process pools; --bytes hands the maps over as
                                                                                           bytes objects instead of file paths
  python tools/prof_pool.py threads [--workers 1,2,4,6]                                   thread pool (StreamPool) vs process pool
  python tools/prof_pool.py sizes   [--edges 200,128,96]                                  entry size: what is PCIe, what is host work
  python tools/prof_pool.py inside  [--workers 1,4]                                       inside a worker: loader / analyzeEntry / CPU time
"""
import argparse
import multiprocessing
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pdb_eda_amd import synthetic, multipleStructures, densityAnalysis, _native


def timed(entry):
    t0, c0 = time.perf_counter(), time.process_time()
    entry.loader()
    t1 = time.perf_counter()
    multipleStructures.analyzeEntry(entry, multipleStructures._worker_context(), {}, True)
    return (t1 - t0, time.perf_counter() - t1, time.process_time() - c0, len(os.sched_getaffinity(0)))


def entries_of(tmp, edge, n, distinct=4, as_paths=True, tag="e"):
    loaders = [synthetic.write_entry_files(tmp, "%s%d_%d" % (tag, edge, k), edge, 100, k, as_paths=as_paths) for k in range(distinct)]
    return [multipleStructures.Entry("e%04d" % i, loaders[i % distinct], cost_hint=0.0) for i in range(n)]


def run_process_pool(entries, workers, params, reps=3, label=""):
    pool = multipleStructures.ProcessPool(0, workers, params=params, silent=True)
    try:
        pool.warm()
        pool.map(entries[:2 * workers])
        best, ok = 1e9, 0
        for _ in range(reps):
            t0 = time.perf_counter()
            recs = pool.map(entries)
            best = min(best, time.perf_counter() - t0)
            ok = sum(1 for r in recs if r)
        print("%sprocesses %d: %.2f ms/entry (%d ok)" % (label, workers, 1e3 * best / len(entries), ok), flush=True)
    finally:
        pool.close()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["pools", "threads", "sizes", "inside"])
    ap.add_argument("--edge", type=int, default=200)
    ap.add_argument("--edges", default="200,128,96")
    ap.add_argument("--entries", type=int, default=64)
    ap.add_argument("--workers", default=None)
    ap.add_argument("--bytes", action="store_true")
    args = ap.parse_args()
    workers = [int(v) for v in (args.workers or {"pools": "1,4", "threads": "1,2,4,6", "sizes": "1,4", "inside": "1,4"}[args.what]).split(",")]
    print("main pinned to", _native.pin_to_device(0), "cpus")
    params = synthetic.synthetic_params()
    densityAnalysis.setGlobals(params)
    tmp = tempfile.mkdtemp(prefix="pdbeda_prof_")
    try:
        if args.what == "pools":
            entries = entries_of(tmp, args.edge, args.entries, as_paths=not args.bytes)
            for w in workers:
                run_process_pool(entries, w, params)
        elif args.what == "threads":
            entries = entries_of(tmp, args.edge, args.entries)
            fn = lambda e, ctx: multipleStructures.analyzeEntry(e, ctx, {}, True)      # noqa: E731
            for w in workers:
                pool = multipleStructures.StreamPool(0, w, silent=True)
                pool.map(fn, entries[:2 * w])
                t0 = time.perf_counter()
                recs = pool.map(fn, entries)
                print("threads %d: %.2f ms/entry (%d ok)" % (w, 1e3 * (time.perf_counter() - t0) / len(entries), sum(1 for r in recs if r)), flush=True)
                pool.close()
            for w in workers:
                run_process_pool(entries, w, params, reps=2)
        elif args.what == "sizes":
            for edge in [int(v) for v in args.edges.split(",")]:
                entries = entries_of(tmp, edge, args.entries, tag="s")
                for w in workers:
                    run_process_pool(entries, w, params, label="edge %d (%.0f MB per entry), " % (edge, 8e-6 * edge ** 3))
        else:
            entries = entries_of(tmp, args.edge, 32, distinct=2, as_paths=False)
            for w in workers:
                pool = multiprocessing.get_context("spawn").Pool(w, multipleStructures._worker_init, (0, params, 0.0, True))
                pool.map(timed, entries[:8], chunksize=1)
                t0 = time.perf_counter()
                out = pool.map(timed, entries, chunksize=1)
                dt, n = time.perf_counter() - t0, len(out)
                print("workers %d: wall %.2f ms/entry; inside a worker: loader %.2f, analyzeEntry (incl. a second load) %.2f ms, cpu %.2f ms; affinity %d cpus" %
                      (w, 1e3 * dt / n, 1e3 * sum(o[0] for o in out) / n, 1e3 * sum(o[1] for o in out) / n, 1e3 * sum(o[2] for o in out) / n, out[0][3]), flush=True)
                pool.close()
                pool.join()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
