"""Inside the workers of the multiple-structure pool: when does the loader of an entry run, when its analysis, who waits for whom.
A traced copy of multipleStructures._worker_chunk (same pipeline, time stamps added) on the bench's configuration:
  python tools/prof_pipeline2.py [workers] [entries] [eager 0/1]"""
import os
import shutil
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def traced_chunk(entries):
    import collections
    from pdb_eda_amd import multipleStructures as ms
    depth = int(os.environ.get("PIPE_DEPTH", ms.N_WORKER_CONTEXTS - 1))
    out, stamps = [], []

    def load(i, box):
        t0 = time.monotonic()
        try:
            box.append(ms.loadEntry(entries[i], ms._worker_context(i % ms.N_WORKER_CONTEXTS)))
        except BaseException as exception:
            box.append(exception)
        stamps.append(("load", i, t0, time.monotonic()))

    def start(i):
        box = []
        thread = threading.Thread(target=load, args=(i, box), daemon=True)
        thread.start()
        return thread, box
    t_chunk = time.monotonic()
    pending = collections.deque(start(k) for k in range(min(depth, len(entries))))
    for i, entry in enumerate(entries):
        thread, box = pending.popleft()
        t0 = time.monotonic()
        thread.join()
        t1 = time.monotonic()
        if i + depth < len(entries):
            pending.append(start(i + depth))
        record = ms.analyzeEntry(entry, ms._worker_context(i % ms.N_WORKER_CONTEXTS), {}, True, loaded=box[0])
        t2 = time.monotonic()
        stamps.append(("wait", i, t0, t1))
        stamps.append(("analyse", i, t1, t2))
        out.append(bool(record))
    return os.getpid(), t_chunk, time.monotonic(), stamps, out


def cpu_stat():
    """The cgroup's CPU accounting (a GPU box gives 16 CPUs' worth of time per 100 ms period, whatever the affinity mask says)."""
    try:
        return {k: int(v) for k, v in (line.split() for line in open("/sys/fs/cgroup/cpu.stat")) if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec")}
    except OSError:
        return {}


if __name__ == "__main__":
    import numpy as np
    from pdb_eda_amd import synthetic, multipleStructures, densityAnalysis, _native
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    n_entries = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    os.environ["PDBEDA_EAGER_DIFF_MAP"] = sys.argv[3] if len(sys.argv) > 3 else "1"
    _native.pin_to_device(0)
    params = synthetic.synthetic_params()
    densityAnalysis.setGlobals(params)
    tmp = tempfile.mkdtemp(prefix="pdbeda_pipe2_")
    try:
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(16)]
        entries = [multipleStructures.Entry("e%04d" % i, loaders[i % 16], cost_hint=0.0) for i in range(n_entries)]
        pool = multipleStructures.ProcessPool(0, workers, params=params, silent=True)
        try:
            pool.warm()
            pool.map(entries[:2 * workers])
            chunk = int(os.environ.get("PIPE_CHUNK", "8"))
            chunks = [entries[k:k + chunk] for k in range(0, len(entries), chunk)]
            pool.run(traced_chunk, chunks[:workers])
            cpu0 = cpu_stat()
            t0 = time.monotonic()
            res = pool.run(traced_chunk, chunks)
            wall = time.monotonic() - t0
            cpu1 = cpu_stat()
        finally:
            pool.close()
        dur = {"load": [], "wait": [], "analyse": []}
        first_load = []
        for pid, c0, c1, stamps, ok in res:
            for kind, i, a, b in stamps:
                dur[kind].append(b - a)
                if kind == "load" and i == 0:
                    first_load.append(b - a)
        print("depth %s chunk %s: workers %d, eager diff map %s: wall %.2f ms/entry (%d entries); per worker and entry: chunk time %.2f ms" %
              (os.environ.get("PIPE_DEPTH", "default"), os.environ.get("PIPE_CHUNK", "8"), workers, os.environ["PDBEDA_EAGER_DIFF_MAP"], 1e3 * wall / n_entries, n_entries, 1e3 * np.mean([(c1 - c0) / len(ok) for _, c0, c1, _, ok in res])))
        for kind in ("load", "analyse", "wait"):
            v = 1e3 * np.array(dur[kind])
            print("  %-8s mean %.2f  median %.2f  p90 %.2f ms" % (kind, v.mean(), np.median(v), np.percentile(v, 90)))
        print("  first load of a chunk (nothing overlaps it): mean %.2f ms" % (1e3 * np.mean(first_load)))
        if cpu0 and cpu1:
            d = {k: cpu1[k] - cpu0[k] for k in cpu0}
            print("  cgroup: %.1f CPUs busy on average; %d of %d periods throttled, %.2f s of thread time throttled" %
                  (1e-6 * d["usage_usec"] / wall, d["nr_throttled"], d["nr_periods"], 1e-6 * d["throttled_usec"]))
        busy = sum(c1 - c0 for _, c0, c1, _, _ in res)
        print("  sum of chunk times / (workers * wall) = %.2f (1.0 = every worker always inside a chunk)" % (busy / (workers * wall)))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
