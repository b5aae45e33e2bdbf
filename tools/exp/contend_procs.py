# Across PROCESSES: P processes (spawn) each analyse a RESIDENT entry over and over on their own context; U uploader threads in each upload
# maps from files beside it.  What an analysis costs alone, beside other processes' analyses, and beside everybody's uploads.
#   python tools/exp/contend_procs.py "P,U[,W]" ...      e.g. 1,0 2,0 4,0 4,1 4,3,1      (P + 1 processes use the GPU: P <= 5 on the pool's boxes;
#   W: only the first W processes run their uploaders)
import os, sys, time, threading, tempfile, shutil, multiprocessing
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def work(args):
    k, loaders, n_up, barrier, seconds = args
    from pdb_eda_amd import _native, ccp4, synthetic, multipleStructures, densityAnalysis
    _native.pin_to_device(0)
    densityAnalysis.setGlobals(synthetic.synthetic_params())
    ctx = _native.Context(0)
    os.environ["PDBEDA_EAGER_DIFF_MAP"] = "1"
    entry = multipleStructures.Entry("e%d" % k, loaders[k % 4])
    loaded = multipleStructures.loadEntry(entry, ctx)
    for _ in range(10):
        multipleStructures.analyzeEntry(entry, ctx, {}, True, loaded=loaded)
    stop = threading.Event()
    moved = [0] * max(n_up, 1)

    def uploader(u):
        c = _native.Context(0)
        head = ccp4.read(loaders[0].density_path, "u", ctx=c, lazy=True)
        geom, off = head.header.geometry(), 1024 + head.header.symmetryBytes
        i = u
        while not stop.is_set():
            l = loaders[i % len(loaders)]
            for path in (l.density_path, l.diff_path):
                _native.DeviceMap.from_file(c, path, off, False, geom).free()
                moved[u] += 4 * 200 ** 3
            i += 1
    threads = [threading.Thread(target=uploader, args=(u,), daemon=True) for u in range(n_up)]
    for t in threads:
        t.start()
    barrier.wait()
    ts, t_end, m0, t0 = [], time.perf_counter() + seconds, sum(moved), time.perf_counter()
    prim = os.environ.get("CONTEND_PRIMITIVE", "")
    dmap = loaded[0]._map if prim else None        # the resident 2Fo-Fc map
    pts = np.random.default_rng(k).integers(10, 190, size=(100, 3)).astype(np.int32)
    big = np.random.default_rng(k).integers(10, 190, size=(20000, 3)).astype(np.int32)
    pure = n_up > 0 and os.environ.get("CONTEND_PURE_UPLOADER", "") == "1"      # a process that uploads does nothing else
    while time.perf_counter() < t_end:
        t1 = time.perf_counter()
        if pure:
            time.sleep(0.05)
        elif prim == "sum":            # two kernels, a result copied out, one wait: nothing read from the host
            dmap.sum_of_abs(0.5)
        elif prim == "points":       # 1.2 KB staged in, one kernel, 800 B out, one wait
            dmap.point_density(pts)
        elif prim == "points_big":   # 240 KB staged in, one kernel, 160 KB out, one wait
            dmap.point_density(big)
        else:
            multipleStructures.analyzeEntry(entry, ctx, {}, True, loaded=loaded)
        ts.append(time.perf_counter() - t1)
    gbs = (sum(moved) - m0) / (time.perf_counter() - t0) / 1e9
    stop.set()
    for t in threads:
        t.join()
    return 1e3 * float(np.median(ts)), 1e3 * float(np.mean(ts)), gbs


if __name__ == "__main__":
    from pdb_eda_amd import synthetic
    tmp = tempfile.mkdtemp(prefix="pdbeda_contend_procs_")
    try:
        loaders = [synthetic.write_entry_files(tmp, "e%d" % j, 200, 100, j, as_paths=True) for j in range(4)]
        mp = multiprocessing.get_context("spawn")
        for spec in (sys.argv[1:] or ["1,0", "2,0", "4,0", "4,1"]):
            parts = [int(v) for v in spec.split(",")]
            n_proc, n_up, n_with = parts[0], parts[1], (parts[2] if len(parts) > 2 else parts[0])
            barrier = mp.Manager().Barrier(n_proc)
            with mp.Pool(n_proc) as pool:
                res = pool.map(work, [(k, loaders, n_up if k < n_with else 0, barrier, 2.0) for k in range(n_proc)])
            print("processes %d, uploader threads %d in %d of them: an analysis median %s ms; uploads %.1f GB/s in total" %
                  (n_proc, n_up, n_with, " ".join("%.2f" % r[0] for r in res), sum(r[2] for r in res)), flush=True)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
