"""Round 6: the HOST side of an 8-rank run on ONE box.

The driver has no 8-GPU node, so the entries/min curve of multiple-structure mode (BASELINE configs[3]) beyond one GPU is unmeasured; DESIGN
section 6 predicts that it bends between four and eight GPUs because every uploaded byte crosses host memory three times (page cache -> pinned
chunk by pread, pinned chunk -> PCIe by the copy engine's read).  What a rank does on the host needs no GPU: this script runs the REAL rank's
multiple-structure pool on GPU 0 while k = 0, 1, 3, 7 PHANTOM ranks -- each four processes x three reader threads, as a rank's pool has --
do the host half of the same work beside it, on the cores the other GPUs' ranks would get:
    pread of their own CCP4 files from the page cache into a chunk buffer (pass 1 + 2 over host memory: read the page, write the chunk),
    a copy of the chunk into a second buffer (pass 3: what the copy engine's read of the pinned chunk costs the memory system),
    and ~0.7 ms of interpreter work per entry (the per-entry Python of a worker).
It reports entries/min of the real rank against k, what the phantoms moved, and the cgroup's CPU throttling counters -- on a box whose CPU
share is a QUOTA (cpu.max) the phantoms compete with the real rank for cycles, not for memory bandwidth, and the table says so.

    python tools/exp/phantom_ranks.py [--ks 0,1,3,7] [--seconds 2.0] [--phantom-files 16]      (GPU box; writes gpurun_out/phantom_ranks.json)
"""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
CHUNK = 8 << 20


def cgroup_cpu():
    out = {}
    for name in ("cpu.max", "cpuset.cpus.effective"):
        try:
            out[name] = open("/sys/fs/cgroup/" + name).read().strip()
        except OSError:
            out[name] = None
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            if k in ("usage_usec", "nr_periods", "nr_throttled", "throttled_usec"):
                out[k] = int(v)
    except OSError:
        pass
    return out


def gpu_local_cpus():
    """The cores of GPU 0's NUMA node, asked for in a CHILD: the call initialises the HIP runtime, and this process goes on to spawn pools."""
    import ast
    import subprocess
    code = "import sys; sys.path.insert(0, %r); from pdb_eda_amd import _native; print(sorted(_native.device_local_cpus(0) or []))" % ROOT
    try:
        text = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600).stdout.strip().splitlines()[-1]
        return list(ast.literal_eval(text))
    except Exception:
        return []


def phantom_process(paths, cpus, stop, ready, moved, burn_ms):
    """One of a phantom rank's four 'workers': three reader threads over its share of the rank's files + the per-entry interpreter work."""
    import numpy as np
    if cpus:
        try:
            os.sched_setaffinity(0, cpus)
        except OSError:
            pass
    total = [0, 0, 0]

    def reader(t):
        a = np.empty(CHUNK, dtype=np.uint8)
        b = np.empty(CHUNK, dtype=np.uint8)
        k = t
        while not stop.is_set():
            path = paths[k % len(paths)]
            k += 3
            fd = os.open(path, os.O_RDONLY)
            try:
                off = 1024
                while not stop.is_set():
                    n = os.preadv(fd, [memoryview(a)], off)      # (releases the GIL)
                    if n <= 0:
                        break
                    np.copyto(b[:n], a[:n])                      # the "DMA read" of the chunk
                    off += n
                    total[t] += n
            finally:
                os.close(fd)
    threads = [threading.Thread(target=reader, args=(t,), daemon=True) for t in range(3)]
    for th in threads:
        th.start()
    ready.set()
    entries = 0
    while not stop.is_set():          # the per-entry Python of a worker: ~burn_ms of interpreter time an entry, GIL held
        if burn_ms <= 0:
            time.sleep(0.01)
            continue
        t0 = time.perf_counter()
        x = 0
        while time.perf_counter() - t0 < burn_ms * 1e-3:
            x += 1
        entries += 1
        time.sleep(0.0005)
    for th in threads:
        th.join(timeout=5)
    with moved.get_lock():
        moved.value += sum(total)


def stream_probe(seconds=0.5, threads=8):
    """What host memory delivers to `threads` numpy copies of 256 MiB each (read + write bytes / s)."""
    import numpy as np
    src = [np.ones(256 << 20, dtype=np.uint8) for _ in range(threads)]
    dst = [np.empty_like(s) for s in src]
    counts = [0] * threads
    stop = threading.Event()

    def run(t):
        while not stop.is_set():
            np.copyto(dst[t], src[t])
            counts[t] += 1
    ths = [threading.Thread(target=run, args=(t,)) for t in range(threads)]
    t0 = time.perf_counter()
    for th in ths:
        th.start()
    time.sleep(seconds)
    stop.set()
    for th in ths:
        th.join()
    dt = time.perf_counter() - t0
    return 2 * sum(counts) * (256 << 20) / dt / 1e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="0,1,3,7")
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--entries", type=int, default=125)
    ap.add_argument("--real-files", type=int, default=64, help="distinct file pairs of the real rank (64 x 64 MB = 4 GB)")
    ap.add_argument("--phantom-files", type=int, default=16, help="distinct map files per phantom rank (16 x 32 MB)")
    ap.add_argument("--burn-ms", type=float, default=0.7, help="interpreter work per phantom entry (0: none -- the phantoms are memory traffic only)")
    ap.add_argument("--phantom-procs", type=int, default=4, help="processes per phantom rank (a box with a CPU quota: fewer, so that the cgroup is not throttled)")
    ap.add_argument("--tag", default="", help="suffix of the output file")
    args = ap.parse_args()
    ks = [int(x) for x in args.ks.split(",")]
    ctx = mp.get_context("spawn")
    import numpy as np
    from pdb_eda_amd import synthetic, multipleStructures, densityAnalysis
    densityAnalysis.setGlobals(synthetic.synthetic_params())
    tmp = tempfile.mkdtemp(prefix="pdbeda_phantom_")
    out = {"cgroup_before": cgroup_cpu(), "logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)), "rows": []}
    try:
        gen = [synthetic.write_entry_files(tmp, "g%d" % k, 200, 100, 7000 + k, as_paths=True) for k in range(4)]
        loaders = list(gen)
        for k in range(4, args.real_files):
            src = gen[k % 4]
            paths = [os.path.join(tmp, "r%d%s.ccp4" % (k, s)) for s in ("", "_diff")]
            shutil.copyfile(src.density_path, paths[0]); shutil.copyfile(src.diff_path, paths[1])
            loaders.append(synthetic.SyntheticEntryFiles(paths[0], paths[1], src.n_residues, src.seed, src.edge, src.spacing, True))
        phantom_paths = []
        for r in range(max(ks)):
            mine = []
            for k in range(args.phantom_files):
                path = os.path.join(tmp, "p%d_%d.ccp4" % (r, k))
                shutil.copyfile(gen[k % 4].density_path, path)
                mine.append(path)
            phantom_paths.append(mine)
        entries = [multipleStructures.Entry("e%04d" % i, loaders[i % len(loaders)], cost_hint=0.0) for i in range(args.entries)]
        # the cores: the real rank's are the GPU's NUMA-local ones (as bench.py pins them); phantom r takes the next 16-core group
        all_cpus = sorted(os.sched_getaffinity(0))
        local = sorted(gpu_local_cpus() or all_cpus)
        # (an 8-GPU node hangs four GPUs off each socket: phantoms 0-2 share the real rank's socket -- the next 16-CPU groups of its NUMA node --,
        #  phantoms 3-6 take the other socket's)
        same = [c for c in local[16:] if c in set(all_cpus)]
        far = [c for c in all_cpus if c not in set(local)]
        groups = []
        for r in range(max(ks + [1])):
            pool_ = same if r < 3 else far
            q = r if r < 3 else r - 3
            grp = pool_[16 * q:16 * q + 16] or pool_[:16] or all_cpus
            groups.append(grp)
        out["phantom_cpus"] = [g[:2] + ["..."] + g[-1:] for g in groups]
        out["real_rank_cpus"] = local[:16]
        out["stream_probe_GBs_8_threads"] = stream_probe()
        for k in ks:
            stop, moved = ctx.Event(), ctx.Value("q", 0)
            procs, readies = [], []
            for r in range(k):
                for w in range(args.phantom_procs):
                    ready = ctx.Event()
                    p = ctx.Process(target=phantom_process, args=(phantom_paths[r][w::args.phantom_procs] or phantom_paths[r], groups[r], stop, ready, moved, args.burn_ms), daemon=True)
                    p.start()
                    procs.append(p); readies.append(ready)
            for ready in readies:
                ready.wait(60)
            row = {"phantom_ranks": k, "phantom_processes": len(procs)}
            stat0, t_ph = cgroup_cpu(), time.perf_counter()
            for mode, eager in (("both_maps", "1"), ("lazy_diff_map", "0")):
                os.environ["PDBEDA_EAGER_DIFF_MAP"] = eager
                pool = multipleStructures.ProcessPool(0, 4, params=synthetic.synthetic_params(), silent=True)
                try:
                    pool.warm()
                    pool.map(entries[:8])
                    t0 = time.perf_counter()
                    first = pool.map(entries)
                    dt = time.perf_counter() - t0
                    passes = max(1, int(1.25 * args.seconds / max(dt, 1e-3)))
                    t0 = time.perf_counter()
                    recs = pool.map(entries * passes)
                    dt = time.perf_counter() - t0
                    assert all(recs), "an entry failed"
                    row[mode + "_entries_per_min"] = 60.0 * len(recs) / dt
                finally:
                    pool.close()
            os.environ["PDBEDA_EAGER_DIFF_MAP"] = "0"
            t_ph = time.perf_counter() - t_ph
            stop.set()
            for p in procs:
                p.join(timeout=20)
            stat1 = cgroup_cpu()
            row["phantom_GBs_read_plus_copied"] = 2 * moved.value / t_ph / 1e9 if k else 0.0      # (bytes pread + bytes copied)
            row["host_memory_passes_GBs"] = 3 * moved.value / t_ph / 1e9 if k else 0.0            # (page read, chunk write + read, copy write: ~3 passes a byte, as a rank's upload)
            if "nr_throttled" in stat0 and "nr_throttled" in stat1:
                row["cgroup_periods_throttled"] = stat1["nr_throttled"] - stat0["nr_throttled"]
                row["cgroup_throttled_ms"] = (stat1["throttled_usec"] - stat0["throttled_usec"]) / 1e3
                row["cgroup_cpu_busy"] = (stat1["usage_usec"] - stat0["usage_usec"]) / 1e6 / t_ph
            out["rows"].append(row)
            print(json.dumps(row), flush=True)
        out["cgroup_after"] = cgroup_cpu()
        base = out["rows"][0]
        for row in out["rows"]:
            row["both_maps_vs_alone"] = row["both_maps_entries_per_min"] / base["both_maps_entries_per_min"]
            row["lazy_vs_alone"] = row["lazy_diff_map_entries_per_min"] / base["lazy_diff_map_entries_per_min"]
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        out["args"] = vars(args)
        with open(os.path.join(ROOT, "gpurun_out", "phantom_ranks%s.json" % args.tag), "w") as fh:
            json.dump(out, fh, indent=1)
        print(json.dumps({k: v for k, v in out.items() if k != "rows"}))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
