#!/usr/bin/env python3
"""bench.py -- whole-map blob labelling throughput on MI355X (BASELINE.json metric).

A *step* is one pass of the hot path over one synthetic entry that is already resident in
HBM: fused green/red blob labelling of a 256^3 map (BASELINE configs[1]) -- threshold at
+-(mean + 1.5 std), 26-neighbour connected components, per-blob fp64 statistics, blobs in
the reference's order, and the dense int32 label volume of both signs.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; every rank labels its own
   entries -- the path shards over entries with no data-path collective, scaling = weak --
   and RCCL carries only the final statistics reduction.)

Prints ONE JSON line on rank 0 (see README / DESIGN.md section "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (MI355X_MICROARCH.md): 8 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=256, help="grid edge (BASELINE config: 256)")
    ap.add_argument("--nsd", type=float, default=1.5, help="cutoff = mean + nsd * std")
    ap.add_argument("--streams", type=int, default=3, help="extra leg: entries in flight per GPU, one HIP stream + host thread each (multiple-structure mode); 1 = skip")
    ap.add_argument("--no-labels", action="store_true", help="skip the dense label volume (not the headline configuration)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-analysis", action="store_true", help="skip the informational densityAnalysis leg")
    ap.add_argument("--no-sigma3", action="store_true", help="skip the informational +-3 sigma leg (a kernel trace of the headline configuration must not mix cutoffs)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the bounded CPU-baseline sample")
    ap.add_argument("--windows", type=int, default=5, help="extra timed windows of --steps steps each (dispersion of ms_per_step)")
    ap.add_argument("--entries", type=int, default=125, help="multiple-structure leg (BASELINE configs[3]): entries per rank (1000 / 8 GPUs); 0 = skip")
    ap.add_argument("--entry-files", type=int, default=125, help="multiple-structure leg: distinct FILES on disk per rank (configs[3]: every entry of the list its own two files -- 125 x 64 MB = 8 GB, "
                                                                 "beyond any last-level cache; the page cache does not deduplicate copies)")
    ap.add_argument("--entry-generated", type=int, default=16, help="multiple-structure leg: distinct entries GENERATED per rank (0.3 s each); the other file names are byte copies of these, dealt round-robin")
    ap.add_argument("--entry-seconds", type=float, default=2.0, help="multiple-structure leg: repeat the entry list until the timed region is at least this long")
    ap.add_argument("--workers", type=int, default=4, help="multiple-structure leg: worker processes (= streams) per GPU")
    ap.add_argument("--entry-size", type=int, default=200, help="multiple-structure leg: grid edge of an entry (configs[3]: 200)")
    ap.add_argument("--entry-residues", type=int, default=100, help="multiple-structure leg: poly-ALA residues per entry (~500 atoms)")
    ap.add_argument("--sweep-entries", type=int, default=63, help="optimise-mode leg (BASELINE configs[4]: a 500-entry list = 63 per rank at 8 GPUs): resident entries per rank; 0 = skip")
    ap.add_argument("--sweep-iterations", type=int, default=3, help="optimise-mode leg: parameter tables evaluated (one changed radius each)")
    ap.add_argument("--sweep-seconds", type=float, default=1.0, help="optimise-mode leg: repeat the sweep over the tables until the timed region is at least this long")
    ap.add_argument("--no-beyond-cache", action="store_true", help="skip the informational leg on a working set larger than the Infinity Cache")
    ap.add_argument("--beyond-seconds", type=float, default=0.5, help="beyond-cache leg: timed region at least this long")
    ap.add_argument("--stream-seconds", type=float, default=0.5, help="multi-stream leg: entries are dealt to the streams until the timed region is at least this long")
    return ap.parse_args()


def csrc_sha16():
    """Hash of the kernel sources (the library's .hip / .h files; not the CPython helper hostwalk.c, which no kernel sees): ties a
    committed counter profile to the code it was collected on."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "pdb_eda_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".h")):
            continue
        with open(os.path.join(csrc, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(kernel, n_vox, n_planes):
    """Algorithmic HBM bytes of ONE launch of `kernel` ('Kernels and rooflines')."""
    words = (n_vox // 64) * n_planes
    table = {
        "k_tile_label": 4 * n_vox + (8 + 4) * words,      # read the f32 grid once; write bit masks + run bases
        "k_labels_tiles": 4 * n_vox + (8 + 4) * words,    # write ONE signed int32 label volume; read masks + run bases
        "k_face_merge": 0,                                 # run lists of the rows on tile faces: KB per tile
    }
    return table.get(kernel)


def analysis_leg(ctx, case="c2_bench_entry", reps=5):
    """Whole per-entry pipeline on a synthetic entry: parse + upload of both maps, aggregateCloud, atom and residue region
    discrepancies, green/red blob statistics (the record `pdb_eda multiple` keeps per entry).  The entry is one of the cases the
    REFERENCE was run on (tests/golden/analysis_big_<case>.npz): the leg checks its result against the reference's."""
    import io
    from pdb_eda_amd import ccp4, synthetic, structure, densityAnalysis as da
    ncrs, n_res, seed, spacing = synthetic.BIG_CASES[case]
    edge = ncrs[0]
    spec, header, st, params, dens, diff, rot = synthetic.cube_entry(ncrs, n_res, seed, spacing)
    da.setGlobals(params)
    files = synthetic.ccp4_bytes(spec, dens), synthetic.ccp4_bytes(spec, diff)
    pdbObj = structure.PDBEntry(structure.PDBHeader(pdbid="synth", resolution=2.0, spaceGroup="P_1", rotationMats=rot))

    def once():
        t = {}
        st.__dict__.pop("_pdbeda_columns", None)          # a fresh entry: the columnar snapshot of the structure is rebuilt inside the timed region
        t0 = time.perf_counter()
        densityObj = ccp4.parse(io.BytesIO(files[0]), "synth", ctx=ctx)
        diffObj = ccp4.parse(io.BytesIO(files[1]), "synth", ctx=ctx)
        da._attachCutoffs(densityObj, diffObj)
        an = da.DensityAnalysis("synth", densityObj, diffObj, st, pdbObj)
        t["parse_upload"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        an.aggregateCloud()
        t["aggregateCloud"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        an.calculateAtomRegionDiscrepancies(3.5, 3.0, "")
        an.calculateResidueRegionDiscrepancies(3.5, 3.0, "")
        t["region_discrepancies"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        an.calculateAtomSpecificBlobStatistics(an.greenBlobList + an.redBlobList)
        t["blob_statistics"] = time.perf_counter() - t0
        return an, t
    once()
    best = None
    for _ in range(reps):
        an, t = once()
        if best is None or sum(t.values()) < sum(best.values()):
            best = t
    total = sum(best.values())
    # the sphere gathers are latency / gather bound -- their primary figure is voxels tested per second
    # (sum over atoms of the (2R+2)^3 search box, cutils.pyx:241-243), from the HIP-event times of the region-sum batch
    atoms = list(st.get_atoms())
    box = 1
    for k in range(3):
        box *= 2 * int(round(3.5 / header.gridLength[k])) + 2
    ctx.profile_begin()
    an.calculateAtomRegionDiscrepancies(3.5, 3.0, "")
    prof = ctx.profile_end()
    dev_ms = sum(ms for _, ms in prof.values())
    sphere = {"atoms": len(atoms), "radius_A": 3.5, "voxels_tested": box * len(atoms), "device_ms": dev_ms,
              "voxels_tested_per_s": box * len(atoms) / (dev_ms * 1e-3) if dev_ms > 0 else None, "kernels_ms": {k: round(ms, 4) for k, (_, ms) in sorted(prof.items())}}
    # the reference's own result on this entry (seeds + numbers only are kept: tests/golden/make_golden_big.py)
    golden = np.load(os.path.join(ROOT, "tests", "golden", "analysis_big_%s.npz" % case), allow_pickle=False)
    want = float(golden["ratio"])
    rel = abs(an.densityElectronRatio - want) / abs(want)
    assert rel < 1e-8 and an.numVoxelsAggregated == int(golden["num_voxels"]), "analysis leg differs from the reference: %r vs %r" % (an.densityElectronRatio, want)
    ref_s = json.loads(str(golden["reference_seconds"]))
    return {"sphere_region_sums": sphere, "workload": "synthetic ~2 A entry: %d^3 grid at 0.5 A, %d atoms (%d with clouds), 2Fo-Fc + Fo-Fc maps parsed from CCP4 bytes" %
                        (edge, len(list(st.get_atoms())), an.cloudCounts[0]),
            "ms": {k: round(1e3 * v, 2) for k, v in best.items()}, "ms_per_entry": 1e3 * total, "entries_per_min": 60.0 / total,
            "density_electron_ratio": an.densityElectronRatio,
            "reference": {"density_electron_ratio": want, "relative_difference": rel, "num_voxels_aggregated_equal": True,
                          "seconds_one_core_build_container": ref_s, "total_s": round(sum(ref_s.values()), 1)},
            "note": "single host thread + one stream; checked here against the reference's result on the same entry (Cython path, one core, build container)"}


def cpu_share():
    """CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota when there is one (cpu.max "quota period": the GPU
    boxes of this pool give a 16-CPU quota under an affinity mask of all 256 logical CPUs -- a pool of 128 processes there delivers 16 CPUs' worth
    and spends the rest throttled; round 6)."""
    cores = len(os.sched_getaffinity(0))
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    return (max(1, min(cores, int(quota + 0.5))) if quota else cores), cores, quota


def host_contention():
    """The committed phantom-ranks runs (tools/exp/phantom_ranks.py, profiles/r*_phantom_ranks_*.json): the real rank's entries/min beside k phantom ranks
    that do the host half of a rank's work -- and the cgroup's throttling beside it.  Not re-measured here (two minutes of pools); carried from profiles/."""
    import glob
    rows, quota = [], None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_phantom_ranks_*.json"))):
        try:
            with open(path) as fh:
                d = json.load(fh)
        except (OSError, ValueError):
            continue
        quota = d.get("cgroup_before", {}).get("cpu.max", quota)
        for r in d.get("rows", []):
            rows.append({"file": os.path.basename(path), "phantom_ranks": r.get("phantom_ranks"), "phantom_processes": r.get("phantom_processes"),
                         "both_maps_k_per_min": round(r.get("both_maps_entries_per_min", 0.0) / 1e3, 1), "lazy_k_per_min": round(r.get("lazy_diff_map_entries_per_min", 0.0) / 1e3, 1),
                         "phantom_memory_passes_GBs": round(r.get("host_memory_passes_GBs", 0.0)), "cgroup_cpu_busy": round(r.get("cgroup_cpu_busy", 0.0), 1),
                         "cgroup_periods_throttled": r.get("cgroup_periods_throttled")})
    if not rows:
        return None
    return {"rows": rows, "cgroup_cpu_max": quota,
            "note": "committed runs of tools/exp/phantom_ranks.py on one box of this pool: a 16-CPU cgroup quota (cpu.max) that one rank's pool half fills -- every row with a phantom is "
                    "throttled, so the real rank's loss measures the quota, not host-memory contention; the 8-rank host side needs an 8-GPU node (DESIGN.md section 6)"}


def link_rate(torch):
    """Pinned host -> HBM copy of 64 MiB, the best of six batches of five copies: GB/s."""
    pin = torch.empty(16 << 20, dtype=torch.float32).pin_memory()
    dev = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    dev.copy_(pin, non_blocking=True)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(6):      # (the best of six batches: the pools' worker processes may still be winding down beside the first ones)
        t1 = time.perf_counter()
        for _ in range(5):
            dev.copy_(pin, non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, 5 * pin.numel() * 4 / (time.perf_counter() - t1) / 1e9)
    del pin, dev
    return best


def multiple_leg(args, pool, rank, local_rank, world, barrier, dist, torch):
    import shutil
    import tempfile
    from pdb_eda_amd import synthetic, multipleStructures, densityAnalysis
    densityAnalysis.setGlobals(synthetic.synthetic_params())
    link_before = link_rate(torch)
    tmp = tempfile.mkdtemp(prefix="pdbeda_bench_%d_" % rank)
    try:
        t0 = time.perf_counter()
        # files on disk: `distinct` names (BASELINE configs[3]: every entry of a rank's list its own two CCP4 files), of which `generated` are
        # distinct synthetic entries (0.3 s each) and the others byte copies of those under their own names -- own inodes, own pages in the
        # page cache: a pass over the list reads 125 x 64 MB = 8 GB, beyond any last-level cache (16 files = 1 GB were re-read 13 x)
        distinct = max(1, min(args.entry_files, args.entries))
        generated = max(1, min(args.entry_generated, distinct))
        try:      # (N ranks write N x 8 GB into one /tmp: a rank takes at most its share of two thirds of the free space)
            pair_bytes = 2 * (4 * args.entry_size ** 3 + 1024)
            room = int(shutil.disk_usage(tmp).free * 2 / 3 / max(world, 1) / pair_bytes)
            if room < distinct:
                print("bench.py: /tmp holds %d of the %d file pairs per rank: fewer distinct names" % (max(room, generated), distinct), file=sys.stderr)
                distinct = max(generated, min(distinct, room))
        except OSError:
            pass
        # entry 0 of every rank is the configs[3] entry the REFERENCE was run on (synthetic.BIG_CASES["c3_multiple_entry"]: 200^3, 100
        # residues, seed 0): its records are checked against the reference's numbers after every pool has returned
        golden_case = synthetic.BIG_CASES["c3_multiple_entry"]
        golden_ok = (args.entry_size, args.entry_residues) == (golden_case[0][0], golden_case[1])
        loaders = [synthetic.write_entry_files(tmp, "e%d" % k, args.entry_size, args.entry_residues, (golden_case[2] if k == 0 else 1000 * rank + k), as_paths=True)
                   for k in range(generated)]
        copy_error = None
        for k in range(generated, distinct):       # name k = a copy of generated entry k % generated (same model, same record)
            src = loaders[k % generated]
            try:
                paths = [os.path.join(tmp, "e%d%s.ccp4" % (k, suffix)) for suffix in ("", "_diff")]
                for a, b in zip((src.density_path, src.diff_path), paths):
                    shutil.copyfile(a, b)
                loaders.append(synthetic.SyntheticEntryFiles(paths[0], paths[1], src.n_residues, src.seed, src.edge, src.spacing, True))
            except OSError as error:               # (a small /tmp: the leg runs on the names it has, and says so)
                copy_error = "%s: %s" % (type(error).__name__, error)
                for path in paths:
                    if os.path.exists(path):
                        os.unlink(path)
                break
        distinct = len(loaders)
        gen_s = time.perf_counter() - t0
        golden = np.load(os.path.join(ROOT, "tests", "golden", "analysis_big_c3_multiple_entry.npz"), allow_pickle=False) if golden_ok else None
        checked = {"records": 0}

        def check_golden(records):
            # file -> upload engine -> pool at 200^3: a wrong-but-non-zero record must not pass the bench
            # (records of the entry list, possibly repeated: entry k of the list reads files k % distinct, and files 0 are the golden entry's)
            if golden is None:
                return
            for i, r in enumerate(records):
                if ((i % args.entries) % distinct) % generated != 0:      # (names 0, generated, 2 * generated, ... hold the golden entry's bytes)
                    continue
                assert r, "the golden entry failed in the pool"
                got = r["stats"]["density_electron_ratio"]
                assert abs(got - float(golden["ratio"])) <= 1e-8 * abs(float(golden["ratio"])), "multiple_structures: golden entry ratio %r vs the reference's %r" % (got, float(golden["ratio"]))
                assert r["stats"]["num_voxels_aggregated"] == int(golden["num_voxels"]), "multiple_structures: golden entry voxel count differs from the reference's"
                checked["records"] += 1
        entries = [multipleStructures.Entry("r%de%04d" % (rank, i), loaders[i % distinct], cost_hint=0.0) for i in range(args.entries)]
        pool.warm()
        pool.map(entries[:2 * args.workers])                          # untimed: first-use costs of every worker (imports, arenas)
        barrier()
        # The timed region is ONE map over the entry list repeated `passes` times -- as many as a first, calibrating pass says it
        # takes to run for --entry-seconds (a 0.2 s region says little).  (Rounds 3-4 timed the passes one by one, each ending in a
        # barrier: a pass of 125 entries is eight tasks, every one with a pipeline to fill, and a drain at its end -- 4-6 % of the
        # region was the leg's own stop-and-go, which a run over a real list does not have.)
        def passes_for(pool_):
            t0 = time.perf_counter()
            first = pool_.map(entries)
            barrier()
            dt = time.perf_counter() - t0
            check_golden(first)
            want = max(1, min(64, int(1.25 * args.entry_seconds / max(dt, 1e-3)) + 1))      # (the one-call form is faster than the calibrating pass)
            if dist is not None:     # (every rank maps the same number of entries)
                t = torch.tensor([want], dtype=torch.int64, device="cuda")
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                want = int(t.item())
            return want
        passes = passes_for(pool)
        t0 = time.perf_counter()
        records = pool.map(entries * passes)
        barrier()
        elapsed = time.perf_counter() - t0
        ok = sum(1 for r in records if r)
        check_golden(records)
        n_done = passes * args.entries
        own_rate = 60.0 * n_done / elapsed
        # ONE cold-cache data point: the pages of the bench's own files are dropped (fsync + POSIX_FADV_DONTNEED: an ordinary user may
        # do that for files it owns; whether the kernel obeys is checked by timing) and one pass over the distinct entries is timed
        cold, dropped, drop_error = None, 0, None
        try:                                     # (rank-local: no barrier in here -- a rank that leaves through the except must not strand the others)
            for l in loaders:
                for path in (l.density_path, l.diff_path):
                    fd = os.open(path, os.O_RDONLY)
                    try:
                        os.fsync(fd)
                        os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
                        dropped += 1
                    finally:
                        os.close(fd)
        except (OSError, AttributeError) as error:
            drop_error = "%s: %s" % (type(error).__name__, error)
        barrier()
        t0 = time.perf_counter()
        cold_records = pool.map(entries[:distinct])      # (every file of the rank once: 8 GB at the default sizes)
        barrier()
        cold_s = time.perf_counter() - t0
        check_golden(cold_records)
        cold = {"entries": distinct, "seconds": cold_s, "entries_per_min": 60.0 * distinct / cold_s, "files_dropped": dropped,
                "file_GBs": distinct * 2 * 4 * args.entry_size ** 3 / cold_s / 1e9,
                "note": "one pass over the distinct entries right after fsync + posix_fadvise(DONTNEED) on their files; a rate near the warm one means the "
                        "kernel kept the pages (tmpfs / a busy page cache) -- read it beside file_GBs"}
        if drop_error:
            cold["drop_error"] = drop_error
        # ONE worker process of the same kind for comparison (the pool is closed first: few processes may share the GPU)
        pool.close()
        sample = entries[:min(16, len(entries))]
        os.environ["PDBEDA_EAGER_DIFF_MAP"] = "1"       # (the same loader as the pool it is compared with)
        one = multipleStructures.ProcessPool(local_rank, 1, params=synthetic.synthetic_params(), silent=True)
        os.environ["PDBEDA_EAGER_DIFF_MAP"] = "0"
        try:
            one.warm()
            one.map(sample[:2])
            single = None
            for _ in range(3):                      # best of three short passes: a pool that has just started finds the GPU at idle clocks
                t1 = time.perf_counter()
                one.map(sample, chunk=16)           # (the chunk size the big pool works with: the worker pipelines inside a chunk)
                dt = (time.perf_counter() - t1) / len(sample)
                single = dt if single is None else min(single, dt)
        finally:
            one.close()
        # the product's default: the Fo-Fc grid of an entry is uploaded only when something reads it -- nothing in this record does
        lazy = multipleStructures.ProcessPool(local_rank, args.workers, params=synthetic.synthetic_params(), silent=True)
        try:
            lazy.warm()
            lazy.map(entries[:2 * args.workers])
            barrier()
            lazy_passes = passes_for(lazy)
            t1 = time.perf_counter()
            lazy_records = lazy.map(entries * lazy_passes)
            barrier()
            lazy_elapsed = time.perf_counter() - t1
            lazy_ok = sum(1 for r in lazy_records if r)
            check_golden(lazy_records)
        finally:
            lazy.close()
        lazy_done = lazy_passes * args.entries
        # ONE load by itself: a 32 MB map file -> HBM through the upload engine, statistics included -- in a lone pool worker (a fresh process;
        # this one holds a dozen streams by now, and streams that exist slow a process's copy streams)
        load_single = None
        try:
            lone = multipleStructures.ProcessPool(local_rank, 1, params=synthetic.synthetic_params(), silent=True)
            try:
                lone.warm()
                times = lone.run(synthetic.time_single_loads, [([l.density_path for l in loaders[:4]], 48)])[0][8:]
            finally:
                lone.close()
            n_bytes = 4 * args.entry_size ** 3
            load_single = {"bytes": n_bytes, "loads": len(times), "median_ms": 1e3 * float(np.median(times)), "GBs_median": n_bytes / float(np.median(times)) / 1e9,
                           "GBs_best": n_bytes / min(times) / 1e9,
                           "note": "pdbeda_map_upload_file_stats of one map alone in a lone pool worker: open, chunked pread -> pinned -> HBM on three reader threads, "
                                   "mean / std, one wait"}
        except Exception as error:
            load_single = {"error": "%s: %s" % (type(error).__name__, error)}
        per_rank = [own_rate]
        total_done = n_done
        if dist is not None:
            t = torch.tensor([elapsed, single, lazy_elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, single, lazy_elapsed = float(t[0].item()), float(t[1].item()), float(t[2].item())
            c = torch.tensor([ok, n_done, lazy_ok, lazy_done], dtype=torch.int64, device="cuda")
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            ok, total_done, lazy_ok, lazy_done = int(c[0].item()), int(c[1].item()), int(c[2].item()), int(c[3].item())
            rates = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
            dist.all_gather(rates, torch.tensor([own_rate], dtype=torch.float64, device="cuda"))
            per_rank = [float(r.item()) for r in rates]
        n_atoms = len(list(loaders[0].structure()[0].get_atoms()))
        file_mb = 2 * 4 * args.entry_size ** 3 / 1e6
        # the leg's roofline is the host link: an entry's map bytes must cross it, whatever else happens.  The link's rate is
        # measured in this run (link_rate: a pinned 64 MiB buffer, the best of six batches of five copies), so the fraction compares like with like.
        h2d_gbs = max(link_before, link_rate(torch))      # (measured before the pools started and after they closed: the better of the two -- one run of
                                                          #  round 5 measured 49 GB/s behind the pools where every other run measures 57)
        per_gpu_rate = total_done / world / elapsed                     # entries / s / GPU
        lazy_rate = lazy_done / world / lazy_elapsed
        pcie = {"bound": "pcie", "unit": "GB/s", "peak": h2d_gbs, "peak_source": "pinned host -> HBM copy of 64 MiB measured in this run (PCIe Gen5 x16: 63 GB/s spec)",
                "both_maps": {"bytes_per_entry": file_mb * 1e6, "achieved": file_mb * 1e6 * per_gpu_rate / 1e9, "frac": file_mb * 1e6 * per_gpu_rate / 1e9 / h2d_gbs},
                "lazy_diff_map": {"bytes_per_entry": file_mb * 1e6 / 2, "achieved": file_mb * 1e6 / 2 * lazy_rate / 1e9, "frac": file_mb * 1e6 / 2 * lazy_rate / 1e9 / h2d_gbs}}
        # the same entries on the host cores of this box (rank 0, N = 1 only): the CPU restatement ("port": the product's host code over
        # oracle/pdbeda_oracle.c) in a multiprocessing.Pool on ALL cores -- the reference's own shape (multipleStructures.py:167-168)
        cpu_pool = None
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_entry
            cores, affinity, quota = cpu_share()
            tasks = [(l.density_path, l.n_residues, l.seed, l.edge, l.spacing) for l in loaders]
            tasks = [tasks[i % len(tasks)] for i in range(max(len(tasks), 2 * cores))]
            cpu_pool = cpu_entry.multiple_baseline(tasks, cores, seconds=args.cpu_seconds)
            cpu_pool.update({"cores": cores, "affinity_cpus": affinity, "cgroup_cpu_quota": quota, "kind": "port", "unit": "entries/min", "value": cpu_pool["entries_per_min"],
                             "sample": "%d x %d^3 entries (%d atoms) per pass, repeated for %.0f s: read the 2Fo-Fc CCP4 file, numpy-tree mean / std, aggregateCloud "
                                       "(flattening + oracle composite + statistics tail), the entry's diffs -- multiprocessing.Pool(%d), one entry per task"
                                       % (len(tasks), args.entry_size, n_atoms, cpu_pool["seconds"], cores)})
        return {"workload": "configs[3]: %d entries per rank, the list mapped `passes` times over in ONE call (%d distinct files-pairs on disk = %.0f MB of CCP4 files per rank; %d generated entries, "
                            "the other names byte copies of them), each two CCP4 files of a %d^3 grid + a "
                            "%d-atom model: read, parse, upload (BOTH maps: PDBEDA_EAGER_DIFF_MAP=1), aggregateCloud + the per-entry record of `pdb_eda multiple`"
                            % (args.entries, distinct, distinct * file_mb, generated, args.entry_size, n_atoms),
                "distinct_files": distinct, "generated_entries": generated, "file_copy_error": copy_error,
                "entries": total_done, "entries_ok": ok, "passes": passes, "workers_per_gpu": args.workers, "seconds": elapsed,
                "entries_per_min": 60.0 * total_done / elapsed, "entries_per_min_per_gpu": 60.0 * total_done / world / elapsed,
                "entries_per_min_per_rank": [round(v, 1) for v in per_rank],
                "one_worker_ms_per_entry": 1e3 * single, "one_worker_entries_per_min": 60.0 / single,
                "pool_vs_one_worker": (total_done / world / elapsed) * single, "generation_s": gen_s,
                "roofline": pcie, "load_GBs_single": (load_single or {}).get("GBs_median"), "load_single": load_single,
                "golden_records_checked": checked["records"] if golden is not None else None,
                "lazy_diff_map": {"entries": lazy_done, "entries_ok": lazy_ok, "seconds": lazy_elapsed, "entries_per_min": 60.0 * lazy_done / lazy_elapsed,
                                  "note": "the same entry list with the product's default loader: the Fo-Fc file's header is read, its grid would follow on first use "
                                          "and nothing in the record of `pdb_eda multiple` uses it (32 MB per entry over PCIe instead of 64); same records"},
                "cpu_baseline": cpu_pool,
                "host_contention": host_contention() if rank == 0 else None,
                "cold_pass": cold,
                "page_cache": "warm for `entries_per_min`: the files were written by this process moments earlier and every one is read again on each pass; "
                              "`cold_pass` is one pass after the files' pages were dropped",
                "note": "worker processes (spawn), one HIP stream each, sharing the GPU of the rank; sharding over ranks is one entry list per rank, no collective"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sweep_leg(args, rank, local_rank, world, barrier, dist, torch):
    """BASELINE configs[4]: one iteration of the radius optimisation = every resident entry re-analysed under a candidate
    parameter table + the statistics reduction (all-gather of rows, all-reduce of counters: RCCL when world > 1)."""
    import shutil
    import tempfile
    from pdb_eda_amd import synthetic, multipleStructures, optimizeSweep
    tmp = tempfile.mkdtemp(prefix="pdbeda_sweep_%d_" % rank)
    sw = None
    try:
        distinct = 4
        loaders = [synthetic.write_entry_files(tmp, "s%d" % k, args.entry_size, args.entry_residues, 5000 + 1000 * rank + k, as_paths=True) for k in range(distinct)]
        entries = [multipleStructures.Entry("r%ds%04d" % (rank, i), loaders[i % distinct], cost_hint=0.0) for i in range(args.sweep_entries)]
        t0 = time.perf_counter()
        sw = optimizeSweep.ProcessSweep(entries, local_rank, args.workers)
        load_s = time.perf_counter() - t0
        sets = synthetic.sweep_param_sets()
        sets = [sets[k % len(sets)] for k in range(1, args.sweep_iterations + 1)]
        sw.iteration(sets[0])                                   # untimed: first-use costs of the workers
        barrier()
        # the timed region is whole sweeps over the parameter tables, repeated until it has run for at least --sweep-seconds
        # (three iterations over 32 entries are 35 ms: not a measurement)
        elapsed, n_iter, reduce_s = 0.0, 0, 0.0
        while True:
            t0 = time.perf_counter()
            for params in sets:
                reduction, records = sw.iteration(params)
            barrier()
            elapsed += time.perf_counter() - t0
            n_iter += len(sets)
            if not keep_going(elapsed < args.sweep_seconds and n_iter < 4096, dist, torch):   # (every rank runs the same passes)
                break
        ok = sum(1 for r in records if r)
        # the reduction by itself: with a group (N > 1) here, on the job's own ranks; at N = 1 through a 1-rank RCCL group in a child
        from pdb_eda_amd import optimizeStats
        rccl = None
        if dist is not None:
            barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                optimizeStats.calculateMedianDiffsSlopes(records, sets[-1])
            barrier()
            reduce_s = (time.perf_counter() - t0) / 20
            t = torch.tensor([elapsed, reduce_s], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, reduce_s = float(t[0].item()), float(t[1].item())
            rccl = {"reduction_ms": 1e3 * reduce_s, "ranks": world, "records_per_rank": len(records), "calls": 20}
        else:
            sw.close()                           # (few processes may share the GPU: the workers leave before the child comes)
            sw = None
            rccl = reduction_through_rccl(records, sets[-1], local_rank, tmp)
            rccl["ranks"] = 1
        per = elapsed / (n_iter * max(1, args.sweep_entries))
        return {"workload": "configs[4]: %d resident entries per rank (%d^3 maps, ~%d atoms), %d parameter tables of a radius sweep: per table every entry is "
                            "re-analysed (aggregateCloud -> diffs / slopes / overlap counters) and the records are reduced over all ranks" %
                            (args.sweep_entries, args.entry_size, 5 * args.entry_residues, len(sets)),
                "entries": args.sweep_entries * world, "iterations": n_iter, "parameter_tables": len(sets), "workers_per_gpu": args.workers, "seconds": elapsed,
                "ms_per_entry_iteration_per_gpu": 1e3 * per, "entry_iterations_per_s": world / per, "entries_ok_last_iteration": ok,
                "reduced_types": len(reduction[0]), "load_s": load_s,
                "reduction": "optimizeStats: all_gather of per-entry rows + all_reduce of counters (%s)" %
                             ("RCCL" if dist is not None else "inside the timed iterations: single process, no group; timed by itself through a 1-rank RCCL group in a fresh child process"),
                "reduction_ms": rccl.get("reduction_ms") if rccl else None, "reduction_rccl": rccl,
                "note": "worker processes keep their lane of entries resident in HBM between iterations; load_s includes spawning them"}
    finally:
        if sw is not None:
            sw.close()
        shutil.rmtree(tmp, ignore_errors=True)


def keep_going(local_wish, dist, torch):
    """Whether a leg runs another pass, decided by ALL ranks together: a pass ends in a barrier (and, in the sweep, in the
    statistics reduction), so ranks that decided on their own clocks could run different numbers of passes -- one would wait in a
    collective the other never enters.  Everybody continues while anybody wants to."""
    if dist is None:
        return bool(local_wish)
    t = torch.tensor([1 if local_wish else 0], dtype=torch.int64, device="cuda")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))


REDUCTION_CHILD = r'''
import json, os, pickle, sys, time
sys.path.insert(0, %(root)r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%(port)d")
import torch, torch.distributed as dist
# the process group comes first: nothing in this process has touched the GPU yet
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", %(device)d))
torch.cuda.set_device(%(device)d)
from pdb_eda_amd import optimizeStats
records, params = pickle.load(open(%(inp)r, "rb"))
for _ in range(5):
    through = optimizeStats.calculateMedianDiffsSlopes(records, params)
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n):
    through = optimizeStats.calculateMedianDiffsSlopes(records, params)
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0) / n
dist.barrier()
dist.destroy_process_group()
t0 = time.perf_counter()
for _ in range(n):
    plain = optimizeStats.calculateMedianDiffsSlopes(records, params)      # no group: the single-process formula
plain_ms = 1e3 * (time.perf_counter() - t0) / n
same = [through[0], through[2], through[4], through[5]] == [plain[0], plain[2], plain[4], plain[5]]
json.dump({"reduction_ms": ms, "plain_ms": plain_ms, "equal_to_plain": bool(same), "records": len(records), "calls": n}, open(%(out)r, "w"))
'''


def reduction_through_rccl(records, params, device, tmp):
    """The path's one collective (optimizeParams.py:400-406 -> optimizeStats) in a record even at one GPU: a FRESH child process
    starts a 1-rank `nccl` (= RCCL) group before it touches the GPU -- a process group, once initialised, always goes through
    all_gather / all_reduce on device tensors, the same code at every world size -- and times the reduction of this rank's last
    iteration's records.  (A child, not this process: the group must exist before the first GPU call.)"""
    import pickle
    import subprocess
    inp, out, script = os.path.join(tmp, "reduce_in.pkl"), os.path.join(tmp, "reduce_out.json"), os.path.join(tmp, "reduce_child.py")
    with open(inp, "wb") as fh:
        pickle.dump((records, params), fh)
    with open(script, "w") as fh:
        fh.write(REDUCTION_CHILD % {"root": ROOT, "port": 29500 + os.getpid() % 2000, "device": device, "inp": inp, "out": out})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "TORCHELASTIC_RUN_ID", "GROUP_RANK", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    proc = subprocess.run([sys.executable, script], env=env, capture_output=True, text=True, timeout=240)
    if proc.returncode != 0:
        return {"error": proc.stderr[-400:]}
    with open(out) as fh:
        return json.load(fh)


def guarded(name, leg):
    """An informational leg must not take the headline line down with it: its failure is reported in its place."""
    try:
        return leg()
    except BaseException as exception:
        if isinstance(exception, (KeyboardInterrupt, SystemExit)):
            raise
        print("bench.py: leg %s failed: %s: %s" % (name, type(exception).__name__, exception), file=sys.stderr)
        return {"error": "%s: %s" % (type(exception).__name__, exception)}


def _import_native():
    from pdb_eda_amd import _native
    return _native


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
    as a CHILD process (this process has not touched the GPU and never will), relay its output and return its exit code.
    Fewer than N visible devices is an error, not a one-GPU measurement."""
    import socket
    import subprocess
    import torch                                   # (import only; device_count() does not initialise the GPU on this image)
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus:
        print("bench.py: --gpus %d but only %d device(s) are visible: refusing to measure fewer GPUs than asked for" % (args.gpus, n_dev), file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))

    # torch is imported first (importing does not touch the GPU): it bundles its own HIP runtime, and the library must resolve
    # libamdhip64 to THAT copy in this process -- two runtimes in one process do not see the device.
    import torch
    import torch.distributed as dist
    # Build (rank 0 of the node compiles if anything is stale; the others wait for the file) BEFORE this process touches the GPU:
    # hipcc / make must not be spawned from a process that already holds the device.
    import __graft_entry__ as entry
    if local_rank == 0:
        entry.build()
    else:
        csrc = os.path.join(ROOT, "pdb_eda_amd", "csrc")
        newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
        deadline = time.time() + 600
        while (not os.path.exists(entry.HIP_SO) or os.path.getmtime(entry.HIP_SO) < newest) and time.time() < deadline:
            time.sleep(0.5)
        entry.build()
    from pdb_eda_amd import _native, ccp4, synthetic, multipleStructures

    # multiple-structure leg: its worker processes are spawned now, before the GPU is initialised here
    pool = None
    if args.entries > 0:
        # the leg's figure is quoted with BOTH maps of an entry uploaded, as the reference's loader reads both (its workers inherit
        # the switch); the product's default -- the Fo-Fc grid follows only if the analysis asks for it -- is timed beside it
        os.environ["PDBEDA_EAGER_DIFF_MAP"] = "1"
        pool = multipleStructures.ProcessPool(local_rank, args.workers, params=synthetic.synthetic_params(), silent=True)
        os.environ["PDBEDA_EAGER_DIFF_MAP"] = "0"

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    pinned_cpus = _native.pin_to_device(local_rank)      # host side of this rank on the GPU's NUMA node (the workers above do the same)
    rccl_ranks = 1
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        rccl_ranks = dist.get_world_size()

    # ---- synthetic entry: smooth noise, orthogonal cell, resident in HBM ----
    n = args.size
    spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
    grid = synthetic.smooth_noise((n, n, n), seed=7 + rank, sigma_voxels=1.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    # torch owns the device buffer (plumbing); the library borrows the pointer (zero copy)
    dens = torch.from_numpy(grid).to("cuda:%d" % local_rank)
    ctx = _native.Context(local_rank)
    dmap = _native.DeviceMap(ctx, dens, header.geometry(), device_ptr=dens.data_ptr())
    mean, std = dmap.stats()
    cut = mean + args.nsd * std
    n_vox = n * n * n
    labels = not args.no_labels

    def step():
        g, r = dmap.full_blobs_pm(cut, -cut, labels=labels)
        return g, r

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    keep = None
    for _ in range(args.warmup):
        keep = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        keep = step()           # dropping the previous lists recycles their device arena (stream ordered)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    windows = []                      # dispersion: further windows of K steps each (never `value`)
    for _ in range(max(0, args.windows)):
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            keep = step()
        barrier()
        windows.append(1e3 * (time.perf_counter() - t1) / args.steps)
    green, red = keep
    n_green, n_red = len(green), len(red)
    sig_vox = int(green.stats()["n"].sum() + red.stats()["n"].sum())

    # ---- the same map at pdb_eda's own green / red default, +-(mean + 3 std) (densityAnalysis.py:148) -- informational, never `value`
    sigma3 = None
    if args.nsd != 3.0 and not args.no_sigma3:
        cut3 = mean + 3.0 * std
        for _ in range(max(args.warmup, 2)):
            keep3 = dmap.full_blobs_pm(cut3, -cut3, labels=labels)
        barrier()
        n3 = max(args.steps, 50)
        t1 = time.perf_counter()
        for _ in range(n3):
            keep3 = dmap.full_blobs_pm(cut3, -cut3, labels=labels)
        barrier()
        el3 = time.perf_counter() - t1
        ctx.profile_begin()
        for _ in range(n3):
            keep3 = dmap.full_blobs_pm(cut3, -cut3, labels=labels)
        prof3 = ctx.profile_end()
        ev3 = 1e3 * sum(ms for _, ms in prof3.values()) / n3
        gap3 = max(0.0, (ev3 - 1e6 * el3 / n3) / (sum(c for c, _ in prof3.values()) / n3))
        sigma3 = {"cutoff_sigma": 3.0, "steps": n3, "ms_per_step": 1e3 * el3 / n3, "value": n_vox * n3 / el3 / 1e6, "unit": "Mvoxels/s",
                  "blobs": {"green": len(keep3[0]), "red": len(keep3[1])},
                  "kernels_us": {k: round(max(1e3 * ms / c - gap3, 0.0), 2) for k, (c, ms) in sorted(prof3.items())},
                  "note": "pdb_eda's default green / red cutoff on the same resident map; parity at this cutoff: tests/test_gpu_voxel.py (nsd 3.0)"}
        del keep3

    # final statistics reduction (the only collective of the path; KB-scale, outside the voxel work)
    totals = torch.tensor([n_green, n_red, sig_vox], dtype=torch.int64, device="cuda")
    if world > 1:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)

    # ---- multiple-structure mode on one GPU (informational, never `value`): S entries in flight, each on its own context
    # (HIP stream + arena pool) driven by its own host thread (ctypes releases the GIL); K steps dealt round-robin ----
    multi = None
    if args.streams > 1:
        import threading
        lanes = []
        for k in range(args.streams):
            c = ctx if k == 0 else _native.Context(local_rank)
            gk = grid if k == 0 else synthetic.smooth_noise((n, n, n), seed=1000 * (k + 1) + rank, sigma_voxels=1.5)
            tk = dens if k == 0 else torch.from_numpy(gk).to("cuda:%d" % local_rank)
            mk = dmap if k == 0 else _native.DeviceMap(c, tk, header.geometry(), device_ptr=tk.data_ptr())
            mu, sd = mk.stats()
            lanes.append({"ctx": c, "map": mk, "tensor": tk, "cut": mu + args.nsd * sd, "keep": None})

        # one host thread per stream for the life of the leg: started and parked on a barrier BEFORE the clock starts (thread
        # start-up inside a 4 ms region made the driver's figure at --steps 20 contradict the design notes), then every thread
        # labels its share of the entries and drains its stream
        gate = threading.Barrier(args.streams + 1, timeout=600)
        plan = {"counts": [0] * args.streams, "stop": False, "error": None}

        def lane_thread(k):
            lane = lanes[k]
            try:
                while True:
                    gate.wait()                      # start of a round
                    if plan["stop"]:
                        return
                    for _ in range(plan["counts"][k]):
                        lane["keep"] = lane["map"].full_blobs_pm(lane["cut"], -lane["cut"], labels=labels)
                    lane["ctx"].synchronize()
                    gate.wait()                      # end of the round
            except threading.BrokenBarrierError:
                return
            except BaseException as exception:       # a lane that fails must not leave the others parked on the barrier for ever
                plan["error"] = exception
                gate.abort()

        threads = [threading.Thread(target=lane_thread, args=(k,), daemon=True) for k in range(args.streams)]
        for t in threads:
            t.start()

        def run_all(total):
            plan["counts"] = [total // args.streams + (1 if k < total % args.streams else 0) for k in range(args.streams)]
            try:
                gate.wait()
                t_start = time.perf_counter()
                gate.wait()
            except threading.BrokenBarrierError:
                raise RuntimeError("multi-stream leg: a lane failed: %r" % (plan["error"],))
            return time.perf_counter() - t_start
        run_all(max(args.warmup, 2) * args.streams)
        # entries of the timed round: what a single stream would need --stream-seconds for, so the region is at least that long
        n_multi = max(2 * args.steps, int(args.stream_seconds / max(0.75 * elapsed / args.steps, 1e-6)) + 1)
        barrier()
        el = run_all(n_multi)
        barrier()
        plan["stop"] = True
        gate.wait()
        for t in threads:
            t.join()
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        multi = {"streams_per_gpu": args.streams, "entries": n_multi, "seconds": el, "value": world * n_vox * n_multi / el / 1e6, "unit": "Mvoxels/s",
                 "ms_per_entry": 1e3 * el / n_multi,
                 "note": "different 256^3 entries resident in HBM, one per stream, host thread per stream (parked before the clock starts); the latency-bound merge kernels of one entry overlap the tile kernel of another"}
        for lane in lanes[1:]:
            lane["keep"] = None
        lanes = lanes[:1]

    # ---- per-kernel durations with HIP events on the library's stream (separate K-step pass) ----
    ctx.profile_begin()
    for _ in range(args.steps):
        keep = step()
    prof = ctx.profile_end()
    # ONE clock for the roofline: HIP events on the launch stream.  An event pair around a launch also times the launch gap
    # (~2 us): the gap is calibrated in this same run as (sum of the event times of a step - host-timed step) / launches and
    # taken off every kernel, so the corrected kernel times add up to ms_per_step by construction (the rocprofv3 averages
    # under profiles/ are the cross-check).
    n_launch = sum(c for c, _ in prof.values()) / args.steps
    step_events_us = 1e3 * sum(ms for _, ms in prof.values()) / args.steps
    step_wall_us = 1e6 * elapsed / args.steps
    gap_us = max(0.0, (step_events_us - step_wall_us) / n_launch)
    per_kernel = {k: {"calls": c, "event_us": 1e3 * ms / c, "avg_us": max(1e3 * ms / c - gap_us, 0.0)} for k, (c, ms) in prof.items()}
    dominant = max((k for k in prof if algorithmic_bytes(k, n_vox, 2)), key=lambda k: per_kernel[k]["avg_us"])
    dom_avg_s = per_kernel[dominant]["avg_us"] * 1e-6
    dom_bytes = algorithmic_bytes(dominant, n_vox, 2)
    achieved = dom_bytes / dom_avg_s / 1e9
    step_kernel_s = sum(v["avg_us"] * v["calls"] for v in per_kernel.values()) / args.steps * 1e-6

    # measured HBM traffic of the dominant kernel: the PMC passes committed under profiles/ count -- but only if they were
    # taken on THESE kernel sources (the file carries the hash of pdb_eda_amd/csrc at collection time); otherwise null
    traffic, rocprof_avg_us, kernels_traffic = None, None, None
    try:
        import glob
        sha = csrc_sha16()
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            with open(path) as fh:
                pmc = json.load(fh)
            if pmc.get("csrc_sha16") == sha and dominant in pmc["kernels"] and n == 256 and labels and args.nsd == 1.5:
                traffic = pmc["kernels"][dominant]["hbm_bytes_per_launch_corrected"]
                rocprof_avg_us = pmc.get("rocprofv3_avg_us", {}).get(dominant)     # (the kernel-trace average of the same sources, for comparison)
                # every kernel of the step: measured HBM bytes per launch, its algorithmic bytes where it has any, and the ratio;
                # "step": all of them against the 8 B / voxel of the labelling pass
                kernels_traffic = {}
                for k in per_kernel:
                    if k in pmc["kernels"]:
                        b, a = pmc["kernels"][k]["hbm_bytes_per_launch_corrected"], algorithmic_bytes(k, n_vox, 2)
                        kernels_traffic[k] = {"hbm_bytes": b, "algorithmic_bytes": a or None, "traffic_ratio": round(b / a, 3) if a else None}
                total = sum(v["hbm_bytes"] for v in kernels_traffic.values())
                kernels_traffic["step"] = {"hbm_bytes": total, "algorithmic_bytes": 8 * n_vox, "traffic_ratio": round(total / (8 * n_vox), 3)}
                break
    except Exception:
        traffic = None

    # ---- the same fused step on a working set LARGER than the 256 MiB Infinity Cache (informational, never `value`): the headline
    # re-labels ONE resident 64 MiB map, and FETCH_SIZE counts Infinity-Cache hits -- so this leg visits `n_bc` distinct resident maps
    # round-robin on ONE stream (each a rolled copy of the entry: other memory, the same statistics; inputs alone 8 x 64 MiB, and
    # every step writes a 64 MiB label volume): what it shows is whether the headline is inflated by the cache ----
    beyond = None
    if not args.no_beyond_cache and n == 256:
        n_bc = 8
        bc_t = [dens] + [torch.roll(dens, shifts=(17 * k, 31 * k, 5 * k), dims=(0, 1, 2)).contiguous() for k in range(1, n_bc)]
        bc_m = [dmap] + [_native.DeviceMap(ctx, t, header.geometry(), device_ptr=t.data_ptr()) for t in bc_t[1:]]
        bc_cut = []
        for mk in bc_m:
            mu, sd = mk.stats()
            bc_cut.append(mu + args.nsd * sd)
        for k in range(2 * n_bc):
            keep = bc_m[k % n_bc].full_blobs_pm(bc_cut[k % n_bc], -bc_cut[k % n_bc], labels=labels)
        barrier()
        n_steps_bc = n_bc * max(4, int(args.beyond_seconds / max(elapsed / args.steps, 1e-6) / n_bc) + 1)
        t1 = time.perf_counter()
        for k in range(n_steps_bc):
            keep = bc_m[k % n_bc].full_blobs_pm(bc_cut[k % n_bc], -bc_cut[k % n_bc], labels=labels)
        barrier()
        el_bc = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el_bc], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el_bc = float(t.item())
        ctx.profile_begin()
        for k in range(4 * n_bc):
            keep = bc_m[k % n_bc].full_blobs_pm(bc_cut[k % n_bc], -bc_cut[k % n_bc], labels=labels)
        prof_bc = ctx.profile_end()
        ms_bc = 1e3 * el_bc / n_steps_bc
        # the launch gap inside an event pair is calibrated on THIS pass, as for the headline (a host that falls behind the device between an event and its
        # launch adds its own delay to every pair: one profile run of round 5 showed +10 us on all four kernels with the step time unchanged)
        gap_bc = max(0.0, (1e3 * sum(ms for _, ms in prof_bc.values()) / (4 * n_bc) - 1e3 * ms_bc) / max(sum(c for c, _ in prof_bc.values()) / (4 * n_bc), 1))
        ker_bc = {k: max(1e3 * ms / c - gap_bc, 0.0) for k, (c, ms) in prof_bc.items()}
        beyond = {"maps": n_bc, "working_set_MiB": n_bc * 4 * n_vox / 2 ** 20 + (4 * n_vox / 2 ** 20 if labels else 0), "infinity_cache_MiB": 256,
                  "steps": n_steps_bc, "seconds": el_bc, "ms_per_step": ms_bc, "value": world * n_vox * n_steps_bc / el_bc / 1e6, "unit": "Mvoxels/s",
                  "vs_one_resident_map": (elapsed / args.steps) / (el_bc / n_steps_bc),
                  "kernels_us": {k: round(v, 2) for k, v in sorted(ker_bc.items())},
                  "pass_8B_per_voxel": {"bytes": 8 * n_vox, "achieved": 8 * n_vox / (ms_bc * 1e-3) / 1e9, "frac": 8 * n_vox / (ms_bc * 1e-3) / 1e9 / HBM_PEAK_GBS},
                  "roofline": {"bound": "hbm", "kernel": dominant, "achieved": dom_bytes / (ker_bc[dominant] * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": dom_bytes / (ker_bc[dominant] * 1e-6) / 1e9 / HBM_PEAK_GBS, "avg_launch_us": ker_bc[dominant]},
                  "note": "%d distinct 256^3 maps resident in HBM, visited round-robin on one stream: every map and every label volume has left the "
                          "Infinity Cache before it is touched again; event times minus the launch gap calibrated on this pass (%.2f us)" % (n_bc, gap_bc)}
        for mk in bc_m[1:]:
            mk.free()
        del bc_m, bc_t
        keep = step()

    # on-box device-to-device copy ceiling (report the roofline against the datasheet peak AND a measured ceiling)
    src_t = torch.empty(64 << 20, dtype=torch.float32, device="cuda")   # 256 MiB
    dst_t = torch.empty_like(src_t)
    dst_t.copy_(src_t)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        dst_t.copy_(src_t)
    ev1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * 2 * src_t.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9   # read + write bytes
    del src_t, dst_t

    # densityAnalysis leg (informational): a "~2 A entry" (SURVEY 8d config 3 stand-in) through aggregateCloud + atom / residue
    # region discrepancies + blob statistics, maps uploaded from host buffers -> entries/min of the whole per-entry pipeline
    analysis = None
    if rank == 0 and not args.no_analysis:
        analysis = guarded("analysis_entry", lambda: analysis_leg(ctx))

    # ---- BASELINE configs[3]: multiple-structure mode.  Every rank analyses ITS shard of the entries (no data-path collective):
    # `entries` synthetic entries per rank -- each = two CCP4 files of a 200^3 grid + a ~500-atom model -- read from files, parsed,
    # uploaded and analysed (aggregateCloud -> the record `pdb_eda multiple` keeps, plus atom region discrepancies) by a pool of
    # worker processes, one HIP stream each.  entries/min = entries of all ranks / max-over-ranks time. ----
    multiple = None
    if pool is not None:
        multiple = guarded("multiple_structures", lambda: multiple_leg(args, pool, rank, local_rank, world, barrier, dist if world > 1 else None, torch))

    # ---- BASELINE configs[4]: optimise-mode iterations over resident entries + the statistics reduction (informational) ----
    sweep = None
    if args.sweep_entries > 0 and args.sweep_iterations > 0:
        sweep = guarded("radius_sweep", lambda: sweep_leg(args, rank, local_rank, world, barrier, dist if world > 1 else None, torch))

    # host -> HBM upload of one entry (the boundary hands over a host buffer); never part of `value`
    t1 = time.perf_counter()
    tmp = _native.DeviceMap(ctx, grid, header.geometry())
    h2d_s = time.perf_counter() - t1
    tmp.free()

    value = world * n_vox * args.steps / elapsed / 1e6
    out = {
        "metric": "Mvoxels/s blob-labelled",
        "value": value,
        "unit": "Mvoxels/s",
        "n_gpus": world,
        "rccl_ranks": rccl_ranks,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "configs[1]: %d^3 synthetic CCP4 grid (gaussian-filtered noise, sigma_filter 1.5 voxels), fused green/red blob "
                               "labelling at +-(mean+%.1f*std), per-blob fp64 stats%s, map resident in HBM" % (n, args.nsd, " + dense int32 labels of both signs" if labels else ""),
                   "grid": [n, n, n], "cutoff_sigma": args.nsd, "labels": labels, "entries_per_rank": 1, "sharding": "one entry per rank, no data-path collective",
                   "host_affinity": ("%d cores of the GPU's NUMA node" % pinned_cpus) if pinned_cpus else "unchanged"},
        "entries_per_min": 60.0 * world * args.steps / elapsed,
        "blobs": {"green": n_green, "red": n_red, "significant_voxels": sig_vox, "all_ranks": [int(x) for x in totals.tolist()]},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": traffic, "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": 1e6 * dom_avg_s,
                     # the conservative figure: the raw HIP-event time of the launch (it includes the launch gap) -- and, when the committed
                     # rocprofv3 kernel trace was taken on these very sources, its average for the same kernel
                     "event_us": per_kernel[dominant]["event_us"], "achieved_on_event_time": dom_bytes / (per_kernel[dominant]["event_us"] * 1e-6) / 1e9,
                     "frac_on_event_time": dom_bytes / (per_kernel[dominant]["event_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                     "rocprofv3_avg_us": rocprof_avg_us,
                     "d2d_copy_ceiling_GBs": copy_gbs, "frac_of_copy_ceiling": achieved / copy_gbs,
                     # (the dispersion of the host-timed step rides in this object too: the driver's record keeps `roofline` whole)
                     "ms_per_step_windows": {"n": len(windows), "steps_each": args.steps, "min": min(windows) if windows else None,
                                             "median": float(np.median(windows)) if windows else None, "max": max(windows) if windows else None},
                     "timing": "HIP events on the launch stream (separate %d-step pass), minus the per-launch event gap calibrated in this run: "
                               "%.2f us = (sum of event times %.1f us - host-timed step %.1f us) / %.0f launches" % (args.steps, gap_us, step_events_us, step_wall_us, n_launch),
                     "pass_8B_per_voxel": {"bytes": 8 * n_vox, "kernel_sum_us": 1e6 * step_kernel_s,
                                           "achieved": 8 * n_vox / step_kernel_s / 1e9, "frac": 8 * n_vox / step_kernel_s / 1e9 / HBM_PEAK_GBS}},
        "kernels_us": {k: round(v["avg_us"] * v["calls"] / args.steps, 2) for k, v in sorted(per_kernel.items())},
        "kernels_traffic": kernels_traffic,
        "windows_ms_per_step": {"windows": [round(w, 5) for w in windows], "min": min(windows) if windows else None,
                                "median": float(np.median(windows)) if windows else None, "steps_per_window": args.steps},
        "h2d": {"upload_ms": 1e3 * h2d_s, "pcie_inclusive_Mvoxels_per_s": n_vox / (h2d_s + elapsed / args.steps) / 1e6,
                "note": "pageable host buffer -> HBM through pdbeda_map_upload; reported for information, never part of value"},
        "fallback_tiles": green.counters(),
    }
    if sigma3:
        out["sigma3"] = sigma3
    if multi:
        out["multi_stream"] = multi
    if beyond:
        out["beyond_cache"] = beyond
    if analysis:
        out["analysis_entry"] = analysis
    if multiple:
        out["multiple_structures"] = multiple
    if sweep:
        out["radius_sweep"] = sweep

    # ---- CPU baseline: the oracle (CPU restatement, O(N) clustering) on the same entry, 1 core ----
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as ora
        o = ora.Oracle(header, grid)
        done, spent = 0, 0.0
        while spent < args.cpu_seconds and done < 8:
            t1 = time.perf_counter()
            a = o.full_blobs(cut, labels=labels)
            b = o.full_blobs(-cut, labels=labels)
            spent += time.perf_counter() - t1
            done += 1
        assert len(a["n"]) == n_green and len(b["n"]) == n_red, "GPU / oracle blob counts differ"
        out["cpu_baseline"] = {"value": done * n_vox / spent / 1e6, "unit": "Mvoxels/s", "cores": 1, "kind": "port",
                               "sample": "%d x the same %d^3 entry, green+red, oracle/pdbeda_oracle.c ora_full_blobs (single thread); "
                                         "the reference's own O(N^2) clustering cannot run this size (SURVEY.md 6)" % (done, n)}
        if multiple and isinstance(multiple.get("cpu_baseline"), dict):      # the all-core baseline of the entries/min leg rides along
            out["cpu_baseline"]["multiple_structures"] = multiple["cpu_baseline"]
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
