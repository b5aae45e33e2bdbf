"""Multiple-structure mode on one GPU: entries dealt to a pool of HIP streams (one context per
worker thread) give the same records as sequential processing, failed entries are dropped without
poisoning the pool, and the record matches the reference-derived goldens."""
import os

import numpy as np
import pytest

from conftest import load_analysis_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _entries(n_copies=3):
    from pdb_eda_amd import synthetic, multipleStructures
    out = []
    for k in range(n_copies):
        for name in ("orth", "hex"):
            z, spec, st, pdb, params = load_analysis_case(name)
            dens, diff = synthetic.ccp4_bytes(spec, z["dens"]), synthetic.ccp4_bytes(spec, z["diff"])
            out.append(multipleStructures.Entry("%s%d" % (name, k), (lambda d=dens, f=diff, s=st, p=pdb: (d, f, s, p)), cost_hint=len(dens)))
    out.append(multipleStructures.Entry("broken", lambda: (b"not a map", b"", None, None)))
    return out


def test_stream_pool_matches_sequential(gpu_ctx):
    from pdb_eda_amd import synthetic, densityAnalysis, multipleStructures
    densityAnalysis.setGlobals(synthetic.synthetic_params())
    entries = _entries()
    seq = {e.pdbid: multipleStructures.analyzeEntry(e, gpu_ctx, silent=True) for e in entries}
    reasons = {}
    par = multipleStructures.processEntries(entries, device=0, n_streams=3, silent=True, failures=reasons)
    assert "broken" not in par and seq["broken"] == 0
    assert set(reasons) == {"broken"} and reasons["broken"]            # the dropped entry carries its reason (ref 277-282)
    assert set(par) == {k for k, v in seq.items() if v}
    for k, rec in par.items():
        want = seq[k]
        for key in ("density_electron_ratio", "num_voxels_aggregated", "total_aggregated_electrons", "num_atoms_analyzed",
                    "num_residue_clouds_analyzed", "num_domain_clouds_analyzed", "atom_overlap_completeness"):
            assert rec["stats"][key] == pytest.approx(want["stats"][key], rel=1e-12), key
        assert rec["diffs"].keys() == want["diffs"].keys()
        for t in rec["diffs"]:
            assert rec["diffs"][t] == pytest.approx(want["diffs"][t], rel=1e-9, abs=1e-12)
        z = load_analysis_case(k[:-1])[0]
        assert rec["stats"]["density_electron_ratio"] == pytest.approx(float(z["ratio"]), rel=1e-8)
        assert rec["stats"]["num_voxels_aggregated"] == int(z["num_voxels"])
    # the reduction of optimise mode over these records (single rank here; 2 ranks in the gloo test)
    from pdb_eda_amd import optimizeStats
    med = optimizeStats.calculateMedianDiffsSlopes(list(par.values()), densityAnalysis.paramsGlobal)[0]
    assert set(med) == set(densityAnalysis.paramsGlobal["radii"])
    order = multipleStructures.shard(entries, 0, 2)
    assert order[0].cost_hint >= order[-1].cost_hint


@pytest.mark.timeout(240)
@pytest.mark.parametrize("shape,roughness", [((40, 48, 256), 0.5), ((24, 40, 600), 0.5),      # one tile column; three (the last one partial)
                                             ((40, 48, 256), 1.5)])                                # smooth: dense and wide tiles instead of unit tiles
def test_unit_fallback_on_many_streams(shape, roughness):
    """Dense maps (tiles overflow LDS: "unit tiles", labelled and united by two launches of their own in a second run of the
    job -- round 5; rounds 3-4 did both inside k_face_merge behind flags the workgroups polled, and a late dispatch on a busy
    GPU failed the job) labelled concurrently on EIGHT streams: every stream gets the oracle's answer (counts, keys and the
    label volume), every job ran twice and none failed."""
    import io
    from oracle import oracle as ora
    from pdb_eda_amd import ccp4, synthetic, multipleStructures
    g = synthetic.smooth_noise(shape, 9, roughness)  # 0.5 = nearly white: ~30 word-runs a mask word, far beyond the 4 096 a tile holds in LDS
    spec = synthetic.MapSpec(ncrs=shape[::-1])
    blob = synthetic.ccp4_bytes(spec, g)
    header = ccp4.DensityHeader.fromFileHeader(blob[:1024])
    mean, std = float(np.mean(g, dtype=np.float64)), float(np.std(g.astype(np.float64)))
    cut = mean + 0.3 * std
    want = ora.Oracle(header, g).full_blobs(cut, labels=True)

    def work(k, ctx):
        dm = ccp4.parse(io.BytesIO(blob), "dense%d" % k, ctx=ctx)
        out = []
        for _ in range(4):
            green, red = dm._map.full_blobs_pm(cut, -cut, labels=True)
            st = green.stats()
            same = np.array_equal(st["n"], want["n"]) and np.array_equal(st["firstKey"], want["firstKey"]) and \
                np.array_equal(green.labels(dm._map.unique_shape), want["labels"])
            c = green.counters()
            out.append((same, c["unit_tiles_runs"] + c["unit_tiles_comps"] if roughness < 1.0 else c["run_ids"], c["reruns"]))
        return out
    res = multipleStructures.StreamPool(device=0, n_streams=8).map(work, list(range(16)))
    assert all(r != 0 for r in res)
    assert all(ok for r in res for ok, _, _ in r)
    assert all(n_unit > 0 for r in res for _, n_unit, _ in r)      # the fallback path really ran
    assert all(reruns == (1 if roughness < 1.0 else 0) for r in res for _, _, reruns in r)   # ... as the second run of its job; dense / wide tiles need none


def test_device_failure_stops_the_pool():
    """A library / device failure is not an entry failure: the pool re-raises it instead of reporting 0."""
    from pdb_eda_amd import _native, multipleStructures

    def work(k, ctx):
        if k == 2:
            ctx.check(-1, "synthetic device failure")
        return k + 1
    with pytest.raises(_native.PdbedaError):
        multipleStructures.StreamPool(device=0, n_streams=2, silent=True).map(work, list(range(6)))


@pytest.mark.timeout(120)
def test_watchdog_abandons_a_context():
    """--time-out (multipleStructures.py:297-304, 359-377): a stream that does not drain in time fails ITS entry with the
    reason "Timeout"; the context is abandoned (every later call on it fails at once, destroy does not wait) and the
    worker carries on with a fresh one.  The slow job is real work that takes ~0.5 s on ONE compute unit: a single
    testOverlap pair of two far-apart sets of 10^5 voxels each (10^10 coordinate comparisons in one workgroup)."""
    from pdb_eda_amd import _native, multipleStructures
    n = 100000
    a = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.int32)
    b = a + np.array([0, 50, 50], dtype=np.int32)
    crs, off = np.concatenate([a, b]), np.array([0, n, 2 * n], dtype=np.int64)

    def work(k, ctx):
        if k == 1:
            return bool(ctx.test_overlap(crs, off, [0], [1])[0]) + 10
        return bool(ctx.test_overlap(crs[:50], np.array([0, 25, 50], dtype=np.int64), [0], [1])[0]) + 1
    pool = multipleStructures.StreamPool(device=0, n_streams=1, time_out=0.05, silent=True)
    res = pool.map(work, [0, 1, 2])
    assert res == [2, 0, 2]                      # (25 consecutive voxels next to the following 25: they touch)
    assert pool.failures == {1: "Timeout"}
    # a timed-out context stays dead; a disarmed one waits as long as it takes
    ctx = _native.Context(0)
    ctx.set_timeout(0.02)
    with pytest.raises(_native.PdbedaTimeout):
        ctx.test_overlap(crs, off, [0], [1])
    with pytest.raises(_native.PdbedaTimeout):
        ctx.synchronize()
    ctx2 = _native.Context(0)
    assert not ctx2.test_overlap(crs, off, [0], [1])[0]
    # abandoned contexts are parked, not leaked: once their streams have drained the library reaps them (every context
    # creation tries; so does an allocation that is about to fail) -- with the arenas their lost handles held
    ctx.close()
    import time
    lib = _native.lib()
    deadline = time.time() + 30
    while lib.pdbeda_reap_abandoned() != 0 and time.time() < deadline:
        time.sleep(0.05)
    assert lib.pdbeda_reap_abandoned() == 0


def _sorted_lists(crs, off):
    """Voxel lists with every blob's rows sorted (the order inside a blob is the order its runs took their places in: not fixed)."""
    crs = np.asarray(crs).copy()
    for a, b in zip(off[:-1], off[1:]):
        rows = crs[a:b]
        crs[a:b] = rows[np.lexsort((rows[:, 2], rows[:, 1], rows[:, 0]))]
    return crs, np.asarray(off)


@pytest.mark.timeout(120)
def test_staged_results_under_the_watchdog_equal_those_without(gpu_ctx):
    """with PDBEDA_COPY_KERNELS (the default) every staged result -- packed tables, counters, regional sums -- is written by a
    KERNEL into the context's pinned block, and under the watchdog the host reads it after a hipStreamQuery POLL instead of a stream
    synchronize.  The same calls on a context with an armed (generous) time-out and on one without: identical results, bit for bit."""
    from pdb_eda_amd import _native, ccp4, synthetic
    g = synthetic.smooth_noise((40, 48, 96), 17, 1.5)
    spec = synthetic.MapSpec(ncrs=(96, 48, 40), spacing=0.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    xyz = np.array([[6.0, 7.0, 8.0], [12.5, 9.25, 6.0], [20.0, 11.0, 9.5], [30.0, 12.0, 10.0]])
    rad = np.array([1.5, 2.0, 1.2, 2.4])
    out = []
    for timed in (False, True):
        ctx = _native.Context(0)
        if timed:
            ctx.set_timeout(60.0)
        m = _native.DeviceMap(ctx, g, header.geometry())
        mean, std = m.stats()
        cut = mean + 1.5 * std
        green, red = m.full_blobs_pm(cut, -cut, labels=True)
        sp = m.sphere_blobs(xyz, rad, np.arange(5), cut)
        sums = m.region_sums(xyz, rad, np.arange(5), cut) if hasattr(m, "region_sums") else None
        rec = {"mean": mean, "std": std, "green": green.stats(), "red": red.stats(), "labels": green.labels(m.unique_shape), "sphere": sp.stats(), "vox": _sorted_lists(*sp.voxels()), "sums": sums}
        out.append(rec)
        ctx.close()

    def same(a, b):
        if isinstance(a, dict):
            return set(a) == set(b) and all(same(a[k], b[k]) for k in a)
        if isinstance(a, (tuple, list)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if a is None:
            return b is None
        return np.array_equal(np.asarray(a), np.asarray(b))
    for key in out[0]:
        assert same(out[0][key], out[1][key]), key


@pytest.mark.timeout(120)
def test_watchdog_deadline_is_per_entry_not_per_wait():
    """One deadline for ALL waits of an entry (the reference's SIGALRM is around the whole of analyzePDBID): many short waits
    that each stay below the time-out still time out together once their sum passes it; re-arming starts a new entry."""
    import time
    from pdb_eda_amd import _native
    n = 20000
    a = np.stack([np.arange(n), np.zeros(n), np.zeros(n)], axis=1).astype(np.int32)
    crs, off = np.concatenate([a, a + np.array([0, 50, 50], dtype=np.int32)]), np.array([0, n, 2 * n], dtype=np.int64)
    ctx = _native.Context(0)
    ctx.test_overlap(crs, off, [0], [1])                     # warm
    t0 = time.perf_counter()
    ctx.test_overlap(crs, off, [0], [1])
    one = time.perf_counter() - t0                           # one call = one wait of ~20 ms of work on one compute unit
    ctx.set_timeout(max(4 * one, 0.02))
    with pytest.raises(_native.PdbedaTimeout):
        for _ in range(40):                                  # 40 waits, each well below the time-out: the ENTRY is over time
            ctx.test_overlap(crs, off, [0], [1])
    ctx.close()
    ctx = _native.Context(0)
    for _ in range(6):                                       # re-armed per entry: never expires
        ctx.set_timeout(max(4 * one, 0.02))
        ctx.test_overlap(crs, off, [0], [1])
        ctx.test_overlap(crs, off, [0], [1])


@pytest.mark.timeout(120)
def test_a_timed_out_upload_does_not_wedge_the_upload_engine(tmp_path):
    """The readers of the upload engine belong to the process, not to a context: an upload whose entry runs out of time (its context is
    abandoned) must leave them serving everybody else.  A context with a deadline that has all but passed uploads a 27 MB file --
    PdbedaTimeout, or, if the box was quick, a map --, while and after which other contexts upload the same file and get its bytes."""
    import threading
    import numpy as np
    from pdb_eda_amd import _native, ccp4, synthetic
    spec = synthetic.MapSpec(ncrs=(200, 176, 190), spacing=0.45)
    grid = synthetic.smooth_noise((190, 176, 200), seed=8, sigma_voxels=1.3)
    path = tmp_path / "t.ccp4"
    path.write_bytes(synthetic.ccp4_bytes(spec, grid))
    geom = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec)).geometry()
    good, errors = [], []

    def ordinary(k):
        try:
            ctx = _native.Context(0)
            for _ in range(4):
                m = _native.DeviceMap.from_file(ctx, str(path), 1024, False, geom)
                good.append(bool(np.array_equal(m.download().reshape(grid.shape), grid)))
                m.free()
            ctx.close()
        except BaseException as exception:
            errors.append(exception)
    others = [threading.Thread(target=ordinary, args=(k,)) for k in range(2)]
    for t in others:
        t.start()
    timed_out = 0
    for _ in range(6):
        ctx = _native.Context(0)
        ctx.set_timeout(2e-4)                       # 0.2 ms: the file needs ~1 ms alone, more beside two other uploaders
        try:
            m = _native.DeviceMap.from_file(ctx, str(path), 1024, False, geom)
            m.free()
        except _native.PdbedaTimeout:
            timed_out += 1
        ctx.close()                                 # (an abandoned context: parked, reaped once the copies behind it have drained)
    for t in others:
        t.join()
    assert not errors, errors
    assert len(good) == 8 and all(good)
    assert timed_out >= 1
    ctx = _native.Context(0)                        # ... and afterwards
    m = _native.DeviceMap.from_file(ctx, str(path), 1024, False, geom)
    assert np.array_equal(m.download().reshape(grid.shape), grid)
    m.free()
    ctx.close()


@pytest.mark.timeout(300)
def test_process_pool_matches_sequential(tmp_path, gpu_ctx):
    """BASELINE configs[3] shape: entries read from CCP4 files by worker PROCESSES (spawn; one stream each) give the records of
    sequential processing; a broken file is dropped with its reason and the pool carries on."""
    from pdb_eda_amd import synthetic, densityAnalysis, multipleStructures
    params = synthetic.synthetic_params()
    densityAnalysis.setGlobals(params)
    loaders = [synthetic.write_entry_files(str(tmp_path), "e%d" % k, 72, 24, 300 + k) for k in range(2)]
    entries = [multipleStructures.Entry("x%02d" % i, loaders[i % 2]) for i in range(6)]
    bad = synthetic.SyntheticEntryFiles(str(tmp_path / "missing.ccp4"), loaders[0].diff_path, 24, 300, 72, 0.5)
    entries.insert(3, multipleStructures.Entry("gone", bad))
    by_path = synthetic.SyntheticEntryFiles(loaders[1].density_path, loaders[1].diff_path, 24, 301, 72, 0.5, as_paths=True)
    entries[5] = multipleStructures.Entry("x04", by_path)        # the same entry handed over as file paths (page cache -> HBM)
    seq = [multipleStructures.analyzeEntry(e, gpu_ctx, silent=True) for e in entries]
    pool = multipleStructures.ProcessPool(device=0, n_workers=2, params=params, silent=True)
    try:
        pool.warm()
        par = pool.map(entries)
        failures = dict(pool.failures)
        again = pool.map(entries[:2])            # the workers (and their contexts) are reused
    finally:
        pool.close()
    assert [bool(r) for r in par] == [bool(r) for r in seq] == [True, True, True, False, True, True, True]
    assert par[3] == 0 and set(failures) == {"gone"} and "Error" in failures["gone"]
    for a, b in zip(par + again, seq + seq[:2]):
        if not b:
            continue
        assert a["pdbid"] == b["pdbid"] and a["stats"]["num_voxels_aggregated"] == b["stats"]["num_voxels_aggregated"]
        assert a["stats"]["density_electron_ratio"] == pytest.approx(b["stats"]["density_electron_ratio"], rel=1e-12)
        assert a["diffs"] == pytest.approx(b["diffs"], rel=1e-9, abs=1e-12)


def test_ccp4_file_goes_straight_to_the_device(tmp_path, gpu_ctx):
    """``ccp4.read`` of a mode-2 file: header parsed on the host, grid from the file to HBM through the library's pinned double
    buffer -- the same resident map as parsing the bytes (both endiannesses, symmetry records skipped, more than two chunks),
    the host copy only on demand; a truncated file is an entry-level error (OSError), not a device failure."""
    import numpy as np
    from pdb_eda_amd import _native, ccp4, synthetic
    spec = synthetic.MapSpec(ncrs=(160, 130, 150), spacing=0.45)          # 12.5 MB grid: four 4 MiB chunks, the last one partial
    grid = synthetic.smooth_noise((150, 130, 160), seed=11, sigma_voxels=1.2)
    raw = synthetic.ccp4_bytes(spec, grid, symmetry_bytes=b"X" * 160)
    big = synthetic.ccp4_bytes(spec, grid, big_endian=True)
    for name, payload in (("little.ccp4", raw), ("big.ccp4", bytes(big))):
        path = tmp_path / name
        path.write_bytes(payload)
        want = ccp4.parse(__import__("io").BytesIO(payload), "ref", ctx=gpu_ctx)
        got = ccp4.read(str(path), "file", ctx=gpu_ctx)
        assert got._density is None                                        # no host copy was made
        assert got.header.ncrs == want.header.ncrs and got.header.endian == want.header.endian
        assert (got.meanDensity, got.stdDensity) == (want.meanDensity, want.stdDensity)
        assert got.numStoredVoxels == want.densityArray.size
        cut = want.meanDensity + 2.0 * want.stdDensity
        a, b = got.createFullBlobList(cut), want.createFullBlobList(cut)
        assert len(a) == len(b) > 10 and all(x.numVoxels == y.numVoxels and x.totalDensity == y.totalDensity for x, y in zip(a, b))
        assert np.array_equal(got.density, want.density) and got.densityArray.shape == want.densityArray.shape
    short = tmp_path / "short.ccp4"
    short.write_bytes(raw[:len(raw) - 4096])
    with pytest.raises(Exception) as err:                                  # the size check sends it down the parsing path, which fails on the count
        ccp4.read(str(short), "short", ctx=gpu_ctx)
    assert not isinstance(err.value, _native.PdbedaError)
    from pdb_eda_amd._native import DeviceMap
    hdr = ccp4.DensityHeader.fromFileHeader(raw[:1024])
    with pytest.raises(OSError):                                           # the library's own check of the file length
        DeviceMap.from_file(gpu_ctx, str(short), 1024 + 160, False, hdr.geometry())
    with pytest.raises(OSError):
        DeviceMap.from_file(gpu_ctx, str(tmp_path / "absent.ccp4"), 1024, False, hdr.geometry())
    assert len(ccp4.read(str(tmp_path / "little.ccp4"), ctx=gpu_ctx).createFullBlobList(cut)) == len(b)      # the context is still good


def test_lazy_diff_map(tmp_path, gpu_ctx, monkeypatch):
    """``ccp4.read(..., lazy=True)``: header now, grid when first asked for -- the same map as the eager read; the loader of
    multiple-structure mode leaves the Fo-Fc grid of an entry on disk (its record reads the header only) and gives the record of
    the loader that uploads both maps (PDBEDA_EAGER_DIFF_MAP=1), file by file and through the process pool."""
    from pdb_eda_amd import ccp4, synthetic, densityAnalysis, multipleStructures
    params = synthetic.synthetic_params()
    densityAnalysis.setGlobals(params)
    loader = synthetic.write_entry_files(str(tmp_path), "lazy", 72, 24, 411, as_paths=True)
    eager = ccp4.read(loader.diff_path, "e", ctx=gpu_ctx)
    lazy = ccp4.read(loader.diff_path, "l", ctx=gpu_ctx, lazy=True)
    assert eager.resident and not lazy.resident
    assert lazy.header.ncrs == eager.header.ncrs and lazy.header.densityMean == eager.header.densityMean and not lazy.resident
    assert lazy.diffDensityCutoff == eager.meanDensity + 3 * eager.stdDensity and lazy.resident        # first use brought it in
    assert (lazy.meanDensity, lazy.stdDensity) == (eager.meanDensity, eager.stdDensity)
    green = lazy.createFullBlobList(lazy.diffDensityCutoff)
    assert [b.numVoxels for b in green] == [b.numVoxels for b in eager.createFullBlobList(eager.diffDensityCutoff)]
    # the loader of multiple-structure mode
    entry = multipleStructures.Entry("lazy", loader)
    assert multipleStructures.lazyDiffMap()
    loaded = multipleStructures.loadEntry(entry, gpu_ctx)
    assert loaded[0].resident and not loaded[1].resident
    rec_lazy = multipleStructures.analyzeEntry(entry, gpu_ctx, silent=True, loaded=loaded)
    assert rec_lazy and not loaded[1].resident                              # the record never touched the Fo-Fc grid
    an = densityAnalysis.DensityAnalysis("lazy", *loaded)
    assert len(an.greenBlobList) == len(green) and loaded[1].resident       # ... and an analysis that does gets it
    monkeypatch.setenv("PDBEDA_EAGER_DIFF_MAP", "1")
    assert not multipleStructures.lazyDiffMap()
    loaded = multipleStructures.loadEntry(entry, gpu_ctx)
    assert loaded[1].resident
    rec_eager = multipleStructures.analyzeEntry(entry, gpu_ctx, silent=True, loaded=loaded)
    for rec in (rec_lazy, rec_eager):
        rec.pop("execution_time")
    assert rec_lazy == rec_eager and rec_lazy["stats"]["diff_density_mean"] == eager.header.densityMean
    monkeypatch.delenv("PDBEDA_EAGER_DIFF_MAP")
    entries = [multipleStructures.Entry("p%d" % i, loader) for i in range(5)]
    pool = multipleStructures.ProcessPool(device=0, n_workers=2, params=params, silent=True)
    try:
        par = pool.map(entries)
    finally:
        pool.close()
    for rec in par:
        rec.pop("execution_time"); rec.pop("pdbid")
    want = dict(rec_lazy); want.pop("pdbid")
    assert all(rec == want for rec in par)


def test_file_upload_into_a_recycled_arena(tmp_path, monkeypatch):
    """the helper reader's chunks go through a second stream, and the pool hands out arenas whose last user may still
    be queued on the context's stream (maps and lists are freed without a host sync) -- the second stream has to wait for that
    work before its first chunk lands.  A whole-map job is queued on a large map, lists and map are freed with no accessor in
    between, and a file larger than one 4 MiB chunk goes straight into the arena that comes back (poisoned: 0xFF is queued on
    the context's stream too); the download must equal the file."""
    import numpy as np
    from pdb_eda_amd import _native, ccp4, synthetic
    monkeypatch.setenv("PDBEDA_DEBUG_POISON", "1")
    ctx = _native.Context(0)
    spec = synthetic.MapSpec(ncrs=(200, 176, 190), spacing=0.45)           # 26.8 MB: seven chunks, both readers busy
    grid = synthetic.smooth_noise((190, 176, 200), seed=5, sigma_voxels=1.3)
    path = tmp_path / "m.ccp4"
    path.write_bytes(synthetic.ccp4_bytes(spec, grid))
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    for rep in range(4):
        dmap = _native.DeviceMap(ctx, grid if rep % 2 == 0 else grid[::-1].copy(), header.geometry())     # same size: its arena is the one that comes back
        mean, std = dmap.stats()
        green, red = dmap.full_blobs_pm(mean + 1.5 * std, -(mean + 1.5 * std), labels=True)
        green.free(); red.free(); dmap.free()                                # kernels still queued; nothing has synchronised
        got = _native.DeviceMap.from_file(ctx, str(path), 1024, False, header.geometry())
        assert np.array_equal(got.download().reshape(grid.shape), grid), rep
        got.free()
    ctx.close()


@pytest.mark.timeout(240)
def test_upload_engine_serves_many_loads_at_once(tmp_path):
    """Round 5: file uploads go through ONE engine per process -- three reader threads take the chunks of every load in flight from
    a FIFO, chunk sizes ramp up, a context's stream waits for an event behind each reader's copies.  Six threads (a context each)
    upload files of very different sizes at once, over and over -- a few voxels (one short chunk), sizes that end in the middle of
    the ramp, a big-endian file (swapped on the device), 27 MB (full chunks on every reader): every download must equal its file,
    and the statistics that ride along must be numpy's."""
    import threading
    import numpy as np
    from pdb_eda_amd import _native, ccp4, synthetic
    shapes = [(3, 5, 7), (40, 33, 29), (64, 64, 64), (100, 90, 75), (128, 120, 110), (200, 176, 190)]
    files = []
    for k, shp in enumerate(shapes):
        spec = synthetic.MapSpec(ncrs=shp[::-1], spacing=0.5)
        grid = synthetic.smooth_noise(shp, seed=40 + k, sigma_voxels=1.2)
        path = tmp_path / ("u%d.ccp4" % k)
        path.write_bytes(synthetic.ccp4_bytes(spec, grid, big_endian=(k == 3)))
        header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec, big_endian=(k == 3)))
        files.append((str(path), grid, header.geometry(), k == 3))
    errors = []

    def work(t):
        try:
            ctx = _native.Context(0)
            for rep in range(6):
                path, grid, geom, swapped = files[(t + rep) % len(files)]
                got = _native.DeviceMap.from_file(ctx, path, 1024, swapped, geom)
                mean, std = got.stats()
                assert np.array_equal(got.download().reshape(grid.shape), grid), (t, rep)
                assert mean == float(np.mean(grid, dtype=np.float64)) or abs(mean - float(np.mean(grid.astype(np.float64)))) < 1e-12
                assert abs(std - float(np.std(grid.astype(np.float64)))) < 1e-9
                got.free()
            ctx.close()
        except BaseException as exception:
            errors.append((t, exception))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors


def test_file_upload_brings_the_statistics_along(gpu_ctx, tmp_path):
    """pdbeda_map_upload_file_stats: a map read from a file comes with its mean / std from the same wait (and the quantum of
    its blob sums): exactly the numbers pdbeda_map_stats gives on the same grid uploaded from memory (== np.mean / np.std)."""
    import io
    from pdb_eda_amd import ccp4, synthetic
    g = synthetic.smooth_noise((37, 50, 71), 12, 1.5)
    spec = synthetic.MapSpec(ncrs=(71, 50, 37))
    path = tmp_path / "stats.ccp4"
    path.write_bytes(synthetic.ccp4_bytes(spec, g))
    from_file = ccp4.read(str(path), ctx=gpu_ctx)
    assert from_file._map.__dict__.get("_file_stats") is not None
    from_memory = ccp4.parse(io.BytesIO(path.read_bytes()), "m", ctx=gpu_ctx)
    assert (from_file.meanDensity, from_file.stdDensity) == (from_memory.meanDensity, from_memory.stdDensity)
    assert from_file.meanDensity == float(np.mean(g, dtype=np.float64)) and from_file.stdDensity == float(np.std(g.astype(np.float64)))
    a, b = from_file.createFullBlobList(from_file.meanDensity + 1.5 * from_file.stdDensity), from_memory.createFullBlobList(from_memory.meanDensity + 1.5 * from_memory.stdDensity)
    assert len(a) == len(b) and [x.totalDensity for x in a] == [y.totalDensity for y in b]      # (the same quantum: bit-equal sums)


AB_WORKER = r'''
import io, json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, densityAnalysis, structure
ctx = _native.Context(0)
spec, header, st, params, dens, diff, rot = synthetic.cube_entry((64, 60, 56), 60, 21, 0.55)
densityAnalysis.setGlobals(params)
d0 = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), "ab", ctx=ctx)
d1 = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), "ab", ctx=ctx)
densityAnalysis._attachCutoffs(d0, d1)
pdb = structure.PDBEntry(structure.PDBHeader(pdbid="ab", resolution=2.0, spaceGroup="P_1", rotationMats=rot))
an = densityAnalysis.DensityAnalysis("ab", d0, d1, st, pdb)
an.aggregateCloud()
out = {"ratio": an.densityElectronRatio, "voxels": an.numVoxelsAggregated, "density": an.totalAggregatedDensity,
       "medians": {k: {t: float(v) for t, v in d.items()} for k, d in an.medians.items()},
       "residues": [[r[1], r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.residueCloudDescriptions],
       "atoms": an.calculateAtomRegionDiscrepancies(3.5, 3.0), "blobs": [list(map(repr, row)) for row in an.calculateAtomSpecificBlobStatistics(an.greenBlobList)],
       # (groups of several atoms -- a residue's: round 6 makes their boxes and volumes on the host too, PDBEDA_HOST_BOXES=0 on the device)
       "residue_regions": an.calculateResidueRegionDiscrepancies(3.5, 3.0), "residue_density": an.calculateResidueRegionDensity(2.0)}
cols = structure.columns(st)
off = np.concatenate([[0], np.cumsum(np.bincount(cols.res_of_atom, minlength=len(cols.residues)))]).astype(np.int64)
gb = d0._map.sphere_blobs(cols.coord, np.full(len(cols.atoms), 1.4, dtype=np.float32), off, float(d0.meanDensity + 1.5 * d0.stdDensity))
gst = gb.stats()
out["grouped_blobs"] = {"n": gst["n"].tolist(), "key": gst["firstKey"].tolist(), "group": gst["group"].tolist(), "total": [float.hex(float(x)) for x in gst["totalDensity"]]}
json.dump(out, open(%(out)r, "w"), default=lambda o: o.tolist() if hasattr(o, "tolist") else repr(o))
'''


@pytest.mark.timeout(300)
def test_the_ab_switches_change_no_result(tmp_path):
    """PDBEDA_COPY_KERNELS=0 (the runtime's copies instead of kernels over the pinned block) and PDBEDA_HOST_BOXES=0 (the device's three kernels
    instead of host-made boxes and volumes) are the A/B switches of round 5: the same analysis in a fresh process under each setting gives
    the same numbers to the last bit (aggregateCloud's tables and medians, a region table, a blob table)."""
    import json
    import subprocess
    import sys
    outs = []
    # (round 6: PDBEDA_ATOM_ENGINE=0 -- the five generic kernels instead of the one fused launch for per-atom sphere batches;
    #  PDBEDA_UNORDERED_UNION=0 -- aggregateCloud's union job with painted keys, ranks and a packed blob table instead of k_union_finish;
    #  PDBEDA_ATOM_REGION=0 -- per-atom regional sums by paint + reduce + pack instead of the one fused launch)
    for k, env_extra in enumerate(({}, {"PDBEDA_COPY_KERNELS": "0"}, {"PDBEDA_HOST_BOXES": "0"}, {"PDBEDA_ATOM_ENGINE": "0"}, {"PDBEDA_UNORDERED_UNION": "0"}, {"PDBEDA_ATOM_REGION": "0"})):
        out, script = tmp_path / ("out%d.json" % k), tmp_path / ("worker%d.py" % k)
        script.write_text(AB_WORKER % {"root": ROOT, "out": str(out)})
        proc = subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=280)
        assert proc.returncode == 0, proc.stderr[-3000:]
        outs.append(json.loads(out.read_text()))
    assert outs[0]["voxels"] > 0 and len(outs[0]["atoms"]) > 100 and len(outs[0]["blobs"]) > 5 and len(outs[0]["residue_regions"]) > 20 and len(outs[0]["grouped_blobs"]["n"]) > 20
    assert all(o == outs[0] for o in outs[1:])

