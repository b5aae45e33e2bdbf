"""BASELINE configs[4] (optimise mode) and the value check of every `pdb_eda single` table on the MI355X path, against
goldens produced by the reference's DensityAnalysis (tests/golden/make_golden_sweep.py -> analysis_sweep.npz).
Bar: counts exact, floats 1e-5 relative (north_star) -- asserted at 1e-7."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, ANALYSIS_CASES, load_analysis_case

pytestmark = pytest.mark.gpu
REL = 1e-7


def _entries():
    from pdb_eda_amd import synthetic, multipleStructures
    out = []
    for name in ANALYSIS_CASES:
        z, spec, st, pdb, _ = load_analysis_case(name)
        dens, diff = synthetic.ccp4_bytes(spec, z["dens"]), synthetic.ccp4_bytes(spec, z["diff"])
        out.append(multipleStructures.Entry(name, (lambda d=dens, f=diff, s=st, p=pdb: (d, f, s, p)), cost_hint=len(dens)))
    return out


def _close(a, b, rel=REL):
    return a == pytest.approx(b, rel=rel, abs=1e-10)


def _serial_reduction(records, params):
    """optimizeParams.py:360-406 on one process, from the REFERENCE's records."""
    types = list(params["radii"])
    diffs = {t: [r["diffs"][t] for r in records if t in r["diffs"]] for t in types}
    slopes = {t: [r["slopes"][t] for r in records if t in r["slopes"]] for t in params["slopes"]}
    comp = {t: sum(r["atomtype_overlap_completeness"].get(t, 0) for r in records) for t in types}
    inc = {t: sum(r["atomtype_overlap_incompleteness"].get(t, 0) for r in records) for t in types}
    completeness = {t: (comp[t] / (comp[t] + inc[t]) if (comp[t] > 0 or inc[t] > 0) else 1) for t in types}
    median = {t: (np.nanmedian(v) if (v and not np.isnan(v).all()) else 0) for t, v in diffs.items()}
    size = {t: int(sum(~np.isnan(v))) if v else 0 for t, v in diffs.items()}
    sq = [x ** 2 for v in diffs.values() for x in v if not np.isnan(x)]
    std = np.sqrt(sum(sq) / (len(sq) - 1))
    mslopes = {t: np.nanmedian(v) for t, v in slopes.items() if v}
    return median, std, mslopes, size, completeness


def test_radius_sweep_vs_reference():
    """Every iteration of the sweep re-analyses the resident entries under the changed radii / slopes: per-entry records
    (diffs, slopes, overlap counters: optimizeParams.py:428-436) and the reduction over them (341-408) equal the reference's."""
    from pdb_eda_amd import synthetic, optimizeSweep
    z = np.load(os.path.join(GOLDEN, "analysis_sweep.npz"))
    entries = _entries()
    sw = optimizeSweep.Sweep(entries, device=0, n_streams=2)
    assert not sw.failures
    ratios = []
    for k, params in enumerate(synthetic.sweep_param_sets()):
        (median, mean, std, mslopes, size, completeness), records = sw.iteration(params)
        want = [json.loads(str(z["%s_k%d" % (e.pdbid, k)])) for e in entries]
        for rec, w in zip(records, want):
            assert rec and set(rec["diffs"]) == set(w["diffs"]) and set(rec["slopes"]) == set(w["slopes"])
            for t in w["diffs"]:
                assert _close(rec["diffs"][t], w["diffs"][t]), (k, rec["pdbid"], t)
            for t in w["slopes"]:
                assert _close(rec["slopes"][t], w["slopes"][t]), (k, rec["pdbid"], t)
            assert rec["atomtype_overlap_completeness"] == w["atomtype_overlap_completeness"]
            assert rec["atomtype_overlap_incompleteness"] == w["atomtype_overlap_incompleteness"]
        wm, wstd, wsl, wsize, wcomp = _serial_reduction(want, params)
        assert size == wsize and set(mslopes) == set(wsl)
        for t in wm:
            assert _close(median[t], wm[t]) and _close(completeness[t], wcomp[t])
        for t in wsl:
            assert _close(mslopes[t], wsl[t])
        assert _close(std, wstd)
        pen = optimizeSweep.penalties(median, completeness)
        assert set(pen) == set(median)
        ratios.append([w["ratio"] for w in want])
    assert len({tuple(r) for r in ratios}) == len(ratios)        # the iterations really differ (the radii changed the clouds)
    sw.close()


def test_sweep_function_shards_by_rank():
    from pdb_eda_amd import synthetic, optimizeSweep
    entries = _entries()
    sets = synthetic.sweep_param_sets()[:2]
    whole = optimizeSweep.sweep(entries, sets, device=0, n_streams=2)
    half = optimizeSweep.sweep(entries, sets, device=0, n_streams=1, rank=0, world_size=2)      # no process group: this rank's shard only
    assert len(whole) == len(half) == 2
    assert whole[0][4] != half[0][4] and max(half[0][4].values()) == (len(entries) + 1) // 2      # rank 0 of 2: every other entry, longest first


@pytest.mark.timeout(300)
def test_process_sweep_matches_thread_sweep(tmp_path):
    """Worker PROCESSES that keep their lane of entries resident give the records and reductions of the thread sweep, iteration
    by iteration; an entry whose file is missing is dropped with its reason at load time."""
    from pdb_eda_amd import synthetic, multipleStructures, optimizeSweep
    loaders = [synthetic.write_entry_files(str(tmp_path), "e%d" % k, 72, 24, 500 + k, as_paths=(k % 2 == 0)) for k in range(3)]
    entries = [multipleStructures.Entry("p%02d" % i, loaders[i % 3]) for i in range(5)]
    bad = synthetic.SyntheticEntryFiles(str(tmp_path / "missing.ccp4"), loaders[0].diff_path, 24, 500, 72, 0.5, as_paths=True)
    entries.insert(2, multipleStructures.Entry("gone", bad))
    sets = synthetic.sweep_param_sets()[:3]
    threads = optimizeSweep.Sweep(entries, device=0, n_streams=2)
    procs = optimizeSweep.ProcessSweep(entries, device=0, n_workers=2)
    try:
        assert set(threads.failures) == set(procs.failures) == {"gone"}
        for params in sets:
            (ta, tb, tc, td, te, tf), trec = threads.iteration(params)
            (pa, pb, pc, pd, pe, pf), prec = procs.iteration(params)
            assert [bool(r) for r in prec] == [bool(r) for r in trec] == [True, True, False, True, True, True]
            for a, b in zip(prec, trec):
                if b:
                    assert a["pdbid"] == b["pdbid"] and a["diffs"] == pytest.approx(b["diffs"], rel=1e-12, abs=1e-15)
                    assert a["slopes"] == pytest.approx(b["slopes"], rel=1e-12, abs=1e-15)
                    assert a["atomtype_overlap_completeness"] == b["atomtype_overlap_completeness"]
            assert pa == pytest.approx(ta, rel=1e-12, abs=1e-15) and pe == te and pf == pytest.approx(tf, rel=1e-12)
    finally:
        threads.close()
        procs.close()
    whole = optimizeSweep.sweep(entries, sets[:1], device=0, n_streams=2, processes=True)
    assert whole[0][4] == te or len(whole) == 1


NCCL_WORKER = r'''
import io, json, os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%(port)d")
import torch, torch.distributed as dist
# the process group comes first: nothing in this process has touched the GPU yet
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
import numpy as np
from conftest import ANALYSIS_CASES, load_analysis_case
from pdb_eda_amd import synthetic, multipleStructures, optimizeSweep, optimizeStats
entries = []
for name in ANALYSIS_CASES:
    z, spec, st, pdb, _ = load_analysis_case(name)
    dens, diff = synthetic.ccp4_bytes(spec, z["dens"]), synthetic.ccp4_bytes(spec, z["diff"])
    entries.append(multipleStructures.Entry(name, (lambda d=dens, f=diff, s=st, p=pdb: (d, f, s, p))))
sw = optimizeSweep.Sweep(entries, device=0, n_streams=2)
params = synthetic.sweep_param_sets()[1]
through_rccl, records = sw.iteration(params)                 # all_gather + all_reduce on device tensors over RCCL
rows = optimizeStats.gather_rows([[1.0, float("nan")], [2.0, 3.0]], 2)
counts = optimizeStats.reduce_counts(np.array([[1, 2], [3, 4]]))
dist.barrier()
dist.destroy_process_group()
plain = optimizeStats.calculateMedianDiffsSlopes(records, params)        # no group: the single-process formula
json.dump({"rccl": [through_rccl[0], through_rccl[2], through_rccl[4], through_rccl[5]], "plain": [plain[0], plain[2], plain[4], plain[5]],
           "rows": np.nan_to_num(rows, nan=-1.0).tolist(), "counts": counts.tolist(), "backend": "nccl"}, open(%(out)r, "w"))
'''


@pytest.mark.timeout(300)
def test_single_rank_rccl_reduction(tmp_path):
    """The collective branch on real hardware: a 1-rank `nccl` (= RCCL) group started in a FRESH child process before any
    GPU call; the sweep's reduction goes through all_gather / all_reduce on device tensors and equals the plain formula."""
    out = tmp_path / "out.json"
    script = tmp_path / "worker.py"
    script.write_text(NCCL_WORKER % {"root": ROOT, "port": 29000 + os.getpid() % 2000, "out": str(out)})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert proc.returncode == 0, proc.stderr[-3000:]
    got = json.loads(out.read_text())
    assert got["rccl"] == got["plain"]
    assert got["rows"] == [[1.0, -1.0], [2.0, 3.0]] and got["counts"] == [[1, 2], [3, 4]]


# ---- every `pdb_eda single` table against the reference's method outputs -------------------------------------------------

def _same(got, want, path=""):
    if isinstance(want, list):
        assert isinstance(got, (list, tuple, np.ndarray)) and len(got) == len(want), path
        for i, (g, w) in enumerate(zip(got, want)):
            _same(g, w, "%s[%d]" % (path, i))
    elif isinstance(want, bool) or isinstance(want, str) or want is None:
        assert (bool(got) if isinstance(want, bool) else got) == want, path
    elif isinstance(want, int):
        assert int(got) == want, path
    else:
        g = float(got)
        assert (np.isnan(g) and np.isnan(want)) or g == pytest.approx(want, rel=REL, abs=1e-9), (path, g, want)


@pytest.fixture(scope="module", params=ANALYSIS_CASES)
def tables(request, gpu_ctx):
    from pdb_eda_amd import ccp4, synthetic, densityAnalysis
    z, spec, st, pdb, params = load_analysis_case(request.param)
    densityAnalysis.setGlobals(params)
    dens = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["dens"])), request.param, ctx=gpu_ctx)
    diff = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["diff"])), request.param, ctx=gpu_ctx)
    densityAnalysis._attachCutoffs(dens, diff)
    st.header = {"resolution": 2.0}
    an = densityAnalysis.DensityAnalysis(request.param, dens, diff, st, pdb)
    want = json.loads(str(np.load(os.path.join(GOLDEN, "analysis_sweep.npz"))["%s_tables" % request.param]))
    return an, want


SUBMODES = [("cloud", "atom", {}), ("cloud", "residue", {}), ("cloud", "domain", {}),
            ("density", "atom", {}), ("density", "residue", {}), ("density", "symmetry-atom", {}),
            ("difference", "atom", {}), ("difference", "residue", {}), ("difference", "symmetry-atom", {}),
            ("blob", "green", {"green": True}), ("blob", "red", {"red": True}), ("blob", "blue", {}),
            ("statistics", "atom", {}), ("statistics", "residue", {})]


@pytest.mark.parametrize("mode,level,kw", SUBMODES, ids=["%s/%s" % (m, lv) for m, lv, _ in SUBMODES])
def test_single_table_values(tables, mode, level, kw):
    """The CLI defaults (radius 3.5; numSD 3.0 for difference / green / red, else 1.5): headers, row order and every value."""
    from pdb_eda_amd import singleStructure
    an, want_all = tables
    header, rows = singleStructure.rows(an, mode, "atom" if mode == "blob" else level, **kw)
    want = want_all["%s/%s" % (mode, level)]
    rows = json.loads(json.dumps(rows, default=singleStructure.numpyConverter))      # what the JSON writer emits
    if mode == "cloud":
        assert all(r[-1] == pytest.approx(want_all["ratio"], rel=REL) for r in rows)
        rows = [r[:-1] for r in rows]                      # (the ratio column main() appends, singleStructure.py:100-108)
        if level == "atom":
            assert header[:-1] == want_all["cloud/atom/names"]
        else:                                              # the reference's order inside a residue / of equal ratios follows CPython set order
            key = lambda r: (r[1], r[4], round(r[7][0], 4)) if level == "residue" else (r[4], round(r[7][0], 4))
            rows, want = sorted(rows, key=key), sorted(want, key=key)
            if level == "domain":                          # representative residue of a domain cloud: arbitrary in the reference too
                rows, want = [r[3:] for r in rows], [w[3:] for w in want]
    if level == "symmetry-atom":
        # Q11: main() post-processes columns 4 and 5 of these rows (singleStructure.py:119-121, 133-135) -- written for the
        # atom-metrics layout, where they are the symmetry tag and the coordinate; here they are the atom NAME (which becomes
        # a list of its characters) and the symmetry tag (which becomes floats).  Kept: outputs must diff cleanly.
        want = [w[:4] + [list(w[4]), [float(v) for v in w[5]]] + w[6:] for w in want]
        assert isinstance(rows[0][4], list) and isinstance(rows[0][5][0], float)
    assert len(header) - (1 if mode == "cloud" else 0) == len(want[0]) + (3 if (mode, level) == ("cloud", "domain") else 0)
    _same(rows, want, "%s/%s" % (mode, level))
