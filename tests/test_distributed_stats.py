"""The N > 1 path on CPU: world_size-2 gloo run of the optimise-mode statistics reduction
(all-gather of per-entry rows + all-reduce of counters) against the single-process formula of
optimizeParams.calculateMedianDiffsSlopes; and the longest-first round-robin sharding."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from pdb_eda_amd import optimizeStats, multipleStructures
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
records = json.load(open(%(rec)r))
params = json.load(open(%(par)r))
entries = [multipleStructures.Entry(r["pdbid"] if r else "bad%%d" %% i, None, cost_hint=(r["execution_time"] if r else 0)) for i, r in enumerate(records)]
mine = multipleStructures.shard(entries, dist.get_rank(), 2)
by_id = {(r["pdbid"] if r else "bad%%d" %% i): r for i, r in enumerate(records)}
out = optimizeStats.calculateMedianDiffsSlopes([by_id[e.pdbid] for e in mine], params)
if dist.get_rank() == 0:
    json.dump([out[0], out[1], out[2], out[3], out[4], out[5], [e.pdbid for e in mine]], open(%(out)r, "w"))
dist.barrier()
dist.destroy_process_group()
'''


def _records(params, n=13, seed=5):
    rng = np.random.default_rng(seed)
    types = list(params["radii"])
    recs = []
    for i in range(n):
        if i % 6 == 5:
            recs.append(0)          # a failed entry
            continue
        present = [t for t in types if rng.random() > 0.25]
        recs.append({"pdbid": "e%03d" % i, "diffs": {t: float(rng.normal(0, 0.2)) for t in present},
                     "slopes": {t: float(rng.normal(-0.5, 0.1)) for t in present if rng.random() > 0.3},
                     "atomtype_overlap_completeness": {t: int(rng.integers(0, 40)) for t in present},
                     "atomtype_overlap_incompleteness": {t: int(rng.integers(0, 5)) for t in present},
                     "execution_time": float(rng.uniform(0.1, 3.0))})
    return recs


def _serial(records, params):
    """optimizeParams.py:360-406 restated on one process."""
    types = list(params["radii"])
    diffs = {t: [] for t in types}
    slopes = {t: [] for t in types}
    comp = {t: 0 for t in types}
    inc = {t: 0 for t in types}
    for r in records:
        if not r:
            continue
        for t, v in r["diffs"].items():
            diffs[t].append(v)
        for t, v in r["slopes"].items():
            slopes[t].append(v)
        for t, v in r["atomtype_overlap_completeness"].items():
            comp[t] += v
        for t, v in r["atomtype_overlap_incompleteness"].items():
            inc[t] += v
    med = {t: (np.nanmedian(v) if v else 0) for t, v in diffs.items()}
    mean = {t: (np.nanmean(v) if v else 0) for t, v in diffs.items()}
    size = {t: len(v) for t, v in diffs.items()}
    sq = [x ** 2 for v in diffs.values() for x in v]
    std = np.sqrt(sum(sq) / (len(sq) - 1))
    ms = {t: np.nanmedian(v) for t, v in slopes.items() if v}
    cc = {t: (comp[t] / (comp[t] + inc[t]) if (comp[t] > 0 or inc[t] > 0) else 1) for t in types}
    return med, mean, std, ms, size, cc


def test_two_rank_reduction(tmp_path):
    from pdb_eda_amd import synthetic
    params = synthetic.synthetic_params()
    records = _records(params)
    rec, par, out = tmp_path / "rec.json", tmp_path / "par.json", tmp_path / "out.json"
    rec.write_text(json.dumps(records))
    par.write_text(json.dumps(params))
    port = 29500 + os.getpid() % 2000
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, port=port, rec=str(rec), par=str(par), out=str(out)))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = json.loads(out.read_text())
    want = _serial(records, params)
    for g, w in zip(got[:2], want[:2]):
        for t in w:
            assert g[t] == np.float64(w[t]) or abs(g[t] - w[t]) < 1e-15
    assert abs(got[2] - want[2]) < 1e-12
    assert got[3].keys() == want[3].keys()
    for t in want[3]:
        assert abs(got[3][t] - want[3][t]) < 1e-15
    assert got[4] == want[4]
    for t in want[5]:
        assert abs(got[5][t] - want[5][t]) < 1e-15
    # sharding: longest first, round robin
    valid = sorted([r for r in records if r], key=lambda r: -r["execution_time"])
    assert got[6][0] == valid[0]["pdbid"] and got[6][1] == valid[2]["pdbid"]


def test_single_process_matches_serial():
    from pdb_eda_amd import synthetic, optimizeStats
    params = synthetic.synthetic_params()
    records = _records(params, n=9, seed=2)
    got = optimizeStats.calculateMedianDiffsSlopes(records, params)
    want = _serial(records, params)
    assert got[0] == {t: (float(v) if not isinstance(v, int) else v) for t, v in want[0].items()} or all(abs(got[0][t] - want[0][t]) < 1e-15 for t in want[0])
    assert abs(got[2] - want[2]) < 1e-12
    assert got[4] == want[4]


OK_WORKER = """
import sys, os, json
sys.path.insert(0, %(root)r)
import torch.distributed as dist
rank = int(sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="%(port)d")
dist.init_process_group("gloo", rank=rank, world_size=2)
from pdb_eda_amd import optimizeStats
optimizeStats.all_ranks_ok(None)                       # everybody fine: returns on both ranks
try:
    optimizeStats.all_ranks_ok(ValueError("device lost") if rank == 1 else None)
    outcome = "returned"
except ValueError as e:
    outcome = "own:" + str(e)
except RuntimeError as e:
    outcome = "other:" + str(e)[:20]
open(%(out)r + str(rank), "w").write(outcome)
dist.destroy_process_group()
"""


def test_a_failing_rank_ends_the_iteration_on_every_rank(tmp_path):
    """A rank whose device / worker failed does not leave the others waiting in the all-gather: the 'ok' flag is all-reduced
    first and every rank raises (world_size 2, gloo)."""
    port = 31500 + os.getpid() % 2000
    script = tmp_path / "ok_worker.py"
    out = str(tmp_path / "outcome")
    script.write_text(OK_WORKER % dict(root=ROOT, port=port, out=out))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)]) for r in range(2)]
    for p in procs:
        assert p.wait(timeout=240) == 0
    assert open(out + "0").read().startswith("other:another rank failed")
    assert open(out + "1").read() == "own:device lost"
