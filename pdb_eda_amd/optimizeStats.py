"""The statistics reduction of optimise mode across ranks (one process per GPU).

``optimizeParams.calculateMedianDiffsSlopes`` (optimizeParams.py:341-408) fans entries out with
``Pool.starmap`` and reduces the per-entry records in the parent: per-atom-type nan-median / mean of
the ``diffs`` and ``slopes`` vectors, and integer sums of the overlap-completeness counters.  Here
every rank analyses its shard of entries on its GPU and the reduction is the path's ONE exchange
step: an all-gather of the per-entry rows (a median is not sum-reducible) and an all-reduce(sum) of
the int64 counters -- ``torch.distributed`` (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in
the CPU tests).  Messages are KB-scale: latency-bound, nothing to tune.
"""
import numpy as np


def _dist():
    import torch
    import torch.distributed as dist
    return torch, dist


def gather_rows(local_rows, width):
    """All-gather a ragged list of fixed-width float64 rows (NaN = missing)."""
    torch, dist = _dist()
    rows = np.asarray(local_rows, dtype=np.float64).reshape(-1, width)
    if not (dist.is_available() and dist.is_initialized()):
        return rows                      # no process group: a single process holds every record
    # (a 1-rank group still goes through the collectives: the RCCL path is the same code at every world size)
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=device)
    counts = [torch.zeros_like(n) for _ in range(dist.get_world_size())]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.full((cap, width), float("nan"), dtype=torch.float64, device=device)
    if rows.shape[0]:
        padded[:rows.shape[0]] = torch.from_numpy(rows).to(device)
    parts = [torch.empty_like(padded) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, padded)
    return np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)], axis=0)


def reduce_counts(local_counts):
    torch, dist = _dist()
    c = np.asarray(local_counts, dtype=np.int64)
    if not (dist.is_available() and dist.is_initialized()):
        return c
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.from_numpy(c.copy()).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def all_ranks_ok(error=None):
    """Every rank calls this between its local work and the reduction: a rank whose device or worker failed must not simply
    leave -- the others would wait in the all-gather for ever.  All-reduce (min) of one 'ok' flag; if any rank reports a failure,
    EVERY rank raises (the failing one with its own error), so the job ends instead of hanging."""
    torch, dist = _dist()
    if not (dist.is_available() and dist.is_initialized()):
        if error is not None:
            raise error
        return
    device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    flag = torch.tensor([0 if error is not None else 1], dtype=torch.int64, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if error is not None:
        raise error
    if int(flag.item()) == 0:
        raise RuntimeError("another rank failed during this iteration: the statistics reduction is abandoned on every rank")


def calculateMedianDiffsSlopes(local_records, params):
    """ref optimizeParams.py:341-408 over the records of ALL ranks.

    ``local_records``: this rank's per-entry records (``multipleStructures.analyzeEntry``); failed
    entries (0) are skipped, as the reference skips missing result files.
    Returns (medianDiffs, meanDiffs, overallStdDevDiffs, medianSlopes, sizeDiffs, overlapCompleteness).
    """
    types = list(params["radii"])
    col = {t: i for i, t in enumerate(types)}
    nt = len(types)
    d_rows, s_rows = [], []
    counts = np.zeros((2, nt), dtype=np.int64)
    for rec in local_records:
        if not rec:
            continue
        d = np.full(nt, np.nan)
        s = np.full(nt, np.nan)
        for t, v in rec["diffs"].items():
            if t in col:
                d[col[t]] = v
        for t, v in rec.get("slopes", {}).items():
            if t in col:
                s[col[t]] = v
        d_rows.append(d)
        s_rows.append(s)
        for t, n in rec["atomtype_overlap_completeness"].items():
            if t in col:
                counts[0, col[t]] += n
        for t, n in rec["atomtype_overlap_incompleteness"].items():
            if t in col:
                counts[1, col[t]] += n
    diffs = gather_rows(d_rows, nt)
    slopes = gather_rows(s_rows, nt)
    counts = reduce_counts(counts)

    def colstat(a, fn, empty):
        out = {}
        for t, i in col.items():
            v = a[:, i] if len(a) else np.zeros(0)
            out[t] = fn(v) if (len(v) and not np.isnan(v).all()) else empty
        return out

    medianDiffs = colstat(diffs, np.nanmedian, 0)
    meanDiffs = colstat(diffs, np.nanmean, 0)
    sizeDiffs = {t: int((~np.isnan(diffs[:, i])).sum()) if len(diffs) else 0 for t, i in col.items()}
    sq = diffs[~np.isnan(diffs)] ** 2 if len(diffs) else np.zeros(0)
    overallStdDevDiffs = float(np.sqrt(sq.sum() / (len(sq) - 1))) if len(sq) > 1 else float("nan")
    medianSlopes = {t: v for t, v in colstat(slopes, np.nanmedian, float("nan")).items() if not np.isnan(v)}
    completeness = {}
    for t, i in col.items():
        c, n = int(counts[0, i]), int(counts[1, i])
        completeness[t] = c / (c + n) if (c > 0 or n > 0) else 1
    return medianDiffs, meanDiffs, overallStdDevDiffs, medianSlopes, sizeDiffs, completeness
