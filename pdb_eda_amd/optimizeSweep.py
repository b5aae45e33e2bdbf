"""Optimise mode on MI355X: the per-iteration work of ``pdb_eda optimize`` (BASELINE configs[4]).

One iteration of the reference's radius optimisation (optimizeParams.py:232-243) is
``calculateMedianDiffsSlopes(pdbids, {**params, "radii": currentRadii, "slopes": currentSlopes})``: every entry is
re-analysed with the candidate parameter table (``processFunction``, 410-448: ``setGlobals`` -> ``aggregateCloud`` ->
record of diffs / slopes / overlap counters) and the records are reduced to per-atom-type medians, sizes and an overlap
completeness (341-408).  Here every rank re-analyses ITS shard of the entries on its GPU (stream pool, one entry per
stream) and the reduction is the path's one exchange step (``optimizeStats``: all-gather of the rows + all-reduce of the
counters over RCCL / xGMI).  The descent logic that picks the next radius (optimizeParams.py:176-327) is control plane
and stays with the caller: ``sweep`` takes the parameter tables to evaluate.

Entries keep their parsed maps resident in HBM between iterations (``ResidentEntry``): an iteration changes radii and
slopes, never the maps, so only the analysis is repeated.
"""
import time

import numpy as np

from . import _native, densityAnalysis, multipleStructures, optimizeStats


class ResidentEntry(object):
    """An entry whose two maps were parsed and uploaded once (by the worker thread that owns ``ctx``)."""

    def __init__(self, entry, ctx):
        dens, diff, self.biopdbObj, self.pdbObj = entry.loader()
        self.pdbid = entry.pdbid
        self.densityObj = multipleStructures.loadMap(dens, entry.pdbid, ctx)
        self.diffDensityObj = multipleStructures.loadMap(diff, entry.pdbid, ctx)
        densityAnalysis._attachCutoffs(self.densityObj, self.diffDensityObj)


def processRecord(analyzer, params, startTime=None):
    """ref optimizeParams.py:423-436: the record one entry contributes to an iteration, or 0 (Q7: no ratio)."""
    ratio = analyzer.densityElectronRatio
    if not ratio:
        return 0
    corrected = analyzer.medians['corrected_density_electron_ratio']
    diffs = {t: float((corrected[t] - ratio) / ratio) for t in params["radii"] if t in corrected and not np.isnan(corrected[t])}
    slopes = {t: float(analyzer.medians['slopes'][t]) for t in params["slopes"]
              if t in analyzer.medians['slopes'] and not np.isnan(analyzer.medians['slopes'][t])}
    return {"pdbid": analyzer.pdbid, "diffs": diffs, "slopes": slopes, "resolution": analyzer.pdbObj.header.resolution,
            "execution_time": (time.thread_time() - startTime) if startTime is not None else 0.0,
            "atomtype_overlap_completeness": dict(analyzer.atomTypeOverlapCompleteness),
            "atomtype_overlap_incompleteness": dict(analyzer.atomTypeOverlapIncompleteness)}


class Sweep(object):
    """The entries of ONE rank, resident on its GPU, ready to be re-analysed under changing parameter tables.

    ``entries``: this rank's shard (``multipleStructures.shard``).  Stream k of the pool owns entries k, k + S, ...:
    a resident map is only ever touched by the context that uploaded it."""

    def __init__(self, entries, device=0, n_streams=4, silent=True):
        self.silent = silent
        self.failures = {}
        self.n_entries = len(entries)
        n = max(1, min(int(n_streams), max(1, len(entries))))
        self._pool = multipleStructures.StreamPool(device, n, silent=silent)
        self.lanes = [[] for _ in range(n)]             # stream -> [(index, ResidentEntry)]

        def load(k, ctx):
            for i in range(k, len(entries), n):
                try:
                    self.lanes[k].append((i, ResidentEntry(entries[i], ctx)))
                except _native.PdbedaError:
                    raise
                except Exception as exception:
                    multipleStructures._drop(entries[i].pdbid, "%s: %s" % (type(exception).__name__, exception), self.failures, silent)
            return 1
        self._pool.each(load)

    def iteration(self, params):
        """One ``calculateMedianDiffsSlopes`` (optimizeParams.py:341-408) over the entries of ALL ranks for ``params``:
        ((medianDiffs, meanDiffs, overallStdDevDiffs, medianSlopes, sizeDiffs, overlapCompleteness), this rank's records)."""
        densityAnalysis.setGlobals(params)               # ref 421-422 (every worker of the reference loads the same table)
        records = [0] * self.n_entries

        def run(k, ctx):
            for i, res in self.lanes[k]:
                t0 = time.thread_time()
                analyzer = densityAnalysis.DensityAnalysis(res.pdbid, res.densityObj, res.diffDensityObj, res.biopdbObj, res.pdbObj)
                try:
                    records[i] = processRecord(analyzer, params, t0)
                except _native.PdbedaError:
                    raise
                except Exception as exception:
                    multipleStructures._drop(res.pdbid, "%s: %s" % (type(exception).__name__, exception), self.failures, self.silent)
            return 1
        error = None
        try:
            self._pool.each(run)
        except Exception as exception:          # (a device failure of this rank: the other ranks must hear of it before the reduction)
            error = exception
        optimizeStats.all_ranks_ok(error)
        return optimizeStats.calculateMedianDiffsSlopes(records, params), records

    def close(self):
        self.lanes = []
        self._pool.close()


_FIXED_TABLES = ("full_atom_name_map_atom_type", "full_atom_name_map_electrons", "bonded_atoms")


def _splice_fixed_tables(params, kept):
    """Replace the name tables of ``params`` by the objects kept from an earlier, equal table (and keep new ones)."""
    for name in _FIXED_TABLES:
        if name in params:
            if name in kept and kept[name] == params[name]:
                params[name] = kept[name]
            else:
                kept[name] = params[name]
    return params


def _sweep_worker(conn, device, lane, silent):
    """A worker process of ProcessSweep: owns one context and the resident maps of its lane of entries; serves iterations."""
    try:
        _native.pin_to_device(device)
        ctx = _native.Context(device)
        failures, resident = {}, []
        for i, entry in lane:
            try:
                resident.append((i, ResidentEntry(entry, ctx)))
            except _native.PdbedaError:
                raise
            except Exception as exception:
                multipleStructures._drop(entry.pdbid, "%s: %s" % (type(exception).__name__, exception), failures, silent)
        conn.send(("ready", failures))
        kept = {}
        while True:
            params = conn.recv()
            if params is None:
                return
            # Every iteration arrives as a freshly unpickled dict: equal tables, new objects.  The per-structure flattening is
            # cached on the IDENTITY of the three name tables (densityAnalysis._cloudInputs), so the first objects are kept and
            # spliced into later tables that compare equal: only radii / slopes change between the iterations of a sweep.
            _splice_fixed_tables(params, kept)
            densityAnalysis.setGlobals(params)
            records, failures = {}, {}
            for i, res in resident:
                t0 = time.thread_time()
                analyzer = densityAnalysis.DensityAnalysis(res.pdbid, res.densityObj, res.diffDensityObj, res.biopdbObj, res.pdbObj)
                try:
                    records[i] = processRecord(analyzer, params, t0)
                except _native.PdbedaError:
                    raise
                except Exception as exception:
                    multipleStructures._drop(res.pdbid, "%s: %s" % (type(exception).__name__, exception), failures, silent)
            conn.send(("records", records, failures))
    except BaseException as exception:            # a device / library failure (or a broken pipe): tell the parent, then die
        try:
            conn.send(("error", "%s: %s" % (type(exception).__name__, exception)))
        except Exception:
            pass


class ProcessSweep(object):
    """``Sweep`` with worker PROCESSES (spawned; one context = stream each): the host side of an iteration -- flattening the
    structure, the statistics tail -- holds the GIL, so threads do not scale it (measured: 1.36 ms per entry and iteration
    with four threads against 1.50 with one); a process owns its entries' resident maps for the life of the sweep.
    Same interface: ``iteration(params)`` -> (reduction over ALL ranks, this rank's records), ``failures``, ``close()``."""

    def __init__(self, entries, device=0, n_workers=4, silent=True):
        import multiprocessing
        self.silent = silent
        self.failures = {}
        self.n_entries = len(entries)
        n = max(1, min(int(n_workers), max(1, len(entries))))
        mp = multiprocessing.get_context("spawn")
        self._workers = []
        for k in range(n):
            parent, child = mp.Pipe()
            lane = [(i, entries[i]) for i in range(k, len(entries), n)]
            proc = mp.Process(target=_sweep_worker, args=(child, device, lane, silent), daemon=True)
            proc.start()
            child.close()
            self._workers.append((proc, parent))
        for message in self._gather():
            self.failures.update(message[1])

    def _gather(self):
        out = []
        for proc, conn in self._workers:
            try:
                message = conn.recv()
            except EOFError:
                message = ("error", "worker process ended")
            if message[0] == "error":
                self.close()
                raise _native.PdbedaError("sweep worker failed: %s" % message[1])
            out.append(message)
        return out

    def iteration(self, params):
        records = [0] * self.n_entries
        error = None
        try:
            for _, conn in self._workers:
                conn.send(params)
            for _, part, failures in self._gather():
                self.failures.update(failures)
                for i, record in part.items():
                    records[i] = record
        except Exception as exception:          # (a worker of this rank failed: tell the other ranks before they enter the reduction)
            error = exception
        optimizeStats.all_ranks_ok(error)
        densityAnalysis.setGlobals(params)
        return optimizeStats.calculateMedianDiffsSlopes(records, params), records

    def close(self):
        for proc, conn in self._workers:
            try:
                conn.send(None)
            except Exception:
                pass
        for proc, conn in self._workers:
            proc.join(timeout=30)
            if proc.is_alive():
                proc.kill()            # (the exact process this object started)
            conn.close()
        self._workers = []


def sweep(entries, paramSets, device=0, n_streams=4, rank=0, world_size=1, silent=True, processes=False):
    """Evaluate ``paramSets`` (a sequence of parameter tables, e.g. one changed radius per step) over ``entries``:
    this rank keeps its shard resident, and every iteration ends in the RCCL reduction.  Returns one reduction tuple per set.
    ``processes``: worker processes instead of threads (``ProcessSweep``)."""
    mine = multipleStructures.shard(entries, rank, world_size)
    sw = ProcessSweep(mine, device, n_streams, silent) if processes else Sweep(mine, device, n_streams, silent)
    try:
        return [sw.iteration(p)[0] for p in paramSets]
    finally:
        sw.close()


def penalties(medianDiffs, overlapCompleteness, inversePenaltyWeight=2.0):
    """ref optimizeParams.py:166-167: the quantity the optimiser compares between iterations."""
    top = max(overlapCompleteness.values())
    return {t: medianDiffs[t] + (overlapCompleteness[t] - top) / inversePenaltyWeight for t in medianDiffs}
