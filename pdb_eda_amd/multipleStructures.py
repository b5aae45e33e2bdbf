"""Multiple-structure mode on MI355X: entries are independent, so they are sharded over ranks
(one process per GPU) and, inside a rank, over a small pool of HIP streams.

Replaces the *dispatch* of ``pdb_eda/multipleStructures.py`` -- ``multiprocessing.Pool().map(
processFunction, pdbids, chunksize=1)`` (multipleStructures.py:167-168) and the temp-JSON hand-off
(fileUtils.py:12-28) -- and reproduces the per-entry record of ``analyzePDBID``
(multipleStructures.py:320-356).  HIP contexts do not survive ``fork`` (what ``Pool`` does), so the
pool here is threads: one ``pdbeda_ctx`` (= one stream) per worker thread; ctypes releases the GIL
inside every C-ABI call, so streams overlap.  ``writeResults`` emits the reference's JSON / CSV (182-194).
Filters and reload: out of scope.
"""
import collections
import csv
import io
import os
import json
import sys
import queue
import threading
import time

import numpy as np

from . import _native, ccp4, densityAnalysis
from . import structure as _structure

statsHeaders = ['density_electron_ratio', 'voxel_volume', 'f000', 'num_voxels_aggregated', 'total_aggregated_electrons', 'density_mean',
                'diff_density_mean', 'resolution', 'space_group', 'num_atoms_analyzed', 'num_residue_clouds_analyzed',
                'num_domain_clouds_analyzed', 'atom_overlap_completeness']


class Entry(object):
    """One PDB entry: a loader returning (2Fo-Fc map, Fo-Fc map, structure, pdbObj); a map is either the bytes of a CCP4
    file (a download held in memory, densityAnalysis.py:96-99 of the reference) or the path of one on disk (its local-mirror
    mode), which goes from the page cache to HBM without a host copy (``ccp4.read``)."""

    def __init__(self, pdbid, loader, cost_hint=0.0):
        self.pdbid = pdbid
        self.loader = loader
        self.cost_hint = cost_hint      # e.g. last iteration's execution_time (optimizeParams.py:392-393)


def shard(entries, rank, world_size):
    """Longest-first (the reference's heuristic, optimizeParams.py:392-393), then dealt round-robin."""
    order = sorted(range(len(entries)), key=lambda i: -entries[i].cost_hint)
    return [entries[i] for i in order[rank::world_size]]


def _drop(pdbid, reason, failures, silent):
    """ref multipleStructures.py:277-282, 297-304: a failed entry is reported on stderr (unless --silent) and dropped."""
    if failures is not None:
        failures[pdbid] = reason
    if not silent:
        print(pdbid, reason, file=sys.stderr)
    return 0


def loadMap(source, pdbid, ctx=None, lazy=False):
    """A DensityMatrix from what an entry loader hands over: CCP4 bytes, or the path of a CCP4 file (``lazy``: resident on
    first use, see ``ccp4.read``)."""
    if isinstance(source, (str, os.PathLike)):
        return ccp4.read(os.fspath(source), pdbid, ctx=ctx, lazy=lazy)
    return ccp4.parse(io.BytesIO(source), pdbid, ctx=ctx)


def lazyDiffMap():
    """The record of ``pdb_eda multiple`` reads the Fo-Fc map's HEADER (``diff_density_mean``) and nothing of its grid, so the
    grid of an entry's Fo-Fc file is uploaded only if something asks for it.  PDBEDA_EAGER_DIFF_MAP=1 brings it in with the
    2Fo-Fc map as the reference's loader does (densityAnalysis.py:145-148) -- bench.py times both."""
    return os.environ.get("PDBEDA_EAGER_DIFF_MAP", "0") in ("", "0")


def loadEntry(entry, ctx=None):
    """The first half of an entry: its loader, the 2Fo-Fc map resident on ``ctx`` (file or bytes -> HBM) with mean / std and the
    default cutoff, the Fo-Fc map's header read and its grid ready to follow (``lazyDiffMap``).  Returns (densityObj,
    diffDensityObj, biopdbObj, pdbObj); raises what the loader / parser raise."""
    dens, diff, biopdbObj, pdbObj = entry.loader()
    densityObj = loadMap(dens, entry.pdbid, ctx)
    diffDensityObj = loadMap(diff, entry.pdbid, ctx, lazy=lazyDiffMap())
    densityAnalysis._attachCutoffs(densityObj, diffDensityObj)
    return densityObj, diffDensityObj, biopdbObj, pdbObj


def analyzeEntry(entry, ctx=None, failures=None, silent=False, loaded=None):
    """ref multipleStructures.py:320-356: one entry -> result record, or 0 when the ENTRY fails (its files do not load or
    parse, the analysis raises, or there is no density-electron ratio: Q7); the reason goes to ``failures[pdbid]`` and to
    stderr like the reference's processFunction (277-282).  Device / library failures (``_native.PdbedaError``: no memory,
    a HIP fault, a missing library, the watchdog) are NOT entry failures: they propagate.
    ``loaded``: the result of ``loadEntry`` (or the exception it raised) when the maps were brought in ahead of time."""
    startTime = time.thread_time()
    try:
        if isinstance(loaded, BaseException):
            raise loaded
        densityObj, diffDensityObj, biopdbObj, pdbObj = loaded if loaded is not None else loadEntry(entry, ctx)
        analyzer = densityAnalysis.DensityAnalysis(entry.pdbid, densityObj, diffDensityObj, biopdbObj, pdbObj)
        ratio = analyzer.densityElectronRatio
    except _native.PdbedaError:
        raise
    except Exception as exception:
        return _drop(entry.pdbid, "%s: %s" % (type(exception).__name__, exception), failures, silent)
    if not ratio:
        return _drop(entry.pdbid, "no density-electron ratio (total aggregated electrons below the minimum)", failures, True)
    corrected = analyzer.medians['corrected_density_electron_ratio']
    diffs = {atomType: ((corrected[atomType] - ratio) / ratio) if atomType in corrected else 0 for atomType in sorted(densityAnalysis.paramsGlobal["radii"])}
    complete = sum(analyzer.atomTypeOverlapCompleteness.values())
    incomplete = sum(analyzer.atomTypeOverlapIncompleteness.values())
    if complete > 0 or incomplete > 0:
        complete = complete / (complete + incomplete)
    # 'f000' needs the structure factors' F000 estimate (densityAnalysis.py F000, out of scope: DESIGN.md 7): None
    stats = {'density_electron_ratio': ratio, 'voxel_volume': densityObj.header.unitVolume, 'f000': None,
             'num_voxels_aggregated': analyzer.numVoxelsAggregated, 'total_aggregated_electrons': analyzer.totalAggregatedElectrons,
             'density_mean': densityObj.header.densityMean, 'diff_density_mean': diffDensityObj.header.densityMean,
             'resolution': pdbObj.header.resolution, 'space_group': pdbObj.header.spaceGroup,
             'num_atoms_analyzed': analyzer.cloudCounts[0], 'num_residue_clouds_analyzed': analyzer.cloudCounts[1],
             'num_domain_clouds_analyzed': analyzer.cloudCounts[2], 'atom_overlap_completeness': complete}
    properties = dict(getattr(biopdbObj, "header", None) or {})          # the structure header items (ref 346)
    cols = _structure.columns(biopdbObj)          # (the snapshot the analysis made: its lists, not a second walk of the object tree)
    properties['residue_counts'] = dict(collections.Counter(cols.res_name))
    properties['element_counts'] = dict(collections.Counter([atom.element for atom in cols.atoms]))
    slopes = {t: float(v) for t, v in analyzer.medians['slopes'].items() if not np.isnan(v)}
    return {"pdbid": entry.pdbid, "diffs": {k: float(v) for k, v in diffs.items()}, "stats": stats, "slopes": slopes,
            "atomtype_overlap_completeness": dict(analyzer.atomTypeOverlapCompleteness),
            "atomtype_overlap_incompleteness": dict(analyzer.atomTypeOverlapIncompleteness),
            "execution_time": time.thread_time() - startTime, "properties": properties}


class StreamPool(object):
    """N worker threads on one GPU, each with its own context (HIP stream + device arena cache); the contexts live as long
    as the pool, so repeated ``map`` / ``each`` calls (the iterations of optimise mode) reuse streams and arenas.

    ``map(fn, items)`` calls ``fn(item, ctx)`` from whichever worker is free and returns the results in order;
    ``each(fn)`` calls ``fn(k, ctx_k)`` once on every worker k (work pinned to a stream).  Per item:
      * an ordinary exception drops the item (result 0) with its reason in ``self.failures[index]`` and on stderr
        (multipleStructures.py:297-304);
      * ``time_out`` seconds (the reference's --time-out, 359-377: a SIGALRM around the whole of analyzePDBID) arm the
        library's watchdog with ONE deadline per item (re-armed when the item starts): a stream that does not drain by
        then fails the item with reason "Timeout", the worker ABANDONS that context (never waits on it again, never
        re-execs the process; the library reaps its memory when its stream has drained) and carries on with a fresh
        one.  Host-side phases cannot be interrupted from a thread: an item whose wall time exceeded ``time_out`` is
        dropped as "Timeout" when it returns (weaker than the reference's signal, same outcome for the result set);
      * any other ``PdbedaError`` is a device failure: the pool stops handing out work, joins its threads and re-raises it.
    """

    def __init__(self, device=0, n_streams=4, time_out=0.0, silent=False):
        self.device = device
        self.n_streams = max(1, int(n_streams))
        self.time_out = float(time_out or 0.0)
        self.silent = silent
        self.failures = {}
        self._ctx = [None] * self.n_streams

    def context(self, k, fresh=False):
        if fresh or self._ctx[k] is None:
            ctx = _native.Context(self.device)
            if self.time_out > 0:
                ctx.set_timeout(self.time_out)
            self._ctx[k] = ctx
        return self._ctx[k]

    def close(self):
        self._ctx = [None] * self.n_streams

    def _run(self, n_workers, body):
        fatal = []
        stop = threading.Event()

        def work(k):
            try:
                body(k, stop)
                ctx = self._ctx[k]
                if ctx is not None and not stop.is_set():
                    try:
                        ctx.synchronize()
                    except _native.PdbedaTimeout:
                        self.context(k, fresh=True)
            except BaseException as exception:       # device failure (or interpreter shutdown): stop the pool, report once
                fatal.append(exception)
                stop.set()
        threads = [threading.Thread(target=work, args=(k,)) for k in range(n_workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if fatal:
            raise fatal[0]

    def _call(self, fn, k, index, item, results):
        try:
            ctx = self.context(k)
            t0 = time.monotonic()
            if self.time_out > 0:
                ctx.set_timeout(self.time_out)      # the item's deadline starts now
            result = fn(item, ctx)
            if self.time_out > 0 and time.monotonic() - t0 > self.time_out:
                raise _native.PdbedaTimeout("the item took %.3f s (host phases included)" % (time.monotonic() - t0))
            results[index] = result
        except _native.PdbedaTimeout:
            self.failures[index] = "Timeout"
            _drop(getattr(item, "pdbid", index), "Timeout", None, self.silent)
            self.context(k, fresh=True)              # the old context is abandoned with whatever still runs on it
        except _native.PdbedaError:
            raise
        except Exception as exception:
            self.failures[index] = "%s: %s" % (type(exception).__name__, exception)
            _drop(getattr(item, "pdbid", index), self.failures[index], None, self.silent)

    def map(self, fn, entries):
        todo = queue.Queue()
        for i, e in enumerate(entries):
            todo.put((i, e))
        results = [0] * len(entries)
        self.failures = {}

        def body(k, stop):
            while not stop.is_set():
                try:
                    i, e = todo.get_nowait()
                except queue.Empty:
                    return
                self._call(fn, k, i, e, results)
        self._run(min(self.n_streams, max(1, len(entries))), body)
        return results

    def each(self, fn):
        results = [0] * self.n_streams
        self.failures = {}
        self._run(self.n_streams, lambda k, stop: self._call(fn, k, k, k, results))
        return results


# ---- process pool: the reference's own shape (multiprocessing.Pool, multipleStructures.py:167-168) on a GPU -----------------------
# Threads share one interpreter: the host-side part of an entry (building Python tables) holds the GIL, so a thread pool
# scales only the part spent inside the library.  Worker PROCESSES scale all of it: each is a fresh interpreter (spawn --
# HIP state does not survive fork) with its own context = stream on the same GPU.

_worker_state = {}


def _worker_init(device, params, time_out, silent):
    _worker_state.update(device=device, time_out=time_out, silent=silent)
    _native.pin_to_device(device)          # the worker's reads, parsing and uploads stay on the GPU's socket
    if params is not None:
        densityAnalysis.setGlobals(params)


def _worker_context(k=0):
    ctxs = _worker_state.setdefault("ctxs", [None] * (N_WORKER_CONTEXTS * max(1, WORKER_LANES)))
    if ctxs[k] is None:
        ctxs[k] = _native.Context(_worker_state["device"])
        if _worker_state["time_out"] > 0:
            ctxs[k].set_timeout(_worker_state["time_out"])
    return ctxs[k]


def _worker_entry(entry):
    """One entry in a worker process: (record or 0, failure reason or None).  A time-out abandons the worker's context."""
    return _worker_chunk([entry])[0]


N_WORKER_CONTEXTS = 3    # contexts (= HIP streams, arena pools) ONE lane of a pool worker rotates its entries over
WORKER_LANES = int(os.environ.get("PDBEDA_WORKER_LANES", "1"))      # lanes of a pool worker: entries analysed at once (threads of the worker)
WORKER_DEPTH = int(os.environ.get("PDBEDA_WORKER_DEPTH", "2"))      # entries a lane loads ahead of the one it analyses (<= N_WORKER_CONTEXTS - 1)


def _worker_chunk(entries):
    """A few entries in a worker process, as a pipeline: while entry i is analysed (host-side table building: holds the GIL)
    helper threads bring the maps of the next ``WORKER_DEPTH`` entries into HBM on the lane's other contexts (file reads and
    PCIe copies happen inside the library, GIL released).  A worker may also run several such lanes (``WORKER_LANES`` threads
    over interleaved shares of the chunk, each with its own contexts).  Round 5 measured the shapes against each other with
    the process-wide upload engine underneath (tools/prof_pool.py pools, four workers, both maps): one lane two entries ahead
    1.44 ms per entry, two lanes one ahead 1.48, two lanes two ahead 1.46, three lanes 1.47 -- and three workers 1.51, two
    1.57: the pool sits on the link's rate for chunked copies plus ~0.17 ms of per-entry work the processes do not overlap,
    whatever its shape; the default stays the simplest (one lane, two ahead).  With a time-out there is one lane, one entry
    deep (an entry's ONE deadline starts with its upload: it must not spend it queueing).  Returns [(record or 0, failure
    reason or None)]; a time-out abandons the context it happened on."""
    time_out = _worker_state["time_out"]
    lanes = 1 if time_out > 0 else max(1, min(WORKER_LANES, len(entries)))
    if lanes == 1:
        return _worker_lane(entries, 0)
    out = [None] * len(entries)
    errors = []

    def run(k):
        try:
            for j, pair in enumerate(_worker_lane(entries[k::lanes], k * N_WORKER_CONTEXTS)):
                out[k + j * lanes] = pair
        except BaseException as exception:       # a device failure in one lane: the chunk fails with it (after the other lane has returned)
            errors.append(exception)
    threads = [threading.Thread(target=run, args=(k,), daemon=True) for k in range(1, lanes)]
    for t in threads:
        t.start()
    run(0)
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out


def _worker_lane(entries, ctx_base):
    """One lane of ``_worker_chunk``: its entries in order, on the contexts ``ctx_base`` .. ``ctx_base + N_WORKER_CONTEXTS - 1``."""
    silent = _worker_state["silent"]

    time_out = _worker_state["time_out"]
    depth = 1 if time_out > 0 else max(1, min(WORKER_DEPTH, N_WORKER_CONTEXTS - 1))
    started = {}

    def context(i):
        return _worker_context(ctx_base + i % N_WORKER_CONTEXTS)

    def load(i, box):
        try:
            ctx = context(i)
            started[i] = time.monotonic()
            if time_out > 0:
                ctx.set_timeout(time_out)            # the entry's ONE deadline starts with its upload
            box.append(loadEntry(entries[i], ctx))
        except BaseException as exception:           # handed to the analysing side, which sorts entry errors from device errors
            box.append(exception)

    def start(i):
        box = []
        thread = threading.Thread(target=load, args=(i, box), daemon=True)
        thread.start()
        return thread, box

    out = []
    pending = collections.deque(start(k) for k in range(min(depth, len(entries))))
    for i, entry in enumerate(entries):
        thread, box = pending.popleft()
        thread.join()
        if i + depth < len(entries):
            pending.append(start(i + depth))         # (its context is free: the entry that used it has returned its record)
        reasons = {}
        try:
            record = analyzeEntry(entry, context(i), reasons, silent, loaded=box[0])
            if time_out > 0 and record and time.monotonic() - started.get(i, time.monotonic()) > time_out:
                _drop(entry.pdbid, "Timeout", reasons, silent)     # (host phases cannot be interrupted: judged when the entry returns)
                record = 0
        except _native.PdbedaTimeout:
            _worker_state["ctxs"][ctx_base + i % N_WORKER_CONTEXTS] = None
            _drop(entry.pdbid, "Timeout", reasons, silent)
            record = 0
        except BaseException:
            for other, _ in pending:
                other.join()                         # (do not leave the helpers running into a dying call)
            raise
        out.append((record, reasons.get(entry.pdbid)))
    return out


class ProcessPool(object):
    """``n_workers`` worker processes on one GPU (one stream each), started with the ``spawn`` method when the pool is made
    (ideally before the parent touches the GPU) and reused across ``map`` calls.  Entries and their loaders must be picklable.
    A worker that DIES (killed for memory, a fault that takes the process down) does not hang the pool: the call in flight
    raises ``PdbedaError`` -- like any other device failure it is not an entry-level condition."""

    def __init__(self, device=0, n_workers=4, params=None, time_out=0.0, silent=True):
        import concurrent.futures
        import multiprocessing
        self.n_workers = max(1, int(n_workers))
        self._pool = concurrent.futures.ProcessPoolExecutor(self.n_workers, mp_context=multiprocessing.get_context("spawn"),
                                                            initializer=_worker_init, initargs=(device, params, float(time_out or 0.0), silent))
        self.failures = {}
        self.run(_worker_hold, range(self.n_workers))       # the executor starts a process per task it cannot hand to an idle one

    def run(self, fn, items):
        """``fn(item)`` for every item on the workers (results in order); a dead worker raises ``PdbedaError``."""
        from concurrent.futures.process import BrokenProcessPool
        try:
            return list(self._pool.map(fn, items, chunksize=1))
        except BrokenProcessPool as broken:
            raise _native.PdbedaError("a worker process of the pool died (%s): its GPU context is gone, the entries in flight are lost" % broken)

    def warm(self):
        """Make every worker import the package, load the library and create its context (first-use costs out of the way)."""
        self.run(_worker_warm, range(4 * self.n_workers))

    def map(self, entries, chunk=None):
        """Records of ``entries`` in order (0 for a failed entry, its reason in ``self.failures``).  The entries go to the
        workers in chunks (default: at most 16 entries each, and a whole number of rounds over the workers -- 125 entries on 4
        workers are 8 chunks of 16, not 9 of 14 with a round left over) so that a worker can bring the next entries' maps in
        while it analyses the current one (``_worker_chunk``); every chunk starts with an upload nothing overlaps, so chunks of
        16 instead of 8 took the pool from 1.77 to 1.51 ms per entry (tools/prof_pipeline2.py)."""
        entries = list(entries)
        if chunk is None and os.environ.get("PDBEDA_POOL_CHUNK"):
            chunk = int(os.environ["PDBEDA_POOL_CHUNK"])      # (experiments)
        if chunk is None:
            rounds = max(1, -(-len(entries) // (16 * self.n_workers)))
            chunk = max(1, -(-len(entries) // (rounds * self.n_workers)))
        chunks = [entries[k:k + chunk] for k in range(0, len(entries), chunk)]
        results = [pair for part in self.run(_worker_chunk, chunks) for pair in part]
        self.failures = {e.pdbid: why for e, (rec, why) in zip(entries, results) if why}
        return [rec for rec, _ in results]

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None


def _worker_hold(_):
    import time as _time
    _time.sleep(0.2)       # (long enough that the next task finds no idle worker and a new process is started for it)
    return 1


def _worker_warm(_):
    _worker_context(0)
    _worker_context(1)
    import time as _time
    _time.sleep(0.05)      # (keeps the task on this worker long enough for the others to take the next ones)
    return 1


def processEntries(entries, device=0, n_streams=4, time_out=0.0, silent=False, failures=None):
    """The per-GPU part of ``pdb_eda multiple``: {pdbid: record} for the entries that succeed; the reasons of the ones that
    do not are collected in ``failures`` ({pdbid: reason}) when a dict is passed."""
    pool = StreamPool(device, n_streams, time_out, silent)
    reasons = {} if failures is None else failures
    records = pool.map(lambda e, ctx: analyzeEntry(e, ctx, reasons, silent), entries)
    for i, why in pool.failures.items():
        reasons.setdefault(entries[i].pdbid, why)
    return {r["pdbid"]: r for r in records if r}


def writeResults(fullResults, outFile="-", outFormat="json"):
    """ref multipleStructures.py:182-194: the merged per-entry records as JSON (indent 2, sorted keys) or as the CSV table
    pdbid + statsHeaders + one diff column per atom type (sorted)."""
    if outFormat in ('csv', 'txt'):
        atomTypes = sorted(densityAnalysis.paramsGlobal["radii"])
        fh = open(outFile, "w", newline='') if outFile != "-" else sys.stdout
        try:
            writer = csv.writer(fh)
            writer.writerow(['pdbid'] + statsHeaders + atomTypes)
            for result in fullResults.values():
                writer.writerow([result['pdbid']] + [result["stats"][h] for h in statsHeaders] + [result["diffs"][t] for t in atomTypes])
        finally:
            if fh is not sys.stdout:
                fh.close()
    else:
        text = json.dumps(fullResults, indent=2, sort_keys=True) + "\n"
        if outFile == "-":
            sys.stdout.write(text)
        else:
            with open(outFile, "w") as fh:
                fh.write(text)
