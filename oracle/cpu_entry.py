"""The per-entry CPU path of `pdb_eda multiple` with the oracle in the device's place.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/test_oracle_cloud.py (the composite pinned on the reference's
goldens), by tests/test_gpu_analysis_big.py (checker for entries of thousands of atoms) and by bench.py's cpu_baseline leg (the
"port" timed on the host cores of the GPU box, where the reference itself cannot travel).  The host-side code around the composite --
flattening the structure, the statistics tail, the record `pdb_eda multiple` keeps -- is the product's own host code: only the
voxel work (what the HIP library does on the GPU) is replaced by oracle/pdbeda_oracle.c.
"""
import numpy as np

from . import oracle as ora


class HostDensity(object):
    """What aggregateCloud reads from its DensityMatrix, with the oracle in the device's place."""

    def __init__(self, header, grid):
        self.header = header
        self._map = ora.Oracle(header, grid)
        mean, std = self._map.mean_std()
        self.meanDensity, self.stdDensity = mean, std
        self.densityCutoff = mean + 1.5 * std            # densityAnalysis.py:131


def analyse(header, grid, structure, params):
    """aggregateCloud of one entry on the CPU: the product's DensityAnalysis host code over the oracle composite."""
    from pdb_eda_amd import densityAnalysis
    densityAnalysis.setGlobals(params)
    an = densityAnalysis.DensityAnalysis("oracle", HostDensity(header, grid), None, structure, None)
    an.aggregateCloud()
    return an


def read_ccp4(path):
    """(header, float32 grid [s][r][c]) of an uncompressed mode-2 CCP4 file (host-side header code of the product: ccp4.py:77-127)."""
    from pdb_eda_amd import ccp4
    with open(path, "rb") as fh:
        head = fh.read(1024)
        header = ccp4.DensityHeader.fromFileHeader(head)
        fh.seek(1024 + int(header.symmetryBytes))
        nc, nr, ns = header.ncrs
        grid = np.frombuffer(fh.read(4 * nc * nr * ns), dtype=header.endian + "f4").astype(np.float32).reshape(ns, nr, nc)
    return header, grid


def multiple_entry(task):
    """One entry of the multiple-structure leg on one core: read the 2Fo-Fc map, aggregateCloud, the numbers `pdb_eda multiple`
    keeps (multipleStructures.py:320-356).  ``task`` = (density path, residues, seed, edge, spacing): the model is rebuilt from its
    seed like the GPU leg's workers do (cached per process)."""
    from pdb_eda_amd import synthetic
    path, n_residues, seed, edge, spacing = task
    loader = synthetic.SyntheticEntryFiles(path, path, n_residues, seed, edge, spacing, as_paths=True)
    st, pdb = loader.structure()
    st.__dict__.pop("_pdbeda_columns", None)
    header, grid = read_ccp4(path)
    an = analyse(header, grid, st, synthetic.synthetic_params())
    if not an.densityElectronRatio:
        return 0
    diffs = {t: float((v - an.densityElectronRatio) / an.densityElectronRatio) for t, v in an.medians["corrected_density_electron_ratio"].items()}
    return {"ratio": float(an.densityElectronRatio), "num_voxels": int(an.numVoxelsAggregated), "diffs": diffs}


def _pool_init():
    import os
    os.environ.setdefault("OMP_NUM_THREADS", "1")


def multiple_baseline(tasks, cores, seconds=20.0):
    """entries/min of ``multiple_entry`` over a multiprocessing.Pool on ``cores`` processes (the reference's own shape:
    multipleStructures.py:167-168 Pool() over all cores): the task list is repeated until about ``seconds`` have passed."""
    import multiprocessing
    import time
    ctx = multiprocessing.get_context("spawn")          # (the caller may hold a GPU: no fork)
    with ctx.Pool(cores, initializer=_pool_init) as pool:
        pool.map(multiple_entry, tasks[:cores])         # untimed: imports, library build / load, first-touch
        done, t0 = 0, time.perf_counter()
        ok = 0
        while done == 0 or time.perf_counter() - t0 < seconds:
            res = pool.map(multiple_entry, tasks, chunksize=1)
            ok += sum(1 for r in res if r)
            done += len(tasks)
        elapsed = time.perf_counter() - t0
    return {"entries": done, "entries_ok": ok, "seconds": elapsed, "entries_per_min": 60.0 * done / elapsed}
