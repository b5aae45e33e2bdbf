"""Electron-density analysis on MI355X behind pdb_eda's ``densityAnalysis`` API surface.

Mirrors ``pdb_eda/densityAnalysis.py`` (``fromFile``/``fromPDBid``/``DensityAnalysis`` with the
reference's attribute, method and result-header names) for the rows of the accelerated path:
blob lists, ``aggregateCloud``, ``calculateAtomSpecificBlobStatistics``, region density /
discrepancy.  All voxel work -- sphere gathers, clustering, unions, regional sums, symmetry
atoms, nearest atom -- runs in ``libpdbeda_hip.so``; what stays on the host is the reference's
own host-side tail: per-atom table bookkeeping and the numpy/scipy statistics over <= nAtoms rows
(densityAnalysis.py:734-767).  Out of scope: downloads, RSCC/RSR, F000.

Parameters (radii, slopes, electrons, bonded atoms) are *reference data*: they are not shipped.
Load the reference's ``conf/optimized_params.json`` (or your own) with :func:`loadParams` /
:func:`setGlobals`, or set ``PDB_EDA_PARAMS``.
"""
import collections
import gzip
import json
import math
import operator
import os

import numpy as np

from . import ccp4
from . import structure as _structure

paramsGlobal = None
radiiGlobal = None
slopesGlobal = None
bondedAtomsGlobal = None
fullAtomNameMapElectronsGlobal = None
fullAtomNameMapAtomTypeGlobal = None
atomTypeLengthGlobal = None

ccp4folder = './ccp4_data/'
pdbfolder = './pdb_data/'


def setGlobals(params):
    """ref densityAnalysis.py:48-68."""
    global paramsGlobal, radiiGlobal, slopesGlobal, bondedAtomsGlobal
    global fullAtomNameMapElectronsGlobal, fullAtomNameMapAtomTypeGlobal, atomTypeLengthGlobal
    paramsGlobal = params
    radiiGlobal = params['radii']
    slopesGlobal = params['slopes']
    bondedAtomsGlobal = params['bonded_atoms']
    fullAtomNameMapElectronsGlobal = params['full_atom_name_map_electrons']
    fullAtomNameMapAtomTypeGlobal = params['full_atom_name_map_atom_type']
    atomTypeLengthGlobal = max(len(t) for t in fullAtomNameMapAtomTypeGlobal.values()) + 5


def loadParams(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, 'rt') as fh:
        setGlobals(json.load(fh))


if os.environ.get("PDB_EDA_PARAMS"):
    loadParams(os.environ["PDB_EDA_PARAMS"])


def _requireParams():
    if paramsGlobal is None:
        raise RuntimeError("no analysis parameters loaded: call pdb_eda_amd.densityAnalysis.loadParams(<optimized_params.json>) "
                           "or set PDB_EDA_PARAMS (the reference's parameter tables are data and are not shipped)")


def fromFile(pdbFile, ccp4DensityFile=None, ccp4DiffDensityFile=None, ctx=None):
    """ref densityAnalysis.py:182-229; returns 0 on any failure, like the reference."""
    pdbid = "xxxx"
    densityObj = None
    diffDensityObj = None
    try:
        if ccp4DensityFile is not None:
            densityObj = ccp4.read(ccp4DensityFile, pdbid, ctx=ctx) if isinstance(ccp4DensityFile, str) else ccp4.parse(ccp4DensityFile, pdbid, ctx=ctx)
            _attachCutoffs(densityObj, None)
        if ccp4DiffDensityFile is not None:
            diffDensityObj = ccp4.read(ccp4DiffDensityFile, pdbid, ctx=ctx) if isinstance(ccp4DiffDensityFile, str) else ccp4.parse(ccp4DiffDensityFile, pdbid, ctx=ctx)
            _attachCutoffs(None, diffDensityObj)
        biopdbObj, pdbObj = _structure.read_pdb(pdbFile, pdbid)
    except Exception:
        return 0
    return DensityAnalysis(pdbid, densityObj, diffDensityObj, biopdbObj, pdbObj)


def fromPDBid(pdbid, ccp4density=True, ccp4diff=True, pdbbio=True, pdbi=True, downloadFile=True, mmcif=False):
    """ref densityAnalysis.py:88-179 for files already present under ./ccp4_data and ./pdb_data
    (there is no network here: nothing is downloaded; a missing file gives 0 like a failed download)."""
    pdbid = pdbid.lower()
    try:
        densityObj = diffDensityObj = biopdbObj = pdbObj = None
        if ccp4density:
            densityObj = ccp4.read(ccp4folder + pdbid + '.ccp4', pdbid)
            _attachCutoffs(densityObj, None)
        if ccp4diff:
            diffDensityObj = ccp4.read(ccp4folder + pdbid + '_diff.ccp4', pdbid)
            _attachCutoffs(None, diffDensityObj)
        if pdbbio or pdbi:
            biopdbObj, pdbObj = _structure.read_pdb(pdbfolder + 'pdb' + pdbid + '.ent.gz', pdbid)
    except Exception:
        return 0
    return DensityAnalysis(pdbid, densityObj, diffDensityObj, biopdbObj, pdbObj)


def _attachCutoffs(densityObj, diffDensityObj):
    """ref densityAnalysis.py:131-132, 148."""
    if densityObj is not None:
        densityObj.densityCutoff = densityObj.meanDensity + 1.5 * densityObj.stdDensity
        densityObj.densityCutoffFromHeader = densityObj.header.densityMean + 1.5 * densityObj.header.rmsd
    if diffDensityObj is not None and getattr(diffDensityObj, "resident", True):
        diffDensityObj.diffDensityCutoff = diffDensityObj.meanDensity + 3 * diffDensityObj.stdDensity
    # (a lazily read Fo-Fc map computes the same value when it is first asked for: ccp4.DensityMatrix.diffDensityCutoff)


def residueAtomName(atom):
    """ref densityAnalysis.py:1243-1252."""
    return atom.parent.resname.strip() + '_' + atom.name


class SymAtom(object):
    """ref cutils.pyx:105-123: an atom with its own symmetry tag and coordinate."""

    def __init__(self, atom, coord, symmetry):
        self.atom = atom
        self.coord = coord
        self.symmetry = symmetry

    def __getattr__(self, attr):
        return getattr(self.atom, attr)


def _typeRegressions(x, y, bfactor, group, n_types):
    """``scipy.stats.linregress(x[sel], y[sel])`` for every atom type at once (ref densityAnalysis.py:752-757 calls it per type:
    five calls were 0.3 ms of a 500-atom entry's 3 ms).  Returns (slope, two-sided p-value, fitted) per type, where ``fitted`` is
    the reference's own condition for fitting at all: more than two rows and not all b-factors equal.  Same formulas as scipy
    (biased covariances about the means, r clipped to [-1, 1], t = r sqrt(df / ((1 - r + TINY)(1 + r + TINY))), p = 2 stdtr(df, -|t|)),
    with the sums taken per type by ``np.bincount``; pinned against scipy itself in tests/test_structure.py."""
    from scipy import special
    n = np.bincount(group, minlength=n_types).astype(np.float64)
    safe_n = np.maximum(n, 1.0)
    xm = np.bincount(group, weights=x, minlength=n_types) / safe_n
    ym = np.bincount(group, weights=y, minlength=n_types) / safe_n
    dx, dy = x - xm[group], y - ym[group]
    ssxm = np.bincount(group, weights=dx * dx, minlength=n_types) / safe_n
    ssym = np.bincount(group, weights=dy * dy, minlength=n_types) / safe_n
    ssxym = np.bincount(group, weights=dx * dy, minlength=n_types) / safe_n
    order = np.argsort(group, kind="stable")
    first = np.searchsorted(group[order], np.arange(n_types))
    present = n > 0
    b_sorted = bfactor[order]
    # the reference's test is `len(np.unique(bfactor)) == 1` (densityAnalysis.py:752), and np.unique counts all NaNs as ONE value:
    # a type whose b-factors are all NaN (no positive B: its median is NaN, and so is every normalised value) is NOT fitted
    b_lo = np.where(present, np.fmin.reduceat(b_sorted, np.minimum(first, max(len(b_sorted) - 1, 0))), 0.0) if len(b_sorted) else np.zeros(n_types)
    b_hi = np.where(present, np.fmax.reduceat(b_sorted, np.minimum(first, max(len(b_sorted) - 1, 0))), 0.0) if len(b_sorted) else np.zeros(n_types)
    n_nan = np.bincount(group, weights=np.isnan(bfactor), minlength=n_types)
    one_value = (n_nan == n) | ((n_nan == 0) & (b_lo == b_hi))
    fitted = (n > 2) & ~one_value
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where((ssxm == 0.0) | (ssym == 0.0), 0.0, ssxym / np.sqrt(ssxm * ssym))
        r = np.clip(r, -1.0, 1.0)
        slope = ssxym / ssxm
        df = n - 2.0
        t = r * np.sqrt(df / ((1.0 - r + 1.0e-20) * (1.0 + r + 1.0e-20)))
        p = 2.0 * special.stdtr(np.maximum(df, 1.0), -np.abs(t))
    return slope, p, fitted


def _lastWithSameCoord(coord32):
    """For every row of float32 coordinates, the index of the LAST row with an equal triple (``allAtomClouds[tuple(atom.coord)]``
    keeps the last atom of a coordinate, ref densityAnalysis.py:604; equal as floats: -0.0 is 0.0).  One 64-bit sort of a mixed
    key instead of a sort of 12-byte records; rows whose keys collide without being equal send the call to the slow exact path."""
    n = len(coord32)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    bits = np.ascontiguousarray(coord32 + np.float32(0.0)).view(np.uint32).astype(np.uint64)     # (+0.0: -0.0 becomes 0.0)
    key = ((bits[:, 0] << np.uint64(32)) | bits[:, 1]) * np.uint64(0x9E3779B97F4A7C15) ^ (bits[:, 2] * np.uint64(0xC2B2AE3D27D4EB4F))
    order = np.argsort(key, kind="stable")
    sorted_bits, sorted_key = bits[order], key[order]
    same_row = (sorted_bits[1:] == sorted_bits[:-1]).all(axis=1)
    if ((sorted_key[1:] == sorted_key[:-1]) & ~same_row).any():
        triple = np.ascontiguousarray(bits.astype(np.uint32)).view(np.dtype((np.void, 12))).ravel()
        _, same = np.unique(triple, return_inverse=True)
        last = np.full(int(same.max()) + 1, -1, dtype=np.int64)
        np.maximum.at(last, same.reshape(-1), np.arange(n))
        return last[same.reshape(-1)]
    starts = np.flatnonzero(np.concatenate([[True], ~same_row]))
    group = np.cumsum(np.concatenate([[0], (~same_row).astype(np.int64)]))
    alias = np.empty(n, dtype=np.int64)
    alias[order] = np.maximum.reduceat(order, starts)[group]
    return alias


_pairTableCache = collections.OrderedDict()


def _pairTables(names, n_pairs):
    """What the parameter tables say about a structure's distinct 'RES_ATOM' names (``structure.Columns.pair_names``): atom type,
    electrons, and the bonded names that exist among them (densityAnalysis.py:617-621, 653-656), as arrays over the names.
    Entries of one run share their names (the same residues with the same atoms, met in the same order) and the tables are the
    same objects from entry to entry, so the answer is kept per (names, tables): a few dict look-ups per entry instead of a
    Python loop over ~160 names and their bonded lists."""
    typeMap, electronsMap, bonded = fullAtomNameMapAtomTypeGlobal, fullAtomNameMapElectronsGlobal, bondedAtomsGlobal
    key = (tuple(names), id(typeMap), id(electronsMap), id(bonded))
    hit = _pairTableCache.get(key)
    if hit is not None and hit["tables"][0] is typeMap and hit["tables"][1] is electronsMap and hit["tables"][2] is bonded:
        _pairTableCache.move_to_end(key)
        return hit
    known = np.fromiter((name in typeMap for name in names), dtype=bool, count=len(names))
    pair_id = {name: k for k, name in enumerate(names)}
    pair_type = [typeMap[name] if name in typeMap else None for name in names] + [None] * (n_pairs - len(names))
    electrons = np.full(n_pairs, np.nan)
    nb_off = np.zeros(n_pairs + 1, dtype=np.int64)
    nb = []
    for k, name in enumerate(names):
        if name in electronsMap:
            electrons[k] = electronsMap[name]
        nb.extend(pair_id[other] for other in bonded.get(name, ()) if other in pair_id)
        nb_off[k + 1] = len(nb)
    nb_off[len(names) + 1:] = len(nb)
    type_names = sorted({t for t in pair_type if t is not None})
    type_id = {t: k for k, t in enumerate(type_names)}
    pair_type_id = np.asarray([type_id[t] if t is not None else -1 for t in pair_type], dtype=np.int64)
    hit = {"tables": (typeMap, electronsMap, bonded), "known": known, "pair_type": pair_type, "pair_type_id": pair_type_id, "type_names": type_names,
           "electrons": electrons, "nb_off": nb_off, "nb": np.asarray(nb, dtype=np.int64)}
    _pairTableCache[key] = hit
    while len(_pairTableCache) > 32:
        _pairTableCache.popitem(last=False)
    return hit


class _SymAtomList(object):
    """The list of SymAtom objects createSymmetryAtoms returns (cutils.pyx:73-103), materialised item by item; the tables that
    list thousands of them read whole columns instead (``columns``)."""

    def __init__(self, cols, idx, sym, xyz, ident, rows=None):
        self._cols, self._idx, self._sym, self._xyz, self._ident = cols, idx, sym, xyz, ident
        self._rows = np.arange(len(idx)) if rows is None else np.asarray(rows)
        self._made = {}

    def subset(self, rows):
        return _SymAtomList(self._cols, self._idx, self._sym, self._xyz, self._ident, self._rows[rows])

    def __len__(self):
        return len(self._rows)

    def columns(self, items=None, type=""):
        """(rows of the structure columns, symmetry operators as an n x 4 int64 array -- a tuple per row once ``_rows`` has made the table --,
        coordinates as the SymAtom objects would hold them) of the listed
        items (all by default), optionally only those whose atom name is ``type``."""
        r = self._rows if items is None else self._rows[np.asarray(items, dtype=np.int64)]
        atom_rows = self._idx[r]
        if type:
            keep = np.asarray(self._cols.name)[atom_rows] == type if len(r) else np.zeros(0, dtype=bool)
            r, atom_rows = r[keep], atom_rows[keep]
        symmetry = np.ascontiguousarray(self._sym[r], dtype=np.int64)      # (n x 4 ints: _rows makes the tuples)
        own = self._cols.atom_lists("coord")                               # (most listed atoms are the structure's own: their coordinate objects as they are)
        picked = atom_rows.tolist()
        coords = list(operator.itemgetter(*picked)(own)) if len(picked) > 1 else [own[a] for a in picked]
        moved = np.nonzero(~self._ident[r])[0]
        if len(moved):
            xyz = self._xyz[r[moved]]
            for k, row in zip(moved.tolist(), xyz):                          # (a row view per symmetry copy only: iterating every row was 0.2 ms of a 1 400-blob table)
                coords[k] = row
        return atom_rows, symmetry, coords

    def _make(self, r):
        atom = self._cols.atoms[int(self._idx[r])]
        return SymAtom(atom, atom.coord if self._ident[r] else self._xyz[r], tuple(int(v) for v in self._sym[r]))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        r = int(self._rows[i])
        if r not in self._made:
            self._made[r] = self._make(r)
        return self._made[r]

    def __iter__(self):
        return (self[k] for k in range(len(self)))


def _segmentMeans(values, counts):
    """``np.mean`` of consecutive segments of ``values`` (``counts`` items each), bit-equal to calling it per segment: rows of
    equal length are reduced together along their contiguous axis, which is the summation numpy uses for a 1-D array."""
    counts = np.asarray(counts, dtype=np.int64)
    start = np.cumsum(counts) - counts
    out = np.full(len(counts), np.nan)
    for k in np.unique(counts).tolist():
        if k == 0:
            continue
        which = np.nonzero(counts == k)[0]
        out[which] = np.add.reduce(values[start[which][:, None] + np.arange(k)[None, :]], axis=1) / k
    return out


def _norm3(x):
    """np.linalg.norm of a 1-D float64 vector, as numpy computes it (sqrt(x.dot(x))), without the dispatch overhead."""
    return math.sqrt(float(x.dot(x)))


def _crs_keys(crs):
    """Pack raw (c, r, s) triples into sortable int64 keys (21 bits each, offset)."""
    c = np.asarray(crs, dtype=np.int64) + (1 << 20)
    return (c[:, 0] << 42) | (c[:, 1] << 21) | c[:, 2]


class DensityAnalysis(object):
    """ref densityAnalysis.py:278-1240 (the accelerated subset)."""

    residueCloudHeader = ['chain', 'residue_number', 'residue_name', 'local_density_electron_ratio', 'num_voxels', 'electrons', 'volume', 'centroid_xyz']
    domainCloudHeader = residueCloudHeader
    blobStatisticsHeader = ['distance_to_atom', 'sign', 'electrons_of_discrepancy', 'num_voxels', 'volume', 'chain', 'residue_number', 'residue_name',
                            'atom_name', 'atom_symmetry', 'atom_xyz', 'centroid_xyz']
    regionDensityHeader = ["actual_significant_regional_density", "num_electrons_actual_significant_regional_density"]
    atomRegionDensityHeader = ['model', 'chain', 'residue_number', 'residue_name', "atom_name", "occupancy"] + regionDensityHeader
    symmetryAtomRegionDensityHeader = ['model', 'chain', 'residue_number', 'residue_name', "atom_name", "symmetry", "atom_xyz", "fully_within_density_map"] + regionDensityHeader
    residueRegionDensityHeader = ['model', 'chain', 'residue_number', 'residue_name', "mean_occupancy"] + regionDensityHeader
    regionDiscrepancyHeader = ["actual_abs_significant_regional_discrepancy", "num_electrons_actual_abs_significant_regional_discrepancy",
                               "expected_abs_significant_regional_discrepancy", "num_electrons_expected_abs_significant_regional_discrepancy",
                               "actual_significant_regional_discrepancy", "num_electrons_actual_significant_regional_discrepancy",
                               "actual_positive_significant_regional_discrepancy", "num_electrons_actual_positive_significant_regional_discrepancy",
                               "actual_negative_significant_regional_discrepancy", "num_electrons_actual_negative_significant_regional_discrepancy"]
    atomRegionDiscrepancyHeader = ['model', 'chain', 'residue_number', 'residue_name', "atom_name", "occupancy"] + regionDiscrepancyHeader
    symmetryAtomRegionDiscrepancyHeader = ['model', 'chain', 'residue_number', 'residue_name', "atom_name", "symmetry", "atom_xyz", "fully_within_density_map"] + regionDiscrepancyHeader
    residueRegionDiscrepancyHeader = ['model', 'chain', 'residue_number', 'residue_name', "mean_occupancy"] + regionDiscrepancyHeader

    def __init__(self, pdbid, densityObj=None, diffDensityObj=None, biopdbObj=None, pdbObj=None):
        self.pdbid = pdbid
        self.densityObj = densityObj
        self.diffDensityObj = diffDensityObj
        self.biopdbObj = biopdbObj
        self.pdbObj = pdbObj
        self._symmetryAtoms = None
        self._symmetryOnlyAtoms = None
        self._asymmetryAtoms = None
        self._symmetryAtomCoords = None
        self._symmetryOnlyAtomCoords = None
        self._asymmetryAtomCoords = None
        self._greenBlobList = None
        self._redBlobList = None
        self._blueBlobList = None
        self._medians = None
        self._atomCloudDescriptions = None
        self._residueCloudDescriptions = None
        self._domainCloudDescriptions = None
        self._densityElectronRatio = None
        self._numVoxelsAggregated = None
        self._totalAggregatedElectrons = None
        self._totalAggregatedDensity = None
        self._atomTypeOverlapCompleteness = None
        self._atomTypeOverlapIncompleteness = None
        self._cloudTables = None
        self._fc = None

    # ---- lazy properties (ref densityAnalysis.py:326-565) ------------------------------------
    def _lazy(name, trigger):
        def getter(self):
            if getattr(self, name) is None:
                getattr(self, trigger)()
            return getattr(self, name)
        return property(getter)

    symmetryAtoms = _lazy('_symmetryAtoms', '_calculateSymmetryAtoms')
    symmetryOnlyAtoms = _lazy('_symmetryOnlyAtoms', '_splitSymmetryAtoms')
    asymmetryAtoms = _lazy('_asymmetryAtoms', '_splitSymmetryAtoms')
    symmetryAtomCoords = _lazy('_symmetryAtomCoords', '_calculateSymmetryAtoms')
    symmetryOnlyAtomCoords = _lazy('_symmetryOnlyAtomCoords', '_splitSymmetryAtoms')
    asymmetryAtomCoords = _lazy('_asymmetryAtomCoords', '_splitSymmetryAtoms')
    medians = _lazy('_medians', 'aggregateCloud')

    def _described(name, which):
        """The description tables (ref densityAnalysis.py:682-731, 734-767) are MADE on first access: aggregateCloud keeps what they
        are made from (numeric columns, row indices) -- `pdb_eda multiple` and the optimiser read the medians and the number of rows,
        never the rows (a 2 000-atom entry spent 0.4 ms of its 1.6 on string columns and lists of Python rows nobody looked at)."""
        def getter(self):
            if getattr(self, name) is None:
                if self._cloudTables is None:
                    self.aggregateCloud()
                if self._cloudTables is not None and getattr(self, name) is None:
                    setattr(self, name, self._cloudTables[which]())
            return getattr(self, name)
        return property(getter)

    atomCloudDescriptions = _described('_atomCloudDescriptions', 'atoms')
    residueCloudDescriptions = _described('_residueCloudDescriptions', 'residues')
    domainCloudDescriptions = _described('_domainCloudDescriptions', 'domains')
    del _described

    @property
    def cloudCounts(self):
        """(atoms, residue clouds, domain clouds) analysed = the lengths of the three description tables, without making them."""
        if self._cloudTables is None:
            self.aggregateCloud()
        return self._cloudTables["counts"] if self._cloudTables is not None else None
    numVoxelsAggregated = _lazy('_numVoxelsAggregated', 'aggregateCloud')
    totalAggregatedElectrons = _lazy('_totalAggregatedElectrons', 'aggregateCloud')
    totalAggregatedDensity = _lazy('_totalAggregatedDensity', 'aggregateCloud')
    densityElectronRatio = _lazy('_densityElectronRatio', 'aggregateCloud')
    atomTypeOverlapCompleteness = _lazy('_atomTypeOverlapCompleteness', 'aggregateCloud')
    atomTypeOverlapIncompleteness = _lazy('_atomTypeOverlapIncompleteness', 'aggregateCloud')
    del _lazy

    def _greenRed(self):
        # ONE fused pass over the Fo-Fc grid gives both lists (the reference thresholds it twice)
        cut = self.diffDensityObj.diffDensityCutoff
        self._greenBlobList, self._redBlobList = self.diffDensityObj.createFullBlobLists(cut)

    @property
    def greenBlobList(self):
        """ref densityAnalysis.py:392-401."""
        if self._greenBlobList is None:
            self._greenRed()
        return self._greenBlobList

    @property
    def redBlobList(self):
        """ref densityAnalysis.py:403-412."""
        if self._redBlobList is None:
            self._greenRed()
        return self._redBlobList

    @property
    def blueBlobList(self):
        """ref densityAnalysis.py:414-423."""
        if self._blueBlobList is None:
            self._blueBlobList = self.densityObj.createFullBlobList(self.densityObj.densityCutoff)
        return self._blueBlobList

    # ---- aggregateCloud (ref densityAnalysis.py:571-780) --------------------------------------
    def _cloudInputs(self):
        """The arrays of ``pdbeda_cloud_atoms`` for the current parameter tables.  Everything but the radius column depends on
        the structure and on the name -> type / electrons / bonded tables only, so it is kept on the structure's snapshot
        while those tables are the same objects: the iterations of optimise mode (optimizeParams.py:232-243 passes
        ``{**params, "radii": ..., "slopes": ...}``) change radii and slopes, nothing else."""
        cols = _structure.columns(self.biopdbObj)
        tables = (fullAtomNameMapAtomTypeGlobal, fullAtomNameMapElectronsGlobal, bondedAtomsGlobal)
        cached = cols.__dict__.get("_cloud_inputs")
        if cached is None or any(a is not b for a, b in zip(cached[0], tables)):
            cached = (tables, self._cloudInputsFixed(cols))            # (holding the tables keeps their identity meaningful)
            cols._cloud_inputs = cached
        inp = dict(cached[1])
        pair_radius = np.zeros(len(inp["pair_type"]), dtype=np.float32)
        for k in inp["used_pairs"]:
            pair_radius[k] = radiiGlobal[inp["pair_type"][k]]
        inp["radius"] = pair_radius[inp["pair"]]
        return inp

    @staticmethod
    def _cloudInputsFixed(cols, native=None):
        """Flatten what aggregateCloud reads from the structure (densityAnalysis.py:596-603, 617-621, 653-656) into the arrays
        of ``pdbeda_cloud_atoms``: the eligible atoms in the reference's iteration order, a key per (residue, residue_atom
        name), the bonded-name table restricted to each residue, and the 'owners' of the completeness count.  Works on the
        columnar snapshot of the structure (``structure.columns``): no per-atom Python.  The index work is one pass in C
        (``_hostwalk.cloud_inputs``: fifty small numpy calls were 0.3 ms of a 2 000-atom entry) with the numpy form below as
        its fallback and its check (``native``: None = C when built, True / False force one; tests/test_cloud_inputs.py)."""
        names = cols.pair_names
        walk = _structure._hostwalk() if native is not False else None
        if walk is None and native:
            raise ImportError("pdb_eda_amd/_hostwalk.so is not built (python __graft_entry__.py)")
        if walk is not None and hasattr(walk, "cloud_inputs"):
            n_pairs = max(len(names), 1)
            tables = _pairTables(names, n_pairs)
            known = np.zeros(n_pairs, dtype=np.uint8)
            known[:len(names)] = tables["known"]
            out = walk.cloud_inputs(cols.res_of_atom, cols.pair_of_atom, (~cols.res_het).astype(np.uint8), known, cols.occupancy,
                                    np.ascontiguousarray(cols.coord32), tables["nb_off"], tables["nb"])
            sel, residue_of, pair_of, key_of, alias, bonded_off, bonded, owner_key, owner_pair, plain_residues = (
                np.frombuffer(b, dtype=t) for b, t in zip(out, (np.int64, np.int32, np.int64, np.int32, np.int32, np.int64, np.int32, np.int32, np.int64, np.int64)))
            used = np.unique(pair_of)
            electrons = tables["electrons"]
            if len(used) and np.isnan(electrons[used]).any():     # (the reference's electronsMap[name] raises the same KeyError)
                raise KeyError(names[int(used[np.isnan(electrons[used])][0])])
            return {"cols": cols, "rows": sel, "plain_residues": plain_residues, "xyz": cols.coord[sel], "occupancy": cols.occupancy[sel],
                    "electrons": electrons[pair_of], "used_pairs": used.tolist(), "pair": pair_of, "pair_type": tables["pair_type"],
                    "residue": residue_of, "alias": alias, "key": key_of, "bonded_off": bonded_off, "bonded": bonded, "owner_key": owner_key,
                    "owner_type_id": tables["pair_type_id"][owner_pair], "type_names": tables["type_names"], "pair_type_id": tables["pair_type_id"]}
        n_pairs = max(len(names), 1)
        known = _pairTables(names, n_pairs)["known"]
        plain = ~cols.res_het                                                      # residues with id[0] == ' ' (596)
        ordinal = np.cumsum(plain) - 1                                             # their running number
        child = np.nonzero(plain[cols.res_of_atom])[0]                             # EVERY child atom of those residues, in order
        child_res = ordinal[cols.res_of_atom[child]]
        child_pair = cols.pair_of_atom[child]
        eligible = known[child_pair] & (cols.occupancy[child] != 0) if len(child) else np.zeros(0, dtype=bool)
        sel = child[eligible]                                                      # rows of cols.* of the eligible atoms
        residue_of, pair_of = child_res[eligible], child_pair[eligible]
        n = len(sel)
        # key per (residue, name), numbered by first appearance
        code = residue_of * n_pairs + pair_of
        distinct, first, inverse = np.unique(code, return_index=True, return_inverse=True)
        by_first = np.argsort(first, kind="stable")
        rank = np.empty(len(distinct), dtype=np.int64)
        rank[by_first] = np.arange(len(distinct))
        key_of = rank[inverse]

        def key_lookup(codes):
            pos = np.minimum(np.searchsorted(distinct, codes), max(len(distinct) - 1, 0))
            found = distinct[pos] == codes if len(distinct) else np.zeros(len(codes), dtype=bool)
            return found, rank[pos[found]]
        # allAtomClouds is keyed by the coordinate: the last atom with the same float32 triple wins (604)
        alias = _lastWithSameCoord(cols.coord32[sel])
        # the pair tables: type, electrons, radius, and the bonded names that exist in this structure
        used = np.unique(pair_of)
        tables = _pairTables(names, n_pairs)
        pair_type, pair_type_id, type_names, nb_off, nb = tables["pair_type"], tables["pair_type_id"], tables["type_names"], tables["nb_off"], tables["nb"]
        pair_electrons = tables["electrons"]
        if len(used) and np.isnan(pair_electrons[used]).any():     # (the reference's electronsMap[name] raises the same KeyError)
            raise KeyError(names[int(used[np.isnan(pair_electrons[used])][0])])
        # bonded keys of every key, in key order then table order
        key_code = distinct[by_first]
        key_res, key_pair = key_code // n_pairs, key_code % n_pairs
        counts = nb_off[key_pair + 1] - nb_off[key_pair]
        owner = np.repeat(np.arange(len(key_code)), counts)
        within = np.arange(int(counts.sum())) - np.repeat(np.cumsum(counts) - counts, counts)
        found, bonded = key_lookup(key_res[owner] * n_pairs + nb[nb_off[key_pair][owner] + within]) if len(owner) else (np.zeros(0, dtype=bool), np.zeros(0, dtype=np.int64))
        bonded_off = np.concatenate([[0], np.cumsum(np.bincount(owner[found], minlength=len(key_code)))]).astype(np.int64)
        # owners of the completeness count: every child atom whose (residue, name) has a key (653-656)
        found, owner_key = key_lookup(child_res * n_pairs + child_pair)
        return {"cols": cols, "rows": sel, "plain_residues": np.nonzero(plain)[0], "xyz": cols.coord[sel], "occupancy": cols.occupancy[sel],
                "electrons": pair_electrons[pair_of], "used_pairs": used.tolist(), "pair": pair_of, "pair_type": pair_type,
                "residue": residue_of.astype(np.int32), "alias": alias.astype(np.int32), "key": key_of.astype(np.int32),
                "bonded_off": bonded_off, "bonded": bonded.astype(np.int32), "owner_key": owner_key.astype(np.int32),
                "owner_type_id": pair_type_id[child_pair[found]], "type_names": type_names, "pair_type_id": pair_type_id}

    def aggregateCloud(self, minCloudElectrons=25.0, minTotalElectrons=400.0):
        """Aggregate the 2Fo-Fc clouds by atom, residue and domain; sets ``densityElectronRatio``,
        ``medians`` and the description tables.  Same silent-failure contract as the reference
        (Q7): everything stays ``None`` below ``minTotalElectrons`` or if the statistics tail fails.

        Everything that touches voxels -- the clouds of every atom, best cloud and pooling, bonded-atom completeness, the
        residue and domain unions -- is ONE library call (``pdbeda_aggregate_cloud``: voxel lists stay on the device); the
        host flattens the structure into arrays before it and keeps the reference's host-side statistics tail after it."""
        _requireParams()
        densityObj = self.densityObj
        unitVolume = densityObj.header.unitVolume
        typeMap = fullAtomNameMapAtomTypeGlobal
        inp = self._cloudInputs()
        if not len(inp["rows"]):
            return
        res = densityObj._map.aggregate_cloud(inp["xyz"], inp["radius"], inp["electrons"] * inp["occupancy"], inp["residue"], inp["alias"], inp["key"],
                                              inp["bonded_off"], inp["bonded"], inp["owner_key"], densityObj.densityCutoff, minCloudElectrons)
        if not len(res["atom"]):
            return
        completely = collections.defaultdict(int)
        incompletely = collections.defaultdict(int)
        n_types = len(inp["type_names"])
        for counter, state in ((completely, 1), (incompletely, 2)):
            per_type = np.bincount(inp["owner_type_id"][res["owner_state"] == state], minlength=n_types).tolist()
            counter.update({t: c for t, c in zip(inp["type_names"], per_type) if c})
        cols = inp["cols"]
        plain = inp["plain_residues"].tolist()

        def cloudRows(t, by_ratio=False):
            which = [plain[ri] for ri in t["residue"].tolist()]
            rows = [[cols.res_chain[k], cols.res_number[k], cols.res_name[k], tot / el, nv, el, nv * unitVolume, cen]
                    for k, tot, nv, el, cen in zip(which, t["total"].tolist(), t["n"].tolist(), t["electrons"].tolist(), t["centroid"].tolist())]
            if by_ratio:
                rows.sort(key=lambda x: x[3])
            return rows
        numVoxels, totalElectrons, totalDensity = res["numVoxels"], res["totalElectrons"], res["totalDensity"]
        if totalElectrons < minTotalElectrons:
            return
        densityElectronRatio = totalDensity / totalElectrons

        try:
            atoms, n_atoms, medians = self._cloudStatistics(inp, res, densityElectronRatio, unitVolume, typeMap)
        except Exception:
            return
        self._cloudTables = {"atoms": atoms, "residues": lambda: cloudRows(res["res"]), "domains": lambda: cloudRows(res["dom"], True),
                             "counts": (n_atoms, len(res["res"]["n"]), len(res["dom"]["n"]))}
        self._atomCloudDescriptions = self._residueCloudDescriptions = self._domainCloudDescriptions = None

        self._densityElectronRatio = densityElectronRatio
        self._numVoxelsAggregated = numVoxels
        self._totalAggregatedElectrons = totalElectrons
        self._totalAggregatedDensity = totalDensity
        self._medians = medians
        self._atomTypeOverlapCompleteness = completely
        self._atomTypeOverlapIncompleteness = incompletely

    @staticmethod
    def _cloudStatistics(inp, res, ratio, unitVolume, typeMap, native=None):
        """The host-side statistics over the atom table (what densityAnalysis.py:734-767 computes), on whole columns: the
        per-atom-type medians come from ONE sort per column instead of a masked nanmedian per (column, type)."""
        idx = res["atom"]
        dist = res["atom_distance"]
        total, count, centroid = res["atom_total"], res["atom_n"], res["atom_centroid"]
        if not np.isnan(dist).all():
            walk0 = _structure._hostwalk() if native is not False else None
            if walk0 is not None and hasattr(walk0, "nan_cutoff"):                # (numpy's own nanmedian + nanstd * 2, to the bit, without their 0.08 ms of Python)
                near = dist < walk0.nan_cutoff(np.ascontiguousarray(dist, dtype=np.float64), 2.0)
            else:
                near = dist < np.nanmedian(dist) + np.nanstd(dist) * 2          # (the reference filters the finished table: same rows)
            idx, dist, total, count, centroid = idx[near], dist[near], total[near], count[near], centroid[near]
        n = len(idx)
        cols, rows = inp["cols"], inp["rows"][idx]
        of_residue = cols.res_of_atom[rows]
        # atom types as numbers: inp["type_names"] is sorted, so the types present, in np.unique's order, are the ids present
        type_id = inp["pair_type_id"][inp["pair"][idx]]
        present = np.flatnonzero(np.bincount(type_id, minlength=len(inp["type_names"])))
        atom_types = np.asarray(inp["type_names"])[present] if len(present) else np.zeros(0, dtype='U1')
        n_types = len(atom_types)
        group = np.searchsorted(present, type_id)
        # the numeric columns of the atom table (the reference's structured array, 734-741); the table itself -- with its string
        # columns -- is made from them when somebody asks for it (``make_table`` below)
        table = {'density_electron_ratio': total / inp["electrons"][idx] / inp["occupancy"][idx], 'num_voxels': np.asarray(count, dtype=np.int64),
                 'bfactor': np.array(cols.bfactor[rows], dtype=np.float64), 'centroid_distance': np.asarray(dist, dtype=np.float64)}

        def make_table():
            out = np.zeros(n, dtype=np.dtype([
                ('chain', 'U20'), ('residue_number', int), ('residue_name', 'U10'), ('atom_name', 'U10'), ('atom_type', 'U%d' % atomTypeLengthGlobal),
                ('density_electron_ratio', float), ('num_voxels', int), ('electrons', int), ('bfactor', float), ('centroid_distance', float),
                ('centroid_xyz', float, (3,)), ('adj_density_electron_ratio', float), ('domain_fraction', float), ('corrected_fraction', float),
                ('corrected_density_electron_ratio', float), ('volume', float)]))
            out['chain'] = np.asarray(cols.res_chain)[of_residue]
            out['residue_number'] = np.asarray(cols.res_number)[of_residue]
            out['residue_name'] = np.asarray(cols.res_name)[of_residue]
            out['atom_name'] = np.asarray(cols.atom_names)[cols.name_of_atom[rows]] if n else ''
            out['atom_type'] = atom_types[group] if n else ''
            out['electrons'] = inp["electrons"][idx]
            out['centroid_xyz'] = centroid
            for field, values in table.items():
                out[field] = values
            return out

        asDict = lambda per_type: dict(zip(atom_types.tolist(), per_type))     # noqa: E731
        walk = _structure._hostwalk() if native is not False else None
        if walk is None and native:
            raise ImportError("pdb_eda_amd/_hostwalk.so is not built (python __graft_entry__.py)")
        if walk is not None and hasattr(walk, "cloud_stats") and n:
            # the medians, the b-factor regressions and the corrected columns in one pass in C (round 5: ~60 small numpy calls were
            # 0.45 ms of a 2 000-atom entry's 1.3 ms); the numpy form below stays as its fallback and its check (tests/test_structure.py)
            table_slopes = np.array([slopesGlobal[t] for t in atom_types.tolist()], dtype=np.float64)
            rows_b, types_b = walk.cloud_stats(np.ascontiguousarray(group, dtype=np.int64), n_types, np.ascontiguousarray(table['density_electron_ratio'], dtype=np.float64),
                                               table['num_voxels'], table['bfactor'], table['centroid_distance'], table_slopes, float(ratio), float(unitVolume))
            r6 = np.frombuffer(rows_b, dtype=np.float64).reshape(6, n)
            t10 = np.frombuffer(types_b, dtype=np.float64).reshape(10, n_types)
            for k, field in enumerate(('adj_density_electron_ratio', 'bfactor', 'domain_fraction', 'corrected_fraction', 'corrected_density_electron_ratio', 'volume')):
                table[field] = r6[k]
            per_type = dict(zip(('num_voxels', 'volume', 'density_electron_ratio', 'centroid_distance', 'adj_density_electron_ratio', 'bfactor',
                                 'domain_fraction', 'slopes', 'corrected_fraction', 'corrected_density_electron_ratio'), t10))
            medians = {field: asDict(per_type[field]) for field in ('num_voxels', 'density_electron_ratio', 'centroid_distance', 'adj_density_electron_ratio', 'volume',
                                                                    'bfactor', 'slopes', 'domain_fraction', 'corrected_fraction', 'corrected_density_electron_ratio')}
            return make_table, n, medians

        # rows in type order, once: every median below sorts the values of one type at a time, in place
        by_type = np.argsort(group.astype(np.int16 if n_types < 32768 else np.int64), kind="stable")
        start = np.searchsorted(group[by_type], np.arange(n_types + 1))
        runs = list(zip(start[:-1].tolist(), start[1:].tolist()))

        def typeMedians(values, keep=None, also=()):
            """np.nanmedian of ``values`` per atom type (NaNs and dropped rows sort last within a type; the middle one or the mean of
            the middle two of what is left).  ``also``: weakly monotone functions of the values whose per-type medians are wanted as
            well -- the median of f(values) is the mean of f at the same one or two order statistics, so no second sort."""
            v = np.asarray(values, dtype=np.float64)
            if keep is not None:
                v = np.where(keep, v, np.nan)
            sv = v[by_type]
            for lo, hi in runs:
                sv[lo:hi].sort()
            count = np.bincount(group, weights=~np.isnan(v), minlength=n_types).astype(np.int64)
            safe = np.minimum(np.stack([start[:-1] + np.maximum(count - 1, 0) // 2, start[:-1] + count // 2]), max(n - 1, 0))
            middle = sv[safe] if n else np.full((2, n_types), np.nan)
            out = [np.where(count > 0, (f(middle[0]) + f(middle[1])) / 2.0, np.nan) for f in (lambda x: x,) + tuple(also)]
            return out[0] if not also else out

        medians = {}
        m_vox, m_volume = typeMedians(table['num_voxels'], also=(lambda nv: nv * unitVolume,))
        medians['num_voxels'] = asDict(m_vox)
        table['adj_density_electron_ratio'] = table['density_electron_ratio'] / table['num_voxels'] * m_vox[group]
        table['volume'] = table['num_voxels'] * unitVolume
        medians['density_electron_ratio'] = asDict(typeMedians(table['density_electron_ratio']))
        medians['centroid_distance'] = asDict(typeMedians(table['centroid_distance']))
        m_adj, m_fraction = typeMedians(table['adj_density_electron_ratio'], also=(lambda adj: (adj - ratio) / ratio,))
        medians['adj_density_electron_ratio'] = asDict(m_adj)
        medians['volume'] = asDict(m_volume)
        m_b = typeMedians(table['bfactor'], table['bfactor'] > 0)
        medians['bfactor'] = asDict(m_b)
        missing = table['bfactor'] <= 0
        table['bfactor'][missing] = m_b[group][missing]
        # slope of the b-factor dependence per atom type: linear regression where there is something to fit, else the table's slope
        fraction = (table['adj_density_electron_ratio'] - ratio) / ratio
        log_b = np.log(table['bfactor'])
        fit_slope, fit_p, fitted = _typeRegressions(log_b, fraction, table['bfactor'], group, n_types)
        table_slopes = np.array([slopesGlobal[t] for t in atom_types.tolist()], dtype=np.float64)
        slopes = np.where(fitted & ~(fit_p > 0.05), fit_slope, table_slopes)
        medians['slopes'] = asDict(slopes)
        table['domain_fraction'] = fraction
        table['corrected_fraction'] = fraction - (log_b - np.log(m_b[group])) * slopes[group]
        table['corrected_density_electron_ratio'] = table['corrected_fraction'] * ratio + ratio
        medians['domain_fraction'] = asDict(m_fraction)
        m_corrected, m_corrected_ratio = typeMedians(table['corrected_fraction'], also=(lambda c: c * ratio + ratio,))
        medians['corrected_fraction'] = asDict(m_corrected)
        medians['corrected_density_electron_ratio'] = asDict(m_corrected_ratio)
        return make_table, n, medians

    # ---- symmetry atoms (ref densityAnalysis.py:885-912 + cutils.pyx:73-103) -------------------
    def _calculateSymmetryAtoms(self):
        densityObj = self.densityObj
        header = densityObj.header
        ncrs = header.ncrs
        corners = np.array([[c, r, s] for c in [0, ncrs[0] - 1] for r in [0, ncrs[1] - 1] for s in [0, ncrs[2] - 1]], dtype=np.int32)
        box = densityObj._map.crs2xyz(corners)
        lo, hi = box.min(axis=0), box.max(axis=0)
        cols = _structure.columns(self.biopdbObj)
        coords = cols.coord
        rot = np.array([np.asarray(m, dtype=np.float64) for m in self.pdbObj.header.rotationMats])
        idx, sym, xyz = densityObj._ctx.symmetry_atoms(coords, rot, np.asarray(header.orthoMat, dtype=np.float64), lo, hi)
        ident = ~np.any(sym != 0, axis=1)
        allCoords = np.where(ident[:, None], coords[idx], xyz)      # == np.asarray([atom.coord ...]): float32 coordinates promote exactly
        # the SymAtom objects are made on demand: a blob-statistics table touches a few hundred of the thousands there are
        allAtoms = _SymAtomList(cols, idx, sym, xyz, ident)
        self._symmetryAtoms = allAtoms
        self._symmetryAtomCoords = allCoords
        self._symmetryOnlyAtoms = self._asymmetryAtoms = self._symmetryOnlyAtomCoords = self._asymmetryAtomCoords = None

    def _splitSymmetryAtoms(self):
        """The symmetry-only / asymmetric-unit halves of the list (ref densityAnalysis.py:905-912), made when somebody asks for them: the blob
        statistics read the whole list only."""
        allAtoms, allCoords = self.symmetryAtoms, self.symmetryAtomCoords
        ident = allAtoms._ident
        self._symmetryOnlyAtoms = allAtoms.subset(np.nonzero(~ident)[0])
        self._symmetryOnlyAtomCoords = allCoords[~ident]
        self._asymmetryAtoms = allAtoms.subset(np.nonzero(ident)[0])
        self._asymmetryAtomCoords = allCoords[ident]

    # ---- blob statistics (ref densityAnalysis.py:914-939) --------------------------------------
    def calculateAtomSpecificBlobStatistics(self, blobList):
        symmetryAtoms = self.symmetryAtoms
        symmetryAtomCoords = self.symmetryAtomCoords
        if not self.densityElectronRatio:
            raise RuntimeError("Failed to calculate densityElectronRatio, probably due to total aggregated electrons less than the minimum.")
        ratio = self.densityElectronRatio
        if not blobList:
            return []
        if len(symmetryAtomCoords) == 0:
            # no symmetry atoms (a file without REMARK 290 has no operators): the reference's cdist() of a centroid against an
            # empty coordinate array raises this ValueError (densityAnalysis.py:933) -- an ordinary per-entry failure, not a device error
            raise ValueError("XB must be a 2-dimensional array.")
        listed = blobList.columns() if isinstance(blobList, ccp4.DeviceBlobs) else None
        if listed is not None:                   # straight from the device list's columns: no blob object is made for the table
            centroid_xyz = np.ascontiguousarray(listed["centroid"], dtype=np.float64)
            centroid = centroid_xyz                                  # (n x 3: _rows makes a list of three floats per row)
            total = listed["totalDensity"]
            num_voxels, volume = np.asarray(listed["n"], dtype=np.int64), np.asarray(listed["volume"], dtype=np.float64)
        else:
            centroid = [blob.centroid for blob in blobList]
            centroid_xyz = np.array(centroid, dtype=np.float64)
            total = np.array([blob.totalDensity for blob in blobList], dtype=np.float64)
            num_voxels, volume = [blob.numVoxels for blob in blobList], [blob.volume for blob in blobList]
        idx, dist = self.densityObj._ctx.nearest_atom(centroid_xyz, np.asarray(symmetryAtomCoords, dtype=np.float64))
        rows, symmetry, coords = symmetryAtoms.columns(idx)
        cols = _structure.columns(self.biopdbObj)
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        chain, number, resname = (cols.atom_lists(which) for which in ("chain", "number", "resname"))
        sign = np.where(total >= 0, '+', '-').tolist()
        return self._rows(list(dist), sign, np.abs(total / ratio), num_voxels, volume,
                          (chain, rows), (number, rows), (resname, rows), (cols.name if isinstance(cols.name, (list, tuple)) else list(cols.name), rows),       # (picked by _rows)
                          symmetry, coords, centroid)

    # ---- Fo / Fc maps, RSCC / RSR (ref densityAnalysis.py:426-446, 783-882) -------------------
    @property
    def fo(self):
        """ref densityAnalysis.py:438-446: the Fo map is the 2Fo-Fc map."""
        return self.densityObj

    @property
    def fc(self):
        """ref densityAnalysis.py:426-435: Fc = 2Fo-Fc - 2 (Fo-Fc), as a DensityMatrix.

        Reference behaviour kept on purpose: the reference makes a ``deepcopy`` of the 2Fo-Fc object and replaces only
        ``.density`` -- ``densityArray`` and the cached ``meanDensity`` / ``stdDensity`` / ``getTotalAbsDensity`` stay those
        of the Fo map (golden ``fc_mean_std``).  The grid is computed on the device (``pdbeda_map_combine``) and stored as
        float32 (the reference's is float64):
        the metrics below therefore take Fc voxel values as ``fo - 2 * diff`` in float64 from the two gathered float32
        values, which is exact."""
        if self._fc is None:
            d = self.densityObj
            fc = ccp4.DensityMatrix.fromDeviceMap(d.header, d.origin, type(d._map).combine(d._map, self.diffDensityObj._map, -2.0), d.pdbid, d._ctx)
            fc._flatOf = d                      # (== fc.densityArray = d.densityArray, without fetching a host copy)
            fc._meanDensity, fc._stdDensity = d.meanDensity, d.stdDensity
            fc._totalAbsDensity = d._totalAbsDensity
            self._fc = fc
        return self._fc

    def medianAbsFoFc(self):
        """ref densityAnalysis.py:783-801: medians of |Fo| and |Fc| over the voxels of the unique box whose |Fo| and |Fc| are
        both below mean + 1 sigma (the Fc statistics are those of Fo: see ``fc``).  Exact order statistics by a device-side
        radix select (``pdbeda_abs_select_hist``): the maps do not move; Fc values are fo - 2 diff in float64 like the reference's."""
        fo, fc = self.fo, self.fc
        foCut = fo.meanDensity + 1.0 * fo.stdDensity
        fcCut = fc.meanDensity + 1.0 * fc.stdDensity
        a, b = fo._map, self.diffDensityObj._map
        n = a.abs_order_statistics(b, -2.0, foCut, fcCut, 0)
        if n == 0:
            return (float("nan"), float("nan"))
        ranks = [n // 2] if n % 2 else [n // 2 - 1, n // 2]
        return tuple(float(np.mean(a.abs_order_statistics(b, -2.0, foCut, fcCut, which, ranks))) for which in (0, 1))

    residueMetricsHeaderList = ['chain', 'residue_number', 'residue_name', "rscc", "rsr", "mean_occupancy", "occupancy_weighted_mean_bfactor"]
    atomMetricsHeaderList = ['chain', 'residue_number', 'residue_name', "atom_name", "symmetry", "xyz", "rscc", "rsr", "occupancy", "bfactor"]

    def _metricsRadius(self):
        """ref densityAnalysis.py:813-818 / 850-855."""
        resolution = self.biopdbObj.header['resolution']
        radius = 0.7
        if 0.6 <= resolution <= 3:
            radius = (resolution - 0.6) / 3 + 0.7
        elif resolution > 3:
            radius = resolution * 0.5
        return radius

    def _rsccRsr(self, groups, radius):
        """RSCC and RSR of every group of coordinates: ONE sphere batch (all voxels of the sphere union, de-duplicated on the
        raw crs like the reference's set), two gathers (2Fo-Fc and Fo-Fc values under the wrap contract), then segmented
        float64 sums.  Pearson r as scipy.stats.pearsonr computes it (centre, normalise, dot, clip)."""
        xyz = np.array([c for g in groups for c in g], dtype=np.float64).reshape(-1, 3)
        off = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int64)
        bl = self.densityObj._map.sphere_blobs(xyz, np.full(len(xyz), radius, dtype=np.float32), off, 0.0)
        st = bl.stats()
        crs, voff = bl.voxels()
        vgroup = np.repeat(st["group"].astype(np.int64), np.diff(voff))
        order = np.argsort(vgroup, kind="stable")
        crs, vgroup = crs[order], vgroup[order]
        fo = self.densityObj._map.point_density(crs)
        fc = fo - self.diffDensityObj._map.point_density(crs) * 2
        n_groups = len(groups)
        cnt = np.bincount(vgroup, minlength=n_groups).astype(np.int64)
        start = np.concatenate([[0], np.cumsum(cnt)])[:-1]
        rscc = np.full(n_groups, np.nan)
        rsr = np.full(n_groups, np.nan)
        has = cnt > 0
        if has.any():
            seg = start[has]
            mean_fo = np.add.reduceat(fo, seg) / cnt[has]
            mean_fc = np.add.reduceat(fc, seg) / cnt[has]
            idx = np.cumsum(has) - 1                       # group -> row of the non-empty tables
            xm = fo - mean_fo[idx[vgroup]]
            ym = fc - mean_fc[idx[vgroup]]
            nx = np.sqrt(np.add.reduceat(xm * xm, seg))
            ny = np.sqrt(np.add.reduceat(ym * ym, seg))
            with np.errstate(invalid="ignore", divide="ignore"):
                r = np.add.reduceat((xm / nx[idx[vgroup]]) * (ym / ny[idx[vgroup]]), seg)
                rscc[has] = np.clip(r, -1.0, 1.0)
                rsr[has] = np.add.reduceat(np.abs(fo - fc), seg) / np.add.reduceat(np.abs(fo + fc), seg)
            rscc[cnt < 2] = np.nan                          # scipy raises for fewer than two points
        return rscc, rsr

    def calculateRsccRsrMetrics(self, crsList):
        """ref densityAnalysis.py:864-882 for one explicit voxel set."""
        crs = np.asarray(sorted(set(map(tuple, crsList))), dtype=np.int32).reshape(-1, 3)
        fo = self.densityObj._map.point_density(crs)
        fc = fo - self.diffDensityObj._map.point_density(crs) * 2
        xm, ym = fo - fo.mean(), fc - fc.mean()
        rscc = float(np.clip(np.dot(xm / np.linalg.norm(xm), ym / np.linalg.norm(ym)), -1.0, 1.0)) if len(crs) > 1 else float("nan")
        rsr = float(np.abs(fo - fc).sum() / np.abs(fo + fc).sum())
        return (rscc, rsr)

    def residueMetrics(self, residueList=None):
        """ref densityAnalysis.py:803-838: [chain, resnum, resname, rscc, rsr, mean occupancy, occupancy-weighted mean B]."""
        radius = self._metricsRadius()
        if residueList is None:
            residueList = list(self.biopdbObj.get_residues())
        rscc, rsr = self._rsccRsr([[atom.coord for atom in residue.child_list] for residue in residueList], radius)
        results = []
        for residue, cc, rr in zip(residueList, rscc, rsr):
            bfactorWeightedSum = occupancySum = 0.0
            for atom in residue.child_list:
                bfactorWeightedSum += atom.get_bfactor() * atom.get_occupancy()
                occupancySum += atom.get_occupancy()
            results.append([residue.parent.id, residue.id[1], residue.resname, float(cc), float(rr), occupancySum / len(residue.child_list),
                            bfactorWeightedSum / occupancySum])
        return results

    def atomMetrics(self, atomList=None):
        """ref densityAnalysis.py:840-862: [chain, resnum, resname, atom, symmetry, xyz, rscc, rsr, occupancy, bfactor]."""
        radius = self._metricsRadius()
        if atomList is None:
            atomList = self.asymmetryAtoms
        rscc, rsr = self._rsccRsr([[atom.coord] for atom in atomList], radius)
        return [[atom.parent.parent.id, atom.parent.id[1], atom.parent.resname, atom.name, atom.symmetry, atom.coord, float(cc), float(rr),
                 atom.get_occupancy(), atom.get_bfactor()] for atom, cc, rr in zip(atomList, rscc, rsr)]

    # ---- region density / discrepancy (ref densityAnalysis.py:948-1211), batched ---------------
    @staticmethod
    def _flatten(groups, radii):
        """Lists of coordinate lists + per-coordinate radius lists -> (xyz[n, 3], radius[n], offsets[groups + 1])."""
        sizes = np.fromiter((len(g) for g in groups), dtype=np.int64, count=len(groups))
        xyz = np.array([c for g in groups for c in g], dtype=np.float64).reshape(-1, 3)
        rad = np.fromiter((r for g in radii for r in g), dtype=np.float32, count=int(sizes.sum()))
        return xyz, rad, np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    def _needRatio(self):
        if not self.densityElectronRatio:
            raise RuntimeError("Failed to calculate densityElectronRatio, probably due to total aggregated electrons less than the minimum.")
        return self.densityElectronRatio

    def _densityColumns(self, xyz, rad, off, numSD):
        """The two columns of regionDensityHeader for every group of spheres (ONE device batch), and the validity flags."""
        ratio = self._needRatio()
        dm = self.densityObj
        pos, neg, cnt, valid = dm._map.region_sums(xyz, rad, off, dm.meanDensity + numSD * dm.stdDensity)
        return [pos, pos / ratio], valid

    def _discrepancyColumns(self, xyz, rad, off, numSD):
        """The ten columns of regionDiscrepancyHeader for every group, computed on whole columns (densityAnalysis.py:1200-1211)."""
        ratio = self._needRatio()
        dm = self.diffDensityObj
        cutoff = dm.meanDensity + numSD * dm.stdDensity
        pos, neg, cnt, valid = dm._map.region_sums(xyz, rad, off, cutoff)
        avg = dm.getTotalAbsDensity(cutoff) / dm.numStoredVoxels
        absd = np.abs(pos) + np.abs(neg)
        expected = avg * cnt
        net = pos + neg
        return [absd, absd / ratio, expected, expected / ratio, net, net / ratio, pos, pos / ratio, neg, neg / ratio], valid      # (whole columns: _rows makes the Python numbers)

    @staticmethod
    def _rows(*columns):
        """The table rows (lists) of whole columns: lists, numpy arrays (their ``.tolist()`` values; a 2-D array of ints gives a TUPLE per row), or
        (list, int64 index array) picks.
        One pass in C (``_hostwalk.table_rows``) when the helper is built; ``list(map(list, zip(...)))`` over plain lists otherwise."""
        walk = _structure._hostwalk()
        if walk is not None and hasattr(walk, "table_rows") and 1 <= len(columns) <= 32:
            prepared = [np.ascontiguousarray(c) if isinstance(c, np.ndarray) else c for c in columns]
            if all(not isinstance(c, np.ndarray) or ((c.ndim == 1 and c.dtype in (np.float64, np.int64, np.int32, np.bool_)) or (c.ndim == 2 and c.dtype in (np.float64, np.int64)))
                   for c in prepared):
                return walk.table_rows(prepared)
        plain = []
        for c in columns:
            if isinstance(c, tuple) and len(c) == 2 and isinstance(c[1], np.ndarray):
                plain.append([c[0][r] for r in c[1].tolist()])
            elif isinstance(c, np.ndarray) and c.ndim == 2 and c.dtype.kind in "iu":
                plain.append(list(map(tuple, c.tolist())))                     # (rows of ints are tuples: the symmetry operators)
            else:
                plain.append(c.tolist() if isinstance(c, np.ndarray) else c)
        return list(map(list, zip(*plain)))

    def _atomPick(self, type):
        """Rows of the structure columns of the atoms named ``type`` (all when empty) and their leading table columns (model,
        chain, residue number, residue name, atom name, occupancy: densityAnalysis.py:966-971)."""
        cols = _structure.columns(self.biopdbObj)
        lead = [cols.atom_lists("model"), cols.atom_lists("chain"), cols.atom_lists("number"), cols.atom_lists("resname"), cols.name, cols.occupancy_raw]
        if not type:
            return cols, np.arange(len(cols.atoms)), lead
        pick = np.nonzero(np.asarray(cols.name) == type)[0] if cols.atoms else np.zeros(0, dtype=np.int64)
        rows = pick.tolist()
        return cols, pick, [[column[r] for r in rows] for column in lead]

    def _pairRadius(self, cols, radius, useOptimizedRadii):
        """Per-atom radius of the structure columns: the optimised radius of the atom type where the name is known, else ``radius``."""
        if not useOptimizedRadii:
            return np.full(len(cols.atoms), radius, dtype=np.float32)
        typeMap = fullAtomNameMapAtomTypeGlobal
        per_pair = np.array([radiiGlobal[typeMap[name]] if name in typeMap else radius for name in cols.pair_names] + [radius], dtype=np.float32)
        return per_pair[cols.pair_of_atom]

    def _symmetryPick(self, type):
        atom_rows, symmetry, coords = self.symmetryAtoms.columns(None, type)
        cols = _structure.columns(self.biopdbObj)
        rows = atom_rows.tolist()
        lead = [[column[r] for r in rows] for column in (cols.atom_lists("model"), cols.atom_lists("chain"), cols.atom_lists("number"), cols.atom_lists("resname"), cols.name)]
        xyz = np.array(coords, dtype=np.float64).reshape(-1, 3)
        return cols, atom_rows, lead + [symmetry, coords], xyz

    def _residuePick(self, type, keepAtom, dropEmpty):
        """Residues (optionally only those named ``type``) with the atoms ``keepAtom(residue name, atom name)`` lets through:
        (lead columns incl. the mean occupancy, atom rows, offsets)."""
        cols = _structure.columns(self.biopdbObj)
        n_res = len(cols.residues)
        res_keep = np.ones(n_res, dtype=bool) if not type else np.asarray(cols.res_name) == type
        atom_keep = res_keep[cols.res_of_atom] if len(cols.atoms) else np.zeros(0, dtype=bool)
        if keepAtom is not None:
            resname = cols.atom_lists("resname")
            atom_keep = atom_keep & np.fromiter((keepAtom(rn, an) for rn, an in zip(resname, cols.name)), dtype=bool, count=len(cols.atoms))
        counts = np.bincount(cols.res_of_atom[atom_keep], minlength=n_res) if len(cols.atoms) else np.zeros(n_res, dtype=np.int64)
        if dropEmpty:
            res_keep = res_keep & (counts > 0)
        which = np.nonzero(res_keep)[0]
        counts = counts[which]
        atom_rows = np.nonzero(atom_keep)[0]
        mean_occupancy = _segmentMeans(cols.occupancy[atom_rows], counts)
        rows = which.tolist()
        lead = [[column[r] for r in rows] for column in (cols.res_model, cols.res_chain, cols.res_number, cols.res_name)] + [list(mean_occupancy)]
        return cols, lead, atom_rows, np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def calculateRegionDensity(self, xyzCoordList, radius, numSD=1.5, testValidCrs=False):
        """ref densityAnalysis.py:1037-1068."""
        rad = list(radius) if isinstance(radius, (list, tuple, np.ndarray)) else [radius] * len(xyzCoordList)
        columns, valid = self._densityColumns(*self._flatten([list(xyzCoordList)], [rad]), numSD)
        row = [column.tolist()[0] for column in columns]
        return (row, bool(valid[0])) if testValidCrs else row

    def calculateRegionDiscrepancy(self, xyzCoordList, radius, numSD=3.0, testValidCrs=False):
        """ref densityAnalysis.py:1160-1211."""
        rad = list(radius) if isinstance(radius, (list, tuple, np.ndarray)) else [radius] * len(xyzCoordList)
        columns, valid = self._discrepancyColumns(*self._flatten([list(xyzCoordList)], [rad]), numSD)
        row = [column.tolist()[0] for column in columns]
        return (row, bool(valid[0])) if testValidCrs else row

    def calculateAtomRegionDensity(self, radius, numSD=1.5, type="", useOptimizedRadii=False):
        """ref densityAnalysis.py:948-973 (all atoms in ONE device batch)."""
        cols, pick, lead = self._atomPick(type)
        columns, _ = self._densityColumns(cols.coord[pick], self._pairRadius(cols, radius, useOptimizedRadii)[pick], np.arange(len(pick) + 1, dtype=np.int64), numSD)
        return self._rows(*lead, *columns)

    def calculateSymmetryAtomRegionDensity(self, radius, numSD=1.5, type="", useOptimizedRadii=False):
        """ref densityAnalysis.py:975-999."""
        cols, atom_rows, lead, xyz = self._symmetryPick(type)
        columns, valid = self._densityColumns(xyz, self._pairRadius(cols, radius, useOptimizedRadii)[atom_rows], np.arange(len(atom_rows) + 1, dtype=np.int64), numSD)
        return self._rows(*lead, valid.astype(bool), *columns)

    def calculateResidueRegionDensity(self, radius, numSD=1.5, type="", atomMask=None, useOptimizedRadii=False):
        """ref densityAnalysis.py:1001-1035 (residues left without atoms by the mask are skipped)."""
        keep = None if not atomMask else (lambda resname, name: resname not in atomMask or name in atomMask[resname])
        cols, lead, atom_rows, off = self._residuePick(type, keep, True)
        columns, _ = self._densityColumns(cols.coord[atom_rows], self._pairRadius(cols, radius, useOptimizedRadii)[atom_rows], off, numSD)
        return self._rows(*lead, *columns)

    def calculateAtomRegionDiscrepancies(self, radius, numSD=3.0, type=""):
        """ref densityAnalysis.py:1081-1104 (33 ms/atom in the reference; ONE device batch here)."""
        cols, pick, lead = self._atomPick(type)
        columns, _ = self._discrepancyColumns(cols.coord[pick], np.full(len(pick), radius, dtype=np.float32), np.arange(len(pick) + 1, dtype=np.int64), numSD)
        return self._rows(*lead, *columns)

    def calculateSymmetryAtomRegionDiscrepancies(self, radius, numSD=3.0, type=""):
        """ref densityAnalysis.py:1106-1128."""
        cols, atom_rows, lead, xyz = self._symmetryPick(type)
        columns, valid = self._discrepancyColumns(xyz, np.full(len(atom_rows), radius, dtype=np.float32), np.arange(len(atom_rows) + 1, dtype=np.int64), numSD)
        return self._rows(*lead, valid.astype(bool), *columns)

    def calculateResidueRegionDiscrepancies(self, radius, numSD=3.0, type="", atomMask=None):
        """ref densityAnalysis.py:1130-1158 (with a mask, only atoms the mask names; residues it empties stay in the table)."""
        keep = None if not atomMask else (lambda resname, name: resname in atomMask and name in atomMask[resname])
        cols, lead, atom_rows, off = self._residuePick(type, keep, False)
        columns, _ = self._discrepancyColumns(cols.coord[atom_rows], np.full(len(atom_rows), radius, dtype=np.float32), off, numSD)
        return self._rows(*lead, *columns)
