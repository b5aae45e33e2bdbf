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
    if diffDensityObj is not None:
        diffDensityObj.diffDensityCutoff = diffDensityObj.meanDensity + 3 * diffDensityObj.stdDensity


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


class _SymAtomList(object):
    """The list of SymAtom objects createSymmetryAtoms returns (cutils.pyx:73-103), materialised item by item."""

    def __init__(self, atoms, idx, sym, xyz, ident, rows=None):
        self._atoms, self._idx, self._sym, self._xyz, self._ident = atoms, idx, sym, xyz, ident
        self._rows = np.arange(len(idx)) if rows is None else np.asarray(rows)
        self._made = {}

    def subset(self, rows):
        return _SymAtomList(self._atoms, self._idx, self._sym, self._xyz, self._ident, self._rows[rows])

    def __len__(self):
        return len(self._rows)

    def _make(self, r):
        atom = self._atoms[int(self._idx[r])]
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
        self._fc = None
        self._atomTableCache = None

    # ---- lazy properties (ref densityAnalysis.py:326-565) ------------------------------------
    def _lazy(name, trigger):
        def getter(self):
            if getattr(self, name) is None:
                getattr(self, trigger)()
            return getattr(self, name)
        return property(getter)

    symmetryAtoms = _lazy('_symmetryAtoms', '_calculateSymmetryAtoms')
    symmetryOnlyAtoms = _lazy('_symmetryOnlyAtoms', '_calculateSymmetryAtoms')
    asymmetryAtoms = _lazy('_asymmetryAtoms', '_calculateSymmetryAtoms')
    symmetryAtomCoords = _lazy('_symmetryAtomCoords', '_calculateSymmetryAtoms')
    symmetryOnlyAtomCoords = _lazy('_symmetryOnlyAtomCoords', '_calculateSymmetryAtoms')
    asymmetryAtomCoords = _lazy('_asymmetryAtomCoords', '_calculateSymmetryAtoms')
    medians = _lazy('_medians', 'aggregateCloud')
    atomCloudDescriptions = _lazy('_atomCloudDescriptions', 'aggregateCloud')
    residueCloudDescriptions = _lazy('_residueCloudDescriptions', 'aggregateCloud')
    domainCloudDescriptions = _lazy('_domainCloudDescriptions', 'aggregateCloud')
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
        """Flatten what aggregateCloud reads from the structure (densityAnalysis.py:596-603, 617-621, 653-656) into the arrays
        of ``pdbeda_cloud_atoms``: the eligible atoms in the reference's iteration order, a key per (residue, residue_atom
        name), the bonded-name table restricted to each residue, and the 'owners' of the completeness count."""
        typeMap, electronsMap, radii = fullAtomNameMapAtomTypeGlobal, fullAtomNameMapElectronsGlobal, radiiGlobal
        residues = [res for res in self.biopdbObj.get_residues() if res.id[0] == ' ']
        atoms, resAtoms, residue_of, key_of = [], [], [], []
        keys = {}                       # (residue ordinal, residue_atom name) -> key id
        children = []                   # (residue ordinal, residue_atom name) of EVERY child atom, for the owners
        for ri, residue in enumerate(residues):
            for atom in residue.child_list:
                resAtom = residueAtomName(atom)
                children.append((ri, resAtom))
                if resAtom not in typeMap or atom.get_occupancy() == 0:
                    continue
                atoms.append(atom)
                resAtoms.append(resAtom)
                residue_of.append(ri)
                key_of.append(keys.setdefault((ri, resAtom), len(keys)))
        n = len(atoms)
        coords = np.array([a.coord for a in atoms], dtype=np.float64).reshape(n, 3)
        last = {}
        for i, a in enumerate(atoms):
            last[a.coord.tobytes()] = i                       # allAtomClouds is keyed by the coordinate: the last one wins (604)
        alias = np.fromiter((last[a.coord.tobytes()] for a in atoms), dtype=np.int32, count=n)
        bonded_off, bonded = [0], []
        for (ri, resAtom) in keys:                            # (dicts keep insertion order = key id order)
            bonded.extend(keys[(ri, r2)] for r2 in bondedAtomsGlobal[resAtom] if (ri, r2) in keys)
            bonded_off.append(len(bonded))
        owner_key = [keys[c] for c in children if c in keys]
        owner_type = [typeMap[c[1]] for c in children if c in keys]
        occupancy = np.fromiter((a.get_occupancy() for a in atoms), dtype=np.float64, count=n)
        electrons = np.fromiter((electronsMap[ra] for ra in resAtoms), dtype=np.float64, count=n)
        return {"residues": residues, "atoms": atoms, "resAtoms": resAtoms, "xyz": coords, "occupancy": occupancy, "electrons": electrons,
                "radius": np.fromiter((radii[typeMap[ra]] for ra in resAtoms), dtype=np.float32, count=n),
                "residue": np.asarray(residue_of, dtype=np.int32), "alias": alias, "key": np.asarray(key_of, dtype=np.int32),
                "bonded_off": np.asarray(bonded_off, dtype=np.int64), "bonded": np.asarray(bonded, dtype=np.int32),
                "owner_key": np.asarray(owner_key, dtype=np.int32), "owner_type": owner_type}

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
        if not inp["atoms"]:
            return
        res = densityObj._map.aggregate_cloud(inp["xyz"], inp["radius"], inp["electrons"] * inp["occupancy"], inp["residue"], inp["alias"], inp["key"],
                                              inp["bonded_off"], inp["bonded"], inp["owner_key"], densityObj.densityCutoff, minCloudElectrons)
        if not len(res["atom"]):
            return
        completely = collections.defaultdict(int)
        incompletely = collections.defaultdict(int)
        for atomType, state in zip(inp["owner_type"], res["owner_state"].tolist()):
            if state == 1:
                completely[atomType] += 1
            elif state == 2:
                incompletely[atomType] += 1
        residues = inp["residues"]

        def cloudRows(t):
            return [[residues[ri].parent.id, residues[ri].id[1], residues[ri].resname, tot / el, int(nv), el, int(nv) * unitVolume, list(cen)]
                    for ri, tot, nv, el, cen in zip(t["residue"].tolist(), t["total"].tolist(), t["n"].tolist(), t["electrons"].tolist(), t["centroid"].tolist())]
        residueList = cloudRows(res["res"])
        domainList = cloudRows(res["dom"])
        numVoxels, totalElectrons, totalDensity = res["numVoxels"], res["totalElectrons"], res["totalDensity"]
        if totalElectrons < minTotalElectrons:
            return
        densityElectronRatio = totalDensity / totalElectrons
        domainList.sort(key=lambda x: x[3])

        try:
            atoms, medians = self._cloudStatistics(inp, res, densityElectronRatio, unitVolume, typeMap)
        except Exception:
            return

        self._densityElectronRatio = densityElectronRatio
        self._numVoxelsAggregated = numVoxels
        self._totalAggregatedElectrons = totalElectrons
        self._totalAggregatedDensity = totalDensity
        self._medians = medians
        self._atomCloudDescriptions = atoms
        self._residueCloudDescriptions = residueList
        self._domainCloudDescriptions = domainList
        self._atomTypeOverlapCompleteness = completely
        self._atomTypeOverlapIncompleteness = incompletely

    @staticmethod
    def _cloudStatistics(inp, res, ratio, unitVolume, typeMap):
        """The host-side statistics over the atom table (what densityAnalysis.py:734-767 computes), on whole columns: the
        per-atom-type medians come from ONE sort per column instead of a masked nanmedian per (column, type)."""
        from scipy import stats
        idx = res["atom"]
        n = len(idx)
        table = np.zeros(n, dtype=np.dtype([
            ('chain', 'U20'), ('residue_number', int), ('residue_name', 'U10'), ('atom_name', 'U10'), ('atom_type', 'U%d' % atomTypeLengthGlobal),
            ('density_electron_ratio', float), ('num_voxels', int), ('electrons', int), ('bfactor', float), ('centroid_distance', float),
            ('centroid_xyz', float, (3,)), ('adj_density_electron_ratio', float), ('domain_fraction', float), ('corrected_fraction', float),
            ('corrected_density_electron_ratio', float), ('volume', float)]))
        picked = [inp["atoms"][i] for i in idx.tolist()]
        table['chain'] = [a.parent.parent.id for a in picked]
        table['residue_number'] = [a.parent.id[1] for a in picked]
        table['residue_name'] = [a.parent.resname for a in picked]
        table['atom_name'] = [a.name for a in picked]
        table['atom_type'] = [typeMap[inp["resAtoms"][i]] for i in idx.tolist()]
        table['density_electron_ratio'] = res["atom_total"] / inp["electrons"][idx] / inp["occupancy"][idx]
        table['num_voxels'] = res["atom_n"]
        table['electrons'] = inp["electrons"][idx]
        table['bfactor'] = [a.get_bfactor() for a in picked]
        table['centroid_distance'] = res["atom_distance"]
        table['centroid_xyz'] = res["atom_centroid"]
        dist = table['centroid_distance']
        if not np.isnan(dist).all():
            table = table[dist < np.nanmedian(dist) + np.nanstd(dist) * 2]
        atom_types, group = np.unique(table['atom_type'], return_inverse=True)
        n_types = len(atom_types)

        def typeMedians(values, keep=None):
            """np.nanmedian of ``values`` per atom type: sort once by (type, value) -- NaNs and dropped rows last -- and take
            the middle one or the mean of the middle two of every type's run."""
            v = np.asarray(values, dtype=np.float64)
            if keep is not None:
                v = np.where(keep, v, np.nan)
            order = np.lexsort((v, group))
            sv = v[order]
            start = np.searchsorted(group[order], np.arange(n_types))
            count = np.bincount(group, weights=~np.isnan(v), minlength=n_types).astype(np.int64)
            lo, hi = start + np.maximum(count - 1, 0) // 2, start + count // 2
            safe = np.minimum(np.stack([lo, hi]), max(len(sv) - 1, 0))
            med = (sv[safe[0]] + sv[safe[1]]) / 2.0 if len(sv) else np.full(n_types, np.nan)
            return np.where(count > 0, med, np.nan)

        def asDict(per_type):
            return dict(zip(atom_types.tolist(), per_type))
        medians = {}
        m_vox = typeMedians(table['num_voxels'])
        medians['num_voxels'] = asDict(m_vox)
        table['adj_density_electron_ratio'] = table['density_electron_ratio'] / table['num_voxels'] * m_vox[group]
        table['volume'] = table['num_voxels'] * unitVolume
        for column in ('density_electron_ratio', 'centroid_distance', 'adj_density_electron_ratio', 'volume'):
            medians[column] = asDict(typeMedians(table[column]))
        m_b = typeMedians(table['bfactor'], table['bfactor'] > 0)
        medians['bfactor'] = asDict(m_b)
        missing = table['bfactor'] <= 0
        table['bfactor'][missing] = m_b[group][missing]
        # slope of the b-factor dependence per atom type: linear regression where there is something to fit, else the table's slope
        fraction = (table['adj_density_electron_ratio'] - ratio) / ratio
        log_b = np.log(table['bfactor'])
        slopes = np.zeros(n_types)
        for k, atom_type in enumerate(atom_types.tolist()):
            sel = group == k
            fit = stats.linregress(log_b[sel], fraction[sel]) if (sel.sum() > 2 and len(np.unique(table['bfactor'][sel])) != 1) else None
            slopes[k] = slopesGlobal[atom_type] if (fit is None or fit.pvalue > 0.05) else fit.slope
        medians['slopes'] = asDict(slopes)
        table['domain_fraction'] = fraction
        table['corrected_fraction'] = fraction - (log_b - np.log(m_b[group])) * slopes[group]
        table['corrected_density_electron_ratio'] = table['corrected_fraction'] * ratio + ratio
        for column in ('domain_fraction', 'corrected_fraction', 'corrected_density_electron_ratio'):
            medians[column] = asDict(typeMedians(table[column]))
        return table, medians

    # ---- symmetry atoms (ref densityAnalysis.py:885-912 + cutils.pyx:73-103) -------------------
    def _calculateSymmetryAtoms(self):
        densityObj = self.densityObj
        header = densityObj.header
        ncrs = header.ncrs
        corners = np.array([[c, r, s] for c in [0, ncrs[0] - 1] for r in [0, ncrs[1] - 1] for s in [0, ncrs[2] - 1]], dtype=np.int32)
        box = densityObj._map.crs2xyz(corners)
        lo, hi = box.min(axis=0), box.max(axis=0)
        atoms = list(self.biopdbObj.get_atoms())
        coords = np.array([a.coord for a in atoms], dtype=np.float64)
        rot = np.array([np.asarray(m, dtype=np.float64) for m in self.pdbObj.header.rotationMats])
        idx, sym, xyz = densityObj._ctx.symmetry_atoms(coords, rot, np.asarray(header.orthoMat, dtype=np.float64), lo, hi)
        ident = ~np.any(sym != 0, axis=1)
        allCoords = np.where(ident[:, None], coords[idx], xyz)      # == np.asarray([atom.coord ...]): float32 coordinates promote exactly
        # the SymAtom objects are made on demand: a blob-statistics table touches a few hundred of the thousands there are
        allAtoms = _SymAtomList(atoms, idx, sym, xyz, ident)
        self._symmetryAtoms = allAtoms
        self._symmetryAtomCoords = allCoords
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
        centroids = np.array([blob.centroid for blob in blobList], dtype=np.float64)
        idx, dist = self.densityObj._ctx.nearest_atom(centroids, np.asarray(symmetryAtomCoords, dtype=np.float64))
        blobStats = []
        for blob, i, d in zip(blobList, idx, dist):
            atom = symmetryAtoms[int(i)]
            sign = '+' if blob.totalDensity >= 0 else '-'
            blobStats.append([d, sign, abs(blob.totalDensity / ratio), blob.numVoxels, blob.volume, atom.parent.parent.id, atom.parent.id[1], atom.parent.resname,
                              atom.name, atom.symmetry, atom.coord, blob.centroid])
        return blobStats

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
            fc.densityArray = d.densityArray
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
    def _regionBatch(self, dm, groups, radii, cutoff):
        """groups: list of lists of coordinates; radii: per-coordinate list of lists.  One device call."""
        sizes = np.fromiter((len(g) for g in groups), dtype=np.int64, count=len(groups))
        xyz = np.array([c for g in groups for c in g], dtype=np.float64).reshape(-1, 3)
        rad = np.fromiter((r for g in radii for r in g), dtype=np.float32, count=int(sizes.sum()))
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        return dm._map.region_sums(xyz, rad, off, cutoff)

    def _needRatio(self):
        if not self.densityElectronRatio:
            raise RuntimeError("Failed to calculate densityElectronRatio, probably due to total aggregated electrons less than the minimum.")
        return self.densityElectronRatio

    def _densityRows(self, groups, radii, numSD, want_valid):
        ratio = self._needRatio()
        dm = self.densityObj
        cutoff = dm.meanDensity + numSD * dm.stdDensity
        pos, neg, cnt, valid = self._regionBatch(dm, groups, radii, cutoff)
        rows = np.stack([pos, pos / ratio], axis=1).tolist()
        return (rows, valid) if want_valid else rows

    def _discrepancyRows(self, groups, radii, numSD, want_valid):
        """The ten columns of regionDiscrepancyHeader for every group, computed on whole columns (densityAnalysis.py:1200-1211)."""
        ratio = self._needRatio()
        dm = self.diffDensityObj
        cutoff = dm.meanDensity + numSD * dm.stdDensity
        pos, neg, cnt, valid = self._regionBatch(dm, groups, radii, cutoff)
        avg = dm.getTotalAbsDensity(cutoff) / dm.densityArray.size
        absd = np.abs(pos) + np.abs(neg)
        expected = avg * cnt
        net = pos + neg
        rows = np.stack([absd, absd / ratio, expected, expected / ratio, net, net / ratio, pos, pos / ratio, neg, neg / ratio], axis=1).tolist()
        return (rows, valid) if want_valid else rows

    def _atomTable(self):
        """All atoms of the structure with the leading columns of the per-atom tables (model, chain, residue number, residue name,
        atom name, occupancy: densityAnalysis.py:966-971), built once per analysis."""
        if self._atomTableCache is None:
            atoms = list(self.biopdbObj.get_atoms())
            lead = [[a.parent.parent.parent.id, a.parent.parent.id, a.parent.id[1], a.parent.resname, a.name, a.get_occupancy()] for a in atoms]
            self._atomTableCache = (atoms, lead)
        return self._atomTableCache

    def calculateRegionDensity(self, xyzCoordList, radius, numSD=1.5, testValidCrs=False):
        """ref densityAnalysis.py:1037-1068."""
        rad = list(radius) if isinstance(radius, (list, tuple, np.ndarray)) else [radius] * len(xyzCoordList)
        out = self._densityRows([list(xyzCoordList)], [rad], numSD, testValidCrs)
        return (out[0][0], bool(out[1][0])) if testValidCrs else out[0]

    def calculateRegionDiscrepancy(self, xyzCoordList, radius, numSD=3.0, testValidCrs=False):
        """ref densityAnalysis.py:1160-1211."""
        rad = list(radius) if isinstance(radius, (list, tuple, np.ndarray)) else [radius] * len(xyzCoordList)
        out = self._discrepancyRows([list(xyzCoordList)], [rad], numSD, testValidCrs)
        return (out[0][0], bool(out[1][0])) if testValidCrs else out[0]

    def _atomRadius(self, atom, radius, useOptimizedRadii):
        resAtom = residueAtomName(atom)
        return radiiGlobal[fullAtomNameMapAtomTypeGlobal[resAtom]] if useOptimizedRadii and resAtom in fullAtomNameMapAtomTypeGlobal else radius

    def calculateAtomRegionDensity(self, radius, numSD=1.5, type="", useOptimizedRadii=False):
        """ref densityAnalysis.py:948-973 (all atoms in ONE device batch)."""
        atoms, lead = self._atomTable()
        pick = [i for i, a in enumerate(atoms) if not type or a.name == type]
        rows = self._densityRows([[atoms[i].coord] for i in pick], [[self._atomRadius(atoms[i], radius, useOptimizedRadii)] for i in pick], numSD, False)
        return [lead[i] + r for i, r in zip(pick, rows)]

    def calculateSymmetryAtomRegionDensity(self, radius, numSD=1.5, type="", useOptimizedRadii=False):
        """ref densityAnalysis.py:975-999."""
        atoms = [a for a in self.symmetryAtoms if not type or a.name == type]
        rows, valid = self._densityRows([[a.coord] for a in atoms], [[self._atomRadius(a, radius, useOptimizedRadii)] for a in atoms], numSD, True)
        return [[a.parent.parent.parent.id, a.parent.parent.id, a.parent.id[1], a.parent.resname, a.name, a.symmetry, a.coord, bool(v)] + r
                for a, r, v in zip(atoms, rows, valid)]

    def calculateResidueRegionDensity(self, radius, numSD=1.5, type="", atomMask=None, useOptimizedRadii=False):
        """ref densityAnalysis.py:1001-1035."""
        sel = []
        for residue in self.biopdbObj.get_residues():
            if type and residue.resname != type:
                continue
            atoms = [a for a in residue.get_atoms() if not atomMask or residue.resname not in atomMask or a.name in atomMask[residue.resname]]
            if atoms:
                sel.append((residue, atoms))
        rows = self._densityRows([[a.coord for a in atoms] for _, atoms in sel],
                                 [[self._atomRadius(a, radius, useOptimizedRadii) for a in atoms] for _, atoms in sel], numSD, False)
        return [[res.parent.parent.id, res.parent.id, res.id[1], res.resname, np.mean([a.get_occupancy() for a in atoms])] + r
                for (res, atoms), r in zip(sel, rows)]

    def calculateAtomRegionDiscrepancies(self, radius, numSD=3.0, type=""):
        """ref densityAnalysis.py:1081-1104 (33 ms/atom in the reference; ONE device batch here)."""
        atoms, lead = self._atomTable()
        pick = [i for i, a in enumerate(atoms) if not type or a.name == type]
        rows = self._discrepancyRows([[atoms[i].coord] for i in pick], [[radius]] * len(pick), numSD, False)
        return [lead[i] + r for i, r in zip(pick, rows)]

    def calculateSymmetryAtomRegionDiscrepancies(self, radius, numSD=3.0, type=""):
        """ref densityAnalysis.py:1106-1128."""
        atoms = [a for a in self.symmetryAtoms if not type or a.name == type]
        rows, valid = self._discrepancyRows([[a.coord] for a in atoms], [[radius] for a in atoms], numSD, True)
        return [[a.parent.parent.parent.id, a.parent.parent.id, a.parent.id[1], a.parent.resname, a.name, a.symmetry, a.coord, bool(v)] + r
                for a, r, v in zip(atoms, rows, valid)]

    def calculateResidueRegionDiscrepancies(self, radius, numSD=3.0, type="", atomMask=None):
        """ref densityAnalysis.py:1130-1158."""
        sel = []
        for residue in self.biopdbObj.get_residues():
            if type and residue.resname != type:
                continue
            atoms = [a for a in residue.get_atoms() if not atomMask or (residue.resname in atomMask and a.name in atomMask[residue.resname])]
            sel.append((residue, atoms))
        rows = self._discrepancyRows([[a.coord for a in atoms] for _, atoms in sel], [[radius] * len(atoms) for _, atoms in sel], numSD, False)
        return [[res.parent.parent.id, res.parent.id, res.id[1], res.resname, np.mean([a.get_occupancy() for a in atoms])] + r
                for (res, atoms), r in zip(sel, rows)]
