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
            densityObj = ccp4.read(ccp4DensityFile, pdbid) if isinstance(ccp4DensityFile, str) else ccp4.parse(ccp4DensityFile, pdbid, ctx=ctx)
            _attachCutoffs(densityObj, None)
        if ccp4DiffDensityFile is not None:
            diffDensityObj = ccp4.read(ccp4DiffDensityFile, pdbid) if isinstance(ccp4DiffDensityFile, str) else ccp4.parse(ccp4DiffDensityFile, pdbid, ctx=ctx)
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

    @property
    def fo(self):
        return self.densityObj

    # ---- aggregateCloud (ref densityAnalysis.py:571-780) --------------------------------------
    def aggregateCloud(self, minCloudElectrons=25.0, minTotalElectrons=400.0):
        """Aggregate the 2Fo-Fc clouds by atom, residue and domain; sets ``densityElectronRatio``,
        ``medians`` and the description tables.  Same silent-failure contract as the reference
        (Q7): everything stays ``None`` below ``minTotalElectrons`` or if the statistics tail fails.

        Device work: (1) one sphere batch -- a group per atom -- gives every atom's clouds
        (getSphereCrsFromXyz + createCrsLists + fromCrsList); (2) the all-pairs ``testOverlap``
        clustering of the reference equals the 26-connected components of the union of the pooled
        clouds' voxels, so residue clouds are ONE list batch with a group per residue and domain
        clouds one more with a single group; (3) the bonded-atom completeness tests are one batched
        voxel-set adjacency call.  The host only keeps the tables.
        """
        _requireParams()
        from scipy import stats
        densityObj = self.densityObj
        dmap = densityObj._map
        unitVolume = densityObj.header.unitVolume
        radii, electronsMap, typeMap = radiiGlobal, fullAtomNameMapElectronsGlobal, fullAtomNameMapAtomTypeGlobal

        # eligible atoms in the reference's iteration order
        residues = [res for res in self.biopdbObj.get_residues() if res.id[0] == ' ']
        elig = []            # (residue index, atom, resAtom)
        for ri, residue in enumerate(residues):
            for atom in residue.child_list:
                resAtom = residueAtomName(atom)
                if resAtom not in typeMap or atom.get_occupancy() == 0:
                    continue
                elig.append((ri, atom, resAtom))
        if not elig:
            return
        xyz = np.array([a.coord for _, a, _ in elig], dtype=np.float64)
        rad = np.array([radii[typeMap[ra]] for _, _, ra in elig], dtype=np.float32)
        clouds = dmap.sphere_blobs(xyz, rad, np.arange(len(elig) + 1), densityObj.densityCutoff)
        cst = clouds.stats()
        ccrs, coff = clouds.voxels()
        cgroup = cst["group"]
        first_cloud = np.searchsorted(cgroup, np.arange(len(elig)), side="left")
        last_cloud = np.searchsorted(cgroup, np.arange(len(elig)), side="right")

        # duplicates of an atom coordinate share one dict entry in the reference (last one wins)
        by_coord = {}
        for ai, (_, atom, _) in enumerate(elig):
            by_coord[tuple(atom.coord)] = ai
        src = [by_coord[tuple(atom.coord)] for _, atom, _ in elig]

        # distance of every cloud centroid to its atom, all atoms at once: the same row-wise float64 arithmetic as the
        # reference's per-atom np.linalg.norm(coord - centroids, axis=1) (float32 coordinates promote exactly)
        cloud_dist = np.linalg.norm(xyz[cgroup] - cst["centroid"], axis=1) if len(cgroup) else np.zeros(0)
        src = np.asarray(src, dtype=np.int64)
        has_cloud = last_cloud[src] > first_cloud[src]
        min_dist = np.full(len(elig), np.nan)
        owners_with = np.nonzero(last_cloud > first_cloud)[0]
        if len(owners_with):
            min_dist[owners_with] = np.minimum.reduceat(cloud_dist, first_cloud[owners_with])
        centroidDistances = min_dist[src][has_cloud]
        centroidDistanceCutoff = np.nanmedian(centroidDistances) + 2.5 * np.nanstd(centroidDistances)

        # pass 2: best cloud per atom, pooled clouds per residue
        atomList = []
        pool_cloud = []      # pooled cloud -> device cloud index
        pool_atom = []       # pooled cloud -> eligible atom index
        pool_res = []        # pooled cloud -> residue index
        res_atom_clouds = collections.defaultdict(dict)   # residue -> {resAtom: [pool indices]}
        for ai, (ri, atom, resAtom) in enumerate(elig):
            s = src[ai]
            lo, hi = first_cloud[s], last_cloud[s]
            if hi == lo:
                continue
            if hi - lo == 1:
                best = lo
            else:
                if min_dist[s] > centroidDistanceCutoff:
                    continue
                best = lo + int(np.argmin(cloud_dist[lo:hi]))
            res_atom_clouds[ri][resAtom] = list(range(len(pool_cloud), len(pool_cloud) + (hi - lo)))
            for ci in range(lo, hi):
                pool_cloud.append(ci)
                pool_atom.append(ai)
                pool_res.append(ri)
            residue = residues[ri]
            centroid = list(cst["centroid"][best])
            atomList.append([residue.parent.id, residue.id[1], atom.parent.resname, atom.name, typeMap[resAtom],
                             cst["totalDensity"][best] / electronsMap[resAtom] / atom.get_occupancy(), int(cst["n"][best]),
                             electronsMap[resAtom], atom.get_bfactor(), _norm3(atom.coord - cst["centroid"][best]), centroid])
        if not pool_cloud:
            return
        pool_cloud = np.asarray(pool_cloud)
        pool_atom = np.asarray(pool_atom)
        pool_res = np.asarray(pool_res)
        weights = np.array([electronsMap[ra] * a.get_occupancy() for _, a, ra in elig], dtype=np.float64)

        # bonded-atom overlap completeness (densityAnalysis.py:652-659): one batched adjacency call.
        # Voxel sets: all clouds of an atom = a contiguous slice of the sphere batch's voxel list.
        completely = collections.defaultdict(int)
        incompletely = collections.defaultdict(int)
        pair_a, pair_b, pair_owner = [], [], []
        atom_of_key = {}
        for ai, (ri, atom, resAtom) in enumerate(elig):
            if resAtom in res_atom_clouds.get(ri, {}):
                atom_of_key[(ri, resAtom)] = ai      # the reference's dict keeps the LAST atom of a name
        owners = []
        for ri, residue in enumerate(residues):
            have = res_atom_clouds.get(ri, {})
            for atom in residue.child_list:
                resAtom = residueAtomName(atom)
                if resAtom in have:
                    owner = len(owners)
                    owners.append((typeMap[resAtom], 0))
                    for resAtom2 in bondedAtomsGlobal[resAtom]:
                        if resAtom2 in have:
                            pair_a.append(src[atom_of_key[(ri, resAtom)]])
                            pair_b.append(src[atom_of_key[(ri, resAtom2)]])
                            pair_owner.append(owner)
        atom_set_off = np.concatenate([coff[first_cloud], [coff[-1]]]).astype(np.int64)   # voxel slice (all clouds) per eligible atom
        touching = dmap._ctx.test_overlap(ccrs, atom_set_off, pair_a, pair_b) if pair_a else np.zeros(0, bool)
        fails = np.zeros(len(owners), dtype=np.int64)
        np.add.at(fails, np.asarray(pair_owner, dtype=np.int64), (~touching).astype(np.int64))
        for (atomType, _), bad in zip(owners, fails):
            if bad == 0:
                completely[atomType] += 1
            else:
                incompletely[atomType] += 1

        # residue clouds: 26-connected components of each residue's pooled voxels
        def components(group_of_pool, n_groups):
            """Union the pooled clouds' voxels per group on the device (createBlobList on the union
            == the reference's all-pairs testOverlap clustering + merge); returns the component
            statistics and, for every pooled cloud, the component that contains it."""
            order = np.argsort(group_of_pool, kind="stable")
            starts = coff[pool_cloud[order]]
            lens = coff[pool_cloud[order] + 1] - starts
            ends = np.cumsum(lens)
            idx = np.repeat(starts - (ends - lens), lens) + np.arange(int(ends[-1]) if len(ends) else 0)
            goff = np.zeros(n_groups + 1, dtype=np.int64)
            np.add.at(goff, group_of_pool[order] + 1, lens)
            goff = np.cumsum(goff)
            bl = dmap.list_blobs(ccrs[idx], goff)
            st = bl.stats()
            vcrs, voff = bl.voxels()
            comp_of_voxel = np.repeat(np.arange(len(st["n"])), np.diff(voff))
            vgroup = st["group"][comp_of_voxel].astype(np.int64)
            vk = _crs_keys(vcrs)
            qk = _crs_keys(ccrs[coff[pool_cloud]])          # first voxel of every pooled cloud
            uniq, inv = np.unique(np.concatenate([vk, qk]), return_inverse=True)
            big = len(uniq)
            vcomb = vgroup * big + inv[:len(vk)]
            qcomb = group_of_pool.astype(np.int64) * big + inv[len(vk):]
            srt = np.argsort(vcomb)
            return st, comp_of_voxel[srt][np.searchsorted(vcomb[srt], qcomb)]

        res_ids, res_group = np.unique(pool_res, return_inverse=True)
        rst, res_comp_of_pool = components(res_group, len(res_ids))
        residueList = []
        n_rcomp = len(rst["n"])
        # electrons of a residue cloud = distinct atoms that contribute a pooled cloud to it
        pairs = np.unique(np.stack([res_comp_of_pool, pool_atom], axis=1), axis=0)
        relectrons = np.zeros(n_rcomp)
        np.add.at(relectrons, pairs[:, 0], weights[pairs[:, 1]])
        # emission order of the reference: by residue, then by the lowest pooled-cloud index in the component
        first_pool = np.full(n_rcomp, len(pool_cloud), dtype=np.int64)
        np.minimum.at(first_pool, res_comp_of_pool, np.arange(len(pool_cloud)))
        for k in np.argsort(first_pool, kind="stable"):
            if relectrons[k] >= minCloudElectrons:
                residue = residues[res_ids[rst["group"][k]]]
                residueList.append([residue.parent.id, residue.id[1], residue.resname, rst["totalDensity"][k] / relectrons[k], int(rst["n"][k]),
                                    relectrons[k], int(rst["n"][k]) * unitVolume, list(rst["centroid"][k])])

        # domain clouds: components of the union of everything pooled
        dst, dom_comp_of_pool = components(np.zeros(len(pool_cloud), dtype=np.int64), 1)
        n_dcomp = len(dst["n"])
        pairs = np.unique(np.stack([dom_comp_of_pool, pool_atom], axis=1), axis=0)
        delectrons = np.zeros(n_dcomp)
        np.add.at(delectrons, pairs[:, 0], weights[pairs[:, 1]])
        first_pool = np.full(n_dcomp, len(pool_cloud), dtype=np.int64)
        np.minimum.at(first_pool, dom_comp_of_pool, np.arange(len(pool_cloud)))
        numVoxels = 0
        totalElectrons = 0
        totalDensity = 0
        domainList = []
        for k in np.argsort(first_pool, kind="stable"):
            totalElectrons += delectrons[k]
            numVoxels += int(dst["n"][k])
            totalDensity += dst["totalDensity"][k]
            if delectrons[k] >= minCloudElectrons:
                rep = residues[pool_res[first_pool[k]]]   # a representative residue (the reference's is set-order dependent)
                domainList.append([rep.parent.id, rep.id[1], rep.resname, dst["totalDensity"][k] / delectrons[k], int(dst["n"][k]), delectrons[k],
                                   int(dst["n"][k]) * unitVolume, list(dst["centroid"][k])])
        if totalElectrons < minTotalElectrons:
            return
        densityElectronRatio = totalDensity / totalElectrons
        domainList.sort(key=lambda x: x[3])

        # ---- host-side statistics tail (densityAnalysis.py:734-767), numpy/scipy like the reference ----
        currentSlopes = slopesGlobal

        def calcSlope(data, atom_type):
            if len(data['chain']) <= 2 or len(np.unique(data['bfactor'])) == 1:
                return currentSlopes[atom_type]
            slope, intercept, r_value, p_value, std_err = stats.linregress(np.log(data['bfactor']), (data['adj_density_electron_ratio'] - densityElectronRatio) / densityElectronRatio)
            return currentSlopes[atom_type] if p_value > 0.05 else slope

        try:
            dataType = np.dtype([('chain', np.dtype(('U', 20))), ('residue_number', int), ('residue_name', np.dtype(('U', 10))), ('atom_name', np.dtype(('U', 10))),
                                 ('atom_type', np.dtype(('U', atomTypeLengthGlobal))), ('density_electron_ratio', float), ('num_voxels', int), ('electrons', int),
                                 ('bfactor', float), ('centroid_distance', float), ('centroid_xyz', float, (3,)), ('adj_density_electron_ratio', float),
                                 ('domain_fraction', float), ('corrected_fraction', float), ('corrected_density_electron_ratio', float), ('volume', float)])
            atoms = np.asarray([tuple(atom + [0.0 for x in range(5)]) for atom in atomList], dataType)
            if not np.isnan(atoms['centroid_distance']).all():
                centroidCutoff = np.nanmedian(atoms['centroid_distance']) + np.nanstd(atoms['centroid_distance']) * 2
                atoms = atoms[atoms['centroid_distance'] < centroidCutoff]
            atom_types = np.unique(atoms['atom_type'])

            def med(column, mask_extra=None):
                out = {}
                for t in atom_types:
                    sel = atoms['atom_type'] == t
                    if mask_extra is not None:
                        sel = sel & mask_extra
                    out[t] = np.nanmedian(atoms[column][sel])
                return out

            medians = {'num_voxels': med('num_voxels')}
            lookup = np.vectorize(lambda column, atom_type: medians[column][atom_type])
            atoms['adj_density_electron_ratio'] = atoms['density_electron_ratio'] / atoms['num_voxels'] * lookup('num_voxels', atoms['atom_type'])
            atoms['volume'] = atoms['num_voxels'] * unitVolume
            for column in ['density_electron_ratio', 'centroid_distance', 'adj_density_electron_ratio', 'volume']:
                medians[column] = med(column)
            medians['bfactor'] = med('bfactor', atoms['bfactor'] > 0)
            atoms['bfactor'][atoms['bfactor'] <= 0] = lookup('bfactor', atoms['atom_type'])[atoms['bfactor'] <= 0]
            medians['slopes'] = {t: calcSlope(atoms[atoms['atom_type'] == t], t) for t in atom_types}
            atoms['domain_fraction'] = (atoms['adj_density_electron_ratio'] - densityElectronRatio) / densityElectronRatio
            atoms['corrected_fraction'] = atoms['domain_fraction'] - (np.log(atoms['bfactor']) - np.log(lookup('bfactor', atoms['atom_type']))) * lookup('slopes', atoms['atom_type'])
            atoms['corrected_density_electron_ratio'] = atoms['corrected_fraction'] * densityElectronRatio + densityElectronRatio
            for column in ['domain_fraction', 'corrected_fraction', 'corrected_density_electron_ratio']:
                medians[column] = med(column)
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
        allAtoms = []
        identity = (0, 0, 0, 0)
        for a, s4, x in zip(idx.tolist(), map(tuple, sym.tolist()), xyz):
            allAtoms.append(SymAtom(atoms[a], atoms[a].coord if s4 == identity else x, s4))
        ident = ~np.any(sym != 0, axis=1)
        allCoords = np.where(ident[:, None], coords[idx], xyz)      # == np.asarray([atom.coord ...]): float32 coordinates promote exactly
        self._symmetryAtoms = allAtoms
        self._symmetryAtomCoords = allCoords
        self._symmetryOnlyAtoms = [atom for atom, i in zip(allAtoms, ident.tolist()) if not i]
        self._symmetryOnlyAtomCoords = allCoords[~ident]
        self._asymmetryAtoms = [atom for atom, i in zip(allAtoms, ident.tolist()) if i]
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
        xyz = np.array([c for g in groups for c in g], dtype=np.float64).reshape(-1, 3)
        rad = np.array([r for g in radii for r in g], dtype=np.float32)
        off = np.concatenate([[0], np.cumsum([len(g) for g in groups])]).astype(np.int64)
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
        rows = [[float(p), float(p) / ratio] for p in pos]
        return (rows, valid) if want_valid else rows

    def _discrepancyRows(self, groups, radii, numSD, want_valid):
        ratio = self._needRatio()
        dm = self.diffDensityObj
        cutoff = dm.meanDensity + numSD * dm.stdDensity
        pos, neg, cnt, valid = self._regionBatch(dm, groups, radii, cutoff)
        avg = dm.getTotalAbsDensity(cutoff) / dm.densityArray.size
        rows = []
        for p, n, c in zip(pos, neg, cnt):
            p, n = float(p), float(n)
            absd = abs(p) + abs(n)
            expected = avg * int(c)
            rows.append([absd, absd / ratio, expected, expected / ratio, p + n, (p + n) / ratio, p, p / ratio, n, n / ratio])
        return (rows, valid) if want_valid else rows

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
        atoms = [a for a in self.biopdbObj.get_atoms() if not type or a.name == type]
        rows = self._densityRows([[a.coord] for a in atoms], [[self._atomRadius(a, radius, useOptimizedRadii)] for a in atoms], numSD, False)
        return [[a.parent.parent.parent.id, a.parent.parent.id, a.parent.id[1], a.parent.resname, a.name, a.get_occupancy()] + r for a, r in zip(atoms, rows)]

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
        atoms = [a for a in self.biopdbObj.get_atoms() if not type or a.name == type]
        rows = self._discrepancyRows([[a.coord] for a in atoms], [[radius] for a in atoms], numSD, False)
        return [[a.parent.parent.parent.id, a.parent.parent.id, a.parent.id[1], a.parent.resname, a.name, a.get_occupancy()] + r for a, r in zip(atoms, rows)]

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
