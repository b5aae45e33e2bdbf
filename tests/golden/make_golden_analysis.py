"""Golden vectors for the densityAnalysis rows: run the REFERENCE DensityAnalysis
(aggregateCloud, region density / discrepancy, blob statistics) on synthetic entries.

Run through make_golden.py (build container only).  The parameter table used here is
synthetic (pdb_eda_amd.synthetic.synthetic_params) -- the reference's optimized_params.json is
reference data and is neither copied nor needed.
"""
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

from pdb_eda_amd import synthetic  # noqa: E402
from pdb_eda_amd import ccp4 as my_ccp4  # noqa: E402  (host-side header only: no device work)
from pdb_eda_amd import structure as my_structure  # noqa: E402

CASES = {
    # "alias" (round 4): atoms that SHARE a float32 coordinate.  The reference keys allAtomClouds by tuple(atom.coord)
    # (densityAnalysis.py:605, read back at :622): all eligible atoms of one coordinate use the clouds of the LAST of them (found
    # with ITS radius), as the same DensityBlob objects -- whose .atoms the loop overwrites (:639).  Three kinds, see alias_sites().
    "alias": dict(spec=dict(ncrs=(72, 64, 68), spacing=0.5), n_res=60, seed=27, alias=True),
    "orth": dict(spec=dict(ncrs=(72, 64, 68), spacing=0.5), n_res=56, seed=21),
    "hex": dict(spec=dict(ncrs=(64, 60, 56), interval=(72, 80, 64), crs_start=(-4, 6, 3), axis_order=(2, 1, 3),
                          cell=(40.0, 36.0, 32.0), angles=(90.0, 90.0, 120.0)), n_res=40, seed=22),
}


def entry(name):
    cfg = CASES[name]
    spec = synthetic.MapSpec(**cfg["spec"])
    header = my_ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    nc, nr, ns = spec.ncrs
    corners = np.array([header.crs2xyzCoord([c, r, s]) for c in (4, nc - 5) for r in (4, nr - 5) for s in (4, ns - 5)], dtype=np.float64)
    lo, hi = corners.min(axis=0), corners.max(axis=0)
    if not header.orthogonal:       # keep the chain inside the skewed cell: shrink the box around its centre
        mid = (lo + hi) / 2
        lo, hi = mid - (hi - lo) / 4, mid + (hi - lo) / 4
    st = synthetic.chain_structure(cfg["n_res"], cfg["seed"], lo, hi, hetero_every=9, zero_occupancy_every=37)
    if cfg.get("alias"):
        alias_sites(st)
    params = synthetic.synthetic_params()
    dens = synthetic.gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=cfg["seed"])
    diff = (synthetic.noise_grid(spec, cfg["seed"] + 100, 1.2) * 0.12).astype(np.float32)
    rot = [np.hstack([np.eye(3), np.zeros((3, 1))]),
           np.array([[-1.0, 0.0, 0.0, 0.5 * header.xlength], [0.0, -1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.5 * header.zlength]])]
    return spec, st, params, dens, diff, rot


def alias_sites(st):
    """Coincident coordinates of three kinds (alternate locations collapsed onto one site, placeholders): (i) two atoms of ONE
    residue -- the side-chain atom put on the C-alpha, and, in another residue, the EARLIER atom put on a later one; (ii) atoms of
    two DIFFERENT residues with different atom types, hence different radii -- a carbonyl O on the next residue's N, and a
    C-beta on the C of a residue further on; (iii) an atom with a ZERO-occupancy twin (the twin is skipped by both loops and must
    not become the alias).  Plain (non-hetero) residues only, occupancies untouched elsewhere."""
    res = [r for r in st.get_residues() if r.id[0] == " "]
    def atom(r, name):
        return next(a for a in r.child_list if a.name == name)
    atom(res[8], "CB").coord = atom(res[8], "CA").coord.copy()            # (i) later atom takes the earlier one's place
    atom(res[14], "N").coord = atom(res[14], "O").coord.copy()            # (i) earlier atom onto a later one: the O's radius finds both
    atom(res[20], "O").coord = atom(res[21], "N").coord.copy()            # (ii) O (residue k) and N (residue k + 1)
    atom(res[30], "CB").coord = atom(res[33], "C").coord.copy()           # (ii) three residues apart
    twin = atom(res[40], "CB")                                            # (iii) zero-occupancy twin of the C-alpha, listed after it
    twin.coord = atom(res[40], "CA").coord.copy()
    twin.occupancy = 0.0
    twin2 = atom(res[44], "N")                                            # (iii) ... and listed BEFORE the atom it shadows
    twin2.coord = atom(res[44], "C").coord.copy()
    twin2.occupancy = 0.0


def structure_arrays(st):
    atoms = list(st.get_atoms())
    return {"atom_name": np.array([a.name for a in atoms]), "atom_coord": np.array([a.coord for a in atoms], dtype=np.float32),
            "atom_occ": np.array([a.get_occupancy() for a in atoms]), "atom_b": np.array([a.get_bfactor() for a in atoms]),
            "atom_element": np.array([a.element for a in atoms]), "atom_resnum": np.array([a.parent.id[1] for a in atoms]),
            "atom_het": np.array([a.parent.id[0] for a in atoms]), "atom_resname": np.array([a.parent.resname for a in atoms])}


def main(ccp4, da, only=None):
    for name in CASES:
        if only and name not in only:
            continue
        spec, st, params, dens, diff, rot = entry(name)
        da.setGlobals(params)
        densityObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, dens)), name)
        diffObj = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, diff)), name)
        densityObj.densityCutoff = densityObj.meanDensity + 1.5 * densityObj.stdDensity
        diffObj.diffDensityCutoff = diffObj.meanDensity + 3 * diffObj.stdDensity
        pdbObj = my_structure.PDBEntry(my_structure.PDBHeader(pdbid=name, resolution=2.0, spaceGroup="P_1", rotationMats=rot))
        an = da.DensityAnalysis(name, densityObj, diffObj, st, pdbObj)
        out = {"dens": dens, "diff": diff, "rot": np.array(rot), "spec": np.array(json.dumps(CASES[name]["spec"]))}
        out.update(structure_arrays(st))
        an.aggregateCloud()
        assert an.densityElectronRatio, "synthetic entry failed the reference's own minimum-electrons gate"
        out["ratio"] = np.float64(an.densityElectronRatio)
        out["num_voxels"] = np.int64(an.numVoxelsAggregated)
        out["total_electrons"] = np.float64(an.totalAggregatedElectrons)
        out["total_density"] = np.float64(an.totalAggregatedDensity)
        atoms = an.atomCloudDescriptions
        for f in atoms.dtype.names:
            out["acd_" + f] = np.asarray(atoms[f])
        out["res_rows"] = np.array([[r[1]] + [r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.residueCloudDescriptions], dtype=np.float64).reshape(-1, 8)
        out["dom_rows"] = np.array([[r[3], r[4], r[5], r[6]] + list(r[7]) for r in an.domainCloudDescriptions], dtype=np.float64).reshape(-1, 7)
        out["medians"] = np.array(json.dumps({k: {t: float(v) for t, v in d.items()} for k, d in an.medians.items()}))
        out["overlap_complete"] = np.array(json.dumps(dict(an.atomTypeOverlapCompleteness)))
        out["overlap_incomplete"] = np.array(json.dumps(dict(an.atomTypeOverlapIncompleteness)))
        # region statistics
        out["atom_discrepancy"] = np.array([r[6:] for r in an.calculateAtomRegionDiscrepancies(3.5, 3.0)], dtype=np.float64)
        out["atom_discrepancy_r2"] = np.array([r[6:] for r in an.calculateAtomRegionDiscrepancies(2.0, 2.5, type="CA")], dtype=np.float64)
        out["residue_discrepancy"] = np.array([r[5:] for r in an.calculateResidueRegionDiscrepancies(3.5, 3.0)], dtype=np.float64)
        out["atom_density"] = np.array([r[6:] for r in an.calculateAtomRegionDensity(1.0, 1.5, useOptimizedRadii=True)], dtype=np.float64)
        out["residue_density"] = np.array([r[5:] for r in an.calculateResidueRegionDensity(1.2, 1.5)], dtype=np.float64)
        sym = an.symmetryAtoms
        pick = list(range(0, len(sym), max(1, len(sym) // 60)))[:60]
        rows, valid = [], []
        for i in pick:
            res, ok = an.calculateRegionDiscrepancy([sym[i].coord], 3.5, 3.0, testValidCrs=True)
            rows.append(res)
            valid.append(ok)
        out["sym_pick"] = np.array(pick, dtype=np.int64)
        out["sym_count"] = np.int64(len(sym))
        out["sym_discrepancy"] = np.array(rows, dtype=np.float64)
        out["sym_valid"] = np.array(valid, dtype=np.uint8)
        out["sym_coords"] = np.array([np.asarray(sym[i].coord, dtype=np.float64) for i in pick])
        out["sym_tags"] = np.array([list(sym[i].symmetry) for i in pick], dtype=np.int64)
        # blob statistics for the green / red lists
        for tag, blobs in (("green", an.greenBlobList), ("red", an.redBlobList)):
            stats = an.calculateAtomSpecificBlobStatistics(blobs)
            out["blob_%s_num" % tag] = np.array([[s[0], s[2], s[3], s[4]] for s in stats], dtype=np.float64).reshape(-1, 4)
            out["blob_%s_sign" % tag] = np.array([s[1] for s in stats])
            out["blob_%s_atom" % tag] = np.array(["%s|%s|%s|%s" % (s[6], s[7], s[8], tuple(int(v) for v in s[9])) for s in stats])
            out["blob_%s_centroid" % tag] = np.array([list(s[11]) for s in stats], dtype=np.float64).reshape(-1, 3)
        # RSCC / RSR metrics (densityAnalysis.py:803-882) at the entry's resolution, and the Fo / Fc scale check (783-801)
        st.header = {"resolution": 2.0}
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")      # scipy.stats.stats deprecation in the reference's pearsonr call
            rm = an.residueMetrics()
            am = an.atomMetrics()
            out["residue_metrics"] = np.array([[r[3], r[4], r[5], r[6]] for r in rm], dtype=np.float64)
            out["residue_metrics_id"] = np.array(["%s|%s|%s" % (r[0], r[1], r[2]) for r in rm])
            out["atom_metrics"] = np.array([[r[6], r[7], r[8], r[9]] for r in am], dtype=np.float64)
            out["atom_metrics_id"] = np.array(["%s|%s|%s|%s" % (r[0], r[1], r[2], r[3]) for r in am])
            out["median_abs_fo_fc"] = np.array(an.medianAbsFoFc(), dtype=np.float64)
            fc = an.fc
            out["fc_mean_std"] = np.array([fc.meanDensity, fc.stdDensity], dtype=np.float64)
        path = os.path.join(HERE, "analysis_%s.npz" % name)
        np.savez_compressed(path, **out)
        print("wrote", path, os.path.getsize(path) // 1024, "KiB")
