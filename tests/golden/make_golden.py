"""Generate golden vectors by running the REFERENCE pdb_eda (Cython cutils path) here.

Run in the build container only:  python tests/golden/make_golden.py
Writes tests/golden/voxel_<case>.npz and tests/golden/analysis_<case>.npz.  Fixtures hold
inputs (synthetic grids / atoms generated from seeds by pdb_eda_amd.synthetic) and the
reference's numeric outputs -- never reference source.  numpy 2.2.6 / scipy 1.15.3 /
Python 3.10 / Cython 3.2.9 semantics (SURVEY.md 8c caveat).
"""
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.normpath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refload  # noqa: E402
from pdb_eda_amd import synthetic  # noqa: E402

CASES = {
    # name: (MapSpec kwargs, seed, sigma_filter)
    "orth": (dict(ncrs=(30, 28, 26), spacing=0.45), 11, 1.3),
    "orth_sub": (dict(ncrs=(24, 20, 22), interval=(40, 36, 30), crs_start=(-3, 5, 2), spacing=0.4), 12, 1.2),
    "orth_rep": (dict(ncrs=(34, 30, 28), interval=(30, 30, 24), crs_start=(2, -4, 0), spacing=0.5), 13, 1.4),
    "orth_perm": (dict(ncrs=(26, 30, 22), axis_order=(3, 1, 2), crs_start=(4, 0, -6), spacing=0.42), 14, 1.3),
    "hex": (dict(ncrs=(28, 26, 24), interval=(32, 36, 30), crs_start=(-5, 3, 7), axis_order=(2, 1, 3),
                 cell=(14.4, 12.8, 12.0), angles=(90.0, 90.0, 120.0)), 15, 1.3),
    "tric": (dict(ncrs=(26, 24, 28), cell=(11.7, 10.4, 12.6), angles=(82.0, 97.5, 108.25), crs_start=(1, 2, 3)), 16, 1.2),
    "wide": (dict(ncrs=(150, 9, 7), spacing=0.4), 17, 1.5),   # rows longer than two mask words
}


def ragged(list_of_arrays, dtype):
    off = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    for i, a in enumerate(list_of_arrays):
        off[i + 1] = off[i] + len(a)
    flat = np.concatenate([np.asarray(a, dtype=dtype).reshape(-1, 3) for a in list_of_arrays]) if list_of_arrays else np.zeros((0, 3), dtype)
    return flat, off


def blob_record(blobs):
    crs, off = ragged([sorted(b.crsList) for b in blobs], np.int32)
    return {"crs": crs.reshape(-1, 3), "off": off,
            "total": np.array([b.totalDensity for b in blobs], dtype=np.float64),
            "centroid": np.array([list(b.centroid) for b in blobs], dtype=np.float64).reshape(-1, 3),
            "center": np.array([list(b.coordCenter) for b in blobs], dtype=np.float64).reshape(-1, 3),
            "volume": np.array([b.volume for b in blobs], dtype=np.float64)}


def voxel_case(name, ccp4):
    kw, seed, sig = CASES[name]
    spec = synthetic.MapSpec(**kw)
    grid = synthetic.noise_grid(spec, seed, sig)
    # make the repeated part periodic where ncrs > interval (a real map is)
    nc, nr, ns = spec.ncrs
    ci = [spec.interval[a - 1] for a in spec.axis_order]
    if nc > ci[0]:
        grid[:, :, ci[0]:] = grid[:, :, :nc - ci[0]]
    if nr > ci[1]:
        grid[:, ci[1]:, :] = grid[:, :nr - ci[1], :]
    if ns > ci[2]:
        grid[ci[2]:, :, :] = grid[:ns - ci[2], :, :]
    raw = synthetic.ccp4_bytes(spec, grid, big_endian=(name == "orth_perm"), symmetry_bytes=(b"X" * 160 if name == "hex" else b""))
    dm = ccp4.parse(io.BytesIO(raw), name)
    h = dm.header
    out = {"ccp4_bytes": np.frombuffer(raw, dtype=np.uint8)}
    out["h_origin"] = np.asarray(h.origin, dtype=np.float64)
    out["h_ortho"] = np.asarray(h.orthoMat, dtype=np.float64)
    out["h_deortho"] = np.asarray(h.deOrthoMat, dtype=np.float64)
    out["h_unit_volume"] = np.float64(h.unitVolume)
    out["h_grid_length"] = np.asarray(h.gridLength, dtype=np.float64)
    out["h_ints"] = np.array(list(h.ncrs) + list(h.crsStart) + list(h.xyzInterval) + list(h.map2xyz) + list(h.map2crs) +
                             list(h.crsInterval) + list(h.uniqueNcrs), dtype=np.int64)
    mean, std = dm.meanDensity, dm.stdDensity
    out["mean"], out["std"] = np.float64(mean), np.float64(std)
    rng = np.random.default_rng(seed + 1000)

    # point densities incl. far out-of-range crs
    pts = rng.integers(-70, 110, size=(400, 3)).astype(np.int32)
    pts[:50] = rng.integers(0, 20, size=(50, 3))
    out["pt_crs"] = pts
    out["pt_density"] = np.array([dm.getPointDensityFromCrs([int(x) for x in p]) for p in pts], dtype=np.float64)
    out["pt_valid"] = np.array([ccp4.utils.testValidCrs(dm, [int(x) for x in p]) for p in pts], dtype=np.uint8)
    out["pt_xyz"] = np.array([np.asarray(h.crs2xyzCoord([int(x) for x in p]), dtype=np.float64) for p in pts])

    # xyz -> crs on float32 "atom" coordinates spread over and around the cell
    lo = np.min(out["pt_xyz"], axis=0)
    hi = np.max(out["pt_xyz"], axis=0)
    xyz32 = rng.uniform(lo, hi, size=(400, 3)).astype(np.float32)
    out["x2c_xyz"] = xyz32
    out["x2c_crs"] = np.array([h.xyz2crsCoord(p) for p in xyz32], dtype=np.int32)

    # sumOfAbs at several cutoffs
    cuts = [mean + 3 * std, mean + 1.5 * std, 0.0, float(np.float32(0.05))]
    out["soa_cut"] = np.array(cuts, dtype=np.float64)
    out["soa"] = np.array([ccp4.utils.sumOfAbs(dm.densityArray, c) for c in cuts], dtype=np.float64)

    # whole-map blobs
    for tag, cut in (("p30", mean + 3 * std), ("n30", -(mean + 3 * std)), ("p15", mean + 1.5 * std), ("n20", -(mean + 2 * std))):
        blobs = dm.createFullBlobList(cut)
        out["full_%s_cut" % tag] = np.float64(cut)
        out["full_%s_list" % tag] = np.array(ccp4.utils.createFullCrsList(dm, cut), dtype=np.int32).reshape(-1, 3)
        for k, v in blob_record(blobs).items():
            out["full_%s_%s" % (tag, k)] = v

    # spheres: "atoms" near interesting places (inside, near faces, outside the stored box)
    centre_crs = rng.integers(-4, max(spec.ncrs) + 4, size=(24, 3))
    atoms = np.array([np.asarray(h.crs2xyzCoord([int(x) for x in c]), dtype=np.float64) for c in centre_crs])
    atoms = (atoms + rng.uniform(-0.3, 0.3, atoms.shape)).astype(np.float32)
    radii = rng.choice([0.62, 0.75, 1.1, 1.24, 2.0, 3.5], size=len(atoms)).astype(np.float64)
    cutoffs = [0.0, mean + 1.5 * std, -(mean + 1.5 * std), mean + 0.5 * std]
    out["sph_xyz"], out["sph_radius"], out["sph_cut"] = atoms, radii, np.array(cutoffs)
    for ci_, cut in enumerate(cutoffs):
        lists = [np.array(dm.getSphereCrsFromXyz(a, float(r), cut), dtype=np.int32).reshape(-1, 3) for a, r in zip(atoms, radii)]
        flat, off = ragged(lists, np.int32)
        out["sph%d_crs" % ci_], out["sph%d_off" % ci_] = flat.reshape(-1, 3), off
        # per-atom blobs (findAberrantBlobs single coordinate); blobs in the reference's emission order
        nb, recs = [], []
        for a, r in zip(atoms, radii):
            try:
                blobs = dm.findAberrantBlobs(a, float(r), cut)
            except Exception:   # createCrsLists raises on an empty list in the reference (np.matrix of [])
                blobs = []
            nb.append(len(blobs))
            recs.extend(blobs)
        out["sphb%d_nblobs" % ci_] = np.array(nb, dtype=np.int64)
        for k, v in blob_record(recs).items():
            out["sphb%d_%s" % (ci_, k)] = v
    out["sph_valid"] = np.array([ccp4.utils.testValidXyz(dm, a, float(r)) for a, r in zip(atoms, radii)], dtype=np.uint8)

    # sphere unions (residue-like groups of 4 atoms, scalar radius and per-atom radii)
    groups = [list(range(i, i + 4)) for i in range(0, 24, 4)]
    for gi, grp in enumerate(groups):
        xyz_list = [atoms[i] for i in grp]
        for tag, rad in (("s", 1.9), ("l", [float(radii[i]) for i in grp])):
            for ci_, cut in enumerate(cutoffs[:3]):
                s = ccp4.utils.getSphereCrsFromXyzList(dm, xyz_list, rad, cut)
                out["uni_%s%d_g%d" % (tag, ci_, gi)] = np.array(sorted(s), dtype=np.int32).reshape(-1, 3)
                if s:
                    blobs = dm.findAberrantBlobs(xyz_list, rad, cut)
                    rec = blob_record(sorted(blobs, key=lambda b: min(b.crsList)))
                    for k, v in rec.items():
                        out["unib_%s%d_g%d_%s" % (tag, ci_, gi, k)] = v
        out["uni_valid_g%d" % gi] = np.uint8(ccp4.utils.testValidXyzList(dm, xyz_list, 1.9))

    # testOverlap on pairs of whole-map blobs
    blobs = dm.createFullBlobList(mean + 1.5 * std)[:12]
    pairs, res = [], []
    for i in range(len(blobs)):
        for j in range(i + 1, len(blobs)):
            pairs.append((i, j))
            res.append(bool(blobs[i].testOverlap(blobs[j])))
    out["ovl_pairs"] = np.array(pairs, dtype=np.int32).reshape(-1, 2)
    out["ovl_res"] = np.array(res, dtype=np.uint8)

    # symmetry atoms (cutils.createSymmetryAtoms) with two operators
    class _A(object):
        def __init__(self, c):
            self.coord = c
    rot = [np.hstack([np.eye(3), np.zeros((3, 1))]),
           np.array([[-1.0, 0.0, 0.0, 0.5 * h.xlength], [0.0, -1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.5 * h.zlength]])]
    ncrs = h.ncrs
    box = [h.crs2xyzCoord(i) for i in [[c, r, s] for c in [0, ncrs[0] - 1] for r in [0, ncrs[1] - 1] for s in [0, ncrs[2] - 1]]]
    xs, ys, zs = sorted(i[0] for i in box), sorted(i[1] for i in box), sorted(i[2] for i in box)
    alist = [_A(a) for a in atoms]
    sym = ccp4.utils.createSymmetryAtoms(alist, rot, h.orthoMat, xs, ys, zs)
    out["sym_rot"] = np.array(rot, dtype=np.float64)
    out["sym_box"] = np.array([[xs[0], ys[0], zs[0]], [xs[-1], ys[-1], zs[-1]]], dtype=np.float64)
    out["sym_atom"] = np.array([alist.index(s.atom) for s in sym], dtype=np.int32)
    out["sym_sym"] = np.array([list(s.symmetry) for s in sym], dtype=np.int32).reshape(-1, 4)
    out["sym_xyz"] = np.array([np.asarray(s.coord, dtype=np.float64) for s in sym]).reshape(-1, 3)

    path = os.path.join(HERE, "voxel_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    ccp4, da = refload.load()
    which = sys.argv[1:] or list(CASES)
    for name in which:
        if name in CASES:
            voxel_case(name, ccp4)
    if not sys.argv[1:] or "analysis" in sys.argv[1:]:
        import make_golden_analysis
        # (`analysis alias`: only the named analysis cases)
        make_golden_analysis.main(ccp4, da, only=[a for a in sys.argv[1:] if a in make_golden_analysis.CASES] or None)
