"""ctypes wrapper of the CPU oracle (oracle/pdbeda_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpdbeda_oracle.so")


class OraMap(C.Structure):
    _fields_ = [("ncrs", C.c_int32 * 3), ("crs_start", C.c_int32 * 3), ("xyz_interval", C.c_int32 * 3),
                ("map2xyz", C.c_int32 * 3), ("map2crs", C.c_int32 * 3), ("crs_interval", C.c_int32 * 3),
                ("unique_ncrs", C.c_int32 * 3), ("orthogonal", C.c_int32),
                ("ortho", C.c_double * 9), ("deortho", C.c_double * 9), ("origin", C.c_double * 3),
                ("grid_len", C.c_double * 3), ("unit_volume", C.c_double), ("density", C.c_void_p)]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "pdbeda_oracle.c")):
        tmp = "libpdbeda_oracle.%d.tmp.so" % os.getpid()     # never leave a half-written library where another process may load it
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "OUT=" + tmp, tmp])
        os.replace(os.path.join(_HERE, tmp), _SO)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        p, i64 = C.c_void_p, C.c_int64
        mp = C.POINTER(OraMap)
        _lib.ora_point_density.restype = C.c_double
        _lib.ora_point_density.argtypes = [mp, p]
        _lib.ora_valid_crs.restype = C.c_int
        _lib.ora_valid_crs.argtypes = [mp, p]
        _lib.ora_crs2xyz.restype = None
        _lib.ora_crs2xyz.argtypes = [mp, p, p]
        _lib.ora_xyz2crs.restype = None
        _lib.ora_xyz2crs.argtypes = [mp, p, p]
        _lib.ora_full_crs_list.restype = i64
        _lib.ora_full_crs_list.argtypes = [mp, C.c_float, p, i64]
        _lib.ora_cluster.restype = i64
        _lib.ora_cluster.argtypes = [p, i64, p]
        _lib.ora_blob_stats.restype = None
        _lib.ora_blob_stats.argtypes = [mp, p, i64, p]
        _lib.ora_sphere_crs.restype = i64
        _lib.ora_sphere_crs.argtypes = [mp, p, C.c_float, C.c_float, p, i64]
        _lib.ora_sphere_crs_list.restype = i64
        _lib.ora_sphere_crs_list.argtypes = [mp, p, p, i64, C.c_float, p, i64]
        _lib.ora_valid_xyz.restype = C.c_int
        _lib.ora_valid_xyz.argtypes = [mp, p, C.c_float]
        _lib.ora_sum_of_abs.restype = C.c_double
        _lib.ora_sum_of_abs.argtypes = [p, i64, C.c_float]
        _lib.ora_mean_std.restype = None
        _lib.ora_mean_std.argtypes = [p, i64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib.ora_test_overlap.restype = C.c_int
        _lib.ora_test_overlap.argtypes = [p, i64, p, i64]
        _lib.ora_symmetry_atoms.restype = i64
        _lib.ora_symmetry_atoms.argtypes = [p, i64, p, C.c_int32, p, p, p, p, p, p, i64]
        _lib.ora_full_blobs.restype = i64
        _lib.ora_full_blobs.argtypes = [mp, C.c_float, p, p, p, i64, p]
        _lib.ora_cloud_begin.restype = C.c_void_p
        _lib.ora_cloud_begin.argtypes = [mp, i64, p, p, p, p, p, p, i64, p, p, i64, p, C.c_float, p]
        _lib.ora_cloud_finish.restype = C.c_int
        _lib.ora_cloud_finish.argtypes = [C.c_void_p, C.c_double, C.c_double, i64] + [p] * 16 + [p, p]
        _lib.ora_cloud_end.restype = None
        _lib.ora_cloud_end.argtypes = [C.c_void_p]
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle(object):
    """The reference algorithm on one map.  ``header`` is any object with the reference's
    DensityHeader attribute names (the product's pdb_eda_amd.ccp4.DensityHeader, whose
    derived fields are themselves pinned against the reference by the golden vectors)."""

    def __init__(self, header, density):
        self.header = header
        self.density = np.ascontiguousarray(density, dtype=np.float32).reshape(header.ncrs[2], header.ncrs[1], header.ncrs[0])
        m = OraMap()
        for k in range(3):
            m.ncrs[k] = header.ncrs[k]
            m.crs_start[k] = header.crsStart[k]
            m.xyz_interval[k] = header.xyzInterval[k]
            m.map2xyz[k] = header.map2xyz[k]
            m.map2crs[k] = header.map2crs[k]
            m.crs_interval[k] = header.crsInterval[k]
            m.unique_ncrs[k] = header.uniqueNcrs[k]
            m.origin[k] = float(header.origin[k])
            m.grid_len[k] = float(header.gridLength[k])
        o = np.asarray(header.orthoMat, dtype=np.float64).reshape(9)
        d = np.asarray(header.deOrthoMat, dtype=np.float64).reshape(9)
        for k in range(9):
            m.ortho[k] = float(o[k])
            m.deortho[k] = float(d[k])
        m.orthogonal = 1 if (header.alpha == header.beta == header.gamma == 90) else 0
        m.unit_volume = float(header.unitVolume)
        m.density = self.density.ctypes.data
        self.m = m
        self.L = lib()

    def point_density(self, crs):
        c = np.asarray(crs, dtype=np.int32)
        return self.L.ora_point_density(C.byref(self.m), _ptr(c))

    def valid_crs(self, crs):
        c = np.asarray(crs, dtype=np.int32)
        return bool(self.L.ora_valid_crs(C.byref(self.m), _ptr(c)))

    def crs2xyz(self, crs):
        c = np.asarray(crs, dtype=np.int32)
        out = np.zeros(3)
        self.L.ora_crs2xyz(C.byref(self.m), _ptr(c), _ptr(out))
        return out

    def xyz2crs(self, xyz):
        x = np.asarray(xyz, dtype=np.float64)
        out = np.zeros(3, dtype=np.int32)
        self.L.ora_xyz2crs(C.byref(self.m), _ptr(x), _ptr(out))
        return out

    def full_crs_list(self, cutoff):
        n = self.L.ora_full_crs_list(C.byref(self.m), C.c_float(cutoff), None, 0)
        if n < 0:
            return None
        out = np.zeros((n, 3), dtype=np.int32)
        self.L.ora_full_crs_list(C.byref(self.m), C.c_float(cutoff), _ptr(out), n)
        return out

    def cluster(self, crs):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        lab = np.zeros(len(crs), dtype=np.int32)
        n = self.L.ora_cluster(_ptr(crs), len(crs), _ptr(lab))
        assert n >= 0
        return [crs[lab == k] for k in range(n)]

    def blob_stats(self, crs):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        st = np.zeros(8)
        self.L.ora_blob_stats(C.byref(self.m), _ptr(crs), len(crs), _ptr(st))
        return {"totalDensity": st[0], "centroid": st[1:4].copy(), "coordCenter": st[4:7].copy(), "volume": st[7], "n": len(crs)}

    def blob_list(self, crs):
        """createBlobList: clusters + stats, in the reference's emission order."""
        return [dict(self.blob_stats(c), crs=c) for c in self.cluster(crs)]

    def sphere_crs(self, xyz, radius, cutoff=0.0):
        x = np.asarray(xyz, dtype=np.float64)
        n = self.L.ora_sphere_crs(C.byref(self.m), _ptr(x), C.c_float(radius), C.c_float(cutoff), None, 0)
        out = np.zeros((n, 3), dtype=np.int32)
        self.L.ora_sphere_crs(C.byref(self.m), _ptr(x), C.c_float(radius), C.c_float(cutoff), _ptr(out), n)
        return out

    def sphere_crs_list(self, xyz_list, radii, cutoff=0.0):
        x = np.ascontiguousarray(xyz_list, dtype=np.float64).reshape(-1, 3)
        r = np.ascontiguousarray(np.broadcast_to(np.asarray(radii, dtype=np.float32), (len(x),)))
        cap = self.L.ora_sphere_crs_list(C.byref(self.m), _ptr(x), _ptr(r), len(x), C.c_float(cutoff), None, 0)
        out = np.zeros((max(cap, 1), 3), dtype=np.int32)
        n = self.L.ora_sphere_crs_list(C.byref(self.m), _ptr(x), _ptr(r), len(x), C.c_float(cutoff), _ptr(out), cap)
        assert n >= 0
        return out[:n]

    def find_aberrant_blobs(self, xyz_list, radii, cutoff=0.0):
        x = np.asarray(xyz_list, dtype=np.float64).reshape(-1, 3)
        if len(x) > 1:
            crs = self.sphere_crs_list(x, radii, cutoff)
        else:
            crs = self.sphere_crs(x[0], np.asarray(radii, dtype=np.float32).reshape(-1)[0], cutoff)
        return self.blob_list(crs)

    def aggregate_cloud(self, xyz, radius, weight, residue, alias, key, bonded_off, bonded, owner_key, cutoff, min_cloud_electrons):
        """The composite behind DensityAnalysis.aggregateCloud (densityAnalysis.py:571-731) on the flattened structure: same
        arguments and same result dict as the product's DeviceMap.aggregate_cloud, so it can stand in for the device in the
        host-side table code.  The centroid-distance cut-off between the two phases is numpy's own (607)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        a = {"radius": np.ascontiguousarray(radius, dtype=np.float32), "weight": np.ascontiguousarray(weight, dtype=np.float64),
             "residue": np.ascontiguousarray(residue, dtype=np.int32), "alias": np.ascontiguousarray(alias, dtype=np.int32),
             "key": np.ascontiguousarray(key, dtype=np.int32), "bonded_off": np.ascontiguousarray(bonded_off, dtype=np.int64),
             "bonded": np.ascontiguousarray(bonded, dtype=np.int32), "owner_key": np.ascontiguousarray(owner_key, dtype=np.int32)}
        n, no = len(xyz), len(a["owner_key"])
        dist = np.zeros(max(n, 1))
        st = self.L.ora_cloud_begin(C.byref(self.m), n, _ptr(xyz), _ptr(a["radius"]), _ptr(a["weight"]), _ptr(a["residue"]), _ptr(a["alias"]), _ptr(a["key"]),
                                    len(a["bonded_off"]) - 1, _ptr(a["bonded_off"]), _ptr(a["bonded"]), no, _ptr(a["owner_key"]), C.c_float(cutoff), _ptr(dist))
        if not st:
            raise ValueError("ora_cloud_begin failed (allocation, or an alias index out of range)")
        try:
            d = dist[:n][~np.isnan(dist[:n])]
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                cut = float(np.nanmedian(d) + 2.5 * np.nanstd(d)) if len(d) else float("nan")     # densityAnalysis.py:607
            cap = max(4 * n, 16)
            at = {"atom": np.zeros(cap, np.int32), "atom_total": np.zeros(cap), "atom_n": np.zeros(cap, np.int64), "atom_centroid": np.zeros((cap, 3)),
                  "atom_distance": np.zeros(cap)}
            tabs = {tag: {"residue": np.zeros(cap, np.int32), "total": np.zeros(cap), "n": np.zeros(cap, np.int64), "electrons": np.zeros(cap),
                          "centroid": np.zeros((cap, 3))} for tag in ("res", "dom")}
            owner = np.zeros(max(no, 1), np.uint8)
            counts, totals = np.zeros(3, np.int64), np.zeros(3)
            rc = self.L.ora_cloud_finish(st, cut, float(min_cloud_electrons), cap,
                                         _ptr(at["atom"]), _ptr(at["atom_total"]), _ptr(at["atom_n"]), _ptr(at["atom_centroid"]), _ptr(at["atom_distance"]),
                                         *[_ptr(tabs[tag][f]) for tag in ("res", "dom") for f in ("residue", "total", "n", "electrons", "centroid")],
                                         _ptr(owner), _ptr(counts), _ptr(totals))
            assert rc == 0, rc
        finally:
            self.L.ora_cloud_end(st)
        na, nr, nd = (int(v) for v in counts)
        out = {"numVoxels": int(totals[0]), "totalElectrons": float(totals[1]), "totalDensity": float(totals[2]), "centroidDistanceCutoff": cut,
               "owner_state": owner[:no].copy()}
        out.update({k: v[:na].copy() for k, v in at.items()})
        out["res"] = {k: v[:nr].copy() for k, v in tabs["res"].items()}
        out["dom"] = {k: v[:nd].copy() for k, v in tabs["dom"].items()}
        return out

    def valid_xyz(self, xyz, radius):
        x = np.asarray(xyz, dtype=np.float64)
        return bool(self.L.ora_valid_xyz(C.byref(self.m), _ptr(x), C.c_float(radius)))

    def mean_std(self):
        """(np.mean, np.std) of all stored voxels with numpy's own summation tree (ccp4.py:343-363)."""
        a = self.density.reshape(-1)
        m, s = C.c_double(), C.c_double()
        self.L.ora_mean_std(_ptr(a), a.size, C.byref(m), C.byref(s))
        return m.value, s.value

    def sum_of_abs(self, cutoff):
        a = self.density.reshape(-1)
        return self.L.ora_sum_of_abs(_ptr(a), a.size, C.c_float(cutoff))

    def full_blobs(self, cutoff, labels=False):
        uc, ur, us = self.header.uniqueNcrs
        cap = ((uc + 1) // 2) * ((ur + 1) // 2) * ((us + 1) // 2) + 1
        n = np.zeros(cap, dtype=np.int64)
        st = np.zeros((cap, 8))
        key = np.zeros(cap, dtype=np.int64)
        lab = np.zeros((us, ur, uc), dtype=np.int32) if labels else None
        nb = self.L.ora_full_blobs(C.byref(self.m), C.c_float(cutoff), _ptr(n), _ptr(st), _ptr(key), cap, _ptr(lab))
        if nb == -1:
            return None
        assert nb >= 0, nb
        out = {"n": n[:nb].copy(), "totalDensity": st[:nb, 0].copy(), "centroid": st[:nb, 1:4].copy(),
               "coordCenter": st[:nb, 4:7].copy(), "volume": st[:nb, 7].copy(), "firstKey": key[:nb].copy()}
        if labels:
            out["labels"] = lab
        return out


def test_overlap(a, b):
    a = np.ascontiguousarray(a, dtype=np.int32).reshape(-1, 3)
    b = np.ascontiguousarray(b, dtype=np.int32).reshape(-1, 3)
    return bool(lib().ora_test_overlap(_ptr(a), len(a), _ptr(b), len(b)))


def symmetry_atoms(coords, rot, ortho, bbox_lo, bbox_hi):
    x = np.ascontiguousarray(coords, dtype=np.float64).reshape(-1, 3)
    rot = np.ascontiguousarray(rot, dtype=np.float64).reshape(-1, 12)
    o = np.ascontiguousarray(ortho, dtype=np.float64).reshape(9)
    lo = np.ascontiguousarray(bbox_lo, dtype=np.float64)
    hi = np.ascontiguousarray(bbox_hi, dtype=np.float64)
    cap = 27 * len(rot) * len(x)
    idx = np.zeros(cap, dtype=np.int32)
    sym = np.zeros((cap, 4), dtype=np.int32)
    out = np.zeros((cap, 3))
    n = lib().ora_symmetry_atoms(_ptr(x), len(x), _ptr(rot), len(rot), _ptr(o), _ptr(lo), _ptr(hi), _ptr(idx), _ptr(sym), _ptr(out), cap)
    return idx[:n].copy(), sym[:n].copy(), out[:n].copy()
