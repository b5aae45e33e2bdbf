"""ctypes binding of ``libpdbeda_hip.so`` (C-ABI in ``include/pdbeda.h``).

This is the reference-side stub a pdb_eda maintainer would add:
plain pointers and sizes, no torch types.  There is deliberately NO fallback: if the
shared library is missing, or no gfx950 device is usable, every operation raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDBEDA_LIB") or os.path.join(_HERE, "libpdbeda_hip.so")   # PDBEDA_LIB: A/B builds of the same library

PDBEDA_FLAG_LABELS = 1


class PdbedaError(RuntimeError):
    """A device / library failure (never an entry-level condition): callers must not swallow it."""


class PdbedaTimeout(PdbedaError):
    """The per-entry watchdog of a context expired (PDBEDA_ERR_TIMEOUT): that context is abandoned."""


PDBEDA_ERR_TIMEOUT = -6
PDBEDA_ERR_ARGUMENT = -2


class Geometry(C.Structure):
    _fields_ = [("ncrs", C.c_int32 * 3), ("crs_start", C.c_int32 * 3), ("xyz_interval", C.c_int32 * 3),
                ("map2xyz", C.c_int32 * 3), ("map2crs", C.c_int32 * 3), ("orthogonal", C.c_int32),
                ("ortho", C.c_double * 9), ("deortho", C.c_double * 9), ("origin", C.c_double * 3),
                ("grid_len", C.c_double * 3), ("unit_volume", C.c_double)]


class CloudAtoms(C.Structure):
    """pdbeda_cloud_atoms (include/pdbeda.h): the flattened input of pdbeda_aggregate_cloud."""
    _fields_ = [("n", C.c_int64), ("xyz", C.c_void_p), ("radius", C.c_void_p), ("weight", C.c_void_p), ("residue", C.c_void_p),
                ("alias", C.c_void_p), ("key", C.c_void_p), ("n_keys", C.c_int64), ("bonded_off", C.c_void_p), ("bonded", C.c_void_p),
                ("n_owners", C.c_int64), ("owner_key", C.c_void_p)]


def make_geometry(ncrs, crs_start, xyz_interval, map2xyz, map2crs, orthogonal, ortho, deortho, origin, grid_len, unit_volume):
    g = Geometry()
    for k in range(3):
        g.ncrs[k] = int(ncrs[k])
        g.crs_start[k] = int(crs_start[k])
        g.xyz_interval[k] = int(xyz_interval[k])
        g.map2xyz[k] = int(map2xyz[k])
        g.map2crs[k] = int(map2crs[k])
        g.origin[k] = float(origin[k])
        g.grid_len[k] = float(grid_len[k])
    g.orthogonal = 1 if orthogonal else 0
    o = np.asarray(ortho, dtype=np.float64).reshape(9)
    d = np.asarray(deortho, dtype=np.float64).reshape(9)
    for k in range(9):
        g.ortho[k] = float(o[k])
        g.deortho[k] = float(d[k])
    g.unit_volume = float(unit_volume)
    return g


_lib = None
_lib_lock = threading.Lock()

_p = C.c_void_p
_i64 = C.c_int64
_SIGS = {
    "pdbeda_version": (C.c_char_p, []),
    "pdbeda_device_count": (C.c_int, []),
    "pdbeda_device_pci_address": (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    "pdbeda_ctx_create": (C.c_int, [C.c_int, C.POINTER(_p)]),
    "pdbeda_ctx_create_on_stream": (C.c_int, [C.c_int, _p, C.POINTER(_p)]),
    "pdbeda_ctx_destroy": (C.c_int, [_p]),
    "pdbeda_ctx_synchronize": (C.c_int, [_p]),
    "pdbeda_ctx_stream": (_p, [_p]),
    "pdbeda_ctx_set_timeout": (C.c_int, [_p, C.c_double]),
    "pdbeda_reap_abandoned": (C.c_int64, []),
    "pdbeda_last_error": (C.c_char_p, [_p]),
    "pdbeda_ctx_profile_begin": (C.c_int, [_p]),
    "pdbeda_ctx_profile_end": (C.c_int, [_p, C.c_char_p, _i64]),
    "pdbeda_map_upload": (C.c_int, [_p, _p, C.POINTER(Geometry), C.POINTER(_p)]),
    "pdbeda_map_upload_stats": (C.c_int, [_p, _p, C.POINTER(Geometry), C.POINTER(_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pdbeda_map_upload_file": (C.c_int, [_p, C.c_char_p, _i64, C.c_int, C.POINTER(Geometry), C.POINTER(_p)]),
    "pdbeda_map_upload_file_stats": (C.c_int, [_p, C.c_char_p, _i64, C.c_int, C.POINTER(Geometry), C.POINTER(_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pdbeda_map_from_device": (C.c_int, [_p, _p, C.POINTER(Geometry), C.POINTER(_p)]),
    "pdbeda_map_invalidate": (C.c_int, [_p]),
    "pdbeda_map_free": (C.c_int, [_p]),
    "pdbeda_map_combine": (C.c_int, [_p, _p, C.c_double, C.POINTER(_p)]),
    "pdbeda_map_download": (C.c_int, [_p, _p]),
    "pdbeda_abs_select_hist": (C.c_int, [_p, _p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_ulonglong, C.c_ulonglong, _p]),
    "pdbeda_map_stats": (C.c_int, [_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pdbeda_sum_of_abs": (C.c_int, [_p, C.c_float, C.POINTER(C.c_double)]),
    "pdbeda_point_density": (C.c_int, [_p, _p, _i64, _p]),
    "pdbeda_valid_crs": (C.c_int, [_p, _p, _i64, _p]),
    "pdbeda_crs2xyz": (C.c_int, [_p, _p, _i64, _p]),
    "pdbeda_xyz2crs": (C.c_int, [_p, _p, _i64, _p]),
    "pdbeda_full_blobs": (C.c_int, [_p, C.c_float, C.c_uint32, C.POINTER(_p)]),
    "pdbeda_full_blobs_pm": (C.c_int, [_p, C.c_float, C.c_float, C.c_uint32, C.POINTER(_p), C.POINTER(_p)]),
    "pdbeda_sphere_blobs": (C.c_int, [_p, _p, _p, _i64, _p, _i64, C.c_float, C.POINTER(_p)]),
    "pdbeda_list_blobs": (C.c_int, [_p, _p, _i64, _p, _i64, C.POINTER(_p)]),
    "pdbeda_bloblist_count": (_i64, [_p]),
    "pdbeda_bloblist_num_voxels": (_i64, [_p]),
    "pdbeda_bloblist_stats": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "pdbeda_bloblist_voxels": (C.c_int, [_p, _p, _p]),
    "pdbeda_bloblist_labels": (C.c_int, [_p, _p]),
    "pdbeda_bloblist_free": (C.c_int, [_p]),
    "pdbeda_bloblist_counters": (C.c_int, [_p, _p]),
    "pdbeda_region_sums": (C.c_int, [_p, _p, _p, _i64, _p, _i64, C.c_float, _p, _p, _p, _p]),
    "pdbeda_aggregate_cloud": (C.c_int, [_p, C.POINTER(CloudAtoms), C.c_float, C.c_double, C.POINTER(_p)]),
    "pdbeda_cloud_counts": (C.c_int, [_p, _p, _p]),
    "pdbeda_cloud_atom_rows": (C.c_int, [_p, _p, _p, _p, _p, _p]),
    "pdbeda_cloud_residue_rows": (C.c_int, [_p, _p, _p, _p, _p, _p]),
    "pdbeda_cloud_domain_rows": (C.c_int, [_p, _p, _p, _p, _p, _p]),
    "pdbeda_cloud_owner_states": (C.c_int, [_p, _p]),
    "pdbeda_cloud_free": (C.c_int, [_p]),
    "pdbeda_test_overlap": (C.c_int, [_p, _p, _p, _i64, _p, _p, _i64, _p]),
    "pdbeda_symmetry_atoms": (C.c_int, [_p, _p, _i64, _p, C.c_int32, _p, _p, _p, _p, _p, _p, _i64, C.POINTER(_i64)]),
    "pdbeda_nearest_atom": (C.c_int, [_p, _p, _i64, _p, _i64, _p, _p]),
}
EXPORTED_SYMBOLS = tuple(sorted(_SIGS))


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise PdbedaError("libpdbeda_hip.so is not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`; "
                                  "there is no CPU fallback" % LIB_PATH)
            handle = C.CDLL(LIB_PATH)
            for name, (res, args) in _SIGS.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Context(object):
    """One device + one HIP stream (+ a cache of device arenas)."""

    def __init__(self, device=0, stream=None):
        self._lib = lib()
        h = C.c_void_p()
        rc = self._lib.pdbeda_ctx_create_on_stream(int(device), C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise PdbedaError("pdbeda_ctx_create(device=%d) failed with status %d: no usable gfx950 device "
                              "(the hot path has no CPU fallback)" % (device, rc))
        self._h = h
        self.device = int(device)

    def check(self, rc, what):
        if rc != 0:
            msg = self._lib.pdbeda_last_error(self._h)
            cls = PdbedaTimeout if rc == PDBEDA_ERR_TIMEOUT else PdbedaError
            error = cls("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
            error.code = rc
            raise error

    def set_timeout(self, seconds):
        """Arm (seconds > 0: ONE deadline, now + seconds, for every wait until the next call -- re-arm it when an entry starts) or
        disarm (0) the per-entry watchdog: see pdbeda_ctx_set_timeout in include/pdbeda.h."""
        self.check(self._lib.pdbeda_ctx_set_timeout(self._h, C.c_double(float(seconds))), "pdbeda_ctx_set_timeout")

    def synchronize(self):
        self.check(self._lib.pdbeda_ctx_synchronize(self._h), "pdbeda_ctx_synchronize")

    @property
    def stream(self):
        return self._lib.pdbeda_ctx_stream(self._h)

    def profile_begin(self):
        self.check(self._lib.pdbeda_ctx_profile_begin(self._h), "pdbeda_ctx_profile_begin")

    def profile_end(self):
        """{kernel: (calls, total_ms)} measured with HIP events on this context's stream."""
        buf = C.create_string_buffer(1 << 16)
        self.check(self._lib.pdbeda_ctx_profile_end(self._h, buf, len(buf)), "pdbeda_ctx_profile_end")
        out = {}
        for line in buf.value.decode().splitlines():
            name, calls, ms = line.rsplit(" ", 2)
            out[name] = (int(calls), float(ms))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pdbeda_ctx_destroy(self._h)
            self._h = None

    # -- context-level helpers -------------------------------------------------------
    def test_overlap(self, crs, set_offsets, a_idx, b_idx):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        off = np.ascontiguousarray(set_offsets, dtype=np.int64)
        a = np.ascontiguousarray(a_idx, dtype=np.int32)
        b = np.ascontiguousarray(b_idx, dtype=np.int32)
        out = np.zeros(len(a), dtype=np.uint8)
        self.check(self._lib.pdbeda_test_overlap(self._h, _ptr(crs), _ptr(off), len(off) - 1, _ptr(a), _ptr(b), len(a), _ptr(out)),
                   "pdbeda_test_overlap")
        return out.astype(bool)

    def symmetry_atoms(self, xyz, rot, ortho, bbox_lo, bbox_hi):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        rot = np.ascontiguousarray(rot, dtype=np.float64).reshape(-1, 12)
        ortho = np.ascontiguousarray(ortho, dtype=np.float64).reshape(9)
        lo = np.ascontiguousarray(bbox_lo, dtype=np.float64)
        hi = np.ascontiguousarray(bbox_hi, dtype=np.float64)
        cap = 27 * len(rot) * len(xyz)
        idx = np.zeros(cap, dtype=np.int32)
        sym = np.zeros((cap, 4), dtype=np.int32)
        out = np.zeros((cap, 3), dtype=np.float64)
        n = C.c_int64(0)
        self.check(self._lib.pdbeda_symmetry_atoms(self._h, _ptr(xyz), len(xyz), _ptr(rot), len(rot), _ptr(ortho), _ptr(lo), _ptr(hi),
                                                   _ptr(idx), _ptr(sym), _ptr(out), cap, C.byref(n)), "pdbeda_symmetry_atoms")
        k = n.value
        return idx[:k].copy(), sym[:k].copy(), out[:k].copy()

    def nearest_atom(self, centroids, atom_xyz):
        cen = np.ascontiguousarray(centroids, dtype=np.float64).reshape(-1, 3)
        at = np.ascontiguousarray(atom_xyz, dtype=np.float64).reshape(-1, 3)
        idx = np.zeros(len(cen), dtype=np.int64)
        dist = np.zeros(len(cen), dtype=np.float64)
        self.check(self._lib.pdbeda_nearest_atom(self._h, _ptr(cen), len(cen), _ptr(at), len(at), _ptr(idx), _ptr(dist)),
                   "pdbeda_nearest_atom")
        return idx, dist

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = {}
_default_lock = threading.Lock()


def device_local_cpus(device=0):
    """Host cores on the NUMA node of GPU ``device`` (sysfs ``local_cpulist`` of its PCI function), or None when the
    platform does not say (no sysfs, a single node, a container that hides it)."""
    buf = C.create_string_buffer(64)
    if lib().pdbeda_device_pci_address(int(device), buf, 64) != 0:
        return None
    address = buf.value.decode().strip().lower()
    try:
        with open("/sys/bus/pci/devices/%s/local_cpulist" % address) as fh:
            text = fh.read().strip()
    except OSError:
        return None
    cpus = set()
    for part in text.split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus or None


def pin_to_device(device=0):
    """Restrict the calling thread (and the threads and processes it starts afterwards) to the cores of the GPU's NUMA node:
    file reads, parsing and the pageable side of every upload then stay on the socket the GPU hangs off.  Returns the number
    of cores kept, or 0 when nothing was changed."""
    if not hasattr(os, "sched_setaffinity"):
        return 0
    local = device_local_cpus(device)
    if not local:
        return 0
    keep = local & os.sched_getaffinity(0)
    if not keep or keep == os.sched_getaffinity(0):
        return 0
    os.sched_setaffinity(0, keep)
    return len(keep)


def default_context(device=None):
    """Per-thread default context (device from PDBEDA_DEVICE / LOCAL_RANK, else 0)."""
    if device is None:
        device = int(os.environ.get("PDBEDA_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    key = (threading.get_ident(), device)
    with _default_lock:
        ctx = _default_ctx.get(key)
        if ctx is None:
            ctx = Context(device)
            _default_ctx[key] = ctx
    return ctx


class BlobList(object):
    """Result of a labelling call; statistics and voxel membership are fetched on demand."""

    def __init__(self, ctx, handle, keepalive=None):
        self._ctx = ctx
        self._h = handle
        self._keep = keepalive
        self._stats = None
        self._vox = None

    def __len__(self):
        n = self._ctx._lib.pdbeda_bloblist_count(self._h)
        if n < 0:
            self._ctx.check(int(n), "pdbeda_bloblist_count")
        return int(n)

    def stats(self):
        if self._stats is None:
            n = len(self)
            st = {"n": np.zeros(n, np.int64), "totalDensity": np.zeros(n, np.float64), "centroid": np.zeros((n, 3), np.float64),
                  "coordCenter": np.zeros((n, 3), np.float64), "volume": np.zeros(n, np.float64), "firstKey": np.zeros(n, np.int64),
                  "group": np.zeros(n, np.int32)}
            self._ctx.check(self._ctx._lib.pdbeda_bloblist_stats(self._h, _ptr(st["n"]), _ptr(st["totalDensity"]), _ptr(st["centroid"]),
                                                                 _ptr(st["coordCenter"]), _ptr(st["volume"]), _ptr(st["firstKey"]),
                                                                 _ptr(st["group"])), "pdbeda_bloblist_stats")
            self._stats = st
        return self._stats

    def voxels(self):
        """(crs[N,3] int32 grouped by blob, offsets[count+1])."""
        if self._vox is None:
            nv = self._ctx._lib.pdbeda_bloblist_num_voxels(self._h)
            if nv < 0:
                self._ctx.check(int(nv), "pdbeda_bloblist_num_voxels")
            crs = np.zeros((int(nv), 3), dtype=np.int32)
            off = np.zeros(len(self) + 1, dtype=np.int64)
            self._ctx.check(self._ctx._lib.pdbeda_bloblist_voxels(self._h, _ptr(crs), _ptr(off)), "pdbeda_bloblist_voxels")
            self._vox = (crs, off)
        return self._vox

    def voxels_of(self, i):
        crs, off = self.voxels()
        return crs[off[i]:off[i + 1]]

    def labels(self, shape):
        out = np.zeros(shape, dtype=np.int32)
        self._ctx.check(self._ctx._lib.pdbeda_bloblist_labels(self._h, _ptr(out)), "pdbeda_bloblist_labels")
        return out

    def counters(self):
        out = np.zeros(8, dtype=np.int64)
        self._ctx.check(self._ctx._lib.pdbeda_bloblist_counters(self._h, _ptr(out)), "pdbeda_bloblist_counters")
        return {"run_ids": int(out[0]), "component_ids": int(out[1]), "blobs": int(out[3]), "unit_tiles_runs": int(out[4]), "wide_tiles": int(out[5]), "unit_tiles_comps": int(out[6]),
                "reruns": int(out[2]), "arena_bytes": int(out[7])}

    def free(self):
        if self._h is not None and self._ctx._h:
            self._ctx._lib.pdbeda_bloblist_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceMap(object):
    """A density grid resident in HBM + its unit-cell basis."""

    def __init__(self, ctx, density, geometry, device_ptr=None):
        self._ctx = ctx
        self._geom = geometry
        h = C.c_void_p()
        if device_ptr is not None:
            rc = ctx._lib.pdbeda_map_from_device(ctx._h, C.c_void_p(device_ptr), C.byref(geometry), C.byref(h))
            self._keep = density
        else:
            grid = np.ascontiguousarray(density, dtype=np.float32)
            assert grid.size == geometry.ncrs[0] * geometry.ncrs[1] * geometry.ncrs[2], "grid size does not match header ncrs"
            mean, std = C.c_double(), C.c_double()      # (with the map's mean / std from the same wait: every caller asks for them next)
            rc = ctx._lib.pdbeda_map_upload_stats(ctx._h, _ptr(grid), C.byref(geometry), C.byref(h), C.byref(mean), C.byref(std))
            if rc == 0:
                self._file_stats = (mean.value, std.value)
            self._keep = None
        ctx.check(rc, "pdbeda_map_upload")
        self._h = h
        self.unique_shape = tuple(min(geometry.ncrs[k], geometry.xyz_interval[geometry.map2crs[k]]) for k in (2, 1, 0))

    def invalidate(self):
        """The caller rewrote a borrowed device buffer in place: drop what the library cached about its contents."""
        self._ctx.check(self._ctx._lib.pdbeda_map_invalidate(self._h), "pdbeda_map_invalidate")
        self.__dict__.pop("_file_stats", None)

    @classmethod
    def from_file(cls, ctx, path, offset, byteswap, geometry):
        """The float32 grid stored at byte ``offset`` of ``path`` straight into HBM through the process's upload engine (three reader
        threads with pinned chunks and copy streams of their own: file reads and PCIe copies overlap; no host copy of the map is kept)."""
        self = cls.__new__(cls)
        self._ctx, self._geom, self._keep = ctx, geometry, None
        h = C.c_void_p()
        mean, std = C.c_double(), C.c_double()
        try:
            # (with the map's mean / std from the same wait: every caller asks for them next)
            ctx.check(ctx._lib.pdbeda_map_upload_file_stats(ctx._h, os.fsencode(path), int(offset), 1 if byteswap else 0, C.byref(geometry), C.byref(h),
                                                            C.byref(mean), C.byref(std)), "pdbeda_map_upload_file")
            self._file_stats = (mean.value, std.value)
        except PdbedaError as error:
            if getattr(error, "code", 0) == PDBEDA_ERR_ARGUMENT:       # the FILE is at fault (unreadable, truncated): an entry-level condition
                raise OSError(str(error))
            raise
        self._h = h
        self.unique_shape = tuple(min(geometry.ncrs[k], geometry.xyz_interval[geometry.map2crs[k]]) for k in (2, 1, 0))
        return self

    @classmethod
    def combine(cls, a, b, alpha):
        """A new resident map: float32(double(a) + alpha * double(b)) per voxel, on the geometry of ``a`` (device side)."""
        self = cls.__new__(cls)
        self._ctx, self._geom, self._keep, self.unique_shape = a._ctx, a._geom, None, a.unique_shape
        h = C.c_void_p()
        a._ctx.check(a._ctx._lib.pdbeda_map_combine(a._h, b._h, C.c_double(alpha), C.byref(h)), "pdbeda_map_combine")
        self._h = h
        return self

    def abs_order_statistics(self, other, alpha, cut_a, cut_b, which, ranks=None):
        """Exact order statistics of |a| (which = 0) or |a + alpha * other| (which = 1) over the voxels of the unique box with
        |a| < cut_a and |a + alpha other| < cut_b: a radix select, 16 bits per device pass.  ranks=None -> the number of
        selected voxels; else the values at those 0-based ranks (ascending), as float64."""
        hist = np.zeros(65536, dtype=np.uint32)
        oh = other._h if other is not None else None

        def digit(shift, prefix, mask):
            self._ctx.check(self._ctx._lib.pdbeda_abs_select_hist(self._h, oh, C.c_double(alpha), C.c_double(cut_a), C.c_double(cut_b), which, shift,
                                                                  C.c_ulonglong(prefix), C.c_ulonglong(mask), _ptr(hist)), "pdbeda_abs_select_hist")
            return hist.astype(np.int64)
        bits = 64 if which else 32
        if ranks is None:
            return int(digit(bits - 16, 0, 0).sum())
        out = []
        for rank in ranks:
            prefix, mask, remaining = 0, 0, int(rank)
            for shift in range(bits - 16, -1, -16):
                cs = np.cumsum(digit(shift, prefix, mask))
                d = int(np.searchsorted(cs, remaining, side="right"))
                remaining -= int(cs[d - 1]) if d else 0
                prefix |= d << shift
                mask |= 0xffff << shift
            out.append(np.array([prefix], dtype=np.uint64).view(np.float64)[0] if which else float(np.array([prefix], dtype=np.uint32).view(np.float32)[0]))
        return out

    def download(self):
        """The float32 grid [ns][nr][nc] copied back to the host."""
        g = self._geom
        out = np.empty((g.ncrs[2], g.ncrs[1], g.ncrs[0]), dtype=np.float32)
        self._ctx.check(self._ctx._lib.pdbeda_map_download(self._h, _ptr(out)), "pdbeda_map_download")
        return out

    # -- reductions ------------------------------------------------------------------
    def stats(self):
        known = self.__dict__.get("_file_stats")           # (a map uploaded from a file brought them along)
        if known is not None:
            return known
        mean, std = C.c_double(), C.c_double()
        self._ctx.check(self._ctx._lib.pdbeda_map_stats(self._h, C.byref(mean), C.byref(std)), "pdbeda_map_stats")
        return mean.value, std.value

    def sum_of_abs(self, cutoff):
        out = C.c_double()
        self._ctx.check(self._ctx._lib.pdbeda_sum_of_abs(self._h, C.c_float(cutoff), C.byref(out)), "pdbeda_sum_of_abs")
        return out.value

    # -- points ----------------------------------------------------------------------
    def point_density(self, crs):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        out = np.zeros(len(crs), dtype=np.float64)
        self._ctx.check(self._ctx._lib.pdbeda_point_density(self._h, _ptr(crs), len(crs), _ptr(out)), "pdbeda_point_density")
        return out

    def valid_crs(self, crs):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        out = np.zeros(len(crs), dtype=np.uint8)
        self._ctx.check(self._ctx._lib.pdbeda_valid_crs(self._h, _ptr(crs), len(crs), _ptr(out)), "pdbeda_valid_crs")
        return out.astype(bool)

    def crs2xyz(self, crs):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        out = np.zeros((len(crs), 3), dtype=np.float64)
        self._ctx.check(self._ctx._lib.pdbeda_crs2xyz(self._h, _ptr(crs), len(crs), _ptr(out)), "pdbeda_crs2xyz")
        return out

    def xyz2crs(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        out = np.zeros((len(xyz), 3), dtype=np.int32)
        self._ctx.check(self._ctx._lib.pdbeda_xyz2crs(self._h, _ptr(xyz), len(xyz), _ptr(out)), "pdbeda_xyz2crs")
        return out

    # -- labelling -------------------------------------------------------------------
    def full_blobs(self, cutoff, labels=False):
        h = C.c_void_p()
        self._ctx.check(self._ctx._lib.pdbeda_full_blobs(self._h, C.c_float(cutoff), PDBEDA_FLAG_LABELS if labels else 0, C.byref(h)),
                        "pdbeda_full_blobs")
        return BlobList(self._ctx, h, self)

    def full_blobs_pm(self, cutoff_pos, cutoff_neg, labels=False):
        g, r = C.c_void_p(), C.c_void_p()
        self._ctx.check(self._ctx._lib.pdbeda_full_blobs_pm(self._h, C.c_float(cutoff_pos), C.c_float(cutoff_neg),
                                                            PDBEDA_FLAG_LABELS if labels else 0, C.byref(g), C.byref(r)),
                        "pdbeda_full_blobs_pm")
        return BlobList(self._ctx, g, self), BlobList(self._ctx, r, self)

    def sphere_blobs(self, xyz, radii, group_offsets, cutoff):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        off = np.ascontiguousarray(group_offsets, dtype=np.int64)
        h = C.c_void_p()
        self._ctx.check(self._ctx._lib.pdbeda_sphere_blobs(self._h, _ptr(xyz), _ptr(radii), len(xyz), _ptr(off), len(off) - 1,
                                                           C.c_float(cutoff), C.byref(h)), "pdbeda_sphere_blobs")
        return BlobList(self._ctx, h, self)

    def list_blobs(self, crs, group_offsets=None):
        crs = np.ascontiguousarray(crs, dtype=np.int32).reshape(-1, 3)
        off = np.array([0, len(crs)], dtype=np.int64) if group_offsets is None else np.ascontiguousarray(group_offsets, dtype=np.int64)
        h = C.c_void_p()
        self._ctx.check(self._ctx._lib.pdbeda_list_blobs(self._h, _ptr(crs), len(crs), _ptr(off), len(off) - 1, C.byref(h)),
                        "pdbeda_list_blobs")
        return BlobList(self._ctx, h, self)

    def list_stats(self, crs):
        """Statistics of one explicit voxel set as a single blob (DensityBlob.fromCrsList).

        The set need not be connected: the per-component sums of the device job are
        combined here (a handful of rows)."""
        bl = self.list_blobs(crs)
        st = bl.stats()
        n = st["n"].astype(np.float64)
        tot = float(st["totalDensity"].sum())
        # a component whose own total density is 0 has no density-weighted centroid (0 / 0 on the device): it contributes
        # nothing to the weighted sum -- leaving it in would poison the whole centroid with NaN * 0
        w = st["totalDensity"]
        has = w != 0
        cen = (st["centroid"][has] * w[has, None]).sum(axis=0) / tot if (len(n) and tot != 0) else np.full(3, np.nan)
        cc = (st["coordCenter"] * n[:, None]).sum(axis=0) / n.sum() if len(n) else np.full(3, np.nan)
        return {"totalDensity": tot, "centroid": cen, "coordCenter": cc, "volume": float(st["volume"].sum()), "n": int(n.sum())}

    def region_sums(self, xyz, radii, group_offsets, cutoff):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        radii = np.ascontiguousarray(radii, dtype=np.float32)
        off = np.ascontiguousarray(group_offsets, dtype=np.int64)
        ng = len(off) - 1
        pos, neg = np.zeros(ng), np.zeros(ng)
        cnt = np.zeros(ng, dtype=np.int64)
        valid = np.zeros(ng, dtype=np.uint8)
        self._ctx.check(self._ctx._lib.pdbeda_region_sums(self._h, _ptr(xyz), _ptr(radii), len(xyz), _ptr(off), ng, C.c_float(cutoff),
                                                          _ptr(pos), _ptr(neg), _ptr(cnt), _ptr(valid)), "pdbeda_region_sums")
        return pos, neg, cnt, valid.astype(bool)

    def aggregate_cloud(self, xyz, radius, weight, residue, alias, key, bonded_off, bonded, owner_key, cutoff, min_cloud_electrons):
        """pdbeda_aggregate_cloud: everything of aggregateCloud that touches voxels, in one call.  Returns a dict of arrays:
        atoms (eligible index, totalDensity, voxels, centroid, distance), residue / domain cloud rows, owner states, totals."""
        lib, ctx = self._ctx._lib, self._ctx
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        arrs = {"radius": np.ascontiguousarray(radius, dtype=np.float32), "weight": np.ascontiguousarray(weight, dtype=np.float64),
                "residue": np.ascontiguousarray(residue, dtype=np.int32), "alias": np.ascontiguousarray(alias, dtype=np.int32),
                "key": np.ascontiguousarray(key, dtype=np.int32), "bonded_off": np.ascontiguousarray(bonded_off, dtype=np.int64),
                "bonded": np.ascontiguousarray(bonded, dtype=np.int32), "owner_key": np.ascontiguousarray(owner_key, dtype=np.int32)}
        n = len(xyz)
        assert all(len(arrs[k]) == n for k in ("radius", "weight", "residue", "alias", "key"))
        at = CloudAtoms(n, xyz.ctypes.data, *[arrs[k].ctypes.data for k in ("radius", "weight", "residue", "alias", "key")],
                        len(arrs["bonded_off"]) - 1, arrs["bonded_off"].ctypes.data, arrs["bonded"].ctypes.data,
                        len(arrs["owner_key"]), arrs["owner_key"].ctypes.data)
        h = C.c_void_p()
        ctx.check(lib.pdbeda_aggregate_cloud(self._h, C.byref(at), C.c_float(cutoff), C.c_double(min_cloud_electrons), C.byref(h)), "pdbeda_aggregate_cloud")
        try:
            counts, totals = np.zeros(4, np.int64), np.zeros(4, np.float64)
            ctx.check(lib.pdbeda_cloud_counts(h, _ptr(counts), _ptr(totals)), "pdbeda_cloud_counts")
            na, nr, nd, no = (int(v) for v in counts)
            out = {"numVoxels": int(totals[0]), "totalElectrons": float(totals[1]), "totalDensity": float(totals[2]), "centroidDistanceCutoff": float(totals[3]),
                   "atom": np.zeros(na, np.int32), "atom_total": np.zeros(na), "atom_n": np.zeros(na, np.int64), "atom_centroid": np.zeros((na, 3)),
                   "atom_distance": np.zeros(na), "owner_state": np.zeros(no, np.uint8)}
            ctx.check(lib.pdbeda_cloud_atom_rows(h, _ptr(out["atom"]), _ptr(out["atom_total"]), _ptr(out["atom_n"]), _ptr(out["atom_centroid"]),
                                                 _ptr(out["atom_distance"])), "pdbeda_cloud_atom_rows")
            for tag, cnt, fn in (("res", nr, lib.pdbeda_cloud_residue_rows), ("dom", nd, lib.pdbeda_cloud_domain_rows)):
                t = {"residue": np.zeros(cnt, np.int32), "total": np.zeros(cnt), "n": np.zeros(cnt, np.int64), "electrons": np.zeros(cnt), "centroid": np.zeros((cnt, 3))}
                ctx.check(fn(h, _ptr(t["residue"]), _ptr(t["total"]), _ptr(t["n"]), _ptr(t["electrons"]), _ptr(t["centroid"])), "pdbeda_cloud_rows")
                out[tag] = t
            ctx.check(lib.pdbeda_cloud_owner_states(h, _ptr(out["owner_state"])), "pdbeda_cloud_owner_states")
        finally:
            lib.pdbeda_cloud_free(h)
        return out

    def test_overlap(self, a, b):
        crs = np.concatenate([np.asarray(a, np.int32).reshape(-1, 3), np.asarray(b, np.int32).reshape(-1, 3)])
        off = np.array([0, len(a), len(a) + len(b)], dtype=np.int64)
        return bool(self._ctx.test_overlap(crs, off, [0], [1])[0])

    def free(self):
        if getattr(self, "_h", None) is not None and self._ctx._h:
            self._ctx._lib.pdbeda_map_free(self._h)
        self._h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
