"""Seeded synthetic CCP4 maps and model structures.

Real PDB entries / PDBe maps cannot be downloaded (no network), so tests, golden
fixtures and bench.py all run on these generators.  Everything here is host-side
input preparation; nothing is on the hot path.

The CCP4 byte layout written by :func:`ccp4_bytes` is the one the reference parses in
``pdb_eda/ccp4.py:145-150`` (56 header words + 800 label bytes, mode 2 data).
"""
import numpy as np

__all__ = ["ccp4_bytes", "ccp4_header_bytes", "synthetic_params", "noise_grid", "MapSpec", "chain_structure", "gaussian_sum_grid", "smooth_noise",
           "sweep_param_sets", "SyntheticEntryFiles", "write_entry_files"]


class MapSpec(object):
    """Geometry of a synthetic map: everything the CCP4 header carries."""

    def __init__(self, ncrs, cell=None, angles=(90.0, 90.0, 90.0), interval=None, crs_start=(0, 0, 0),
                 axis_order=(1, 2, 3), spacing=0.4):
        self.ncrs = tuple(int(x) for x in ncrs)                # columns, rows, sections
        self.axis_order = tuple(int(x) for x in axis_order)    # MAPC, MAPR, MAPS (1=X,2=Y,3=Z)
        self.crs_start = tuple(int(x) for x in crs_start)
        if interval is None:
            # interval is per xyz axis; default = ncrs of the crs axis that maps onto it
            interval = [0, 0, 0]
            for crs_axis, xyz_axis in enumerate(self.axis_order):
                interval[xyz_axis - 1] = self.ncrs[crs_axis]
        self.interval = tuple(int(x) for x in interval)
        if cell is None:
            cell = [self.interval[i] * spacing for i in range(3)]
        self.cell = tuple(float(np.float32(x)) for x in cell)
        self.angles = tuple(float(np.float32(x)) for x in angles)


def ccp4_header_bytes(spec, stats=(0.0, 0.0, 0.0, 0.0), big_endian=False, n_symmetry_bytes=0, origin_em=(0.0, 0.0, 0.0)):
    """The 1024-byte CCP4 header for ``spec``; stats = (min, max, mean, rms)."""
    e = ">" if big_endian else "<"
    words = np.zeros(56, dtype=e + "i4")
    fl = words.view(e + "f4")
    words[0:3] = spec.ncrs
    words[3] = 2
    words[4:7] = spec.crs_start
    words[7:10] = spec.interval
    fl[10:13] = spec.cell
    fl[13:16] = spec.angles
    words[16:19] = spec.axis_order
    fl[19], fl[20], fl[21] = stats[0], stats[1], stats[2]
    words[22] = 1
    words[23] = n_symmetry_bytes
    fl[49:52] = origin_em
    fl[54] = stats[3]
    words[55] = 1
    head = bytearray(words.tobytes())
    head[208:212] = b"MAP "
    head[212:216] = bytes([0x11, 0x11, 0, 0]) if big_endian else bytes([0x44, 0x41, 0, 0])
    labels = b"synthetic map (pdb_eda_amd.synthetic)".ljust(800, b" ")
    return bytes(head) + labels


def ccp4_bytes(spec, grid, big_endian=False, symmetry_bytes=b"", origin_em=(0.0, 0.0, 0.0)):
    """Serialise ``grid`` ([ns][nr][nc] float32) with the header described by ``spec``."""
    grid = np.ascontiguousarray(grid, dtype=np.float32)
    nc, nr, ns = spec.ncrs
    assert grid.shape == (ns, nr, nc), (grid.shape, spec.ncrs)
    e = ">" if big_endian else "<"
    stats = (grid.min(), grid.max(), grid.mean(dtype=np.float64), grid.std(dtype=np.float64)) if grid.size else (0.0,) * 4
    head = ccp4_header_bytes(spec, stats, big_endian, len(symmetry_bytes), origin_em)
    return head + bytes(symmetry_bytes) + grid.astype(e + "f4").tobytes()


def smooth_noise(shape, seed, sigma_voxels=1.5):
    """Gaussian-filtered white noise, periodic, float32 -- the config-2/4 map generator."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    g = rng.standard_normal(shape, dtype=np.float32)
    if sigma_voxels > 0:
        g = gaussian_filter(g, sigma=sigma_voxels, mode="wrap")
    return np.ascontiguousarray(g, dtype=np.float32)


def noise_grid(spec, seed, sigma_voxels=1.5):
    nc, nr, ns = spec.ncrs
    return smooth_noise((ns, nr, nc), seed, sigma_voxels)


# ---- synthetic model: poly-ALA random-walk chain ---------------------------------------

_ALA = (("N", "N", (0.0, 0.0, 0.0)), ("CA", "C", (1.458, 0.0, 0.0)), ("C", "C", (2.009, 1.420, 0.0)),
        ("O", "O", (1.251, 2.390, 0.0)), ("CB", "C", (1.988, -0.773, -1.199)))


def chain_structure(n_residues, seed, box_lo, box_hi, hetero_every=0, zero_occupancy_every=0, resname="ALA"):
    """Return a :class:`pdb_eda_amd.structure.Structure` of ``n_residues`` ALA residues.

    CA positions follow a 3.8 A random walk reflected inside [box_lo, box_hi]; the other
    backbone atoms + CB are placed with a random rigid rotation of an ideal residue.
    """
    from .structure import Structure, Model, Chain, Residue, Atom
    rng = np.random.default_rng(seed)
    lo = np.asarray(box_lo, dtype=np.float64) + 3.0
    hi = np.asarray(box_hi, dtype=np.float64) - 3.0
    st = Structure("synth")
    model = Model(0, st)
    chain = Chain("A", model)
    pos = (lo + hi) / 2 + rng.uniform(-1, 1, 3)
    serial = 0
    for i in range(n_residues):
        step = rng.standard_normal(3)
        step *= 3.8 / np.linalg.norm(step)
        new = pos + step
        for k in range(3):
            if new[k] < lo[k] or new[k] > hi[k]:
                new[k] = pos[k] - step[k]
        pos = new
        q = rng.standard_normal(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                        [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                        [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        het = " " if not (hetero_every and (i + 1) % hetero_every == 0) else "H_LIG"
        res = Residue((het, i + 1, " "), resname, chain)
        for name, element, off in _ALA:
            serial += 1
            coord = (pos + rot.dot(np.asarray(off) - np.asarray(_ALA[1][2]))).astype(np.float32)
            occ = 0.0 if (zero_occupancy_every and serial % zero_occupancy_every == 0) else 1.0
            bfac = float(np.round(rng.uniform(8.0, 45.0), 2))
            Atom(name, coord, occ, bfac, element, res, serial)
    return st


def gaussian_sum_grid(header, structure, electrons, sigma=0.55, noise=0.02, seed=0, scale=0.05):
    """2Fo-Fc-like grid: sum over atoms of isotropic Gaussians weighted by electrons.

    ``header`` is a :class:`pdb_eda_amd.ccp4.DensityHeader`; the Gaussian is evaluated on
    the stored grid in a cubic neighbourhood of each atom (periodic in the cell).
    ``electrons`` maps "RES_ATOM" names to electron counts (the reference's
    ``full_atom_name_map_electrons`` table, loaded by the caller at run time).
    """
    nc, nr, ns = header.ncrs
    rng = np.random.default_rng(seed)
    grid = (rng.standard_normal((ns, nr, nc)) * noise).astype(np.float64)
    reach = int(np.ceil(4 * sigma / min(header.gridLength))) + 1
    off = np.arange(-reach, reach + 1)
    oc, orr, os_ = np.meshgrid(off, off, off, indexing="ij")
    for atom in structure.get_atoms():
        key = atom.parent.resname.strip() + "_" + atom.name
        w = float(electrons.get(key, 6.0)) * atom.get_occupancy()
        if w == 0:
            continue
        c0, r0, s0 = header.xyz2crsCoord(atom.coord)
        cc, rr, ss = c0 + oc, r0 + orr, s0 + os_
        crs = np.stack([cc, rr, ss], axis=-1).reshape(-1, 3)
        xyz = header.crs2xyz_array(crs)
        d2 = ((xyz - atom.coord.astype(np.float64)) ** 2).sum(axis=1)
        val = w * scale * np.exp(-d2 / (2 * sigma * sigma))
        ci = np.mod(crs[:, 0], header.crsInterval[0])
        ri = np.mod(crs[:, 1], header.crsInterval[1])
        si = np.mod(crs[:, 2], header.crsInterval[2])
        ok = (ci < nc) & (ri < nr) & (si < ns)
        np.add.at(grid, (si[ok], ri[ok], ci[ok]), val[ok])
    ic, ir, is_ = header.crsInterval
    if nc > ic:
        grid[:, :, ic:] = grid[:, :, :nc - ic]
    if nr > ir:
        grid[:, ir:, :] = grid[:, :nr - ir, :]
    if ns > is_:
        grid[is_:, :, :] = grid[:ns - is_, :, :]
    return grid.astype(np.float32)


def synthetic_params():
    """A small, made-up analysis parameter table (same schema as the reference's
    ``conf/optimized_params.json``: radii, slopes, electrons, atom types, bonded atoms) covering the
    poly-ALA model of :func:`chain_structure`.  The values are NOT the reference's."""
    types = {"ALA_N": "N.syn.amide", "ALA_CA": "C.syn.alpha", "ALA_C": "C.syn.carbonyl", "ALA_O": "O.syn.carbonyl", "ALA_CB": "C.syn.methyl"}
    return {
        "radii": {"N.syn.amide": 0.78, "C.syn.alpha": 0.81, "C.syn.carbonyl": 0.74, "O.syn.carbonyl": 0.86, "C.syn.methyl": 0.9},
        "slopes": {"N.syn.amide": -0.52, "C.syn.alpha": -0.61, "C.syn.carbonyl": -0.55, "O.syn.carbonyl": -0.48, "C.syn.methyl": -0.7},
        "full_atom_name_map_atom_type": types,
        "full_atom_name_map_electrons": {"ALA_N": 8.0, "ALA_CA": 7.0, "ALA_C": 6.0, "ALA_O": 8.0, "ALA_CB": 9.0},
        "bonded_atoms": {"ALA_N": ["ALA_CA"], "ALA_CA": ["ALA_N", "ALA_C", "ALA_CB"], "ALA_C": ["ALA_CA", "ALA_O", "ALA_OXT"],
                         "ALA_O": ["ALA_C"], "ALA_CB": ["ALA_CA"]},
        "leaving_atoms": [],
    }


def sweep_param_sets():
    """Parameter sets of a short optimise-mode radius sweep (BASELINE configs[4]; optimizeParams.py:232-243 changes ONE atom
    type's radius per iteration and carries the slopes of the last accepted iteration): the base table, then three
    steps.  Shared by the golden generator (reference run) and the tests / bench (MI355X run).  Made-up values."""
    base = synthetic_params()
    steps = [("C.syn.methyl", +0.10, {}), ("O.syn.carbonyl", -0.08, {"O.syn.carbonyl": -0.41}), ("N.syn.amide", +0.05, {"C.syn.alpha": -0.66, "N.syn.amide": -0.5})]
    sets = [base]
    for atom_type, delta, slopes in steps:
        prev = sets[-1]
        radii = dict(prev["radii"])
        radii[atom_type] = round(radii[atom_type] + delta, 6)
        sets.append({**prev, "radii": radii, "slopes": {**prev["slopes"], **slopes}})
    return sets


# ---- one synthetic entry in memory: the generator behind bench.py's analysis leg, the entry files of the multiple-structure leg and
#      the reference goldens at the BASELINE sizes (tests/golden/make_golden_big.py keeps only seeds + the reference's numbers) ----
BIG_CASES = {
    # name: (ncrs, residues of the poly-ALA chain (5 atoms each), seed, grid spacing in Angstrom) -- SURVEY.md 8d
    "c0_1stp_like": ((100, 108, 96), 200, 31, 0.45),     # BASELINE configs[0] stand-in: ~100^3, ~1 k atoms
    "c2_bench_entry": ((128, 128, 128), 400, 5, 0.5),    # bench.py's analysis_entry: 128^3, 2 000 atoms ("~2 A entry")
    "c3_multiple_entry": ((200, 200, 200), 100, 0, 0.5),  # one entry of configs[3]: 200^3, 500 atoms
    "c5_hex_perm": ((128, 128, 128), 400, 11, 0.5),      # round 5: a NON-orthogonal cell at a BASELINE size (gamma = 120, Y/X/Z axis order, crsStart != 0)
}
# what a case's CCP4 header carries beyond (ncrs, spacing): a case without an entry here is an orthogonal cube whose cell is its grid
BIG_CASE_SPECS = {
    "c5_hex_perm": dict(interval=(144, 160, 136), crs_start=(-8, 12, 6), axis_order=(2, 1, 3), cell=(72.0, 80.0, 68.0), angles=(90.0, 90.0, 120.0)),
}


def cube_entry(ncrs, n_residues, seed, spacing=0.5, spec_kwargs=None):
    """(spec, header, structure, params, 2Fo-Fc grid, Fo-Fc grid, rotation matrices) of one synthetic entry.
    ``spec_kwargs``: the rest of the map's geometry (``BIG_CASE_SPECS``: cell, angles, interval, crs_start, axis_order)."""
    from . import ccp4
    spec = MapSpec(ncrs=tuple(ncrs), spacing=spacing, **(spec_kwargs or {}))
    header = ccp4.DensityHeader.fromFileHeader(ccp4_header_bytes(spec))
    if spec_kwargs:
        nc, nr, ns = spec.ncrs
        corners = np.array([header.crs2xyzCoord([c, r, s]) for c in (6, nc - 7) for r in (6, nr - 7) for s in (6, ns - 7)], dtype=np.float64)
        lo, hi = corners.min(axis=0), corners.max(axis=0)
        if not header.orthogonal:       # keep the chain inside the skewed cell: shrink the box around its centre
            mid = (lo + hi) / 2
            lo, hi = mid - (hi - lo) / 4, mid + (hi - lo) / 4
    else:
        lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([n - 7 for n in ncrs]))
    st = chain_structure(n_residues, seed, lo, hi, hetero_every=9, zero_occupancy_every=37)
    params = synthetic_params()
    dens = gaussian_sum_grid(header, st, params["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=seed)
    diff = (noise_grid(spec, seed + 100, 1.2) * 0.12).astype(np.float32)
    rot = [np.hstack([np.eye(3), np.zeros((3, 1))]),
           np.array([[-1.0, 0, 0, 0.5 * header.xlength], [0, -1.0, 0, 0], [0, 0, 1.0, 0.5 * header.zlength]])]
    return spec, header, st, params, dens, diff, rot


# ---- synthetic entries on disk (BASELINE configs[3] / [4]: "1 000 synthetic 200^3 grids ... ~500 atoms each") ---------------------

def write_entry_files(directory, tag, edge, n_residues, seed, spacing=0.5, as_paths=False):
    """Generate ONE synthetic entry -- a 2Fo-Fc-like map (sum of atomic Gaussians + noise) and an Fo-Fc-like map (filtered noise)
    on an ``edge``^3 grid around a poly-ALA chain of ``n_residues`` -- and write its two CCP4 files.  Returns the picklable
    loader (:class:`SyntheticEntryFiles`) the worker processes of a pool use."""
    import os
    from . import ccp4
    spec = MapSpec(ncrs=(edge, edge, edge), spacing=spacing)
    header = ccp4.DensityHeader.fromFileHeader(ccp4_header_bytes(spec))
    lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([edge - 7] * 3))
    st = chain_structure(n_residues, seed, lo, hi, hetero_every=9, zero_occupancy_every=37)
    dens = gaussian_sum_grid(header, st, synthetic_params()["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=seed)
    diff = (noise_grid(spec, seed + 100, 1.2) * 0.12).astype(np.float32)
    paths = [os.path.join(directory, "%s%s.ccp4" % (tag, suffix)) for suffix in ("", "_diff")]
    for path, grid in zip(paths, (dens, diff)):
        with open(path, "wb") as fh:
            fh.write(ccp4_bytes(spec, grid))
    return SyntheticEntryFiles(paths[0], paths[1], n_residues, seed, edge, spacing, as_paths)


class SyntheticEntryFiles(object):
    """Loader of one synthetic entry for ``multipleStructures.Entry``: reads the two CCP4 files (as a real run reads the files
    ``fromPDBid`` downloaded) and rebuilds the model structure from its seed (cached per process: the files are the per-entry
    cost that matters, a coordinate file parser is outside the path).  Picklable: pools hand it to worker processes."""
    _structures = {}

    def __init__(self, density_path, diff_path, n_residues, seed, edge, spacing, as_paths=False):
        self.density_path, self.diff_path = density_path, diff_path
        self.n_residues, self.seed, self.edge, self.spacing = n_residues, seed, edge, spacing
        self.as_paths = as_paths          # hand the file PATHS to the analysis (local-mirror mode) instead of their bytes

    def structure(self):
        from . import ccp4, structure
        key = (self.n_residues, self.seed, self.edge, self.spacing)
        if key not in SyntheticEntryFiles._structures:
            spec = MapSpec(ncrs=(self.edge,) * 3, spacing=self.spacing)
            header = ccp4.DensityHeader.fromFileHeader(ccp4_header_bytes(spec))
            lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([self.edge - 7] * 3))
            st = chain_structure(self.n_residues, self.seed, lo, hi, hetero_every=9, zero_occupancy_every=37)
            rot = [np.hstack([np.eye(3), np.zeros((3, 1))]),
                   np.array([[-1.0, 0, 0, 0.5 * header.xlength], [0, -1.0, 0, 0], [0, 0, 1.0, 0.5 * header.zlength]])]
            pdb = structure.PDBEntry(structure.PDBHeader(pdbid="synth%d" % self.seed, resolution=2.0, spaceGroup="P_1", rotationMats=rot))
            SyntheticEntryFiles._structures[key] = (st, pdb)
        return SyntheticEntryFiles._structures[key]

    def __call__(self):
        if self.as_paths:
            st, pdb = self.structure()
            st.__dict__.pop("_pdbeda_columns", None)
            return self.density_path, self.diff_path, st, pdb
        with open(self.density_path, "rb") as fh:
            dens = fh.read()
        with open(self.diff_path, "rb") as fh:
            diff = fh.read()
        st, pdb = self.structure()
        st.__dict__.pop("_pdbeda_columns", None)        # (a real entry arrives with a fresh structure: no snapshot carried over)
        return dens, diff, st, pdb


def time_single_loads(task):
    """bench.py's ``load_single``: ``task`` = (paths, repetitions) -> seconds of every ``DeviceMap.from_file`` of ONE map at a time on a
    context of its own (run in a pool worker: a fresh process with nothing else on the card -- the bench's own process holds a dozen
    streams by then, which slows every copy stream of the process)."""
    import time
    from . import _native, ccp4, multipleStructures
    paths, repetitions = task
    ctx = _native.Context(multipleStructures._worker_state.get("device", 0))
    try:
        head = ccp4.read(paths[0], "lone", ctx=ctx, lazy=True)
        geom, off = head.header.geometry(), 1024 + head.header.symmetryBytes
        times = []
        for k in range(repetitions):
            t1 = time.perf_counter()
            one_map = _native.DeviceMap.from_file(ctx, paths[k % len(paths)], off, False, geom)
            times.append(time.perf_counter() - t1)
            one_map.free()
        return times
    finally:
        ctx.close()

