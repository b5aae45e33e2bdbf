"""The minimal PDB reader (ATOM / HETATM / REMARK 290 for the fields the path consumes) and the file-based
drop-in entry points ``fromFile`` / ``fromPDBid`` (ref densityAnalysis.py:88-229)."""
import gzip
import io
import os

import numpy as np
import pytest

PDB_TEXT = """HEADER    TEST PROTEIN                            01-JAN-00   1ABC
REMARK   2 RESOLUTION.    1.80 ANGSTROMS.
REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP: P 21 21 21
REMARK 290   SMTRY1   1  1.000000  0.000000  0.000000        0.00000
REMARK 290   SMTRY2   1  0.000000  1.000000  0.000000        0.00000
REMARK 290   SMTRY3   1  0.000000  0.000000  1.000000        0.00000
REMARK 290   SMTRY1   2 -1.000000  0.000000  0.000000       10.50000
REMARK 290   SMTRY2   2  0.000000 -1.000000  0.000000        0.00000
REMARK 290   SMTRY3   2  0.000000  0.000000  1.000000       12.25000
MODEL        1
ATOM      1  N   ALA A   1      11.104   6.134  -6.504  1.00 10.00           N
ATOM      2  CA AALA A   1      11.639   6.071  -5.147  0.40 11.00           C
ATOM      3  CA BALA A   1      11.700   6.100  -5.100  0.60 12.00           C
ATOM      4  C   ALA A   1      13.100   6.500  -5.100  1.00 13.00           C
ATOM      5  O   ALA A   1      13.500   7.600  -5.500  0.00 14.00           O
ATOM      6  N   GLY B   2       1.000   2.000   3.000  1.00 15.00           N
HETATM    7  O   HOH A 101       5.000   5.000   5.000  1.00 30.00           O
HETATM    8 ZN    ZN A 201       7.000   7.000   7.000  1.00 20.00          ZN
ENDMDL
MODEL        2
ATOM      9  N   ALA A   1      99.000  99.000  99.000  1.00 10.00           N
ENDMDL
END
"""


def test_read_pdb_fields_order_and_selection():
    from pdb_eda_amd import structure
    st, pdb = structure.read_pdb(io.StringIO(PDB_TEXT), "1abc")
    atoms = list(st.get_atoms())
    assert [a.name for a in atoms] == ["N", "CA", "C", "O", "O", "ZN", "N"]            # first model only; chain -> residue -> atom (Bio.PDB order)
    ca = atoms[1]
    assert ca.get_occupancy() == pytest.approx(0.6) and ca.get_bfactor() == 12.0        # the higher-occupancy conformer
    assert ca.coord.dtype == np.float32 and np.allclose(ca.coord, [11.7, 6.1, -5.1])
    assert atoms[3].get_occupancy() == 0.0
    residues = list(st.get_residues())
    assert [(r.parent.id, r.id[0], r.id[1]) for r in residues] == [("A", " ", 1), ("A", "W", 101), ("A", "H_ZN", 201), ("B", " ", 2)]
    assert residues[0].resname == "ALA" and residues[3].resname == "GLY"
    assert atoms[0].parent.parent.parent.id == 0                                          # model id
    assert atoms[5].element == "ZN"
    assert st.header["resolution"] == pytest.approx(1.8)
    h = pdb.header
    assert h.pdbid == "1ABC" and h.spaceGroup == "P_21_21_21" and len(h.rotationMats) == 2
    assert np.allclose(h.rotationMats[1], [[-1, 0, 0, 10.5], [0, -1, 0, 0], [0, 0, 1, 12.25]])


def test_read_pdb_gz_path(tmp_path):
    from pdb_eda_amd import structure
    path = tmp_path / "pdb1abc.ent.gz"
    with gzip.open(path, "wt") as fh:
        fh.write(PDB_TEXT)
    st, pdb = structure.read_pdb(str(path), "1abc")
    assert len(list(st.get_atoms())) == 7


@pytest.mark.gpu
def test_from_file_and_from_pdbid(tmp_path, monkeypatch, gpu_ctx):
    """The reference's file-based constructors on files on disk: CCP4 maps written by the synthetic generator and a PDB file
    written from the golden structure give the golden density-electron ratio; a missing file gives 0 (Q7-style contract)."""
    from conftest import load_analysis_case
    from pdb_eda_amd import synthetic, densityAnalysis
    z, spec, st, pdb, params = load_analysis_case("orth")
    densityAnalysis.setGlobals(params)
    (tmp_path / "ccp4_data").mkdir()
    (tmp_path / "pdb_data").mkdir()
    with open(tmp_path / "ccp4_data" / "9xyz.ccp4", "wb") as fh:
        fh.write(synthetic.ccp4_bytes(spec, z["dens"]))
    with open(tmp_path / "ccp4_data" / "9xyz_diff.ccp4", "wb") as fh:
        fh.write(synthetic.ccp4_bytes(spec, z["diff"]))
    lines = ["HEADER    SYNTHETIC                               01-JAN-00   9XYZ", "REMARK   2 RESOLUTION.    2.00 ANGSTROMS."]
    for k, m in enumerate(pdb.header.rotationMats):
        for row in range(3):
            lines.append("REMARK 290   SMTRY%d %3d%10.6f%10.6f%10.6f%15.5f" % (row + 1, k + 1, m[row][0], m[row][1], m[row][2], m[row][3]))
    serial = 0
    for a in st.get_atoms():
        serial += 1
        res = a.parent
        tag = "ATOM  " if res.id[0] == " " else "HETATM"
        name = a.name if len(a.name) == 4 else " " + a.name.ljust(3)
        lines.append("%s%5d %s %3s %s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s" % (tag, serial, name, res.resname, res.parent.id, res.id[1],
                                                                                    a.coord[0], a.coord[1], a.coord[2], a.get_occupancy(), a.get_bfactor(), a.element.rjust(2)))
    lines.append("END")
    with gzip.open(tmp_path / "pdb_data" / "pdb9xyz.ent.gz", "wt") as fh:
        fh.write("\n".join(lines) + "\n")
    monkeypatch.chdir(tmp_path)
    an = densityAnalysis.fromPDBid("9XYZ")
    assert an != 0 and an.pdbid == "9xyz"
    # coordinates went through %8.3f: the ratio is close to, not bit-equal with, the golden one
    assert an.densityElectronRatio == pytest.approx(float(z["ratio"]), rel=2e-2)
    an2 = densityAnalysis.fromFile(str(tmp_path / "pdb_data" / "pdb9xyz.ent.gz"), str(tmp_path / "ccp4_data" / "9xyz.ccp4"), str(tmp_path / "ccp4_data" / "9xyz_diff.ccp4"))
    assert an2 != 0 and an2.densityElectronRatio == pytest.approx(an.densityElectronRatio, rel=1e-12)
    assert len(an2.greenBlobList) == len(an.greenBlobList)
    assert densityAnalysis.fromPDBid("0000") == 0
    assert densityAnalysis.fromFile(str(tmp_path / "nope.pdb")) == 0


def test_read_pdb_header_equals_reference_pdbparser():
    """SURVEY 8f.2, header half: every field of the reference's pdbParser.parse (lite mode) on five synthetic header texts
    (tests/golden/make_golden_pdbheader.py ran the reference).  The ATOM / HETATM half stays unpinned (Bio.PDB is not installed)."""
    import json
    from pdb_eda_amd import structure
    with open(os.path.join(os.path.dirname(__file__), "golden", "pdbheader.json")) as fh:
        cases = json.load(fh)
    assert len(cases) >= 5
    for name, case in cases.items():
        _, pdb = structure.read_pdb(io.StringIO(case["text"]), name)
        h, want = pdb.header, case["header"]
        for field in ("pdbid", "date", "method", "resolution", "rValue", "rFree", "program", "spaceGroup"):
            assert getattr(h, field) == want[field], (name, field, getattr(h, field), want[field])
            assert type(getattr(h, field)) is type(want[field]), (name, field)       # 0 stays the integer 0, text stays text
        assert len(h.rotationMats) == len(want["rotationMats"]), name
        for got, ref in zip(h.rotationMats, want["rotationMats"]):
            assert np.array_equal(np.asarray(got), np.asarray(ref)), name


def test_bench_gpus_flag_refuses_to_measure_fewer_devices():
    """`python bench.py --gpus N` starts N ranks itself (torch.distributed.run as a child) -- and with fewer than N devices visible
    it exits non-zero instead of measuring one GPU."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "device(s) are visible" in p.stderr, (p.returncode, p.stderr[-500:])
    env["WORLD_SIZE"] = "3"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "does not match WORLD_SIZE" in p.stderr, (p.returncode, p.stderr[-500:])


def test_type_regressions_equal_scipy_linregress():
    """The per-atom-type b-factor regression of the statistics tail (ref densityAnalysis.py:752-757: one scipy.stats.linregress
    per type) is computed for all types at once; slopes, p-values and the decision to fit must be scipy's."""
    import warnings
    import numpy as np
    from scipy import stats
    from pdb_eda_amd import densityAnalysis as da
    rng = np.random.default_rng(5)
    for trial in range(40):
        n_types = int(rng.integers(1, 7))
        n = int(rng.integers(n_types, 400))
        group = np.concatenate([np.arange(n_types), rng.integers(0, n_types, n - n_types)])
        rng.shuffle(group)
        b = np.round(rng.uniform(5.0, 60.0, n), 2)
        if trial % 4 == 0:
            b[group == 0] = 20.0                      # a type whose b-factors are all equal: no fit
        if trial % 5 == 0 and n_types > 1:
            keep = np.ones(n, dtype=bool)
            keep[np.nonzero(group == 1)[0][2:]] = False   # a type with two rows: no fit
            group, b = group[keep], b[keep]
        if trial % 7 == 0 and n_types > 2:
            b[group == 2] = np.nan                    # a type with no positive B: its normalised b-factors are all NaN -- ONE value for np.unique: no fit
        if trial % 9 == 0 and n_types > 3 and (group == 3).sum() > 3:
            b[np.nonzero(group == 3)[0][0]] = np.nan  # NaN beside numbers: two values or more, fitted (as the reference does)
        with np.errstate(invalid="ignore"):
            x = np.log(b)
        y = 0.02 * np.nan_to_num(x) * rng.uniform(-1, 1) + rng.normal(0, 0.05, len(x))
        slope, p, fitted = da._typeRegressions(x, y, b, group, n_types)
        for k in range(n_types):
            sel = group == k
            want_fit = sel.sum() > 2 and len(np.unique(b[sel])) != 1
            assert bool(fitted[k]) == bool(want_fit)
            if want_fit and not np.isnan(x[sel]).any():
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    fit = stats.linregress(x[sel], y[sel])
                assert slope[k] == pytest.approx(fit.slope, rel=1e-10, abs=1e-14)
                assert p[k] == pytest.approx(fit.pvalue, rel=1e-8, abs=1e-14)


def test_columns_c_walk_equals_python_walk():
    """structure.Columns reads the object tree once per entry: the C walk (pdb_eda_amd/_hostwalk.so) and the Python loops it
    replaces must give the same snapshot -- on our own tree, with float64 coordinates and a missing occupancy (Bio.PDB has
    both), and an object the C walk does not read falls back to the loops."""
    import numpy as np
    import __graft_entry__
    __graft_entry__.build()
    from pdb_eda_amd import structure, synthetic
    lo, hi = np.zeros(3), np.full(3, 40.0)
    st = synthetic.chain_structure(60, 3, lo, hi, hetero_every=7, zero_occupancy_every=11)
    atoms = list(st.get_atoms())
    atoms[5].coord = atoms[5].coord.astype(np.float64)
    atoms[9].occupancy = None
    atoms[12].coord = np.asarray([1.0, 2.0, 3.0, 4.0], dtype=np.float32)[::1][:3][::-1][::-1]      # a view with the same layout
    fast, slow = structure.Columns(st, native=True), structure.Columns(st, native=False)
    for field in ("res_model", "res_chain", "res_number", "res_name", "name", "atom_names", "occupancy_raw", "pair_names"):
        assert getattr(fast, field) == getattr(slow, field), field
    assert all(a is b for a, b in zip(fast.atoms, slow.atoms)) and all(a is b for a, b in zip(fast.residues, slow.residues))
    for field in ("res_het", "res_start", "res_of_atom", "name_of_atom", "occupancy", "bfactor", "coord32", "coord", "pair_of_atom"):
        a, b = getattr(fast, field), getattr(slow, field)
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b, equal_nan=True), field
    assert np.isnan(fast.occupancy[9])
    atoms[3].coord = [1.0, 2.0, 3.0]                     # a plain list: not a buffer
    with pytest.raises(TypeError):
        structure.Columns(st, native=True)
    assert np.array_equal(structure.Columns(st).coord32, structure.Columns(st, native=False).coord32)


def test_last_atom_with_the_same_coordinate():
    """The alias rule's index column (the last atom of a float32 coordinate triple, ref densityAnalysis.py:604-605) against a dict
    keyed the way the reference keys allAtomClouds, duplicates, -0.0 and all."""
    import numpy as np
    from pdb_eda_amd import densityAnalysis as da
    rng = np.random.default_rng(11)
    for n in (0, 1, 7, 500):
        xyz = rng.normal(0, 10, (n, 3)).astype(np.float32)
        if n >= 7:
            xyz[3] = xyz[1]; xyz[6] = xyz[1]; xyz[5] = xyz[0]
            xyz[2] = [0.0, 1.0, 2.0]; xyz[4] = [-0.0, 1.0, 2.0]
        last = {}
        for i in range(n):
            last[tuple(xyz[i])] = i
        want = np.array([last[tuple(xyz[i])] for i in range(n)], dtype=np.int64)
        assert np.array_equal(da._lastWithSameCoord(xyz), want)


def test_device_blobs_sequence_semantics_without_a_device():
    """ccp4.DeviceBlobs over a stand-in for a device list: objects on first access only, `+` of two is one (sharing the
    objects), list semantics for ==, + with a plain list, indexing and slices; columns() until an object exists."""
    import numpy as np
    from pdb_eda_amd import ccp4

    class FakeList(object):
        def __init__(self, n, base):
            self.calls = 0
            self._st = {"centroid": np.arange(3 * n, dtype=np.float64).reshape(n, 3) + base, "coordCenter": np.zeros((n, 3)),
                        "totalDensity": np.linspace(1.0, 2.0, n) + base, "volume": np.full(n, 0.125), "n": np.arange(1, n + 1), "firstKey": np.arange(n) * 7}

        def stats(self):
            self.calls += 1
            return self._st
    a_list, b_list = FakeList(4, 0.0), FakeList(3, 100.0)
    a, b = ccp4.DensityBlob.listFromDevice(a_list, None), ccp4.DensityBlob.listFromDevice(b_list, None)
    assert isinstance(a, ccp4.DeviceBlobs) and len(a) == 4 and len(b) == 3 and a_list.calls == 1
    both = a + b
    cols = both.columns()
    assert cols is not None and cols["n"].tolist() == [1, 2, 3, 4, 1, 2, 3] and cols["centroid"].shape == (7, 3)
    assert a.columns() is not None                      # nothing materialised yet
    first = both[0]
    assert isinstance(first, ccp4.DensityBlob) and first.numVoxels == 1 and first is a[0]
    assert both.columns() is None and a.columns() is None      # (objects exist: they are the truth from now on)
    assert [blob.numVoxels for blob in both] == [1, 2, 3, 4, 1, 2, 3] and both[-1] is b[2]
    assert both[1:3] == [a[1], a[2]] and a == list(a) and list(b) == b and a != b
    assert (a + [1])[-1] == 1 and ([0] + b)[0] == 0
    assert ccp4.DeviceBlobs([]) == [] and not ccp4.DeviceBlobs([]) and ccp4.DeviceBlobs([]).columns()["n"].size == 0


def test_cloud_statistics_in_c_equal_the_numpy_form():
    """The statistics tail of aggregateCloud (per-type medians, b-factor regressions, corrected fractions) runs in C
    (_hostwalk.cloud_stats, round 5); the numpy form it replaces stays as the fallback and is the check here: random tables with
    types of one row, b-factors that are all zero in a type (no fit, NaN medians), a single distinct b-factor, missing b-factors
    among valid ones, NaN distances.  Medians are the same order statistics: equal; the regressions sum in the same order: 1e-12."""
    import types
    import numpy as np
    import __graft_entry__
    __graft_entry__.build()
    from pdb_eda_amd import densityAnalysis as da, synthetic
    params = synthetic.synthetic_params()
    da.setGlobals(params)
    type_names = sorted(params["radii"])
    rng = np.random.default_rng(12)
    for trial in range(30):
        n = int(rng.integers(1, 600))
        n_pairs = int(rng.integers(1, 40))
        pair_type_id = rng.integers(0, len(type_names), n_pairs)
        pair = rng.integers(0, n_pairs, n)
        b = np.round(rng.uniform(5.0, 60.0, n), 2)
        tid = pair_type_id[pair]
        if trial % 3 == 0:
            b[tid == tid[0]] = 0.0                      # a type without a positive b-factor
        if trial % 4 == 0:
            b[tid == tid[-1]] = 17.5                    # one distinct value: no fit
        if trial % 5 == 0:
            b[rng.integers(0, n, max(1, n // 10))] = 0.0   # missing b-factors among valid ones: filled with the type's median
        dist = rng.uniform(0.0, 0.6, n)
        if trial % 7 == 0:
            dist[rng.integers(0, n, 1)] = np.nan
        cols = types.SimpleNamespace(res_of_atom=np.zeros(n, dtype=np.int64), bfactor=b, res_chain=["A"], res_number=[1], res_name=["ALA"],
                                     atom_names=["CA"], name_of_atom=np.zeros(n, dtype=np.int64))
        inp = {"cols": cols, "rows": np.arange(n), "pair_type_id": pair_type_id, "pair": pair, "type_names": type_names,
               "electrons": rng.integers(1, 9, n).astype(np.float64), "occupancy": rng.choice([1.0, 0.5], n)}
        res = {"atom": np.arange(n), "atom_distance": dist, "atom_total": rng.uniform(0.5, 20.0, n), "atom_n": rng.integers(1, 60, n),
               "atom_centroid": rng.uniform(0, 30, (n, 3))}
        with np.errstate(all="ignore"):
            make_c, n_c, med_c = da.DensityAnalysis._cloudStatistics(inp, res, 0.67, 0.125, None, native=True)
            make_p, n_p, med_p = da.DensityAnalysis._cloudStatistics(inp, res, 0.67, 0.125, None, native=False)
            tab_c, tab_p = make_c(), make_p()
        assert n_c == n_p and list(med_c) == list(med_p)
        for field in med_p:
            assert list(med_c[field]) == list(med_p[field])
            for t in med_p[field]:
                a, w = float(med_c[field][t]), float(med_p[field][t])
                assert (np.isnan(a) and np.isnan(w)) or a == pytest.approx(w, rel=1e-12, abs=1e-15), (trial, field, t, a, w)
        for field in tab_p.dtype.names:
            if tab_p.dtype[field].kind == "f":
                assert np.allclose(tab_c[field], tab_p[field], rtol=1e-12, atol=1e-15, equal_nan=True), (trial, field)
            else:
                assert np.array_equal(tab_c[field], tab_p[field]), (trial, field)


def test_table_rows_in_c_equal_the_zip_form():
    """densityAnalysis._rows: the result tables' rows made in one C pass (_hostwalk.table_rows) from whole columns -- lists, numpy arrays,
    (list, index) picks -- are what list(map(list, zip(...))) makes of the columns' .tolist(): the same values AND the same Python types."""
    from pdb_eda_amd import densityAnalysis, structure
    walk = structure._hostwalk()
    assert walk is not None and hasattr(walk, "table_rows")
    rng = np.random.default_rng(5)
    n = 257
    f = rng.standard_normal(n)
    i64 = rng.integers(-5, 5000, n)
    i32 = i64.astype(np.int32)
    flag = rng.random(n) < 0.3
    xyz = rng.standard_normal((n, 3))
    names = ["A%d" % k for k in range(40)]
    pick = rng.integers(0, 40, n).astype(np.int64)
    tuples = [(k, k + 1) for k in range(n)]
    scalars = list(np.float64(f))                   # (numpy scalars in a list stay what they are)
    ops = rng.integers(-1, 3, (n, 4))
    got = densityAnalysis.DensityAnalysis._rows(f, i64, i32, flag, xyz, (names, pick), tuples, scalars, xyz[:, ::2], ops)     # (xyz[:, ::2]: not contiguous)
    want = list(map(list, zip(f.tolist(), i64.tolist(), i32.tolist(), flag.tolist(), xyz.tolist(), [names[r] for r in pick.tolist()], tuples, scalars, xyz[:, ::2].tolist(),
                              [tuple(t) for t in ops.tolist()])))
    assert got == want
    for a, b in zip(got[3], want[3]):
        assert type(a) is type(b)
    assert all(type(row[0]) is float and type(row[1]) is int and type(row[3]) is bool and type(row[4]) is list and type(row[9]) is tuple for row in got)
    assert densityAnalysis.DensityAnalysis._rows(np.zeros(0), []) == []
    with pytest.raises(ValueError):
        walk.table_rows([f, i64[:-1]])
    with pytest.raises(IndexError):
        walk.table_rows([(names, np.array([40], dtype=np.int64))])
    with pytest.raises(TypeError):
        walk.table_rows([f.astype(np.float32)])
    # columns the helper does not take (float32) go the plain way inside _rows
    assert densityAnalysis.DensityAnalysis._rows(f.astype(np.float32), i64, ops) == list(map(list, zip(f.astype(np.float32).tolist(), i64.tolist(), [tuple(t) for t in ops.tolist()])))


def test_columns_keep_the_atoms_own_coordinate_objects():
    """structure.Columns.atom_lists('coord'): the very objects atom.coord returns (the symmetry-atom tables list them, as the reference's SymAtom
    holds them), from the C walk and from the plain one."""
    from pdb_eda_amd import structure, synthetic
    spec, header, st, params, dens, diff, rot = synthetic.cube_entry((40, 40, 40), 12, 3, 0.6)
    for native in (None, False):
        cols = structure.Columns(st, native=native) if "native" in structure.Columns.__init__.__code__.co_varnames else structure.Columns(st)
        own = cols.atom_lists("coord")
        assert len(own) == len(cols.atoms) > 20
        assert all(a is atom.coord for a, atom in zip(own, cols.atoms))


def test_nan_cutoff_is_numpys_to_the_bit():
    """_hostwalk.nan_cutoff(values, k) == np.nanmedian(values) + np.nanstd(values) * k exactly (the row filter of aggregateCloud's atom table,
    densityAnalysis.py:731): lengths around numpy's pairwise blocks (8, 128, 8192), with and without NaNs, all NaN."""
    import warnings
    from pdb_eda_amd import structure
    walk = structure._hostwalk()
    assert walk is not None and hasattr(walk, "nan_cutoff")
    rng = np.random.default_rng(7)
    for n in (1, 2, 3, 7, 8, 9, 16, 17, 127, 128, 129, 136, 255, 256, 257, 1000, 2001, 8191, 8192, 8193, 9001, 20000):
        for nan_share in (0.0, 0.05, 0.6):
            for scale, shift in ((1.0, 0.0), (1e-3, 5.0)):
                x = rng.random(n) * scale + shift
                x[rng.random(n) < nan_share] = np.nan
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    want = np.nanmedian(x) + np.nanstd(x) * 2.5
                got = walk.nan_cutoff(x, 2.5)
                assert got == want or (np.isnan(got) and np.isnan(want)), (n, nan_share, scale)
    assert np.isnan(walk.nan_cutoff(np.full(5, np.nan), 2.0)) and np.isnan(walk.nan_cutoff(np.zeros(0), 2.0))

