"""Host logic: the columnar flattening of a structure for ``pdbeda_aggregate_cloud`` against a per-atom walk that follows
the reference's loops (densityAnalysis.py:596-604, 617-621, 653-656) statement by statement."""
import numpy as np
import pytest

from pdb_eda_amd import densityAnalysis as da, structure, synthetic


def walk(biopdbObj):
    """The plain loops: eligible atoms in iteration order, key per (residue, name) by first appearance, last atom per
    coordinate, bonded keys within the residue, owners = every child atom whose (residue, name) has a key."""
    typeMap, electronsMap, radii, bondedMap = da.fullAtomNameMapAtomTypeGlobal, da.fullAtomNameMapElectronsGlobal, da.radiiGlobal, da.bondedAtomsGlobal
    residues = [res for res in biopdbObj.get_residues() if res.id[0] == ' ']
    atoms, names, residue_of, key_of, keys, children = [], [], [], [], {}, []
    for ri, residue in enumerate(residues):
        for atom in residue.child_list:
            name = da.residueAtomName(atom)
            children.append((ri, name))
            if name not in typeMap or atom.get_occupancy() == 0:
                continue
            atoms.append(atom)
            names.append(name)
            residue_of.append(ri)
            key_of.append(keys.setdefault((ri, name), len(keys)))
    last = {}
    for i, a in enumerate(atoms):
        last[a.coord.tobytes()] = i
    bonded_off, bonded = [0], []
    for (ri, name) in keys:
        bonded.extend(keys[(ri, other)] for other in bondedMap[name] if (ri, other) in keys)
        bonded_off.append(len(bonded))
    return {"atoms": atoms, "xyz": np.array([a.coord for a in atoms], dtype=np.float64).reshape(len(atoms), 3),
            "occupancy": np.array([a.get_occupancy() for a in atoms], dtype=np.float64),
            "electrons": np.array([electronsMap[n] for n in names], dtype=np.float64),
            "radius": np.array([radii[typeMap[n]] for n in names], dtype=np.float32),
            "residue": np.array(residue_of, dtype=np.int32), "key": np.array(key_of, dtype=np.int32),
            "alias": np.array([last[a.coord.tobytes()] for a in atoms], dtype=np.int32),
            "bonded_off": np.array(bonded_off, dtype=np.int64), "bonded": np.array(bonded, dtype=np.int32),
            "owner_key": np.array([keys[c] for c in children if c in keys], dtype=np.int32),
            "owner_type": [typeMap[c[1]] for c in children if c in keys], "residues": residues}


def make_structure(seed, n_res):
    rng = np.random.default_rng(seed)
    st = synthetic.chain_structure(n_res, 3, np.array([2.0, 2.0, 2.0]), np.array([40.0, 40.0, 40.0]), hetero_every=5, zero_occupancy_every=7)
    atoms = list(st.get_atoms())
    for k in rng.choice(len(atoms), size=max(2, len(atoms) // 9), replace=False):          # shared coordinates (alternate models of a site)
        atoms[int(k)].coord = atoms[int(rng.integers(len(atoms)))].coord.copy()
    res = list(st.get_residues())
    extra = res[min(1, len(res) - 1)]
    structure.Atom("XX9", np.array([5.0, 5.0, 5.0]), 1.0, 10.0, "X", extra)                  # a name the tables do not know
    structure.Atom(extra.child_list[0].name, np.array([6.0, 5.0, 5.0]), 0.5, 10.0, "C", extra)   # a repeated (residue, name)
    return st


@pytest.mark.parametrize("native", [True, False])       # the one pass in C (pdb_eda_amd/_hostwalk.so) and the numpy form it replaces
@pytest.mark.parametrize("seed,n_res", [(1, 1), (2, 7), (3, 60), (4, 211)])
def test_cloud_inputs_match_the_plain_walk(seed, n_res, native):
    import __graft_entry__
    __graft_entry__.build()
    da.setGlobals(synthetic.synthetic_params())
    st = make_structure(seed, n_res)
    fixed = da.DensityAnalysis._cloudInputsFixed(structure.columns(st), native=native)
    if native:          # what the analysis itself takes is the C pass when it is built
        structure.columns(st).__dict__.pop("_cloud_inputs", None)
    inp = da.DensityAnalysis("t", None, None, st, None)._cloudInputs()
    for name in ("rows", "residue", "alias", "key", "bonded_off", "bonded", "owner_key", "owner_type_id", "pair", "electrons", "plain_residues"):
        a, b = np.asarray(fixed[name]), np.asarray(inp[name])
        assert a.shape == b.shape and (a == b).all(), name
    inp = dict(inp)
    inp.update({k: fixed[k] for k in ("rows", "residue", "alias", "key", "bonded_off", "bonded", "owner_key", "owner_type_id", "plain_residues")})
    ref = walk(st)
    cols = inp["cols"]
    assert [cols.atoms[i] for i in inp["rows"].tolist()] == ref["atoms"]
    for name in ("xyz", "occupancy", "electrons", "radius", "residue", "key", "alias", "bonded_off", "bonded", "owner_key"):
        got, want = np.asarray(inp[name]), ref[name]
        assert got.dtype == want.dtype and got.shape == want.shape and (got == want).all(), name
    assert [inp["type_names"][k] for k in inp["owner_type_id"].tolist()] == ref["owner_type"]
    assert [cols.residues[k] for k in inp["plain_residues"].tolist()] == ref["residues"]


def test_cloud_inputs_empty_structure():
    da.setGlobals(synthetic.synthetic_params())
    st = structure.Structure("e")
    structure.Chain("A", structure.Model(0, st))
    inp = da.DensityAnalysis("t", None, None, st, None)._cloudInputs()
    assert len(inp["rows"]) == 0


def test_columns_snapshot():
    st = make_structure(9, 12)
    cols = structure.columns(st)
    assert structure.columns(st) is cols and structure.columns(st, refresh=True) is not cols
    atoms = list(st.get_atoms())
    assert cols.atoms == atoms and cols.residues == list(st.get_residues())
    assert (cols.coord == np.array([a.coord for a in atoms], dtype=np.float64)).all()
    assert [cols.pair_names[k] for k in cols.pair_of_atom.tolist()] == [da.residueAtomName(a) for a in atoms]
    assert cols.atom_lists("chain") == [a.parent.parent.id for a in atoms]
    assert cols.atom_lists("number") == [a.parent.id[1] for a in atoms]
    assert [cols.residues[k] for k in cols.res_of_atom.tolist()] == [a.parent for a in atoms]


def test_cloud_inputs_follow_the_radius_table():
    """Only the radius column changes between the iterations of optimise mode: the rest is kept on the structure's snapshot
    while the name tables are the same objects, and rebuilt when they are not."""
    base = synthetic.synthetic_params()
    da.setGlobals(base)
    st = make_structure(5, 40)
    an = da.DensityAnalysis("t", None, None, st, None)
    first = an._cloudInputs()
    radii = dict(base["radii"])
    some_type = first["pair_type"][int(first["pair"][0])]
    radii[some_type] = radii[some_type] + 0.25
    da.setGlobals({**base, "radii": radii})
    second = an._cloudInputs()
    assert second["key"] is first["key"] and second["xyz"] is first["xyz"]                  # the kept part
    changed = np.asarray(second["radius"]) != np.asarray(first["radius"])
    assert changed.any() and all(first["pair_type"][k] == some_type for k in np.asarray(first["pair"])[changed].tolist())
    ref = walk(st)
    assert (np.asarray(second["radius"]) == ref["radius"]).all()
    da.setGlobals(synthetic.synthetic_params())                                             # new table objects: everything is rebuilt
    third = an._cloudInputs()
    assert third["key"] is not first["key"] and (np.asarray(third["key"]) == np.asarray(first["key"])).all()


def test_segment_means_are_numpy_means():
    """``_segmentMeans`` (rows of equal length reduced along their contiguous axis) is bit-equal to ``np.mean`` per segment:
    the per-residue mean occupancy of the region tables (densityAnalysis.py:1033, 1156), also for long residues where numpy
    switches to its unrolled pairwise summation."""
    rng = np.random.default_rng(3)
    for trial in range(50):
        counts = rng.integers(0, 45, size=60)
        values = rng.random(int(counts.sum())) * rng.choice([1.0, 0.5, 0.37, 0.01], size=int(counts.sum()))
        got = da._segmentMeans(values, counts)
        start = np.concatenate([[0], np.cumsum(counts)])
        for k in range(len(counts)):
            seg = values[start[k]:start[k + 1]]
            if len(seg) == 0:
                assert np.isnan(got[k])
            else:
                assert got[k] == np.mean(list(seg))


def test_sweep_worker_keeps_the_flattened_structure_between_iterations():
    """ProcessSweep sends every iteration's parameter table through a pipe: equal tables, NEW objects.  The worker splices the
    first objects back in, so the per-structure cache of densityAnalysis._cloudInputs (keyed on the identity of the name tables)
    hits from the second iteration on: only the radius column is rebuilt."""
    import pickle
    from pdb_eda_amd import densityAnalysis, optimizeSweep, structure, synthetic
    sets = synthetic.sweep_param_sets()
    spec, header, st, params, dens, diff, rot = synthetic.cube_entry((40, 40, 40), 12, 3, 0.5)
    kept = {}
    flattened = []
    for p in sets[:3]:
        p = optimizeSweep._splice_fixed_tables(pickle.loads(pickle.dumps(p)), kept)     # what a worker receives, then does
        densityAnalysis.setGlobals(p)
        an = densityAnalysis.DensityAnalysis.__new__(densityAnalysis.DensityAnalysis)
        an.biopdbObj = st
        inp = an._cloudInputs()
        flattened.append(structure.columns(st)._cloud_inputs[1])
        assert np.array_equal(inp["radius"], np.array([p["radii"][t] for t in np.asarray(inp["pair_type"], dtype=object)[inp["pair"]]], dtype=np.float32))
    assert flattened[0] is flattened[1] is flattened[2]
    # without the splice the cache would miss: the unpickled tables are equal but not identical
    densityAnalysis.setGlobals(pickle.loads(pickle.dumps(sets[0])))
    an._cloudInputs()
    assert structure.columns(st)._cloud_inputs[1] is not flattened[0]
