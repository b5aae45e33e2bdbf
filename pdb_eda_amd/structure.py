"""Minimal structure object model with the Bio.PDB attribute surface the hot path reads.

The reference walks a ``Bio.PDB`` structure (``densityAnalysis.py:596-643, 962-971,
1019-1033``) using only: ``structure.get_residues()/get_atoms()``, ``residue.id``
(hetero flag, number, icode), ``residue.resname``, ``residue.child_list``,
``residue.parent`` (chain, ``.id``), ``atom.name``, ``atom.coord`` (float32[3]),
``atom.parent``, ``atom.get_occupancy()``, ``atom.get_bfactor()``, ``atom.element``.
Biopython is not installed on the target image, so this module provides those names and
nothing else; a real ``Bio.PDB`` structure can be passed to ``DensityAnalysis`` unchanged.

:func:`read_pdb` is the SURVEY.md 8f-2 "minimal ATOM/HETATM + REMARK 290 reader" (the
fields ``pdbParser.py:67-95`` and Bio.PDB feed into the path); it is host-side input
preparation, not part of the accelerated path.
"""
import re

import itertools
import os
import sys
import numpy as np

__all__ = ["Structure", "Model", "Chain", "Residue", "Atom", "PDBHeader", "PDBEntry", "read_pdb", "columns", "Columns"]


class _Entity(object):
    def __init__(self, id_, parent=None):
        self.id = id_
        self.parent = parent
        self.child_list = []
        if parent is not None:
            parent.child_list.append(self)

    def get_parent(self):
        return self.parent

    def get_id(self):
        return self.id

    def __iter__(self):
        return iter(self.child_list)

    def __len__(self):
        return len(self.child_list)


class Structure(_Entity):
    def __init__(self, id_):
        super().__init__(id_, None)
        self.header = {}

    def get_models(self):
        return iter(self.child_list)

    def get_chains(self):
        for m in self.child_list:
            for c in m.child_list:
                yield c

    def get_residues(self):
        for c in self.get_chains():
            for r in c.child_list:
                yield r

    def get_atoms(self):
        for r in self.get_residues():
            for a in r.child_list:
                yield a


class Model(_Entity):
    pass


class Chain(_Entity):
    pass


class Residue(_Entity):
    def __init__(self, id_, resname, parent=None):
        super().__init__(id_, parent)
        self.resname = resname

    def get_resname(self):
        return self.resname

    def get_atoms(self):
        return iter(self.child_list)


class Atom(object):
    def __init__(self, name, coord, occupancy, bfactor, element, parent=None, serial_number=0, altloc=" "):
        self.name = name
        self.coord = np.asarray(coord, dtype=np.float32)
        self.occupancy = occupancy
        self.bfactor = bfactor
        self.element = element
        self.parent = parent
        self.serial_number = serial_number
        self.altloc = altloc
        if parent is not None:
            parent.child_list.append(self)

    def get_occupancy(self):
        return self.occupancy

    def get_bfactor(self):
        return self.bfactor

    def get_coord(self):
        return self.coord

    def get_name(self):
        return self.name

    def get_parent(self):
        return self.parent


class PDBHeader(object):
    """The fields of ``pdbParser.PDBheader`` the analysis reads (pdbParser.py:97-98)."""

    def __init__(self, pdbid="", resolution=0, spaceGroup=0, rotationMats=None, **extra):
        self.pdbid = pdbid
        self.resolution = resolution
        self.spaceGroup = spaceGroup
        self.rotationMats = rotationMats if rotationMats is not None else [np.hstack([np.eye(3), np.zeros((3, 1))])]
        for k, v in extra.items():
            setattr(self, k, v)


class PDBEntry(object):
    def __init__(self, header, atoms=None):
        self.header = header
        self.atoms = atoms or []


def _first_appearance_ids(values):
    """(distinct values in order of first appearance, int64 id of every value)."""
    ids = dict.fromkeys(values)
    for k, value in enumerate(ids):
        ids[value] = k
    return list(ids), np.fromiter(map(ids.__getitem__, values), dtype=np.int64, count=len(values))


_hostwalk_module = [False]


def _hostwalk():
    """The C walk of the object tree (pdb_eda_amd/_hostwalk.so, csrc/hostwalk.c), or None when it is not built."""
    if _hostwalk_module[0] is False:
        try:
            from . import _hostwalk as module
        except ImportError:
            module = None
        _hostwalk_module[0] = module
    return _hostwalk_module[0]


def _hostwalk_fell_back(where, exception):
    """The C walk raised and the Python loops take over: silent by default (an object tree it does not read is an ordinary case),
    but PDBEDA_DEBUG_HOSTWALK=1 says so on stderr -- a C module that fails on trees it SHOULD read must not hide behind its
    fallback -- and PDBEDA_DEBUG_HOSTWALK=raise turns the fallback into the error."""
    mode = os.environ.get("PDBEDA_DEBUG_HOSTWALK", "")
    if mode == "raise":
        raise exception
    if mode not in ("", "0"):
        print("pdb_eda_amd: _hostwalk.%s fell back to the Python loops: %s: %s" % (where, type(exception).__name__, exception), file=sys.stderr)


class Columns(object):
    """A columnar snapshot of a structure: what the analysis reads from the object tree (see the module text), gathered in
    ONE walk so that the per-entry host work runs on arrays instead of on 10^3..10^4 Python objects.

    Order is the order of ``structure.get_residues()`` / ``get_atoms()`` (the atoms of a residue are contiguous).
    Per residue: ``residues`` (the objects), ``res_model``, ``res_chain``, ``res_number``, ``res_name`` (lists),
    ``res_het`` (bool array: ``id[0] != ' '``), ``res_start`` (int64[n_res + 1], atom range of every residue).
    Per atom: ``atoms`` (the objects), ``name`` (list; ``atom_names`` the distinct ones and ``name_of_atom`` the int64 index into them), ``res_of_atom`` (int64), ``coord32`` (float32[n, 3], the stored
    coordinates), ``coord`` (float64, promoted exactly), ``occupancy``, ``bfactor`` (float64; ``occupancy_raw``: the objects as read), ``pair_of_atom`` (int64 index
    into ``pair_names``, the distinct 'RES_ATOM' names of densityAnalysis.residueAtomName).
    The snapshot is cached on the structure object: edit the tree and call ``columns(structure, refresh=True)``."""

    def __init__(self, structure, native=None):
        """``native``: None = use the C walk (``_hostwalk``) when it is built and understands the tree, else the Python loops;
        True / False force one of them (tests)."""
        residues = list(structure.get_residues())
        walked = None
        if native is not False:
            walk = _hostwalk()
            if walk is None and native:
                raise ImportError("pdb_eda_amd/_hostwalk.so is not built (python __graft_entry__.py)")
            if walk is not None:
                try:
                    res_model, res_chain, res_number, res_name, het, children = walk.residue_columns(residues)
                    walked = walk.atom_columns(children)
                except Exception as exception:
                    if native:
                        raise
                    _hostwalk_fell_back("Columns", exception)
                    walked = None                      # an object tree the C walk does not read: the loops below do
        if walked is not None:
            atoms, name, occupancy, counts, occ, bfac, xyz, name_id, distinct = walked[:9]
            if len(walked) > 9:
                self.__dict__.setdefault("_atom_lists", {})["coord"] = walked[9]      # (the atoms' coordinate objects, for the symmetry-atom tables)
            n = len(atoms)
            res_het = np.frombuffer(het, dtype=np.uint8).astype(bool)
            counts = np.frombuffer(counts, dtype=np.int64)
            atom_names = (distinct, np.frombuffer(name_id, dtype=np.int64))
            occupancy_f, bfactor_f = np.frombuffer(occ, dtype=np.float64), np.frombuffer(bfac, dtype=np.float64)
            coord32 = np.frombuffer(xyz, dtype=np.float32).reshape(n, 3)
        else:
            chains = [residue.parent for residue in residues]
            ids = [residue.id for residue in residues]
            res_model = [chain.parent.id for chain in chains]
            res_chain = [chain.id for chain in chains]
            res_number = [rid[1] for rid in ids]
            res_name = [residue.resname for residue in residues]
            res_het = np.asarray([rid[0] != ' ' for rid in ids], dtype=bool)
            children = [residue.child_list for residue in residues]
            atoms = list(itertools.chain.from_iterable(children))
            n = len(atoms)
            counts = np.fromiter(map(len, children), dtype=np.int64, count=len(residues))
            name = [atom.name for atom in atoms]
            atom_names = _first_appearance_ids(name)
            try:                                       # (the accessors of our atoms and of Bio.PDB's return these attributes)
                occupancy = [atom.occupancy for atom in atoms]
                bfactor = [atom.bfactor for atom in atoms]
            except AttributeError:
                occupancy = [atom.get_occupancy() for atom in atoms]
                bfactor = [atom.get_bfactor() for atom in atoms]
            occupancy_f, bfactor_f = np.asarray(occupancy, dtype=np.float64), np.asarray(bfactor, dtype=np.float64)
            coord = [atom.coord for atom in atoms]
            coord32 = (np.concatenate(coord).astype(np.float32, copy=False) if n else np.zeros(0, dtype=np.float32)).reshape(n, 3)
        res_start = np.zeros(len(residues) + 1, dtype=np.int64)
        np.cumsum(counts, out=res_start[1:])
        res_of_atom = np.repeat(np.arange(len(residues), dtype=np.int64), counts)
        # the distinct 'RES_ATOM' names, numbered by first appearance: ids of the (stripped) residue names and of the atom names
        # through C-level dict lookups, then one np.unique over the combined code -- no string is built per atom
        res_names = _first_appearance_ids([resname.strip() for resname in res_name])
        code = res_names[1][res_of_atom] * max(len(atom_names[0]), 1) + atom_names[1]
        distinct, first, inverse = np.unique(code, return_index=True, return_inverse=True)
        by_first = np.argsort(first, kind="stable")
        rank = np.empty(len(distinct), dtype=np.int64)
        rank[by_first] = np.arange(len(distinct))
        pair_names = [res_names[0][c // max(len(atom_names[0]), 1)] + '_' + atom_names[0][c % max(len(atom_names[0]), 1)] for c in distinct[by_first].tolist()]
        self.residues, self.atoms = residues, atoms
        self.res_model, self.res_chain, self.res_number, self.res_name = res_model, res_chain, res_number, res_name
        self.res_het = res_het
        self.res_start = res_start
        self.res_of_atom = res_of_atom
        self.name = name
        self.atom_names, self.name_of_atom = atom_names
        self.occupancy_raw = occupancy
        self.occupancy = occupancy_f
        self.bfactor = bfactor_f
        self.coord32 = coord32
        self.coord = self.coord32.astype(np.float64)
        self.pair_of_atom = rank[inverse.reshape(-1)] if n else np.zeros(0, dtype=np.int64)
        self.pair_names = pair_names

    def atom_lists(self, which):
        """Per-atom Python lists of residue-level columns ('model', 'chain', 'number', 'resname') -- and 'coord': the atoms' own
        coordinate objects --, made on first use."""
        cache = self.__dict__.setdefault("_atom_lists", {})
        if which == "coord" and which not in cache:
            cache[which] = [atom.coord for atom in self.atoms]
        if which not in cache:
            source = {"model": self.res_model, "chain": self.res_chain, "number": self.res_number, "resname": self.res_name}[which]
            column = np.fromiter(source, dtype=object, count=len(source))      # (the very objects, repeated by numpy)
            cache[which] = np.repeat(column, np.diff(self.res_start)).tolist()
        return cache[which]


def columns(structure, refresh=False):
    """The :class:`Columns` snapshot of ``structure`` (ours or a Bio.PDB one), built on first use and kept on the object."""
    cols = None if refresh else getattr(structure, "_pdbeda_columns", None)
    if cols is None:
        cols = Columns(structure)
        try:
            structure._pdbeda_columns = cols
        except AttributeError:
            pass
    return cols


def read_pdb(handle_or_path, structure_id="xxxx"):
    """Parse ATOM/HETATM + the REMARK 290 SMTRY operators of a PDB-format file.

    Returns ``(Structure, PDBEntry)``.  First model only; for alternate locations the
    highest-occupancy (first on ties) conformer of an atom name is kept, as Bio.PDB's
    DisorderedAtom selection does; hetero flag 'W' for waters and 'H_<res>' otherwise.
    """
    import gzip
    if isinstance(handle_or_path, str):
        opener = gzip.open if handle_or_path.endswith(".gz") else open
        with opener(handle_or_path, "rt") as fh:
            lines = fh.readlines()
    else:
        lines = handle_or_path.readlines()
    st = Structure(structure_id)
    model = Model(0, st)
    chains = {}
    residues = {}
    atom_slots = {}
    # Header fields exactly as the reference's pdbParser.parse reads them in its 'lite' mode (pdbParser.py:24-98; pinned by
    # tests/golden/pdbheader.json): every field starts as 0, the date is the two characters at columns 58-59, blanks inside
    # names become '_', header records count only up to the first record that starts with 'ATOM', and a file without
    # REMARK 290 has NO operators (an empty list, so no symmetry atoms -- not an identity).
    rot = []
    pdbid = date = method = resolution = r_value = r_free = program = space_group = 0
    header_open = True
    n_models = 0
    for rec in lines:
        tag = rec[:6]
        if rec.startswith("ATOM"):
            header_open = False
        if header_open and tag != "HETATM":
            if tag == "HEADER":
                date = rec[57:59].strip()
                pdbid = rec[62:66].strip()
            elif tag == "EXPDTA":
                method = rec[6:36].strip().replace(" ", "_")
            elif rec.startswith("REMARK   2 RESOLUTION"):
                m = re.search("RESOLUTION.(.+)ANGSTROMS", rec)
                if m:
                    resolution = m.group(1).strip()
            elif rec.startswith("REMARK   3   R VALUE"):
                m = re.search(r"^REMARK   3   R VALUE            \(WORKING SET\) : (.+)$", rec)
                if m:
                    r_value = m.group(1).strip()
            elif rec.startswith("REMARK   3   FREE R VALUE"):
                m = re.search(r"^REMARK   3   FREE R VALUE                     : (.+)$", rec)
                if m:
                    r_free = m.group(1).strip()
            elif rec.startswith("REMARK   3   PROGRAM"):
                m = re.search(r"^REMARK   3   PROGRAM     : (.+)$", rec)
                if m:
                    program = m.group(1).strip().replace(" ", "_")
            elif rec.startswith("REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP:"):
                m = re.search(r"^REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP: (.+)$", rec)
                if m:
                    space_group = m.group(1).strip().replace(" ", "_")
            elif rec.startswith("REMARK 290   SMTRY"):
                items = rec[18:].split()
                row, op = int(items[0]) - 1, int(items[1]) - 1
                while len(rot) <= op:
                    rot.append(np.zeros((3, 4)))
                rot[op][row] = [float(x) for x in items[2:6]]
        if tag == "MODEL ":
            n_models += 1
            if n_models > 1:
                break
        elif tag in ("ATOM  ", "HETATM"):
            name = rec[12:16].strip()
            altloc = rec[16]
            resname = rec[17:20]
            chain_id = rec[21]
            resseq = int(rec[22:26])
            icode = rec[26]
            xyz = np.array([float(rec[30:38]), float(rec[38:46]), float(rec[46:54])], dtype=np.float32)
            occ_txt = rec[54:60].strip()
            occ = float(occ_txt) if occ_txt else 1.0
            b_txt = rec[60:66].strip()
            bfac = float(b_txt) if b_txt else 0.0
            element = rec[76:78].strip().upper() if len(rec) > 76 else ""
            if tag == "HETATM":
                het = "W" if resname.strip() in ("HOH", "WAT") else "H_" + resname.strip()
            else:
                het = " "
            if chain_id not in chains:
                chains[chain_id] = Chain(chain_id, model)
            rkey = (chain_id, het, resseq, icode)
            if rkey not in residues:
                residues[rkey] = Residue((het, resseq, icode), resname, chains[chain_id])
            akey = rkey + (name,)
            prev = atom_slots.get(akey)
            if prev is None:
                atom_slots[akey] = Atom(name, xyz, occ, bfac, element, residues[rkey], int(rec[6:11] or 0), altloc)
            elif occ > prev.occupancy:
                prev.coord, prev.occupancy, prev.bfactor, prev.altloc = xyz, occ, bfac, altloc
    st.header = {"resolution": float(resolution) if resolution not in (0, "") else None}
    hdr = PDBHeader(pdbid=pdbid, resolution=resolution, spaceGroup=space_group, rotationMats=rot, date=date, method=method, rValue=r_value, rFree=r_free,
                    program=program)
    return st, PDBEntry(hdr)
