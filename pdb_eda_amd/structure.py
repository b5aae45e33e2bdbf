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
import numpy as np

__all__ = ["Structure", "Model", "Chain", "Residue", "Atom", "PDBHeader", "PDBEntry", "read_pdb"]


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
    rot = []
    resolution = 0
    space_group = 0
    pdbid = ""
    n_models = 0
    for rec in lines:
        tag = rec[:6]
        if tag == "HEADER":
            pdbid = rec[62:66].strip()
        elif rec.startswith("REMARK   2 RESOLUTION"):
            body = rec.split("RESOLUTION.", 1)[-1]
            if "ANGSTROMS" in body:
                resolution = body.split("ANGSTROMS")[0].strip()
        elif rec.startswith("REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP:"):
            space_group = rec.split(":", 1)[1].strip().replace(" ", "_")
        elif rec.startswith("REMARK 290   SMTRY"):
            items = rec[18:].split()
            row, op = int(items[0]) - 1, int(items[1]) - 1
            while len(rot) <= op:
                rot.append(np.zeros((3, 4)))
            rot[op][row] = [float(x) for x in items[2:6]]
        elif tag == "MODEL ":
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
    hdr = PDBHeader(pdbid=pdbid, resolution=resolution, spaceGroup=space_group, rotationMats=rot or None)
    return st, PDBEntry(hdr)
