"""Golden vectors for the header half of the minimal PDB reader: the REFERENCE's pdbParser.parse (lite mode, the
mode densityAnalysis.fromFile uses, pdbParser.py:24-98) on synthetic PDB header texts.  The texts are written here (no real
entry is copied); the fixture holds them and the reference's parsed fields.  Build container only:
    python tests/golden/make_golden_pdbheader.py   -> tests/golden/pdbheader.json
The ATOM / HETATM half (Bio.PDB's atom order and altloc choice) stays unpinned: Bio is not installed anywhere in reach."""
import importlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import refload  # noqa: E402

ATOMS = ("ATOM      1  N   ALA A   1      11.104   6.134  -6.504  1.00 10.00           N\n"
         "ATOM      2  CA  ALA A   1      11.639   6.071  -5.147  1.00 11.00           C\n")


def smtry(op, rows):
    return "".join("REMARK 290   SMTRY%d %3d %9.6f %9.6f %9.6f %14.5f\n" % (r + 1, op, *rows[r]) for r in range(3))


CASES = {
    "p212121": ("HEADER    HYDROLASE                               17-MAR-99   1XYZ\n"
                "EXPDTA    X-RAY DIFFRACTION\n"
                "REMARK   2 RESOLUTION.    1.80 ANGSTROMS.\n"
                "REMARK   3   PROGRAM     : REFMAC 5.8.0135\n"
                "REMARK   3   R VALUE            (WORKING SET) : 0.187\n"
                "REMARK   3   FREE R VALUE                     : 0.221\n"
                "REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP: P 21 21 21\n"
                + smtry(1, [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0)])
                + smtry(2, [(-1, 0, 0, 26.35), (0, -1, 0, 0), (0, 0, 1, 40.455)])
                + smtry(3, [(-1, 0, 0, 0), (0, 1, 0, 31.2), (0, 0, -1, 40.455)])
                + smtry(4, [(1, 0, 0, 26.35), (0, -1, 0, 31.2), (0, 0, -1, 0)])
                + "MODEL        1\n" + ATOMS + "ENDMDL\n"),
    "no_resolution": ("HEADER    DNA                                     02-FEB-01   2ABC\n"
                      "EXPDTA    SOLUTION NMR\n"
                      "REMARK   2 RESOLUTION. NOT APPLICABLE.\n"
                      "REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP: P 1\n"
                      + smtry(1, [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0)]) + ATOMS),
    "hexagonal_blank_in_group": ("HEADER    TRANSFERASE                             30-NOV-12   4HEX\n"
                                 "EXPDTA    X-RAY DIFFRACTION\n"
                                 "REMARK   2 RESOLUTION.    2.35 ANGSTROMS.\n"
                                 "REMARK   3   PROGRAM     : PHENIX (PHENIX.REFINE: 1.8_1069)\n"
                                 "REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP: P 65 2 2\n"
                                 + smtry(1, [(1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0)])
                                 + smtry(2, [(0.5, -0.866025, 0, 0), (0.866025, 0.5, 0, 0), (0, 0, 1, 81.66667)])
                                 + smtry(3, [(-0.5, -0.866025, 0, 0), (0.866025, -0.5, 0, 0), (0, 0, 1, 65.33333)])
                                 + ATOMS),
    "odd_spacing_and_order": ("HEADER    X                                                     \n"
                              "REMARK   2 RESOLUTION.  0.95  ANGSTROMS.\n"
                              "REMARK 290 SYMMETRY OPERATORS FOR SPACE GROUP:   C 1 2 1  \n"
                              "REMARK 290   SMTRY1   1  1.000000  0.000000  0.000000        0.00000\n"
                              "REMARK 290   SMTRY1   2 -1.000000  0.000000  0.000000        0.00000\n"
                              "REMARK 290   SMTRY2   1  0.000000  1.000000  0.000000        0.00000\n"
                              "REMARK 290   SMTRY2   2  0.000000  1.000000  0.000000        0.00000\n"
                              "REMARK 290   SMTRY3   1  0.000000  0.000000  1.000000        0.00000\n"
                              "REMARK 290   SMTRY3   2  0.000000  0.000000 -1.000000        0.00000\n"
                              "HETATM    1  O   HOH A 101       5.000   5.000   5.000  1.00 30.00           O\n"
                              "REMARK   3   FREE R VALUE                     : 0.150\n" + ATOMS
                              + "REMARK   3   R VALUE            (WORKING SET) : 0.999\n"),
    "no_remarks": ("HEADER    EMPTY                                   01-JAN-70   9ZZZ\n" + ATOMS),
}


def main():
    refload.load(with_density_analysis=False)
    pp = importlib.import_module("pdb_eda.pdbParser")
    out = {}
    for name, text in CASES.items():
        h = pp.parse(io.StringIO(text)).header
        out[name] = {"text": text,
                     "header": {"pdbid": h.pdbid, "date": h.date, "method": h.method, "resolution": h.resolution, "rValue": h.rValue, "rFree": h.rFree,
                                "program": h.program, "spaceGroup": h.spaceGroup, "rotationMats": [m.tolist() for m in h.rotationMats]}}
        print(name, {k: v for k, v in out[name]["header"].items() if k != "rotationMats"}, len(h.rotationMats))
    with open(os.path.join(HERE, "pdbheader.json"), "w") as fh:
        json.dump(out, fh, indent=1)
        fh.write("\n")


if __name__ == "__main__":
    main()
