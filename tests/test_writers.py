"""Result writers (ref singleStructure.py:169-178, multipleStructures.py:182-194): text layout on CPU, and the row
builders of every ``pdb_eda single`` sub-mode on the MI355X path."""
import csv
import io
import json

import numpy as np
import pytest


def test_single_writer_text_layout():
    from pdb_eda_amd import singleStructure
    header = ["chain", "residue_number", "xyz", "value"]
    rows = [["A", 7, [1.0, 2.5, -3.0], 0.25], ["B", np.int64(8), [0.0, 0.0, 0.0], np.float64(1.5)]]
    text = singleStructure.dumps(header, rows, "csv")
    assert text.splitlines()[0] == "chain,residue_number,xyz,value"
    assert text.splitlines()[1] == "A,7,[1.0, 2.5, -3.0],0.25"          # str() of every cell, like the reference
    back = json.loads(singleStructure.dumps(header, rows, "json"))
    assert back == [{"chain": "A", "residue_number": 7, "xyz": [1.0, 2.5, -3.0], "value": 0.25},
                    {"chain": "B", "residue_number": 8, "xyz": [0.0, 0.0, 0.0], "value": 1.5}]
    assert singleStructure.dumps(header, rows, "json").startswith('[\n  {\n    "chain": "A",')   # indent 2, sorted keys


def test_multiple_writer_text_layout(tmp_path):
    from pdb_eda_amd import densityAnalysis, multipleStructures, synthetic
    densityAnalysis.setGlobals(synthetic.synthetic_params())
    types = sorted(densityAnalysis.paramsGlobal["radii"])
    rec = {"pdbid": "1abc", "diffs": {t: 0.01 * i for i, t in enumerate(types)},
           "stats": {h: i for i, h in enumerate(multipleStructures.statsHeaders)}}
    out = tmp_path / "r.csv"
    multipleStructures.writeResults({"1abc": rec}, str(out), "csv")
    table = list(csv.reader(io.StringIO(out.read_text())))
    assert table[0] == ["pdbid"] + multipleStructures.statsHeaders + types
    assert table[1][0] == "1abc" and len(table[1]) == len(table[0])
    out = tmp_path / "r.json"
    multipleStructures.writeResults({"1abc": rec}, str(out), "json")
    assert json.loads(out.read_text()) == {"1abc": rec}


@pytest.mark.gpu
def test_single_structure_rows(gpu_ctx):
    from conftest import load_analysis_case
    from pdb_eda_amd import ccp4, synthetic, densityAnalysis, singleStructure
    z, spec, st, pdb, params = load_analysis_case("orth")
    densityAnalysis.setGlobals(params)
    dens = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["dens"])), "orth", ctx=gpu_ctx)
    diff = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, z["diff"])), "orth", ctx=gpu_ctx)
    densityAnalysis._attachCutoffs(dens, diff)
    an = densityAnalysis.DensityAnalysis("orth", dens, diff, st, pdb)
    st.header = {"resolution": 2.0}
    # difference / atom: the golden discrepancy columns behind the reference's header
    header, rows = singleStructure.rows(an, "difference", "atom", radius=3.5)
    assert header == densityAnalysis.DensityAnalysis.atomRegionDiscrepancyHeader
    assert np.allclose(np.array([r[6:] for r in rows], dtype=np.float64), z["atom_discrepancy"], rtol=1e-8, atol=1e-12)
    back = json.loads(singleStructure.dumps(header, rows, "json"))
    assert len(back) == len(rows) and set(back[0]) == set(header)
    # every sub-mode builds, serialises to both formats, and has as many cells as headers
    for mode, level, kw in (("cloud", "atom", {}), ("cloud", "residue", {}), ("cloud", "domain", {}), ("density", "atom", {"radius": 1.0}),
                            ("density", "residue", {"radius": 1.2}), ("density", "symmetry-atom", {"radius": 1.0}), ("difference", "residue", {}),
                            ("difference", "symmetry-atom", {}), ("blob", "atom", {"green": True, "red": True}), ("blob", "atom", {"green": True}),
                            ("blob", "atom", {}), ("statistics", "atom", {}), ("statistics", "residue", {})):
        header, rows = singleStructure.rows(an, mode, level, includePdbid=True, **kw)
        assert rows and all(len(r) == len(header) for r in rows), (mode, level)
        assert header[0] == "pdbid" and rows[0][0] == "orth"
        json.loads(singleStructure.dumps(header, rows, "json"))
        assert len(singleStructure.dumps(header, rows, "csv").splitlines()) == len(rows) + 1
    # the fused green+red call equals the two separate lists, in the reference's order (green rows first)
    _, both = singleStructure.rows(an, "blob", green=True, red=True)
    _, g = singleStructure.rows(an, "blob", green=True)
    _, r = singleStructure.rows(an, "blob", red=True)
    assert json.dumps(both, default=float) == json.dumps(g + r, default=float)
    assert "Relative Difference" in singleStructure.validationLine(an)
