import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
VOXEL_CASES = ["orth", "orth_sub", "orth_rep", "orth_perm", "hex", "tric", "wide"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_case(name):
    """Golden fixture -> (npz dict, header, grid) using the PRODUCT's host-side parser."""
    import io
    from pdb_eda_amd import ccp4
    z = np.load(os.path.join(GOLDEN, "voxel_%s.npz" % name))
    header, grid = ccp4.read_grid(io.BytesIO(z["ccp4_bytes"].tobytes()))
    grid = np.ascontiguousarray(grid.astype(np.float32)).reshape(header.ncrs[2], header.ncrs[1], header.ncrs[0])
    return z, header, grid


@pytest.fixture(scope="session")
def gpu_ctx():
    from pdb_eda_amd import _native
    return _native.default_context()


def blobs_from_record(z, prefix):
    off = z[prefix + "_off"]
    crs = z[prefix + "_crs"]
    return [{"crs": crs[off[i]:off[i + 1]], "totalDensity": z[prefix + "_total"][i], "centroid": z[prefix + "_centroid"][i],
             "coordCenter": z[prefix + "_center"][i], "volume": z[prefix + "_volume"][i]} for i in range(len(off) - 1)]


def crs_set(a):
    return {tuple(int(x) for x in v) for v in np.asarray(a).reshape(-1, 3)}


ANALYSIS_CASES = ["orth", "hex", "alias"]   # alias: atoms that share a coordinate (round 4)


def load_analysis_case(name):
    """Analysis golden fixture -> (npz, MapSpec, Structure, PDBEntry, params)."""
    import json
    from pdb_eda_amd import synthetic, structure
    z = np.load(os.path.join(GOLDEN, "analysis_%s.npz" % name))
    spec = synthetic.MapSpec(**json.loads(str(z["spec"])))
    st = structure.Structure("synth")
    model = structure.Model(0, st)
    chain = structure.Chain("A", model)
    res = None
    last = None
    for i in range(len(z["atom_name"])):
        key = (str(z["atom_het"][i]), int(z["atom_resnum"][i]))
        if key != last:
            res = structure.Residue((key[0], key[1], " "), str(z["atom_resname"][i]), chain)
            last = key
        structure.Atom(str(z["atom_name"][i]), z["atom_coord"][i], float(z["atom_occ"][i]), float(z["atom_b"][i]), str(z["atom_element"][i]), res, i + 1)
    pdb = structure.PDBEntry(structure.PDBHeader(pdbid=name, resolution=2.0, spaceGroup="P_1", rotationMats=[m for m in z["rot"]]))
    return z, spec, st, pdb, synthetic.synthetic_params()
