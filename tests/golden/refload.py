"""Load the *reference* pdb_eda (read-only at /root/reference) in the build container.

TEST INFRASTRUCTURE ONLY.  This is how the golden vectors under tests/golden/ were
produced; it is never imported by the product, by `-m gpu` tests, by smoke() or by
bench.py (``/root/reference`` does not exist on the GPU box).

Recipe:
  * the reference's only native module, ``pdb_eda/cutils.pyx``, is cythonized from
    where it lies with the same flags as the reference's setup.py:42-44 (``-O3``) and
    the build products go to ``$TMPDIR/pdbeda_pyref`` (``PDBEDA_PYREF``), outside the repository;
  * ``import pdb_eda`` itself fails here (biopython/docopt are not installed), so an
    empty package module whose ``__path__`` is [built cutils dir, reference dir] is
    registered and ``pdb_eda.ccp4`` / ``pdb_eda.densityAnalysis`` are imported from it;
  * ``Bio.PDB`` is replaced by an empty stub (only ``PDBParser`` is referenced, in the
    file readers we do not call).
"""
import importlib
import os
import sys
import types

REF_ROOT = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# build products of the reference's Cython module: OUTSIDE the repository (round 4 -- the cythonized cutils.c is derived from
# the reference and must not sit under the repo root, ignored or not)
PYREF = os.environ.get("PDBEDA_PYREF") or os.path.join(os.environ.get("TMPDIR") or "/tmp", "pdbeda_pyref")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "pdb_eda"))


def build_cutils():
    import numpy
    from setuptools import Extension
    from setuptools.dist import Distribution
    from Cython.Build import cythonize

    lib = os.path.join(PYREF, "lib")
    tmp = os.path.join(PYREF, "build")
    os.makedirs(tmp, exist_ok=True)
    ext = Extension("pdb_eda.cutils", sources=[os.path.join(REF_ROOT, "pdb_eda", "cutils.pyx")],
                    extra_compile_args=["-O3"], include_dirs=[numpy.get_include()])
    exts = cythonize([ext], build_dir=tmp, language_level=3, quiet=True)
    dist = Distribution({"ext_modules": exts})
    cmd = dist.get_command_obj("build_ext")
    cmd.build_lib = lib
    cmd.build_temp = tmp
    cmd.ensure_finalized()
    cmd.run()
    return os.path.join(lib, "pdb_eda")


def load(with_density_analysis=True, use_cython=True):
    """Return (ccp4_module, densityAnalysis_module_or_None) of the reference."""
    if "pdb_eda" in sys.modules and getattr(sys.modules["pdb_eda"], "_graft_ref", False):
        pkg = sys.modules["pdb_eda"]
        return pkg.ccp4, getattr(pkg, "densityAnalysis", None)
    path = [os.path.join(REF_ROOT, "pdb_eda")]
    if use_cython:
        path.insert(0, build_cutils())
    pkg = types.ModuleType("pdb_eda")
    pkg.__path__ = path
    pkg._graft_ref = True
    sys.modules["pdb_eda"] = pkg
    ccp4 = importlib.import_module("pdb_eda.ccp4")
    pkg.ccp4 = ccp4
    if use_cython:
        assert ccp4.utils.__name__ == "pdb_eda.cutils", ccp4.utils.__name__
    da = None
    if with_density_analysis:
        if "Bio" not in sys.modules:
            bio = types.ModuleType("Bio")
            biopdb = types.ModuleType("Bio.PDB")
            biopdb.PDBParser = object
            bio.PDB = biopdb
            sys.modules["Bio"] = bio
            sys.modules["Bio.PDB"] = biopdb
        da = importlib.import_module("pdb_eda.densityAnalysis")
        pkg.densityAnalysis = da
    return ccp4, da
