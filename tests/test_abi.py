"""The drop-in boundary without a GPU: libpdbeda_hip.so loads and exports every function include/pdbeda.h declares, the
ctypes stub (pdb_eda_amd/_native.py, the binding INTEGRATION.md shows) binds exactly those, and the product fails loudly
instead of falling back when no device is usable.  No compute call is made here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "pdbeda.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdbeda_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as entry
    entry.build()
    from pdb_eda_amd import _native
    handle = ctypes.CDLL(_native.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for name in names:
        assert hasattr(handle, name), "%s is declared in include/pdbeda.h but not exported" % name
    assert sorted(_native.EXPORTED_SYMBOLS) == names, "the ctypes stub and the header disagree"
    assert handle.pdbeda_version is not None and b"gfx950" in ctypes.c_char_p(ctypes.cast(handle.pdbeda_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()).value


def test_no_cpu_fallback():
    """Without a device every entry point refuses: the product never routes through the oracle or any CPU path."""
    from pdb_eda_amd import _native
    if _native.lib().pdbeda_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.PdbedaError):
        _native.Context(0)
    import pdb_eda_amd
    src = ""
    for dirpath, _, files in os.walk(os.path.dirname(pdb_eda_amd.__file__)):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                src += open(os.path.join(dirpath, f)).read()
    assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) and "libpdbeda_oracle" not in src


def test_numa_helpers_degrade_without_a_device():
    """``device_local_cpus`` / ``pin_to_device`` never raise: no GPU, no sysfs or a single node leave the affinity alone."""
    from pdb_eda_amd import _native
    before = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    cpus = _native.device_local_cpus(0)
    assert cpus is None or (isinstance(cpus, set) and cpus)
    kept = _native.pin_to_device(0)
    assert isinstance(kept, int) and kept >= 0
    if before is not None:
        if kept == 0:
            assert os.sched_getaffinity(0) == before
        os.sched_setaffinity(0, before)


def _die(_):
    os._exit(3)


def _double(x):
    return 2 * x


def test_process_pool_survives_nothing_silently():
    """The worker pool without a GPU: plain functions run on the spawned workers; a worker that dies raises PdbedaError in the
    parent instead of hanging the map (multiprocessing.Pool would wait for the lost task for ever)."""
    from pdb_eda_amd import _native, multipleStructures
    pool = multipleStructures.ProcessPool(device=0, n_workers=2)
    try:
        assert pool.run(_double, range(5)) == [0, 2, 4, 6, 8]
        with pytest.raises(_native.PdbedaError):
            pool.run(_die, range(2))
    finally:
        pool.close()
