"""pdb_eda_amd -- MI355X-native electron-density voxel core behind pdb_eda's API surface.

Only the accelerated hot path of pdb_eda lives here: CCP4 grid ->
HBM, significant-density test + 26-neighbour blob labelling, CRS<->XYZ, per-atom sphere
gathers, atom/residue regional sums -- all executed by ``libpdbeda_hip.so`` (HIP, gfx950).
"""
__version__ = "0.1.0"

from . import _native  # noqa: F401  (ctypes binding; loading the .so is deferred to first use)


def fromFile(*args, **kwargs):
    from .densityAnalysis import fromFile as _f
    return _f(*args, **kwargs)


def fromPDBid(*args, **kwargs):
    from .densityAnalysis import fromPDBid as _f
    return _f(*args, **kwargs)
