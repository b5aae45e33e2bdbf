# Diagnostic (library built with -DPDBEDA_COUNT_FIND): find calls / steps / unite retries of ONE labelling step.
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
which = sys.argv[1] if len(sys.argv) > 1 else "noise"
edge = 256
if which == "noise":
    spec = synthetic.MapSpec(ncrs=(edge,) * 3, spacing=0.4)
    dens = synthetic.smooth_noise((edge,) * 3, seed=7, sigma_voxels=1.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
else:
    spec = synthetic.MapSpec(ncrs=(edge,) * 3, spacing=0.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    lo, hi = np.array(header.crs2xyzCoord([6, 6, 6])), np.array(header.crs2xyzCoord([edge - 7] * 3))
    st = synthetic.chain_structure(6000, 5, lo, hi)
    dens = synthetic.gaussian_sum_grid(header, st, synthetic.synthetic_params()["full_atom_name_map_electrons"], sigma=0.55, noise=0.02, seed=5)
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, dens, header.geometry())
mean, std = dmap.stats()
cut = mean + 1.5 * std
lib = _native.lib()
out = (C.c_ulonglong * 6)()
k = dmap.full_blobs(cut, labels=True); ctx.synchronize()
lib.pdbeda_debug_find_counters(out, 1)
k = dmap.full_blobs(cut, labels=True); ctx.synchronize()
lib.pdbeda_debug_find_counters(out, 1)
print(which, "unites %d mean ticks %.0f max ticks %d;" % (out[5], out[4] / max(out[5], 1), out[3]), "max retries of one unite %d; find steps %d, unite retries %d blobs %d" % (out[0] >> 40, out[1], out[2], len(k)))
