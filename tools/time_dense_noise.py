# Whole-map step on smooth noise at dense cutoffs (tiles over their run slots / component slots): python tools/time_dense_noise.py [edge]
import sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
edge = int(sys.argv[1]) if len(sys.argv) > 1 else 256
spec = synthetic.MapSpec(ncrs=(edge,) * 3, spacing=0.4)
grid = synthetic.smooth_noise((edge,) * 3, seed=7, sigma_voxels=1.5)
header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
ctx = _native.Context(0)
dmap = _native.DeviceMap(ctx, grid, header.geometry())
mean, std = dmap.stats()
for nsd in (1.5, 1.35, 1.25, 1.15, 1.0, 0.75, 0.5, 0.25):
    cut = mean + nsd * std
    for _ in range(3):
        g, r = dmap.full_blobs_pm(cut, -cut, labels=True)
        len(g)
    ctx.profile_begin()
    g, r = dmap.full_blobs_pm(cut, -cut, labels=True)
    n = len(g) + len(r)
    prof = ctx.profile_end()
    c = g.counters()
    print("nsd %.2f blobs %7d  step %7.1f us  %s  wide %d unit(runs) %d unit(ids) %d reruns %d" % (
        nsd, n, 1e3 * sum(ms for _, ms in prof.values()), {k: round(1e3 * ms, 1) for k, (_, ms) in sorted(prof.items())},
        c["wide_tiles"], c["unit_tiles_runs"], c["unit_tiles_comps"], c["reruns"]))
