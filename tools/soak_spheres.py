# Soak test of the sphere batches (findAberrantBlobs over groups of atoms, regional sums) against the oracle: random cells
# (orthogonal / triclinic, permuted axes, crsStart != 0, interval > ncrs), atoms inside, at the edges and outside the box,
# radii from sub-voxel to several words wide, all three cutoff regimes.
#   python tools/soak_spheres.py [cases] [seed]
import sys, os, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic
from oracle import oracle as ora

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
ctx = _native.Context(0)
t0 = time.time()
bad = 0
checked = 0
for k in range(n_cases):
    nc, nr, ns = int(rng.integers(6, 150)), int(rng.integers(6, 60)), int(rng.integers(6, 60))
    kind = k % 4
    kw = {}
    if kind == 1:
        kw = dict(angles=(90.0, 90.0, 120.0))
    elif kind == 2:
        kw = dict(angles=(82.0, 97.0, 110.0), axis_order=(2, 1, 3), crs_start=(-3, 5, 2))
    elif kind == 3:
        order = (3, 1, 2)
        ncrs = (nc, nr, ns)
        interval = [0, 0, 0]
        for crs_axis, xyz_axis in enumerate(order):
            interval[xyz_axis - 1] = ncrs[crs_axis] + int(rng.integers(0, 9))      # interval > ncrs: part of the cell is not stored
        kw = dict(axis_order=order, interval=interval, crs_start=(4, -2, 0))
    spacing = float(rng.choice([0.3, 0.45, 0.7]))
    spec = synthetic.MapSpec(ncrs=(nc, nr, ns), spacing=spacing, **kw)
    g = synthetic.smooth_noise((ns, nr, nc), 9000 + k, float(rng.choice([0.8, 1.6])))
    dm = ccp4.parse(io.BytesIO(synthetic.ccp4_bytes(spec, g)), "soak", ctx=ctx)
    o = ora.Oracle(dm.header, g)
    n_atoms = int(rng.integers(1, 40))
    crs = np.stack([rng.integers(-6, nc + 6, n_atoms), rng.integers(-6, nr + 6, n_atoms), rng.integers(-6, ns + 6, n_atoms)], axis=1).astype(np.int32)
    xyz = np.asarray(o.crs2xyz(crs), dtype=np.float64) + rng.normal(0.0, 0.2, (n_atoms, 3))
    xyz = xyz.astype(np.float32).astype(np.float64)                                  # atom coordinates are float32 in the structure
    radii = rng.choice([0.2, 0.9, 1.7, 2.6, 3.5, 5.0], n_atoms).astype(np.float32)
    cuts = [0.0, float(dm.meanDensity + 1.0 * dm.stdDensity), -float(dm.meanDensity + 1.2 * dm.stdDensity)]
    # groups: singles, then random runs
    sizes = []
    left = n_atoms
    while left > 0:
        s = int(min(left, rng.choice([1, 1, 2, 5, 9])))
        sizes.append(s)
        left -= s
    goff = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    for cut in cuts:
        bl = dm._map.sphere_blobs(xyz, radii, goff, cut)
        st = bl.stats()
        vox, voff = bl.voxels()
        for gi in range(len(sizes)):
            a, b = int(goff[gi]), int(goff[gi + 1])
            want = o.find_aberrant_blobs(xyz[a:b], radii[a:b], cut)
            rows = np.nonzero(st["group"] == gi)[0]
            checked += 1
            got_sets = sorted(tuple(sorted(map(tuple, vox[voff[r]:voff[r + 1]].tolist()))) for r in rows)
            want_sets = sorted(tuple(sorted(map(tuple, np.asarray(w["crs"]).reshape(-1, 3).tolist()))) for w in want)
            ok = got_sets == want_sets
            if ok and len(want):
                wt = sorted((len(np.asarray(w["crs"]).reshape(-1, 3)), w["totalDensity"]) for w in want)
                gt = sorted((int(st["n"][r]), float(st["totalDensity"][r])) for r in rows)
                ok = all(x[0] == y[0] and abs(x[1] - y[1]) <= 1e-9 * max(1.0, abs(y[1])) for x, y in zip(gt, wt))
            if not ok:
                bad += 1
                print("MISMATCH blobs: case", k, "kind", kind, (ns, nr, nc), "cut", cut, "group", gi, flush=True)
        bl.free()
    cut = abs(cuts[1])
    pos, neg, cnt, valid = dm._map.region_sums(xyz, radii, goff, cut)
    for gi in range(len(sizes)):
        a, b = int(goff[gi]), int(goff[gi + 1])
        sel = o.sphere_crs_list(xyz[a:b], radii[a:b], 0.0) if b - a > 1 else o.sphere_crs(xyz[a], radii[a], 0.0)
        d = np.array([o.point_density(v) for v in np.asarray(sel).reshape(-1, 3)], dtype=np.float64)
        c32 = float(np.float32(cut))
        wp, wn = float(d[d > c32].sum()), float(d[d < -c32].sum())
        wv = all(o.valid_xyz(xyz[i], radii[i]) for i in range(a, b))
        checked += 1
        if cnt[gi] != len(sel) or abs(pos[gi] - wp) > 1e-9 * max(1.0, abs(wp)) or abs(neg[gi] - wn) > 1e-9 * max(1.0, abs(wn)) or bool(valid[gi]) != wv:
            bad += 1
            print("MISMATCH region: case", k, "kind", kind, (ns, nr, nc), "group", gi, (cnt[gi], len(sel)), (pos[gi], wp), (neg[gi], wn), (bool(valid[gi]), wv), flush=True)
    if k % 20 == 0:
        print("case", k, "kind", kind, (ns, nr, nc), "%d checks, %.0f s" % (checked, time.time() - t0), flush=True)
print("done: %d cases, %d checks, %d mismatches, %.0f s" % (n_cases, checked, bad, time.time() - t0))
sys.exit(1 if bad else 0)
