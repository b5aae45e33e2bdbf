"""Run-to-run bit stability: the blob sums are folded by atomics whose order the hardware picks, and
aggregateCloud DECIDES with them (best cloud, centroid-distance cut-off, pooling).  They are order-independent integers now
(FixSums, pdbeda_kernels.h): 20 repetitions on 3 concurrent streams give bit-identical tables -- whole-map blob lists and every
table of pdbeda_aggregate_cloud."""
import io
import threading

import numpy as np
import pytest

from conftest import load_analysis_case

pytestmark = pytest.mark.gpu
REPS, STREAMS = 20, 3


def _run_streams(make, work):
    """make(k) -> state of stream k (own context); work(state) -> comparable result; REPS repetitions per stream, all streams at once."""
    states = [make(k) for k in range(STREAMS)]
    results = [[None] * REPS for _ in range(STREAMS)]
    errors = []

    def lane(k):
        try:
            for r in range(REPS):
                results[k][r] = work(states[k])
        except BaseException as e:       # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=lane, args=(k,)) for k in range(STREAMS)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return [res for lane_results in results for res in lane_results]


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert a.keys() == b.keys(), path
        for k in a:
            _same(a[k], b[k], path + "/" + str(k))
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, path + "[%d]" % i)
    else:
        a, b = np.asarray(a), np.asarray(b)
        assert a.shape == b.shape and a.tobytes() == b.tobytes(), "not bit-identical: " + path


def test_whole_map_blob_tables_are_bit_identical():
    from pdb_eda_amd import _native, ccp4, synthetic
    n = 160
    spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
    grid = synthetic.smooth_noise((n, n, n), seed=21, sigma_voxels=1.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))

    def make(k):
        ctx = _native.Context(0)
        dmap = _native.DeviceMap(ctx, grid, header.geometry())
        mean, std = dmap.stats()
        return ctx, dmap, mean + 1.5 * std

    def work(state):
        ctx, dmap, cut = state
        out = []
        for bl in dmap.full_blobs_pm(cut, -cut, labels=True):
            st = bl.stats()
            out.append({k: np.array(st[k]) for k in ("n", "totalDensity", "centroid", "coordCenter", "firstKey")})
            bl.free()
        return out
    res = _run_streams(make, work)
    assert len(res[0][0]["n"]) > 3000          # a real table: thousands of blobs, most of them folded across tiles
    for other in res[1:]:
        _same(res[0], other)


def test_aggregate_cloud_tables_are_bit_identical():
    from pdb_eda_amd import _native, ccp4, synthetic, densityAnalysis
    z, spec, st, pdb, params = load_analysis_case("orth")
    densityAnalysis.setGlobals(params)
    blob = synthetic.ccp4_bytes(spec, z["dens"])

    def make(k):
        ctx = _native.Context(0)
        dens = ccp4.parse(io.BytesIO(blob), "det", ctx=ctx)
        densityAnalysis._attachCutoffs(dens, None)
        an = densityAnalysis.DensityAnalysis("det", dens, None, st, pdb)
        inp = an._cloudInputs()
        return dens, (inp["xyz"], inp["radius"], inp["electrons"] * inp["occupancy"], inp["residue"], inp["alias"], inp["key"], inp["bonded_off"],
                      inp["bonded"], inp["owner_key"])

    def work(state):
        dens, args = state
        return dens._map.aggregate_cloud(*args, dens.densityCutoff, 25.0)
    res = _run_streams(make, work)
    assert len(res[0]["atom"]) > 100
    for other in res[1:]:
        _same(res[0], other)
