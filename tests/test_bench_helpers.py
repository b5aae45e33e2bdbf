"""bench.py's host-side helpers that need no GPU: the CPU share of the cgroup, the carried phantom-ranks table, the algorithmic bytes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_cpu_share_is_the_quota_or_the_mask():
    import bench
    cores, affinity, quota = bench.cpu_share()
    assert 1 <= cores <= affinity == len(os.sched_getaffinity(0))
    if quota is not None:
        assert cores == max(1, min(affinity, int(quota + 0.5)))
    else:
        assert cores == affinity


def test_host_contention_carries_the_committed_runs():
    import bench
    hc = bench.host_contention()
    assert hc is not None and hc["rows"], "profiles/r*_phantom_ranks_*.json are committed"
    base = [r for r in hc["rows"] if r["phantom_ranks"] == 0]
    assert base and all(r["both_maps_k_per_min"] > 0 and r["lazy_k_per_min"] > 0 for r in hc["rows"])
    assert "quota" in hc["note"]
    json.dumps(hc)      # (goes into the bench line)


def test_algorithmic_bytes_of_the_labelling_pass():
    import bench
    n = 256 ** 3
    assert bench.algorithmic_bytes("k_tile_label", n, 2) == 4 * n + 12 * (n // 64) * 2
    assert bench.algorithmic_bytes("k_labels_tiles", n, 2) == bench.algorithmic_bytes("k_tile_label", n, 2)
    assert not bench.algorithmic_bytes("k_face_merge", n, 2)
