import sys, os, io, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pdb_eda_amd import _native
ctx = _native.Context(0)
bench.analysis_leg(ctx, reps=1)
pr = cProfile.Profile(); pr.enable(); r = bench.analysis_leg(ctx, reps=3); pr.disable()
print(r["ms"])
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
