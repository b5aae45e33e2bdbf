#!/bin/bash
# the whole-map step by grid edge (fused +-1.5 sigma, labels on): bash tools/exp/size_sweep.sh
for n in 64 100 128 160 200 256 320; do
  python3 bench.py --size $n --steps 50 --warmup 5 --windows 0 --streams 1 --no-cpu-baseline --no-analysis --no-sigma3 --sweep-entries 0 --entries 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('edge $n: %.4f ms/step = %.1f Gvoxel/s  kernels %s' % (d['ms_per_step'], d['value'] / 1e3, d['roofline'].get('kernels_us') or d.get('kernels_us')))"
done
