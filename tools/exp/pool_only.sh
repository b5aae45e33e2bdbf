#!/bin/bash
# The multiple-structure leg of bench.py by itself, N times (both loaders, the lone worker, one load alone): bash tools/exp/pool_only.sh [N]
for i in $(seq 1 ${1:-2}); do
python3 bench.py --steps 5 --no-cpu-baseline --no-analysis --no-sigma3 --no-beyond-cache --streams 1 --sweep-entries 0 --windows 0 2>/dev/null | python3 -c "
import sys,json
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=p['multiple_structures']
print('both', round(m['entries_per_min']), round(m['roofline']['both_maps']['frac'],3), 'lazy', round(m['lazy_diff_map']['entries_per_min']), round(m['roofline']['lazy_diff_map']['frac'],3), 'one worker ms', round(m['one_worker_ms_per_entry'],3), 'lone load GB/s', round(m['load_GBs_single'],1), 'link', round(m['roofline']['peak'],1))"
done
