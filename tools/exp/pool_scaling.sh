#!/bin/bash
# entries/min of the multiple-structure leg by worker count (both maps uploaded, and the lazy loader): bash tools/exp/pool_scaling.sh [counts]
# (the GPU boxes of this pool allow six processes on the card: five workers + the bench is the most that runs; 5 measured 34 k / 64 k against 41 k / 75 k at 4)
for w in ${@:-1 2 3 4}; do
  python3 bench.py --steps 5 --warmup 2 --windows 0 --streams 1 --no-cpu-baseline --no-analysis --no-sigma3 --no-beyond-cache --sweep-entries 0 --workers $w 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])['multiple_structures']
print('workers $w: %.0f entries/min (%.2f of PCIe), lazy %.0f (%.2f), one worker %.2f ms/entry' % (d['entries_per_min'], d['roofline']['both_maps']['frac'], d['lazy_diff_map']['entries_per_min'], d['roofline']['lazy_diff_map']['frac'], d['one_worker_ms_per_entry']))"
done
