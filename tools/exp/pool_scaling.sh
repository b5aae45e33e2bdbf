#!/bin/bash
# entries/min of the multiple-structure leg by worker count (both maps uploaded, and the lazy loader): bash tools/exp/pool_scaling.sh
for w in 1 2 3 4 5; do
  python3 bench.py --steps 5 --warmup 2 --windows 0 --streams 1 --no-cpu-baseline --no-analysis --no-sigma3 --sweep-entries 0 --workers $w 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])['multiple_structures']
print('workers $w: %.0f entries/min (%.2f of PCIe), lazy %.0f, one worker %.2f ms/entry' % (d['entries_per_min'], d['roofline']['both_maps']['frac'], d['lazy_diff_map']['entries_per_min'], d['one_worker_ms_per_entry']))"
done
