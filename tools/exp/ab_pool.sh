#!/bin/bash
# A/B of library builds on the multiple-structure leg (same box, alternating): bash tools/exp/ab_pool.sh main ablx/libBASE.so ...
for lib in "$@"; do
  if [ "$lib" = main ]; then unset PDBEDA_LIB; else export PDBEDA_LIB=$PWD/$lib; fi
  python3 bench.py --steps 5 --warmup 2 --windows 0 --streams 1 --no-cpu-baseline --no-analysis --no-sigma3 --sweep-entries 0 --workers 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])['multiple_structures']
print('$lib: %.0f entries/min, lazy %.0f, one worker %.2f ms' % (d['entries_per_min'], d['lazy_diff_map']['entries_per_min'], d['one_worker_ms_per_entry']))"
done
