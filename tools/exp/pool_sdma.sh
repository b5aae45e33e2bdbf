#!/bin/bash
# the multiple-structure leg with the runtime's copies on SDMA engines (default) and on blit kernels (HSA_ENABLE_SDMA=0)
for sdma in 1 0; do
 for w in 2 4; do
  HSA_ENABLE_SDMA=$sdma python3 bench.py --steps 5 --warmup 2 --windows 0 --streams 1 --no-cpu-baseline --no-analysis --no-sigma3 --sweep-entries 0 --workers $w 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])['multiple_structures']
print('HSA_ENABLE_SDMA=$sdma workers $w: %.0f entries/min, lazy %.0f, one worker %.2f ms/entry, h2d peak %.1f GB/s' % (d['entries_per_min'], d['lazy_diff_map']['entries_per_min'], d['one_worker_ms_per_entry'], d['roofline']['peak']))"
 done
done
