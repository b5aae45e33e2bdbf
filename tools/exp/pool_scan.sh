# Round 6: entries/min of the multiple-structure leg by worker processes x reader threads per process (PDBEDA_FILE_READERS).
# NEVER more than 5 workers: the box's process guard allows 6 processes on the GPU, and bench.py itself is one of them.
# (4 x 3: 44.6 k / 85.4 k = 0.83 / 0.80 of the link; 4 x 2: 0.79 / 0.82; 5 x 2: 0.18 / 0.62; 5 x 3: 0.46 / 0.48 -- the cgroup's 16-CPU quota is spent at four workers)
for cfg in "4 3" "4 2" "5 2" "5 3" "3 3"; do set -- $cfg
PDBEDA_FILE_READERS=$2 python3 bench.py --steps 5 --no-cpu-baseline --no-analysis --no-sigma3 --no-beyond-cache --streams 1 --sweep-entries 0 --windows 0 --entry-files 48 --workers $1 2>/dev/null | python3 -c "
import sys,json
p=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=p['multiple_structures']
print('workers $1 readers $2', round(m['entries_per_min']), round(m['roofline']['both_maps']['frac'],3), round(m['lazy_diff_map']['entries_per_min']), round(m['roofline']['lazy_diff_map']['frac'],3), round(m['one_worker_ms_per_entry'],3), round(m['load_GBs_single'],1))"
done
