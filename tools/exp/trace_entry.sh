#!/bin/bash
# every launch (kernels, runtime copies / fills) of ONE analysis entry of bench.py, in order, with the gaps between them
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/trace_entry; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 "$root/tools/prof_analysis.py" 1 > "$out/log.txt" 2>&1
cd "$root"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/trace_entry/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0] for r in rows]
# the last entry = everything after the last pair of map uploads' statistics: find the last two k_np_final pairs
idx = [i for i, n in enumerate(names) if n == "k_np_chunk_sums"]
start = idx[-4] if len(idx) >= 4 else 0
t0 = int(rows[start]["Start_Timestamp"])
count = collections.Counter()
busy = 0
for r, n in list(zip(rows, names))[start:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy += d
    count[n] += 1
    print("%9.1f us  %-28s %7.1f us" % ((int(r["Start_Timestamp"]) - t0) / 1e3, n, d))
span = (int(rows[-1]["End_Timestamp"]) - t0) / 1e3
print("launches %d, device busy %.0f us of %.0f us" % (sum(count.values()), busy, span))
print(sorted(count.items(), key=lambda kv: -kv[1]))
PY
