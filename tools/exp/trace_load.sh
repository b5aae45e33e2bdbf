#!/bin/bash
# the copies and kernels of ONE file upload on the device's clock (rocprofv3 memory-copy + kernel trace of tools/exp/single_load.py): bash tools/exp/trace_load.sh
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/trace_load; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$out" -o t -- python3 "$root/tools/exp/single_load.py" > "$out/log.txt" 2>&1
cd "$root"
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob("gpurun_out/trace_load/**/t_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy %s" % r.get("Direction", "?"), int(r.get("Bytes", r.get("Size", 0)) or 0)))
for f in glob.glob("gpurun_out/trace_load/**/t_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], 0))
rows.sort()
# the last load = everything after the last but one k_np_final
finals = [i for i, r in enumerate(rows) if r[2].startswith("k_np_final")]
lo = finals[-3] + 1 if len(finals) >= 3 else 0
t0 = rows[lo][0]
for s, e, name, size in rows[lo:finals[-1] + 1]:
    print("%9.1f us  +%7.1f us  %-28s %s" % ((s - t0) / 1e3, (e - s) / 1e3, name, ("%.2f MB  %.1f GB/s" % (size / 1e6, size / max(e - s, 1))) if size else ""))
PY
