#!/bin/bash
# per-launch kernel durations of one aggregateCloud (the last of tools/prof_cloud_kernels.py), in launch order
set -e -o pipefail
root=$(pwd); out=$root/gpurun_out/trace_cloud; rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 "$root/tools/prof_cloud_kernels.py" > "$out/log.txt" 2>&1
cd "$root"
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_cloud/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0] for r in rows]
# the last aggregateCloud ends with the last k_pool_component; its first launch is the copy kernel in front of the last but one k_sphere_paint
last = max(i for i, n in enumerate(names) if n in ("k_pool_component", "k_union_finish"))      # (round 6: the union job ends in k_union_finish)
paints = [i for i in range(last) if names[i] in ("k_sphere_paint", "k_atom_engine")]
start = (paints[-1] - 1) if paints else last - 30
t0 = int(rows[max(start, 0)]["Start_Timestamp"])
for r, n in list(zip(rows, names))[max(start, 0):last + 3]:
    print("%9.1f us  %-20s %7.1f us  grid %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
PY
