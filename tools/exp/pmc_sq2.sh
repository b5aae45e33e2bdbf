#!/bin/bash
# SQ unit-busy counters (which issue port a kernel keeps busy): bash tools/exp/pmc_sq2.sh [lib.so] -> gpurun_out/pmc_sq2_<name>
set -e -o pipefail
root=$(pwd)
lib=${1:-main}
name=$(basename "$lib" .so)
out=$root/gpurun_out/pmc_sq2_$name
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
if [ "$lib" != main ]; then export PDBEDA_LIB=$root/$lib; fi
cd /tmp
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES --output-format csv -d "$out" -o p -- python3 "$root/tools/profile_step.py" > "$out/log.txt" 2>&1
cd "$root"
python3 - "$out" <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(f)):
    acc[row["Kernel_Name"].split("(")[0][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    if "k_" not in k: continue
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(d.items())}, "launches", len(next(iter(d.values()))))
PY
