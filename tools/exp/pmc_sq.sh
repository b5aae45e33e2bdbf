#!/bin/bash
# SQ counters of the labelling step's kernels (one pass, 8 SQ slots): bash tools/exp/pmc_sq.sh [lib.so] -> gpurun_out/pmc_sq_<name>.csv
set -e -o pipefail
root=$(pwd)
lib=${1:-main}
name=$(basename "$lib" .so)
out=$root/gpurun_out/pmc_sq_$name
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
if [ "$lib" != main ]; then export PDBEDA_LIB=$root/$lib; fi
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE --output-format csv -d "$out" -o p -- python3 "$root/tools/profile_step.py" > "$out/log.txt" 2>&1
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
