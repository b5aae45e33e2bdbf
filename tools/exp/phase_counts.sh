#!/bin/bash
# Dynamic instruction counts of k_tile_label PER PHASE: variants of the library whose tile kernel returns after barrier 1 / 2 / 3
# (and whose later kernels return at once: they would read what the truncated kernel did not write), one SQ counter pass each.
#   here:  bash tools/exp/phase_counts.sh build       on the GPU box:  bash tools/exp/phase_counts.sh run
set -e -o pipefail
root=$(pwd)
OFF=( '    /*@F0*/=>    return; /*@F0*/' '    /*@R0*/=>    return; /*@R0*/' '    /*@L0*/=>    return; /*@L0*/' '    if ((int)blockIdx.x >= n_tiles) {=>    if (true) return; if ((int)blockIdx.x >= n_tiles) {' )
if [ "$1" = build ]; then
  python3 tools/exp/variant.py PH1 "${OFF[@]}" '    __syncthreads();   // ---- barrier 1 ----=>    __syncthreads(); if (n_planes >= 0) return;  // ---- barrier 1 ----'
  python3 tools/exp/variant.py PH2 "${OFF[@]}" '    __syncthreads();   // ---- barrier 2: all unions done ----=>    __syncthreads(); if (n_planes >= 0) return;  // ---- barrier 2 ----'
  python3 tools/exp/variant.py PH3 "${OFF[@]}" '    __syncthreads();   // ---- barrier 3 ----=>    __syncthreads(); if (n_planes >= 0) return;  // ---- barrier 3 ----'
  python3 tools/exp/variant.py PH4 "${OFF[@]}"
else
  export TMPDIR=/tmp
  for v in PH1 PH2 PH3 PH4; do
    out=$root/gpurun_out/phase_$v; rm -rf "$out"; mkdir -p "$out"
    ( cd /tmp && PDBEDA_LIB=$root/ablx/lib$v.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES --output-format csv -d "$out" -o p -- python3 "$root/tools/exp/run_step_noaccess.py" > "$out/log.txt" 2>&1 ) || true
    python3 - "$out" $v <<'PY'
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print(sys.argv[2], "no counters"); sys.exit(0)
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f[0])):
    if "k_tile_label" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print(sys.argv[2], {c: round(sum(v) / len(v)) for c, v in sorted(acc.items())})
PY
  done
fi
