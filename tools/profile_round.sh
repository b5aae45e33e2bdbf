#!/bin/bash
# Refresh the measurements kept under profiles/ (run on the GPU box through gpurun, from the repo root):
#   bash tools/profile_round.sh r03
# 1. rocprofv3 --kernel-trace --stats     -> gpurun_out/prof_<tag>/trace/... (bench.py under the profiler; the informational
#    legs -- CPU baseline, multi-stream, densityAnalysis -- are switched off so the trace holds only the single-stream steps)
# 2. rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE in SEPARATE passes over tools/profile_step.py
# 3. bench.py alone (after the counters are condensed, so that its roofline.traffic is this round's) -> gpurun_out/prof_<tag>/bench.json
# tools/make_profiles.py then condenses 1-3 into profiles/<tag>_*.  (Counter passes never carry --stats/traces.)
set -e -o pipefail
tag=${1:-r04}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o bench -- python3 "$root/bench.py" --steps 30 --warmup 5 --windows 0 --entries 0 --sweep-entries 0 --no-cpu-baseline --no-analysis --no-sigma3 --no-beyond-cache --streams 1 > "$out/bench_under_rocprof.json" 2> "$out/trace.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -o p -- python3 "$root/tools/profile_step.py" > "$out/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -o p -- python3 "$root/tools/profile_step.py" > "$out/pmc_write.log" 2>&1
cd "$root"
python3 tools/make_profiles.py "$tag" "$out" > /dev/null      # the counters first: bench.py prints roofline.traffic from them
python3 bench.py --steps 50 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"
python3 tools/make_profiles.py "$tag" "$out"
echo "profiles refreshed from $out"
