#!/bin/bash
# CPU sanitizers over the host-side C of the repository (build container only -- GPU boxes refuse ASan):
# the structure walk (_hostwalk.so) and the CPU oracle (libpdbeda_oracle.so) are rebuilt with -fsanitize=address,undefined and
# the CPU test suite runs on them with the sanitizer runtime preloaded; the ordinary builds are put back afterwards.
#   bash tools/sanitize_cpu.sh [pytest args]        -> profiles/<tag>_sanitize_cpu.txt is what the round commits of it
set -e -o pipefail
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root"
inc=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
asan=$(gcc -print-file-name=libasan.so)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
python3 __graft_entry__.py > /dev/null                                    # the ordinary builds exist (and are newer than their sources)
cp pdb_eda_amd/_hostwalk.so /tmp/_hostwalk.plain.so
cp oracle/libpdbeda_oracle.so /tmp/libpdbeda_oracle.plain.so
restore() { cp /tmp/_hostwalk.plain.so pdb_eda_amd/_hostwalk.so; cp /tmp/libpdbeda_oracle.plain.so oracle/libpdbeda_oracle.so; }
trap restore EXIT
gcc $SAN -ffp-contract=off -shared -fPIC -Wall -I"$inc" -o pdb_eda_amd/_hostwalk.so pdb_eda_amd/csrc/hostwalk.c -lm
gcc $SAN -shared -fPIC -ffp-contract=off -fno-fast-math -Wall -Wextra -o oracle/libpdbeda_oracle.so oracle/pdbeda_oracle.c -lm
# (Python itself is not instrumented: leaks of the interpreter are not ours to report; PDBEDA_DEBUG_HOSTWALK=raise: a C walk that
#  fails must fail the test instead of hiding behind its Python fallback)
export LD_PRELOAD="$asan" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export PDBEDA_DEBUG_HOSTWALK=${PDBEDA_DEBUG_HOSTWALK:-1}
python3 -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"
unset LD_PRELOAD
# Round 6: the upload engine of libpdbeda_hip.so (pdb_eda_amd/csrc/pdbeda_upload.h: reader threads, FIFO, slot recycling, deadlines,
# stalled-stream replacement) on the host stand-in of tests/upload_harness.cpp, under ThreadSanitizer and again under ASan + UBSan
for san in "thread" "address,undefined -fno-sanitize-recover=undefined"; do
  g++ -std=c++17 -O1 -g -pthread -fno-omit-frame-pointer -fsanitize=$san -Ipdb_eda_amd/csrc tests/upload_harness.cpp -o /tmp/upload_harness_san
  for scenario in many deadline stall; do
    echo "upload engine, -fsanitize=${san%% *}, $scenario:"
    TSAN_OPTIONS=halt_on_error=1:second_deadlock_stack=1 ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 /tmp/upload_harness_san $scenario
  done
done
