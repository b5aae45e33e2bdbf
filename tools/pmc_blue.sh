#!/bin/bash
root=$(pwd); out=$root/gpurun_out/pmc_blue; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/a -o p -- python3 $root/tools/time_blue.py 256 6000 > $out/a.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -o p -- python3 $root/tools/time_blue.py 256 6000 > $out/t.log 2>&1
cd $root
python3 tools/pmc_summary.py $out/a 2>&1 | grep -E "Counter|k_union_edges|k_resolve|k_face_merge"
