#!/bin/bash
# instruction counters of k_tile_label per cumulative phase (stop-after builds under build/abl)
set -o pipefail
root=$(pwd); out=$root/gpurun_out/pmc_ph; rm -rf $out; mkdir -p $out
export TMPDIR=/tmp; cd /tmp
for k in S11 S1 S12 S2 main; do
  lib=$root/build/abl/lib$k.so; [ $k = main ] && lib=$root/pdb_eda_amd/libpdbeda_hip.so
  PDBEDA_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $out/$k -o p -- python3 $root/tools/profile_step.py > $out/$k.log 2>&1 || echo "fail $k"
done
cd $root
python3 tools/pmc_summary.py $out/S11 $out/S1 $out/S12 $out/S2 $out/main 2>&1 | grep -E "pmc_ph|Counter|k_tile_label|k_face_merge|k_union|k_resolve|k_labels"
