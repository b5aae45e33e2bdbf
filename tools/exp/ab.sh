#!/bin/bash
# A/B timing of library builds: bash tools/exp/ab.sh main ablx/libX.so ...  (per-kernel HIP-event times of the 256^3 step)
for lib in "$@"; do
  if [ "$lib" = main ]; then python3 tools/profile_step.py 2>&1 | tail -1; else PDBEDA_LIB=$PWD/$lib python3 tools/profile_step.py 2>&1 | tail -1; fi
done
