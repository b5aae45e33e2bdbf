#!/bin/bash
# A/B of library builds on the analysis entry (same box, alternating): bash tools/exp/ab_entry.sh main ablx/libBASE.so main ablx/libBASE.so
for lib in "$@"; do
  if [ "$lib" = main ]; then python3 tools/prof_entry_parts.py 10 128 400 2>&1 | grep "analysis_entry" | sed "s/^/main: /"; else PDBEDA_LIB=$PWD/$lib python3 tools/prof_entry_parts.py 10 128 400 2>&1 | grep "analysis_entry" | sed "s|^|$lib: |"; fi
done
