#!/bin/bash
# Register / LDS / scratch figures of the step's kernels from the compiler's metadata (no GPU needed): bash tools/exp/isa_meta.sh [csrc-dir]
src=${1:-pdb_eda_amd/csrc}
out=/tmp/isa_meta_$$; mkdir -p $out
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -std=c++17 -Wno-unused-function --save-temps=obj -o $out/lib.so $src/pdbeda_hip.hip 2>/dev/null
python3 - $out/pdbeda_hip-hip-amdgcn-amd-amdhsa-gfx950.s <<'PY'
import sys, re
t = open(sys.argv[1]).read()
for blk in re.findall(r"  - \.agpr_count:.*?\.wavefront_size: +\d+", t, re.S):
    name = re.search(r"\.name: +(\S+)", blk).group(1)
    if not any(k in name for k in ("k_tile_labelILi4", "k_face_mergeILi4ELi384", "k_resolve_tiles", "k_labels_tilesILi4ELb1")): continue
    g = lambda k: re.search(r"\.%s: +(\d+)" % k, blk).group(1)
    print("%-70s vgpr %s sgpr %s spills v%s s%s scratch %s lds %s" % (name[:70], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
PY
rm -rf $out
