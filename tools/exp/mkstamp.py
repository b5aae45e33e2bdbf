# Diagnostic build (never shipped): a COPY of csrc with s_memrealtime stamps in k_tile_label -- after every barrier (thread 0) and,
# per wave, at the end of the stream (A1), of the numbering (A2), of the unions (B) and of the folds (C2).
#   python tools/exp/mkstamp.py  -> ablx/libSTAMP.so ;  on the GPU box: PDBEDA_LIB=ablx/libSTAMP.so python tools/exp/stamps.py
#   python tools/exp/mkstamp.py [NAME [csrc-dir]]  (another source tree, e.g. last round's, for a side-by-side)
import os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "STAMP"
src, dst = (sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "pdb_eda_amd", "csrc")), "/tmp/csrc_stamp_" + name
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)
inc = os.path.join(root, "include")
t = open(os.path.join(dst, "pdbeda_tile.h")).read()
a = t.index("__global__ void __launch_bounds__(512, 8) k_tile_label")
b = t.index("// Cross-tile pairs of one mask word.")
body = t[a:b]
ST = "((unsigned long long *)stamps_p)[(size_t)bid * 64 + %s] = __builtin_amdgcn_s_memrealtime();"
def once(old, new):
    global body
    assert body.count(old) == 1, old
    body = body.replace(old, new)
once("    const uint32_t bid = ((uint32_t)st * (uint32_t)td.rtiles + (uint32_t)rt) * (uint32_t)td.ctiles + (uint32_t)ct;\n",
     "    const uint32_t bid = ((uint32_t)st * (uint32_t)td.rtiles + (uint32_t)rt) * (uint32_t)td.ctiles + (uint32_t)ct;\n    unsigned long long *stamps_p = job.stamps;\n    if (tid == 0) " + ST % "0" + "\n")
for k in (1, 2, 3, 4):
    once("    __syncthreads();   // ---- barrier %d" % k, "    __syncthreads();   if (tid == 0) " + (ST % str(k)) + "  // ---- barrier %d" % k)
once("    __syncthreads();   if (tid == 0) " + (ST % "1"), "    if (lane == 0) " + (ST % "(8 + wv)") + "\n    __syncthreads();   if (tid == 0) " + (ST % "1"))
bmark = "    // ---- B: touching pairs -> unions." if "    // ---- B: touching pairs -> unions." in body else "    // ---- B: 26-connected components inside the tile"
once(bmark, "    if (lane == 0) " + (ST % "(16 + wv)") + "\n" + bmark)
once("    __syncthreads();   if (tid == 0) " + (ST % "2"), "    if (lane == 0) " + (ST % "(24 + wv)") + "\n    __syncthreads();   if (tid == 0) " + (ST % "2"))
once("    __syncthreads();   if (tid == 0) " + (ST % "4"), "    if (lane == 0) " + (ST % "(32 + wv)") + "\n    __syncthreads();   if (tid == 0) " + (ST % "4"))
once("    if (tid == 0) lj.tile_runs[bid] = n_runs;\n}", "    if (tid == 0) lj.tile_runs[bid] = n_runs;\n    if (tid == 0) " + (ST % "5") + "\n}")
t = t[:a] + body + t[b:]
open(os.path.join(dst, "pdbeda_tile.h"), "w").write(t)
k = open(os.path.join(dst, "pdbeda_kernels.h")).read()
k = k.replace("    uint32_t run_cap, comp_cap, blob_cap;", "    uint32_t run_cap, comp_cap, blob_cap;\n    unsigned long long *stamps;", 1)
open(os.path.join(dst, "pdbeda_kernels.h"), "w").write(k)
h = open(os.path.join(dst, "pdbeda_hip.hip")).read()
h = h.replace("    job.inbox = n_tiles ?", "    job.stamps = n_tiles ? cv.take<unsigned long long>(64 * n_tiles) : nullptr;\n    job.inbox = n_tiles ?", 1)
h += '''
extern "C" int pdbeda_bloblist_stamps(pdbeda_bloblist *bl, unsigned long long *out, int64_t n_tiles) {
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, ctx_sync(ctx));
    HIP_TRY(ctx, hipMemcpy(out, bl->job.stamps, 512 * n_tiles, hipMemcpyDeviceToHost));
    return 0;
}
'''
for f, txt in (("pdbeda_hip.hip", h), ("pdbeda_device.h", open(os.path.join(dst, "pdbeda_device.h")).read())):
    open(os.path.join(dst, f), "w").write(txt.replace('#include "../../include/pdbeda.h"', '#include "%s/pdbeda.h"' % inc).replace('#include "/root/repo/include/pdbeda.h"', '#include "%s/pdbeda.h"' % inc))
os.makedirs(os.path.join(root, "ablx"), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                       "-o", os.path.join(root, "ablx", "lib%s.so" % name), os.path.join(dst, "pdbeda_hip.hip")])
print("built ablx/lib%s.so" % name)
