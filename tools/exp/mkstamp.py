# Diagnostic build (never shipped): a COPY of csrc with s_memrealtime stamps after the barriers of k_tile_label.
#   python tools/exp/mkstamp.py  -> abl/libSTAMP.so ;  on the GPU box: PDBEDA_LIB=abl/libSTAMP.so python tools/exp/stamps.py
import os, re, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src, dst = os.path.join(root, "pdb_eda_amd", "csrc"), "/tmp/csrc_stamp"
shutil.rmtree(dst, ignore_errors=True)
shutil.copytree(src, dst)
inc = os.path.join(root, "include")
t = open(os.path.join(dst, "pdbeda_tile.h")).read()
a = t.index("__global__ void __launch_bounds__(NT, NT == 512 ? 8 : 1) k_tile_label")
b = t.index("// Generic labelling of the tiles k_tile_label could not hold in LDS")
body = t[a:b]
n = [0]
def stamp(m):
    n[0] += 1
    return m.group(0) + "\n    if (threadIdx.x == 0 && %d < 32) job.stamps[(size_t)blockIdx.x * 32 + %d] = __builtin_amdgcn_s_memrealtime();" % (n[0], n[0])
lines = body.split("\n")
out = []
for i, ln in enumerate(lines):
    out.append(ln)
    if ln.strip().startswith("__syncthreads();"):
        n[0] += 1
        out.append("%sif (threadIdx.x == 0) job.stamps[(size_t)blockIdx.x * 32 + %d] = __builtin_amdgcn_s_memrealtime();  // line %d" % (ln[:len(ln) - len(ln.lstrip())], min(n[0], 31), i))
body = "\n".join(out)
# stamp 0 at kernel entry
body = body.replace("    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;", "    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;\n    if (threadIdx.x == 0) { for (int k = 0; k < 32; ++k) job.stamps[(size_t)blockIdx.x * 32 + k] = 0ull; job.stamps[(size_t)blockIdx.x * 32] = __builtin_amdgcn_s_memrealtime(); }", 1)
body = body.replace("    if (tid == 0) job.tile_runs[blockIdx.x] = n_wordruns;", "    if (tid == 0) job.tile_runs[blockIdx.x] = n_wordruns;\n    if (threadIdx.x == 0) job.stamps[(size_t)blockIdx.x * 32 + 16] = __builtin_amdgcn_s_memrealtime();")
body = body.replace("                if (tid == 0) s_changed = 0;", "                if (tid == 0) { s_changed = 0; job.stamps[(size_t)blockIdx.x * 32 + 20] += 1ull; }")
body = body.replace("            uint32_t wmax = n_edges;   // wave maximum of the pair counts", "            if (threadIdx.x == 0) job.stamps[(size_t)blockIdx.x * 32 + 21] = __builtin_amdgcn_s_memrealtime();\n            uint32_t wmax = n_edges;   // wave maximum of the pair counts")
t = t[:a] + body + t[b:]
open(os.path.join(dst, "pdbeda_tile.h"), "w").write(t)
k = open(os.path.join(dst, "pdbeda_kernels.h")).read()
k = k.replace("    double2 *run_sums;", "    unsigned long long *stamps;\n    double2 *run_sums;", 1)
open(os.path.join(dst, "pdbeda_kernels.h"), "w").write(k)
h = open(os.path.join(dst, "pdbeda_hip.hip")).read()
h = h.replace("    job.run_sums = n_tiles ?", "    job.stamps = n_tiles ? cv.take<unsigned long long>(32 * n_tiles) : nullptr;\n    job.run_sums = n_tiles ?", 1)
h += '''
extern "C" int pdbeda_bloblist_stamps(pdbeda_bloblist *bl, unsigned long long *out, int64_t n_tiles) {
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, ctx_sync(ctx));
    HIP_TRY(ctx, hipMemcpy(out, bl->job.stamps, 256 * n_tiles, hipMemcpyDeviceToHost));
    return 0;
}
'''
h = h.replace('#include "../../include/pdbeda.h"', '#include "%s/pdbeda.h"' % inc)
open(os.path.join(dst, "pdbeda_hip.hip"), "w").write(h)
d = open(os.path.join(dst, "pdbeda_device.h")).read().replace('#include "../../include/pdbeda.h"', '#include "%s/pdbeda.h"' % inc)
open(os.path.join(dst, "pdbeda_device.h"), "w").write(d)
os.makedirs(os.path.join(root, "abl"), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                       "-o", os.path.join(root, "abl", "libSTAMP.so"), os.path.join(dst, "pdbeda_hip.hip")])
print("stamps:", n[0])
for i, ln in enumerate(out):
    if "job.stamps[(size_t)blockIdx.x * 32 +" in ln and "line" in ln:
        print(ln.strip()[-60:], "<-", out[i - 2].strip()[:70])
