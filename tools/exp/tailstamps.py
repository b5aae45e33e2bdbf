# Diagnostic build + run: s_memrealtime stamps inside the tail kernels (k_resolve_tiles, k_paint_tiles, k_emit): where a
# latency-bound kernel spends its microseconds.  python tools/exp/tailstamps.py build   (here) ;  ... run   (GPU box, PDBEDA_LIB=abl/libTSTAMP.so)
import os, sys, subprocess, shutil
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "build":
    src, dst = os.path.join(root, "pdb_eda_amd", "csrc"), "/tmp/csrc_tstamp"
    shutil.rmtree(dst, ignore_errors=True)
    shutil.copytree(src, dst)
    inc = os.path.join(root, "include")
    ST = "if (threadIdx.x == 0) job.stamps[((size_t)%d * 2048 + blockIdx.x) * 8 + %d] = __builtin_amdgcn_s_memrealtime();"
    t = open(os.path.join(dst, "pdbeda_tile.h")).read()
    def once(old, new):
        global t
        assert t.count(old) == 1, old
        t = t.replace(old, new)
    # resolve (kernel 0)
    once("    static_assert(CCAP == 256, \"one thread per component id of a tile\");\n    const int tid = threadIdx.x;", "    static_assert(CCAP == 256, \"one thread per component id of a tile\");\n    const int tid = threadIdx.x;\n    " + ST % (0, 0))
    once("    int root = -1;\n    bool member = false;   // non-root component with voxels", "    if (n_i == 0xdeadbeefu) return;\n    " + ST % (0, 1) + "\n    int root = -1;\n    bool member = false;   // non-root component with voxels")
    once("    __syncthreads();\n    if (__syncthreads_or(member ? 1 : 0) == 0) return;   // nothing to fold in this tile", "    __syncthreads();\n    " + ST % (0, 2) + "\n    if (__syncthreads_or(member ? 1 : 0) == 0) return;   // nothing to fold in this tile")
    once("            fold(root, s_cnt[k], s_f[k], s_i[0][k], s_i[1][k], s_i[2][k], s_key[k]);\n        }\n    }\n}", "            fold(root, s_cnt[k], s_f[k], s_i[0][k], s_i[1][k], s_i[2][k], s_key[k]);\n        }\n    }\n    " + ST % (0, 3) + "\n}")
    # paint (kernel 1)
    once("    __shared__ uint32_t s_cnt[CCAP];\n    const uint32_t i = (uint32_t)blockIdx.x * CCAP + tid;\n    const uint32_t n_in", "    __shared__ uint32_t s_cnt[CCAP];\n    " + ST % (1, 0) + "\n    const uint32_t i = (uint32_t)blockIdx.x * CCAP + tid;\n    const uint32_t n_in")
    once("    const bool root = par == (int32_t)i && n_i != 0u;\n    {   // the tile's roots as four ballots", "    const bool root = par == (int32_t)i && n_i != 0u;\n    if (n_i == 0xdeadbeefu) return;\n    " + ST % (1, 1) + "\n    {   // the tile's roots as four ballots")
    once("    if (root) paint(key);\n}", "    " + ST % (1, 2) + "\n    if (root) paint(key);\n    " + ST % (1, 3) + "\n}")
    open(os.path.join(dst, "pdbeda_tile.h"), "w").write(t)
    k = open(os.path.join(dst, "pdbeda_kernels.h")).read()
    k = k.replace("    uint64_t *root_mask;", "    unsigned long long *stamps;\n    uint64_t *root_mask;", 1)
    # emit (kernel 2)
    ke = "    const uint32_t total = rank_table_lds(job, s_pre, s_wave);"
    assert k.count(ke) == 1
    k = k.replace(ke, "    " + ST % (2, 0) + "\n" + ke + "\n    " + ST % (2, 1))
    ke2 = "    if (blockIdx.x == 0 && threadIdx.x == 0) {   // the table's totals, for the host and for k_labels_tiles"
    assert k.count(ke2) == 1
    k = k.replace(ke2, "    " + ST % (2, 2) + "\n" + ke2)
    open(os.path.join(dst, "pdbeda_kernels.h"), "w").write(k)
    h = open(os.path.join(dst, "pdbeda_hip.hip")).read()
    h = h.replace("    job.inbox = n_tiles ?", "    job.stamps = n_tiles ? cv.take<unsigned long long>(3 * 2048 * 8) : nullptr;\n    job.inbox = n_tiles ?", 1)
    h += '''
extern "C" int pdbeda_bloblist_stamps(pdbeda_bloblist *bl, unsigned long long *out, int64_t n) {
    pdbeda_ctx *ctx = bl->ctx;
    HIP_TRY(ctx, ctx_sync(ctx));
    HIP_TRY(ctx, hipMemcpy(out, bl->job.stamps, 8 * n, hipMemcpyDeviceToHost));
    return 0;
}
'''
    for f, txt in (("pdbeda_hip.hip", h), ("pdbeda_device.h", open(os.path.join(dst, "pdbeda_device.h")).read())):
        open(os.path.join(dst, f), "w").write(txt.replace('#include "../../include/pdbeda.h"', '#include "%s/pdbeda.h"' % inc))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                           "-o", os.path.join(root, "abl", "libTSTAMP.so"), os.path.join(dst, "pdbeda_hip.hip")])
    print("built abl/libTSTAMP.so")
else:
    sys.path.insert(0, root)
    import ctypes as C
    import numpy as np
    from pdb_eda_amd import _native, ccp4, synthetic
    n = 256
    spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
    grid = synthetic.smooth_noise((n, n, n), seed=7, sigma_voxels=1.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    ctx = _native.Context(0)
    dmap = _native.DeviceMap(ctx, grid, header.geometry())
    mean, std = dmap.stats()
    cut = mean + 1.5 * std
    for _ in range(3):
        g, r = dmap.full_blobs_pm(cut, -cut, labels=True)
    ctx.synchronize()
    lib = _native.lib()
    out = np.zeros((3, 2048, 8), dtype=np.uint64)
    lib.pdbeda_bloblist_stamps.restype = C.c_int
    lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), out.size) == 0
    t = out.astype(np.int64)
    names = ["k_resolve_tiles", "k_paint_tiles", "k_emit"]
    labels = [["entry", "first loads back", "roots found (barrier)", "posted"], ["entry", "first loads back", "absorbed", "painted"], ["entry", "rank table built", "roots emitted", ""]]
    for k in range(3):
        used = t[k, :, 0] > 0
        tt = t[k][used]
        t0 = tt[:, 0].min()
        print(names[k], "workgroups", used.sum(), "start spread %.1f us" % ((tt[:, 0].max() - t0) / 100.0))
        for j in range(1, 4):
            ok = tt[:, j] > 0
            if ok.any():
                print("   %-24s at median %5.1f  p90 %5.1f  max %5.1f us after the first entry" % (labels[k][j], np.median(tt[ok, j] - t0) / 100.0, np.percentile(tt[ok, j] - t0, 90) / 100.0, (tt[ok, j] - t0).max() / 100.0))
