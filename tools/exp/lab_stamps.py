# Diagnostic build + run: s_memrealtime stamps inside k_labels_tiles (an instrumented copy, never the product).
#   python tools/exp/lab_stamps.py build (here);  PDBEDA_LIB=$PWD/abl/libLST.so python tools/exp/lab_stamps.py run (GPU box)
import os, sys, subprocess, shutil
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "build":
    src, dst = os.path.join(root, "pdb_eda_amd", "csrc"), "/tmp/csrc_lst"
    shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
    inc = os.path.join(root, "include")
    t = open(os.path.join(dst, "pdbeda_tile.h")).read()
    ST = "if (threadIdx.x == %d) job.stamps[(size_t)blockIdx.x * 8 + %d] = __builtin_amdgcn_s_memrealtime();"
    def once(old, new):
        global t
        assert t.count(old) == 1, old
        t = t.replace(old, new)
    once("    const int wvs = __builtin_amdgcn_readfirstlane(wv);\n    const int uc = td.uc, ur = td.ur, us = td.us, row_words = td.row_words;\n    int t = blockIdx.x;",
         "    const int wvs = __builtin_amdgcn_readfirstlane(wv);\n    " + ST % (0, 0) + "\n    const int uc = td.uc, ur = td.ur, us = td.us, row_words = td.row_words;\n    int t = blockIdx.x;")
    once("    __syncthreads();\n    // wave wv writes RPW rows of the tile;", "    " + ST % (0, 1) + "\n    __syncthreads();\n    " + ST % (0, 2) + "\n    // wave wv writes RPW rows of the tile;")
    once("    else { if (inside) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }\n}",
         "    else { if (inside) rows(std::false_type{}, std::true_type{}); else rows(std::false_type{}, std::false_type{}); }\n    " + ST % (0, 3) + "\n    " + ST % (448, 4) + "\n}")
    open(os.path.join(dst, "pdbeda_tile.h"), "w").write(t)
    k = open(os.path.join(dst, "pdbeda_kernels.h")).read()
    k = k.replace("    uint64_t *root_mask;", "    unsigned long long *stamps;\n    uint64_t *root_mask;", 1)
    open(os.path.join(dst, "pdbeda_kernels.h"), "w").write(k)
    h = open(os.path.join(dst, "pdbeda_hip.hip")).read()
    h = h.replace("    job.inbox = n_tiles ?", "    job.stamps = n_tiles ? cv.take<unsigned long long>(2048 * 8) : nullptr;\n    job.inbox = n_tiles ?", 1)
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
                           "-o", os.path.join(root, "abl", "libLST.so"), os.path.join(dst, "pdbeda_hip.hip")])
    print("built abl/libLST.so")
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
    out = np.zeros((2048, 8), dtype=np.uint64)
    lib.pdbeda_bloblist_stamps.restype = C.c_int
    lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), out.size) == 0
    t = out.astype(np.int64)[:1024]
    t0 = t[:, 0].min()
    for j, name in enumerate(["entry", "wave 0: tables in LDS", "after the barrier", "wave 0: rows stored", "wave 7: rows stored"]):
        d = (t[:, j] - t0) / 100.0
        print("%-24s median %5.1f  p90 %5.1f  max %5.1f us" % (name, np.median(d), np.percentile(d, 90), d.max()))
