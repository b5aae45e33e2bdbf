# Diagnostic build + run: s_memrealtime stamps at the /*@<letter><n>*/ marker comments of the kernel sources (an instrumented
# COPY of csrc, never the product).  Threads 0 and 256 of every workgroup stamp (wave 0 and wave 4): slots n and 8 + n.
#   python tools/exp/markstamps.py build L        -> ablx/libMST_L.so   (markers /*@L0*/ ... of k_labels_tiles; R: k_resolve_tiles; F: k_face_merge)
#   PDBEDA_LIB=$PWD/ablx/libMST_L.so python tools/exp/markstamps.py run L [nsd]   (GPU box)
import os, re, sys, subprocess, shutil
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
letter = sys.argv[2]
JOBREF = {"L": "lj", "R": "job", "F": "job", "T": "lj"}[letter]
BLOCK = {"L": "bid", "R": "blockIdx.x", "F": "blockIdx.x", "T": "bid"}[letter]
if sys.argv[1] == "build":
    src, dst = os.path.join(root, "pdb_eda_amd", "csrc"), "/tmp/csrc_mst_" + letter
    shutil.rmtree(dst, ignore_errors=True); shutil.copytree(src, dst)
    inc = os.path.join(root, "include")
    t = open(os.path.join(dst, "pdbeda_tile.h")).read()
    def stamp(m):
        n = int(m.group(1))
        return ("{ if ((threadIdx.x & 255) == 0 && threadIdx.x < 512) %s.stamps[(size_t)%s * 16 + (threadIdx.x >> 8) * 8 + %d] = __builtin_amdgcn_s_memrealtime(); }"
                % (JOBREF, BLOCK, n))
    t, cnt = re.subn(r"/\*@%s(\d)\*/" % letter, stamp, t)
    assert cnt > 0, "no markers"
    open(os.path.join(dst, "pdbeda_tile.h"), "w").write(t)
    k = open(os.path.join(dst, "pdbeda_kernels.h")).read()
    k = k.replace("    int32_t unit_form;\n};", "    int32_t unit_form;\n    unsigned long long *stamps;\n};", 1)
    assert "stamps" in k
    open(os.path.join(dst, "pdbeda_kernels.h"), "w").write(k)
    h = open(os.path.join(dst, "pdbeda_hip.hip")).read()
    h = h.replace("    job.inbox = n_tiles ?", "    job.stamps = n_tiles ? cv.take<unsigned long long>(4096 * 16) : nullptr;\n    job.inbox = n_tiles ?", 1)
    assert "job.stamps" in h
    anchor = "    int pair_slots = PAIR_SLOTS;"
    assert h.count(anchor) == 1
    h = h.replace(anchor, "    (void)hipMemsetAsync(job.stamps, 0, 4096 * 16 * 8, ctx->stream);\n" + anchor)
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
    out = os.path.join(root, "ablx", "libMST_%s.so" % letter)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-ffp-contract=off", "-std=c++17", "-Wno-unused-function",
                           "-o", out, os.path.join(dst, "pdbeda_hip.hip")])
    print("built", out, "with", cnt, "stamps")
else:
    sys.path.insert(0, root)
    import ctypes as C
    import numpy as np
    from pdb_eda_amd import _native, ccp4, synthetic
    n = 256
    nsd = float(sys.argv[3]) if len(sys.argv) > 3 else 1.5
    spec = synthetic.MapSpec(ncrs=(n, n, n), spacing=0.4)
    grid = synthetic.smooth_noise((n, n, n), seed=7, sigma_voxels=1.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    ctx = _native.Context(0)
    dmap = _native.DeviceMap(ctx, grid, header.geometry())
    mean, std = dmap.stats()
    cut = mean + nsd * std
    for _ in range(4):
        g, r = dmap.full_blobs_pm(cut, -cut, labels=True)
    ctx.synchronize()
    lib = _native.lib()
    out = np.zeros((4096, 16), dtype=np.uint64)
    lib.pdbeda_bloblist_stamps.restype = C.c_int
    lib.pdbeda_bloblist_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    assert lib.pdbeda_bloblist_stamps(g._h, out.ctypes.data_as(C.c_void_p), out.size) == 0
    t = out.astype(np.int64)[:1024]
    valid = t[:, 0] > 0
    t0 = t[valid, 0].min()
    for half, name in ((0, "wave 0"), (1, "wave 4")):
        marks = sorted(set(int(x) for x in re.findall(r"/\*@%s(\d)\*/" % letter, open(os.path.join(root, "pdb_eda_amd", "csrc", "pdbeda_tile.h")).read())))
        for j in marks:
            col = t[valid, half * 8 + j]
            if not (col > 0).any():
                continue
            d = (col[col > 0] - t0) / 100.0
            print("%s stamp %s%d: median %5.1f  p10 %5.1f  p90 %5.1f  max %5.1f us  (%d workgroups)" % (name, letter, j, np.median(d), np.percentile(d, 10), np.percentile(d, 90), d.max(), len(d)))
