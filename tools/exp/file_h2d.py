# File (warm page cache) -> HBM: the library's pread ring against one pageable copy out of an mmap of the file, alone and with
# several processes at once (each on the GPU's NUMA node).   python tools/exp/file_h2d.py [processes]
import sys, os, time, tempfile, mmap, multiprocessing
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np


def work(args):
    path, edge, mode, reps, barrier = args
    from pdb_eda_amd import _native, ccp4, synthetic
    _native.pin_to_device(0)
    ctx = _native.Context(0)
    spec = synthetic.MapSpec(ncrs=(edge, edge, edge), spacing=0.5)
    header = ccp4.DensityHeader.fromFileHeader(synthetic.ccp4_header_bytes(spec))
    geom = header.geometry()
    n = edge ** 3
    fd = os.open(path, os.O_RDONLY)
    mm = mmap.mmap(fd, 0, prot=mmap.PROT_READ)
    arr = np.frombuffer(mm, dtype=np.float32, count=n, offset=1024)
    for _ in range(2):
        (_native.DeviceMap.from_file(ctx, path, 1024, False, geom) if mode == "ring" else _native.DeviceMap(ctx, arr, geom)).free()
    barrier.wait()                      # all processes time the same interval
    t0 = time.perf_counter()
    for _ in range(reps):
        if mode == "ring":
            m = _native.DeviceMap.from_file(ctx, path, 1024, False, geom)
        elif mode == "mmap":                      # one mapping, made before the clock started: its pages are mapped already
            m = _native.DeviceMap(ctx, arr, geom)
        else:                                     # a fresh mapping per upload, as an entry's file would be
            mm2 = mmap.mmap(fd, 0, flags=mmap.MAP_PRIVATE | (mmap.MAP_POPULATE if mode == "fresh+populate" else 0), prot=mmap.PROT_READ)
            a2 = np.frombuffer(mm2, dtype=np.float32, count=n, offset=1024)
            m = _native.DeviceMap(ctx, a2, geom)
            del a2
            mm2.close()
        m.free()
    dt = (time.perf_counter() - t0) / reps
    return dt


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    edge = 200
    tmp = tempfile.mkdtemp(prefix="pdbeda_fh2d_")
    paths = []
    for k in range(procs):
        p = os.path.join(tmp, "m%d.ccp4" % k)
        with open(p, "wb") as fh:
            fh.write(b"\0" * 1024)
            fh.write(np.random.default_rng(k).standard_normal(edge ** 3).astype(np.float32).tobytes())
        paths.append(p)
    mb = 4 * edge ** 3 / 1e6
    ctxm = multiprocessing.get_context("spawn")
    for mode in ("ring", "mmap", "fresh", "fresh+populate"):
        for n in sorted({1, procs}):
            with ctxm.Manager() as manager, ctxm.Pool(n) as pool:
                barrier = manager.Barrier(n)
                dts = pool.map(work, [(paths[k], edge, mode, 300, barrier) for k in range(n)])
            print("%s x%d: %.2f ms per %d MB map per process = %.1f GB/s each, %.1f GB/s together" % (mode, n, 1e3 * np.mean(dts), mb, mb / 1e3 / np.mean(dts), n * mb / 1e3 / np.mean(dts)), flush=True)
    import shutil; shutil.rmtree(tmp)
