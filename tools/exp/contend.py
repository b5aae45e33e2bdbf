# Does an entry's analysis slow down when uploads run beside it?  One process: thread A analyses RESIDENT entries over and over
# (no upload of its own), U threads upload maps from files on their own contexts (inside the library, GIL released).
#   python tools/exp/contend.py [uploaders ...]      e.g. 0 1 2 4
import os, sys, time, threading, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from pdb_eda_amd import _native, ccp4, synthetic, multipleStructures, densityAnalysis
_native.pin_to_device(0)
densityAnalysis.setGlobals(synthetic.synthetic_params())
tmp = tempfile.mkdtemp(prefix="pdbeda_contend_")
try:
    loaders = [synthetic.write_entry_files(tmp, "e%d" % k, 200, 100, k, as_paths=True) for k in range(8)]
    ctx = _native.Context(0)
    os.environ["PDBEDA_EAGER_DIFF_MAP"] = "1"
    entry = multipleStructures.Entry("e0", loaders[0])
    loaded = multipleStructures.loadEntry(entry, ctx)

    def analyse(n):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            r = multipleStructures.analyzeEntry(entry, ctx, {}, True, loaded=loaded)
            ts.append(time.perf_counter() - t0)
            assert r
        return 1e3 * float(np.median(ts)), 1e3 * float(np.mean(ts))

    analyse(10)
    for n_up in [int(a) for a in (sys.argv[1:] or ["0", "1", "2", "4"])]:
        stop = threading.Event()
        moved = [0] * max(n_up, 1)
        call_s = [0.0] * max(n_up, 1)

        def uploader(k):
            # (as little Python as possible beside the analysis thread: the header is parsed once, the loop is one ctypes call -- GIL
            #  released -- and a free per map)
            c = _native.Context(0)
            i = k
            head = ccp4.read(loaders[0].density_path, "u", ctx=c, lazy=True)
            geom, off = head.header.geometry(), 1024 + head.header.symmetryBytes
            while not stop.is_set():
                l = loaders[i % len(loaders)]
                for path in (l.density_path, l.diff_path):
                    t1 = time.perf_counter()
                    m = _native.DeviceMap.from_file(c, path, off, False, geom)
                    call_s[k] += time.perf_counter() - t1
                    moved[k] += 4 * 200 ** 3
                    m.free()
                i += n_up
        threads = [threading.Thread(target=uploader, args=(k,), daemon=True) for k in range(n_up)]
        for t in threads:
            t.start()
        time.sleep(0.2)
        m0, t0 = sum(moved), time.perf_counter()
        med, mean = analyse(150)
        dt = time.perf_counter() - t0
        gbs = (sum(moved) - m0) / dt / 1e9
        stop.set()
        for t in threads:
            t.join()
        per_call = (sum(moved) / max(sum(call_s), 1e-9) / 1e9) if n_up else 0.0
        print("uploaders %d: analysis of a resident entry median %.2f ms mean %.2f ms; uploads beside it %.1f GB/s (inside one upload call: %.1f GB/s)" % (n_up, med, mean, gbs, per_call), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
