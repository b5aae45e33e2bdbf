# Condense one tools/profile_round.sh run into the small files kept under profiles/.
#   python tools/make_profiles.py r01 gpurun_out/prof_r01
import glob, json, os, shutil, sys
import pandas as pd

tag, src = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def last_json_line(path):
    with open(path) as fh:
        lines = [l for l in fh.read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


bench = None
if os.path.exists(os.path.join(src, "bench.json")):   # (profile_round.sh calls this once before bench.py has run: the counters first)
    bench = last_json_line(os.path.join(src, "bench.json"))
    with open(os.path.join(dst, tag + "_bench.json"), "w") as fh:
        json.dump(bench, fh)
        fh.write("\n")
under = last_json_line(os.path.join(src, "bench_under_rocprof.json"))
with open(os.path.join(dst, tag + "_bench_under_rocprof.json"), "w") as fh:
    json.dump(under, fh)
    fh.write("\n")
stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)
assert stats, "no kernel_stats.csv under " + src
shutil.copy(stats[0], os.path.join(dst, tag + "_bench_kernel_stats.csv"))


def pmc_means(sub, counter):
    f = glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True)
    assert f, "no counter_collection.csv under " + sub
    d = pd.read_csv(f[0])
    d = d[d["Counter_Name"] == counter]
    d["k"] = d["Kernel_Name"].str.extract(r"(k_\w+)")
    return d.groupby("k")["Counter_Value"].mean().to_dict()


fetch, write = pmc_means("pmc_fetch", "FETCH_SIZE"), pmc_means("pmc_write", "WRITE_SIZE")
kern = {}
for k in sorted(set(fetch) | set(write)):
    f_kb, w_kb = float(fetch.get(k, 0.0)), float(write.get(k, 0.0))
    kern[k] = {"FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB": w_kb, "hbm_bytes_per_launch_corrected": int(round((2.0 * f_kb + w_kb) * 1024))}
note = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/profile_step.py (256^3, +-1.5 sigma, fused, labels on); "
        "per-launch means. gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request -> x2; calibrated on "
        "k_reduce_partials (a float4 stream of exactly 65536 KB reports ~32792 KB). WRITE_SIZE is exact for streaming stores.")
sys.path.insert(0, root)
import bench as bench_module  # noqa: E402  (csrc_sha16: ties these counters to the kernel sources they were collected on)
avg = pd.read_csv(stats[0])
avg["k"] = avg["Name"].str.extract(r"(k_\w+)")
rocprof_avg = {k: round(float(v) / 1e3, 3) for k, v in avg.dropna(subset=["k"]).groupby("k")["AverageNs"].mean().items()}
with open(os.path.join(dst, tag + "_pmc_traffic.json"), "w") as fh:
    json.dump({"note": note, "csrc_sha16": bench_module.csrc_sha16(), "kernels": kern, "rocprofv3_avg_us": rocprof_avg}, fh, indent=1)
    fh.write("\n")
if bench is not None:
    print(json.dumps({"ms_per_step": bench["ms_per_step"], "value": bench["value"], "roofline": bench["roofline"], "kernels_us": bench["kernels_us"]}, indent=1))
