"""gpurun_out/<dir> (bench_n1.json, stats/, fetch/, write/ as produced by the commands in profiles/README.md) -> profiles/."""
import collections, csv, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1])
dst = os.path.join(ROOT, "profiles")
line = [l for l in open(os.path.join(src, "bench_n1.json")) if l.startswith("{")][-1]
open(os.path.join(dst, "r01_bench_n1.json"), "w").write(line)
rows = list(csv.DictReader(open(os.path.join(src, "stats", "r_kernel_stats.csv"))))
with open(os.path.join(dst, "r01_bench_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        w.writerow([n[:110] + ("..." if len(n) > 110 else ""), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        d[k].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return d


f, w = load(os.path.join(src, "fetch", "r_counter_collection.csv")), load(os.path.join(src, "write", "r_counter_collection.csv"))
out = {}
for k in ["dw_fwd_strip_kernel", "dw_bwd_strip_kernel", "na_fwd_kernel<1>", "na_fwd_kernel<2>", "na_fwd_kernel<4>", "na_fwd_kernel<8>"]:
    if k not in f:
        continue
    tf, tw, n = sum(v for _, v in f[k]) * 2 * 1024, sum(v for _, v in w[k]) * 1024, len(f[k])
    per = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for (g, v) in f[k]:
        per[g][0] += v * 2 * 1024; per[g][2] += 1
    for (g, v) in w[k]:
        per[g][1] += v * 1024
    out[k] = dict(launches_sampled=n, fetch_bytes_per_launch=tf / n, write_bytes_per_launch=tw / n, hbm_bytes_per_launch=(tf + tw) / n,
                  by_grid={str(g): dict(fetch_MB=round(a / c / 1e6, 1), write_MB=round(b / c / 1e6, 1)) for g, (a, b, c) in sorted(per.items(), reverse=True)})
json.dump(dict(note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 --warmup 1 "
                    "--no-cpu-baseline`; counters are in KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests "
                    "as 64 B); averages over all launches of the kernel (all four resolutions), by_grid = per launch shape",
               kernels=out), open(os.path.join(dst, "r01_pmc_hbm_traffic.json"), "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 1) for k, v in out.items()}))
