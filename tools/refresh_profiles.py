"""gpurun_out/<dir> (as produced by tools/collect_profiles.sh) -> profiles/r02_*.

    python tools/refresh_profiles.py r02b

  r02_bench_n1.json             the bench line (python bench.py)
  r02_bench_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `bench.py --steps 5 --warmup 2 --no-cpu-baseline`
  r02_pmc.json                  per kernel: HBM bytes per launch (FETCH_SIZE x2 per the gfx950 correction of
                                MI355X_MICROARCH.md + WRITE_SIZE, separate passes), MFMA busy fraction
                                (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); the dispatches of a PMC pass
                                run serialized, i.e. every kernel alone on the GPU), and whole-step totals
  r02_bench_<variant>.json      the other bench lines of the run (bf16, batch 64, 512x512, batch 1)"""
import csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", sys.argv[1])
dst = os.path.join(ROOT, "profiles")
PFX = (sys.argv[2] if len(sys.argv) > 2 else "r04") + "_"      # round prefix of the files written


def last_json(path):
    return [l for l in open(path) if l.startswith("{")][-1]


open(os.path.join(dst, PFX + "bench_n1.json"), "w").write(last_json(os.path.join(src, "bench_n1.json")))
for f in glob.glob(os.path.join(src, "bench_*.json")):
    name = os.path.basename(f)
    if name != "bench_n1.json":
        open(os.path.join(dst, PFX + name), "w").write(last_json(f))
rows = list(csv.DictReader(open(os.path.join(src, "stats", "r_kernel_stats.csv"))))
with open(os.path.join(dst, PFX + "bench_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        n = r["Name"].replace("(anonymous namespace)::", "")
        w.writerow([n[:110] + ("..." if len(n) > 110 else ""), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])

F = json.load(open(os.path.join(src, "fetch.json")))
W = json.load(open(os.path.join(src, "write.json")))
M = json.load(open(os.path.join(src, "mfma.json")))
STEPS = int(F["adamw_kernel"]["counters"]["FETCH_SIZE"][1])     # training steps in the PMC run = AdamW launches
out, tot_f, tot_w, tot_busy, tot_cyc = {}, 0.0, 0.0, 0.0, 0.0
for k in sorted(set(F) | set(W) | set(M)):
    e = {}
    if k in F and "FETCH_SIZE" in F[k]["counters"]:
        s, n = F[k]["counters"]["FETCH_SIZE"]
        e["fetch_bytes_per_launch"] = s * 2 * 1024 / n
        e["launches_sampled"] = n
        tot_f += s * 2 * 1024
    if k in W and "WRITE_SIZE" in W[k]["counters"]:
        s, n = W[k]["counters"]["WRITE_SIZE"]
        e["write_bytes_per_launch"] = s * 1024 / n
        tot_w += s * 1024
    if "fetch_bytes_per_launch" in e and "write_bytes_per_launch" in e:
        e["hbm_bytes_per_launch"] = e["fetch_bytes_per_launch"] + e["write_bytes_per_launch"]
    if k in M:
        c = M[k]["counters"]
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"][0] > 0:
            cyc = c["GRBM_GUI_ACTIVE"][0] / 8.0
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 1])[0]
            e["mfma_busy_frac"] = round(busy / (cyc * 1024), 4)
            e["gpu_cycles_per_launch"] = round(cyc / c["GRBM_GUI_ACTIVE"][1])
            tot_busy += busy
            tot_cyc += cyc
    if e:
        out[k] = e
json.dump(dict(note="rocprofv3 --pmc passes over `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` (FETCH_SIZE, WRITE_SIZE and "
                    "{SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES, GRBM_GUI_ACTIVE, SQ_INSTS_VALU_MFMA_MOPS_F32} in three separate "
                    "runs); FETCH_SIZE / WRITE_SIZE are in KB, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                    "requests as 64 B); averages over all launches of a kernel name (all resolutions); mfma_busy_frac = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) with the kernel ALONE on the GPU (PMC "
                    "passes serialize dispatches); whole_step: sums over every kernel of the run / %d steps" % STEPS,
               whole_step=dict(hbm_fetch_MB=round(tot_f / STEPS / 1e6, 1), hbm_write_MB=round(tot_w / STEPS / 1e6, 1),
                               hbm_total_MB=round((tot_f + tot_w) / STEPS / 1e6, 1), algorithmic_MB=round(2085.54 * 8, 1),
                               mfma_busy_ms_per_simd_at_2p4GHz=round(tot_busy / STEPS / 1024 / 2.4e6, 3),
                               serialized_gpu_ms_at_2p4GHz=round(tot_cyc / STEPS / 2.4e6, 2)),
               kernels=out), open(os.path.join(dst, PFX + "pmc.json"), "w"), indent=1)
print(json.dumps(json.load(open(os.path.join(dst, PFX + "pmc.json")))["whole_step"]))
for k in ("wgrad3_kernel<2, 2, 0>", "dw_fwd_kernel<float, true>", "dw_bwd_kernel<float, 0, true, true, 2>", "dw_stats1_kernel<float, true>",
          "na_fwd_kernel<1, float>"):
    print(k, {a: (round(b) if b > 10 else b) for a, b in out.get(k, {}).items()})
