"""Aggregate a rocprofv3 kernel trace (…_kernel_trace.csv) by (kernel, grid size): launches, average / total duration.
   python tools/trace_by_grid.py <kernel_trace.csv> [substring ...]  -> one line per (kernel, grid), largest total first."""
import csv, sys, collections
rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "")
    d = 0
    for i, c in enumerate(n):            # cut the argument list: the first "(" outside <...>
        d += (c == "<") - (c == ">")
        if c == "(" and d == 0 and i > 0:
            n = n[:i]
            break
    if len(sys.argv) > 2 and not any(s in n for s in sys.argv[2:]):
        continue
    g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    rows[(n, g // max(wg, 1), wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in rows.values())
print("%-64s %8s %5s %6s %9s %9s %6s" % ("kernel", "blocks", "wg", "n", "avg_us", "min_us", "share"))
for (n, g, wg), v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    print("%-64s %8d %5d %6d %9.1f %9.1f %6.3f" % (n[:64], g, wg, len(v), sum(v) / len(v), min(v), sum(v) / tot))
