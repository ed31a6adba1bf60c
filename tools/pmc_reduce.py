"""rocprofv3 --pmc ... --output-format csv (one row per dispatch and counter) -> JSON {kernel: {counter: [sum, launches], by_grid}}."""
import collections, csv, json, sys

agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
grid = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0])))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    c, v = r["Counter_Name"], float(r["Counter_Value"])
    agg[k][c][0] += v; agg[k][c][1] += 1
    g = grid[k][r["Grid_Size"]][c]
    g[0] += v; g[1] += 1
print(json.dumps({k: {"counters": {c: v for c, v in cs.items()},
                      "by_grid": {g: {c: v for c, v in cc.items()} for g, cc in grid[k].items()}} for k, cs in agg.items()}))
