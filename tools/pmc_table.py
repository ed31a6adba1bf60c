"""Pivot a rocprofv3 --pmc csv (counter_collection.csv) into one row per dispatch: python tools/pmc_table.py FILE [name-filter]"""
import csv, sys, collections
rows = collections.OrderedDict()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = (r["Dispatch_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        rows.setdefault(k, {})[r["Counter_Name"]] = float(r["Counter_Value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = sorted({c for v in rows.values() for c in v})
print("kernel," + ",".join(names))
for (d, k), v in rows.items():
    if flt in k:
        print(k.split("(")[0][:40] + "," + ",".join("%.4g" % v.get(c, float("nan")) for c in names))
