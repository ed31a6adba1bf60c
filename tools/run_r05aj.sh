#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r05aj; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/p1 -o r -- python3 $R/tools/gpu_dw_probe.py 0 6 > $O/p1.log 2>&1; echo "rc $?"; tail -3 $O/p1.log
timeout 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES --kernel-trace --output-format csv -d $O/p2 -o r -- python3 $R/tools/gpu_dw_probe.py 0 6 > $O/p2.log 2>&1; echo "rc $?"
find $O -name "*counter_collection.csv" | head
for f in $(find $O -name "*counter_collection.csv"); do
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    if "dw_" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
for k, d in sorted(acc.items()):
    print(k[:50], {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
done
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
