# SQ counters of the depthwise kernels alone (tools/gpu_dw_probe.py): bash tools/pmc_dw.sh [lib.so|""] [tag] [kernel-name substring]   (every pass under a timeout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export LMNET_HIP_LIB=$1
TAG=${2:-new}
export PMC_K=${3:-dw_bwd_kernel}   # substring of the kernel names to report (dw_fwd_kernel, dw_stats0_kernel, dw_stats1_kernel, dw_)
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES" ; do
  i=$((i+1))
  rm -rf $R/gpurun_out/ppd$i
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/ppd$i -o r -- python3 $R/tools/gpu_dw_probe.py > /dev/null 2>&1
  f=$R/gpurun_out/ppd$i/r_counter_collection.csv
  [ -f $f ] && python3 - "$f" "$TAG" <<'PY'
import csv, sys, collections, os
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if os.environ.get("PMC_K", "dw_bwd_kernel") not in k: continue
    key = (k, r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key, d in sorted(acc.items()):
    print(sys.argv[2], key[0][:60], "grid", key[1], {c: round(v / n[(key, c)]) for c, v in d.items()})
PY
  rm -rf $R/gpurun_out/ppd$i
done
