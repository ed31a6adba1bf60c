# SQ counters of the neighborhood-attention kernels alone (tools/gpu_na_probe.py): bash tools/pmc_na.sh   (every pass under a timeout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES" ; do
  i=$((i+1))
  rm -rf $R/gpurun_out/ppn$i
  timeout 150 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/ppn$i -o r -- python3 $R/tools/gpu_na_probe.py > /dev/null 2>&1
  f=$R/gpurun_out/ppn$i/r_counter_collection.csv
  [ -f $f ] && python3 $R/tools/pmc_reduce.py $f > $R/gpurun_out/ppn$i.json
  rm -rf $R/gpurun_out/ppn$i
done
python3 - <<PY
import json
for i in (1,2):
    try: d=json.load(open("$R/gpurun_out/ppn%d.json"%i))
    except Exception as e: print("pass",i,"failed",e); continue
    for k,v in d.items():
        if "na_" in k:
            for g,cc in v["by_grid"].items():
                print(k[:36], "grid", g, {c:"%.3g"%(x[0]/x[1]) for c,x in cc.items()})
PY
