# SQ / TA counters of a probe case: bash tools/pmcprobe.sh conv1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-conv1}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_WAVES" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" ; do
  i=$((i+1))
  rm -rf $R/gpurun_out/pp$i
  timeout 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pp$i -o r -- python3 $R/tools/gpu_pmc_probe.py $W > /dev/null 2>&1
  f=$R/gpurun_out/pp$i/r_counter_collection.csv
  [ -f $f ] && python3 $R/tools/pmc_reduce.py $f > $R/gpurun_out/pp$i.json
  rm -rf $R/gpurun_out/pp$i
done
python3 - <<PY
import json
for i in (1,2,3,4):
    try: d=json.load(open("$R/gpurun_out/pp%d.json"%i))
    except Exception as e: print("pass",i,"failed",e); continue
    for k,v in d.items():
        if "conv" in k or "dw_" in k or "wgrad" in k or "na_" in k:
            for g,cc in v["by_grid"].items():
                print(k[:40], "grid", g, {c:"%.3g"%(x[0]/x[1]) for c,x in cc.items()})
PY
