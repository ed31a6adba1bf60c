# per-dispatch durations of the weight-gradient kernels of a probe case: bash tools/wgprobe.sh wg3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-wgrad}
rm -rf $R/gpurun_out/wgp; timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wgp -- python3 $R/tools/gpu_pmc_probe.py $W > /dev/null 2>&1
python3 - $(find $R/gpurun_out/wgp -name "*.db" | head -1) <<'PY'
import sqlite3,sys,re
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
prev=None
for r in db.execute("select s.kernel_name,d.start,d.end,d.grid_size_x/d.workgroup_size_x,d.grid_size_y from %s d join %s s on d.kernel_id=s.id order by d.start"%(kd,ks)):
    if "wgrad" in r[0]:
        print("%-60s %7.1f us  grid %d x %d   gap before %.1f us"%(r[0][:60], (r[2]-r[1])/1e3, r[3], r[4], (r[1]-prev)/1e3 if prev else 0))
    prev=r[2]
PY
rm -rf $R/gpurun_out/wgp
