cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python3 -m pytest $R/tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -2
rm -rf $R/gpurun_out/wgp; timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/wgp -- python3 $R/tools/gpu_pmc_probe.py wgrad > /dev/null 2>&1
python3 - $(find $R/gpurun_out/wgp -name "*.db" | head -1) <<'PY'
import sqlite3,sys,re
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
for r in db.execute("select s.kernel_name,d.end-d.start,d.grid_size_x/d.workgroup_size_x,d.grid_size_y from %s d join %s s on d.kernel_id=s.id order by d.start"%(kd,ks)):
    if "wgrad_lds" in r[0] or "wgrad_1x1" in r[0]:
        m=re.search(r"(wgrad\w+?)ILi(\d+)ELi(\d+)ELi(\d+)",r[0]) or re.search(r"(wgrad\w+?)ILi(\d+)ELi(\d+)",r[0])
        print(m.groups(), "%.1f us"%(r[1]/1e3), "grid", r[2], r[3])
PY
