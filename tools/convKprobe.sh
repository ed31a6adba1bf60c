# GPU-side durations (rocprofv3 kernel trace) of the launches of tools/gpu_convK_probe.py: bash tools/convKprobe.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ckp; timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/ckp -- python3 $R/tools/gpu_convK_probe.py > /dev/null 2>&1
python3 - $(find $R/gpurun_out/ckp -name "*.db" | head -1) <<'PY'
import sqlite3,sys,collections
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks=[t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
seq=[]
for r in db.execute("select s.kernel_name,d.start,d.end,d.grid_size_x/d.workgroup_size_x,d.grid_size_y from %s d join %s s on d.kernel_id=s.id order by d.start"%(kd,ks)):
    if "conv_tile" in r[0] or "fill" in r[0]:
        key=(r[0][:50],r[3],r[4])
        if seq and seq[-1][0]==key: seq[-1][1].append((r[2]-r[1])/1e3)
        else: seq.append((key,[(r[2]-r[1])/1e3]))
for key,d in seq:
    d=d[5:] if len(d)>10 else d
    print("%-52s grid %4d x %2d  n=%3d  avg %6.1f us  min %6.1f"%(key[0],key[1],key[2],len(d),sum(d)/len(d),min(d)))
PY
rm -rf $R/gpurun_out/ckp
