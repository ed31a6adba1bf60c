"""Host-side cost of replaying the recorded forward / backward plans, per C-ABI entry point (lmn_plan_host_profile).
   python tools/gpu_plan_host_profile.py [batch] [size]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 352
run = bench.Run(torch.device("cuda:0"), 1, 0, "f32", B, S, plans=True)
for _ in range(6):
    run.step()
torch.cuda.synchronize()
ps = [p for p in run.net._plans.values() if p.fwd is not None][0]
for name, plan in (("forward", ps.fwd), ("backward", ps.bwd)):
    for rep in range(2):
        torch.cuda.synchronize()
        prof = plan.host_profile()
    torch.cuda.synchronize()
    tot = sum(v[1] for v in prof.values())
    print("%s: %d ops, %.0f us on the host" % (name, sum(v[0] for v in prof.values()), tot))
    for k, (n, us) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
        print("   %-28s %4d ops %8.0f us  (%.1f us/op)" % (k, n, us, us / n))
