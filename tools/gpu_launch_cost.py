"""Host-side cost of issuing one training step vs its wall time (batch given on the command line): is the small-batch step
bound by the launch rate of lmn_plan_run, by the GPU's dependent-launch gaps, or by kernel time?
   python tools/gpu_launch_cost.py [batch] [size]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 352
dev = torch.device("cuda:0")
for mode in ("plans", "graphs", "host"):
    run = bench.Run(dev, 1, 0, "f32", B, S, plans=mode == "plans", graphs=mode == "graphs")
    for _ in range(6):
        run.step()
    torch.cuda.synchronize()
    issue, wall = [], []
    for _ in range(20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3); wall.append((t2 - t0) * 1e3)
    issue.sort(); wall.sort()
    # back-to-back (the host runs ahead of the GPU)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run.step()
    torch.cuda.synchronize()
    bb = (time.perf_counter() - t0) / 20 * 1e3
    print("%-6s batch %d: host issue %.2f ms, issue+drain %.2f ms, back-to-back %.2f ms/step" % (mode, B, issue[10], wall[10], bb), flush=True)
    del run
