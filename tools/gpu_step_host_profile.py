"""cProfile of the host side of replayed training steps (where do the milliseconds outside lmn_plan_run go?).
   python tools/gpu_step_host_profile.py [batch] [size]"""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 352
run = bench.Run(torch.device("cuda:0"), 1, 0, "f32", B, S, plans=True)
for _ in range(6):
    run.step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    run.step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(30)
