"""Upper bound of what faster matrix-core arithmetic could buy the fp32 step: the same step with fp32 storage and ONE bf16 MFMA
per K chunk ('bf16-mma') against fp32 MFMA.  (A split-operand emulation of fp32 products needs 3 .. 6 of those per chunk.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
for cd in ("fp32", "bf16-mma", "fp32", "bf16-mma"):
    run = bench.Run(dev, 1, 0, "f32", 8, 352)
    run.net.compute_dtype = cd
    for _ in range(6):
        run.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        run.step()
    torch.cuda.synchronize()
    print("%-9s %.3f ms/step" % (cd, (time.perf_counter() - t0) / 20 * 1e3), flush=True)
    del run
