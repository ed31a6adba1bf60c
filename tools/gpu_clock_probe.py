"""Which shader clock does the GPU hold under the step's kernels?  (the cycle-based floors assume 2.4 GHz)
    python tools/gpu_clock_probe.py
Loops one workload for ~3 s at a time while a thread polls `rocm-smi --showclocks --showpower` (the sampled sclk / power are instantaneous
values of the SMU, ~10 samples per workload): idle, a level-0 3x3 conv, the level-0 depthwise forward, the level-0 depthwise backward, the
whole training step."""
import os, subprocess, sys, threading, time, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from lm_net_amd import hip

samples = []
stop = [False]


def poll():
    while not stop[0]:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
        except Exception as e:   # noqa: BLE001
            out = ""
        s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        p = re.search(r"Power \(W\): ([\d.]+)", out)
        samples.append((int(s.group(1)) if s else -1, float(p.group(1)) if p else -1.0))
        time.sleep(0.05)


def run(name, fn, seconds=3.0):
    samples.clear()
    stop[0] = False
    th = threading.Thread(target=poll)
    torch.cuda.synchronize()
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        n += 20
    dt = time.time() - t0
    stop[0] = True
    th.join()
    sc = [s for s, _ in samples if s > 0][1:]
    pw = [p for _, p in samples if p > 0][1:]
    print("%-28s %7.1f us/iter | sclk MHz min %s median %s max %s | power W median %s  (%d samples)" % (
        name, dt / n * 1e6, min(sc) if sc else "-", sorted(sc)[len(sc) // 2] if sc else "-", max(sc) if sc else "-",
        sorted(pw)[len(pw) // 2] if pw else "-", len(sc)), flush=True)


dev = "cuda"
B, H = 8, 352
run("idle (host sleep)", lambda: time.sleep(0.002))
x = torch.randn(B, H, H, 24, device=dev); w = torch.randn(12, 24, 3, 3, device=dev); wp = hip.conv_pack(w, 3, [24]); y = torch.empty(B, H, H, 12, device=dev)
run("3x3 conv 24->12 at 352^2", lambda: hip.conv_fwd([x], wp, y, B=B, Hin=H, Win=H, Hout=H, Wout=H, Cout=12, ksize=3))
E = 24
z = hip.rp4(torch.randn(B, H, H, E, device=dev)); pre = hip.rp4(torch.empty(B, H, H, E, device=dev)); gsum = torch.zeros(B, E, device=dev)
keff, beff = torch.randn(E, 25, device=dev) * 0.1, torch.zeros(E, device=dev)
run("depthwise forward, level 0", lambda: hip.dw_fwd(z, pre, gsum, keff, beff))
a = torch.randn(4096, 4096, device=dev)
run("torch fp32 matmul 4096^3", lambda: torch.mm(a, a))
big = torch.empty(1 << 28, device=dev); big2 = torch.empty(1 << 28, device=dev)
run("copy 1 GiB", lambda: big2.copy_(big))
