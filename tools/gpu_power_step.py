"""Shader clock and board power while the training step runs (bench.py in a child process; this process only polls rocm-smi).
    python tools/gpu_power_step.py [extra bench.py flags]"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
child = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "600", "--warmup", "5", "--no-cpu-baseline", "--no-other-configs"] + sys.argv[1:],
                         stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
samples = []
t0 = time.time()
while child.poll() is None:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True).stdout
    s = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    p = re.search(r"Power \(W\): ([\d.]+)", out)
    u = re.search(r"GPU use \(%\): (\d+)", out)
    samples.append((time.time() - t0, int(s.group(1)) if s else -1, float(p.group(1)) if p else -1.0, int(u.group(1)) if u else -1))
    time.sleep(0.05)
line = [l for l in child.stdout.read().split("\n") if l.startswith("{")]
import json
if line:
    d = json.loads(line[-1]); print("bench: %.3f ms/step, %.1f img/s" % (d["ms_per_step"], d["value"]))
busy = [s for s in samples if s[3] >= 90]
print("%d samples, %d with GPU use >= 90 %%" % (len(samples), len(busy)))
for name, sel in (("all busy samples", busy),):
    if not sel: continue
    sc = sorted(x[1] for x in sel); pw = sorted(x[2] for x in sel)
    print("%s: sclk MHz min %d / median %d / max %d; power W min %.0f / median %.0f / max %.0f" % (name, sc[0], sc[len(sc) // 2], sc[-1], pw[0], pw[len(pw) // 2], pw[-1]))
print("trace (t s, sclk MHz, W, use %):", [(round(a, 1), b, int(c), e) for a, b, c, e in samples[::max(1, len(samples) // 40)]])
