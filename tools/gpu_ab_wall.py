"""A/B of the training-step wall time under environment switches, interleaved in ONE process is impossible (the switches are read
at engine construction), so: N alternating child runs per setting, median reported.
    python tools/gpu_ab_wall.py "LMN_ZPATH=0" "LMN_ZPATH=1" [--runs 3]"""
import os, subprocess, sys, json, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[sys.argv.index("--runs") + 1]) if "--runs" in sys.argv else 3
_skip = {sys.argv.index("--runs") + 1} if "--runs" in sys.argv else set()
sets = [a for i, a in enumerate(sys.argv) if i >= 1 and not a.startswith("--") and i not in _skip]
res = {s: [] for s in sets}
for r in range(runs):
    for s in sets:
        env = dict(os.environ)
        for kv in s.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1); env[k] = v
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-other-configs", "--steps", "30"],
                             env=env, capture_output=True, text=True)
        try:
            res[s].append(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"])
        except Exception:
            res[s].append(float("nan"))
for s in sets:
    print("%-40s median %.3f ms  runs %s" % (s, statistics.median(res[s]), res[s]))
