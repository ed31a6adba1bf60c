"""Per-stream timeline of the LAST training step in a rocprofv3 kernel trace: `python tools/rocprof_timeline.py DB [n_steps]`.

Splits the trace at the AdamW launches (one per step), then for the last full step prints, per HIP stream (queue):
kernel count, busy time, and -- for the busiest stream -- the largest idle gaps with the kernels on either side, and the
kernels on that stream sorted by total time."""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"\(.*$", "", name)[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    qcol = "stream_id" if "stream_id" in cols else "queue_id"
    rows = db.execute("select s.kernel_name, d.start, d.end, d.%s, (d.grid_size_x/d.workgroup_size_x)*(d.grid_size_y/d.workgroup_size_y)*(d.grid_size_z/d.workgroup_size_z) "
                      "from %s d join %s s on d.kernel_id=s.id order by d.start" % (qcol, kd, ks)).fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    if len(marks) < 2:
        raise SystemExit("need >= 2 steps in the trace")
    lo, hi = marks[-2] + 1, marks[-1] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print("step: %d kernels, %.2f ms wall" % (len(step), (t1 - t0) / 1e6))
    # occupancy of the GPU over the step: sweep over kernel start / end events
    ev = sorted([(r[1], 1, i) for i, r in enumerate(step)] + [(r[2], -1, i) for i, r in enumerate(step)])
    active, last, by_n, solo = set(), t0, defaultdict(int), defaultdict(int)
    for t, d, i in ev:
        if t > last:
            by_n[len(active)] += t - last
            if len(active) == 1:
                k = step[next(iter(active))]
                solo[(short(k[0]), k[4] < 256)] += t - last
        last = t
        active.add(i) if d > 0 else active.discard(i)
    print("kernels in flight: " + ", ".join("%d: %.2f ms" % (n, v / 1e6) for n, v in sorted(by_n.items())))
    small = sum(v for (k, sm), v in solo.items() if sm)
    print("alone on the GPU with < 256 workgroups: %.2f ms; top:" % (small / 1e6))
    for (k, sm), v in sorted(solo.items(), key=lambda kv: -kv[1])[:12]:
        print("  %-60s %s %7.1f us alone" % (k, "small" if sm else "     ", v / 1e3))
    per = defaultdict(list)
    for r in step:
        per[r[3]].append(r)
    main_q = max(per, key=lambda q: sum(r[2] - r[1] for r in per[q]))
    for q, rs in sorted(per.items(), key=lambda kv: -sum(r[2] - r[1] for r in kv[1])):
        print("stream %s: %4d kernels, busy %.2f ms, first %.2f ms, last %.2f ms" %
              (q, len(rs), sum(r[2] - r[1] for r in rs) / 1e6, (rs[0][1] - t0) / 1e6, (max(r[2] for r in rs) - t0) / 1e6))
    rs = per[main_q]
    gaps = []
    for a, b in zip(rs[:-1], rs[1:]):
        gaps.append((b[1] - a[2], a, b))
    tot_gap = sum(g[0] for g in gaps if g[0] > 0)
    print("busiest stream: idle between kernels %.2f ms; gaps > 20 us: %d (%.2f ms)" %
          (tot_gap / 1e6, sum(1 for g in gaps if g[0] > 20000), sum(g[0] for g in gaps if g[0] > 20000) / 1e6))
    for g, a, b in sorted(gaps, key=lambda g: -g[0])[:15]:
        print("  gap %7.1f us at %.2f ms: %s -> %s" % (g / 1e3, (a[2] - t0) / 1e6, short(a[0]), short(b[0])))
    agg = defaultdict(lambda: [0, 0])
    for r in rs:
        agg[short(r[0])][0] += 1
        agg[short(r[0])][1] += r[2] - r[1]
    print("busiest stream by kernel:")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %-70s %4d %8.1f us" % (k, n, t / 1e3))


if __name__ == "__main__":
    main()
