"""Every kernel launch of the LAST step in a rocprofv3 kernel trace, longest first: `python tools/rocprof_calls.py DB [top]`."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if m:  # mangled: <len><identifier>[I Li<n>E ... E]
        n = int(m.group(1))
        ident = name[m.end():m.end() + n]
        rest = name[m.end() + n:]
        t = re.match(r"I((?:Li\d+E)+)E", rest)
        return ident + ("<" + ",".join(re.findall(r"Li(\d+)E", t.group(1))) + ">" if t else "")
    return re.sub(r"\(.*$", "", name)[:60]


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute("select s.kernel_name, d.start, d.end, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, d.stream_id "
                      "from %s d join %s s on d.kernel_id=s.id order by d.start" % (kd, ks)).fetchall()
    marks = [i for i, r in enumerate(rows) if "adamw_kernel" in r[0]]
    step = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = step[0][1]
    print("%d launches, %.2f ms of kernel time" % (len(step), sum(r[2] - r[1] for r in step) / 1e6))
    for r in sorted(step, key=lambda r: r[1] - r[2])[:top]:
        print("%8.1f us  at %6.2f ms  grid %5d x %-4d stream %s  %s" % ((r[2] - r[1]) / 1e3, (r[1] - t0) / 1e6, r[3], r[4], r[5], short(r[0])))


if __name__ == "__main__":
    main()
