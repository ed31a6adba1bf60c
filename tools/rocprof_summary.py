"""Summarise a rocprofv3 results database (kernel trace) into per-kernel totals: `python tools/rocprof_summary.py DB [steps]`."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
        "group by s.kernel_name order by 3 desc" % (kd, ks)).fetchall()
    tot = sum(r[2] for r in rows)
    print("name,calls_per_step,total_us_per_step,avg_us,min_us,max_us,percent")
    for name, n, t, mn, mx in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*$", "", name)[:90]
        print("%s,%.1f,%.1f,%.2f,%.2f,%.2f,%.2f" % (name.replace(",", ";"), n / steps, t / steps / 1e3, t / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
    print("TOTAL,,%.1f,,,," % (tot / steps / 1e3))


if __name__ == "__main__":
    main()
