"""Summarise a rocprofv3 run: per-kernel count / total / avg / min / max, as a markdown table.
Input: the rocpd sqlite .db (default output format) or the *_kernel_stats.csv of `--output-format csv`.
usage: python tools/rocprof_stats.py gpurun_out/prof_x/name_results.db|name_kernel_stats.csv [out.md]"""
import csv
import sqlite3
import sys


def stats_csv(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                     float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
    return sorted(rows, key=lambda t: -t[2])


def stats(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, "
         "min(d.end-d.start)/1e3, max(d.end-d.start)/1e3 from %s d join %s s on d.kernel_id=s.id "
         "group by s.kernel_name order by 3 desc" % (kd, ks))
    return list(cur.execute(q))


def main():
    rows = stats_csv(sys.argv[1]) if sys.argv[1].endswith(".csv") else stats(sys.argv[1])
    lines = ["| kernel | calls | total ms | avg us | min us | max us |", "|---|---|---|---|---|---|"]
    for r in rows:
        name = r[0].replace("|", "/")
        if len(name) > 110:
            name = name[:107] + "..."
        lines.append("| `%s` | %d | %.3f | %.1f | %.1f | %.1f |" % (name, r[1], r[2], r[3], r[4], r[5]))
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
