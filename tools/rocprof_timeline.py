"""Timeline of the last `window_ms` of kernel dispatches of a rocprofv3 --kernel-trace run (rocpd sqlite .db):
start offset, duration, queue, kernel -- the critical path of one proof read off the streams it ran on.
usage: python tools/rocprof_timeline.py results.db [window_ms] [out.md]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"dg16::", "", name)
    name = re.sub(r"Fp<(\w+)_params>", r"\1", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def main():
    db = sqlite3.connect(sys.argv[1])
    window = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = list(cur.execute("select d.start, d.end, d.%s, s.kernel_name from %s d join %s s on d.kernel_id=s.id "
                            "order by d.start" % (qcol, kd, ks)))
    t_end = max(r[1] for r in rows)
    t0 = t_end - window * 1e6
    lines = ["| start us | dur us | queue | kernel |", "|---|---|---|---|"]
    for st, en, q, name in rows:
        if en >= t0:
            lines.append("| %.1f | %.1f | %s | `%s` |" % ((st - t0) / 1e3, (en - st) / 1e3, q, short(name)))
    text = "\n".join(lines)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(text + "\n")
    else:
        print(text)


if __name__ == "__main__":
    main()
