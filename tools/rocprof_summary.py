#!/usr/bin/env python3
"""Turns a rocprofv3 (rocpd sqlite) results.db into a small text summary for profiles/ (kernel stats + PMC sums)."""
import sqlite3
import sys


def short_name(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    head = name[:name.index(">(") + 1] if ">(" in name else name.split("(")[0]
    return head[:90]


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    lines = ["# rocprofv3 summary of %s" % path.split("/")[-1], "", "## kernel stats (top_kernels view; durations in us)",
             "| kernel | calls | total_us | avg_us | pct |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append("| %s | %d | %.1f | %.2f | %.2f |" % (short_name(name), calls, total, avg, pct))
    try:
        rows = list(cur.execute(
            "select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name"))
    except Exception:
        rows = []
    if rows:
        lines += ["", "## PMC counters (per kernel: dispatches, sum, mean per dispatch)",
                  "| kernel | counter | n | sum | mean |", "|---|---|---|---|---|"]
        for k, c, n, s, a in rows:
            lines.append("| %s | %s | %d | %.6g | %.6g |" % (short_name(k)[:60], c, n, s, a))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
