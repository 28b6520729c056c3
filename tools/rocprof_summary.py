#!/usr/bin/env python3
"""Summarise a rocprofv3 results database (rocpd SQLite, the default output of `rocprofv3 --kernel-trace --stats`)
into the per-kernel table committed under profiles/: calls, total / average / min / max duration, share of GPU time.

usage: python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r1_kernel_stats.md
"""
import sqlite3
import sys


def main(path, top=40):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), "
                       f"max(end - start) from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 kernel-trace summary: {path}\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows[:top]:
        short = name if len(name) <= 90 else name[:87] + "..."
        print(f"| `{short}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
