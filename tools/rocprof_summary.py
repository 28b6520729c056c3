#!/usr/bin/env python3
"""Summarise rocprofv3 results databases (rocpd SQLite, the default output of `rocprofv3 --kernel-trace [--stats|--pmc ..]`)
into the small markdown tables committed under profiles/.

  python tools/rocprof_summary.py trace  <results.db>            per-kernel calls / total / avg / min / max / share
  python tools/rocprof_summary.py mfma   <results.db>            MFMA utilisation per kernel from SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE
  python tools/rocprof_summary.py grids  <results.db> <substr>   per (kernel, grid) calls / total / avg of the kernels whose name contains <substr>
  python tools/rocprof_summary.py pmc    <results.db> [scale]    per-kernel sum and per-launch mean of each collected counter
                                                                  (scale multiplies the values, e.g. 2 for FETCH_SIZE on gfx950)
"""
import sqlite3
import sys


def short(name, n=88):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def trace(path, top=40):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                       "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary ({path.split('/')[-1]})\n")
    print(f"total kernel time {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, n, tot, avg, mn, mx in rows[:top]:
        print(f"| `{short(name)}` | {n} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100 * tot / total:.2f} |")


def grids(path, flt):
    """The launches of one kernel family split by launch geometry (= by problem shape)."""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    g = [c for c in ("grid_x", "grid_y", "grid_z", "grid_size_x", "grid_size_y", "grid_size_z") if c in cols]
    if not g:
        print("no grid columns in the kernels view:", cols)
        return
    gs = ", ".join(g)
    rows = cur.execute(f"select name, {gs}, count(*), sum(end - start), avg(end - start), min(end - start) from kernels "
                       f"where name like ? group by name, {gs} order by {len(g) + 3} desc", (f"%{flt}%",)).fetchall()
    total = sum(r[len(g) + 2] for r in rows) or 1
    print(f"# launches of kernels matching '{flt}' by grid ({path.split('/')[-1]}): {total / 1e6:.3f} ms\n")
    print("| kernel | grid (work-items) | calls | total ms | avg us | min us |")
    print("|---|---|---:|---:|---:|---:|")
    for r in rows:
        print(f"| `{short(r[0], 48)}` | {' x '.join(str(x) for x in r[1:1 + len(g)])} | {r[len(g) + 1]} | {r[len(g) + 2] / 1e6:.3f} | "
              f"{r[len(g) + 3] / 1e3:.1f} | {r[len(g) + 4] / 1e3:.1f} |")


def pmc(path, scale=1.0, top=25):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                       "group by kernel_name, counter_name order by 4 desc").fetchall()
    print(f"# rocprofv3 --pmc summary ({path.split('/')[-1]}), values x {scale:g}\n")
    print("| kernel | counter | launches | sum | mean per launch |")
    print("|---|---|---:|---:|---:|")
    for name, ctr, n, tot in rows[:top]:
        tot = (tot or 0) * scale
        print(f"| `{short(name)}` | {ctr} | {n} | {tot:.4g} | {tot / max(n, 1):.4g} |")


def mfma(path, top=16):
    """MFMA utilisation per kernel = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the busy counter
    sums cycles over all SIMDs, GRBM_GUI_ACTIVE sums the active cycles of the 8 XCDs."""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                       "group by kernel_name, counter_name").fetchall()
    per = {}
    for name, ctr, n, tot in rows:
        per.setdefault(name, {})[ctr] = (n, tot or 0.0)
    print(f"# MFMA utilisation per kernel ({path.split('/')[-1]}): SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)\n")
    print("| kernel | launches | MFMA busy cycles (sum) | GRBM_GUI_ACTIVE (sum) | MFMA utilisation |")
    print("|---|---:|---:|---:|---:|")
    items = sorted(per.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1])
    for name, c in items[:top]:
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0.0))[1]
        n, act = c.get("GRBM_GUI_ACTIVE", (0, 0.0))
        if not act:
            continue
        print(f"| `{short(name, 60)}` | {n} | {busy:.4g} | {act:.4g} | {100.0 * busy / (act / 8.0 * 1024.0):.1f} % |")


def pmc_dispatches(path, flt=""):
    """One row per dispatch with every collected counter as a column (for single-kernel experiments)."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)").fetchall()]
    did = "dispatch_id" if "dispatch_id" in cols else cols[0]
    rows = cur.execute(f"select {did}, kernel_name, counter_name, sum(value) from counters_collection "
                       f"group by {did}, kernel_name, counter_name order by {did}").fetchall()
    table, names = {}, []
    for d, k, c, v in rows:
        if flt and flt not in k:
            continue
        table.setdefault((d, k), {})[c] = v
        if c not in names:
            names.append(c)
    print("| dispatch | kernel | " + " | ".join(names) + " |")
    print("|---|---|" + "---:|" * len(names))
    for (d, k), vals in table.items():
        print(f"| {d} | `{short(k, 40)}` | " + " | ".join(f"{vals.get(c, 0):.4g}" for c in names) + " |")


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    if mode == "trace":
        trace(path)
    elif mode == "mfma":
        mfma(path)
    elif mode == "grids":
        grids(path, sys.argv[3])
    elif mode == "pmcd":
        pmc_dispatches(path, sys.argv[3] if len(sys.argv) > 3 else "")
    else:
        pmc(path, float(sys.argv[3]) if len(sys.argv) > 3 else 1.0)
