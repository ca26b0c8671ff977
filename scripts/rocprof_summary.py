#!/usr/bin/env python3
"""Summarise a rocprofv3 `--kernel-trace --stats` results DB (rocpd sqlite) into a small text table:
per kernel (name + grid) calls / total / average / share.   usage: rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = c.execute("select name, grid_x, grid_y, grid_z, workgroup_x, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name, grid_x, grid_y, grid_z order by sum(duration) desc").fetchall()
    tot = sum(r[6] for r in rows)
    by_name = {}
    for r in rows:
        d = by_name.setdefault(r[0], [0, 0])
        d[0] += r[5]
        d[1] += r[6]
    lines = [f"# rocprofv3 kernel summary of {db}", f"# total GPU kernel time {tot/1e6:.2f} ms over {sum(r[5] for r in rows)} dispatches", "",
             "## by kernel", f"{'kernel':70s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'share':>7s}"]
    for n, (cnt, d) in sorted(by_name.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n[:70]:70s} {cnt:8d} {d/1e6:10.2f} {d/cnt/1e3:10.1f} {d/tot*100:6.1f}%")
    lines += ["", "## by kernel and grid (top 40)", f"{'kernel':58s} {'grid':>16s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s}"]
    for r in rows[:40]:
        g = f"{r[1]}x{r[2]}x{r[3]}/{r[4]}"
        lines.append(f"{r[0][:58]:58s} {g:>16s} {r[5]:7d} {r[6]/1e6:9.2f} {r[7]/1e3:9.1f} {r[8]/1e3:8.1f} {r[9]/1e3:8.1f}")
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
