#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small markdown table for profiles/."""
import sqlite3
import sys


def main(db, out, title):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
                     "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    span = c.execute("select (max(end)-min(start))/1e6 from kernels").fetchone()[0]
    with open(out, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats` (rocpd db summarised by tools/rocpd_summary.py)\n\n")
        f.write(f"total kernel time {tot:.1f} ms over a {span:.1f} ms span ({len(rows)} distinct kernels)\n\n")
        f.write("| kernel | calls | total ms | % | avg us | min us | max us | vgpr | agpr | lds B |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for r in rows[:40]:
            f.write(f"| `{r[0][:100]}` | {r[1]} | {r[2]:.2f} | {100 * r[2] / tot:.1f} | {r[3]:.1f} | {r[4]:.1f} | {r[5]:.1f} | {r[6]} | {r[7]} | {r[8]} |\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "kernel stats")
