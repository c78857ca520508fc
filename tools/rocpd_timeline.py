#!/usr/bin/env python
"""Per-launch timeline of the LAST sampling step in a rocprofv3 (rocpd sqlite) kernel trace: every launch between the last two
`ddim_step_kernel`s (one U-Net forward + the sampler glue of a graph replay) with its duration and the idle gap in front of it.
The per-kernel averages of rocpd_summary.py hide which launches of one shape are slow (the tall 256x320 tile spans 45 ... 220 us)."""
import re
import sqlite3
import sys


def short(name):
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)(<[^>]*>)?", name)
    return (m.group(1).split("::")[-1] + (m.group(2) or "")) if m else name[:60]


def main(db, out):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    extra = [k for k in ("grid_size_x", "grid_x", "workgroup_size_x", "workgroup_x") if k in cols]
    rows = c.execute(f"select name, start, end{''.join(', ' + k for k in extra)} from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "ddim_step_kernel" in r[0]]
    if len(marks) < 2:
        raise SystemExit("fewer than two ddim_step_kernel launches in the trace")
    seg = rows[marks[-2] + 1: marks[-1] + 1]
    busy = sum(r[2] - r[1] for r in seg) / 1e3
    span = (seg[-1][2] - seg[0][1]) / 1e3
    with open(out, "w") as f:
        f.write(f"# last sampling step of the trace: {len(seg)} launches, busy {busy:.1f} us, span {span:.1f} us, idle {span - busy:.1f} us\n")
        f.write(f"# columns: index, start offset us, duration us, gap before us, {', '.join(extra)}, kernel\n")
        prev = seg[0][1]
        for i, r in enumerate(seg):
            f.write(f"{i:4d} {(r[1] - seg[0][1]) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:8.1f} {(r[1] - prev) / 1e3:6.1f} "
                    f"{' '.join(str(v) for v in r[3:])} {short(r[0])}\n")
            prev = r[2]


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
