"""Host-side checks of the profiling helpers under tools/ (no GPU): the per-launch timeline and the per-kernel summary read a rocprofv3
`rocpd` sqlite database; a synthetic one with the columns they query keeps them honest."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, rows):
    c = sqlite3.connect(path)
    c.execute("create table kernels (name text, start integer, end integer, vgpr_count integer, accum_vgpr_count integer, lds_size integer, "
              "grid_x integer, workgroup_x integer)")
    c.executemany("insert into kernels values (?, ?, ?, 64, 0, 1024, ?, 256)", rows)
    c.commit()
    c.close()


def test_rocpd_timeline_lists_the_last_step_launch_by_launch(tmp_path):
    t, rows = 1000, []
    for step in range(3):                                    # three "sampling steps": 2 GEMMs + 1 attention + the DDIM step that closes them
        for name, dur in (("void gemm_conv_bf16_buf_kernel<256, 320, 3, 0, 4, 2, true, 5>(ddpo_gemm_desc, int)", 46_000 + 1000 * step),
                          ("void attn_fwd_bf16_dma_kernel<40, 48, 64, 1, true>(float const*, int)", 990_000),
                          ("void gemm_conv_bf16_buf_kernel<256, 320, 3, 0, 4, 2, true, 5>(ddpo_gemm_desc, int)", 91_000),
                          ("ddim_step_kernel(float const*, float const*)", 8_500)):
            rows.append((name, t, t + dur, 131072))
            t += dur + 200                                   # 200 ns idle between launches
    db, out = str(tmp_path / "t.db"), str(tmp_path / "tl.txt")
    _db(db, rows)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_timeline.py"), db, out], check=True)
    lines = open(out).read().splitlines()
    assert lines[0].startswith("# last sampling step of the trace: 4 launches") and "idle 0.6 us" in lines[0]
    body = [l.split(None, 6) for l in lines if not l.startswith("#")]
    assert [b[6] for b in body] == ["gemm_conv_bf16_buf_kernel<256, 320, 3, 0, 4, 2, true, 5>", "attn_fwd_bf16_dma_kernel<40, 48, 64, 1, true>",
                                    "gemm_conv_bf16_buf_kernel<256, 320, 3, 0, 4, 2, true, 5>", "ddim_step_kernel"]
    assert [float(b[2]) for b in body] == [48.0, 990.0, 91.0, 8.5]          # the LAST step's durations (us), in launch order
    assert body[0][4:6] == ["131072", "256"]                                # grid / workgroup columns when the database has them
    # fewer than two DDIM steps: nothing to delimit a step with
    db1 = str(tmp_path / "one.db")
    _db(db1, rows[:4])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_timeline.py"), db1, out], capture_output=True, text=True)
    assert r.returncode != 0 and "fewer than two" in (r.stderr + r.stdout)


def test_rocpd_summary_ranks_kernels_by_total_time(tmp_path):
    rows = [("k_small()", 0, 1_000, 64), ("k_big()", 2_000, 1_002_000, 64), ("k_small()", 1_100_000, 1_101_000, 64)]
    db, out = str(tmp_path / "s.db"), str(tmp_path / "s.md")
    _db(db, rows)
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), db, out, "title"], check=True)
    text = open(out).read()
    assert text.startswith("# title") and "2 distinct kernels" in text
    table = [l for l in text.splitlines() if l.startswith("| `")]
    assert table[0].startswith("| `k_big()` | 1 | 1.00 |") and table[1].startswith("| `k_small()` | 2 | 0.00 |")
