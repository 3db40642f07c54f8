#!/usr/bin/env python
"""Per-dispatch timeline of the LAST `window_ms` of a rocprofv3 rocpd database: kernel, grid, duration, gap to the
previous kernel's end.  Usage: timeline_rocpd.py db [n_last_dispatches | -n_replays (dispatches after the marker sleep kernel / n_replays)]"""
import sqlite3
import sys


def main(db_path, n_last):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    gx = 'grid_size_x' if 'grid_size_x' in cols else None
    wx = 'workgroup_size_x' if 'workgroup_size_x' in cols else None
    sel = "s.kernel_name, d.start, d.end" + (", d.%s, d.grid_size_y, d.grid_size_z, d.%s" % (gx, wx) if gx and wx else "")
    rows = list(cur.execute("select %s from %s d join %s s on d.kernel_id = s.id order by d.start" % (sel, kd, ks)))
    mark = [i for i, r in enumerate(rows) if 'sleep' in r[0].lower() or 'spin' in r[0].lower()]
    if mark and n_last <= 0:
        rows = rows[mark[-1] + 1:]
        per = len(rows) // max(-n_last, 1)
        rows = rows[-per:]
    else:
        rows = rows[-n_last:]
    prev_end = None
    tot_k = tot_gap = 0.0
    print("%-4s %-70s %-16s %9s %8s" % ("#", "kernel", "grid(wg)", "dur_us", "gap_us"))
    for i, r in enumerate(rows):
        name, st, en = r[0], r[1], r[2]
        grid = ""
        if len(r) > 3 and r[6]:
            grid = "%dx%dx%d" % (r[3] // max(r[6], 1), r[4], r[5])
        gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
        dur = (en - st) / 1e3
        tot_k += dur
        tot_gap += max(gap, 0.0)
        short = name.replace("_ZN12_GLOBAL__N_1", "").replace("_ZN3d3f", "")[:70]
        print("%-4d %-70s %-16s %9.2f %8.2f" % (i, short, grid, dur, gap))
        prev_end = max(prev_end, en) if prev_end is not None else en
    print("# %d dispatches: kernel time %.1f us, gaps %.1f us, span %.1f us" % (
        len(rows), tot_k, tot_gap, (rows[-1][2] - rows[0][1]) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 800)
