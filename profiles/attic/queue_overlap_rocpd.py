#!/usr/bin/env python
"""Concurrency picture of the last `window_ms` of a rocprofv3 rocpd kernel trace: dispatches per hardware queue / HIP
stream, busy time of each, how long 1, 2, 3.. of them were busy at once, and the mean duration of the heaviest kernels.
    queue_overlap_rocpd.py db [window_ms=40]"""
import collections
import sqlite3
import sys


def main(db_path, window_ms):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % kd)]
    qcol = 'queue_id' if 'queue_id' in cols else None
    scol = 'stream_id' if 'stream_id' in cols else None
    sel = "s.kernel_name, d.start, d.end, %s, %s" % ("d." + qcol if qcol else "0", "d." + scol if scol else "0")
    rows = list(cur.execute("select %s from %s d join %s s on d.kernel_id = s.id order by d.start" % (sel, kd, ks)))
    t_end = max(r[2] for r in rows)
    rows = [r for r in rows if r[1] >= t_end - window_ms * 1e6]
    span = (t_end - rows[0][1]) / 1e3
    print("# columns: %s" % cols)
    print("# last %.1f ms: %d dispatches" % (span / 1e3, len(rows)))
    by = collections.defaultdict(list)
    for r in rows:
        by[(r[3], r[4])].append(r)
    for key, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
        busy = sum(r[2] - r[1] for r in rs) / 1e3
        print("queue %s stream %s: %5d dispatches, busy %9.1f us (%.2f of the window)" % (key[0], key[1], len(rs), busy, busy / span))
    ev = []
    for r in rows:
        ev.append((r[1], 1))
        ev.append((r[2], -1))
    ev.sort()
    depth, last, hist = 0, ev[0][0], collections.Counter()
    for t, d in ev:
        hist[depth] += t - last
        last, depth = t, depth + d
    print("kernels in flight -> share of the window: " + "  ".join("%d: %.3f" % (k, v / 1e3 / span) for k, v in sorted(hist.items())))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        n = r[0].split("(")[0][:48]
        agg[n][0] += 1
        agg[n][1] += (r[2] - r[1]) / 1e3
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-48s n=%4d total %9.1f us  avg %7.2f us" % (n, c, t, t / c))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 40.0)
