#!/usr/bin/env python
"""Kernel time by family over the dispatches after the marker (sleep) kernel of a rocprofv3 kernel trace.
    family_totals_rocpd.py db [n_replays]"""
import collections
import re
import sqlite3
import sys


def family(name):
    if name.startswith('Cijk'):
        return 'library GEMM'
    m = re.search(r'(\d+)([a-z_0-9]+_kernel)', name)
    if m:
        return m.group(2)
    return 'ATen' if 'at6native' in name else name[:30]


db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))
mark = [i for i, r in enumerate(rows) if 'sleep' in r[0].lower() or 'spin' in r[0].lower()]
rows = rows[mark[-1] + 1:]
agg = collections.defaultdict(lambda: [0, 0.0])
for name, st, en in rows:
    f = family(name)
    agg[f][0] += 1
    agg[f][1] += (en - st) / 1e3
tot = sum(v[1] for v in agg.values())
print("# per replay: %.1f us of kernel time in %d dispatches" % (tot / n, len(rows) // n))
for f, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-34s n=%4d  %8.1f us" % (f, c // n, t / n))
