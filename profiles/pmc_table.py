#!/usr/bin/env python
"""Average PMC counter values per kernel from a rocprofv3 --pmc rocpd database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else '%kpconv%'
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
pmc = [t for t in tabs if t.startswith('rocpd_pmc_event_')][0]
info = [t for t in tabs if t.startswith('rocpd_info_pmc_')][0]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
q = ("select s.kernel_name, i.name, avg(p.value), count(*) from %s p join %s i on p.pmc_id = i.id join %s d on p.event_id = d.event_id "
     "join %s s on d.kernel_id = s.id where s.kernel_name like '%s' group by 1, 2 order by 1, 2" % (pmc, info, kd, ks, pat))
last = None
for name, cname, val, n in cur.execute(q):
    short = name.replace('_ZN3d3f', '')[:40]
    if short != last:
        print(short)
        last = short
    print("    %-34s %16.1f   (n=%d)" % (cname, val, n))
