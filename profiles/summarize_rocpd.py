#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the per-kernel summary committed under profiles/."""
import sqlite3
import sys


def main(db_path, title, steps):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
    tot = cur.execute("select sum(end-start)/1e3 from %s" % kd).fetchone()[0]
    print("# " + title)
    print("# %d steps profiled; total kernel time %.1f us  (%.2f ms/step)" % (steps, tot, tot / steps / 1e3))
    print("%-100s %7s %12s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    q = ("select s.kernel_name, count(*), sum(d.end-d.start)/1e3, avg(d.end-d.start)/1e3 from %s d join %s s "
         "on d.kernel_id = s.id group by s.kernel_name order by 3 desc limit 70" % (kd, ks))
    for r in cur.execute(q):
        print("%-100s %7d %12.1f %10.1f %6.1f%%" % (r[0][:100], r[1], r[2], r[3], 100 * r[2] / tot))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]))
