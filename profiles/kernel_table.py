#!/usr/bin/env python
"""Per-(kernel, grid) average durations from a rocprofv3 rocpd database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else '%'
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch_')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol_')][0]
q = ("select s.kernel_name, d.grid_size_x/d.workgroup_size_x, d.grid_size_y, d.grid_size_z, count(*), "
     "avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, s.arch_vgpr_count, s.accum_vgpr_count, d.group_segment_size from %s d join %s s "
     "on d.kernel_id=s.id where s.kernel_name like '%s' group by 1,2,3,4 order by 1,2 desc" % (kd, ks, pat))
for r in cur.execute(q):
    name = r[0].replace('_ZN3d3f', '').replace('_ZN12_GLOBAL__N_1', '')[:44]
    print("%-44s grid=(%5d,%3d,%3d) n=%4d avg=%8.1f min=%8.1f us vgpr=%3d agpr=%3d lds=%6d" % ((name,) + r[1:]))
