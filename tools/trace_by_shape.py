#!/usr/bin/env python3
"""Aggregate a rocprofv3 --kernel-trace database by (kernel, grid, workgroup): python tools/trace_by_shape.py <results.db> [skip_first_n_dispatches]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, count(*), sum(d.end-d.start)/1e3, "
     f"avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id "
     f"group by 1,2,3,4,5 order by 7 desc")
rows = list(cur.execute(q))
tot = sum(r[6] for r in rows)
print(f"total {tot/1e3:.2f} ms over {sum(r[5] for r in rows)} dispatches")
print("kernel,grid,wg,count,total_us,avg_us,min_us")
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 60]:
    name = r[0].replace("_ZN12_GLOBAL__N_1", "").split("Ev")[0][:60]
    print(f"{name},{r[1]}x{r[2]}x{r[3]},{r[4]},{r[5]},{r[6]:.0f},{r[7]:.1f},{r[8]:.1f}")
