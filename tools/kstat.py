#!/usr/bin/env python3
"""Print the kernel table (calls, avg ns, total ns) of a rocprofv3 --kernel-trace output directory (rocpd .db)."""
import glob
import sqlite3
import sys

for db in glob.glob(sys.argv[1] + "/**/*.db", recursive=True):
    c = sqlite3.connect(db)
    q = "select name, count(*), avg(duration), sum(duration) from kernels group by name order by sum(duration) desc limit %d"
    for r in c.execute(q % int(sys.argv[2] if len(sys.argv) > 2 else 14)):
        print(r[0][:100], r[1], round(r[2]), r[3])
