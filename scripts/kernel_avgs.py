"""Per-kernel duration summary of a rocprofv3 rocpd database: python scripts/kernel_avgs.py results.db [name-substring ...]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
pats = sys.argv[2:] or [""]
for pat in pats:
    for r in c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                       "where name like ? group by name order by 6 desc", ("%" + pat + "%",)):
        print("%-60s calls %4d  avg %9.1f us  min %9.1f  max %9.1f  total %8.2f ms" % (r[0][:60], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e6))
