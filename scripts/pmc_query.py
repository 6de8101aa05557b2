"""Prints per-kernel averages of the counters in rocprofv3 sqlite outputs: pmc_query.py <kernel substring> <db> [<db> ...]"""
import sqlite3, sys
key = sys.argv[1]
for f in sys.argv[2:]:
    con = sqlite3.connect(f)
    rows = list(con.execute("select counter_name, avg(value), count(*), avg(duration) from counters_collection where kernel_name like ? group by counter_name", ('%' + key + '%',)))
    for r in rows:
        print('%-34s %14.6g   (n=%d, avg dur %.0f us)' % (r[0], r[1], r[2], r[3] / 1e3))
