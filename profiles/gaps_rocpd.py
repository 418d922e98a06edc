#!/usr/bin/env python3
"""Idle-gap analysis of a rocprofv3 --kernel-trace run (rocpd SQLite): where does the GPU wait for the host?
Usage: gaps_rocpd.py <db> [min_gap_us]   -- prints the kernels that follow the largest idle gaps, aggregated by name."""
import collections
import sqlite3
import sys

db = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
cur = sqlite3.connect(db).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tables else None
if view is None:
    sys.exit("no `kernels` view in %s: %s" % (db, tables))
rows = list(cur.execute("select name, start, end from kernels order by start"))
busy = sum(e - s for _, s, e in rows)
span = rows[-1][2] - rows[0][1]
print("dispatches %d  span %.3f ms  busy %.3f ms (%.1f%%)" % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span))
gap_after = collections.defaultdict(lambda: [0, 0.0])
prev_end, prev_name = rows[0][2], rows[0][0]
for name, s, e in rows[1:]:
    g = (s - prev_end) / 1e3
    if g >= min_gap:
        k = (prev_name.split("(")[0][-60:], name.split("(")[0][-60:])
        gap_after[k][0] += 1
        gap_after[k][1] += g
    prev_end, prev_name = max(prev_end, e), name
tot = sum(v[1] for v in gap_after.values())
print("idle in gaps >= %.0f us: %.3f ms" % (min_gap, tot / 1e3))
for (a, b), (n, g) in sorted(gap_after.items(), key=lambda kv: -kv[1][1])[:30]:
    print("%8.1f us total  %4d x %7.1f us   %s  ->  %s" % (g, n, g / n, a, b))
