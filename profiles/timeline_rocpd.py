#!/usr/bin/env python3
"""Timeline of the last training steps in a rocprofv3 --kernel-trace (rocpd SQLite) of tools/converged_steps.py: every
dispatch after the last idle gap >= 100 ms, with its queue, start (relative, us) and duration; then per-step spans (a step
ends with the Adam kernel: adam_fused_kernel, or adam_h16grad_kernel in older traces) and, per kernel, mean duration and how much of it ran while the OTHER queue was busy too.
Usage: timeline_rocpd.py <db> [n_steps_to_print]"""
import collections, re, sqlite3, sys

db = sys.argv[1]
n_print = int(sys.argv[2]) if len(sys.argv) > 2 else 2
cur = sqlite3.connect(db).cursor()
cols = [r[1] for r in cur.execute("pragma table_info('kernels')")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = "select name, start, end%s from kernels order by start" % ((", " + qcol) if qcol else "")
rows = [list(r) + ([0] if not qcol else []) for r in cur.execute(sel)]
cut = 0
for i in range(1, len(rows)):
    if rows[i][1] - max(r[2] for r in rows[max(0, i - 8):i]) > 100e6:
        cut = i
rows = rows[cut:]
t0 = rows[0][1]
def short(n):  # mangled or demangled: the kernel's own name (+ its template arguments as they are mangled)
    m = re.search(r"([a-z][a-z0-9_]*_kernel)(I[A-Za-z0-9_]*?E(?=v|E))?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n.split("(")[0].replace("void ", "")[:38]
steps, cur_step = [], []
for r in rows:
    cur_step.append(r)
    if "adam_fused" in r[0] or "adam_h16grad" in r[0]:
        steps.append(cur_step); cur_step = []
print("dispatches after the marker: %d, steps: %d, columns: %s" % (len(rows), len(steps), cols))
spans = [(s[-1][2] - s[0][1]) / 1e3 for s in steps]
print("per-step first-start -> adam end (us):", " ".join("%.0f" % v for v in spans))
ends = [s[-1][2] for s in steps]
print("adam-end to adam-end (us):", " ".join("%.0f" % ((b - a) / 1e3) for a, b in zip(ends, ends[1:])))
for s in steps[-n_print:]:
    print("---- step ----")
    base = s[0][1]
    for name, a, b, q in s:
        print("%9.1f  %8.1f  q%-3s %s" % ((a - base) / 1e3, (b - a) / 1e3, q, short(name)))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
all_iv = [(a, b, q) for _, a, b, q in rows]
for name, a, b, q in rows:
    ov = 0.0
    for a2, b2, q2 in all_iv:
        if q2 != q and b2 > a and a2 < b:
            ov += min(b, b2) - max(a, a2)
    k = short(name)
    agg[k][0] += 1; agg[k][1] += (b - a) / 1e3; agg[k][2] += min(ov, b - a) / 1e3
print("---- per kernel over %d steps: calls/step, mean us, share overlapped by the other queue ----" % max(len(steps), 1))
for k, (n, tot, ov) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-36s %5.1f  %8.1f us  %4.0f%%" % (k, n / max(len(steps), 1), tot / n, 100 * ov / max(tot, 1e-9)))
