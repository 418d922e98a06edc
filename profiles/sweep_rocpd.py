#!/usr/bin/env python3
"""Kernel x phase table out of a rocprofv3 --kernel-trace (rocpd SQLite) of tools/kernel_size_sweep.py: the trace is cut at every
idle gap >= 100 ms (the script pauses 0.3 s in front of each phase's timed steps; the part in front of the first pause -- the
20 000 training iterations -- is dropped), a phase's steps end with adam_fused_kernel, and every kernel's mean duration per launch
and launches per step are printed per phase.  Usage: sweep_rocpd.py <db> [label label ...]  (labels: e.g. the phases' sample counts)"""
import collections, re, sqlite3, sys

db = sys.argv[1]
labels = sys.argv[2:]
cur = sqlite3.connect(db).cursor()
rows = [list(r) for r in cur.execute("select name, start, end from kernels order by start")]
cuts = [0]
run_end = rows[0][2]
for i in range(1, len(rows)):
    if rows[i][1] - run_end > 100e6:
        cuts.append(i)
    run_end = max(run_end, rows[i][2])
phases = [rows[a:b] for a, b in zip(cuts, cuts[1:] + [len(rows)])]
# the sweep pauses in front of AND behind each phase's timed steps: segments 1, 3, 5, ... are the timed steps, the even ones the training
# run / the next phase's batch draws and warm-up steps
phases = phases[1::2]
def short(n):  # mangled or demangled: the kernel's own name (+ its template arguments as they are mangled)
    m = re.search(r"([a-z][a-z0-9_]*_kernel)(I[A-Za-z0-9_]*?E(?=v|E))?", n)
    return (m.group(1) + (m.group(2) or "")) if m else n.split("(")[0].replace("void ", "")[:38]
table = collections.OrderedDict()
steps = []
for pi, p in enumerate(phases):
    n_steps = max(1, sum(1 for r in p if "adam_fused" in r[0]))
    steps.append(n_steps)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, a, b in p:
        agg[short(name)][0] += 1
        agg[short(name)][1] += (b - a) / 1e3
    for k, (n, tot) in agg.items():
        table.setdefault(k, {})[pi] = (n / n_steps, tot / n)
print("phases: %d   steps per phase: %s" % (len(phases), steps))
hdr = "%-40s" % "kernel (launches/step) mean us" + "".join("%16s" % (labels[i] if i < len(labels) else "phase %d" % i) for i in range(len(phases)))
print(hdr)
order = sorted(table.items(), key=lambda kv: -max(v[0] * v[1] for v in kv[1].values()))
tot = [0.0] * len(phases)
for k, d in order:
    line = "%-40s" % k
    for pi in range(len(phases)):
        if pi in d:
            line += "%9.1f x%-5.1f" % (d[pi][1], d[pi][0])
            tot[pi] += d[pi][0] * d[pi][1]
        else:
            line += "%16s" % "-"
    print(line)
print("%-40s" % "sum of kernel time per step (us)" + "".join("%16.1f" % t for t in tot))
