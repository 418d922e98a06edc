#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (gpurun_out/prof_*/.../*_results.db) into the small CSV summaries that are
committed under profiles/.  Usage: summarize_rocpd.py stats <db> <out.csv> | pmc <db> <out.csv>"""
import csv
import sqlite3
import sys


def short(name):
    for k in ("field_shade_fwd_kernel", "field_fwd_kernel", "field_bwd_kernel", "shade_bwd_kernel", "shade_fwd_kernel",
              "hash_bwd_kernel", "hash_gather_planes_kernel", "hash_bin_accumulate_kernel", "hash_bin_kernel", "adam_fused_kernel",
              "adam_h16grad_kernel", "adam_kernel"):
        if k in name:
            tmpl = ""
            if "field_fwd_kernelILi1ELb1ELb1" in name:
                tmpl = "<NH=1,hash,mlp>"
            elif "field_bwd_kernelILi1ELi2" in name:
                tmpl = "<NH=1,planes>"
            elif "field_bwd_kernelILi1ELi1" in name:
                tmpl = "<NH=1,atomics>"
            elif "field_bwd_kernelILi2" in name:
                tmpl = "<NH=2>"
            elif "hash_gather_planes_kernelILb1ELb1" in name:
                tmpl = "<staged,run-combining>"
            elif "hash_gather_planes_kernelILb1ELb0" in name:
                tmpl = "<staged>"
            elif "hash_gather_planes_kernelILb0" in name:
                tmpl = "<unstaged>"
            return k + tmpl
    return name.split("(")[0].replace("void ", "")[:70]


def stats(db, out):
    """One row per kernel (rocprofv3's own top_kernels view), plus one row per launch shape for kernels that are launched with
    more than one grid (e.g. the hash gather: once over every marched sample, once over the 16384 edge samples) -- an
    average over both shapes says nothing about either."""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    shapes, calls = {}, {}
    try:
        for n, g, c, tot in cur.execute("select name, grid_x, count(*), sum(duration) from kernels group by name, grid_x"):
            shapes.setdefault(n, []).append((int(g), int(c), float(tot)))
        for n, d in cur.execute("select name, duration from kernels order by name, start"):
            calls.setdefault(n, []).append(float(d))
    except sqlite3.Error:
        shapes, calls = {}, {}
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        # first_us / median_us: a kernel's first launch of a process can be several times its steady duration (cold
        # instruction cache and TLBs, clocks still ramping); the average over a handful of calls then says little
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent", "first_us", "median_us", "min_us", "max_us"])
        for n, c, t, a, p in rows:
            d = calls.get(n, [])
            unit1 = float(t) / max(sum(d), 1e-9) if d else 0.0
            extra = ["%.2f" % (d[0] * unit1), "%.2f" % (sorted(d)[len(d) // 2] * unit1), "%.2f" % (min(d) * unit1),
                     "%.2f" % (max(d) * unit1)] if d else ["", "", "", ""]
            w.writerow([short(n), c, "%.1f" % float(t), "%.2f" % float(a), "%.2f" % float(p)] + extra)
            sh = shapes.get(n, [])
            if len(sh) > 1:
                unit = float(t) / max(sum(x[2] for x in sh), 1e-9)  # kernels.duration -> the unit top_kernels reports in
                for g, cc, tot in sorted(sh, key=lambda x: -x[2]):
                    w.writerow(["  %s [grid_x=%d]" % (short(n), g), cc, "%.1f" % (tot * unit), "%.2f" % (tot * unit / cc), ""])


def pmc(db, out):
    """Counter averages per kernel AND launch shape (grid size)."""
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, grid_size, counter_name, count(*), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, grid_size, counter_name order by sum(duration) desc")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_size", "counter", "dispatches", "avg_value", "avg_duration_ns"])
        for n, g, cn, c, v, d in cur.execute(q):
            w.writerow([short(n), g, cn, c, "%.3f" % float(v), "%.0f" % float(d)])


def traffic(fetch_csv, write_csv, out_json):
    """HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950: rocprofv3 tallies 128-B read requests at 64 B, see
    MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (uncalibrated), both reported by rocprofv3 in KiB."""
    import json
    acc = {}
    for path, key, mult in ((fetch_csv, "fetch", 2.0), (write_csv, "write", 1.0)):
        best = {}  # per kernel: the launch shape with the longest average duration (the full-batch launch)
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["kernel"].split("(")[0]
                for tag in ("hash_gather_planes_kernel", "hash_bin_accumulate_kernel", "hash_bin_kernel", "field_bwd_kernel",
                            "shade_bwd_kernel", "field_shade_fwd_kernel", "shade_fwd_kernel", "ray_march_kernel<true>", "ray_march_kernel<false>"):
                    if tag in name:
                        name = tag
                if name not in best or float(r["avg_duration_ns"]) > float(best[name]["avg_duration_ns"]):
                    best[name] = r
        for name, r in best.items():
            d = acc.setdefault(name, {"dispatches": int(r["dispatches"]), "grid_size": int(r.get("grid_size", 0) or 0)})
            d[key + "_bytes_per_launch"] = float(r["avg_value"]) * 1024.0 * mult
    for d in acc.values():
        d["hbm_bytes_per_launch"] = d.get("fetch_bytes_per_launch", 0.0) + d.get("write_bytes_per_launch", 0.0)
    with open(out_json, "w") as f:
        json.dump(acc, f, indent=1, sort_keys=True)


def schema(db, out):
    cur = sqlite3.connect(db).cursor()
    with open(out, "w") as f:
        for name, typ in list(cur.execute("select name, type from sqlite_master where type in ('table','view') order by name")):
            try:
                cols = [r[1] for r in cur.execute("pragma table_info('%s')" % name)]
            except Exception as e:  # noqa: BLE001
                cols = [str(e)]
            f.write("%s %s: %s\n" % (typ, name, ", ".join(cols)))


if __name__ == "__main__":
    if sys.argv[1] == "schema":
        schema(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
