#!/usr/bin/env python3
"""Turns rocprofv3's rocpd SQLite output (gpurun_out/prof_*/.../*_results.db) into the small CSV summaries that are
committed under profiles/.  Usage: summarize_rocpd.py stats <db> <out.csv> | pmc <db> <out.csv>"""
import csv
import sqlite3
import sys


def short(name):
    for k in ("field_fwd_kernel", "field_bwd_kernel", "shade_bwd_kernel", "shade_fwd_kernel", "hash_bwd_kernel",
              "adam_h16grad_kernel", "adam_kernel"):
        if k in name:
            tmpl = ""
            if "field_fwd_kernelILi1ELb1ELb1" in name:
                tmpl = "<NH=1,hash,mlp>"
            elif "field_bwd_kernelILi1ELb1" in name:
                tmpl = "<NH=1,hash>"
            return k + tmpl
    return name.split("(")[0].replace("void ", "")[:70]


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for n, c, t, a, p in rows:
            w.writerow([short(n), c, "%.1f" % float(t), "%.2f" % float(a), "%.2f" % float(p)])


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    q = ("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(duration) desc")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "avg_value", "avg_duration_ns"])
        for n, cn, c, v, d in cur.execute(q):
            w.writerow([short(n), cn, c, "%.3f" % float(v), "%.0f" % float(d)])


def traffic(fetch_csv, write_csv, out_json):
    """HBM-side bytes per launch = 2 x FETCH_SIZE (gfx950: rocprofv3 tallies 128-B read requests at 64 B, see
    MI355X_MICROARCH.md, HBM section) + WRITE_SIZE (uncalibrated), both reported by rocprofv3 in KiB."""
    import json
    acc = {}
    for path, key, mult in ((fetch_csv, "fetch", 2.0), (write_csv, "write", 1.0)):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["kernel"].split("(")[0]
                for tag in ("hash_gather_planes_kernel", "hash_bin_kernel", "hash_bin_accumulate_kernel", "field_bwd_kernel",
                            "shade_bwd_kernel", "shade_fwd_kernel", "ray_march_kernel<true>", "ray_march_kernel<false>"):
                    if tag in name:
                        name = tag
                d = acc.setdefault(name, {"dispatches": int(r["dispatches"])})
                d[key + "_bytes_per_launch"] = float(r["avg_value"]) * 1024.0 * mult
    for d in acc.values():
        d["hbm_bytes_per_launch"] = d.get("fetch_bytes_per_launch", 0.0) + d.get("write_bytes_per_launch", 0.0)
    with open(out_json, "w") as f:
        json.dump(acc, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
