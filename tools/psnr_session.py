#!/usr/bin/env python3
"""One more entry of profiles/psnr_estimates.json from the two worker files of tools/psnr_study.sh: (reference numerics - product) PSNR@20k of this
session with its unpaired and paired standard errors, and the runs themselves in profiles/<tag>_psnr_runs.json.
Usage: psnr_session.py <dir with product.json / refnum.json> <session label> <tag>"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, label, tag = sys.argv[1], sys.argv[2], sys.argv[3]


def load(name):
    for n in (name + ".json", name + "_partial.json"):
        try:
            txt = [l for l in open(os.path.join(src, n)).read().splitlines() if l.startswith("{")]
            if txt:
                j = json.loads(txt[-1])
                runs = j["runs"] if isinstance(j.get("runs"), list) else [{"seed": s, "psnr_test_mean": p} for s, p in zip(j["seeds"], j["psnr_per_run"])]
                return {r["seed"]: r["psnr_test_mean"] for r in runs}, n
        except Exception:
            pass
    raise SystemExit("no worker result for " + name)


pa, fa = load("product")
pb, fb = load("refnum")
va, vb = np.array(list(pa.values())), np.array(list(pb.values()))
seeds = sorted(set(pa) & set(pb))
d = np.array([pb[s] - pa[s] for s in seeds])
entry = {"session": label, "runs": "%d + %d (%d paired seeds)" % (len(va), len(vb), len(seeds)),
         "delta_db": round(float(vb.mean() - va.mean()), 4),
         "standard_error_unpaired_db": round(float(np.sqrt(va.var(ddof=1) / len(va) + vb.var(ddof=1) / len(vb))), 4),
         "standard_error_paired_db": round(float(d.std(ddof=1) / np.sqrt(len(d))), 4) if len(d) > 1 else None,
         "paired_delta_db": round(float(d.mean()), 4) if len(d) else None,
         "product": {"mean": round(float(va.mean()), 3), "std": round(float(va.std(ddof=1)), 3)},
         "reference_numerics": {"mean": round(float(vb.mean()), 3), "std": round(float(vb.std(ddof=1)), 3)}}
with open(os.path.join(ROOT, "profiles", tag + "_psnr_runs.json"), "w") as f:
    json.dump({"sources": [fa, fb], "seeds": seeds, "product_db": [pa[s] for s in seeds], "reference_numerics_db": [pb[s] for s in seeds], "estimate": entry}, f, indent=1)
p = os.path.join(ROOT, "profiles", "psnr_estimates.json")
j = json.load(open(p))
j["estimates"] = [e for e in j["estimates"] if e.get("session") != label] + [entry]
json.dump(j, open(p, "w"), indent=1)
print(json.dumps(entry))
