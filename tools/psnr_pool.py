#!/usr/bin/env python3
"""profiles/r05_psnr_study.json from the two worker files of tools/psnr_study.sh: PSNR@20k of the product build against the
reference-numerics build, paired by seed, plus the pooled estimate over the three rounds' studies (inverse-variance weights;
each round's code differs outside the numerics under test -- the comparison inside a round is like for like).
within_0p1_db is only stated when the standard error of the estimate it is read from is below 0.05 dB."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "psnr")


def load(tag):
    for name in (tag + ".json", tag + "_partial.json"):
        p = os.path.join(src, name)
        try:
            txt = [l for l in open(p).read().splitlines() if l.startswith("{")]
            if txt:
                return json.loads(txt[-1]), name
        except Exception:
            pass
    return None, None


a, fa = load("product")
b, fb = load("refnum")
assert a and b, "worker results missing under %s" % src
ra = a["runs"] if isinstance(a.get("runs"), list) else [{"seed": s, "psnr_test_mean": p} for s, p in zip(a["seeds"], a["psnr_per_run"])]
rb = b["runs"] if isinstance(b.get("runs"), list) else [{"seed": s, "psnr_test_mean": p} for s, p in zip(b["seeds"], b["psnr_per_run"])]
pa = {r["seed"]: r["psnr_test_mean"] for r in ra}
pb = {r["seed"]: r["psnr_test_mean"] for r in rb}
seeds = sorted(set(pa) & set(pb))
d = np.array([pb[s] - pa[s] for s in seeds])
k = len(d)
paired = {"seeds": seeds, "product_db": [pa[s] for s in seeds], "reference_numerics_db": [pb[s] for s in seeds],
          "differences_db_reference_minus_product": [round(float(v), 3) for v in d], "mean_db": round(float(d.mean()), 4),
          "standard_error_db": round(float(d.std(ddof=1) / np.sqrt(k)), 4) if k > 1 else None}
va, vb = np.array(list(pa.values())), np.array(list(pb.values()))
unpaired = {"product": {"runs": len(va), "mean": round(float(va.mean()), 3), "std": round(float(va.std(ddof=1)), 3)},
            "reference_numerics": {"runs": len(vb), "mean": round(float(vb.mean()), 3), "std": round(float(vb.std(ddof=1)), 3)},
            "delta_db": round(float(vb.mean() - va.mean()), 4),
            "standard_error_db": round(float(np.sqrt(va.var(ddof=1) / len(va) + vb.var(ddof=1) / len(vb))), 4)}
# earlier rounds (delta = reference numerics - product, its standard error): r03 pooled study, r04 bench line (paired by seed)
earlier = [{"round": 3, "file": "profiles/r03_psnr_study.json", "delta_db": 0.071, "standard_error_db": 0.067, "runs": "16 + 14, one seed"},
           {"round": 4, "file": "BENCH_r04.json psnr_numerics_ab.paired_by_seed", "delta_db": -0.198, "standard_error_db": 0.124, "runs": "3 pairs"}]
est = earlier + [{"round": 5, "file": "this study, paired by seed", "delta_db": paired["mean_db"], "standard_error_db": paired["standard_error_db"], "runs": "%d pairs" % k}]
w = np.array([1.0 / e["standard_error_db"] ** 2 for e in est])
pooled_delta = float((w * np.array([e["delta_db"] for e in est])).sum() / w.sum())
pooled_se = float(1.0 / np.sqrt(w.sum()))
chi2 = float((w * (np.array([e["delta_db"] for e in est]) - pooled_delta) ** 2).sum())


def verdict(delta, se):
    return (bool(abs(delta) <= 0.1) if se < 0.05 else "inconclusive (standard error %.3f dB >= 0.05)" % se)


out = {"what": "PSNR@20k on the ngp_fox test views (reference's definition, ExpRunner.cpp:360-388): libf2n_hip.so (product: fp32 MFMA accumulation, owner-binned "
               "hash-gradient sums) against libf2n_hip_refnum.so (the reference's per-addend f16 atomics and f16 MLP accumulator); seed s of either build draws the "
               "same ray batches, march noise, background and edge samples (host/KeyedDraws.h).  There is no run of the reference itself to compare with.",
       "sources": [fa, fb], "paired_by_seed": paired, "unpaired": unpaired,
       "this_round": {"delta_db_reference_minus_product": paired["mean_db"], "standard_error_db": paired["standard_error_db"],
                      "within_0p1_db": verdict(paired["mean_db"], paired["standard_error_db"])},
       "pooled_over_rounds": {"estimates": est, "delta_db_reference_minus_product": round(pooled_delta, 4), "standard_error_db": round(pooled_se, 4),
                              "heterogeneity_chi2_2dof": round(chi2, 2), "within_0p1_db": verdict(pooled_delta, pooled_se),
                              "interval_95_db": [round(pooled_delta - 1.96 * pooled_se, 3), round(pooled_delta + 1.96 * pooled_se, 3)]},
       "workers": {"product": {k2: a.get(k2) for k2 in ("psnr_mean", "psnr_std", "train_wall_s", "valid_leaves", "per_view_mean", "per_view_std") if k2 in a},
                   "reference_numerics": {k2: b.get(k2) for k2 in ("psnr_mean", "psnr_std", "train_wall_s", "valid_leaves", "per_view_mean", "per_view_std") if k2 in b}}}
json.dump(out, open(os.path.join(ROOT, "profiles", "r05_psnr_study.json"), "w"), indent=1)
print(json.dumps({"paired": paired["mean_db"], "se": paired["standard_error_db"], "pooled": out["pooled_over_rounds"]["delta_db_reference_minus_product"],
                  "pooled_se": out["pooled_over_rounds"]["standard_error_db"], "within_0p1_db": out["pooled_over_rounds"]["within_0p1_db"]}))
