#!/usr/bin/env python3
"""profiles/r03_*_{bench,pmc}.json + *_kernel_stats.csv (tools/r03_profiles.sh, tools/summarize_prof.py) ->
  profiles/r03_dlmc_sweep.md / .csv    BASELINE configs[4] per delta
  profiles/r03_counters.md             counter view of the steps' kernels of every r03 profile
Every number of the tables is computed here from those files (the csv holds the inputs and the results per row)."""
import csv
import json
from pathlib import Path

PROF = Path(__file__).resolve().parent.parent / "profiles"
CLOCK_HZ, SIMDS = 2.4e9, 1024


DENSE = ("denseStream", "denseGroups", "denseTiles", "denseShared", "denseSweep")


def load(tag):
    bench = json.loads((PROF / f"r03_{tag}_bench.json").read_text())
    pmc = json.loads((PROF / f"r03_{tag}_pmc.json").read_text())["pmc"]
    stats = {}
    for r in csv.DictReader(open(PROF / f"r03_{tag}_kernel_stats.csv")):
        stats[r["Name"].split("(")[0].replace("void ", "")] = r
    return bench, pmc, stats


ENGINE_KERNELS = {"stream": ("denseStream", "denseGroups"), "tiles": ("denseTiles",), "shared": ("denseShared",), "sweep": ("denseSweep",)}


def steps_kernel(pmc, names, bench=None, stats=None):
    """the kernel of the timed steps: of the engine the bench line names (a tuned run also launches every other candidate
    a few times), the instantiation whose traced duration is closest to the bench line's kernel time"""
    chosen = ((bench or {}).get("dense_engine") or {}).get("chosen")
    want_ms = None
    if names is DENSE:
        want_ms = bench["kernels_ms"]["dense_ms"] if bench else None
        if chosen in ENGINE_KERNELS:
            names = ENGINE_KERNELS[chosen]
    cands = [k for k, c in pmc.items() if any(n in k for n in names) and "FETCH_SIZE" in c]
    if not cands:
        return None
    if want_ms and stats:
        return min(cands, key=lambda k: abs(float(stats[k]["AverageNs"]) / 1e6 - want_ms) if k in stats else 1e9)
    return max(cands, key=lambda k: pmc[k]["FETCH_SIZE"]["launches"])


def view(pmc, stats, kernel):
    c = pmc[kernel]
    trace_us = float(stats[kernel]["AverageNs"]) / 1e3 if kernel in stats else None
    # a tuned run launches the winner in the steps and every candidate a few times: the trace's average is the steps' for the winner
    mean = lambda n: c[n]["mean"] if n in c else None
    hbm = (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024 if mean("FETCH_SIZE") is not None and mean("WRITE_SIZE") is not None else None
    return {"kernel": kernel, "trace_us": trace_us, "hbm_bytes": hbm,
            "mfma_busy": mean("SQ_VALU_MFMA_BUSY_CYCLES") / (trace_us * 1e-6 * CLOCK_HZ * SIMDS) if trace_us and mean("SQ_VALU_MFMA_BUSY_CYCLES") else 0.0,
            "l2_hit": mean("TCC_HIT_sum") / mean("TCC_REQ_sum") if mean("TCC_REQ_sum") else None,
            "hbm_gbs": hbm / (trace_us * 1e-6) / 1e9 if hbm and trace_us else None}


rows = []
for tag, delta, plan in (("dlmc_dense", 0.0, "RPHM split as is"), ("dlmc_d01", 0.1, "RPHM split as is"), ("dlmc_d03", 0.3, "RPHM split as is"),
                         ("dlmc_d05", 0.5, "RPHM split as is"), ("dlmc_sparse", 1.1, "RPHM split as is"),
                         ("dlmc_rules", "any", "plan rules (every delta promotes to this plan)")):
    bench, pmc, stats = load(tag)
    k, cfg, roof = bench["kernels_ms"], bench["config"], bench["roofline"]
    dense = view(pmc, stats, steps_kernel(pmc, DENSE, bench, stats)) if k["dense_ms"] > 0 else None
    sparse = view(pmc, stats, steps_kernel(pmc, ("sparseEntries",))) if k["sparse_ms"] > 0 else None
    dom = dense if roof["kernel"] == "dense" else sparse
    rows.append({
        "tag": tag, "delta": delta, "plan": plan, "us_per_sddmm": round(bench["ms_per_step"] * 1e3, 2), "useful_tflops": round(bench["value"] / 1e3, 2),
        "convert_us": round(k["convert_ms"] * 1e3, 2), "dense_us": round(k["dense_ms"] * 1e3, 2), "residue_us": round(k["sparse_ms"] * 1e3, 2),
        "dense_blocks": cfg["dense_blocks"], "dense_entries": cfg["dense_nnz"], "residue_entries": cfg["sparse_nnz"],
        "dense_engine": (bench.get("dense_engine") or {}).get("chosen"), "dense_kernel": dense["kernel"] if dense else "",
        "mfma_executed_tflops": roof.get("mfma_executed_tflops"), "mfma_busy_pct": round(100 * dense["mfma_busy"], 1) if dense else "",
        "dominant": roof["kernel"], "algorithmic_bytes": roof["algorithmic_bytes"], "algorithmic_gbs": roof["achieved"], "frac_of_8000": roof["frac"],
        "dominant_trace_us": round(dom["trace_us"], 2), "dominant_hbm_bytes": int(dom["hbm_bytes"]), "dominant_hbm_gbs": round(dom["hbm_gbs"]),
        "traffic_over_algorithmic": round(dom["hbm_bytes"] / roof["algorithmic_bytes"], 2), "dominant_l2_hit_pct": round(100 * dom["l2_hit"], 1)})
with open(PROF / "r03_dlmc_sweep.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0]))
    w.writeheader()
    w.writerows(rows)
md = ["# Round 3: BASELINE configs[4] - 4096^2 Bernoulli(0.1) (nnz 1 678 821), K = 512, bf16 operands, one MI355X", "",
      "Per delta with the RPHM's dense / sparse split taken as is (`BSMR_PROMOTE_AVERAGE=0 BSMR_FOLD_DENSE_BELOW=0`), and the plan the",
      "shipping rules build for every delta.  `tools/r03_profiles.sh dlmc_*` = one `bench.py` line + `rocprofv3 --kernel-trace --stats` +",
      "separate `--pmc` passes per row (`profiles/r03_dlmc_*`); `tools/r03_tables.py` computes this table and `r03_dlmc_sweep.csv`, which",
      "holds every input.  useful = 2 nnz K / time; executed = MFMA tiles x 2 x 256 x K / dense kernel time; MFMA busy =",
      "`SQ_VALU_MFMA_BUSY_CYCLES` / (trace time x 2.4 GHz x 1024 SIMDs); HBM side = 2 x `FETCH_SIZE` + `WRITE_SIZE` of the dominant kernel.", "",
      "| delta | plan | us / SDDMM | useful TFLOP/s | convert / dense / residue us | blocks | dense / residue entries | dense kernel | executed TFLOP/s (of 2 500) | MFMA busy | dominant: alg. GB/s (frac of 8 000) | HBM-side MB per launch (x alg.) | HBM-side GB/s | L2 hit |",
      "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    md.append(f"| {r['delta']} | {r['plan']} | {r['us_per_sddmm']} | {r['useful_tflops']} | {r['convert_us']} / {r['dense_us']} / {r['residue_us']} | {r['dense_blocks']} | "
              f"{r['dense_entries']} / {r['residue_entries']} | `{r['dense_kernel']}` | {r['mfma_executed_tflops']} | {r['mfma_busy_pct']} % | "
              f"{r['dominant']}: {r['algorithmic_gbs']:.0f} ({r['frac_of_8000']}) | {r['dominant_hbm_bytes'] / 1e6:.1f} ({r['traffic_over_algorithmic']} x) | {r['dominant_hbm_gbs']} | {r['dominant_l2_hit_pct']} % |")
(PROF / "r03_dlmc_sweep.md").write_text("\n".join(md) + "\n")

out = ["# Round 3: counter view of the steps' kernels (rocprofv3, separate `--pmc` passes of the same bench command; `profiles/r03_*`)", "",
       "Same definitions as `profiles/r02_results.md`.  A tuned run launches every candidate engine a few times; the kernel listed is the one the timed steps ran.", "",
       "| profile | bench us / step | kernel | us (trace) | MFMA busy | HBM-side MB per launch | x algorithmic | HBM side GB/s (of 8 000) | L2 hit |", "|---|---|---|---|---|---|---|---|---|"]
for tag in ("nips_k128", "nips_k512", "myc15_k128", "reddit_shard", "dlmc_rules", "dlmc_dense", "dlmc_d01", "dlmc_d03", "dlmc_d05", "dlmc_sparse"):
    bench, pmc, stats = load(tag)
    roof = bench["roofline"]
    for fam, names in (("dense", DENSE), ("sparse", ("sparseEntries",)), ("convert", ("convertOperands",))):
        if bench["kernels_ms"][f"{fam}_ms"] <= 0:
            continue
        kname = steps_kernel(pmc, names, bench, stats)
        if not kname:
            continue
        v = view(pmc, stats, kname)
        ratio = f"{v['hbm_bytes'] / roof['algorithmic_bytes']:.2f}" if fam == roof["kernel"] else "-"
        out.append(f"| {tag} | {bench['ms_per_step'] * 1e3:.2f} | `{kname}` | {v['trace_us']:.1f} | {100 * v['mfma_busy']:.1f} % | {v['hbm_bytes'] / 1e6:.1f} | {ratio} | {v['hbm_gbs']:.0f} | {100 * v['l2_hit']:.0f} % |")
(PROF / "r03_counters.md").write_text("\n".join(out) + "\n")
print("\n".join(md[-7:]))
print("\n".join(out[-24:]))
