#!/usr/bin/env python3
"""profiles/r04_*_{bench,pmc}.json + *_kernel_stats.csv (tools/r04_profiles.sh, tools/summarize_prof.py) ->
  profiles/r04_counters.md      per workload: the steps' kernels with launches, durations, HBM-side traffic, MFMA busy, L2 hit
Every number of the table is computed here from those files."""
import csv
import json
from pathlib import Path

PROF = Path(__file__).resolve().parent.parent / "profiles"
CLOCK_HZ, SIMDS, HBM_PEAK = 2.4e9, 1024, 8000.0

rows = []
for pmc_file in sorted(PROF.glob("r04_*_pmc.json")):
    tag = pmc_file.name[len("r04_"):-len("_pmc.json")]
    d = json.loads(pmc_file.read_text())
    bench = json.loads((PROF / f"r04_{tag}_bench.json").read_text())
    stats = {r["Name"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(PROF / f"r04_{tag}_kernel_stats.csv"))}
    eng = bench.get("dense_engine") or {}
    steps = {k: v for k, v in d["launches_in_kernel_trace"].items() if v >= 220}     # the kernels of the steps: launched by every step
    for name, calls in sorted(steps.items(), key=lambda kv: -float(stats[kv[0]]["TotalDurationNs"])):
        c = d["pmc"].get(name, {})
        avg_us = float(stats[name]["AverageNs"]) / 1e3
        mean = lambda n: c[n]["mean"] if n in c else None
        hbm = (2 * mean("FETCH_SIZE") + mean("WRITE_SIZE")) * 1024 if mean("FETCH_SIZE") is not None and mean("WRITE_SIZE") is not None else None
        busy = mean("SQ_VALU_MFMA_BUSY_CYCLES")
        rows.append({
            "tag": tag, "workload": bench["config"]["workload"].split(":")[0], "step_us": bench["ms_per_step"] * 1e3, "tflops": bench["value"] / 1e3,
            "engine": f"{eng.get('chosen')} {eng.get('group')}x{eng.get('blocks_per_item')}" + (" fp32 operands" if eng.get("gemm_fp32") or eng.get("sweep_fp32") else ""),
            "kernel": name.replace("bsmr::", ""), "launches": calls, "pmc_launches": c.get("FETCH_SIZE", {}).get("launches"),
            "avg_us": avg_us, "hbm_mb": hbm / 1e6 if hbm else None,
            "alg_mb": bench["roofline"]["algorithmic_bytes"] / 1e6 if name == d.get("step_kernel", {}).get("name") else None,
            "frac": bench["roofline"]["algorithmic_bytes"] / (avg_us * 1e-6) / 1e9 / HBM_PEAK if name == d.get("step_kernel", {}).get("name") else None,
            "mfma_busy": busy / (avg_us * 1e-6 * CLOCK_HZ * SIMDS) if busy else 0.0,
            "l2_hit": mean("TCC_HIT_sum") / mean("TCC_REQ_sum") if mean("TCC_REQ_sum") else None,
            "breakdown": bench.get("step_breakdown_us"),
        })

out = ["# Round 4: counter view of the steps' kernels", "",
       "Computed by `tools/r04_tables.py` from `profiles/r04_<tag>_{bench,pmc}.json` and `r04_<tag>_kernel_stats.csv` "
       "(`tools/r04_profiles.sh`: the un-profiled bench line first, then `rocprofv3 --kernel-trace --stats` and separate `--pmc` passes that "
       "REPLAY the line's tuned choice).  launches = calls in the kernel trace (200 steps + 20 warm-up + bench.py's event-timed loops) / in the "
       "FETCH_SIZE pass; HBM-side = 2 x FETCH_SIZE + WRITE_SIZE per launch (the guide's correction for 16-byte streams: an upper bound); "
       "frac = algorithmic bytes (SURVEY 8d) / traced average duration / 8 TB/s; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs).", "",
       "| workload | step µs (TFLOP/s) | engine | kernel | launches (trace / pmc) | avg µs | HBM-side MB (x algorithmic) | frac of 8 TB/s | MFMA busy | L2 hit |",
       "|---|---|---|---|---|---|---|---|---|---|"]
for r in rows:
    traffic = f"{r['hbm_mb']:.1f}" + (f" ({r['hbm_mb'] / r['alg_mb']:.2f}x of {r['alg_mb']:.1f})" if r["alg_mb"] else "") if r["hbm_mb"] else "-"
    out.append(f"| {r['workload']} | {r['step_us']:.2f} ({r['tflops']:.1f}) | {r['engine']} | `{r['kernel']}` | {r['launches']} / {r['pmc_launches']} | "
               f"{r['avg_us']:.2f} | {traffic} | {r['frac']:.3f} |" if r["frac"] else
               f"| {r['workload']} | | | `{r['kernel']}` | {r['launches']} / {r['pmc_launches']} | {r['avg_us']:.2f} | {traffic} | - |")
    out[-1] += f" {100 * r['mfma_busy']:.1f} % | {100 * r['l2_hit']:.0f} % |" if r["l2_hit"] is not None else " - | - |"
out += ["", "Step breakdown of the same bench lines (µs per step; `host_enqueue` = the loop without the final synchronise, `gpu` = HIP events "
        "around the same loop, `kernels_sum` = event-timed kernels back to back):", ""]
seen = set()
for r in rows:
    if r["tag"] in seen or not r["breakdown"]:
        continue
    seen.add(r["tag"])
    b = r["breakdown"]
    out.append(f"* {r['workload']}: wall {b['wall']}, host_enqueue {b['host_enqueue']}, gpu {b['gpu']}, kernels_sum {b['kernels_sum']} -> {b['bound']}-bound")
(PROF / "r04_counters.md").write_text("\n".join(out) + "\n")
print("\n".join(out))
