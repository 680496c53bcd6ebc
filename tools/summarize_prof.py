#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into committed evidence:
  profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary
  profiles/<round>_<tag>_pmc.json           per-kernel mean counter values
  profiles/traffic.json                     HBM bytes per launch of the dominant kernel
usage: tools/summarize_prof.py <round> <tag> <workload> <mode>"""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
rnd, tag, workload, mode = sys.argv[1:5]
src = REPO / "gpurun_out" / f"prof_{tag}"
dst = REPO / "profiles"
def newest(pattern):
    """gpurun merges a call's files into gpurun_out/ without removing those of earlier calls: one file per pass, the latest."""
    files = glob.glob(pattern, recursive=True)
    return max(files, key=lambda f: Path(f).stat().st_mtime) if files else None


stats = newest(str(src / "stats" / "**" / "*kernel_stats.csv"))
(dst / f"{rnd}_{tag}_kernel_stats.csv").write_text(Path(stats).read_text())
pmc = collections.defaultdict(dict)
for f in filter(None, (newest(str(d / "**" / "*counter_collection.csv")) for d in sorted(src.glob("pmc_*")) if d.is_dir())):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "bsmr::" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        pmc[k][c] = {"launches": len(v), "mean": sum(v) / len(v)}
bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
(dst / f"{rnd}_{tag}_bench.json").write_text(json.dumps(bench, indent=1))
out = {"bench": {k: bench[k] for k in ("value", "ms_per_step", "kernels_ms", "roofline")}, "pmc": pmc}
# HBM traffic of the dominant kernel, per launch: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950
# FETCH_SIZE counts half of a 16-B-per-lane stream (MI355X_MICROARCH.md "HBM"), so it is doubled.
dom = bench["roofline"]["kernel"]
names = {"dense": ("denseStream", "denseGroups", "denseTiles", "denseShared", "denseSweep", "denseGemm"), "sparse": ("sparseEntries",), "convert": ("convertOperands",)}[dom]
# the steps' kernel is an instantiation of the engine the bench line names; the profiler passes REPLAY the line's tuned choice
# (tools/profile_bench.sh), so that kernel is launched by the steps and the warm-up alone - no candidate engines beside it
chosen = (bench.get("dense_engine") or {}).get("chosen")
if dom == "dense" and chosen in ("stream", "tiles", "shared", "sweep", "gemm"):
    names = {"stream": ("denseStream", "denseGroups"), "tiles": ("denseTiles",), "shared": ("denseShared",), "sweep": ("denseSweep",), "gemm": ("denseGemm",)}[chosen]
rows = list(csv.DictReader(open(stats)))
calls = {r["Name"].split("(")[0].replace("void ", ""): int(r["Calls"]) for r in rows}
out["launches_in_kernel_trace"] = {k: v for k, v in calls.items() if "bsmr::" in k}
# of its instantiations the one with the most launches in the kernel trace: the steps' (200 + 20 warm-up; bench.py's event-timed
# passes add more)
candidates = [k for k in pmc if any(n in k for n in names) and "FETCH_SIZE" in pmc[k] and "WRITE_SIZE" in pmc[k]]
matching = [(calls.get(k, 0), k) for k in candidates]
if matching:
    top = max(matching)
    assert top[0] >= 220, f"the step kernel {top[1]} has {top[0]} launches in the kernel trace: fewer than steps + warm-up"
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        assert pmc[top[1]][c]["launches"] >= 25, f"{c} pass: {pmc[top[1]][c]['launches']} launches of {top[1]}"
    out["step_kernel"] = {"name": top[1], "launches": top[0], "average_ns": next(float(r["AverageNs"]) for r in rows if r["Name"].split("(")[0].replace("void ", "") == top[1])}
for k, c in pmc.items():
    if matching and k == max(matching)[1]:
        traffic = int((2 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024)
        out["traffic_bytes_per_launch"] = traffic
        tfile = dst / "traffic.json"
        t = json.loads(tfile.read_text()) if tfile.exists() else {}
        t[f"{workload}:{mode}:{dom}{bench['roofline'].get('engine_tag', '')}"] = {"bytes": traffic, "source": f"profiles/{rnd}_{tag}_pmc.json (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes of the same bench command; 2 x FETCH_SIZE + WRITE_SIZE)"}
        tfile.write_text(json.dumps(t, indent=1))
(dst / f"{rnd}_{tag}_pmc.json").write_text(json.dumps(out, indent=1))
print(open(dst / f"{rnd}_{tag}_kernel_stats.csv").read()[:1500])
print(json.dumps(out, indent=1)[:3000])
