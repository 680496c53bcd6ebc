#!/usr/bin/env python3
"""Extracts, from the result logs the reference ships
(scripts/results_suiteSparse_dataset/BSMR_results/BSMR_k_*_a_*_d_*.log, written by
sddmm_testMode on an RTX 4090), the records of the SuiteSparse matrices whose sparsity
pattern can be regenerated exactly without the collection (synth.py: Trefethen_20000,
Trefethen_20000b, mycielskian14/15, wathen100/120).  Only data is kept: per matrix its
dimensions and per (alpha, delta, K) the integers the pipeline logged.  Timings and GFLOP/s
are dropped (hardware-dependent).

    python tests/golden/make_reference_logs.py [/root/reference] > tests/golden/reference_logs.json
"""
import glob
import json
import re
import sys
from pathlib import Path

MATRICES = ["Trefethen_20000", "Trefethen_20000b", "mycielskian14", "mycielskian15", "wathen100", "wathen120"]
INT_KEYS = ["NumRowPanel", "original_numDenseBlock", "bsmr_numClusters", "bsmr_numDenseBlock",
            "bsmr_numDenseThreadBlocks", "bsmr_numSparseThreadBlocks", "bsmr_numDenseData", "bsmr_numSparseData"]
TEXT_KEYS = ["original_averageDensity", "bsmr_averageDensity", "bsmr_threadBlockRatio", "bsmr_dataRatio",
             "gridDim_dense", "gridDim_sparse"]


def fields(record: str):
    return {k.strip(): v.strip() for k, v in re.findall(r"\[([^\[\]:]+?)\s*:\s*([^\[\]]*)\]", record)}


def main():
    ref = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
    logs = sorted(glob.glob(str(ref / "scripts/results_suiteSparse_dataset/BSMR_results/BSMR_k_*_a_*_d_*.log")))
    out = {m: {"runs": []} for m in MATRICES}
    for path in logs:
        k, a, d = re.match(r".*BSMR_k_(\d+)_a_([\d.]+)_d_([\d.]+)\.log", path).groups()
        for record in Path(path).read_text().split("---New data---")[1:]:
            f = fields(record)
            name = Path(f["File"]).stem
            if name not in out or Path(f["File"]).parent.name != name:
                continue
            dims = {"M": int(f["M"]), "N": int(f["N"]), "NNZ": int(f["NNZ"])}
            entry = out[name]
            assert entry.setdefault("dims", dims) == dims
            run = {"K": int(f["K"]), "alpha": float(f["bsmr_alpha"]), "delta": float(f["bsmr_delta"])}
            assert (run["K"], abs(run["alpha"] - float(a)) < 1e-6, abs(run["delta"] - float(d)) < 1e-6) == (int(k), True, True)
            run.update({key: int(f[key]) for key in INT_KEYS})
            run.update({key: f[key] for key in TEXT_KEYS})
            entry["runs"].append(run)
    # one record per (alpha, delta): everything but the sparse grid is independent of K
    for entry in out.values():
        merged = {}
        for run in entry["runs"]:
            k = str(run.pop("K"))
            grid = run.pop("gridDim_sparse")
            rec = merged.setdefault((run["alpha"], run["delta"]), dict(run, gridDim_sparse={}))
            assert {key: v for key, v in rec.items() if key != "gridDim_sparse"} == run, "fields vary with K"
            assert rec["gridDim_sparse"].setdefault(k, grid) == grid
        entry["runs"] = [merged[key] for key in sorted(merged)]
    out = {m: e for m, e in out.items() if e["runs"]}
    text = json.dumps({"source": "scripts/results_suiteSparse_dataset/BSMR_results (reference, RTX 4090)",
                       "matrices": out}, sort_keys=True, separators=(",", ":"))
    print(text.replace('{"alpha"', '\n{"alpha"'))


if __name__ == "__main__":
    main()
