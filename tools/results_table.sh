#!/bin/bash
# GPU box: one bench line per workload -> gpurun_out/results/<workload>.json and a markdown table.
# usage: tools/results_table.sh <tag>     (copy gpurun_out/results_<tag>.md into profiles/ afterwards)
TAG=${1:-r01}
OUT=gpurun_out/results
mkdir -p $OUT
WLS=${WLS:-"nips_k32_hybrid nips_k128_dense nips_k512_dense cop20k_k128_hybrid cop20k_blocks_k128_hybrid dlmc_k512_dense:bf16 dlmc_k512_d01:bf16 dlmc_k512_sparse:bf16 reddit_shard_k256 mycielskian15_k32 mycielskian15_k128 mycielskian15_k256 mycielskian15_k512 mycielskian14_k32 mycielskian14_k128 mycielskian14_k256 mycielskian14_k512 trefethen20000_k32 trefethen20000_k128 trefethen20000_k256 trefethen20000_k512 wathen100_k32 wathen100_k128 wathen100_k256 wathen100_k512"}
for item in $WLS; do
  wl=${item%%:*}; mode=f16; [ "$item" != "$wl" ] && mode=${item##*:}
  timeout -k 10 300 python3 bench.py --workload $wl --mode $mode --steps 200 --warmup 20 > $OUT/$wl.json 2> $OUT/$wl.err || echo "$wl failed"
  echo "$wl done"
done
python3 - "$TAG" $WLS > gpurun_out/results_$TAG.md <<'PY'
import json, sys
tag, wls = sys.argv[1], sys.argv[2:]
print(f"# bench.py on one MI355X, all workloads ({tag}); fp32 A/B/P at the boundary, conversion included; dtype per row\n")
print("| workload | dtype | us / SDDMM | GFLOP/s | published (RTX 4090) | ratio | convert / dense / sparse us | dominant kernel: alg. GB/s (frac of 8 TB/s) | pre-converted operands us | CPU port GFLOP/s (cores) | mismatches vs CPU | tuned choices |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for wl in [w.split(":")[0] for w in wls]:
    try:
        d = json.loads(open(f"gpurun_out/results/{wl}.json").read().strip().splitlines()[-1])
    except Exception as e:
        print(f"| {wl} | failed: {e} |")
        continue
    k = d["kernels_ms"]; r = d["roofline"]; pub = d.get("published_reference"); c = d.get("cpu_baseline", {})
    pre = d.get("preconverted_operands")
    e = d.get("dense_engine") or {}
    tuned = e.get("chosen", "-")
    if e.get("group", 0) > 1: tuned += f" x{e['group']}"
    if e.get("blocks_per_item"): tuned += f" ({e['blocks_per_item']} blocks/item)"
    if e.get("b_only", -1) >= 0: tuned = "B-only conversion" if e["b_only"] else "fp32 residue"
    if e.get("overlap", -1) >= 0: tuned += ", two streams" if e["overlap"] else ", one stream"
    if e.get("cvt_in_kernel", -1) == 1: tuned = "fp32 operands rounded in the dense kernel"
    print(f"| {wl} | {d['dtype']} | {d['ms_per_step'] * 1e3:.1f} | {d['value']:.0f} | {pub['gflops'] if pub else '-'} | {d['vs_baseline'] if d['vs_baseline'] else '-'} | "
          f"{k['convert_ms'] * 1e3:.1f} / {k['dense_ms'] * 1e3:.1f} / {k['sparse_ms'] * 1e3:.1f} | {r['kernel']}: {r['achieved']:.0f} ({r['frac']:.3f}) | {(str(round(pre['ms_per_step'] * 1e3, 1))) if pre else '-'} | "
          f"{c.get('value', '-')} ({c.get('cores', '-')}) | {d.get('parity_mismatches_vs_cpu', '-')} | {tuned} |")
PY
cat gpurun_out/results_$TAG.md
