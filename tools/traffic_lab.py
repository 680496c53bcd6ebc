#!/usr/bin/env python3
"""Traffic lab (GPU box): one workload, several plan-time knob sets, the SAME launches per set - so that a counter
pass of this script can be cut into per-set chunks by dispatch order.

  python3 tools/traffic_lab.py WORKLOAD MODE "A=1,B=2" "C=3" ...           times (best of 3 x 50 launches), plan facts
  rocprofv3 --pmc FETCH_SIZE -d DIR -- python3 tools/traffic_lab.py --counted WORKLOAD MODE sets...
  python3 tools/traffic_lab.py --summarize DIR [DIR ...] -- sets...         per set: mean counters of the dense kernel

--counted launches every set exactly LAUNCHES times and nothing else from the dense family."""
import csv
import glob
import json
import os
import sys
from pathlib import Path

LAUNCHES = 6
REPO = Path(__file__).resolve().parent.parent
DENSE = ("denseStream", "denseGroups", "denseTiles", "denseShared", "denseSweep")


def summarize(dirs, sets):
    out = [dict(knobs=s) for s in sets]
    for d in dirs:
        files = glob.glob(str(Path(d) / "**" / "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        f = max(files, key=lambda x: Path(x).stat().st_mtime)
        rows = [r for r in csv.DictReader(open(f)) if any(n in r["Kernel_Name"] for n in DENSE)]
        per = {}
        for r in rows:
            per.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
            per[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
        for name, by in per.items():
            vals = [by[k] for k in sorted(by)]
            if len(vals) != LAUNCHES * len(sets):
                print(f"{name}: {len(vals)} dense dispatches, expected {LAUNCHES * len(sets)}", file=sys.stderr)
                continue
            for i in range(len(sets)):
                chunk = vals[i * LAUNCHES + 1:(i + 1) * LAUNCHES]     # (first launch of a set: cold instruction cache etc.)
                out[i][name] = sum(chunk) / len(chunk)
    for o in out:
        if "FETCH_SIZE" in o:
            o["fetch_MB_x2"] = round(2 * o["FETCH_SIZE"] * 1024 / 1e6, 1)       # KiB; gfx950: half of a 16-B-per-lane stream is counted
        if "WRITE_SIZE" in o:
            o["write_MB"] = round(o["WRITE_SIZE"] * 1024 / 1e6, 1)
        if "TCC_HIT_sum" in o and "TCC_MISS_sum" in o:
            o["l2_hit"] = round(o["TCC_HIT_sum"] / max(1.0, o["TCC_HIT_sum"] + o["TCC_MISS_sum"]), 3)
        print(json.dumps(o))


def main():
    args = sys.argv[1:]
    if args and args[0] == "--summarize":
        cut = args.index("--")
        return summarize(args[1:cut], args[cut + 1:])
    counted = args and args[0] == "--counted"
    if counted:
        args = args[1:]
    name, mode_name, sets = args[0], args[1], args[2:] or [""]
    sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
    sys.path.insert(0, str(REPO))
    import torch

    import bsmr_amd as eng
    import synth
    from bench import WORKLOADS

    mode = {"f16": 0, "bf16": 1, "f32": 2}[mode_name]
    gen, kwargs, K, alpha, delta = WORKLOADS[name]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    dev = torch.device("cuda:0")
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    base = eng.Pipeline(csr, alpha=alpha, delta=delta, device=-1 if counted or os.environ.get("BSMR_CLUSTER") == "host" else 0)
    arrays = base.arrays()
    A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
    B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
    P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    touched = set()
    for kn in sets:
        for k in touched:
            os.environ.pop(k, None)
        env = dict(x.split("=") for x in kn.split(",") if x)
        touched |= set(env)
        os.environ.update(env)
        st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
        assert st == 0, st
        if counted:
            for _ in range(LAUNCHES):
                eng.sddmm(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), mode, s)
            torch.cuda.synchronize()
            print(json.dumps({"knobs": kn, "launches": LAUNCHES}), flush=True)
        else:
            best = None
            for _ in range(3):
                t = eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), mode, s, warmup=5, iters=50)
                if best is None or t["total_ms"] < best["total_ms"]:
                    best = t
            ps = eng.PlanStats()
            eng.hip().bsmr_plan_get_stats(plan, eng.C.byref(ps))
            stats = {k: getattr(ps, k) for k, _ in eng.PlanStats._fields_ if k in ("num_dense_blocks", "dense_work_items", "num_sparse_entries", "num_dense_entries", "group_size")}
            print(json.dumps({"knobs": kn, **{k: round(v * 1e3, 2) for k, v in best.items()}, **stats}), flush=True)
        eng.plan_destroy(plan)


if __name__ == "__main__":
    main()
