#!/usr/bin/env python3
"""What a burst of n SDDMM steps costs outside its kernels: wall time of n back-to-back steps of the
headline workload for several n, ended by torch.cuda.synchronize() alone and by a polling loop (stream.query()) in front of it.
wall(n) = fixed + n * per_step: the fit says how much of `ms_per_step` at --steps 20 is the burst's fixed cost.
  python tools/step_latency_lab.py [workload]"""
import ctypes as C
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import numpy as np
import torch

import bench
import bsmr_amd as eng
import synth

name = sys.argv[1] if len(sys.argv) > 1 else "nips_k128_dense"
gen, kwargs, K, alpha, delta = bench.WORKLOADS[name][:5]
rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
os.environ["BSMR_DENSE_ENGINE"] = "tuned"
dev = torch.device("cuda:0")
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
tA = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
tB = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
tP = torch.zeros(int(ci.size), dtype=torch.float32, device=dev)
sh = torch.cuda.current_stream(dev).cuda_stream
plan = pipe.plan
eng.hip().bsmr_plan_reserve(plan, K)
mode = eng.COMPUTE_F16
t_tune = time.perf_counter()
tuned = eng.plan_tune(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, sh)
print("chosen", tuned["chosen"], tuned["group"], tuned["blocks_per_item"], f"(bsmr_plan_tune took {(time.perf_counter() - t_tune) * 1e3:.0f} ms)", flush=True)
step = lambda: eng.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, sh)


def burst(n, poll):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t1 = time.perf_counter()
    if poll:
        while not torch.cuda.current_stream(dev).query():
            pass
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return (t2 - t0) * 1e6, (t1 - t0) * 1e6


# the first bursts after the tuner, one by one: what a single timed region (bench.py's) sees
# (LAB_PRE=n: n more steps and a synchronise in front of them; LAB_SLEEP_MS: a pause)
for _ in range(int(os.environ.get("LAB_PRE", "0"))):
    step()
torch.cuda.synchronize()
time.sleep(float(os.environ.get("LAB_SLEEP_MS", "0")) / 1e3)
first = [burst(20, False) for _ in range(6)]
print("first bursts of 20 after the tuner (wall, enqueue us):", [(round(w, 1), round(e, 1)) for w, e in first], flush=True)
if os.environ.get("LAB_FIRST_ONLY"):
    sys.exit(0)
for poll in (False, True):
    pts = []
    for n in (1, 2, 5, 10, 20, 50, 100, 200):
        walls = sorted(burst(n, poll) for _ in range(15))
        wall, enq = walls[len(walls) // 2]
        pts.append((n, wall))
        print(f"{'poll' if poll else 'sync'} n={n:4d}: wall {wall:8.1f} us ({wall / n:6.2f} per step), enqueue {enq:7.1f} us", flush=True)
    n = np.array([p[0] for p in pts], float)
    w = np.array([p[1] for p in pts], float)
    b, a = np.polyfit(n, w, 1)
    print(f"{'poll' if poll else 'sync'}: wall(n) = {a:.1f} us + n x {b:.2f} us", flush=True)
