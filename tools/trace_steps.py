#!/usr/bin/env python3
"""Timeline of one bench run from a rocprofv3 kernel trace: per kernel name the durations, and the gaps between
consecutive kernels of the steady-state steps (end of one kernel to start of the next).
usage: trace_steps.py <kernel_trace.csv> [--dump N]"""
import csv
import statistics as st
import sys


def short(name):
    name = name.replace("void ", "").replace("bsmr::", "")
    return name.split("(")[0][:48]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    dump = int(sys.argv[sys.argv.index("--dump") + 1]) if "--dump" in sys.argv else 0
    t0 = rows[0][0]
    byname = {}
    for s, e, n in rows:
        byname.setdefault(n, []).append(e - s)
    print("kernel durations (ns): name calls min p10 median p90 max")
    for n, d in sorted(byname.items(), key=lambda kv: -sum(kv[1])):
        d.sort()
        print(f"  {n:48s} {len(d):6d} {d[0]:7d} {d[len(d) // 10]:7d} {int(st.median(d)):7d} {d[(9 * len(d)) // 10]:7d} {d[-1]:7d}")
    # steady-state pairs: a kernel followed within 50 us by the next one
    gaps = {}
    periods = {}
    last_start = {}
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        g = s1 - e0
        if g < 50000:
            gaps.setdefault((n0, n1), []).append(g)
    for s, e, n in rows:
        if n in last_start and s - last_start[n] < 2000000:
            periods.setdefault(n, []).append(s - last_start[n])
        last_start[n] = s
    print("gaps end->start (ns): from -> to  count min median p90")
    for k, g in sorted(gaps.items(), key=lambda kv: -len(kv[1]))[:12]:
        g.sort()
        print(f"  {k[0]:40s} -> {k[1]:40s} {len(g):6d} {g[0]:7d} {int(st.median(g)):7d} {g[(9 * len(g)) // 10]:7d}")
    print("start-to-start period of the same kernel (ns): name count p10 median p90")
    for n, p in periods.items():
        p.sort()
        print(f"  {n:48s} {len(p):6d} {p[len(p) // 10]:8d} {int(st.median(p)):8d} {p[(9 * len(p)) // 10]:8d}")
    if dump:
        mid = len(rows) // 2
        print(f"timeline of {dump} launches from the middle of the run (start us, duration ns, gap before ns):")
        for i in range(mid, min(len(rows), mid + dump)):
            s, e, n = rows[i]
            print(f"  {(s - t0) / 1e3:12.1f} {e - s:7d} {s - rows[i - 1][1]:7d}  {n}")


if __name__ == "__main__":
    main()
