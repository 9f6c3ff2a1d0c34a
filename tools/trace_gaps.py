#!/usr/bin/env python3
"""Where an ADMM iteration's time goes outside its kernels: reads a `rocprofv3 --kernel-trace --output-format csv`
kernel trace of bench.py, takes the iterations between the third and the last `k_consensus` launch, and prints per
iteration the busy time per kernel and the idle time between consecutive kernels, attributed to the pair around it.

  rocprofv3 --kernel-trace --output-format csv -d out -o t -- python3 bench.py --rows 12500000 --steps 8 --warmup 2 ...
  tools/trace_gaps.py out/t_kernel_trace.csv
"""
import collections
import csv
import sys


def short(name):
    name = name.replace("admm_impl::", "").replace("void ", "")
    if "rocprim" in name:
        parts = name.split("detail::")
        return "rocprim:" + (parts[2][:30] if len(parts) > 2 else name[:30])
    return name.split("(")[0][:40]


def main(path):
    with open(path) as f:
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(f))
    cons = [i for i, e in enumerate(ev) if "k_consensus" in e[2]]
    if len(cons) < 4:
        raise SystemExit("fewer than four ADMM iterations in the trace")
    seg = ev[cons[2]:cons[-1] + 1]
    n_it = len(cons) - 3
    total = (seg[-1][1] - seg[0][1]) / 1e3
    busy, calls, gap, gaps = (collections.Counter() for _ in range(4))
    prev_end, prev = seg[0][1], short(seg[0][2])
    for s, e, n in seg[1:]:
        k = short(n)
        busy[k] += (e - s) / 1e3
        calls[k] += 1
        gap[(prev, k)] += max(0, s - prev_end) / 1e3
        gaps[(prev, k)] += 1
        prev_end, prev = max(prev_end, e), k
    print("iterations %d, us per iteration %.1f, in kernels %.1f, between kernels %.1f"
          % (n_it, total / n_it, sum(busy.values()) / n_it, sum(gap.values()) / n_it))
    print("\nkernel                                      us/iteration  calls/iteration  us each")
    for k, v in busy.most_common(30):
        print("%-42s %12.1f %16.1f %8.1f" % (k, v / n_it, calls[k] / n_it, v / calls[k]))
    print("\nidle between                                                                   us/iteration  times/iteration  us each")
    for k, v in gap.most_common(30):
        print("%-38s -> %-38s %10.1f %16.1f %8.1f" % (k[0], k[1], v / n_it, gaps[k] / n_it, v / gaps[k]))


if __name__ == "__main__":
    main(sys.argv[1])
