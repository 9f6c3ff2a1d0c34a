"""Where one ADMM iteration's time goes outside the sweeps, from a rocprofv3 --kernel-trace CSV of
bench.py: per-kernel busy time and launch counts of one iteration (between two consensus kernels),
and the largest idle gaps on the stream with the kernels on either side.

    python tools/step_timeline.py bench_kernel_trace.csv [iteration-from-the-end, default 2]"""
import csv
import sys
from collections import defaultdict


def short(name):
    return name.split("(")[0][-46:]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    marks = [i for i, e in enumerate(ev) if "k_consensus" in e[2]]
    seg = ev[marks[-back - 1]:marks[-back]]
    span = seg[-1][1] - seg[0][0]
    busy = sum(e - s for s, e, _ in seg)
    print("iteration: %.3f ms, kernels busy %.3f ms, %d launches" % (span / 1e6, busy / 1e6, len(seg)))
    per = defaultdict(lambda: [0, 0])
    for s, e, n in seg:
        per[short(n)][0] += e - s
        per[short(n)][1] += 1
    for k, v in sorted(per.items(), key=lambda x: -x[1][0]):
        print("%-48s %9.1f us %4d" % (k, v[0] / 1e3, v[1]))
    gaps = sorted(((seg[i + 1][0] - seg[i][1], short(seg[i][2]), short(seg[i + 1][2])) for i in range(len(seg) - 1)),
                  reverse=True)
    print("idle between kernels: %.3f ms" % (sum(g[0] for g in gaps) / 1e6))
    bypair = defaultdict(lambda: [0, 0])
    for g, a, b in gaps:
        bypair[(a[-24:], b[-24:])][0] += g
        bypair[(a[-24:], b[-24:])][1] += 1
    for k, v in sorted(bypair.items(), key=lambda x: -x[1][0])[:25]:
        print("%9.1f us %3d x  %s -> %s" % (v[0] / 1e3, v[1], k[0], k[1]))


if __name__ == "__main__":
    main()
