"""Per-launch HBM traffic of the two sweeps from two rocprofv3 counter passes (FETCH_SIZE and
WRITE_SIZE, one pass each) over `bench.py --steps 1 --warmup 0 --no-cpu-baseline`.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv BENCH_LINE.json OUT.json [--tcol-factor F]

Counter values are KB per dispatch.  On gfx950 FETCH_SIZE tallies a coalesced streaming read that
asks for whole 128-byte lines (16 bytes per lane, or a wave-wide run of consecutive dwords: the
forward sweep's raw count did not move when its stream went from the first form to the second) at
half its bytes, so the bytes a kernel streams that way are added once more; the factor for
the 8-byte-per-lane stream of the column-block copy's column field is measured separately (a run of
the transposed sweep with its gathers switched off, whose bytes are known) and passed in.
"""
import argparse
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                cnt[row["Kernel_Name"]] += 1
    return {k: (tot[k] / cnt[k] * 1024.0, cnt[k]) for k in tot}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("bench_line")
    ap.add_argument("out")
    ap.add_argument("--tcol-factor", type=float, default=1.0,
                    help="fraction of the 8-byte-per-lane column-field stream that FETCH_SIZE reports")
    args = ap.parse_args()
    line = json.loads(open(args.bench_line).read().strip().splitlines()[-1])
    cfg = line["config"]
    explicit = "stored as f32" in cfg.get("values", "stored as f32")
    hb = cfg["rank0_hbm_bytes"]
    per_entry_copy = 10 if explicit else 6
    copied = hb["column_block_copy"] / per_entry_copy       # entries not held by the row masks (approx.)
    fetch = per_kernel(args.fetch_csv, "FETCH_SIZE")
    write = per_kernel(args.write_csv, "WRITE_SIZE")
    kern = line["roofline"]["kernels"]
    out = {
        "about": __doc__.strip(),
        "workload": {"rows_total": cfg["rows_total"], "dim": cfg["dim"], "n_gpus": line["n_gpus"],
                     "explicit_values": explicit},
        "tcol_factor": args.tcol_factor,
        "kernels": {},
    }
    for name, (fbytes, n) in fetch.items():
        if "k_forward<1" in name:
            s16 = copied * (8 if explicit else 4)            # col (+ val) of the compact row-major form
            s8, alg = 0.0, kern["forward_csr_sweep"]["algorithmic_bytes"]
        elif "k_transposed<false" in name:
            s16 = copied * (8 if explicit else 4)            # row (+ val) fields, 16 bytes per lane
            s8, alg = copied * 2, kern["transposed_block_sweep"]["algorithmic_bytes"]
        else:
            continue
        w = write.get(name, (0.0, 0))[0]
        traffic = fbytes + s16 / 2 + s8 * (1.0 - args.tcol_factor) + w
        out["kernels"][name] = {
            "launches_profiled": n, "FETCH_SIZE_bytes_raw": fbytes, "WRITE_SIZE_bytes": w,
            "streamed_16B_bytes": s16, "streamed_8B_bytes": s8, "traffic_bytes": traffic,
            "algorithmic_bytes": alg, "traffic_over_algorithmic": traffic / alg,
        }
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    sys.exit(main())
