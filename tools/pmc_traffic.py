"""Per-launch HBM-side traffic of the sweep kernels from two rocprofv3 counter passes (FETCH_SIZE and
WRITE_SIZE, one pass each) over `bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-other-linesearch`,
corrected with the calibration of tools/fetch_size_calib.hip.

    python tools/pmc_traffic.py FETCH.csv WRITE.csv BENCH_LINE.json CALIB.csv OUT.json

Counter values are KB per dispatch.  On gfx950 FETCH_SIZE tallies a coalesced streaming read at a
fixed fraction of its bytes that depends on the bytes per lane (the calibration program reads 1 GiB
with 2, 4, 8 and 16 bytes per lane: one half for 4, 8 and 16, next to nothing for 2); a scattered
4-byte gather that leaves the L2 is tallied as the 64-byte request it is.  So for every kernel

    traffic = FETCH_SIZE + sum over its streams of (1 - factor(width)) * bytes(stream) + WRITE_SIZE

with the streams' bytes taken from the entry counts of the layouts (config.rank0_layout_counts).
What is left of FETCH_SIZE after the streams' own share is the gathers' (and the small vectors')."""
import csv
import json
import sys
from collections import defaultdict


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                n = int(row.get("Dispatches") or 1)      # a file reduced by tools/pmc_reduce.py holds means and counts
                tot[row["Kernel_Name"]] += float(row["Counter_Value"]) * n
                cnt[row["Kernel_Name"]] += n
    return {k: (tot[k] / cnt[k] * 1024.0, cnt[k]) for k in tot}


def calibration(path):
    got = per_kernel(path, "FETCH_SIZE")
    out = {}
    for name, (b, _) in got.items():
        for key, width in (("unsigned short", 2), ("unsigned int __vector(2)", 8), ("unsigned int __vector(4)", 16),
                           ("unsigned int>", 4)):
            if "k_read<" + key in name:
                out[width] = b / float(1 << 30)
                break
    return out


def main():
    fetch_csv, write_csv, bench_line, calib_csv, out_path = sys.argv[1:6]
    line = json.loads(open(bench_line).read().strip().splitlines()[-1])
    cfg = line["config"]
    explicit = "stored as f32" in cfg.get("values", "")
    cnt = cfg["rank0_layout_counts"]
    rows = cfg["rows_total"] // line["n_gpus"]
    factor = calibration(calib_csv)
    fetch = per_kernel(fetch_csv, "FETCH_SIZE")
    write = per_kernel(write_csv, "WRITE_SIZE")
    kern = line["roofline"].get("kernels", {})
    hp, lst, tcw, cell = cnt["hot_stream_words"], cnt["list_entries"], cnt["column_stream_words"], cnt["cell_entries"]
    masked = cnt["mask_entries"] > 0
    ftrl = cfg.get("name") == "ftrl"
    launches = max(1, int(cfg.get("batches_per_pass", 1)))   # ftrl: per-launch shares of the pass's streams
    # streams per kernel: (bytes per lane of the load instruction, bytes per launch)
    streams = {
        "k_forward_rp<1": [(8, 2.0 * hp), (4, 4.0 * lst), (2, 2.0 * lst), (8, 32.0 * rows if masked else 0.0), (4, 4.0 * rows)]
                          + ([(16, 4.0 * hp), (4, 4.0 * lst)] if explicit else []),
        "k_transposed<false": [(16, 4.0 * cell), (8, 2.0 * cell)] + ([(16, 4.0 * cell)] if explicit else []),
        "k_transposed_cs<": [(8, 2.0 * tcw), (4, 4.0 * rows)] + ([(16, 4.0 * tcw)] if explicit else []),
    }
    if ftrl:
        # the mini-batch pass: one forward sweep over a batch's short panels and one cell sweep with the fused
        # update per batch; streams per LAUNCH (the counters are averaged per dispatch)
        streams = {
            "k_forward_rp<2": [(8, 2.0 * hp / launches), (4, 4.0 * lst / launches), (2, 2.0 * lst / launches),
                               (8, 32.0 * rows / launches), (4, 4.0 * rows / launches)],
            "k_transposed<true": [(16, 4.0 * cell / launches), (8, 2.0 * cell / launches)],
        }
    alg = {"k_forward_rp<1": kern.get("forward_csr_sweep", {}).get("algorithmic_bytes"),
           "k_transposed<false": kern.get("transposed_block_sweep", {}).get("algorithmic_bytes"),
           "k_transposed_cs<": None, "k_forward_rp<2": None, "k_transposed<true": None}
    out = {"about": __doc__.strip(), "fetch_size_factor_by_bytes_per_lane": factor,
           "workload": {"rows_total": cfg["rows_total"], "dim": cfg["dim"], "n_gpus": line["n_gpus"],
                        "explicit_values": explicit, "layout_counts": cnt},
           "kernels": {}}
    for name, (fbytes, n) in fetch.items():
        for key, ss in streams.items():
            if key in name:
                w = write.get(name, (0.0, 0))[0]
                stream_bytes = sum(b for _, b in ss)
                uncounted = sum((1.0 - factor.get(width, 0.5)) * b for width, b in ss)
                traffic = fbytes + uncounted + w
                out["kernels"][name] = {
                    "launches_profiled": n, "FETCH_SIZE_bytes_raw": fbytes, "WRITE_SIZE_bytes": w,
                    "streamed_bytes": stream_bytes, "streamed_bytes_not_in_FETCH_SIZE": uncounted,
                    "gather_and_vector_bytes": fbytes - (stream_bytes - uncounted),
                    "traffic_bytes": traffic, "algorithmic_bytes": alg[key],
                    "traffic_over_algorithmic": traffic / alg[key] if alg[key] else None,
                }
    k = out["kernels"]
    if ftrl:
        # everything the pass's launches moved, the small kernels included (their FETCH_SIZE / WRITE_SIZE as tallied)
        total = 0.0
        per_pass_launches = {}
        pass_kernels = ("k_forward_rp<2", "k_transposed<true", "k_ftrl_wsum", "k_max_fold", "k_ftrl_last_row", "k_ftrl_apply",
                        "k_forward<2", "k_ftrl_persist", "k_ps_init_max")
        for name, (fbytes, n) in fetch.items():
            if not any(k_ in name for k_ in pass_kernels):
                continue          # generation, the builders, the test metrics: not the pass
            w = write.get(name, (0.0, 0))[0]
            extra = k[name]["streamed_bytes_not_in_FETCH_SIZE"] if name in k else 0.0
            passes = max(1, int(line["steps"]) + int(line["warmup"]) + 1)      # timed + warm-up + the per-kernel pass
            per_pass = n / passes
            per_pass_launches[name[:60]] = per_pass
            total += (fbytes + w + extra) * per_pass
        out["pass_total"] = {"traffic_bytes": total, "algorithmic_bytes": line["roofline"]["algorithmic_bytes"],
                             "traffic_over_algorithmic": total / line["roofline"]["algorithmic_bytes"],
                             "launches_per_pass": per_pass_launches}
    cs = [v for n, v in k.items() if "k_transposed_cs<" in n]
    cells = [v for n, v in k.items() if "k_transposed<false" in n]
    if cs and cells:
        total = cs[0]["traffic_bytes"] + cells[0]["traffic_bytes"]
        out["transposed_sweep_total"] = {"traffic_bytes": total, "algorithmic_bytes": cells[0]["algorithmic_bytes"],
                                         "traffic_over_algorithmic": total / cells[0]["algorithmic_bytes"]}
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps({n[:40]: {a: b for a, b in v.items() if a in ("traffic_bytes", "traffic_over_algorithmic",
                                                                   "gather_and_vector_bytes")}
                      for n, v in k.items()}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
