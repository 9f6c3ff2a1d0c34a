#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box into gpurun_out/prof_<tag>/:
# kernel statistics of the three bench configurations (and of a shard of 8 of the default one), separate FETCH_SIZE / WRITE_SIZE counter
# passes over one step of the default configuration, and the calibration of FETCH_SIZE for 2-, 4-,
# 8- and 16-byte-per-lane streaming reads (tools/fetch_size_calib.hip, built beforehand).
#   tools/profile_round.sh <tag>
set -e -o pipefail
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
  python3 "$root/bench.py" --steps 4 --warmup 1 --no-other-configs > "$out/bench_line.json" 2> "$out/bench_stats.err"
echo "stats done"
for cfg in c5 ftrl; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$cfg" -o bench -- \
    python3 "$root/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_${cfg}_line.json" 2> "$out/bench_${cfg}_stats.err"
  echo "stats $cfg done"
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_explicit" -o bench -- \
  python3 "$root/bench.py" --explicit-values --steps 2 --warmup 1 --no-cpu-baseline --no-other-linesearch > "$out/bench_explicit_line.json" 2> "$out/bench_explicit_stats.err"
echo "stats explicit done"
# one rank's share of the default workload on 8 GPUs, on this one GPU
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_shard8" -o bench -- \
  python3 "$root/bench.py" --rows 12500000 --steps 8 --warmup 2 --no-cpu-baseline --no-other-linesearch > "$out/bench_shard8_line.json" 2> "$out/bench_shard8_stats.err"
echo "stats shard-of-8 done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o pmc -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-linesearch --no-other-configs > "$out/pmc_fetch_line.json" 2> "$out/pmc_fetch.err"
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o pmc -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-linesearch --no-other-configs > "$out/pmc_write_line.json" 2> "$out/pmc_write.err"
echo "write done"
for cfg in c5 ftrl; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$out/${cfg}_$ctr" -o pmc -- \
      python3 "$root/bench.py" --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-other-linesearch --ftrl-pass-only > "$out/pmc_${cfg}_${ctr}_line.json" 2> "$out/pmc_${cfg}_$ctr.err"
    echo "$cfg $ctr done"
  done
done
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/calib" -o pmc -- \
  "$root/tools/fetch_size_calib" > "$out/calib_line.txt" 2> "$out/calib.err"
echo "calibration done"
# keep what is small: the kernel statistics, and the counters as per-kernel means
mkdir -p "$out/keep"
for d in stats stats_c5 stats_ftrl stats_explicit stats_shard8; do
  f=$(find "$out/$d" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/keep/${d}_kernel_stats.csv"
done
for d in fetch write c5_FETCH_SIZE c5_WRITE_SIZE ftrl_FETCH_SIZE ftrl_WRITE_SIZE calib; do
  f=$(find "$out/$d" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 "$root/tools/pmc_reduce.py" "$f" "$out/keep/${d}_counters.csv"
done
cp "$out"/*.json "$out"/*.txt "$out/keep/" 2>/dev/null || true
for d in stats stats_c5 stats_ftrl stats_explicit stats_shard8 fetch write c5_FETCH_SIZE c5_WRITE_SIZE ftrl_FETCH_SIZE ftrl_WRITE_SIZE calib; do
  rm -rf "$out/$d"
done
ls -la "$out/keep"
