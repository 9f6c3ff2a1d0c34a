#!/bin/bash
# Collects the rocprofv3 evidence of one round on the GPU box into gpurun_out/prof_<tag>/:
# kernel statistics of the default bench run, and separate FETCH_SIZE / WRITE_SIZE counter passes.
#   tools/profile_round.sh <tag>
set -e -o pipefail
tag=${1:-rX}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o bench -- \
  python3 "$root/bench.py" --steps 4 --warmup 1 > "$out/bench_line.json" 2> "$out/bench_stats.err"
echo "stats done"
for cfg in c5 ftrl; do
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$cfg" -o bench -- \
    python3 "$root/bench.py" --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > "$out/bench_${cfg}_line.json" 2> "$out/bench_${cfg}_stats.err"
  echo "stats $cfg done"
done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -o pmc -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-linesearch > "$out/pmc_fetch_line.json" 2> "$out/pmc_fetch.err"
echo "fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -o pmc -- \
  python3 "$root/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-other-linesearch > "$out/pmc_write_line.json" 2> "$out/pmc_write.err"
echo "write done"
# calibration of the 8-byte-per-lane stream: transposed sweep with its gathers switched off
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/calib" -o pmc -- \
  python3 "$root/tools/sweep_bench.py" --no-gather > "$out/calib_line.txt" 2> "$out/calib.err"
echo "calibration done"
find "$out" -name "*.csv" | head -20
