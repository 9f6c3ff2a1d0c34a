#!/bin/bash
# Development aid: hardware counters of the forward sweep on the 10M-row probe, one counter group per run.
#   tools/pmc_forward.sh <tag>   -> gpurun_out/pmcf_<tag>/<group>/...
set -e -o pipefail
tag=${1:-x}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/pmcf_$tag
mkdir -p "$out"
cd /tmp
export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o pmc -- \
    python3 "$root/tools/sweep_bench.py" --reps 2 > "$out/$name.txt" 2> "$out/$name.err"
  echo "$name done"
}
run ta TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run sq SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL
