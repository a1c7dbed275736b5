#!/bin/bash
# memory-pipeline counter passes (TCP / TA / TD / TCC / SQ vmem), one --pmc set per run.
# usage: tools/pmc_mem_passes.sh <tag> -- <program args...>   (GPU box, repo root)
set -u
tag=$1; shift; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; counters=$2; shift 2
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d "$out/$name" -o $name -- "$@" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
}
run tcp "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "$@" && \
run tatd "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE" "$@" && \
run sqmem "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "$@" && \
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "$@" && \
run lat "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum TCP_TA_TCP_STATE_READ_sum" "$@"
