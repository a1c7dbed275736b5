#!/bin/bash
# rocprofv3 counter passes for the bench (one --pmc set per run, as the MI355X guide prescribes).
# usage: tools/pmc_passes.sh <tag> -- <program args...>   (run on the GPU box, from the repo root)
set -u
tag=$1; shift; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; counters=$2; shift 2
  timeout -s KILL 120 rocprofv3 --pmc $counters --output-format csv -d "$out/$name" -o $name -- "$@" > "$out/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$out/$name.log"; return 1; }
}
if [ "${PMC_ONLY:-}" = "icache" ]; then
  run icache "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "$@" && run dcache "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_ACTIVE_INST_SCA" "$@"
  find "$out" -name "*.csv" | head -20
  exit 0
fi
run fetch "FETCH_SIZE" "$@" && run write "WRITE_SIZE" "$@" && \
run sq "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU" "$@" && \
run sq2 "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU" "$@"
find "$out" -name "*.csv" | head -20
