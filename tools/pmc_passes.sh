#!/bin/bash
# Collects PMC counters for the render kernel, one rocprofv3 --pmc pass per run
# (separate from --kernel-trace/--stats runs). Usage: tools_pmc.sh <outdir> <passes> [bench args]
set -u
OUT=$1; PASSES=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
BENCH_ARGS="$*"
run_pass () {
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
for p in $PASSES; do
  case $p in
    sq1) run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY || exit 1;;
    sq2) run_pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_FLAT || exit 1;;
    sq3) run_pass sq3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_VALU_FLOPS_FP32 || exit 1;;
    tcp) run_pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum || exit 1;;
    tcc1) run_pass tcc1 FETCH_SIZE TCC_HIT_sum || exit 1;;
    tcc2) run_pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum || exit 1;;
    grbm) run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1;;
    sqc) run_pass sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE || exit 1;;
    ta) run_pass ta TA_BUSY_avr TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum TCP_GATE_EN1_sum || exit 1;;
    ta2) run_pass ta2 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_GATE_EN2_sum || exit 1;;
  esac
done
