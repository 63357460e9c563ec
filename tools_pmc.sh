#!/bin/bash
# Collects PMC counters for the render kernel, one rocprofv3 --pmc pass per run
# (separate from --kernel-trace/--stats runs). Usage: tools_pmc.sh <outdir> [bench args]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run_pass () {
  local name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
BENCH_ARGS="$*"
run_pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY &&
run_pass sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_FLAT &&
run_pass sq3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_VALU_FLOPS_FP32 &&
run_pass tcc1 FETCH_SIZE TCC_HIT_sum &&
run_pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum &&
run_pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
