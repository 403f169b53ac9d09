#!/bin/bash
# SQ counters of one rank of N (tools/exp_rank_trace.py, 40 frames back to back) in two rocprofv3 --pmc passes:
#   tools/pmc_rank.sh <outdir under gpurun_out> [rank n skew]      -> per-kernel averages printed, CSVs under gpurun_out/<outdir>/
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; shift
mkdir -p "$out"
R="$GRAFT_REPO_ROOT"
for pass in "a:SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" "b:GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$R/$out/pmc_$name" -o run -- python3 "$R/tools/exp_rank_trace.py" "$@" > "$R/$out/pmc_$name.log" 2> "$R/$out/pmc_$name.err") || exit 1
  cp "$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)" "$out/pmc_$name.csv"
  python3 tools/pmc_sq_summary.py "$out/pmc_$name.csv"
done
