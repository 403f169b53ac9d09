#!/bin/bash
# TLB / L2 counters of the tile kernel, one GPU and one rank of N:  tools/pmc_tlb.sh <outdir under gpurun_out> [rank n layout]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; shift
mkdir -p "$out"
R="$GRAFT_REPO_ROOT"
for pass in "a:TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum GRBM_GUI_ACTIVE" "b:TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$R/$out/pmc_$name" -o run -- python3 "$R/tools/exp_rank_trace.py" "$@" > "$R/$out/pmc_$name.log" 2> "$R/$out/pmc_$name.err") || exit 1
  cp "$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)" "$out/pmc_$name.csv"
  python3 tools/pmc_sq_summary.py "$out/pmc_$name.csv"
done
