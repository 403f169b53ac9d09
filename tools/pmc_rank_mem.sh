#!/bin/bash
# Memory-side counters of the tile kernel on one rank of N (tools/exp_rank_trace.py): L1 / L2 hit rates and round-trip latencies.
#   tools/pmc_rank_mem.sh <outdir under gpurun_out> [rank n skew]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; shift
mkdir -p "$out"
R="$GRAFT_REPO_ROOT"
for pass in "m1:TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum" "m2:TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum"; do
  name=${pass%%:*}; ctr=${pass#*:}
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$R/$out/pmc_$name" -o run -- python3 "$R/tools/exp_rank_trace.py" "$@" > "$R/$out/pmc_$name.log" 2> "$R/$out/pmc_$name.err") || { echo "pass $name failed"; exit 1; }
  cp "$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)" "$out/pmc_$name.csv"
  python3 - "$out/pmc_$name.csv" <<'PY'
import csv, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_tile<false, false" not in k and "k_block_setup" not in k: continue
    vals[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, d in vals.items():
    for c, v in d.items(): print(f"  {k:40s} {c:36s} {v / len(disp[k]):18.0f} per launch ({len(disp[k])} launches)")
PY
done
