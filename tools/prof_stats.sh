#!/bin/bash
# rocprofv3 kernel-trace summary of one bench.py run (GPU box): tools/prof_stats.sh <outdir under gpurun_out> [bench args...]
# prints the per-kernel table; the CSVs stay under gpurun_out/<outdir>/
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o run -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra "$@" > "$GRAFT_REPO_ROOT/$out/bench_prof.json" 2> "$GRAFT_REPO_ROOT/$out/bench_prof.err"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
cp "$f" "$out/kernel_stats.csv"
python3 - "$out/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:9.1f} min_us={float(r['MinNs'])/1e3:9.1f} max_us={float(r['MaxNs'])/1e3:9.1f} pct={r['Percentage']}")
PY
