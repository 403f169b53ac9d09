#!/bin/bash
# kernel timeline of one rank of N (tools/exp_rank_trace.py) under rocprofv3: tools/prof_rank.sh <outdir> [rank n]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o run -- python3 "$GRAFT_REPO_ROOT/tools/exp_rank_trace.py" "$@" > "$GRAFT_REPO_ROOT/$out/rank.log" 2> "$GRAFT_REPO_ROOT/$out/rank.err"
cd "$GRAFT_REPO_ROOT"
python3 - "$out/prof/run_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 6 frames: print the timeline relative to the start of a frame's main tile kernel
names = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vf::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
clears = [i for i, n in enumerate(names) if n[0].startswith("k_tile<false, false")]
if len(clears) > 8:
    t0 = names[clears[-6]][1]
    for n, s, e in names[clears[-6]:clears[-3]]:
        print(f"{(s - t0)/1e3:9.1f} us  +{(e - s)/1e3:8.1f} us  {n}")
    per = (names[clears[-1]][1] - names[clears[-6]][1]) / 5e3
    print(f"frame period {per:.1f} us")
PY
