#!/bin/bash
# kernel timeline of the C5 orbit (moving camera: the plan waits for the previous frame's feedback) under rocprofv3: tools/prof_c5.sh <outdir>
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/$out/prof" -o run -- python3 "$GRAFT_REPO_ROOT/bench.py" --workload c5 --steps 24 --warmup 4 --no-extra --no-cpu-baseline > "$GRAFT_REPO_ROOT/$out/c5.json" 2> "$GRAFT_REPO_ROOT/$out/c5.err"
cd "$GRAFT_REPO_ROOT"
python3 - "$out/prof/run_kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [(r["Kernel_Name"].split("(")[0].replace("void ", "").replace("vf::", ""), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
mains = [i for i, n in enumerate(names) if n[0].startswith("k_tile<false, false")]
t0 = names[mains[-6]][1]
for n, s, e in names[mains[-6]:mains[-3]]:
    print(f"{(s - t0)/1e3:9.1f} us  +{(e - s)/1e3:8.1f} us  {n}")
print(f"pose period {(names[mains[-1]][1] - names[mains[-6]][1]) / 5e3:.1f} us")
PY
