#!/bin/bash
# rocprofv3 --pmc passes over the fragment stage alone (tools/exp_fragment.py <camera>), counters only, one pass per counter set:
#   tools/pmc_frag.sh <outdir under gpurun_out> <default|fill> "<COUNTERS>" ["<COUNTERS>" ...]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun)}"
out=gpurun_out/$1; cam=$2; shift; shift
mkdir -p "$out"
n=0
for ctr in "$@"; do
  n=$((n+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc_${cam}_$n" -o run -- python3 "$GRAFT_REPO_ROOT/tools/exp_fragment.py" $cam > "$GRAFT_REPO_ROOT/$out/pmc_${cam}_$n.log" 2>&1) || { echo "pass $n failed"; tail -5 "$out/pmc_${cam}_$n.log"; exit 1; }
  f=$(find "$out/pmc_${cam}_$n" -name "*counter_collection.csv" | head -1)
  cp "$f" "$out/pmc_${cam}_$n.csv"
  python3 - "$out/pmc_${cam}_$n.csv" <<'PY'
import csv, sys, collections
vals = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "k_resolve" not in k: continue
    vals[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
for k, d in vals.items():
    for c, v in d.items(): print(f"  {k:40s} {c:36s} {v / len(disp[k]):18.0f} per launch ({len(disp[k])} launches)")
PY
done
