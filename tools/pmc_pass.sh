#!/bin/bash
# One rocprofv3 --pmc pass over bench.py (GPU box; counters in their own run, as the pool requires):
#   tools/pmc_pass.sh <outdir under gpurun_out> "<COUNTER ...>" [bench args...]
# prints per-kernel averages (tools/pmc_sq_summary.py) for the tile / set-up kernels; CSV stays under gpurun_out/<outdir>/
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
out=gpurun_out/$1; ctr=$2; shift; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$GRAFT_REPO_ROOT/$out/pmc" -o run -- python3 "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra "$@" > "$GRAFT_REPO_ROOT/$out/bench_pmc.json" 2> "$GRAFT_REPO_ROOT/$out/bench_pmc.err"
cd "$GRAFT_REPO_ROOT"
f=$(find "$out/pmc" -name "*counter_collection.csv" | head -1)
cp "$f" "$out/counter_collection.csv"
sha256sum vulkan_forge_amd/libvf_hip.so | cut -d' ' -f1 > "$out/lib_sha256.txt"
python3 tools/pmc_sq_summary.py "$out/counter_collection.csv"
