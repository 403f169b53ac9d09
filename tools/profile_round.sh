#!/bin/bash
# Everything the round's profiles/ files are made of, in one GPU-box session:  tools/profile_round.sh <tag>   (e.g. r02)
# Writes gpurun_out/<tag>/...; tools/install_profiles.py <tag> then copies the summaries into profiles/ and refreshes profiles/pmc_traffic.json.
# rocprofv3 --pmc passes are separate runs with counters only (no trace domains), as the pool requires.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun sets it)}"
# A gpurun call lasts 20 minutes at most: `tools/profile_round.sh <tag> a` runs the first half (tests, bench lines, kernel trace, counter
# passes, fragment stage), `... <tag> b` the second (ranks, schedules, rehearsals, probes, soak); without a letter both.
tag=$1
part=${2:-ab}
out=gpurun_out/$tag
mkdir -p "$out"
R="$GRAFT_REPO_ROOT"
sha256sum vulkan_forge_amd/libvf_hip.so | cut -d' ' -f1 > "$out/lib_sha256.txt"
if [[ $part == *a* ]]; then
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -q > "$out/pytest_gpu.log" 2>&1; tail -2 "$out/pytest_gpu.log"
echo "== bench (C4)"; timeout -k 10 600 python bench.py --steps 20 --warmup 3 --check > "$out/bench.json" 2> "$out/bench.err" || exit 1
echo "== bench (C5)"; timeout -k 10 600 python bench.py --workload c5 --steps 64 --warmup 8 > "$out/bench_c5.json" 2> "$out/bench_c5.err" || exit 1
echo "== kernel trace"; (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/trace" -o run -- python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu-baseline --no-extra > "$R/$out/bench_traced.json" 2> "$R/$out/trace.err") || exit 1
cp "$(find $out/trace -name '*kernel_stats.csv' | head -1)" "$out/kernel_stats.csv"
pmc() {   # pmc <name> "<counters>" <program args...>      (every pass under a timeout: a counter set that cannot be collected hangs the run)
  local name=$1 ctr=$2; shift; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d "$R/$out/pmc_$name" -o run -- python3 "$@" > "$R/$out/pmc_$name.json" 2> "$R/$out/pmc_$name.err") || return 1
  cp "$(find $out/pmc_$name -name '*counter_collection.csv' | head -1)" "$out/pmc_$name.csv"
}
B="$R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra"
echo "== pmc passes"
pmc fetch_default "FETCH_SIZE" $B || exit 1
pmc write_default "WRITE_SIZE" $B || exit 1
pmc fetch_fill "FETCH_SIZE" $B --camera fill || exit 1
pmc write_fill "WRITE_SIZE" $B --camera fill || exit 1
pmc sq_a "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS" $B || exit 1
pmc sq_b "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" $B || exit 1
# round 6: the same two passes for the top-down camera (the slowest configuration) and the C5 orbit, plus their traffic
SQA="SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS"
SQB="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"
pmc sq_a_fill "$SQA" $B --camera fill || exit 1
pmc sq_b_fill "$SQB" $B --camera fill || exit 1
C5="$R/bench.py --workload c5 --steps 64 --warmup 8 --no-cpu-baseline --no-extra"
pmc sq_a_c5 "$SQA" $C5 || exit 1
pmc sq_b_c5 "$SQB" $C5 || exit 1
pmc fetch_c5 "FETCH_SIZE" $C5 || exit 1
pmc write_c5 "WRITE_SIZE" $C5 || exit 1
for cam in default fill; do      # the fragment stage alone (k_resolve4), one camera per pass
  pmc frag_fetch_$cam "FETCH_SIZE" "$R/tools/exp_fragment.py" $cam || exit 1
  pmc frag_write_$cam "WRITE_SIZE" "$R/tools/exp_fragment.py" $cam || exit 1
  pmc frag_l1_$cam "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "$R/tools/exp_fragment.py" $cam || exit 1
done
echo "== fragment stage"; timeout -k 10 120 python tools/exp_fragment.py both > "$out/fragment.log" 2>&1; VF_RESOLVE_PER_PIXEL=1 timeout -k 10 120 python tools/exp_fragment.py both >> "$out/fragment.log" 2>&1; timeout -k 10 120 python tools/exp_fragment.py both exact >> "$out/fragment.log" 2>&1; cat "$out/fragment.log"
fi
if [[ $part != *b* ]]; then echo "== first half done"; exit 0; fi
echo "== ranks (round-robin stripes, and -- :b -- stripes dealt by measured times)"; timeout -k 10 400 python tools/exp_ranks.py default 1:0 2:0:d 2:0:d:b 4:0:d 4:0:d:b 8:0:d 8:0:d:b > "$out/ranks.log" 2>&1; timeout -k 10 400 python tools/exp_ranks.py fill 1:0 2:0:d 2:0:d:b 2:0:0:b 4:0:d 4:0:d:b 8:0:d 8:0:d:b >> "$out/ranks.log" 2>&1; cat "$out/ranks.log"
tools/prof_rank.sh $tag/rank_trace 2 8 0 > "$out/rank_timeline.log" 2>&1; tail -3 "$out/rank_timeline.log"
timeout -k 10 300 python tools/exp_toptiles.py > "$out/top_items.log" 2>&1
timeout -k 10 300 python tools/exp_rank_frames.py 2 8 > "$out/rank_frames.log" 2>&1; grep period "$out/rank_frames.log"
echo "== rank 0's stitch"; timeout -k 10 200 python tools/exp_rank0_stitch.py 8 default > "$out/rank0_stitch.log" 2>&1; timeout -k 10 200 python tools/exp_rank0_stitch.py 8 fill >> "$out/rank0_stitch.log" 2>&1; grep period "$out/rank0_stitch.log" | tail -8
echo "== first frames"; timeout -k 10 200 python tools/exp_cold.py default > "$out/cold.log" 2>&1; timeout -k 10 200 python tools/exp_cold.py fill >> "$out/cold.log" 2>&1; grep rep "$out/cold.log"
echo "== one-shot path, cold process"; timeout -k 10 200 python tools/one_shot.py > "$out/one_shot.json" 2> "$out/one_shot.err"; cat "$out/one_shot.json"
echo "== C5: poses at rest, per-pose times of the orbit, schedules"; (VF_C5_STATIC=1 VF_C5_POSES=1 timeout -k 10 300 python tools/exp_c5.py r06) > "$out/c5_orbit.log" 2>&1; cat "$out/c5_orbit.log"
if [ -f build/variants/libvf_gantt.so ]; then for c in pose8 pose0 orbit7 orbit60; do VF_C5=1 VF_HIP_LIB=$PWD/build/variants/libvf_gantt.so timeout -k 10 200 python tools/exp_gantt.py $c; done > "$out/c5_gantt.log" 2>&1; grep tile_ms "$out/c5_gantt.log"; fi
echo "== eight virtual ranks"; timeout -k 10 600 python tools/rehearse_virtual.py 8 > "$out/rehearse_8ranks.json" 2> "$out/rehearse_8ranks.err"; echo "rc=$?"; tail -c 300 "$out/rehearse_8ranks.json"
echo "== rehearsal"; for n in 2 4; do timeout -k 10 600 python bench.py --gpus $n --rehearse --no-cpu-baseline --steps 5 > "$out/rehearse_${n}ranks.json" 2> "$out/rehearse_${n}ranks.err"; echo "rc=$?"; tail -c 300 "$out/rehearse_${n}ranks.json"; done
# round 4: where the tile kernel's time goes (build/variants/libvf_phase.so = tools/build_variant.sh phase -DVF_PHASE_PROF), one GPU and a rank of eight;
# the rank's SQ counters; when each work item starts and ends (libvf_gantt.so = -DVF_DIAG_ITEM=3); which line loop the handle picks per view
if [ -f build/variants/libvf_phase.so ]; then echo "== phase cycles"; timeout -k 10 200 python tools/exp_phases.py build/variants/libvf_phase.so > "$out/phase_cycles.log" 2>&1; echo "---- rank 2 of 8" >> "$out/phase_cycles.log"; timeout -k 10 200 python tools/exp_phases.py build/variants/libvf_phase.so 2 8 0 >> "$out/phase_cycles.log" 2>&1; grep -c cycles/pair "$out/phase_cycles.log"; fi
if [ -f build/variants/libvf_gantt.so ]; then echo "== schedule"; VF_HIP_LIB=$PWD/build/variants/libvf_gantt.so timeout -k 10 200 python tools/exp_gantt.py default > "$out/gantt.log" 2>&1; VF_HIP_LIB=$PWD/build/variants/libvf_gantt.so timeout -k 10 200 python tools/exp_gantt.py default 2 8 >> "$out/gantt.log" 2>&1; grep "tile_ms" "$out/gantt.log"; fi
echo "== rank SQ counters"; tools/pmc_rank.sh $tag/rank_sq 2 8 0 > "$out/rank_sq_counters.txt" 2>&1; tail -4 "$out/rank_sq_counters.txt"
# round 5: the rank's long wait for its vertex records (phase_cycles.log) is neither the TLB nor the L2: address-translation and L2 counters, a rank of eight and one GPU
echo "== TLB / L2 counters"; (echo "---- rank 2 of 8"; tools/pmc_tlb.sh $tag/tlb_rank 2 8 0; echo "---- one GPU"; tools/pmc_tlb.sh $tag/tlb_1gpu 0 1 0) > "$out/tlb_l2_counters.txt" 2>&1; grep -c UTCL1 "$out/tlb_l2_counters.txt"
echo "== stream packets"; (hipcc --offload-arch=gfx950 -O2 tools/micro/stream_packets.hip -o build/stream_packets && timeout -k 10 60 ./build/stream_packets) > "$out/stream_packets.log" 2>&1; tail -3 "$out/stream_packets.log"
echo "== line loops"; timeout -k 10 300 python tools/exp_groups_auto.py > "$out/line_loops.log" 2>&1; cat "$out/line_loops.log"
echo "== stripe widths"; (timeout -k 10 300 python tools/exp_ranks.py default 1:0 2:0:0 2:0:1 2:0:2 2:0:3 4:0:0 4:0:1 4:0:2 8:0:0 8:0:1 8:0:2; timeout -k 10 300 python tools/exp_ranks.py fill 1:0 2:0:0 2:0:2 4:0:0 4:0:1 8:0:0 8:0:1) > "$out/stripes.log" 2>&1; grep -c period "$out/stripes.log"
# round 4: the hardware finding behind tools/isa_lint.py -- 64-bit shifts by the last allocated VGPR -- reproduced with its controls
echo "== hardware probes"; mkdir -p build; (hipcc --offload-arch=gfx950 -O2 tools/probes/shift64_top.hip -o build/shift64_top && hipcc --offload-arch=gfx950 -O2 tools/probes/mad64_overlap.hip -o build/mad64_overlap && hipcc --offload-arch=gfx950 -O2 tools/probes/vgpr_top.hip -o build/vgpr_top) > "$out/hw_probe_build.log" 2>&1; rm -f ./*.hipfb
(echo "# tools/probes/shift64_top.hip"; timeout -k 10 60 ./build/shift64_top; echo; echo "# tools/probes/vgpr_top.hip (a value parked in the last register survives: the register is intact, the shift's READ fails)"; timeout -k 10 60 ./build/vgpr_top | grep -v "end 3 waves early"; echo; echo "# tools/probes/mad64_overlap.hip (overlapping operands of v_mad_u64_u32: all right)"; timeout -k 10 60 ./build/mad64_overlap; echo; echo "# tools/isa_lint.py on the library under test"; python tools/isa_lint.py; echo; echo "# triangle path of the library under test"; timeout -k 10 100 python tests/tri_check.py vulkan_forge_amd/libvf_hip.so) > "$out/hw_shift64_probe.log" 2>&1; grep -c "lanes wrong" "$out/hw_shift64_probe.log"
echo "== soak"; timeout -k 10 500 python tests/soak_parity.py $((500000 + RANDOM)) 100000 400 > "$out/parity_soak.log" 2>&1; tail -1 "$out/parity_soak.log"
echo "== done"
