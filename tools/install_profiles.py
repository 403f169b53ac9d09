#!/usr/bin/env python3
"""Copy one tools/profile_round.sh session from gpurun_out/<tag>/ into profiles/<prefix>_* and rebuild profiles/pmc_traffic.json.

usage: install_profiles.py <tag under gpurun_out> <prefix in profiles/>      e.g.  install_profiles.py r02_m r02

profiles/pmc_traffic.json is what bench.py reads for roofline.traffic and the SQ utilisation figures.  Every entry carries the
sha256 of the libvf_hip.so the counters were collected on (lib_sha256): bench.py reports them only when that is the library it
loaded, and says "traffic_stale" otherwise.

Units / corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB.  WRITE_SIZE is exact for our stores.
FETCH_SIZE reports 1/2 of the bytes of wide coalesced streams on gfx950, so it is doubled (the guide's correction); our reads are
4 to 16 B per lane, which makes the doubled figure an upper bound -- both values are recorded.
"""
import collections
import csv
import json
import os
import shutil
import sys

tag, prefix = sys.argv[1:3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(root, "gpurun_out", tag), os.path.join(root, "profiles")
lib_hash = open(os.path.join(G, "lib_sha256.txt")).read().strip()


def last_json_line(path):
    return [l for l in open(path).read().strip().splitlines() if l.startswith("{")][-1]


LAUNCHES = collections.Counter()                            # kernel -> launches seen in the pass read last


def per_kernel(path, agg_last=None):
    """{(kernel, counter): average per launch} over all launches (or the last `agg_last`)."""
    vals = collections.defaultdict(lambda: collections.OrderedDict())
    LAUNCHES.clear()
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
        d = int(r["Dispatch_Id"])
        vals[k][d] = vals[k].get(d, 0.0) + float(r["Counter_Value"])
    out = {}
    for k, v in vals.items():
        LAUNCHES[k[0]] = max(LAUNCHES[k[0]], len(v))
        xs = list(v.values())
        xs = xs[-agg_last:] if agg_last else xs
        out[k] = sum(xs) / len(xs)
    return out


def copy(src, dst):
    shutil.copy(os.path.join(G, src), os.path.join(P, f"{prefix}_{dst}"))


open(os.path.join(P, f"{prefix}_bench.json"), "w").write(last_json_line(os.path.join(G, "bench.json")) + "\n")
open(os.path.join(P, f"{prefix}_bench_c5.json"), "w").write(last_json_line(os.path.join(G, "bench_c5.json")) + "\n")
for n in (2, 4, 8):
    if os.path.exists(os.path.join(G, f"rehearse_{n}ranks.json")):
        open(os.path.join(P, f"{prefix}_rehearse_{n}ranks.json"), "w").write(last_json_line(os.path.join(G, f"rehearse_{n}ranks.json")) + "\n")
copy("kernel_stats.csv", "kernel_stats.csv")
for n in ("fetch_default", "write_default", "fetch_fill", "write_fill", "frag_fetch_default", "frag_write_default", "frag_fetch_fill", "frag_write_fill", "frag_l1_default", "frag_l1_fill"):
    copy(f"pmc_{n}.csv", f"pmc_{n}.csv")
for n in ("ranks.log", "rank_timeline.log", "top_items.log", "rank_frames.log", "rank0_stitch.log", "cold.log", "fragment.log", "parity_soak.log"):
    copy(n, n)
for n in ("phase_cycles.log", "gantt.log", "rank_sq_counters.txt", "line_loops.log", "stripes.log", "hw_shift64_probe.log",      # round 4
          "one_shot.json", "c5_orbit.log", "c5_gantt.log", "tlb_l2_counters.txt", "stream_packets.log"):                                  # round 5
    if os.path.exists(os.path.join(G, n)):
        copy(n, n)
open(os.path.join(P, f"{prefix}_pytest_gpu.log"), "w").write("".join(open(os.path.join(G, "pytest_gpu.log")).readlines()[-3:]))

# ---- SQ utilisation: default camera, and (round 6) the top-down camera and the C5 orbit -----------------------------------
def sq_passes(suffix):
    sq, lines = {}, []
    for n in ("sq_a", "sq_b"):
        path = os.path.join(G, f"pmc_{n}{suffix}.csv")
        if not os.path.exists(path):
            return {}, []
        pk = per_kernel(path)
        # the frame's main launch exists with and without line groups (round 4; the handle probes both for a few frames): the one that drew most frames
        mains = [k for k in LAUNCHES if k.startswith("vf::k_tile<false, false")]
        main = max(mains, key=lambda k: LAUNCHES[k]) if mains else ""
        lines.append(f"pmc_{n}{suffix}.csv (averages per launch; main tile kernel of the run: {main}, {LAUNCHES.get(main, 0)} launches)")
        for (kern, ctr), v in sorted(pk.items()):
            if "k_tile" in kern or "k_block_setup" in kern:
                lines.append(f"  {kern:34s} {ctr:26s} {v:16.0f}   ({LAUNCHES[kern]} launches)")
            if kern == main:
                sq[ctr] = v
            if kern == "vf::k_block_setup" and ctr == "SQ_INSTS_VALU":
                sq["setup_SQ_INSTS_VALU"] = v                # the set-up pass runs in the tile kernel's idle issue slots: its instructions count against the same frame
    return sq, lines


def sq_block(sq, what):
    return {
        "valu_busy_frac": sq["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * sq["GRBM_GUI_ACTIVE"] / 8),
        "valu_wave_insts": sq["SQ_INSTS_VALU"], "salu_wave_insts": sq["SQ_INSTS_SALU"], "lds_wave_insts": sq["SQ_INSTS_LDS"],
        "active_lanes_per_valu_inst": sq["SQ_THREAD_CYCLES_VALU"] / sq["SQ_ACTIVE_INST_VALU"],
        "setup_valu_wave_insts": sq.get("setup_SQ_INSTS_VALU"),
        "wave_cycles_active_wait_stall": [sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"], sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
                                          sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]],
        "note": f"k_tile fast variant, {what}; busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs) -- the counter ticks in quad-cycles, "
                f"one per instruction at least, so this is an upper bound of the issue utilisation; passes {prefix}_sq_counters.txt",
    }


sq, lines = sq_passes("")
sq_fill, lines_fill = sq_passes("_fill")
sq_c5, lines_c5 = sq_passes("_c5")
open(os.path.join(P, f"{prefix}_sq_counters.txt"), "w").write("\n".join(lines + lines_fill + lines_c5) + "\n")

path = os.path.join(P, "pmc_traffic.json")
data = json.load(open(path)) if os.path.exists(path) else {}
frame = ["vf::k_block_boxes", "vf::k_block_setup", "vf::k_plan", "vf::k_plan_sort", "vf::k_clear", "vf::k_tile"]
for cam in ("default", "fill"):
    f = per_kernel(os.path.join(G, f"pmc_fetch_{cam}.csv"), agg_last=3)
    lf = dict(LAUNCHES)
    w = per_kernel(os.path.join(G, f"pmc_write_{cam}.csv"), agg_last=3)
    lw = dict(LAUNCHES)

    def fold(d, ctr, launches):
        """the tile kernel runs as two launches (main variant, then the complete variant for items it handed over): one entry.  The main
        launch exists with and without line groups (round 4; a handle probes both for a few frames): only the one that drew most frames counts"""
        mains = [k for k in launches if k.startswith("vf::k_tile<false, false")]
        main = max(mains, key=lambda k: launches[k]) if mains else None
        out = collections.defaultdict(float)
        for (k, c), v in d.items():
            if c == ctr:
                if k.startswith("vf::k_tile<false, false") and k != main:
                    continue
                out["vf::k_tile" if k.startswith("vf::k_tile<false") else k] += v
        return out
    f, w = fold(f, "FETCH_SIZE", lf), fold(w, "WRITE_SIZE", lw)
    tile_f, tile_w = f.get("vf::k_tile", 0.0), w.get("vf::k_tile", 0.0)
    entry = {
        "tag": prefix, "lib_sha256": lib_hash,
        "k_tile": {"FETCH_SIZE_KiB": tile_f, "WRITE_SIZE_KiB": tile_w},
        "frame_kernels": {k: {"FETCH_SIZE_KiB": f.get(k, 0.0), "WRITE_SIZE_KiB": w.get(k, 0.0)} for k in frame},
        "hbm_bytes_per_launch": int((2 * tile_f + tile_w) * 1024),            # k_tile, FETCH doubled (guide's gfx950 correction)
        "hbm_bytes_per_launch_fetch_uncorrected": int((tile_f + tile_w) * 1024),
        "hbm_bytes_per_frame_all_kernels": int((2 * sum(f.get(k, 0.0) for k in frame) + sum(w.get(k, 0.0) for k in frame)) * 1024),
    }
    if cam == "default" and sq:
        entry["sq"] = sq_block(sq, "default camera")
    if cam == "fill" and sq_fill:
        entry["sq"] = sq_block(sq_fill, "top-down camera")
    data[f"4096x4096_g4096_{cam}_n1"] = entry

# ---- C5 (1920x1080, grid 2048, the 64-pose orbit through one batch call): traffic and SQ, per pose (= per tile-kernel launch) -------
if os.path.exists(os.path.join(G, "pmc_fetch_c5.csv")) and os.path.exists(os.path.join(G, "pmc_write_c5.csv")):
    f = per_kernel(os.path.join(G, "pmc_fetch_c5.csv")); lf = dict(LAUNCHES)
    w = per_kernel(os.path.join(G, "pmc_write_c5.csv")); lw = dict(LAUNCHES)

    def fold_c5(d, ctr, launches):
        mains = [k for k in launches if k.startswith("vf::k_tile<false, false")]
        main = max(mains, key=lambda k: launches[k]) if mains else None
        out = collections.defaultdict(float)
        for (k, c), v in d.items():
            if c == ctr and not (k.startswith("vf::k_tile<false, false") and k != main):
                out["vf::k_tile" if k.startswith("vf::k_tile<false") else k] += v
        return out
    f, w = fold_c5(f, "FETCH_SIZE", lf), fold_c5(w, "WRITE_SIZE", lw)
    entry = {"tag": prefix, "lib_sha256": lib_hash, "k_tile": {"FETCH_SIZE_KiB": f.get("vf::k_tile", 0.0), "WRITE_SIZE_KiB": w.get("vf::k_tile", 0.0)},
             "hbm_bytes_per_launch": int((2 * f.get("vf::k_tile", 0.0) + w.get("vf::k_tile", 0.0)) * 1024),
             "hbm_bytes_per_launch_fetch_uncorrected": int((f.get("vf::k_tile", 0.0) + w.get("vf::k_tile", 0.0)) * 1024),
             "note": "averages over the launches of the run: the orbit's poses differ"}
    if sq_c5:
        entry["sq"] = sq_block(sq_c5, "C5 orbit, average pose")
    data["1920x1080_g2048_orbit_n1"] = entry

# ---- the fragment stage on its own (k_resolve4): measured traffic per camera next to the algorithmic bytes -----------------
for cam in ("default", "fill"):
    ff = per_kernel(os.path.join(G, f"pmc_frag_fetch_{cam}.csv"))
    fw = per_kernel(os.path.join(G, f"pmc_frag_write_{cam}.csv"))
    l1 = per_kernel(os.path.join(G, f"pmc_frag_l1_{cam}.csv"))
    fetch = sum(v for (k, c), v in ff.items() if "k_resolve" in k and c == "FETCH_SIZE")
    write = sum(v for (k, c), v in fw.items() if "k_resolve" in k and c == "WRITE_SIZE")
    c = {cn: v for (k, cn), v in l1.items() if "k_resolve" in k}
    data[f"4096x4096_g4096_frag_{cam}"] = {
        "tag": prefix, "lib_sha256": lib_hash, "kernel": "k_resolve4 (vf_terrain_debug_fragment_stage)",
        "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
        "hbm_bytes_per_launch": int((2 * fetch + write) * 1024), "hbm_bytes_per_launch_fetch_uncorrected": int((fetch + write) * 1024),
        "algorithmic_bytes": 201326592,
        "l1": {"accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "requests_to_l2": c.get("TCP_TCC_READ_REQ_sum"), "l2_hits": c.get("TCC_HIT_sum"),
               "l2_misses": c.get("TCC_MISS_sum"),
               "l1_stalled_on_l2_frac": (c["TCP_PENDING_STALL_CYCLES_sum"] / 256.0) / (c["GRBM_GUI_ACTIVE"] / 8.0) if c.get("GRBM_GUI_ACTIVE") else None},
        "note": "averages per launch (tools/exp_fragment.py: one warm-up + 10 timed launches); FETCH doubled per the guide's gfx950 correction in hbm_bytes_per_launch"}
data.pop("fragment_stage_k_resolve", None)
json.dump(data, open(path, "w"), indent=1)
b = json.loads(open(os.path.join(P, f"{prefix}_bench.json")).read())
print("bench:", round(b["value"]), "Mpix/s", round(b["ms_per_step"], 4), "ms; kernel", round(b["roofline"]["kernel_ms"], 4), "ms; hbm frac", round(100 * b["roofline"]["frac"], 2),
      "%; other:", b.get("other_camera"), "check:", b.get("gathered_frame_equals_single_rank_frame"))
print("sq:", data["4096x4096_g4096_default_n1"].get("sq"))
print("traffic default:", data["4096x4096_g4096_default_n1"]["hbm_bytes_per_launch"], "fill:", data["4096x4096_g4096_fill_n1"]["hbm_bytes_per_launch"])
