#!/usr/bin/env python3
"""Copy one profile pass from gpurun_out/ into profiles/ under a tag, replacing the previous tag's files, and refresh
profiles/pmc_traffic.json (FETCH/WRITE passes via tools/pmc_summary.py, SQ passes via tools/pmc_sq_summary.py).

usage: install_profiles.py NEW_TAG OLD_TAG PMC_PREFIX      e.g.  install_profiles.py r01_q r01_p pmc22
expects gpurun_out/: bench_<NEW_TAG-with-last-_-kept>.log, prof_<tag>/run_kernel_stats.csv, <PMC_PREFIX>_{default,fill}_{FETCH,WRITE}_SIZE/,
<PMC_PREFIX>_sq_{a,b}/, shards_<tag>.log, toptiles_<tag>.log, schedule_<tag>.log, phases_<tag>.log, pytest_gpu.log  (tag = NEW_TAG without '_')."""
import json, os, re, shutil, subprocess, sys
new, old, pmc = sys.argv[1:4]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
short = new.replace("_", "", 1) if new.count("_") == 2 else new            # r01_q -> r01q
short = new[:3] + new[4:] if new[3] == "_" else new
names = ["bench.json", "kernel_stats.csv", "pmc_FETCH_SIZE_default.csv", "pmc_WRITE_SIZE_default.csv", "emulated_shards.log", "top_items.log",
         "schedule.log", "phase_cycles.log", "pytest_gpu.log", "sq_counters.txt"]
for n in names:
    f = os.path.join(P, f"{old}_{n}")
    if os.path.exists(f): os.remove(f)
open(os.path.join(P, f"{new}_bench.json"), "w").write(open(os.path.join(G, f"bench_{new}.log")).read().strip().splitlines()[-1] + "\n")
shutil.copy(os.path.join(G, f"prof_{short}", "run_kernel_stats.csv"), os.path.join(P, f"{new}_kernel_stats.csv"))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    shutil.copy(os.path.join(G, f"{pmc}_default_{c}", "run_counter_collection.csv"), os.path.join(P, f"{new}_pmc_{c}_default.csv"))
for src, dst in (("shards", "emulated_shards"), ("toptiles", "top_items"), ("schedule", "schedule"), ("phases", "phase_cycles")):
    shutil.copy(os.path.join(G, f"{src}_{short}.log"), os.path.join(P, f"{new}_{dst}.log"))
open(os.path.join(P, f"{new}_pytest_gpu.log"), "w").write("".join(open(os.path.join(G, "pytest_gpu.log")).readlines()[-3:]))
for cam in ("default", "fill"):
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "pmc_summary.py"), f"4096x4096_g4096_{cam}_n1",
                           os.path.join(G, f"{pmc}_{cam}_FETCH_SIZE", "run_counter_collection.csv"),
                           os.path.join(G, f"{pmc}_{cam}_WRITE_SIZE", "run_counter_collection.csv"), short], stdout=subprocess.DEVNULL)
sq = subprocess.check_output([sys.executable, os.path.join(root, "tools", "pmc_sq_summary.py"),
                              os.path.join(G, f"{pmc}_sq_a", "run_counter_collection.csv"), os.path.join(G, f"{pmc}_sq_b", "run_counter_collection.csv")]).decode()
open(os.path.join(P, f"{new}_sq_counters.txt"), "w").write(sq)
v = {m.group(1): int(m.group(2)) for m in re.finditer(r"vf::k_tile<false, false>\s+(\S+)\s+(\d+)", sq)}
path = os.path.join(P, "pmc_traffic.json")
d = json.load(open(path))
k = "4096x4096_g4096_default_n1"
d[k]["sq"] = {"valu_busy_frac": v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * v["GRBM_GUI_ACTIVE"] / 8), "valu_wave_insts": float(v["SQ_INSTS_VALU"]),
              "salu_wave_insts": float(v["SQ_INSTS_SALU"]), "lds_wave_insts": float(v["SQ_INSTS_LDS"]),
              "active_lanes_per_valu_inst": v["SQ_THREAD_CYCLES_VALU"] / v["SQ_ACTIVE_INST_VALU"],
              "note": f"k_tile fast variant; busy = SQ_ACTIVE_INST_VALU*4 / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs); passes {new}_sq_counters.txt"}
json.dump(d, open(path, "w"), indent=1)
b = json.load(open(os.path.join(P, f"{new}_bench.json")))
print("bench:", round(b["value"]), "Mpix/s", round(b["ms_per_step"], 4), "ms; kernel", round(b["roofline"]["kernel_ms"], 4), "ms; roofline", round(b["roofline"]["achieved"], 1),
      "GB/s", round(100 * b["roofline"]["frac"], 2), "%; other:", b.get("other_camera"), "check:", b.get("gathered_frame_equals_single_rank_frame"), "cpu:", round(b["cpu_baseline"]["value"], 2))
print("sq:", {kk: (round(vv, 3) if isinstance(vv, float) else vv) for kk, vv in d[k]["sq"].items() if kk != "note"})
print("k_tile KiB:", d[k]["k_tile"], "traffic", d[k]["hbm_bytes_per_launch"])
