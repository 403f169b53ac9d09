#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as the MI355X guide prescribes) into
profiles/pmc_traffic.json, the file bench.py reads for roofline.traffic.

usage: pmc_summary.py KEY FETCH_counter_collection.csv WRITE_counter_collection.csv [tag]
  KEY = "<W>x<H>_g<grid>_<camera>_n<gpus>"  (bench.py's lookup key)

Units / corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB.  WRITE_SIZE is exact for our
4-byte-per-lane stores (calibrated in this very trace: k_axis_tables writes 5*4096*4 B = 80 KiB -> 80; k_block_boxes
7168 KiB -> 7184).  FETCH_SIZE reports 1/2 of the bytes of wide coalesced streams on gfx950, so it is doubled; our
reads are 4 B/lane (uncalibrated), which makes the doubled figure an upper bound -- both values are recorded.
"""
import collections, csv, json, os, sys

def per_kernel(path, counter):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = (int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].replace("void ", ""))
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    by = collections.defaultdict(list)
    for (d, name), v in agg.items():
        by[name].append(v)
    return {k: sum(v[-3:]) / len(v[-3:]) for k, v in by.items()}   # average of the last launches

key, fcsv, wcsv = sys.argv[1:4]
tag = sys.argv[4] if len(sys.argv) > 4 else ""
f, w = per_kernel(fcsv, "FETCH_SIZE"), per_kernel(wcsv, "WRITE_SIZE")
def fold(d):
    """the tile kernel runs as two launches (fast variant, then the complete variant for items it handed over): one entry"""
    out = {}
    for k, v in d.items():
        out["vf::k_tile" if k.startswith("vf::k_tile<false") else k] = out.get("vf::k_tile" if k.startswith("vf::k_tile<false") else k, 0) + v
    return out
f, w = fold(f), fold(w)
frame = ["vf::k_block_boxes", "vf::k_plan", "vf::k_plan_sort", "vf::k_clear", "vf::k_tile"]
fetch_kib = sum(f.get(k, 0) for k in frame)
write_kib = sum(w.get(k, 0) for k in frame)
tile_f, tile_w = f.get("vf::k_tile", 0), w.get("vf::k_tile", 0)
entry = {
    "tag": tag,
    "k_tile": {"FETCH_SIZE_KiB": tile_f, "WRITE_SIZE_KiB": tile_w},
    "frame_kernels": {k: {"FETCH_SIZE_KiB": f.get(k, 0), "WRITE_SIZE_KiB": w.get(k, 0)} for k in frame},
    "hbm_bytes_per_launch": int((2 * tile_f + tile_w) * 1024),            # k_tile, FETCH doubled (guide's gfx950 correction)
    "hbm_bytes_per_launch_fetch_uncorrected": int((tile_f + tile_w) * 1024),
    "hbm_bytes_per_frame_all_kernels": int((2 * fetch_kib + write_kib) * 1024),
}
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "pmc_traffic.json")
data = json.load(open(out)) if os.path.exists(out) else {}
if key in data and "sq" in data[key]:
    entry["sq"] = data[key]["sq"]          # SQ utilisation block is maintained separately (r01_j_sq_counters.txt)
data[key] = entry
json.dump(data, open(out, "w"), indent=1)
print(key, json.dumps(entry))
