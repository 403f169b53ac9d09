#!/usr/bin/env python3
"""Print per-kernel averages of every counter in rocprofv3 --pmc counter_collection.csv files (SQ utilisation passes)."""
import collections, csv, sys
for path in sys.argv[1:]:
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
        agg[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    print(path)
    for (kern, c), v in sorted(agg.items()):
        if "k_tile" in kern or "k_block_setup" in kern or "k_resolve" in kern: print(f"  {kern:22s} {c:28s} {v/len(n[(kern,c)]):16.0f} per launch ({len(n[(kern,c)])} launches)")
