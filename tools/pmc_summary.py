"""Sum rocprofv3 --pmc counter values per counter for the dispatches of one kernel family.

    python tools/pmc_summary.py <dir with *_counter_collection.csv (searched recursively)> [kernel substring]

FETCH_SIZE / WRITE_SIZE are reported in KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads (MI355X_MICROARCH.md, HBM section): multiply by 2 before comparing with a byte count.
"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "k_local_attn"
tot, disp = defaultdict(float), defaultdict(set)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"]
            i = name.find(pat)
            key = (r["Counter_Name"], name[i:].split("(")[0][:60])
            tot[key] += float(r["Counter_Value"])
            disp[key].add(r["Dispatch_Id"])
for (c, k), v in sorted(tot.items()):
    n = len(disp[(c, k)])
    print(f"{c:28s} {k:62s} total {v:.6g} over {n} dispatches = {v / n:.6g} per dispatch")
