#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (gpurun_out/pmc/*) per kernel: mean counter value per dispatch.
Applies the gfx950 correction of MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per
128-B request on wide coalesced reads -> x2; FETCH_SIZE/WRITE_SIZE are in KiB."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in sorted(acc.items()):
    if "orbhip" not in k:
        continue
    out[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    out[k]["dispatches"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in out[k]:
        out[k]["hbm_read_bytes_corrected"] = out[k]["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in out[k]:
        out[k]["hbm_write_bytes"] = out[k]["WRITE_SIZE"] * 1024
print(json.dumps(out, indent=1))
