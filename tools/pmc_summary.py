#!/usr/bin/env python
"""Summarise rocprofv3 --pmc passes (one directory per pass, csv output) into a JSON under profiles/.

usage: pmc_summary.py OUT.json NOTE DIR [DIR ...]
Per kernel: the average of each counter over its launches, and the HBM bytes per launch derived from FETCH_SIZE /
WRITE_SIZE exactly as MI355X_MICROARCH.md prescribes for gfx950: both are in KiB, FETCH_SIZE counts 64 B per 128-B
request of a wide streaming read and is doubled, WRITE_SIZE is exact.
The file is stamped with the sha256 of the device sources (bench.source_hash): bench.py quotes `roofline.traffic` from it
only while that hash equals the sources it runs.  `step_hbm_bytes` = sum over the kernels of one CoupledModel step of
(average bytes per launch) x (launches per step: 4 of every fused kernel, 12 forward and 8 inverse A sub-passes)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

out, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void nq::", "").strip()
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"note": note, "source_sha256": bench.source_hash(), "kernels": {}}
for k, cs in sorted(acc.items()):
    e = {}
    for c, v in cs.items():
        e[c] = sum(v) / len(v)
        e["launches_sampled"] = len(v)
    if "FETCH_SIZE" in e:
        e["hbm_read_bytes"] = e["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in e:
        e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024
    if "hbm_read_bytes" in e and "hbm_write_bytes" in e:
        e["hbm_bytes_per_launch"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    res["kernels"][k] = e
PER_STEP = [("k_x_products<", 4), ("k_x_wavepv", 4), ("k_s_q<", 4), ("k_s_phi<", 4), ("k_s_invert<", 4),
            ("k_budget_sums", 1), ("k_budget_accumulate", 1)]
step = 0.0
for prefix, n in PER_STEP:
    for k, e in res["kernels"].items():
        if k.startswith(prefix) and "hbm_bytes_per_launch" in e:
            step += n * e["hbm_bytes_per_launch"]
for k, e in res["kernels"].items():
    if k.startswith("k_y_A<") and "hbm_bytes_per_launch" in e:
        step += (8 if "true" in k else 12) * e["hbm_bytes_per_launch"]
res["step_hbm_bytes"] = step
json.dump(res, open(out, "w"), indent=1)
print("wrote", out, len(res["kernels"]), "kernels; step_hbm_bytes = %.3f GB" % (step / 1e9))
