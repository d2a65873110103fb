#!/usr/bin/env python3
"""Condenses a scripts/collect_profiles.sh output directory into one JSON document:
per kernel the launch count and average duration (kernel trace) and the per-launch PMC
values, plus HBM traffic per launch with FETCH_SIZE calibrated on the exposure kernel,
whose read traffic is known exactly (12 bytes per pixel, float4 loads, nothing written)."""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main(out_dir):
    doc = {"kernels": {}}
    try:
        with open(os.path.join(out_dir, "bench_under_rocprof.json")) as f:
            bench = json.loads(f.read().strip().splitlines()[-1])
        doc["bench"] = {k: bench[k] for k in ("metric", "value", "unit", "ms_per_step", "config")}
        doc["bench"]["roofline_avg_launch_ms"] = bench["roofline"]["avg_launch_ms"]
    except (OSError, ValueError, KeyError, IndexError):
        pass
    for f in glob.glob(os.path.join(out_dir, "kt", "**", "*kernel_trace.csv"), recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in agg.items():
            doc["kernels"].setdefault(k, {})["launches"] = len(v)
            doc["kernels"][k]["avg_ms"] = sum(v) / len(v) / 1e6
            doc["kernels"][k]["min_ms"] = min(v) / 1e6
    for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
        for f in glob.glob(os.path.join(out_dir, sub, "**", "*counter_collection.csv"), recursive=True):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in agg.items():
                d = doc["kernels"].setdefault(k, {}).setdefault("pmc_per_launch", {})
                for c, v in cs.items():
                    d[c] = sum(v) / len(v)
    # calibrate FETCH_SIZE (KiB) on the exposure kernel: 1920x1080 pixels x 12 bytes
    exp = next((v for k, v in doc["kernels"].items() if k.startswith("exposure_kernel")), None)
    factor = None
    if exp and "FETCH_SIZE" in exp.get("pmc_per_launch", {}):
        factor = 1920 * 1080 * 12 / (exp["pmc_per_launch"]["FETCH_SIZE"] * 1024.0)
    doc["fetch_size_calibration"] = {
        "factor": factor,
        "note": "true bytes / (FETCH_SIZE * 1024) on exposure_kernel (known 12 B/pixel stream)"}
    for k, v in doc["kernels"].items():
        p = v.get("pmc_per_launch", {})
        if "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            v["hbm_read_bytes_per_launch"] = p["FETCH_SIZE"] * 1024.0 * (factor or 1.0)
            v["hbm_write_bytes_per_launch"] = p["WRITE_SIZE"] * 1024.0
    print(json.dumps(doc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main(sys.argv[1])
