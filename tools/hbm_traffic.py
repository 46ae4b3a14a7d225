#!/usr/bin/env python3
"""Summarise the HBM traffic of one te_run_chain launch from two rocprofv3 PMC passes.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE -d out/fetch -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path
    rocprofv3 --pmc WRITE_SIZE -d out/write -o p --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path
    python tools/hbm_traffic.py out/fetch out/write > profiles/rNN_hbm_traffic.json

FETCH_SIZE / WRITE_SIZE count KB.  The value per kernel is the average over its dispatches.
"""
import collections
import csv
import glob
import json
import re
import os
import sys


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            m = re.search(r"k_[a-z0-9_]+", r["Kernel_Name"])
            if m and m.group(0) != "k_count_invalid":  # (runs once when the elevation is uploaded, not in a te_run_chain launch)
                acc[m.group(0)].append(float(r["Counter_Value"]) * 1024.0)
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {k: {"FETCH_SIZE_bytes": int(fetch.get(k, 0)), "WRITE_SIZE_bytes": int(write.get(k, 0))}
               for k in sorted(set(fetch) | set(write))}
    fb = sum(v["FETCH_SIZE_bytes"] for v in kernels.values())
    wb = sum(v["WRITE_SIZE_bytes"] for v in kernels.values())
    cells = 4096 * 4096
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    print(json.dumps({
        "kernel_sources_sha16": bench.kernel_sources_sha16(),
        "command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-path "
                   "(two separate passes), summarised by tools/hbm_traffic.py",
        "config": "1 x 4096x4096, radius 9 cells, footprint pass",
        "unit": "bytes per te_run_chain launch",
        "kernels": kernels,
        "note": "FETCH_SIZE/WRITE_SIZE are in KB; loads are 4 B/lane (dword), for which the counter matched the byte "
                "count of a plain read within 15%, so no x2 correction (that applies to 16 B/lane streams) is used; "
                "kernel_sources_sha16 ties the numbers to the kernels they were measured with (bench.py checks it)",
        "fetch_bytes": fb, "write_bytes": wb, "traffic_bytes": fb + wb, "algorithmic_bytes": 24 * cells}, indent=1))


if __name__ == "__main__":
    main()
