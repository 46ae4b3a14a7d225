#!/usr/bin/env python3
"""Turns gpurun_out/r06_record/ (tools/r06_record.sh) into the committed record: profiles/r06_configs.json, r06_bench_line.json,
r06_cfg4_bench_line.json, r06_timeline_overlapped_launch.txt, r06_marker_api_stats.csv.  Run on the build box after the merge."""
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out", "r06_record")
P = os.path.join(ROOT, "profiles")


def last(f):
    t = open(f).read().strip().splitlines()
    return json.loads(t[-1]) if t else None


def main():
    rec = {}
    for f in sorted(glob.glob(O + "/bench_*.json")) + sorted(glob.glob(O + "/holes_*.json")):
        d = last(f)
        r = d["roofline"]
        name = os.path.basename(f)[:-5]
        e = {"ms_per_step": round(d["ms_per_step"], 5), "latency_ms_per_launch": round(d["latency_ms_per_launch"], 5), "value_cells_per_s": round(d["value"]),
             "n_gpus": d["n_gpus"], "ranks": d["ranks"], "roofline_frac": round(r["frac"], 4), "workload": d["config"]["workload"],
             "parity_ok": d.get("parity_check", {}).get("ok"), "parity_windows": d.get("parity_check", {}).get("windows"),
             "parity_max_abs_err": {k: v["max_abs_err"] for k, v in d.get("parity_check", {}).get("layers", {}).items()}}
        if "tick_mode" in d:
            e["tick_mode"] = {k: v for k, v in d["tick_mode"].items() if k != "what"}
            e["ticks_per_s"] = d["ticks_per_s"]
        if "cpu_baseline" in d:
            e["cpu_baseline_cells_per_s"] = round(d["cpu_baseline"]["value"])
        rec[name] = e
        print("%-26s ms/step %.4f  launch %.4f  %.3e cells/s  frac %.4f  parity %s %s" % (name, e["ms_per_step"], e["latency_ms_per_launch"], d["value"], e["roofline_frac"],
                                                                                     e["parity_ok"], e.get("tick_mode", "")))
    rec["defaults (tools/defaults_bench.py): the reference's default parameters, ms per launch"] = {
        k: {a: b["ms"] for a, b in v.items()} for k, v in json.load(open(O + "/defaults.json")).items()}
    for d in ("kt_defaults", "kt_cfg1", "kt_cfg2"):
        for f in glob.glob(O + "/" + d + "/**/*kernel_stats.csv", recursive=True):
            rec[d + ": us per kernel (rocprofv3 --kernel-trace --stats)"] = {
                re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Name"]).group(0): round(float(r["AverageNs"]) / 1e3, 1)
                for r in csv.DictReader(open(f)) if re.search(r"k_[a-z0-9_]+", r["Name"])}
    rec["what"] = ("tools/r06_record.sh on one MI355X box, final kernel sources (profiles/r06_hbm_traffic.json: kernel_sources_sha16): every BASELINE "
                   "configuration through bench.py --config, two gloo ranks through `bench.py --gpus 2` without a launcher (one device: the ranks share it), "
                   "the reference's YAML at 1024^2 / 4096^2 (res 0.05), hole maps (--holes f: speckle; 0.5 + f: unobserved rectangles), each with its parity "
                   "check (whole map up to 4096^2)")
    json.dump(rec, open(P + "/r06_configs.json", "w"), indent=1)
    shutil.copy(O + "/bench_driver.json", P + "/r06_bench_line.json")
    shutil.copy(O + "/bench_cfg4.json", P + "/r06_cfg4_bench_line.json")
    shutil.copy(O + "/timeline.txt", P + "/r06_timeline_overlapped_launch.txt")
    shutil.copy(O + "/marker/m_marker_api_stats.csv", P + "/r06_marker_api_stats.csv")
    d = last(O + "/bench_driver.json")
    r = d["roofline"]
    print({k: r[k] for k in ("frac", "traffic", "formulation_ceiling", "frac_of_ceiling", "ms_per_launch")}, d["ms_per_step"], d["value"])
    h = d.get("host_path", {})
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline_all_cores"]["value"], d["cpu_baseline_all_cores"]["cores"], "host", h.get("ms"), h.get("pinned_ms"),
          h.get("three_plugins_ms"), h.get("three_plugins_prefetch_ms"))


if __name__ == "__main__":
    main()
