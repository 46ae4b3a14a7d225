#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes per kernel: python tools/sq_counters.py <dir with one sub-directory per pass> > out.json

Every pass is a separate `rocprofv3 --pmc ...` run of the same command (tools/profile_round.sh); the value reported
for a kernel and a counter is the mean over that kernel's dispatches in the pass that collected the counter.  The
register / LDS / scratch columns come from the dispatch records themselves.  SQ_*_CYCLES, SQ_WAIT_* and SQ_ACTIVE_*
count quad-cycles summed over waves (MI355X_MICROARCH.md); FETCH_SIZE / WRITE_SIZE are KB.
"""
import collections
import csv
import glob
import json
import re
import sys


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", r["Kernel_Name"])
            if not m:
                continue
            k = m.group(0)
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"grid": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]),
                       "scratch_bytes": int(r["Scratch_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                       "sgpr": int(r["SGPR_Count"])}
    out = {}
    for k in sorted(acc):
        c = {n: sum(v) / len(v) for n, v in acc[k].items()}
        d = dict(meta[k])
        d["dispatches_seen"] = max(len(v) for v in acc[k].values())
        d["counters"] = {n: round(v, 1) for n, v in sorted(c.items())}
        waves = c.get("SQ_WAVES")
        derived = {}
        if waves:
            for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
                if n in c:
                    derived[n.replace("SQ_INSTS_", "insts_per_wave_").lower()] = round(c[n] / waves, 1)
        if c.get("SQ_WAVE_CYCLES"):
            wc = c["SQ_WAVE_CYCLES"]
            for n in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY",
                      "SQ_WAIT_ANY", "SQ_WAIT_INST_LDS"):
                if n in c:
                    derived[n.lower() + "_over_wave_cycles"] = round(c[n] / wc, 4)
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_WAVE_CYCLES"):
            derived["mean_resident_waves_per_busy_sq_cycle"] = round(c["SQ_WAVE_CYCLES"] / c["SQ_BUSY_CYCLES"], 2)
        d["derived"] = derived
        out[k] = d
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:  # ties the counters to the kernel sources they were measured with (bench.py, tools/valu_classes.py check it)
        import bench
        out["kernel_sources_sha16"] = bench.kernel_sources_sha16()
    except Exception:
        pass
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
