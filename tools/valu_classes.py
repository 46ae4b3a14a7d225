#!/usr/bin/env python3
"""Prices the vector instructions of the bench launch's kernels BY CLASS (what `roofline.formulation_ceiling` is built from).

Round 5 priced every executed VALU instruction at 4.2 cycles per SIMD.  The microbenchmarks of profiles/r01_valu_rates.txt and
r03_valu_rates_new_ops.txt say otherwise for part of them: plain float32 add / mul / fma and 2-operand integer / logic
instructions issue once per 2.2-2.5 cycles with >= 4 waves per SIMD, float32 transcendentals once per 8.2, float64 ones once
per 16.2; everything else (fp64, 3-operand integer, min / max, compares, selects, conversions, packed, cross-lane) once per
4.2-4.6.  This script
  1. disassembles the gfx950 code objects of the built library's objects (llvm-objdump --offloading, no GPU needed),
  2. finds, per kernel, its hottest loop -- the backward branch that spans the most instructions: the row loop of the
     marching kernels, the tile loop of the mask kernel --, and counts its VALU instructions by class,
  3. takes the executed VALU instructions per launch from the SQ counter profile (profiles/rNN_sq_counters.json): the fp64
     ones as COUNTED (SQ_INSTS_VALU_{ADD,MUL,FMA}_F64 -- a loop that spans cold fp64 paths, like the mask kernel's
     checkForStep, would overstate them), the others split into the 4.2 / 2.2 / 8.2-cycle classes in the proportions of
     the loop's non-fp64 instructions,
and prints one JSON object: per kernel the static mix, the issue floor at 4.2 cycles for everything (round 5's model) and the
floor by class; the sums; the resulting ceilings of the 24 B/cell launch against 8 TB/s.

usage: python tools/valu_classes.py [profiles/rNN_sq_counters.json] > profiles/rNN_valu_classes.json
"""
import glob
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
N_SIMDS, CLOCK_HZ = 1024, 2.4e9
CYCLES = {"f64": 4.2, "slow32": 4.2, "fast32": 2.2, "trans32": 8.2, "trans64": 16.2}
# measured at <= 2.5 cycles per SIMD with >= 4 waves (profiles/r01_valu_rates.txt); anything not listed is priced at 4.2
FAST32 = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
          "v_and_b32", "v_or_b32", "v_mov_b32"}
# the kernels of one te_run_chain(TE_RUN_FOOTPRINT) on the bench map: (object file stem glob, symbol substring, name in the counter profile)
KERNELS = [("te_normals3.p*", "k_normals3s<81>", "k_normals3s<81>"),
           ("te_step5", "k_step_height5<81, false, 2>", "k_step_height5<81, false, 2>"),
           ("te_step5", "k_step_score5<81, false, 2>", "k_step_score5<81, false, 2>"),
           ("te_footprint", "k_fp_mask<32>", "k_fp_mask<32>"),
           ("te_footprint5.p*", "k_fp_slide5<81>", "k_fp_slide5<81>")]


def classify(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if not base.startswith("v_"):
        return None
    if base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "trans64" if "f64" in base else "trans32"
    if "f64" in base or base in ("v_lshl_add_u64", "v_mad_u64_u32", "v_mad_i64_i32"):
        return "f64"
    if base in FAST32 and not op.endswith(("_dpp", "_sdwa")):
        return "fast32"
    return "slow32"


def disassemble(obj, tmp):
    dst = os.path.join(tmp, os.path.basename(obj))
    shutil.copy(obj, dst)
    subprocess.run([OBJDUMP, "--offloading", dst], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    co = [f for f in glob.glob(dst + ".*") if "amdgcn" in f]
    if not co:
        return ""
    return subprocess.run([OBJDUMP, "-d", "--demangle", co[0]], check=True, capture_output=True, text=True).stdout


def kernel_body(text, key):
    """[(offset, mnemonic, branch target offset or None)] of the first kernel whose demangled symbol contains `key`."""
    lines = text.split("\n")
    start = None
    for n, l in enumerate(lines):
        if re.match(r"^[0-9a-f]{16} <", l) and key in l:
            start = n
            base = int(l.split()[0], 16)
            break
    if start is None:
        return None
    out = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]{16} <", l):
            break
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-F]+):", l) or re.match(r"^\s+(\S+)\s*//\s*([0-9A-F]+):", l)
        if not m:
            continue
        op, addr = m.group(1), int(m.group(2), 16)
        tgt = None
        if op.startswith(("s_cbranch", "s_branch")):
            t = re.search(r"\+0x([0-9a-f]+)>\s*$", l)
            if t:
                tgt = int(t.group(1), 16)
        out.append((addr - base, op, tgt))
    return out


def hottest_loop(body):
    best = None
    for k, (off, op, tgt) in enumerate(body):
        if tgt is not None and tgt <= off:
            first = next(i for i, b in enumerate(body) if b[0] >= tgt)
            if best is None or k - first > best[1] - best[0]:
                best = (first, k)
    return body[best[0]:best[1] + 1] if best else body


def kernel_sources_sha16():
    """As bench.py: identifies the kernel sources a profile belongs to."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "traversability_estimation_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def main():
    sq_path = sys.argv[1] if len(sys.argv) > 1 else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.json")))[-1]
    sq = json.load(open(sq_path))
    objdir = os.path.join(ROOT, "traversability_estimation_amd", "_build")
    out = {"what": __doc__.split("\n")[0], "cycles_per_instruction_per_simd": CYCLES, "fast32_opcodes": sorted(FAST32),
           "counters": os.path.relpath(sq_path, ROOT), "counters_kernel_sources_sha16": sq.get("kernel_sources_sha16"),
           "kernel_sources_sha16": kernel_sources_sha16(), "kernels": {}}
    tot_all, tot_cls = 0.0, 0.0
    with tempfile.TemporaryDirectory() as tmp:
        cache = {}
        for stem, key, cname in KERNELS:
            body = None
            for obj in sorted(glob.glob(os.path.join(objdir, stem + ".hip.o"))):
                if obj not in cache:
                    cache[obj] = disassemble(obj, tmp)
                body = kernel_body(cache[obj], key)
                if body:
                    break
            if not body:
                out["kernels"][cname] = {"error": "kernel not found in the built objects"}
                continue
            loop = hottest_loop(body)
            mix = {c: 0 for c in CYCLES}
            for _, op, _ in loop:
                c = classify(op)
                if c:
                    mix[c] += 1
            nv = sum(mix.values())
            rec = {"kernel_instructions": len(body), "hottest_loop_instructions": len(loop), "hottest_loop_valu": nv,
                   "hottest_loop_valu_by_class": mix}
            ctr = sq.get(cname, {}).get("counters", {})
            V = ctr.get("SQ_INSTS_VALU")
            if V and nv:
                f64_dyn = sum(ctr.get(n, 0.0) for n in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64"))
                rec["valu_per_launch"] = V
                rec["fp64_fraction_static_vs_counters"] = [round(mix["f64"] / nv, 4), round(f64_dyn / V, 4)]
                rest_static = mix["slow32"] + mix["fast32"] + mix["trans32"]
                rest = V - f64_dyn
                cyc = f64_dyn * CYCLES["f64"] + sum(rest * mix[c] / max(rest_static, 1) * CYCLES[c] for c in ("slow32", "fast32", "trans32"))
                rec["valu_per_launch_by_class"] = {"f64 (counted)": round(f64_dyn), **{c: round(rest * mix[c] / max(rest_static, 1)) for c in ("slow32", "fast32", "trans32")}}
                rec["issue_floor_us_all_at_4.2"] = round(V * 4.2 / (N_SIMDS * CLOCK_HZ) * 1e6, 1)
                rec["issue_floor_us_by_class"] = round(cyc / (N_SIMDS * CLOCK_HZ) * 1e6, 1)
                tot_all += rec["issue_floor_us_all_at_4.2"]
                tot_cls += rec["issue_floor_us_by_class"]
            out["kernels"][cname] = rec
    bytes_launch = 4096 * 4096 * 24
    out["sum_issue_floor_us_all_at_4.2"] = round(tot_all, 1)
    out["sum_issue_floor_us_by_class"] = round(tot_cls, 1)
    if tot_cls > 0:
        out["formulation_ceiling_all_at_4.2"] = round(bytes_launch / (tot_all * 1e-6) / 8e12, 4)
        out["formulation_ceiling_by_class"] = round(bytes_launch / (tot_cls * 1e-6) / 8e12, 4)
    out["reading"] = ("the by-class floor assumes >= 4 waves per SIMD for the 2.2-cycle class (k_normals3s runs 3: its fast32 share then "
                      "issues at the 4.2-cycle rate, r01_valu_rates.txt column 8ch,2w), so it is a LOWER bound of the time and the ceiling an "
                      "UPPER bound of what the formulation can reach")
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
