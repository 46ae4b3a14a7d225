#!/usr/bin/env python3
"""What-if A of round 4's review, the part that needs no GPU: WOULD a float32 slide be accurate enough?

k_normals3s slides four z-moments per lane in double (Sz, Siz, Sjz, Szz of dz = z - zref; te_normals3.hip).  The review
asked what a float32 ring with float32 moments (v_pk_add_f32 / v_pk_fma_f32: half the VALU issue slots) would buy, with
an exact fallback for the lanes that need it.  This script replays the kernel's slide on the bench map (4096^2, seed 1235,
R = 9, strips of 87 rows as the launcher cuts them, one reference height per 64-lane block) twice -- float64 and float32,
every float32 operation rounded where the instruction would round (fma = one rounding) -- runs the SAME float64 tail on
both sets of moments and reports the score differences.  The float32 slide is given every benefit: exact dz staging in
float32 where representable, fused multiply-adds, a float64 tail.

Output: one JSON line (profiles/r05_whatif_f32_slide.json keeps it).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from traversability_estimation_amd import synth  # noqa: E402

R, Q = 9, 81
HW = [int(np.floor(np.sqrt(Q - d * d))) for d in range(R + 1)]  # half-height of column |e| = d
f32, f64 = np.float32, np.float64


def fma32(a, b, c):
    """float32 fma: exact product and sum in float64 (24 + 24 bits < 53), one rounding"""
    return (a.astype(f64) * b.astype(f64) + c.astype(f64)).astype(f32)


def direct_moments(dz, j, cols, T):
    """moments of the disc of row j summed directly (strip start), in dtype T; dz: [rows][W] window of the block"""
    Sz = np.zeros(cols.size, T)
    Siz = np.zeros(cols.size, T)
    Sjz = np.zeros(cols.size, T)
    Szz = np.zeros(cols.size, T)
    for e in range(-R, R + 1):
        h = HW[abs(e)]
        for dj in range(-h, h + 1):
            z = dz[j + dj, cols + e].astype(T)
            Sz = Sz + z
            Siz = Siz + T(e) * z
            Sjz = Sjz + T(dj) * z
            Szz = Szz + z * z
    return Sz, Siz, Sjz, Szz


def slide(dz, j, cols, S, T):
    """one row step of te_normals3.hip's slide() in dtype T (float32: every instruction rounds once)"""
    Sz, Siz, Sjz, Szz = S
    Sz0 = Sz
    sv = {}
    for e in range(-R, R + 1):
        h = HW[abs(e)]
        zl = dz[j + 1 + h, cols + e].astype(T)
        zt = dz[j - h, cols + e].astype(T)
        u, v = zl - zt, zl + zt
        Sz = Sz + u
        if T is f32:
            if e != 0:
                Siz = fma32(np.full_like(u, e), u, Siz)
            Szz = fma32(u, v, Szz)
        else:
            Siz = Siz + T(e) * u
            Szz = Szz + u * v
        sv[h] = v if h not in sv else sv[h] + v
    acc = Sjz
    for h, s in sv.items():
        acc = fma32(np.full_like(s, h + 0.5), s, acc) if T is f32 else acc + T(h + 0.5) * s
    Sjz = fma32(np.full_like(acc, -0.5), Sz0 + Sz, acc) if T is f32 else acc - T(0.5) * (Sz0 + Sz)
    return Sz, Siz, Sjz, Szz


def tail(S, res, N, sii, slope_crit, rough_crit):
    """closed-form smallest eigenpair, float64 (the kernel's tail(), without its float32 shortcuts)"""
    Sz, Siz, Sjz, Szz = (np.asarray(x, f64) for x in S)
    D = N * Szz - Sz * Sz
    K1h = 0.5 * N * res * res * sii
    dl = K1h - 0.5 * D
    h2 = (N * res) ** 2 * (Siz * Siz + Sjz * Sjz)
    s = np.sqrt(dl * dl + h2)
    t = dl + s
    m2 = h2 / (s * t)
    nz = np.sqrt(np.maximum(1.0 - 0.5 * m2, 0.0)).astype(f32).astype(f64)  # the reference stores float32 normals
    slope = np.arccos(np.clip(nz, -1, 1))
    lam = np.maximum((0.5 * D + K1h) - s, 0.0)
    rough = np.sqrt(lam / (N * (N - 1.0)))
    return np.maximum(1.0 - slope / slope_crit, 0.0), np.maximum(1.0 - rough / rough_crit, 0.0)


def main():
    n, res = 4096, 0.05
    strip = int(os.environ.get("STRIP_ROWS", "87"))
    nblocks = int(os.environ.get("BLOCKS", "16"))  # 64-lane block columns sampled
    nstrips = int(os.environ.get("STRIPS", "3"))
    elev = synth.perlin_elevation(n, n, seed=1235)  # [col j][row i]
    N = sum(2 * h + 1 for h in [HW[abs(e)] for e in range(-R, R + 1)])
    sii = sum(e * e * (2 * HW[abs(e)] + 1) for e in range(-R, R + 1))
    rng = np.random.default_rng(5)
    err_s, err_r, cells = [], [], 0
    drift = []
    for _ in range(nstrips):
        js = int(rng.integers(R + 1, n - strip - R - 2))
        for _b in range(nblocks):
            i0 = 64 * int(rng.integers(1, n // 64 - 1))
            win = elev[js - R - 1: js + strip + R + 2, i0 - R: i0 + 64 + R].astype(f64)  # rows js-R-1 .., window columns
            zref = f64(elev[js, i0])  # (the kernel: the first finite cell of the strip's first row)
            dz64 = win - zref
            dz32 = dz64.astype(f32)  # what a float32 ring would hold
            cols = np.arange(64) + R
            j = R + 1  # index of row js in `win`
            S64 = direct_moments(dz64, j, cols, f64)
            S32 = direct_moments(dz32, j, cols, f32)
            for k in range(strip):
                a_s, a_r = tail(S64, res, N, sii, 1.0, 0.05)
                b_s, b_r = tail(S32, res, N, sii, 1.0, 0.05)
                err_s.append(np.abs(a_s - b_s))
                err_r.append(np.abs(a_r - b_r))
                cells += 64
                if k + 1 < strip:
                    S64 = slide(dz64, j + k, cols, S64, f64)
                    S32 = slide(dz32, j + k, cols, S32, f32)
            # float32 direct sums of the LAST row against its slid moments: what the slide itself drifted
            Sd = direct_moments(dz32, j + strip - 1, cols, f32)
            drift.append(float(np.max(np.abs(np.asarray(Sd[3], f64) - np.asarray(S32[3], f64)) / np.maximum(np.abs(np.asarray(Sd[3], f64)), 1e-30))))
    es, er = np.concatenate(err_s), np.concatenate(err_r)
    out = {
        "what": "float32 slide of k_normals3s's four z-moments against the float64 slide, same float64 tail; bench map 4096^2 seed 1235, R = 9",
        "cells": int(cells), "strip_rows": strip,
        "slope_score": {"max_abs_diff": float(es.max()), "p99": float(np.percentile(es, 99)), "median": float(np.median(es)),
                        "frac_above_1e-5": float((es > 1e-5).mean()), "frac_above_2e-6": float((es > 2e-6).mean())},
        "roughness_score": {"max_abs_diff": float(er.max()), "p99": float(np.percentile(er, 99)), "median": float(np.median(er)),
                            "frac_above_1e-5": float((er > 1e-5).mean()), "frac_above_2e-6": float((er > 2e-6).mean())},
        "szz_relative_drift_over_a_strip_max": max(drift),
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main()
