"""The strip plan of k_normals3 / k_normals3s, checked on the CPU from the kernel's OWN code: tests/cpu/n3_plan_check.cpp
includes traversability_estimation_amd/csrc/te_n3_plan.h -- the header the kernels and their launch code are compiled
from -- and asserts that every cell of a region has exactly one owner, that closed-form blocks stay clear of the map frame
and that strips keep to their planned height (fixed cases of the BASELINE sizes + a 400-case sweep).  (Rounds 4-5 had a
Python restatement of the same arithmetic here: it could not fail when the kernel's arithmetic changed.)"""
import os
import subprocess

from tests.conftest import ROOT


def test_every_cell_of_a_region_has_one_owner(tmp_path):
    exe = tmp_path / "n3_plan_check"
    src = os.path.join(ROOT, "tests", "cpu", "n3_plan_check.cpp")
    inc = os.path.join(ROOT, "traversability_estimation_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", inc, src, "-o", str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failed checks" in r.stdout
