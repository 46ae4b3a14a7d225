"""te_tie_triple.h on the CPU, from the header the tie marches are compiled from: the cells it names are the lattice points of
the circle for every whole-cell radius the kernels are instantiated for, and the split evaluation of isInside()'s test
(dx*dx per lane, dy*dy per row; an axis cell without its exactly-zero term) is bit-identical to the test in place."""
import os
import subprocess

from tests.conftest import ROOT


def test_circle_cells_and_the_split_test(tmp_path):
    exe = tmp_path / "tie_triple_check"
    src = os.path.join(ROOT, "tests", "cpu", "tie_triple_check.cpp")
    inc = os.path.join(ROOT, "traversability_estimation_amd", "csrc")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-Wall", "-Werror", "-I", inc, src, "-o", str(exe)], check=True, timeout=300)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "0 failed checks" in r.stdout
