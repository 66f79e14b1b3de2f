"""A short run of tools/fuzz_spmv.py: every SpMV kernel family forced in turn on irregular random structures
(empty rows, very long rows, crowded columns, rectangular shapes, sizes around the panel / block limits),
y = A x and y <- A x + y, against the oracle — bit for bit in the reference-order kernels, rounding level
(exact on integer data) in the order-free panels."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fuzz_spmv_kernels(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_spmv.py"), "3", "60"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "60 cases" in r.stdout and ", 0 failures" in r.stdout.splitlines()[-1]


def test_fuzz_spgemm_forms(gpu):
    """tools/fuzz_spgemm.py: hub columns, heavy / empty columns, crowded rows, rectangular shapes, real and complex
    values through the automatic, ordered, compacting, two-pass and split-key forms of mm — bit-identical to the
    oracle in every form"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_spgemm.py"), "4", "40"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "40 cases, 240 products, 0 failures" in r.stdout


def test_fuzz_assembly_operations(gpu):
    """tools/fuzz_assembly.py: compress, transpose, lin / + / - (real and complex), mulM, mulVT, takeDiag, kronecker,
    hcat / vcat / fromBlocks on irregular random inputs — every result bit-identical to the oracle"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_assembly.py"), "6", "60"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "60 cases" in r.stdout and ", 0 failures" in r.stdout.splitlines()[-1]
