"""CPU test of the executor's prime-field arithmetic (csrc/bigfield.h, host only): the fast modular inverse the curve
precompiles use (binary GCD on 62-bit approximations) equals Fermat's y^(p-2) for both fields."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_inverse_equals_fermat(tmp_path):
    exe = str(tmp_path / "inv_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "dvt_circuits_amd", "csrc"),
                           os.path.join(ROOT, "tools", "microbench", "bigfield_inv_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bls: 0 mismatches" in r.stdout and "secp: 0 mismatches" in r.stdout and "BUG" not in r.stdout, r.stdout


def test_witness_solver_long_division(tmp_path):
    """csrc/polyrel.h poly_divmnu (the quotient of a UINT256_MUL row, on the host and on the GPU) against Python integers"""
    exe = str(tmp_path / "div_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "dvt_circuits_amd", "csrc"),
                           os.path.join(ROOT, "tools", "microbench", "polyrel_div_check.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")[:-1]
    assert len(lines) > 20000
    for ln in lines:
        u, v, q, rem = (int(x, 16) for x in ln.split())
        assert u == q * v + rem and rem < v, ln
