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
