"""csrc/field.hpp is host + device code: its two inversions beside the Fermat ladder (round 5: binary GCD, and the
batched-divstep form that d_pp's single inversion runs) are checked on the CPU against that ladder on all six fields
(tests/native/field_host_test.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CXX = "/opt/rocm/lib/llvm/bin/clang++"


def test_inverse_gcd_and_safegcd_equal_fermat_on_all_fields():
    if not os.path.exists(CXX):
        pytest.skip("ROCm host compiler not found (field.hpp uses clang's __builtin_addc / __builtin_subc)")
    out = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "field_host_test")
    r = subprocess.run([CXX, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "zk-saas_amd", "csrc"),
                        os.path.join(ROOT, "tests", "native", "field_host_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(": 0 mismatches") == 6, r.stdout
