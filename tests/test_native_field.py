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


def test_glv_split_and_joint_sparse_form_recombine():
    """csrc/glv.hpp (the dealer's point packing, groth16/src/proving_key.rs:72-86 at l = 2): k = k1 + lambda k2 (mod r) with both
    parts below 2^Glv::BITS, checked with Python integers against the LAMBDA of csrc/glv_params.hpp (itself checked to be a
    primitive cube root of unity of Fr), and the joint sparse form of (|k1|, |k2|): digits in {-1, 0, 1} that sum back to
    the two magnitudes, at most BITS + 1 columns.  300 scalars per curve incl. 0, 1, ... and r - 1."""
    import re
    if not os.path.exists(CXX):
        pytest.skip("ROCm host compiler not found")
    out = os.path.join(ROOT, "tests", "native", "_build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "glv_host_test")
    r = subprocess.run([CXX, "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "zk-saas_amd", "csrc"),
                        os.path.join(ROOT, "tests", "native", "glv_host_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-500:] + r.stderr
    hdr = open(os.path.join(ROOT, "zk-saas_amd", "csrc", "glv_params.hpp")).read()
    from oracle.params import CURVES
    consts = {}
    for name, struct in (("bn254", "Bn254Fr"), ("bls12_381", "Bls381Fr"), ("bls12_377", "Bls377Fr")):
        body = hdr[hdr.index("struct Glv<%s>" % struct):]
        limbs = re.search(r"LAMBDA\[8\] = \{([^}]*)\}", body).group(1)
        lam = sum(int(x.strip().rstrip("u"), 16) << (32 * i) for i, x in enumerate(limbs.split(",")))
        bits = int(re.search(r"BITS = (\d+)", body).group(1))
        rmod = CURVES[name].r
        assert (lam * lam + lam + 1) % rmod == 0 and lam != 1            # a primitive cube root of unity of Fr
        consts[name] = (lam, bits, rmod)
    seen = {k: 0 for k in consts}
    for ln in r.stdout.splitlines():
        f = ln.split()
        if len(f) != 8:
            continue
        lam, bits, rmod = consts[f[0]]
        k, n1, m1, n2, m2 = int(f[1], 16), int(f[2]), int(f[3], 16), int(f[4]), int(f[5], 16)
        k1, k2 = (-m1 if n1 else m1), (-m2 if n2 else m2)
        assert (k1 + lam * k2 - k) % rmod == 0, ln
        assert m1 < (1 << bits) and m2 < (1 << bits), ln
        val = lambda d: sum((1 if c == "+" else -1 if c == "-" else 0) << i for i, c in enumerate(d)) if d != "_" else 0
        assert val(f[6]) == m1 and val(f[7]) == m2, ln
        assert len(f[6]) <= bits + 1
        seen[f[0]] += 1
    assert all(v == 300 for v in seen.values()), seen
