"""CPU-side checks of the product: the C-ABI library loads and exports every declared symbol; host codecs."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "zksaas.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import ctypes
    import zksaas_amd as zk
    if not os.path.exists(zk.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    lib = ctypes.CDLL(zk.LIB_PATH)
    declared = _declared_symbols()
    assert declared, "no symbols parsed from the header"
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(zk.SYMBOLS) == declared      # the Python binding covers the whole header


def test_profile_slots_are_named_and_bench_knows_the_kernel_ones():
    """zk_profile_name needs no context: every slot has a distinct name; the kernel slots bench.py prices are among them
    and the host spans of zk_groth16_prove carry the "host:" prefix that keeps them out of the roofline choice."""
    import ctypes
    import zksaas_amd as zk
    import bench
    lib = ctypes.CDLL(zk.LIB_PATH)
    lib.zk_profile_name.restype = ctypes.c_char_p
    names = [lib.zk_profile_name(i).decode() for i in range(lib.zk_profile_slots())]
    assert all(names) and len(set(names)) == len(names)
    assert lib.zk_profile_name(len(names)).decode() == "" and lib.zk_profile_name(-1).decode() == ""
    kernel_slots = [n for n in names if not n.startswith("host:")]
    assert set(bench.SLOT_BYTES) <= set(kernel_slots)
    assert {"host:prove_launch", "host:prove_wait", "host:prove_tail"} <= set(names)
    prof = [{"kernel": n, "total_ms": 9.0 if n.startswith("host:") else 1.0, "units": 1.0, "launches": 1} for n in names]
    assert not bench.roofline_of(prof, ntt_passes=2, masks_on=False)["kernel"].startswith("host:")


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    import zksaas_amd as zk
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zk.ZkError) as e:
        zk.Context("bn254", 2)
    assert e.value.code == 3


def test_load_then_import_torch_exits_cleanly():
    """loading the library (which shares torch's HIP runtime) and importing torch afterwards must not abort
    the interpreter at exit (a global preload of torch's librccl did: 'double free or corruption', exit 134)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import zksaas_amd as zk; zk.load(); "
            "from zksaas_amd import net; net.StarNet.unique_id(); import torch; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stderr[-400:])


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "zk-saas_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle/" not in src.replace(
                    "oracle/dist.py", "").replace("oracle/prng.py", ""), f


def test_mont_codec_roundtrip():
    import zksaas_amd as zk
    from oracle.params import CURVES
    for name, c in CURVES.items():
        assert zk.fields.FR[name] == c.r and zk.fields.FQ[name] == c.q and zk.fields.FR_GENERATOR[name] == c.r_gen
        codec = zk.fields.MontCodec(c.r)
        vals = [0, 1, c.r - 1, 123456789 ** 3 % c.r]
        assert codec.decode(codec.encode(vals)) == vals
        # Montgomery one = R mod p
        assert int(codec.encode([1])[0][0]) == ((1 << (64 * codec.nl)) % c.r) & ((1 << 64) - 1)


def test_chacha20_block_known_answer():
    """RFC 7539 section 2.3.2 test vector through the library's block function (the share-randomness stream)."""
    import ctypes as C
    import struct
    import zksaas_amd as zk
    lib = zk.load()
    key = (C.c_uint32 * 8)(*struct.unpack("<8I", bytes(range(32))))
    out = (C.c_uint32 * 16)()
    # words 12..15 = (counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00)
    lib.zk_chacha20_block(key, 1 | (0x09000000 << 32), 0x4A000000, out)
    assert struct.pack("<16I", *out).hex() == (
        "10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
        "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")
    # and it is a function of every input word
    ref = list(out)
    lib.zk_chacha20_block(key, 2 | (0x09000000 << 32), 0x4A000000, out)
    assert list(out) != ref


def test_bench_line_survives_a_stuck_auxiliary_leg():
    """bench.AuxGuard: when the legs after the timed K steps do not return, the line is printed ONCE with what is finished,
    `incomplete` and `aux_timeout` (naming the leg), and the process leaves with a NON-ZERO status -- a leg that hung is
    not a pass -- while a leg that finishes in time leaves no trace; results published through `put` while the
    timer fires are either whole in the line or absent (the lock), never a half-mutated dict."""
    import json
    import subprocess
    import sys
    code = ("import sys, time; sys.path.insert(0, %r); import bench\n"
            "res = {'metric': 'm', 'value': 1.5}\n"
            "g = bench.AuxGuard(res, 0.3); g.leg = 'batched 8 x 2 in flight'\n"
            "t0 = time.time()\n"
            "while time.time() - t0 < float(sys.argv[1]): g.put('batched', {'k': list(range(50))}, append=True)\n"
            "res['late'] = True\n"
            "if g.done(): print(__import__('json').dumps(res))\n" % ROOT)
    stuck = subprocess.run([sys.executable, "-c", code, "5"], capture_output=True, text=True, timeout=60)
    assert stuck.returncode == 3, (stuck.returncode, stuck.stderr)
    lines = [ln for ln in stuck.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["value"] == 1.5 and line["incomplete"] is True and "late" not in line
    assert line["aux_timeout"]["leg"] == "batched 8 x 2 in flight" and all(len(b["k"]) == 50 for b in line["batched"])
    fine = subprocess.run([sys.executable, "-c", code, "0"], capture_output=True, text=True, timeout=60)
    lines = [ln for ln in fine.stdout.strip().splitlines() if ln.startswith("{")]
    line = json.loads(lines[-1])
    assert fine.returncode == 0 and len(lines) == 1 and line.get("late") is True and "aux_timeout" not in line and "incomplete" not in line


def test_host_cores_honours_quota_and_affinity():
    import bench
    usable, host, quota = bench.host_cores()
    assert 1 <= usable <= host == (os.cpu_count() or 1)
    assert quota is None or (quota > 0 and usable <= int(quota + 0.999))
