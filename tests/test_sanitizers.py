"""ThreadSanitizer and AddressSanitizer + UndefinedBehaviorSanitizer over the library's multi-threaded HOST code (SURVEY.md
section 5): csrc/hostpool.hpp (the context's worker pool) and csrc/net.hpp (the shm control plane and the
shared-memory data plane of the star network, host mode) compiled from the product headers into tests/native/net_stress
with the host compiler and run on the CPU -- worlds 2 / 4 / 8, three channel threads per rank.  GPU sanitizers are not
available on the pool (gpurun refuses them); the device side is covered by the parity tests."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "native", "net_stress.cpp")
CXX = "/opt/rocm/lib/llvm/bin/clang++"
OUT = os.path.join(ROOT, "tests", "native", "_build")


def _build(kind):
    if not os.path.exists(CXX) or not os.path.isdir("/opt/rocm/include/hip"):
        pytest.skip("ROCm host compiler / HIP headers not found")
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, "net_stress_" + kind)
    flags = {"tsan": ["-fsanitize=thread"], "asan": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]}[kind]
    newest = max(os.path.getmtime(p) for p in (SRC, os.path.join(ROOT, "zk-saas_amd", "csrc", "net.hpp"),
                                                os.path.join(ROOT, "zk-saas_amd", "csrc", "hostpool.hpp")))
    if not os.path.exists(exe) or os.path.getmtime(exe) < newest:
        cmd = [CXX, "-O1", "-g", "-std=c++17", "-fno-omit-frame-pointer"] + flags + [
            "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "zk-saas_amd", "csrc"), SRC, "-o", exe,
            "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-lrt", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return exe


def _run(exe, args, kind):
    env = dict(os.environ)
    env["TSAN_OPTIONS"] = "halt_on_error=1 exitcode=66 second_deadlock_stack=1"
    env["ASAN_OPTIONS"] = "detect_leaks=0 exitcode=67"          # (libamdhip64's own start-up allocations are not ours to free)
    env["UBSAN_OPTIONS"] = "print_stacktrace=1 halt_on_error=1"
    r = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, "%s %s: exit %d\n%s" % (kind, args, r.returncode, (r.stderr or r.stdout)[-4000:])


@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_host_pool_under_sanitizers(kind):
    _run(_build(kind), ["pool"], kind)


@pytest.mark.parametrize("workers", [1, 2, 4])
@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_prover_gate_structure_cannot_block(kind, workers):
    """The batched prover's launch / gate / fold tasks (device taken out) with three batches in flight on pools of 1, 2
    and 4 workers: every bounded wait returns, every task runs (no blocking waits that can starve)."""
    _run(_build(kind), ["gates", str(workers), "40"], kind)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_rccl_code_path_call_sequence_with_stubbed_send_recv(kind, world):
    """csrc/net.hpp's RCCL branches (ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd per gather, scatter, all-to-all)
    driven at worlds 2 / 4 / 8 with the two verbs replaced by shared-memory mailboxes: every byte arrives where the SHM
    transport puts it, and every rank's call sequence is what the rounds imply.  The real thing needs a GPU per rank."""
    _run(_build(kind), ["rccl", str(world), "6"], kind)


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("kind", ["tsan", "asan"])
def test_star_network_host_mode_under_sanitizers(kind, world):
    _run(_build(kind), ["net", str(world), "6"], kind)
