"""ctypes binding of libzksaas_hip.so (include/zksaas.h).  No CPU fallback: if the library is missing the
import of anything that computes fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The one library this package loads.  Nothing in the environment can point it elsewhere; same-box A/B runs
# against another build go through tools/ab_run.py, which sets this attribute before anything is loaded.
LIB_PATH = os.path.join(_HERE, "libzksaas_hip.so")

# every symbol include/zksaas.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "zk_version", "zk_ctx_create", "zk_ctx_destroy", "zk_last_error", "zk_ctx_n", "zk_ctx_l", "zk_fr_bytes",
    "zk_fq_bytes", "zk_malloc", "zk_free", "zk_memcpy_h2d", "zk_memcpy_d2h", "zk_stream_sync", "zk_pss_pack",
    "zk_pss_det_pack", "zk_pss_unpack", "zk_pss_unpack2", "zk_bitrev", "zk_vec_add", "zk_vec_mul_sub", "zk_fft1",
    "zk_fft2_king", "zk_d_fft", "zk_d_ifft", "zk_fft_mask_sample", "zk_deg_red", "zk_degred_mask_sample", "zk_d_pp",
    "zk_msm", "zk_d_msm", "zk_base_mul", "zk_circom_h", "zk_groth16_prove", "zk_profile_enable",
    "zk_profile_slots", "zk_profile_name", "zk_profile_read", "zk_d_msm_local", "zk_group_add", "zk_groth16_assemble",
    "zk_groth16_msms_begin", "zk_groth16_msms_finish", "zk_groth16_prove_async", "zk_groth16_wait", "zk_groth16_abort",
    "zk_net_unique_id", "zk_net_create", "zk_net_destroy", "zk_net_last_error", "zk_net_set_timeout_ms", "zk_net_info",
    "zk_net_enter", "zk_net_gather", "zk_net_scatter", "zk_net_alltoall", "zk_net_stats", "zk_net_gather_host", "zk_net_bcast_host", "zk_net_sync",
    "zk_dist_d_fft", "zk_dist_d_ifft", "zk_dist_deg_red", "zk_dist_d_pp", "zk_dist_d_msm", "zk_dist_circom_h",
    "zk_dist_groth16_prove", "zk_chacha20_block", "zk_deg_red_points", "zk_degred_mask_sample_points",
    "zk_points_decompress", "zk_points_compress", "zk_libsnark_h", "zk_vec_scale", "zk_deg_red_parties", "zk_d_msm_parties", "zk_pss_pack_points", "zk_msm_plan",
    "zk_msm_mask_sample", "zk_r1cs_qap", "zk_fr_to_bytes", "zk_fr_from_bytes", "zk_ctx_set_option", "zk_msm_precompute", "zk_msm_forget", "zk_msm_table_info",
    "zk_groth16_prove_batch", "zk_groth16_prove_batch_async", "zk_groth16_batch_wait", "zk_msm_batch", "zk_pss_unpack_points", "zk_pss_unpack2_points", "zk_groth16_reconstruct", "zk_msm_stats", "zk_dist_groth16_prove_batch", "zk_d_fft_host", "zk_msm_host", "zk_d_msm_host",
    "zk_net_parties", "zk_dist_groth16_prove_async", "zk_dist_groth16_wait", "zk_fq_selftest",
    "zk_dist_deg_red_points", "zk_dist_libsnark_h", "zk_deg_red_host", "zk_d_pp_host", "zk_circom_h_host",
    "zk_groth16_prove_host",
]

_lib = None


class ZkError(RuntimeError):
    """MpcNetError mirror (mpc-net/src/lib.rs:19-24): code 1 Generic, 2 Protocol, 3 NotConnected, 4 BadInput."""

    def __init__(self, code, msg, party=-1):
        super().__init__("zk status %d: %s%s" % (code, msg, (" (party %d)" % party) if party >= 0 else ""))
        self.code, self.msg, self.party = code, msg, party


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 (SONAME libamdhip64.so.7, same as /opt/rocm's).  Two HIP
    runtimes in one process do not coexist reliably (the second one to initialise may see no GPU), and bench.py /
    multigpu.py hand torch tensors to this library, so when torch is installed its runtime is loaded first and the
    dynamic loader then resolves this library's libamdhip64.so.7 dependency to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    # Only the HIP runtime.  RCCL is NOT preloaded: net.hpp's Rccl::load finds torch's copy with RTLD_NOLOAD once torch
    # is imported and dlopens it lazily otherwise, and only for the RCCL transport -- a global preload of torch's
    # librccl before `import torch` made the process abort at exit on a box without a GPU (double free in the two
    # copies' static destructors; tests/test_host.py::test_load_then_import_torch_exits_cleanly).
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libzksaas_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                          "there is no CPU fallback")
    # (the library sets this itself when it is loaded; here as well because the HIP runtime preloaded just below reads its
    # environment once -- api.cpp zk_runtime_defaults says what it buys)
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    missing = [name for name in SYMBOLS if not hasattr(lib, name)]
    if missing:                       # an older or foreign build: refuse it rather than call into a different ABI
        raise ImportError("%s does not export %s: rebuild it (include/zksaas.h is the ABI)" % (LIB_PATH, ", ".join(missing[:5])))
    vp, sz, u64, i32 = C.c_void_p, C.c_size_t, C.c_uint64, C.c_int
    lib.zk_version.restype = C.c_char_p
    lib.zk_ctx_create.argtypes = [i32, i32, i32, C.POINTER(vp)]
    lib.zk_ctx_destroy.argtypes = [vp]
    lib.zk_ctx_destroy.restype = None
    lib.zk_last_error.argtypes = [vp, C.POINTER(i32)]
    lib.zk_last_error.restype = C.c_char_p
    lib.zk_ctx_n.argtypes = [vp]
    lib.zk_ctx_l.argtypes = [vp]
    lib.zk_fr_bytes.argtypes = [vp]
    lib.zk_fr_bytes.restype = sz
    lib.zk_fq_bytes.argtypes = [vp]
    lib.zk_fq_bytes.restype = sz
    lib.zk_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.zk_free.argtypes = [vp, vp]
    lib.zk_memcpy_h2d.argtypes = [vp, vp, vp, sz, vp]
    lib.zk_memcpy_d2h.argtypes = [vp, vp, vp, sz, vp]
    lib.zk_stream_sync.argtypes = [vp, vp]
    lib.zk_pss_pack.argtypes = [vp, vp, sz, i32, u64, vp, vp]
    lib.zk_pss_det_pack.argtypes = [vp, vp, sz, i32, vp, vp]
    lib.zk_pss_unpack.argtypes = [vp, vp, sz, vp, vp]
    lib.zk_pss_unpack2.argtypes = [vp, vp, C.POINTER(C.c_uint32), i32, sz, vp, vp]
    lib.zk_bitrev.argtypes = [vp, vp, i32, vp]
    lib.zk_vec_add.argtypes = [vp, vp, vp, sz, vp]
    lib.zk_vec_mul_sub.argtypes = [vp, vp, vp, vp, vp, sz, vp]
    lib.zk_fft1.argtypes = [vp, vp, i32, i32, sz, vp, vp]
    lib.zk_fft2_king.argtypes = [vp, vp, C.POINTER(C.c_uint32), i32, i32, i32, vp, i32, i32, u64, vp, vp, vp]
    lib.zk_d_fft.argtypes = [vp, vp, vp, vp, i32, i32, u64, vp, vp]
    lib.zk_d_ifft.argtypes = [vp, vp, vp, vp, i32, i32, vp, u64, vp, vp]
    lib.zk_fft_mask_sample.argtypes = [vp, i32, vp, i32, i32, u64, vp, vp, vp]
    lib.zk_deg_red.argtypes = [vp, vp, vp, vp, sz, u64, vp]
    lib.zk_degred_mask_sample.argtypes = [vp, sz, u64, vp, vp, vp]
    lib.zk_d_pp.argtypes = [vp, vp, vp, vp, vp, sz, u64, vp, vp]
    lib.zk_msm.argtypes = [vp, i32, vp, sz, vp, sz, vp, vp]
    lib.zk_d_msm.argtypes = [vp, i32, vp, vp, sz, vp, vp, vp, vp]
    lib.zk_base_mul.argtypes = [vp, i32, vp, vp, sz, vp, vp]
    lib.zk_circom_h.argtypes = [vp, vp, vp, vp, i32, vp, u64, vp, vp]
    lib.zk_groth16_prove.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, u64, vp, vp, vp, vp]
    lib.zk_d_msm_local.argtypes = [vp, i32, vp, vp, sz, i32, i32, vp, vp, vp]
    lib.zk_group_add.argtypes = [vp, i32, vp, vp, vp]
    lib.zk_msm_plan.argtypes = [vp, i32, C.c_size_t, C.POINTER(C.c_int)]
    lib.zk_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_longlong]
    lib.zk_msm_precompute.argtypes = [vp, i32, vp, sz, vp]
    lib.zk_msm_forget.argtypes = [vp, vp]
    lib.zk_msm_table_info.argtypes = [vp, i32, vp, C.POINTER(C.c_int)]
    lib.zk_msm_mask_sample.argtypes = [vp, i32, vp, u64, vp, vp]
    lib.zk_r1cs_qap.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, sz, sz, sz, i32, vp, vp, vp, vp]
    lib.zk_fr_to_bytes.argtypes = [vp, vp, sz, vp, vp]
    lib.zk_fr_from_bytes.argtypes = [vp, vp, sz, vp, vp]
    lib.zk_groth16_assemble.argtypes = [vp, vp, vp, vp, C.POINTER(vp), vp, vp, vp, vp]
    lib.zk_groth16_msms_begin.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, vp]
    lib.zk_groth16_msms_finish.argtypes = [vp, vp, C.POINTER(vp), vp]
    lib.zk_groth16_prove_async.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, u64, vp, C.POINTER(i32)]
    lib.zk_groth16_wait.argtypes = [vp, i32, vp, vp, vp]
    lib.zk_groth16_abort.argtypes = [vp, i32]
    u32p = C.POINTER(C.c_uint32)
    lib.zk_chacha20_block.argtypes = [u32p, u64, u64, u32p]
    lib.zk_chacha20_block.restype = None
    lib.zk_deg_red_points.argtypes = [vp, i32, vp, vp, vp, sz, vp, u64, vp, vp]
    lib.zk_degred_mask_sample_points.argtypes = [vp, i32, vp, sz, u64, vp, vp, vp]
    lib.zk_points_decompress.argtypes = [vp, i32, vp, sz, vp, vp]
    lib.zk_points_compress.argtypes = [vp, i32, vp, sz, vp, vp]
    lib.zk_libsnark_h.argtypes = [vp, vp, vp, vp, i32, C.POINTER(vp), C.POINTER(vp), u64, vp, vp]
    lib.zk_net_unique_id.argtypes = [vp]
    lib.zk_net_create.argtypes = [vp, i32, i32, i32, i32, C.POINTER(i32), vp, sz, C.POINTER(vp)]
    lib.zk_net_destroy.argtypes = [vp]
    lib.zk_net_destroy.restype = None
    lib.zk_net_last_error.argtypes = [vp, C.POINTER(i32)]
    lib.zk_net_last_error.restype = C.c_char_p
    lib.zk_net_set_timeout_ms.argtypes = [vp, u64]
    lib.zk_net_info.argtypes = [vp, C.POINTER(i32)]
    lib.zk_net_enter.argtypes = [vp, i32, u32p]
    lib.zk_net_gather.argtypes = [vp, i32, C.c_uint32, vp, sz, vp]
    lib.zk_net_scatter.argtypes = [vp, i32, C.c_uint32, vp, sz, vp]
    lib.zk_net_alltoall.argtypes = [vp, i32, C.c_uint32, vp, sz, vp]
    lib.zk_net_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.zk_net_gather_host.argtypes = [vp, i32, C.c_uint32, vp, sz, vp]
    lib.zk_net_bcast_host.argtypes = [vp, i32, C.c_uint32, vp, sz]
    lib.zk_net_sync.argtypes = [vp, i32]
    lib.zk_dist_d_fft.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, u64, vp]
    lib.zk_dist_d_ifft.argtypes = [vp, vp, i32, vp, vp, vp, i32, i32, vp, u64, vp]
    lib.zk_dist_deg_red.argtypes = [vp, vp, i32, vp, vp, vp, sz, u64, vp]
    lib.zk_dist_d_pp.argtypes = [vp, vp, i32, vp, vp, vp, vp, sz, u64, vp, vp]
    lib.zk_dist_d_msm.argtypes = [vp, vp, i32, i32, vp, vp, sz, vp, vp, vp, vp]
    lib.zk_dist_circom_h.argtypes = [vp, vp, vp, vp, vp, i32, vp, u64, vp, vp]
    lib.zk_dist_groth16_prove.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, u64, vp, vp, vp, vp]
    lib.zk_fq_selftest.argtypes = [vp, i32, vp, vp, vp, vp, sz, vp, vp]
    lib.zk_net_parties.argtypes = [vp, i32, C.POINTER(i32)]
    lib.zk_dist_groth16_prove_async.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, u64, vp, C.POINTER(i32)]
    lib.zk_dist_groth16_wait.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.zk_dist_groth16_prove_batch.argtypes = [vp, vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                C.POINTER(vp), vp, vp, i32, vp, u64, vp, vp, vp, vp]
    lib.zk_d_fft_host.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, u64, vp]
    lib.zk_deg_red_host.argtypes = [vp, vp, vp, vp, sz, u64, vp]
    lib.zk_d_pp_host.argtypes = [vp, vp, vp, vp, vp, sz, u64, vp, vp]
    lib.zk_circom_h_host.argtypes = [vp, vp, vp, vp, i32, vp, u64, vp, vp]
    lib.zk_groth16_prove_host.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, u64, vp, vp, vp, vp]
    lib.zk_dist_deg_red_points.argtypes = [vp, vp, i32, i32, vp, vp, vp, sz, vp, u64, vp, vp]
    lib.zk_dist_libsnark_h.argtypes = [vp, vp, vp, vp, vp, i32, C.POINTER(vp), C.POINTER(vp), u64, vp, vp]
    lib.zk_msm_host.argtypes = [vp, i32, vp, sz, vp, sz, vp, vp]
    lib.zk_d_msm_host.argtypes = [vp, i32, vp, vp, sz, vp, vp, vp, vp]
    lib.zk_vec_scale.argtypes = [vp, vp, vp, sz, vp]
    lib.zk_deg_red_parties.argtypes = [vp, vp, C.POINTER(C.c_uint32), i32, vp, vp, sz, u64, vp, vp]
    lib.zk_d_msm_parties.argtypes = [vp, i32, vp, vp, sz, C.POINTER(C.c_uint32), i32, vp, vp, vp, vp]
    lib.zk_pss_pack_points.argtypes = [vp, i32, vp, sz, i32, vp, vp]
    lib.zk_groth16_prove_batch.argtypes = [vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                           C.POINTER(vp), vp, vp, i32, vp, u64, vp, vp, vp, vp]
    lib.zk_groth16_prove_batch_async.argtypes = [vp, vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp),
                                                 C.POINTER(vp), vp, vp, i32, vp, u64, vp, C.POINTER(i32)]
    lib.zk_groth16_batch_wait.argtypes = [vp, i32, vp, vp, vp]
    lib.zk_msm_batch.argtypes = [vp, i32, vp, sz, C.POINTER(vp), i32, vp, vp]
    lib.zk_pss_unpack_points.argtypes = [vp, i32, vp, sz, vp, vp]
    lib.zk_pss_unpack2_points.argtypes = [vp, i32, vp, C.POINTER(C.c_uint32), i32, sz, vp, vp]
    lib.zk_groth16_reconstruct.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_uint32), i32, vp, vp, vp]
    lib.zk_msm_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.zk_profile_enable.argtypes = [vp, i32]
    lib.zk_profile_slots.argtypes = []
    lib.zk_profile_name.argtypes = [i32]
    lib.zk_profile_name.restype = C.c_char_p
    lib.zk_profile_read.argtypes = [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_long)]
    _lib = lib
    return lib
