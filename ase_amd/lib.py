"""ctypes binding of ``libase_hip.so`` (C ABI declared in ``include/ase_hip.h``).

The library is the product: there is NO fallback.  Importing this module without the built
shared object raises; creating a ``HipBackend`` without a GPU raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libase_hip.so")

ABI_VERSION = 7
PPO_SCRATCH = 1024 * 72 + 8       # ASE_PPO_SCRATCH: doubles of ase_hip_ppo_head's workspace
TN_SLAB = 65536 + 256        # ASE_TN_SLAB: floats per work item in the grouped weight-gradient launch's workspace
F32, BF16, F32X3, F16, F32H3 = 0, 1, 2, 3, 4
ACT_NONE, ACT_RELU, ACT_TANH, ACT_SILU, ACT_ELU, ACT_GELU, ACT_SIGMOID, ACT_SELU, ACT_SOFTPLUS = range(9)
AUX_NONE, AUX_RELU_MASK, AUX_TANH_GRAD, AUX_RELU_BITS, AUX_PREACT = 0, 1, 2, 3, 4

# accumulator slots (ASE_ACC_*)
(ACC_MASK_SUM, ACC_A_LOSS, ACC_B_LOSS, ACC_ENTROPY, ACC_CLIPPED, ACC_C_LOSS, ACC_KL, ACC_DIV, ACC_BCE_AGENT,
 ACC_BCE_DEMO, ACC_AGENT_ACC, ACC_DEMO_ACC, ACC_GP, ACC_ENC, ACC_ENC_GP, ACC_LOGIT_W2, ACC_DISC_W2, ACC_ENC_W2,
 ACC_GRAD_SQ) = range(19)
ACC_COUNT = 24
# result slots (ASE_RES_*)
(RES_A_LOSS, RES_C_LOSS, RES_B_LOSS, RES_ENTROPY, RES_CLIP_FRAC, RES_KL, RES_DISC_LOSS, RES_DISC_GP,
 RES_DISC_LOGIT_LOSS, RES_DISC_AGENT_ACC, RES_DISC_DEMO_ACC, RES_ENC_LOSS, RES_DIV_LOSS, RES_LOSS,
 RES_MASK_SUM, RES_ENC_GP, RES_LR) = range(17)
RES_COUNT = 20

_p, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> argtypes; every function returns int.  Keep in lock-step with include/ase_hip.h
# (tests/test_abi.py parses the header and checks names + arity against this table).
SIGNATURES = {
    "ase_hip_gemm_nt": [_p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _i, _i, _p, _i, _p, _i64, _i, _i, _i, _i, _i, _i, _f, _p, _i, _p],
    "ase_hip_gemm_tn": [_p, _i64, _p, _i64, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _f, _p, _i, _p],
    "ase_hip_refresh_shadow": [_p, _i, _i, _p, _i64, _p, _i64, _i, _i, _i, _p],
    "ase_hip_refresh_shadow_multi": [_p, _i, _i, _p],
    "ase_hip_gather_multi": [_p, _i, _p, _i, _i, _i, _p],
    "ase_hip_rms_moments": [_p, _i64, _i, _p, _i, _i, _i, _p, _p, _p],
    "ase_hip_rms_finalize": [_p, _i, _p, _p, _i, _p, _p, _p],
    "ase_hip_rms_normalize": [_p, _i64, _i, _p, _i, _i, _i, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i, _p],
    "ase_hip_rms_unnormalize": [_p, _p, _p, _i64, _p],
    "ase_hip_gather_rows": [_p, _i64, _i, _p, _i, _i, _i, _p, _i64, _i, _p],
    "ase_hip_reduce_sum": [_p, _i64, _i, _p, _i, _p],
    "ase_hip_ppo_head": [_p, _i64, _p, _i64] + [_p] * 11 + [_p, _i64, _p, _i64, _p, _p, _p, _p, _p] + [_i] * 8 + [_f] * 6 + [_p, _i, _p],
    "ase_hip_disc_head": [_p, _i64, _p, _i64, _p, _p, _i, _i, _f, _f, _p, _i, _p],
    "ase_hip_enc_head": [_p, _i64, _p, _i64, _p, _i64, _p, _p, _p, _i, _i, _i, _f, _f, _p, _i, _p],
    "ase_hip_gp_seed": [_p, _i64, _p, _p, _i64, _i, _i, _f, _i, _i, _p],
    "ase_hip_gp_second": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _i, _p],
    "ase_hip_colsum": [_p, _i64, _i, _i, _f, _p, _p],
    "ase_hip_sqnorm": [_p, _i64, _i, _i, _p, _i, _d, _p, _i, _p],
    "ase_hip_finalize_scalars": [_p, _p, _i, _i, _i, _i, _i, _i] + [_f] * 11 + [_p, _f, _p],
    "ase_hip_enc_gp_seed": [_p, _i64, _p, _i64, _p, _i64, _i, _i, _f, _i, _p],
    "ase_hip_enc_gp_back": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i, _i, _f, _p, _i, _p],
    "ase_hip_clip_scale": [_p, _i64, _p, _f, _p],
    "ase_hip_begin_step": [_p, _p, _i, _p, _i, _p, _p],
    "ase_hip_adam": [_p, _p, _p, _p, _i64, _p, _p],
    "ase_hip_axpy": [_p, _p, _i64, _f, _p],
    "ase_hip_scaler_check": [_p, _i64, _i, _p, _p],
    "ase_hip_scaler_check_multi": [_p, _i, _i, _p, _p],
    "ase_hip_scaler_fold": [_p, _p, _p],
    "ase_hip_scaler_step": [_p, _p, _p, _p, _i64, _p, _p],
    "ase_hip_disc_reward": [_p, _i64, _p, _i64, _f, _p],
    "ase_hip_enc_reward": [_p, _i64, _p, _i64, _p, _i64, _i, _f, _p],
    "ase_hip_gae": [_p, _p, _p, _p, _p, _p, _f, _f, _f, _d, _d, _p, _p, _i, _i, _p],
    "ase_hip_adv_norm": [_p, _p, _p, _p, _p, _i64, _i, _i, _p],
    "ase_hip_ring_store": [_p, _i64, _i, _p, _i, _i, _i, _p, _i64, _i64, _p],
    "ase_hip_sample_latents": [_p, _i, _i, _p, _i64, _i, _p, _i64, _i, _p],
    "ase_hip_rms_moments_multi": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _p],
    "ase_hip_rms_normalize_multi": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "ase_hip_normalize_rows": [_p, _i64, _p, _i64, _i, _i, _p],
    "ase_hip_sample_actions": [_p, _i64, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "ase_hip_debug_nt_profile": [_p],
    "ase_hip_debug_nt_profile_clock": [_i],
    "ase_hip_motion_state": [_p] * 6 + [_i] + [_p] * 6 + [_i, _p, _p, _i, _p, _i] + [_p] * 8,
    "ase_hip_build_amp_obs": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i, _i, _p, _i, _i, _p],
    "ase_hip_gemm_nt_kernel_id": [_i, _i, _i, _i],
    "ase_hip_apply_multi": [_p, _i, _p, _p, _i, _p],
    "ase_hip_gemm_tn_grouped_plan": [_p, _i, _i, _p, _i, _p, _p, _i, _p],
    "ase_hip_prog_create": [_p],
    "ase_hip_prog_destroy": [_p],
    "ase_hip_prog_begin": [_p],
    "ase_hip_prog_end": [_p],
    "ase_hip_prog_size": [_p],
    "ase_hip_prog_launch": [_p],
    "ase_hip_prog_host": [_p, _p],
    "ase_hip_mark": [_p, _p],
    "ase_hip_wait": [_p, _i],
    "ase_hip_memset": [_p, _i, _i64, _p],
    "ase_hip_memcpy": [_p, _p, _i64, _p],
    "ase_hip_gemm_tn_grouped": [_p, _p, _i, _p, _i, _p, _p, _i, _p],
}


class AseHipError(RuntimeError):
    pass


def load():
    # torch first: PyTorch-ROCm ships its own libamdhip64; libase_hip.so must bind to THAT copy of the HIP runtime (the one
    # that owns torch's streams and allocations).  Loaded before torch, the library resolves libamdhip64 from the system
    # ROCm instead and its first launch fails with "no ROCm-capable device is detected" (two runtimes in one process).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise AseHipError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ase_amd/csrc`.  There is no CPU or PyTorch fallback for the update path.")
    lib = C.CDLL(LIB_PATH)
    lib.ase_hip_abi_version.restype = C.c_int
    lib.ase_hip_last_error.restype = C.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the library does not export it
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.ase_hip_abi_version() != ABI_VERSION:
        raise AseHipError("libase_hip.so ABI version mismatch")
    return lib


_lib = None


def get():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib


def check(rc, name):
    if rc != 0:
        raise AseHipError(f"{name} failed ({rc}): {get().ase_hip_last_error().decode()}")
