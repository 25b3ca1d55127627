"""ctypes binding of libdrqv2_hip.so (include/drqv2_hip.h).  Loading fails loudly: there is no
CPU or PyTorch fallback for the update path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdrqv2_hip.so")

c_float_p = C.c_void_p   # device pointers travel as integers
c_u8_p = C.c_void_p
stream_t = C.c_void_p


class DrqStep(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("global_B", C.c_int), ("C", C.c_int), ("A", C.c_int), ("F", C.c_int), ("H", C.c_int),
        ("obs", c_u8_p), ("next_obs", c_u8_p),
        ("action", c_float_p), ("reward", c_float_p), ("discount", c_float_p),
        ("shift_obs", c_float_p), ("shift_next", c_float_p),
        ("noise_critic", c_float_p), ("noise_actor", c_float_p),
        ("base_grid", c_float_p),
        ("params", c_float_p), ("grads", c_float_p), ("adam_m", c_float_p), ("adam_v", c_float_p),
        ("ws", c_float_p), ("ws_bytes", C.c_size_t),
        ("sums", c_float_p),
        ("lr", C.c_double), ("tau", C.c_double),
        ("std", C.c_float), ("clip", C.c_float),
        ("step_critic", C.c_long), ("step_enc", C.c_long), ("step_actor", C.c_long),
        ("gscale", C.c_float),
        ("stream", stream_t),
        ("sums_host", c_float_p),
        ("store_aug_next", C.c_int),
        ("bf16", C.c_int),
        ("timing_events", C.POINTER(C.c_void_p)),
        ("timing_n", C.c_int),
        ("obs_index", C.c_void_p), ("next_obs_index", C.c_void_p),
        ("flags", C.c_int),
    ]


I, L, F, D, P, SZ = C.c_int, C.c_long, C.c_float, C.c_double, C.c_void_p, C.c_size_t

# name -> (restype, argtypes); mirrors include/drqv2_hip.h one to one
PROTOTYPES = {
    "drq_abi_version": (I, []),
    "drq_aug_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "drq_aug_fwd_f32": (I, [P, P, P, P, I, I, I, I, P]),
    "drq_conv1_aug_fwd": (I, [P, P, P, P, P, P, P, P, P, I, I, P]),
    "drq_conv1_aug_fwd_bf16": (I, [P, P, P, P, P, P, P, P, P, I, I, P]),
    "drq_conv1_aug_fwd_bf16_nhwc": (I, [P, P, P, P, P, P, P, P, P, I, I, P]),
    "drq_conv1_aug_fwd_indexed": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, P]),
    "drq_conv3x3_fwd": (I, [P, P, P, P, I, I, I, I, I, L, L, L, L, P]),
    "drq_conv3x3_dgrad": (I, [P, P, P, P, I, I, L, L, L, L, P]),
    "drq_conv3x3_fwd_wino": (I, [P, P, P, P, I, I, I, L, L, L, L, P]),
    "drq_conv3x3_dgrad_wino": (I, [P, P, P, P, I, I, L, L, L, L, P]),
    "drq_conv3x3_wgrad_wino": (I, [P, P, P, P, I, I, L, L, L, L, P, SZ, P]),
    "drq_conv3x3_wgrad": (I, [P, P, P, P, I, I, I, I, L, L, L, L, P, SZ, P]),
    "drq_conv3x3_wgrad_ws_bytes": (SZ, []),
    "drq_conv3x3_fwd_bf16": (I, [P, P, P, P, I, I, I, L, L, L, L, P]),
    "drq_conv3x3_dgrad_bf16": (I, [P, P, P, P, I, I, L, L, L, L, P]),
    "drq_conv3x3_wgrad_bf16": (I, [P, P, P, P, I, I, L, L, L, L, P, SZ, P]),
    "drq_conv3x3_fwd_bf16_nhwc": (I, [P, P, P, P, I, I, I, I, I, P]),
    "drq_conv3x3_dgrad_bf16_nhwc": (I, [P, P, P, P, I, I, I, I, L, L, L, L, P]),
    "drq_conv3x3_wgrad_bf16_nhwc": (I, [P, P, P, P, I, I, I, L, L, L, L, P, SZ, P]),
    "drq_gemm_f32": (I, [P, L, I, P, L, I, P, L, I, I, I, I, L, L, L, P, L, I, P, I, L, I, I, I, P, SZ, P]),
    "drq_gemm_batched_f32": (I, [I, P, L, I, P, L, I, P, L, I, I, I, P, I, P, I, P, I, I, I, P, SZ, P]),
    "drq_gemm_batched_bf16": (I, [I, P, L, I, P, L, I, P, L, I, I, I, P, I, P, I, P, I, I, P, SZ, P]),
    "drq_gemm_batched_partial": (I, [I, P, L, I, P, L, I, P, L, I, I, I, P, P, SZ, C.POINTER(I), P]),
    "drq_mlp_fwd": (I, [I, P, L, P, L, P, L, I, I, I, P, I, P, P, C.POINTER(I), P]),
    "drq_mlp_dgrad": (I, [I, P, L, P, L, P, L, I, I, I, P, I, P]),
    "drq_mlp_wgrad_dgrad": (I, [I, P, L, P, L, P, P, P, L, P, L, P, I, I, I, I, P]),
    "drq_ln_l1_fwd": (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, L, P]),
    "drq_policy_out_l1_fwd": (I, [P, P, P, P, I, I, I, I, I, F, F, I, P, P, P, L, P, P, P, L, I, P, P, P, P]),
    "drq_qout_fwd": (I, [I, P, P, P, P, I, I, P]),
    "drq_qout_bwd": (I, [I, P, P, P, P, P, P, I, I, P]),
    "drq_ln_tanh_fwd_multi": (I, [I, P, I, P, P, P, P, P, P, I, I, P]),
    "drq_ln_tanh_fwd": (I, [P, I, P, P, P, I, P, P, I, I, P]),
    "drq_ln_tanh_fwd2": (I, [P, P, I, P, P, P, P, P, I, P, I, P, P, P, P, I, I, P]),
    "drq_ln_tanh_bwd": (I, [P, I, P, I, P, I, P, P, P, P, P, P, P, I, I, P]),
    "drq_colsum": (I, [P, L, L, P, L, I, I, I, P]),
    "drq_trunc_normal_sample": (I, [P, P, F, F, I, P, P, L, I, I, P]),
    "drq_copy_cols": (I, [P, I, P, L, I, I, P]),
    "drq_td_mse": (I, [P, P, P, P, P, P, P, P, P, I, F, P]),
    "drq_actor_loss": (I, [P, P, P, L, P, F, P, P, P, I, I, F, P]),
    "drq_actor_dmu": (I, [P, P, L, I, P, P, I, I, P]),
    "drq_adam_flat": (I, [P, P, P, P, L, D, L, F, P, D, P]),
    "drq_ema_flat": (I, [P, P, L, D, P]),
    "drq_sum_slices": (I, [P, L, I, P, L, P]),
    "drq_adam_reduce_flat": (I, [P, P, L, I, P, P, L, D, L, F, P]),
    "drq_fill": (I, [P, L, F, P]),
    "drq_u8_normalize": (I, [P, P, L, P]),
    "drq_nstep_gather": (I, [P, P, P, P, P, I, I, L, I, F, P, P, P, P, P, P]),
    "drq_tanh": (I, [P, P, L, P]),
    "drq_param_layout": (I, [I, I, I, I, C.POINTER(L), I]),
    "drq_step_ws_bytes": (SZ, [I, I, I, I, I]),
    "drq_step_ws_offset": (L, [I, I, I, I, I, I]),
    "drq_update_phase": (I, [C.POINTER(DrqStep), I]),
    "drq_act_forward": (I, [C.POINTER(DrqStep), P, I, P]),
    "drq_publish_sums": (I, [P, P, C.c_uint, P]),
    "drq_rng_draws": (I, [C.c_uint64, C.c_uint64, I, I, I, P, P, P, P, P]),
}

WS_IDS = ["AUG", "ACT1", "ACT2", "ACT3", "FEAT", "Z_NEXT", "Z_OBS", "HA_T", "HA_C", "H_AN", "H_AO", "Q", "TQ",
          "DQ", "MU_O", "DY4", "DY3", "DY2", "DY1", "DZ_C", "DZ_A", "HA_C2", "P1", "P2", "C1", "C2"]

_lib = None


class DrqError(RuntimeError):
    pass


def load(dev=False):
    """Returns the loaded library; raises if it has not been built (python -m drqv2_amd.build).
    dev=True (tools/ only, must be the first load of the process): the -DDRQ_DEV build with the timing ablations
    and time-stamp hooks (python -m drqv2_amd.build --dev); the product path never asks for it."""
    global _lib
    if _lib is not None:
        if dev and not hasattr(_lib, "drq_dev_conv_variant"):
            raise DrqError("the product library is already loaded in this process; load(dev=True) must come first")
        return _lib
    path = os.path.join(HERE, "libdrqv2_hip_dev.so") if dev else LIB_PATH
    if not os.path.exists(path):
        raise DrqError(f"{path} is missing: build it with `python -m drqv2_amd.build{' --dev' if dev else ''}` "
                       "(hipcc --offload-arch=gfx950).  The DrQ-v2 update path has no fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError = ABI mismatch, also loud
        fn.restype = res
        fn.argtypes = args
    if lib.drq_abi_version() != 7:
        raise DrqError("libdrqv2_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc == 0:
        return
    if rc == -1:
        raise DrqError(f"{what}: bad argument / unsupported shape (DRQ_EARG)")
    if rc == -2:
        raise DrqError(f"{what}: workspace too small (DRQ_EWS)")
    raise DrqError(f"{what}: HIP error {rc}")


def ptr(t):
    """data pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def param_layout(Cc, A, Fd, H):
    lib = load()
    buf = (L * 59)()
    n = lib.drq_param_layout(Cc, A, Fd, H, buf, 59)
    if n != 59:
        raise DrqError("drq_param_layout failed")
    v = list(buf)
    return {"enc": v[0:8], "critic": v[8:24], "actor": v[24:34], "target": v[34:50],
            "seg": {"enc": (v[50], v[51]), "critic": (v[52], v[53]), "actor": (v[54], v[55]),
                    "target": (v[56], v[57])}, "total": v[58]}
