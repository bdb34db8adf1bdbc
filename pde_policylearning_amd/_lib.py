"""ctypes binding of the fnoengine C ABI (include/fnoengine.h).

The shared library is built in-tree by `python -m pde_policylearning_amd.build`
(or __graft_entry__.build()) with hipcc for gfx950.  There is NO fallback: if
the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FNO_LIB_PATH") or os.path.join(_HERE, "libfnoengine.so")      # (FNO_LIB_PATH: experiment builds of tools/)

FNO_MAX_LAYERS = 16
NORM_CODES = {"backward": 0, None: 0, "forward": 1, "ortho": 2}

c_float_p = C.POINTER(C.c_float)


class FnoSpecDesc(C.Structure):
    _fields_ = [("ndim", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("dims", C.c_int * 3), ("modes", C.c_int * 3),
                ("weight_last_extent", C.c_int), ("norm", C.c_int), ("input_gelu", C.c_int),
                ("weight_planes", C.c_int)]


class FnoModelDesc(C.Structure):
    _fields_ = [("ndim", C.c_int), ("Cin", C.c_int), ("C", C.c_int), ("Cout", C.c_int),
                ("hidden_proj", C.c_int), ("n_layers", C.c_int),
                ("dims", C.c_int * 3), ("modes", C.c_int * 3),
                ("norm", C.c_int), ("gelu_mask", C.c_uint), ("weight_planes", C.c_int)]


class FnoBlockTail(C.Structure):        # include/fnoengine.h: one-layer block stacks with a tail (RNO regressor layers)
    _fields_ = [("relu_out", C.c_int), ("drop_p", C.c_float), ("drop_seed", C.c_void_p), ("y", C.c_void_p)]


class FnoModelParams(C.Structure):
    _fields_ = [("lift_w", C.c_void_p), ("lift_b", C.c_void_p),
                ("skip_w", C.c_void_p * FNO_MAX_LAYERS),
                ("spec_w", (C.c_void_p * 4) * FNO_MAX_LAYERS),
                ("spec_bias", C.c_void_p),
                ("proj_w1", C.c_void_p), ("proj_b1", C.c_void_p),
                ("proj_w2", C.c_void_p), ("proj_b2", C.c_void_p)]


FnoModelGrads = FnoModelParams   # same layout, mutable pointers


class FnoChanflowGrid(C.Structure):
    _fields_ = [("Nx", C.c_int), ("Ny", C.c_int), ("Nz", C.c_int), ("dx", C.c_double), ("dz", C.c_double), ("nu", C.c_double)]

_lib = None


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"fnoengine: {LIB_PATH} not found. Build it with `python -m pde_policylearning_amd.build` "
            "(needs hipcc; gfx950). There is no CPU / PyTorch fallback.")
    # torch first: its wheel carries its own libamdhip64.so.7 / libhsa-runtime64, and the library works on pointers torch's
    # allocator hands out, so both must sit on ONE HIP runtime.  Loaded after torch, the library's NEEDED libamdhip64.so.7
    # resolves to the copy already in the process; loaded before it, /opt/rocm's copy comes in as a second runtime and
    # every later call fails with "no HIP device" (seen with build() and smoke() in one process).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
    L.fno_version.restype = ci
    L.fno_last_error.restype = C.c_char_p
    L.fno_set_gemm_mode.argtypes = [ci]
    L.fno_set_gemm_mode.restype = None
    L.fno_get_gemm_mode.restype = ci
    L.fno_set_mode_gemm.argtypes = [ci]
    L.fno_set_mode_gemm.restype = None
    L.fno_get_mode_gemm.restype = ci
    L.fno_set_fused_mid.argtypes = [ci]
    L.fno_set_fused_mid.restype = None
    L.fno_get_fused_mid.restype = ci
    L.fno_spec_plan_create.argtypes = [C.POINTER(FnoSpecDesc), C.POINTER(vp)]
    L.fno_spec_plan_destroy.argtypes = [vp]
    L.fno_spec_plan_destroy.restype = None
    L.fno_spec_workspace_bytes.argtypes = [vp, ci]
    L.fno_spec_workspace_bytes.restype = sz
    L.fno_spec_xhat_bytes.argtypes = [vp, ci]
    L.fno_spec_xhat_bytes.restype = sz
    L.fno_spec_forward.argtypes = [vp, ci, vp, C.POINTER(vp), vp, vp, vp, vp, sz, vp]
    L.fno_spec_backward.argtypes = [vp, ci, vp, vp, C.POINTER(vp), vp, C.POINTER(vp), vp, vp, sz, vp]
    L.fno_model_plan_create.argtypes = [C.POINTER(FnoModelDesc), C.POINTER(vp)]
    L.fno_model_plan_destroy.argtypes = [vp]
    L.fno_model_plan_destroy.restype = None
    L.fno_model_workspace_bytes.argtypes = [vp, ci]
    L.fno_model_workspace_bytes.restype = sz
    L.fno_model_saved_bytes.argtypes = [vp, ci]
    L.fno_model_saved_bytes.restype = sz
    L.fno_model_forward.argtypes = [vp, ci, C.POINTER(FnoModelParams), vp, vp, vp, vp, sz, vp]
    L.fno_model_backward.argtypes = [vp, ci, C.POINTER(FnoModelParams), vp, vp, vp,
                                     C.POINTER(FnoModelGrads), vp, sz, vp]
    L.fno_fanout_saved_bytes.argtypes = [vp, ci, ci]
    L.fno_fanout_saved_bytes.restype = sz
    L.fno_fanout_workspace_bytes.argtypes = [vp, ci, ci]
    L.fno_fanout_workspace_bytes.restype = sz
    L.fno_fanout_forward.argtypes = [vp, ci, ci, C.POINTER(FnoModelParams), C.POINTER(vp), vp, C.POINTER(vp), vp, vp, sz, vp]
    L.fno_fanout_backward.argtypes = [vp, ci, ci, C.POINTER(FnoModelParams), vp, C.POINTER(vp), vp,
                                      C.POINTER(FnoModelGrads), C.POINTER(vp), vp, vp, sz, vp]
    fl = C.c_float
    L.fno_lploss_workspace_bytes.argtypes = [ci]
    L.fno_lploss_workspace_bytes.restype = sz
    L.fno_lploss_rel_forward.argtypes = [ci, sz, vp, vp, vp, vp, ci, fl, ci, vp, vp, sz, vp]
    L.fno_lploss_rel_backward.argtypes = [ci, sz, vp, vp, vp, ci, fl, vp, vp, vp, sz, vp]
    db = C.c_double       # Adam hyperparameters cross the ABI as the doubles torch.optim.Adam holds
    L.fno_adam_step.argtypes = [sz, vp, vp, vp, vp, db, db, db, db, db, ci, vp]
    L.fno_adam_step_dev.argtypes = [sz, vp, vp, vp, vp, db, db, db, db, db, vp, vp, vp]
    L.fno_adam_scalars.argtypes = [db, db, db, ci, vp]
    L.fno_adam_scalars.restype = None
    L.fno_adam_prep_dev.argtypes = [vp, vp, db, db, db, vp]
    L.fno_adam_step_range.argtypes = [sz, vp, vp, vp, vp, db, db, db, db, db, ci, vp, vp]
    L.fno_adam_step_live.argtypes = [sz, ci, ci, vp, vp, vp, vp, db, db, db, db, db, ci, vp, vp]
    L.fno_adam_replay_prep.argtypes = [vp, ci, ci, db, db, db, vp]
    L.fno_adam_replay_dead.argtypes = [sz, ci, ci, vp, vp, vp, ci, vp, ci, db, db, db, db, vp]
    L.fno_pointwise_workspace_bytes.argtypes = [ci]
    L.fno_pointwise_workspace_bytes.restype = sz
    L.fno_pointwise_forward.argtypes = [ci, ci, sz, vp, vp, vp, vp, ci, vp, vp]
    L.fno_pointwise_backward.argtypes = [ci, ci, sz, vp, vp, vp, vp, ci, vp, vp, vp, vp, sz, vp]
    L.fno_debug_pino_twopass.argtypes = [ci]
    L.fno_debug_pino_twopass.restype = None
    L.fno_projection_workspace_bytes.argtypes = [ci, ci]
    L.fno_projection_workspace_bytes.restype = sz
    L.fno_projection_forward.argtypes = [ci, ci, ci, ci, sz] + [vp] * 7
    L.fno_projection_backward.argtypes = [ci, ci, ci, ci, sz] + [vp] * 11 + [sz, vp]
    L.fno_projection_forward_act.argtypes = [ci, ci, ci, ci, sz] + [vp] * 5 + [ci, vp, vp]
    L.fno_projection_backward_act.argtypes = [ci, ci, ci, ci, sz] + [vp] * 5 + [ci] + [vp] * 6 + [sz, vp]
    L.fno_lifting_workspace_bytes.argtypes = [ci]
    L.fno_lifting_workspace_bytes.restype = sz
    L.fno_lifting_forward.argtypes = [ci, ci, ci, sz, vp, vp, vp, vp, vp]
    L.fno_lifting_backward.argtypes = [ci, ci, ci, sz, vp, vp, vp, vp, vp, sz, vp]
    L.fno_rno_gate_partials.restype = ci
    L.fno_rno_reset_gate_forward.argtypes = [sz] + [vp] * 7
    L.fno_rno_reset_gate_backward.argtypes = [sz] + [vp] * 7
    L.fno_rno_output_gate_forward.argtypes = [sz] + [vp] * 15
    L.fno_rno_output_gate_backward.argtypes = [sz] + [vp] * 11
    gp, dp = C.POINTER(FnoChanflowGrid), C.POINTER(C.c_double)
    L.fno_chanflow_pack_metrics.argtypes = [ci, dp, dp, dp, dp]
    L.fno_chanflow_rhs.argtypes = [gp, ci, ci, vp, vp, vp, vp, vp, C.c_double, vp, vp, vp, vp]
    L.fno_chanflow_pde_loss_workspace_bytes.argtypes = [gp, ci]
    L.fno_chanflow_pde_loss_workspace_bytes.restype = sz
    L.fno_chanflow_pde_loss_forward.argtypes = [gp, ci, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fno_chanflow_pde_loss_backward.argtypes = [gp, ci, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fno_pino_loss_workspace_bytes.argtypes = [ci, ci, ci]
    L.fno_pino_loss_workspace_bytes.restype = sz
    L.fno_pino_loss_forward.argtypes = [ci, ci, ci, vp, vp, vp, vp, fl, vp, vp, vp, sz, vp]
    L.fno_pino_loss_backward.argtypes = [ci, ci, ci, vp, vp, vp, vp, fl, vp, vp, vp, vp, sz, vp]
    L.fno_model_backward_dx.argtypes = [vp, ci, C.POINTER(FnoModelParams), vp, vp, vp,
                                        C.POINTER(FnoModelGrads), vp, vp, sz, vp]
    L.fno_model_backward_part.argtypes = [vp, ci, C.POINTER(FnoModelParams), vp, vp, vp,
                                          C.POINTER(FnoModelGrads), vp, vp, sz, vp, ci, ci]
    L.fno_model_forward_tail.argtypes = [vp, ci, C.POINTER(FnoModelParams), vp, vp, vp, vp, sz, vp, C.POINTER(FnoBlockTail)]
    L.fno_model_backward_tail.argtypes = [vp, ci, C.POINTER(FnoModelParams), vp, vp, vp,
                                          C.POINTER(FnoModelGrads), vp, vp, sz, vp, C.POINTER(FnoBlockTail)]
    L.fno_dropout_scale.argtypes = [sz, C.c_float, vp, vp, vp]
    L.fno_profile_enable.argtypes = [ci]
    L.fno_profile_enable.restype = None
    L.fno_profile_reset.restype = None
    L.fno_profile_count.restype = ci
    L.fno_profile_get.argtypes = [ci, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.POINTER(ci)]
    L.fno_profile_get_terms.argtypes = [ci]
    L.fno_profile_get_terms.restype = ci
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().fno_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"fnoengine {what} failed (code {rc}): {msg}")


EXPORTED_SYMBOLS = [
    "fno_version", "fno_last_error", "fno_set_gemm_mode", "fno_get_gemm_mode", "fno_set_mode_gemm", "fno_get_mode_gemm",
    "fno_set_fused_mid", "fno_get_fused_mid",
    "fno_spec_plan_create", "fno_spec_plan_destroy", "fno_spec_workspace_bytes", "fno_spec_xhat_bytes",
    "fno_spec_forward", "fno_spec_backward",
    "fno_model_plan_create", "fno_model_plan_destroy", "fno_model_workspace_bytes", "fno_model_saved_bytes",
    "fno_model_forward", "fno_model_backward", "fno_model_backward_dx", "fno_model_backward_part",
    "fno_model_forward_tail", "fno_model_backward_tail", "fno_dropout_scale",
    "fno_fanout_saved_bytes", "fno_fanout_workspace_bytes", "fno_fanout_forward", "fno_fanout_backward",
    "fno_lploss_workspace_bytes", "fno_lploss_rel_forward", "fno_lploss_rel_backward", "fno_adam_step", "fno_adam_step_dev",
    "fno_adam_scalars", "fno_adam_prep_dev", "fno_adam_step_range", "fno_adam_step_live", "fno_adam_replay_prep",
    "fno_adam_replay_dead",
    "fno_pointwise_workspace_bytes", "fno_pointwise_forward", "fno_pointwise_backward",
    "fno_projection_workspace_bytes", "fno_projection_forward", "fno_projection_backward",
    "fno_projection_forward_act", "fno_projection_backward_act",
    "fno_lifting_workspace_bytes", "fno_lifting_forward", "fno_lifting_backward",
    "fno_rno_gate_partials", "fno_rno_reset_gate_forward", "fno_rno_reset_gate_backward",
    "fno_rno_output_gate_forward", "fno_rno_output_gate_backward",
    "fno_pino_loss_workspace_bytes", "fno_pino_loss_forward", "fno_pino_loss_backward",
    "fno_chanflow_pack_metrics", "fno_chanflow_rhs", "fno_chanflow_pde_loss_workspace_bytes",
    "fno_chanflow_pde_loss_forward", "fno_chanflow_pde_loss_backward",
    "fno_profile_enable", "fno_profile_count", "fno_profile_get", "fno_profile_get_terms", "fno_profile_reset",
]


def profile_summary(with_terms=False):
    """[(kernel name, total ms, launches)] recorded since the last reset (profiling on); with_terms: a fourth entry, the
    matrix pipe of the kernel's channel GEMMs (fno_profile_get_terms: 0 not stated, 1 fp32, 2 two fp16 terms, 3 three bf16)."""
    L = lib()
    out = []
    for i in range(L.fno_profile_count()):
        name, ms, n = C.c_char_p(), C.c_float(), C.c_int()
        L.fno_profile_get(i, C.byref(name), C.byref(ms), C.byref(n))
        rec = (name.value.decode(), float(ms.value), int(n.value))
        out.append(rec + (int(L.fno_profile_get_terms(i)),) if with_terms else rec)
    return out
