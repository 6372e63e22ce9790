"""ctypes binding of libpinnfem_hip.so (include/pinnfem_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or does not match the
header this module raises, and every compute entry point of the package goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PINNFEM_LIB: another build of the same library (experiment builds, A/B runs); no fallback of any kind
LIB_PATH = os.environ.get("PINNFEM_LIB") or os.path.join(_HERE, "lib", "libpinnfem_hip.so")

PF_ABI_VERSION = 7
PF_OK, PF_ERR_ARG, PF_ERR_UNSUPPORTED, PF_ERR_HIP = 0, -1, -2, -3
PF_DOF_FIXED, PF_DOF_MEASURED, PF_DOF_SHARED, PF_DOF_GHOST = 1, 2, 4, 8
PF_WG_SHUFFLE, PF_WG_MFMA, PF_WG_MFMA44, PF_WG_MFMA32 = 0, 1, 2, 3
PF_N32_WIDTH_MAX = 30
PF_MLP_F32, PF_MLP_BF16 = 0, 1
PF_FE_REFERENCE, PF_FE_DELTA = 0, 1
PF_HIST_COLS = 6
PF_MAX_BLOCKS = 1024
PF_COMM_ID_BYTES = 128
PF_MAX_NODE_BLOCKS = 4096
PF_NODE_SLOTS = PF_MAX_NODE_BLOCKS + 8
PF_KERNEL_SLOTS = 9
PF_GRAPH_CONT_HEAD, PF_GRAPH_NO_TAIL = 1, 2
PF_FUSED_FORWARD, PF_FUSED_BACKWARD, PF_FUSED_THETA_UPDATE, PF_FUSED_U_PINGPONG, PF_FUSED_U_UPDATE = 1, 2, 4, 8, 16
KERNEL_SLOT_NAMES = ("net_forward_young", "net_forward_area", "node_residual", "elem_adjoint",
                     "net_backward_young", "net_backward_area", "node_gradu_adam", "theta_reduce_adam",
                     "finalize")

c_f32p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class PfMesh(C.Structure):
    _fields_ = [
        ("dim", C.c_int32), ("n_nodes", C.c_int32), ("n_elems", C.c_int32), ("n_dofs", C.c_int32),
        ("conn", C.c_void_p), ("egeo", C.c_void_p), ("ecent", C.c_void_p),
        ("adj_ptr", C.c_void_p), ("adj", C.c_void_p), ("f_ext", C.c_void_p),
        ("dof_flags", C.c_void_p), ("meas_val", C.c_void_p),
        ("n_meas", C.c_int32), ("_pad", C.c_int32),
    ]


class PfNet(C.Structure):
    _fields_ = [
        ("enabled", C.c_int32), ("in_dim", C.c_int32), ("width", C.c_int32),
        ("n_hidden", C.c_int32), ("positive", C.c_int32), ("scale", C.c_float),
        ("theta_off", C.c_int32), ("pad_off", C.c_int32),
    ]


class PfState(C.Structure):
    _fields_ = [
        ("iter", C.c_int32), ("done", C.c_int32), ("converged", C.c_int32), ("theta_half", C.c_int32),
        ("step_size_u", C.c_float), ("step_size_t", C.c_float), ("bc2_sqrt", C.c_float),
        ("_pad2", C.c_float),
        ("loss_total", C.c_float), ("loss_physics", C.c_float), ("loss_data", C.c_float),
        ("u_norm", C.c_float), ("residual_norm", C.c_float), ("theta_norm", C.c_float),
        ("u_half", C.c_int32), ("_pad3", C.c_float),
    ]


class PfProblem(C.Structure):
    _fields_ = [
        ("mesh", PfMesh), ("net", PfNet * 2),
        ("u", C.c_void_p), ("m_u", C.c_void_p), ("v_u", C.c_void_p),
        ("theta", C.c_void_p), ("m_t", C.c_void_p), ("v_t", C.c_void_p),
        ("n_theta", C.c_int32), ("n_theta_active", C.c_int32),
        ("tensor_off", C.c_void_p),
        ("n_tensors", C.c_int32), ("wg_mode", C.c_int32),
        ("lam", C.c_float), ("alpha_physics", C.c_float), ("alpha_data", C.c_float),
        ("lr_u", C.c_float), ("lr_t", C.c_float),
        ("tol", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
        ("use_data", C.c_int32), ("max_iter", C.c_int32),
        ("theta_pad", C.c_void_p), ("prop_e", C.c_void_p), ("prop_a", C.c_void_p),
        ("g_f", C.c_void_p), ("g_ea", C.c_void_p), ("grad_u", C.c_void_p),
        ("grad_theta", C.c_void_p), ("partials", C.c_void_p), ("hist", C.c_void_p),
        ("state", C.c_void_p),
        ("n_part_blocks", C.c_int32), ("pad_total", C.c_int32),
        ("pad_index", C.c_void_p),
        ("n_meas_f", C.c_float), ("fe_mode", C.c_int32),
        ("shared_dofs", C.c_void_p), ("shared_slot", C.c_void_p),
        ("n_shared", C.c_int32), ("n_iface", C.c_int32),
        ("own_lo", C.c_int32), ("own_hi", C.c_int32), ("part_half", C.c_int32), ("prop_double", C.c_int32),
        ("net_op", C.c_void_p), ("op_off", C.c_int32 * 2), ("coord_exp", C.c_int32), ("mlp_dtype", C.c_int32),
        ("elem_k", C.c_void_p), ("theta_alt", C.c_void_p), ("u_alt", C.c_void_p), ("adj_other", C.c_void_p),
    ]


class PfScalarId(C.Structure):
    _fields_ = [
        ("p", C.c_void_p), ("m_p", C.c_void_p), ("v_p", C.c_void_p), ("table", C.c_void_p),
        ("n_rows", C.c_int32), ("has_bounds", C.c_int32),
        ("lo", C.c_float * 2), ("hi", C.c_float * 2),
        ("inv_ea0", C.c_float), ("lr_p", C.c_float), ("n_free_f", C.c_float), ("_pad", C.c_float),
    ]


# every symbol include/pinnfem_hip.h declares: name -> (restype, argtypes)
_PP = C.POINTER(PfProblem)
SYMBOLS = {
    "pf_abi_version": (C.c_int, []),
    "pf_last_error": (C.c_char_p, []),
    "pf_net_param_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pf_padded_width": (C.c_int, [C.c_int]),
    "pf_net_pad_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pf_net_pad_index": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "pf_sizeof": (C.c_int, [C.c_int]),
    "pf_net_op_count": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "pf_partials_count": (C.c_longlong, [_PP]),
    "pf_fusion_info": (C.c_int, [_PP]),
    "pf_pack_theta": (C.c_int, [_PP, C.c_void_p]),
    "pf_net_forward": (C.c_int, [_PP, C.c_int, C.c_void_p]),
    "pf_net_forward_all": (C.c_int, [_PP, C.c_void_p]),
    "pf_net_backward_all": (C.c_int, [_PP, C.c_void_p]),
    "pf_internal_force": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_node_residual": (C.c_int, [_PP, C.c_void_p, C.c_void_p]),
    "pf_elem_adjoint": (C.c_int, [_PP, C.c_void_p]),
    "pf_net_backward": (C.c_int, [_PP, C.c_int, C.c_void_p]),
    "pf_node_gradu": (C.c_int, [_PP, C.c_int, C.c_void_p]),
    "pf_theta_reduce": (C.c_int, [_PP, C.c_int, C.c_void_p]),
    "pf_finalize": (C.c_int, [_PP, C.c_void_p]),
    "pf_reset": (C.c_int, [_PP, C.c_void_p]),
    "pf_gd_iterations": (C.c_int, [_PP, C.c_int, C.c_void_p]),
    "pf_graph_create": (C.c_int, [_PP, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pf_graph_create_ex": (C.c_int, [_PP, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pf_graph_tail": (C.c_int, [_PP, C.c_int, C.c_void_p]),
    "pf_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pf_graph_destroy": (C.c_int, [C.c_void_p]),
    "pf_gd_iterations_timed": (C.c_int, [_PP, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "pf_loss_and_grads": (C.c_int, [_PP, C.c_void_p]),
    "pf_shard_forward": (C.c_int, [_PP, C.c_void_p]),
    "pf_shard_backward": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_shard_flush": (C.c_int, [_PP, C.c_void_p, C.c_void_p]),
    "pf_shard_update_interior": (C.c_int, [_PP, C.c_void_p]),
    "pf_shard_update_shared": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_kv_f64": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_pcg_workspace_count": (C.c_longlong, [_PP]),
    "pf_pcg_begin": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]),
    "pf_pcg_iterations": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_void_p]),
    "pf_pcg_graph_create": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "pf_pcg_state": (C.c_int, [_PP, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]),
    "pf_comm_unique_id": (C.c_int, [C.c_char_p, C.c_void_p]),
    "pf_comm_create": (C.c_int, [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "pf_comm_destroy": (C.c_int, [C.c_void_p]),
    "pf_comm_abort": (C.c_int, [C.c_void_p]),
    "pf_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "pf_comm_all_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "pf_shard_iterations": (C.c_int, [_PP, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pf_shard_graph_create": (C.c_int, [_PP, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.POINTER(C.c_void_p)]),
    "pf_shard_iterations_graph": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "pf_scalar_gd_iterations": (C.c_int, [_PP, C.POINTER(PfScalarId), C.c_int, C.c_void_p]),
    "pf_adam": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                          C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p]),
    "pf_diag_k": (C.c_int, [_PP, C.c_void_p, C.c_void_p]),
    "pf_dense_k": (C.c_int, [_PP, C.c_void_p, C.c_void_p]),
    "pf_coo_k": (C.c_int, [_PP, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


class PinnFemHipError(RuntimeError):
    pass


_lib = None


def load():
    """Load the HIP library; raise loudly when it is missing or inconsistent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7: import it first so that this library binds to the
    # SAME HIP runtime instance (two runtimes in one process see no device)
    import torch  # noqa: F401
    from . import build as _build
    if os.path.exists(LIB_PATH) and os.path.exists(_build.HIPCC) and os.environ.get("PINNFEM_NO_REBUILD", "0") != "1":
        # an existing library older than any kernel source or the header is STALE: rebuild it rather than
        # test or benchmark an old binary (build() returns at once when the library is up to date)
        try:
            _build.build(verbose=False)
        except Exception as e:
            raise PinnFemHipError(f"rebuilding the stale {LIB_PATH} failed: {e}") from e
    if not os.path.exists(LIB_PATH):
        # not a fallback: build the HIP library in-tree when a fresh checkout has none yet
        if not os.path.exists(_build.HIPCC):
            raise PinnFemHipError(
                f"{LIB_PATH} not found and hipcc ({_build.HIPCC}) is not available: the HIP library "
                "cannot be built. Run `python -m pinn_fem_amd.build` on a ROCm machine. "
                "pinn_fem_amd has no CPU fallback.")
        try:
            _build.build(verbose=False)
        except Exception as e:
            raise PinnFemHipError(f"building {LIB_PATH} failed: {e}. pinn_fem_amd has no CPU fallback.") from e
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.pf_abi_version() != PF_ABI_VERSION:
        raise PinnFemHipError("libpinnfem_hip.so ABI version mismatch; rebuild the library")
    for idx, st in enumerate((PfMesh, PfNet, PfState, PfProblem, PfScalarId)):
        if lib.pf_sizeof(idx) != C.sizeof(st):
            raise PinnFemHipError(
                f"struct layout mismatch for {st.__name__}: C {lib.pf_sizeof(idx)} vs ctypes {C.sizeof(st)}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != PF_OK:
        msg = load().pf_last_error().decode(errors="replace")
        exc = NotImplementedError if rc == PF_ERR_UNSUPPORTED else (
            ValueError if rc == PF_ERR_ARG else PinnFemHipError)
        raise exc(f"{what or 'libpinnfem_hip'} failed ({rc}): {msg}")
