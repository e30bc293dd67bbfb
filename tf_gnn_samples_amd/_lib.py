"""ctypes binding of librelgnn.so (C ABI: include/relgnn.h).

This is the ONLY compute backend of the package.  There is no CPU / eager-PyTorch fallback:
if the HIP library is missing, or a tensor is not a float32/int32 CUDA(HIP) tensor, the
call fails loudly.  (The NumPy oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
import ctypes
from pathlib import Path

import torch  # imported first on purpose: makes torch's libamdhip64.so.7 the process-wide HIP runtime

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "librelgnn.so"

OK, EINVAL, ENOSPC, EHIP, EUNSUPPORTED = 0, 1, 2, 3, 4
ERRFLAG_INDEX_OUT_OF_RANGE = 1

AGG_SUM, AGG_MEAN, AGG_SQRT_N, AGG_MAX = 0, 1, 2, 3
MT_MAX = 48
ACT_LINEAR, ACT_TANH, ACT_RELU, ACT_LEAKY_RELU, ACT_ELU, ACT_SELU, ACT_GELU = range(7)
ACT_NAMES = ("linear", "tanh", "relu", "leaky_relu", "elu", "selu", "gelu")      # utils.get_activation's strings, by id

_c_i32, _c_i64, _c_f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
_ptr = ctypes.c_void_p

# name -> (restype, argtypes); must list every function declared in include/relgnn.h
_SIGNATURES = {
    "relgnn_abi_version": (ctypes.c_int, []),
    "relgnn_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "relgnn_relational_keys": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _c_i32, _c_i64, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_relational_keys2": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _c_i32, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_relational_keys_all": (ctypes.c_int, [_ptr, _ptr, _c_i32, _c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_relational_plan_workspace_bytes": (ctypes.c_size_t, [_c_i64, _c_i32]),
    "relgnn_relational_plan": (ctypes.c_int, [_ptr, _ptr, _ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    "relgnn_segment_plan_workspace_bytes": (ctypes.c_size_t, [_c_i64, _c_i64]),
    "relgnn_segment_plan": (ctypes.c_int, [_ptr, _c_i64, _c_i64, _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    "relgnn_gather_i32": (ctypes.c_int, [_ptr, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_gather_div_i32": (ctypes.c_int, [_ptr, _ptr, _c_i64, _c_i32, _ptr, _ptr]),
    "relgnn_gather_f32": (ctypes.c_int, [_ptr, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_invert_perm": (ctypes.c_int, [_ptr, _c_i64, _ptr, _ptr]),
    "relgnn_degree_scale": (ctypes.c_int, [_ptr, _ptr, _c_i32, _c_i32, _c_f32, _ptr, _ptr]),
    "relgnn_segment_counts_scale": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr]),
    "relgnn_seg_reduce_fwd": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _c_i64, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr]),
    "relgnn_seg_reduce_acc64_fwd": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _c_i64, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr]),
    "relgnn_seg_reduce_fwd_rowmax": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _c_i64, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_seg_reduce_msgact_fwd": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _c_i64, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _ptr, _c_i64, _ptr]),
    "relgnn_msg_act_bwd": (ctypes.c_int, [_c_i32, _ptr, _c_i32, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_seg_max_count": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_seg_max_bwd": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "relgnn_act_bwd_from_output": (ctypes.c_int, [_c_i32, _ptr, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_film_fwd": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_film_bwd_film": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_film_bwd_msg_masked": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _c_i32, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "relgnn_film_bwd_msg": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_rgat_fwd": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i32, _c_i32, _ptr, _c_f32, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_rgat_bwd_logits": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i32, _c_i32, _ptr, _c_f32, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr]),
    "relgnn_rgat_bwd_msg": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_rgat_alpha": (ctypes.c_int, [_ptr, _ptr, _c_i32, _ptr, _c_i32, _c_i32, _ptr, _c_f32, _ptr, _ptr]),
    "relgnn_headw_reduce": (ctypes.c_int, [_ptr, _c_i64, _c_i64, _c_i32, _c_i32, _ptr, _c_i64, _c_i32, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr]),
    "relgnn_rgat_dz": (ctypes.c_int, [_ptr, _c_i64, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i32, _c_i32, _ptr, _c_f32, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr]),
    "relgnn_pair_fwd": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i64, _ptr]),
    "relgnn_pair_bwd_q": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _c_i32, _c_i32, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr]),
    "relgnn_pair_bwd_p": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "relgnn_column_sum_workspace_bytes": (ctypes.c_size_t, [_c_i64, _c_i32]),
    "relgnn_column_sum": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i64, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    "relgnn_mt_l2norm_workspace_bytes": (ctypes.c_size_t, []),
    "relgnn_mt_l2norm": (ctypes.c_int, [_ptr, _ptr, _c_i32, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    "relgnn_mt_adam_clip": (ctypes.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i32, _ptr, _c_f32, _c_f32, _c_f32, _c_f32, _c_f32, _ptr]),
    "relgnn_adam_step_size": (ctypes.c_int, [_ptr, _c_f32, _c_f32, _c_f32, _ptr]),
    "relgnn_mt_adam_clip_devlr": (ctypes.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i32, _ptr, _c_f32, _ptr, _c_f32, _c_f32, _c_f32, _ptr]),
    "relgnn_sigmoid_ce_stats_workspace_bytes": (ctypes.c_size_t, []),
    "relgnn_sigmoid_ce_stats": (ctypes.c_int, [_ptr, _ptr, _c_i64, _c_f32, _ptr, _ptr, ctypes.c_size_t, _ptr]),
    "relgnn_sigmoid_ce_bwd": (ctypes.c_int, [_ptr, _ptr, _c_i64, _ptr, _c_f32, _ptr, _ptr, _ptr]),
    "relgnn_sigmoid_ce_bwd_padded": (ctypes.c_int, [_ptr, _ptr, _c_i64, _c_i32, _ptr, _c_f32, _ptr, _ptr, _c_i32, _ptr]),
    "relgnn_gru_gates_fwd": (ctypes.c_int, [_ptr, _ptr, _ptr, _c_i64, _c_i32, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_gru_cell_fwd_supported": (ctypes.c_int, [_c_i32, _c_i32, _c_i32]),
    "relgnn_gru_cell_fwd_xf32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr, _ptr, _c_i32, _ptr, _ptr, _ptr, _ptr, _ptr,
                                                _c_i64, _c_i32, _c_i32, _ptr, _ptr]),
    "relgnn_gru_cell_bwd_xf32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _ptr, _c_i32, _ptr, _ptr, _ptr, _c_i64,
                                                _c_i32, _c_i32, _ptr, _ptr]),
    "relgnn_gru_out_fwd": (ctypes.c_int, [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr]),
    "relgnn_gru_out_bwd": (ctypes.c_int, [_ptr, _ptr, _ptr, _ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_gru_gates_bwd": (ctypes.c_int, [_ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i32, _ptr, _ptr, _ptr]),
    "relgnn_pair_materialize": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _ptr, _c_i64, _ptr, _ptr, _c_i64, _ptr]),
    "relgnn_rgat_scores_groups": (_c_i64, [_c_i64]),
    "relgnn_rgat_scores_fwd": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _ptr, _c_i32, _c_i64, _ptr, _ptr, _ptr]),
    "relgnn_rgat_scores_bwd": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _ptr, _c_i32, _c_i64, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "relgnn_layer_norm_groups": (_c_i64, [_c_i64, _c_i32]),
    "relgnn_layer_norm_fwd": (ctypes.c_int, [_ptr, _c_i64, _c_i64, _c_i32, _ptr, _ptr, _c_f32, _ptr, _c_i64, _ptr, _ptr, _ptr]),
    "relgnn_layer_norm_bwd": (ctypes.c_int, [_ptr, _c_i64, _ptr, _c_i64, _c_i64, _c_i32, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _ptr]),
    "relgnn_panel_gemm_zeros_floats": (ctypes.c_int, []),
    "relgnn_panel_gemm_f32": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _ptr, _c_i64, _ptr, _c_i32, _c_i64, _ptr, _ptr, _ptr,
                                             _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _c_i64, _c_i64, _c_i64, _c_i32, _ptr]),
    "relgnn_limb_elements": (_c_i64, [_c_i64, _c_i64]),
    "relgnn_limb_split_f32": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _c_i32, _ptr, _ptr]),
    "relgnn_limb_gemm_f32": (ctypes.c_int, [_c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_limb_gemm_xf32": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_limb_gemm_xf32_dact": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _c_i32,
                                                  _c_i32, _ptr]),
    "relgnn_limb16_gemm_xf32_dact": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i32, _ptr, _ptr, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr,
                                                    _c_i64, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_rgcn_fused_fwd": (ctypes.c_int, [_ptr, _c_i64, _c_i64, _ptr, _c_i32, _c_i32, _ptr, _ptr, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr,
                                             _c_i64, _c_i32, _c_i32, _ptr, _ptr]),
    "relgnn_limb_gemm_xf32_pc": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _ptr, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _c_i32, _c_i32,
                                                _ptr, _ptr]),
    "relgnn_limb_gemm_xf32_pc_supported": (ctypes.c_int, [_c_i32, _c_i32, _c_i32, _c_i32]),
    "relgnn_limb_gemm_tn_chunks": (_c_i64, [_c_i32, _c_i32, _c_i32]),
    "relgnn_limb_gemm_tn_f32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _c_i64, _ptr, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_limb_gemm_tn_tiles_f32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _ptr, _c_i64, _ptr, _ptr, _c_i32, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_limb16_gemm_tn_f32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _c_i64, _ptr, _c_i32, _ptr, _c_i32, _ptr, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_col_absmax_workspace_bytes": (ctypes.c_int64, [_c_i32, _c_i32]),
    "relgnn_col_absmax_f32": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _c_i32, _ptr, _ptr, _c_i64, _ptr]),
    "relgnn_absmax_f32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _ptr]),
    "relgnn_limb16_elements": (_c_i64, [_c_i64, _c_i64]),
    "relgnn_limb16_split_multi_f32": (ctypes.c_int, [_c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i32, _ptr, _ptr]),
    "relgnn_limb16_gemm_xf32": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _c_i32, _c_i32,
                                               _c_i32, _ptr]),
    "relgnn_limb_split_multi_f32": (ctypes.c_int, [_c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_limb_split_batch_f32": (ctypes.c_int, [_ptr, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _ptr, _ptr]),
    "relgnn_limb_dense_sel_f32": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _ptr, _c_i64, _c_i32, _c_i64, _ptr, _c_i32, _ptr, _ptr,
                                                 _ptr, _c_i64, _ptr, _c_i64, _c_i32, _c_i32, _c_i32, _ptr]),
    "relgnn_limb_gemm_sel_xf32": (ctypes.c_int, [_c_i32, _ptr, _c_i64, _ptr, _ptr, _c_i32, _ptr, _c_i32, _ptr, _ptr, _ptr, _c_i64, _c_i32,
                                                 _c_i32, _c_i32, _ptr]),
    "relgnn_limb_gemm_sel_pc_supported": (ctypes.c_int, [_c_i32, _c_i32, _c_i32, _c_i32]),
    "relgnn_limb_gemm_sel_pc_xf32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _ptr, _c_i32, _ptr, _c_i32, _ptr, _ptr, _c_i64, _c_i32, _c_i32,
                                                    _c_i32, _ptr, _ptr]),
    "relgnn_fill_rows_f32": (ctypes.c_int, [_ptr, _c_i64, _c_i32, _ptr, _c_i64, ctypes.c_float, _ptr]),
    "relgnn_limb_dense_f32": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr, _ptr, _c_i64, _ptr, _c_i64, _c_i32,
                                             _c_i32, _c_i32, _ptr]),
    "relgnn_blaslt_gemm_f32": (ctypes.c_int, [_c_i32, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _ptr, _ptr, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32,
                                              _c_i64, _c_i64, _c_i64, _c_i32, _ptr, _c_i64, _ptr]),
    "relgnn_gemm_tn_stream_workspace_bytes": (_c_i64, [_c_i32, _c_i32, _c_i64]),
    "relgnn_gemm_tn_stream_f32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _c_i32, _c_i64, _c_i32, _ptr, _c_i64, _ptr]),
    "relgnn_gemm_tn_stream_group_workspace_bytes": (_c_i64, [_c_i32, _ptr, _ptr, _c_i64, _c_i32]),
    "relgnn_gemm_tn_stream_group_f32": (ctypes.c_int, [_c_i32, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i64, _ptr, _ptr, _c_i64,
                                                       _ptr]),
    "relgnn_gemm_tn_stream_blocks_f32": (ctypes.c_int, [_ptr, _c_i64, _ptr, _c_i64, _ptr, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, _c_i64,
                                                        _c_i32, _ptr, _c_i64, _ptr]),
    "relgnn_sum_slabs_tail_f32": (ctypes.c_int, [_ptr, _c_i32, _c_i32, _c_i32, _ptr, _c_i64, _ptr, _c_i64, _c_i32, _ptr, _ptr]),
    "relgnn_rgdcn_apply_fwd": (ctypes.c_int, [_c_i32, _c_i32, _c_i32, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _ptr, _ptr, _ptr]),
    "relgnn_rgdcn_apply_bwd": (ctypes.c_int, [_c_i32, _ptr, _ptr, _c_i64, _c_i64, _c_i64, _c_i32, _c_i32, _c_i32, _c_i32, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_batch_gather": (ctypes.c_int, [_ptr, _c_i32, _c_i32, _c_i64] + [_ptr] * 6 + [_c_i64, _c_i64, _c_i64, _c_i32, _ptr, _ptr, _ptr,
                                           _ptr, _ptr, _ptr, _ptr, _ptr, _ptr]),
    "relgnn_plan_assemble": (ctypes.c_int, [_ptr, _c_i32, _c_i32, _c_i64] + [_ptr] * 8 + [_c_i64, _c_i64] + [_ptr] * 16 + [_ptr] * 5 + [_ptr]),
    # host-side batch builder (section 9): host pointers only
    "relgnn_batch_layout_len": (_c_i64, [_c_i32, _c_i32]),
    "relgnn_batch_count": (_c_i64, [_ptr, _ptr, _c_i64, _c_i64, _c_i64]),
    "relgnn_batch_layout": (ctypes.c_int, [_c_i32, _c_i64, _ptr, _ptr, _ptr, _c_i32, _ptr, _ptr]),
    "relgnn_batch_pack": (ctypes.c_int, [_c_i32, _c_i64, _ptr, _ptr, _ptr, _ptr, _ptr, _c_i32, _ptr, _ptr, _ptr, _ptr, ctypes.c_size_t, _c_i32]),
}

_lib = None


class RelGnnLibraryError(RuntimeError):
    pass


def load_library():
    """dlopen librelgnn.so and type every entry point.  Raises if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RelGnnLibraryError(
            "%s not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (needs hipcc). There is no CPU fallback for this path." % LIB_PATH)
    lib = ctypes.CDLL(str(LIB_PATH))
    for name, (restype, argtypes) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.relgnn_abi_version() != 1:
        raise RelGnnLibraryError("librelgnn.so ABI version mismatch")
    _lib = lib
    return lib


def exported_signatures():
    return dict(_SIGNATURES)


def status_string(code: int) -> str:
    return load_library().relgnn_status_string(code).decode()


def check(code: int, what: str):
    if code == OK:
        return
    msg = "%s failed: %s (status %d)" % (what, status_string(code), code)
    if code in (EINVAL, EUNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)


def ptr(t, rows_strided: bool = False):
    """Device pointer of a tensor (None -> NULL).  Refuses anything but HIP device memory.
    rows_strided=True accepts a 2-D tensor whose rows are dense (stride(1) == 1) but whose row
    stride exceeds its width (the C ABI takes the leading dimension separately)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RelGnnLibraryError(
            "librelgnn kernels only run on MI355X device tensors; got a %s tensor. "
            "There is no CPU fallback for this path." % t.device)
    if not t.is_contiguous():
        if not (rows_strided and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
            raise ValueError("librelgnn expects contiguous tensors")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream():
    """hipStream_t of torch's current stream on the current device (an int for ctypes).  The raw accessor avoids
    building a torch.cuda.Stream object per kernel launch (~10 us each, several launches per layer)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream
