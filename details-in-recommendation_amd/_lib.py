"""_lib.py -- ctypes binding of libdir_hip.so (C ABI: include/dir_hip.h).

There is NO fallback: if the shared library is missing or fails to load, every op raises.  The library
is built in-tree by build.py (hipcc --offload-arch=gfx950).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DIR_HIP_LIBRARY: a development switch (tools/*_stress.py A/B builds of one kernel); the product loads the in-tree library
_LIB_PATH = os.environ.get("DIR_HIP_LIBRARY") or os.path.join(_HERE, "libdir_hip.so")

c_i32, c_i64, c_f32p, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p

# name -> (restype, argtypes).  Device pointers travel as c_void_p (tensor.data_ptr()).
SIGNATURES = {
    "dir_version": (c_i32, []),
    "dir_last_error": (ctypes.c_char_p, []),
    "dir_crc32c": (ctypes.c_uint32, [ctypes.c_uint32, c_vp, c_i64]),
    "dir_embedding_bag_f32": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i64,
                                      c_vp, c_i64, c_vp]),
    "dir_embedding_bag_ex_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i32, ctypes.c_float, c_i32,
                                         c_i64, c_vp, c_i64, c_vp]),
    "dir_embedding_bag_ex2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_i32, c_vp, ctypes.c_float, c_i32,
                                          c_i64, c_vp, c_i64, c_vp]),
    "dir_check_ids": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_vp]),
    "dir_fm_second_order_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp]),
    "dir_linear_onehot_rows_f32": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i32, c_i64, c_vp, c_vp]),
    "dir_sparse_ftrl_rows_sorted_f32": (c_i32, [c_vp, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_gather_fm_fused_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_gather_fm_linear_packed_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i64,
                                                c_vp, c_vp, c_vp, c_vp]),
    "dir_gather_fm_rows_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_gather_fm_rows_bits_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp,
                                            c_vp, c_vp, c_vp]),
    "dir_linear_sparse_sum_f32": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_i32, c_i64,
                                          c_vp, c_vp]),
    "dir_dcn_cross_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "dir_dcn_cross_op_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_vp]),
    "dir_dcn_cross_head_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_fm_second_order_backward_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_gated_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_dw_bf16x3_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32]),
    "dir_dense_dw_bf16x3_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "dir_dense_dw_f16x2_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_dense_dw_f16x2_scaled_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_row_absmax_workspace_words": (c_i32, []),
    "dir_row_absmax_bits_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "dir_dense_f16x2_rows_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "dir_dense_dw_small_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32]),
    "dir_dense_dw_small_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "dir_units1_relu_backward_partials": (c_i64, [c_i64, c_i32]),
    "dir_units1_relu_backward_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_units1_relu_backward_bits_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_units1_backward_partials": (c_i64, [c_i64, c_i32]),
    "dir_units1_backward_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "dir_units1_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_bn_train_partials": (c_i64, [c_i64, c_i32]),
    "dir_bn_train_stats_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, ctypes.c_float, ctypes.c_float, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                       c_i64, c_vp]),
    "dir_bn_train_backward_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64,
                                          c_vp]),
    "dir_dice_train_backward_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64,
                                            c_vp]),
    "dir_dense_affine_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_small_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_mid_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_tower_bf16x3_image_bytes": (c_i64, [c_i32, c_i32]),
    "dir_tower_bf16x3_pack_f32": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_tower_bf16x3_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_deepfm_tower_bf16x3_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                            c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_tower_f16x2_pack_f32": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_tower_f16x2_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_deepfm_tower_f16x2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                           c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_esmm_head_f32": (c_i32, [c_vp, c_vp, c_i64, ctypes.c_float, c_vp, c_vp]),
    "dir_tower_cs_image_bytes": (c_i64, [c_i32, c_i32]),
    "dir_tower_cs_f16x2_pack_f32": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_tower_cs_f16x2_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_deepfm_tower_cs_f16x2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i64, c_i32, c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp,
                                              c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_dense_f16x2_pack_strided_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_f16x2_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_bf16x3_image_bytes": (c_i64, [c_i32, c_i32]),
    "dir_dense_bf16x3_pack_strided_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_bf16x3_pack_f32": (c_i32, [c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_bf16x3_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_dense_bf16x3_head_blocks": (c_i32, [c_i32]),
    "dir_dense_bf16x3_head_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_dense_f16x2_head_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_dense_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_din_backward_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "dir_din_attention_pool_backward_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                                    c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dir_cin_dw_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i64]),
    "dir_cin_dx_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "dir_cin_dw_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "dir_dcn_cross_backward_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "dir_dcn_cross_backward_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp,
                                           c_vp, c_vp]),
    "dir_din_backward_rows_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i64]),
    "dir_din_attention_pool_backward_rows_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                                         c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                         c_vp, c_vp, c_i64, c_vp]),
    "dir_din_attention_pool_backward_saved_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                                         c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                                         c_vp, c_vp, c_i64, c_vp]),
    "dir_din_attention_pool_save_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_i64,
                                        c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_sparse_adam_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i64]),
    "dir_sparse_adam_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                    ctypes.c_float, ctypes.c_float, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp]),
    "dir_sparse_adagrad_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, ctypes.c_float, c_i64, c_vp,
                                       c_i64, c_vp, c_vp, c_vp]),
    "dir_sparse_adagrad_sorted_workspace_bytes": (c_i64, [c_i64, c_i32, c_i32, c_i64]),
    "dir_sparse_adagrad_sorted_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, ctypes.c_float, c_i64,
                                              c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_sparse_adagrad_sorted_rows_f32": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp,
                                                   ctypes.c_float, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_sparse_adagrad_sorted_payload_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, ctypes.c_float, c_vp, c_i64, c_vp,
                                                      c_i64, c_vp]),
    "dir_sparse_ftrl_sorted_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, ctypes.c_float,
                                           ctypes.c_float, ctypes.c_float, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_adagrad_dense_f32": (c_i32, [c_vp, c_vp, c_vp, c_i64, ctypes.c_float, ctypes.c_float, c_vp]),
    "dir_adagrad_dense_multi_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, ctypes.c_float, ctypes.c_float, c_vp]),
    "dir_ftrl_dense_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_vp]),
    "dir_sparse_adagrad_sorted_rows_from_f32": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_vp, c_vp,
                                                        ctypes.c_float, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_sparse_ftrl_sorted_from_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_din_attention_pool_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                           c_i32, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "dir_din_activation_rows_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp]),
    "dir_din_feat_rows_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_din_feat_rows_backward_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp]),
    "dir_act_rows_train_f32": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_act_rows_backward_partials": (c_i32, [c_i64, c_i32]),
    "dir_act_rows_backward_f32": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i64, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                                          c_i32, c_vp]),
    "dir_din_pool_rows_f32": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "dir_din_pool_rows_backward_f32": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i64, c_i64, c_i32, c_vp, c_vp, c_vp]),
    "dir_din_attention_pool_act_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                               c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_din_attention_pool_arith_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                                 c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "dir_din_pack_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "dir_din_pack_image_bytes": (c_i64, []),
    "dir_din_pack_weights_f32": (c_i32, [c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp]),
    "dir_din_attention_pool_packed_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp,
                                                  c_i32, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_din_attention_pool_save_arith_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32, c_vp, c_vp, c_i32,
                                                      c_i32, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_layer_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp]),
    "dir_cin_bf16x3_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32]),
    "dir_cin_layer_bf16x3_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_pool_z_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp]),
    "dir_cin_pooled_image_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "dir_cin_pooled_pack_f32": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp]),
    "dir_cin_pooled_last_bf16x3_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_pooled_last_bf16x3_gather_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_layer1_f16x2_gather_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_cin_layer_f16x2_gather_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_cin_pool_z_bits_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp]),
    "dir_cin_pool_dx_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_i32, c_vp]),
    "dir_cin_layer1_bf16x3_workspace_bytes": (c_i64, [c_i32, c_i32]),
    "dir_cin_layer1_bf16x3_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_layer_f16x2_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_layer1_f16x2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp]),
    "dir_cin_layer1_bits_f16x2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_cin_bf16x3_dot_partials": (c_i32, [c_i32, c_i32, c_i32]),
    "dir_cin_dw_bf16x3_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i64]),
    "dir_cin_dw_bf16x3_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp]),
    "dir_cin_dw_f16x2_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i32, c_i64]),
    "dir_cin_dw_f16x2_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_cin_layer_grad_f16x2_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp]),
    "dir_cin_layer_rows_f16x2_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_cin_layer_auto_f16x2_f32": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp]),
    "dir_cin_dw_sym_f16x2_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_cin_layer_dot_add_f16x2_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_cin_layer_dot_add_bf16x3_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_sum_partials_f32": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "dir_cin_dw_sym_bf16x3_workspace_bytes": (c_i64, [c_i32, c_i32, c_i32, c_i64]),
    "dir_cin_dw_sym_bf16x3_f32": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp, c_vp, c_i64, c_vp]),
    "dir_cin_layer_dot_bf16x3_f32": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_debug_cin_stamps": (c_i32, [ctypes.POINTER(ctypes.c_uint64)]),
    "dir_debug_stream_read_f32": (c_i32, [c_vp, c_i64, c_vp, c_vp]),
    "dir_debug_stream_copy_f32": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "dir_debug_radix_sort_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "dir_debug_radix_sort_pairs_u32": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_debug_slot_sort_workspace_bytes": (c_i64, [c_i64, c_i32, c_i64]),
    "dir_debug_slot_sort_entries": (c_i32, [c_vp, c_i64, c_i64, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "dir_fingerprint64": (ctypes.c_uint64, [ctypes.c_char_p, c_i64]),
    "dir_hash_bucket_fast": (c_i32, [ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(c_i64), c_i64, c_i64,
                                     ctypes.POINTER(c_i64)]),
    "dir_hash_bucket_i64_device": (c_i32, [c_vp, c_i64, c_i64, c_vp, c_vp]),
    "dir_hash_bucket_i64_fields_device": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "dir_hash_bucket_bytes_device": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_vp, c_vp]),
    "dir_bucketize_f32": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_vp, c_vp]),
    "dir_shard_div_owner": (None, [c_i64, c_i64, c_i32, ctypes.POINTER(c_i32), ctypes.POINTER(c_i64)]),
    "dir_shard_route": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "dir_gather_rows_f32": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_i64, c_vp, c_vp]),
    "dir_shard_bucket_workspace_bytes": (c_i64, [c_i64, c_i32]),
    "dir_shard_bucket": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dir_shard_bucket_cap_workspace_bytes": (c_i64, [c_i32]),
    "dir_shard_bucket_cap": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dir_shard_bucket_cap_dedup": (c_i32, [c_vp, c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "dir_shard_slab_stat": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_vp]),
    "dir_gather_slabs_f32": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i32, c_i64, c_i32, c_vp, c_vp]),
    "dir_gather_packed_f32": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp, c_vp]),
}

DIR_OK, DIR_E_BADARG, DIR_E_RANGE, DIR_E_HIP, DIR_E_UNSUPPORTED = 0, -1, -2, -3, -4
_ERRNAMES = {-1: "DIR_E_BADARG", -2: "DIR_E_RANGE", -3: "DIR_E_HIP", -4: "DIR_E_UNSUPPORTED"}

_lib = None


class DirError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (_ERRNAMES.get(code, "DIR_E_?"), code, msg))
        self.code = code


def library_path():
    return _LIB_PATH


def load():
    """Load libdir_hip.so and bind every symbol of include/dir_hip.h.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise RuntimeError(
            "libdir_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`python details-in-recommendation_amd/build.py`. There is no CPU fallback." % _LIB_PATH)
    # One HIP runtime per process: torch's wheel carries its own libamdhip64 (same SONAME as /opt/rocm's), and the tensors and
    # streams handed to the C ABI are torch's.  Map torch's copy first so libdir_hip.so binds to it; loaded the other way
    # round the process ends up with /opt/rocm's runtime under torch's HSA and the first launch reports "no ROCm-capable device".
    import torch  # noqa: F401
    lib = ctypes.CDLL(_LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI drifted
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != DIR_OK:
        raise DirError(rc, load().dir_last_error().decode("utf-8", "replace"))
