"""ctypes binding of librecengine.so — the only door from Python into the HIP kernels.

There is no CPU fallback: if the library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librecengine.so")

REC_FLAG_INDEX_OOB = 1
REC_FLAG_EXCHANGE_OVERFLOW = 2


class RecError(RuntimeError):
    pass


class DeepFMDesc(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_slots", C.c_int32), ("num_dense", C.c_int32),
                ("emb_dim", C.c_int32), ("row_stride", C.c_int32), ("num_rows", C.c_int64),
                ("padding_idx", C.c_int64), ("w1_stride", C.c_int32), ("compact_dense", C.c_int32),
                ("feat_stride", C.c_int64)]


class DeepFMNet(C.Structure):
    """rec_deepfm_net (include/recengine.h): the model of rec_deepfm_train_step as pointers into caller-owned memory."""
    MAX_LINEAR = 8
    _fields_ = [("num_slots", C.c_int32), ("dim", C.c_int32), ("dense_dim", C.c_int32), ("n_linear", C.c_int32),
                ("widths", C.c_int32 * 8), ("table_rows", C.c_int64), ("num_rows", C.c_int64),
                ("padding_idx", C.c_int64), ("slot_rows", C.c_int64), ("slot_offset", C.c_void_p),
                ("rec", C.c_void_p), ("rec_stride", C.c_int32), ("mv", C.c_void_p), ("mv_stride", C.c_int32),
                ("v_offset", C.c_int32), ("dense_w", C.c_void_p), ("dense_w_one", C.c_void_p),
                ("g_dense_w", C.c_void_p), ("g_dense_w_one", C.c_void_p), ("w", C.c_void_p * 8), ("b", C.c_void_p * 8),
                ("gw", C.c_void_p * 8), ("gb", C.c_void_p * 8), ("flat_param", C.c_void_p), ("flat_grad", C.c_void_p),
                ("flat_m", C.c_void_p), ("flat_v", C.c_void_p), ("flat_numel", C.c_int64), ("w0_folded", C.c_void_p),
                ("layer0_width", C.c_int32)]


class DinNet(C.Structure):
    """rec_din_net (include/recengine.h): the model of rec_din_train_step as pointers into caller-owned memory."""
    _fields_ = ([("item_dim", C.c_int32), ("cat_dim", C.c_int32), ("item_rows", C.c_int64), ("cat_rows", C.c_int64),
                 ("att_hidden1", C.c_int32), ("att_hidden2", C.c_int32), ("mlp_hidden1", C.c_int32),
                 ("mlp_hidden2", C.c_int32)] +
                [(n, C.c_void_p) for n in (
                    "w_hist_item", "w_hist_cat", "w_tgt_item_seq", "w_tgt_cat_seq", "w_tgt_item", "w_tgt_cat", "w_item_b",
                    "att_w1", "att_w1_t", "att_b1", "att_w2", "att_b2", "att_w3", "att_b3",
                    "w_con", "b_con", "w_l0", "b_l0", "w_l1", "b_l1", "w_l2", "b_l2",
                    "g_w_con", "g_b_con", "g_w_l0", "g_b_l0", "g_w_l1", "g_b_l1", "g_w_l2", "g_b_l2",
                    "flat_param", "flat_grad")] + [("flat_numel", C.c_int64)])


DCN_MAX_LAYERS = 8


class DcnV2Net(C.Structure):
    """rec_dcn_v2_net (include/recengine.h): the model of rec_dcn_v2_train_step as pointers into caller-owned memory."""
    _fields_ = ([("num_slots", C.c_int32), ("dim", C.c_int32), ("dense_dim", C.c_int32), ("num_rows", C.c_int64),
                 ("padding_idx", C.c_int64), ("emb_stride", C.c_int32), ("state_stride", C.c_int32),
                 ("emb", C.c_void_p), ("emb_m", C.c_void_p), ("emb_v", C.c_void_p),
                 ("cross_num", C.c_int32), ("n_dnn", C.c_int32), ("widths", C.c_int32 * DCN_MAX_LAYERS),
                 ("is_stacked", C.c_int32), ("low_rank_mix", C.c_int32), ("num_experts", C.c_int32),
                 ("low_rank", C.c_int32), ("dropout_rate", C.c_float), ("l2_dnn", C.c_float), ("clip_norm", C.c_float),
                 ("dropout_seed", C.c_uint64)] +
                [(n, C.c_void_p) for n in ("dense_emb_w", "dense_emb_b", "g_dense_emb_w", "g_dense_emb_b")] +
                [(n, C.c_void_p * DCN_MAX_LAYERS) for n in ("cross_w", "cross_b", "g_cross_w", "g_cross_b", "mix_u", "mix_v", "mix_c",
                                                "mix_bias", "g_mix_u", "g_mix_v", "g_mix_c", "g_mix_bias")] +
                [(n, C.c_void_p) for n in ("gate_w", "gate_b", "g_gate_w", "g_gate_b")] +
                [(n, C.c_void_p * DCN_MAX_LAYERS) for n in ("dnn_w", "dnn_b", "g_dnn_w", "g_dnn_b")] +
                [(n, C.c_void_p) for n in ("fc_w", "fc_b", "g_fc_w", "g_fc_b", "flat_param", "flat_grad", "flat_m",
                                           "flat_v")] + [("flat_numel", C.c_int64)])


class AdamHyper(C.Structure):
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int64)]


class AdagradHyper(C.Structure):
    _fields_ = [("lr", C.c_float), ("initial_g2sum", C.c_float), ("min_bound", C.c_float),
                ("max_bound", C.c_float)]


class GradLayout(C.Structure):
    _fields_ = [("div", C.c_int32), ("group", C.c_int32), ("group_stride", C.c_int64),
                ("partials", C.c_void_p), ("index", C.c_void_p), ("sorted", C.c_int32)]


class SmallSgdJob(C.Structure):
    _fields_ = [("n", C.c_int64), ("emb_dim", C.c_int32), ("row_stride", C.c_int32), ("num_rows", C.c_int64),
                ("padding_idx", C.c_int64), ("ids", C.c_void_p), ("grad", C.c_void_p), ("grad_layout", GradLayout),
                ("P", C.c_void_p)]


class CinView(C.Structure):
    _fields_ = [("stride_b", C.c_int64), ("stride_j", C.c_int32), ("stride_d", C.c_int32)]


class PsAccessor(C.Structure):
    _fields_ = [("lr", C.c_float), ("initial_g2sum", C.c_float), ("min_bound", C.c_float), ("max_bound", C.c_float),
                ("initial_range", C.c_float),
                ("x_lr", C.c_float), ("x_initial_g2sum", C.c_float), ("x_min_bound", C.c_float),
                ("x_max_bound", C.c_float), ("x_initial_range", C.c_float),
                ("embedx_threshold", C.c_float), ("nonclk_coeff", C.c_float), ("click_coeff", C.c_float),
                ("grad_scale", C.c_float), ("show_scale", C.c_int32), ("embed_zero_init", C.c_int32),
                ("seed", C.c_uint64), ("row_mul", C.c_int64), ("row_add", C.c_int64)]


class LazyInit(C.Structure):
    _fields_ = [("state_offset", C.c_int32), ("init_dims", C.c_int32), ("init_range", C.c_float),
                ("seed", C.c_uint64), ("row_mul", C.c_int64), ("row_add", C.c_int64)]


class PsLayout(C.Structure):
    _fields_ = [("row_stride", C.c_int32), ("embed_off", C.c_int32), ("embedx_off", C.c_int32),
                ("embedx_dim", C.c_int32), ("stat_off", C.c_int32)]


class GradSrc(C.Structure):
    _fields_ = [("grad", C.c_void_p), ("layout", GradLayout), ("pitch", C.c_int32), ("col", C.c_int32)]


class MultislotDesc(C.Structure):
    _fields_ = [("batch", C.c_int64), ("num_slots", C.c_int32), ("emb_dim", C.c_int32),
                ("row_stride", C.c_int32), ("key_mode", C.c_int32), ("num_rows", C.c_int64),
                ("padding_idx", C.c_int64), ("lod_stride", C.c_int64), ("out_stride", C.c_int64),
                ("state_offset", C.c_int32), ("init_dims", C.c_int32), ("init_range", C.c_float),
                ("init_seed", C.c_uint64)]


class DinDesc(C.Structure):
    _fields_ = [("batch", C.c_int64), ("max_len", C.c_int32), ("item_dim", C.c_int32),
                ("cat_dim", C.c_int32), ("hidden1", C.c_int32), ("hidden2", C.c_int32),
                ("item_rows", C.c_int64), ("cat_rows", C.c_int64), ("item_stride", C.c_int32),
                ("cat_stride", C.c_int32)]


class CrossV2Desc(C.Structure):
    _fields_ = [("batch", C.c_int64), ("d", C.c_int32), ("ld_x0", C.c_int32), ("ld_xl", C.c_int32),
                ("ld_out", C.c_int32), ("ld_u", C.c_int32)]


class CrossMixDesc(C.Structure):
    _fields_ = [("batch", C.c_int64), ("d", C.c_int32), ("rank", C.c_int32), ("experts", C.c_int32),
                ("ld_x0", C.c_int32), ("ld_xl", C.c_int32), ("ld_out", C.c_int32)]


class GemmDesc(C.Structure):
    _fields_ = [("m", C.c_int64), ("n", C.c_int32), ("k", C.c_int32), ("lda", C.c_int32),
                ("ldb", C.c_int32), ("ldc", C.c_int32), ("trans_a", C.c_int32),
                ("trans_b", C.c_int32), ("epilogue", C.c_int32), ("split_k", C.c_int32), ("num_cus", C.c_int32)]


class GemmEpilogueArgs(C.Structure):
    _fields_ = [("bias", C.c_void_p), ("aux0", C.c_void_p), ("ld_aux0", C.c_int32),
                ("aux1", C.c_void_p), ("ld_aux1", C.c_int32), ("row_scale", C.c_void_p),
                ("row_scale_stride", C.c_int32), ("out2", C.c_void_p), ("ld_out2", C.c_int32),
                ("b_colsum", C.c_void_p), ("b_image", C.c_void_p), ("relu_bits", C.c_void_p)]


class GemmBImage(C.Structure):
    _fields_ = [("B", C.c_void_p), ("ldb", C.c_int32), ("k", C.c_int32), ("n", C.c_int32), ("trans_b", C.c_int32),
                ("image", C.c_void_p)]


EPI = dict(none=0, bias=1, bias_relu=2, relu_mask=3, cross=4, bias_sigmoid=5, bias_tanh=6, add=7, moe=8,
           dsigmoid=9, dtanh=10)

_P = C.c_void_p
_I64, _I32, _F, _SZ = C.c_int64, C.c_int32, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/recengine.h declares
SIGNATURES = {
    "rec_last_error": (C.c_char_p, []),
    "rec_version": (C.c_int, []),
    "rec_deepfm_fm_fwd": (C.c_int, [C.POINTER(DeepFMDesc)] + [_P] * 13),
    "rec_deepfm_fm_bwd_workspace_bytes": (C.c_int, [C.POINTER(DeepFMDesc), C.POINTER(_SZ)]),
    "rec_deepfm_fm_bwd": (C.c_int, [C.POINTER(DeepFMDesc)] + [_P] * 11 + [_SZ, _P]),
    "rec_deepfm_fm_bwd_sorted": (C.c_int, [C.POINTER(DeepFMDesc)] + [_P] * 12 + [_SZ, _P]),
    "rec_dense_fold_fwd": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "rec_dense_fold_bwd": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "rec_dense_fold_fwd_full": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    "rec_dense_fold_bwd_full": (C.c_int, [_I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "rec_emb_gather": (C.c_int, [_I64, _I32, _I32, _I64, _I64, _P, _P, _P, _I32, _I64, _P, _P]),
    "rec_emb_gather_sumpool": (C.c_int, [_I64, _I32, _I32, _I64, _I64, _P, _P, _P, _P, _P, _P, _P]),
    "rec_emb_sumpool_bwd": (C.c_int, [_I64, _I32, _P, _P, _P, _P]),
    "rec_ids_group_workspace_bytes": (C.c_int, [_I64, _I64, C.POINTER(_SZ)]),
    "rec_ids_group": (C.c_int, [_I64, _I32, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rec_ids_group_payload": (C.c_int, [_I64, _I32, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rec_ids_group_slots_workspace_bytes": (C.c_int, [_I64, _I32, _I64, C.POINTER(_SZ)]),
    "rec_ids_group_slots": (C.c_int, [_I64, _I32, _I64, _I64, _P, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rec_ids_rank": (C.c_int, [_I64, _P, _P, _P, _P]),
    "rec_segment_partials_bytes": (C.c_int, [_I64, _I32, C.POINTER(C.c_size_t)]),
    "rec_segment_partials": (C.c_int, [_I64, _I32, _P, _P, _P, _P, C.POINTER(GradLayout), _P, _P]),
    "rec_sparse_adam_rows": (C.c_int, [_I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(GradLayout), _P,
                                       _P, _P, _P, C.POINTER(AdamHyper), _P]),
    "rec_sparse_adam_record": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(GradLayout), _P,
                                         C.POINTER(GradLayout), _P, _P, _P, C.POINTER(AdamHyper), _P]),
    "rec_adam_record_all": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(GradLayout), _P,
                                      C.POINTER(GradLayout), _P, _P, _P, C.POINTER(AdamHyper), _P]),
    "rec_multislot_sumpool_fwd": (C.c_int, [C.POINTER(MultislotDesc)] + [_P] * 10),
    "rec_multislot_sumpool_bwd": (C.c_int, [C.POINTER(MultislotDesc), _I64, _P, _P, _P, _P]),
    "rec_feasign_rows": (C.c_int, [_I64, _I64, _P, _P, _P]),
    "rec_feasign_rows_host": (C.c_int, [_I64, _I64, _P, _P]),
    "rec_record_gather": (C.c_int, [_I64, _I32, _I32, _I64, _P, _P, _P, _P, C.POINTER(LazyInit), _P, _P]),
    "rec_dedup_plan_workspace_bytes": (C.c_int, [_I64, _I32, _I64, C.POINTER(_SZ)]),
    "rec_dedup_plan": (C.c_int, [_I64, _I32, _I64, _I64, _I32, _I64, _I32] + [_P] * 11 + [_P, _SZ, _P]),
    "rec_link_emulate": (C.c_int, [_SZ, _F, _F, _I32, _P, _P, _SZ, _P]),
    "rec_comm_unique_id": (C.c_int, [_P]),
    "rec_comm_init": (C.c_int, [_P, _I32, _I32, C.POINTER(C.c_void_p)]),
    "rec_comm_destroy": (C.c_int, [_P]),
    "rec_comm_size": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "rec_comm_available": (C.c_int, []),
    "rec_alltoall_exchange": (C.c_int, [_P, _P, C.POINTER(_I64), _P, C.POINTER(_I64), _I32, _P]),
    "rec_allreduce_sum_f32": (C.c_int, [_P, _P, _I64, _P]),
    "rec_ps_push_rows": (C.c_int, [_I64, _I32, C.POINTER(PsLayout), _P, _P, _P, _P, C.POINTER(GradSrc),
                                   C.POINTER(GradSrc), _P, _P, _P, C.POINTER(PsAccessor), _P]),
    "rec_ps_init_value_host": (C.c_float, [C.c_uint64, _I64, _I32, _F]),
    "rec_ps_shrink_rows": (C.c_int, [_I64, C.POINTER(PsLayout), _P, _F, _F, _F, C.POINTER(PsAccessor), _P, _P]),
    "rec_ps_save_select": (C.c_int, [_I64, C.POINTER(PsLayout), _P, _I32, _F, _F, _F, C.POINTER(PsAccessor), _P, _P,
                                     _P]),
    "rec_sparse_adagrad_rows": (C.c_int, [_I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(GradLayout), _P,
                                          _P, C.POINTER(AdagradHyper), _P]),
    "rec_adam_rows_all": (C.c_int, [_I64, _I32, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(GradLayout), _P,
                                    _P, _P, _P, C.POINTER(AdamHyper), _P]),
    "rec_adam_dense": (C.c_int, [_I64, _P, _P, _P, _P, _P, C.POINTER(AdamHyper), _P]),
    "rec_sumsq_workspace_bytes": (C.c_int, [C.POINTER(_SZ)]),
    "rec_sumsq": (C.c_int, [_I64, _P, _P, _I32, _P, _SZ, _P]),
    "rec_sparse_rows_sumsq": (C.c_int, [_I64, _I32, _P, _P, _P, _P, C.POINTER(GradLayout), _P, _I32, _P,
                                        _SZ, _P]),
    "rec_clip_scale": (C.c_int, [_P, _F, _P, _P]),
    "rec_mlp_head_bwd_workspace_bytes": (C.c_int, [_I64, _I32, C.POINTER(C.c_size_t)]),
    "rec_mlp_head_bwd": (C.c_int, [_I64, _I32, _P, _I64, _P, _P, _I32, _P, _I64, _P, _P, _P, _SZ, _P]),
    "rec_dropout": (C.c_int, [_I64, _I32, _I64, _I64, _P, _P, _F, C.c_uint64, C.c_uint64, C.c_uint64, _I32, _P]),
    "rec_l2_decay_grad": (C.c_int, [_I64, _P, _P, _F, _P, _P]),
    "rec_din_saves_act1": (C.c_int, [C.POINTER(DinDesc)]),
    "rec_din_attention_pool_fwd": (C.c_int, [C.POINTER(DinDesc)] + [_P] * 20),
    "rec_din_attention_pool_fwd_workspace_bytes": (C.c_int, [C.POINTER(DinDesc), C.POINTER(C.c_size_t)]),
    "rec_din_attention_pool_fwd_ws": (C.c_int, [C.POINTER(DinDesc)] + [_P] * 20 + [_SZ, _P]),
    "rec_din_attention_pool_bwd": (C.c_int, [C.POINTER(DinDesc)] + [_P] * 21),
    "rec_din_attention_pool_bwd_workspace_bytes": (C.c_int, [C.POINTER(DinDesc), C.POINTER(C.c_size_t)]),
    "rec_din_attention_pool_bwd_ws": (C.c_int, [C.POINTER(DinDesc)] + [_P] * 21 + [_SZ, _P]),
    "rec_sparse_sgd_rows": (C.c_int, [_I64, _I32, _I32, _P, _P, _P, _P, _P, C.POINTER(GradLayout), _P, _F, _P]),
    "rec_sgd_dense": (C.c_int, [_I64, _P, _P, _F, _P]),
    "rec_crossnet_v2_layer_workspace_bytes": (C.c_int, [C.POINTER(CrossV2Desc), C.POINTER(_SZ), C.POINTER(_SZ)]),
    "rec_crossnet_v2_layer_fwd": (C.c_int, [C.POINTER(CrossV2Desc), _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rec_crossnet_v2_layer_bwd": (C.c_int, [C.POINTER(CrossV2Desc), _P, _P, _P, _P, _P, _I32, _P, _I32, _I32, _I32, _P,
                                            _I32, _P, _P, _P, _SZ, _P]),
    "rec_crossnet_mix_layer_workspace_bytes": (C.c_int, [C.POINTER(CrossMixDesc), C.POINTER(_SZ), C.POINTER(_SZ)]),
    "rec_crossnet_mix_layer_fwd": (C.c_int, [C.POINTER(CrossMixDesc)] + [_P] * 13 + [_SZ, _P]),
    "rec_crossnet_mix_layer_bwd": (C.c_int, [C.POINTER(CrossMixDesc)] + [_P] * 11 + [_I32, _P, _I32, _I32, _I32, _P, _I32]
                                   + [_P] * 6 + [_I32, _P, _SZ, _P]),
    "rec_sparse_sgd_small": (C.c_int, [_I64, _I32, _I32, _I64, _I64, _P, _P, C.POINTER(GradLayout), _P, _F, _P, _P]),
    "rec_sparse_sgd_small_multi": (C.c_int, [_I32, C.POINTER(SmallSgdJob), _F, _P, _P]),
    "rec_sparse_adam_record_small": (C.c_int, [_I64, _I32, _I32, _I32, _I32, _I32, _I64, _I64, _P, _P, _P, C.POINTER(GradLayout),
                                               _P, C.POINTER(GradLayout), _P, _P, _P, C.POINTER(AdamHyper), _P, _P, _P]),
    "rec_bce_with_logits": (C.c_int, [_I64, _I64, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "rec_moe_bwd_prep": (C.c_int, [_I64, _I32, _P, _I32, _P, _I32, _P, _I32, _P, _I32, _P, _I32, _P, _I32, _I32,
                                   _P, _I32, _P]),
    "rec_softmax_rows_bwd": (C.c_int, [_I64, _I32, _P, _I32, _P, _I32, _P, _I32, _P]),
    "rec_softmax_rows": (C.c_int, [_I64, _I32, _P, _I32, _P, _I32, _P]),
    "rec_batchnorm_workspace_bytes": (C.c_int, [_I64, _I32, C.POINTER(_SZ)]),
    "rec_batchnorm_fwd": (C.c_int, [_I64, _I32, _P, _I64, _P, _P, _P, _P, _F, _F, _I32, _P, _I64, _P, _P, _P, _SZ, _P]),
    "rec_batchnorm_bwd": (C.c_int, [_I64, _I32, _P, _I64, _P, _I64, _P, _P, _P, _I32, _P, _I64, _P, _P, _P, _SZ, _P]),
    "rec_dot_interact_fwd": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _I64, _P]),
    "rec_dot_interact_bwd": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _I64, _P, _I64, _P]),
    "rec_accuracy_count": (C.c_int, [_I64, _P, _P, _P, _P]),
    "rec_cin_outer_fwd": (C.c_int, [_I64, _I32, _I32, _I32, _P, C.POINTER(CinView), _P, C.POINTER(CinView), _P, _I64, _P]),
    "rec_cin_outer_bwd": (C.c_int, [_I64, _I32, _I32, _I32, _P, _I64, _P, C.POINTER(CinView), _P, C.POINTER(CinView),
                                    _P, C.POINTER(CinView), _I32, _P, C.POINTER(CinView), _I32, _P, _I64, _P]),
    "rec_cin_contract_fwd": (C.c_int, [_I64, _I32, _I32, _I32, _P, _I64, _P, C.POINTER(CinView), _P, _I64, _P]),
    "rec_cin_contract_bwd": (C.c_int, [_I64, _I32, _I32, _I32, _P, _I64, _P, _I64, _P, C.POINTER(CinView), _P, _I64,
                                       _P, C.POINTER(CinView), _I32, _P]),
    "rec_cin_sumpool": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _I64, _P]),
    "rec_cin_sumpool_bwd": (C.c_int, [_I64, _I32, _I32, _P, _I64, _P, _I64, _P]),
    "rec_cross_bwd_prep": (C.c_int, [_I64, _I32, _P, _I32, _P, _I32, _P, _I32, _P, _I32, _P, _I32, _I32, _P]),
    "rec_logloss_workspace_bytes": (C.c_int, [_I64, C.POINTER(_SZ)]),
    "rec_sigmoid_logloss": (C.c_int, [_I64, _I64, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _SZ, _P]),
    "rec_auc_histogram": (C.c_int, [_I64, _P, _P, _I32, _P, _P, _P]),
    "rec_shard_route_workspace_bytes": (C.c_int, [_I64, _I32, C.POINTER(_SZ)]),
    "rec_shard_route": (C.c_int, [_I64, _I32, _I64, _I64, _I32] + [_P] * 9 + [_SZ, _P]),
    "rec_gemm_f32_workspace_bytes": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(_SZ)]),
    "rec_gemm_plan_splits": (C.c_int, [C.POINTER(GemmDesc), _I32, C.POINTER(_I32)]),
    "rec_gemm_relu_bits_bytes": (C.c_int, [C.POINTER(GemmDesc), C.POINTER(_I32), C.POINTER(_SZ)]),
    "rec_gemm_b_image_bytes": (C.c_int, [_I32, _I32, C.POINTER(_I32), C.POINTER(_SZ)]),
    "rec_gemm_b_images": (C.c_int, [_I32, C.POINTER(GemmBImage), _P]),
    "rec_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, C.POINTER(GemmEpilogueArgs), _P, _SZ, _P]),
    "rec_gemm_f32_pair": (C.c_int, [C.POINTER(GemmDesc), _P, _P, _P, C.POINTER(GemmEpilogueArgs), C.POINTER(GemmDesc), _P, _P,
                                    _P, C.POINTER(GemmEpilogueArgs), _P, _SZ, _P]),
    "rec_colsum_workspace_bytes": (C.c_int, [_I64, _I32, C.POINTER(_SZ)]),
    "rec_colsum": (C.c_int, [_I64, _I32, _I32, _P, _P, _P, _SZ, _P]),
    "rec_xxh32": (C.c_uint32, [C.c_char_p, _SZ, C.c_uint32]),
    "rec_xxh32_hash_mod": (C.c_int, [C.POINTER(C.c_char_p), C.POINTER(_I32), _I64, C.c_uint32,
                                     C.POINTER(_I64)]),
    "rec_count_lines": (C.c_int, [C.c_char_p, _SZ, _I32, C.POINTER(_I64)]),
    "rec_blank_lines": (C.c_int, [C.c_char_p, _SZ, _I32, _I64, _P, C.POINTER(_I64)]),
    "rec_count_byte": (C.c_int, [C.c_char_p, _SZ, _I32, _I32, C.POINTER(_I64)]),
    "rec_csr_cut": (C.c_int, [_I32, _I32, _P, _P, _P, _P, _P, _P, _I32, _P, _I64, _P, _P]),
    "rec_parse_slot_text": (C.c_int, [C.c_char_p, _SZ, _I32, _I32, _I32, _I64, _I32, _P, _P, _P,
                                      C.POINTER(_I64)]),
    "rec_parse_criteo_tsv": (C.c_int, [C.c_char_p, _SZ, _I32, _I32, _P, _P, C.c_uint32, _I64, _I32, _P, _P, _P,
                                       C.POINTER(_I64)]),
    "rec_parse_feasign_slots": (C.c_int, [C.c_char_p, _SZ, _I32, _I32, C.c_uint64, _I64, _I64, _I32, _P, _P, _P,
                                          C.POINTER(_I64), C.POINTER(_I64)]),
    "rec_fill_uniform": (C.c_int, [_I64, _P, _F, _F, C.c_uint64, _P]),
    "rec_stream_spin": (C.c_int, [_I32, _P]),
    "rec_copy_async": (C.c_int, [_P, _P, _SZ, _P]),
    "rec_copy_2d_async": (C.c_int, [_P, _SZ, _P, _SZ, _SZ, _SZ, _P]),
    "rec_transpose_f32": (C.c_int, [_I64, _I64, _P, _P, _P]),
    "rec_cast_f32_i64": (C.c_int, [_I64, _P, _I64, _P, _P]),
    "rec_stream_create_cu_range": (C.c_int, [_I32, _I32, C.POINTER(C.c_void_p)]),
    "rec_stream_create_cu_stride": (C.c_int, [_I32, _I32, _I32, C.POINTER(C.c_void_p)]),
    "rec_ctr_head_workspace_bytes": (C.c_int, [_I64, _I32, C.POINTER(C.c_size_t)]),
    "rec_ctr_head_fwd_bwd": (C.c_int, [_I64, _I32, _I64, _P, _I64, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float,
                                       _I32, _P, _P, _P, _P, _P, _I64, _P, _P, _P, _SZ, _P]),
    "rec_din_train_step_workspace_bytes": (C.c_int, [C.POINTER(DinNet), _I64, _I32, C.POINTER(C.c_size_t)]),
    "rec_din_train_step": (C.c_int, [C.POINTER(DinNet), _I64, _I32, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                                     _SZ, _P, _P]),
    "rec_dcn_v2_train_step_workspace_bytes": (C.c_int, [C.POINTER(DcnV2Net), _I64, C.POINTER(C.c_size_t)]),
    "rec_dcn_v2_train_step": (C.c_int, [C.POINTER(DcnV2Net), _I64, _P, _P, _P, C.POINTER(AdamHyper), _P, _P, _I32, _P, _P,
                                        _P, _P, _SZ, _P]),
    "rec_deepfm_train_step_workspace_bytes": (C.c_int, [C.POINTER(DeepFMNet), _I64, C.POINTER(C.c_size_t)]),
    "rec_deepfm_train_step": (C.c_int, [C.POINTER(DeepFMNet), _I64, _P, _P, _P, C.POINTER(AdamHyper), _P, _P, _I32, _P, _P,
                                        _P, _P, _SZ, _P, _P]),
    "rec_stream_destroy": (C.c_int, [_P]),
}

_lib = None


def lib():
    """Load librecengine.so (once).  Raises RecError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RecError(
                "librecengine.so not found at %s — run `python -m paddlerec_amd.build` "
                "(there is no CPU fallback)" % LIB_PATH)
        # PyTorch-ROCm wheels bundle their own libamdhip64; it must be the HIP runtime of the process.  If
        # librecengine.so were loaded first it would pull in /opt/rocm's copy and the process would end up
        # with two runtimes ("no ROCm-capable device is detected" on the first launch) — so torch goes first.
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().rec_last_error().decode("utf-8", "replace")
        raise RecError("%s failed (rc=%d): %s" % (what or "recengine call", rc, msg))
