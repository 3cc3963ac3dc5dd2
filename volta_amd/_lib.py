"""ctypes binding of libvolta_hip.so (include/volta_hip.h).  There is no CPU fallback: if the HIP
library is missing the import fails loudly (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C volta_amd/csrc`)."""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: libvolta_hip.so has to bind to the HIP runtime torch already loaded

_HERE = os.path.dirname(os.path.abspath(__file__))
# VK_LIB=study loads the measurement build (tools/ only: ablation switches, geometry overrides, the 4-phase study kernel); any other
# name loads libvolta_hip_<name>.so, a second build of the same sources for an A/B in one box session (tools/ab_bench.sh)
LIB_PATH = os.path.join(_HERE, "libvolta_hip_%s.so" % os.environ["VK_LIB"] if os.environ.get("VK_LIB") else "libvolta_hip.so")
GEMM_PERSISTENT, GEMM_ONE_TILE_PER_WG, GEMM_SOFT_START = 0x1000, 0x2000, 0x4000


class VoltaHipError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "volta_amd: %s not found. The HIP kernels are the product; there is no fallback path. "
        "Build with `make -C volta_amd/csrc` (hipcc --offload-arch=gfx950)." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

c_p = C.c_void_p
i32 = C.c_int32
CHUNK_SKIP = 255          # VK_CHUNK_SKIP


class Dropout(C.Structure):
    _fields_ = [("seed", c_p), ("site", C.c_uint32), ("threshold", C.c_uint32), ("scale", C.c_float)]


def dropout_cfg(seed_ptr, site, p):
    if p <= 0.0 or not seed_ptr:
        return Dropout(None, 0, 0, 1.0)
    return Dropout(seed_ptr, site, min(int(p * 4294967296.0), 0xFFFFFFFF), 1.0 / (1.0 - p))


def rng_cfg(seed_ptr, site):
    """A counter-based stream without a keep threshold (vk_nce_negatives): the seed word and the site id only."""
    return Dropout(seed_ptr, site, 0, 1.0)


class GemmProblem(C.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("C", c_p), ("C2", c_p), ("bias", c_p), ("R", c_p), ("bias_grad", c_p),
                ("dyn", c_p), ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32), ("ldc", i32),
                ("ldr", i32), ("n_store", i32), ("ws", c_p), ("cnt", c_p), ("part", i32), ("nparts", i32),
                ("sig", c_p), ("dep", c_p), ("err", c_p), ("dep_need", i32), ("reserved_", i32), ("retire_flag", c_p), ("retire_stamp", c_p)]


class GemmFp8Problem(C.Structure):
    _fields_ = [("p", GemmProblem), ("scale_a", c_p), ("scale_b", c_p), ("c8", c_p), ("c8_mul", C.c_float), ("ldc8", i32)]


GEMM_FP8_MAX_GROUP = 4


class DropRows(C.Structure):
    _fields_ = [("site", C.c_uint32), ("div", i32), ("mul", i32), ("off", i32)]


class LnArgs(C.Structure):
    _fields_ = [("d", c_p), ("x", c_p), ("addvec", c_p), ("gamma", c_p), ("beta", c_p), ("y", c_p), ("z", c_p),
                ("mean", c_p), ("rstd", c_p), ("dyn", c_p), ("M", i32), ("H", i32), ("split_row", i32), ("post", i32),
                ("out_scale", C.c_float), ("drop", Dropout), ("seg", DropRows * 2), ("y8", c_p), ("y8_scale", c_p), ("ld8", C.c_int64)]


class LnBwdArgs(C.Structure):
    _fields_ = [("dy", c_p), ("z", c_p), ("mean", c_p), ("rstd", c_p), ("gamma", c_p), ("dz", c_p), ("dd", c_p),
                ("partial", c_p), ("dgamma", c_p), ("dbeta", c_p), ("dyn", c_p), ("M", i32), ("H", i32),
                ("split_row", i32), ("post", i32), ("out_scale", C.c_float), ("accumulate", i32), ("drop", Dropout),
                ("seg", DropRows * 2)]


class ConcapArgs(C.Structure):
    _fields_ = [(n, c_p) for n in ("cap_tokens", "cap_len", "cap_index", "feat", "cls", "boxes", "num_boxes", "img_wh", "input_ids", "input_mask",
                                   "segment_ids", "lm_label_ids", "is_match", "image_feat", "image_loc", "image_cls", "image_label", "image_mask")] + \
               [("seed", C.c_uint64)] + [(n, i32) for n in ("B", "T", "R", "F", "C", "n_caps", "cap_ld", "vocab_size", "cls_id", "sep_id", "mask_id",
                                                            "add_global", "objective", "visualization")]


class ConcapRecord(C.Structure):
    _fields_ = [(n, c_p) for n in ("feat", "cls", "attr", "boxes", "obj_labels", "obj_confs", "attr_labels", "attr_confs")] + \
               [(n, i32) for n in ("R", "F", "C", "A", "num_boxes")] + [("img_w", C.c_float), ("img_h", C.c_float), ("caption_len", i32),
                                                                        ("caption", c_p), ("image_id", C.c_char * 64)]


class EmbedArgs(C.Structure):
    _fields_ = [("ids", c_p), ("type_ids", c_p), ("pos_ids", c_p), ("word", c_p), ("pos", c_p), ("type", c_p),
                ("extra", c_p), ("z", c_p), ("M", i32), ("T", i32), ("H", i32), ("V", i32), ("P", i32), ("n_types", i32)]


class EmbedBwdArgs(C.Structure):
    _fields_ = [("dz", c_p), ("ids", c_p), ("type_ids", c_p), ("pos_ids", c_p), ("dword", c_p), ("dpos", c_p),
                ("dtype", c_p), ("M", i32), ("T", i32), ("H", i32), ("n_types", i32), ("V", i32), ("P", i32)]


class XentArgs(C.Structure):
    _fields_ = [("logits", c_p), ("labels", c_p), ("pos", c_p), ("count", c_p), ("lse", c_p), ("loss_sum", c_p),
                ("V", i32), ("ld", i32), ("max_rows", i32)]


class KlArgs(C.Structure):
    _fields_ = [("logits", c_p), ("target", c_p), ("pos", c_p), ("count", c_p), ("lse", c_p), ("tsum", c_p),
                ("loss_sum", c_p), ("weight", C.c_float), ("V", i32), ("ld", i32), ("max_rows", i32)]


class VisLossArgs(C.Structure):
    _fields_ = [("logits", c_p), ("target", c_p), ("labels", c_p), ("conf", c_p), ("pos", c_p), ("count", c_p), ("neg_index", c_p),
                ("lse", c_p), ("aux", c_p), ("loss_sum", c_p), ("weight", C.c_float), ("V", i32), ("ld", i32), ("max_rows", i32),
                ("kind", i32), ("n_neg", i32)]


VIS_MSE, VIS_NCE, VIS_XENT, VIS_HUBER = 1, 2, 3, 5
NCE_ACROSS, NCE_INSIDE, NCE_MAX_SAMPLES = 89, 38, 128
FUSE_MUL, FUSE_SUM, FUSE_TEXT = 0, 1, 2


class AdamwArgs(C.Structure):
    _fields_ = [("p", c_p), ("g", c_p), ("m", c_p), ("v", c_p), ("shadow", c_p), ("chunk_class", c_p), ("clip", c_p),
                ("n", C.c_int64), ("cls_lr_mult", C.c_float * 8), ("cls_wd", C.c_float * 8), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("step_mult", C.c_float),
                ("grad_scale", C.c_float)]


class TailJob(C.Structure):
    _fields_ = [("dst", c_p), ("dst2", c_p), ("src", c_p), ("src2", c_p), ("stride", C.c_int64), ("n", C.c_int64), ("kind", i32), ("count", i32),
                ("accumulate", i32), ("block_start", i32)]


TAIL_MAX_JOBS = 16


class GenericArgs(C.Structure):
    _fields_ = [("fn", i32), ("p", c_p * 6), ("n", C.c_int64 * 6), ("f", C.c_float * 2), ("drop", Dropout)]


class Op(C.Structure):
    _fields_ = [("kind", i32), ("i0", i32), ("i1", i32), ("i2", i32), ("a", c_p), ("b", c_p), ("c", c_p)]


(OP_GEMM, OP_LN_FWD, OP_LN_BWD, OP_ATTN_FWD, OP_ATTN_BWD, OP_EMBED_FWD, OP_EMBED_BWD, OP_XENT_FWD, OP_XENT_BWD,
 OP_KL_FWD, OP_KL_BWD, OP_GENERIC, OP_SIDE_BEGIN, OP_SIDE_END, OP_WAIT_SIDE, OP_JOIN, OP_LN_FINALIZE, OP_GEMM_FP8, OP_GEMM_CHAIN) = range(1, 20)
(FN_CAST, FN_MEMSET, FN_LOC_FWD, FN_LOC_BWD, FN_ADD_DROPOUT, FN_COLSUM, FN_SELECT, FN_GATHER, FN_SCATTER_ADD,
 FN_LOSS_FINAL, FN_POOL_FWD, FN_POOL_BWD, FN_MASK_PREP, FN_MUL, FN_VLBERT_PREP, FN_VLBERT_MASKGRAD, FN_ROWGROUP_SUM,
 FN_RELU_BWD, FN_COPY, FN_SUM_SLABS, FN_SUM_SLABS_BF16, FN_SIDE_TAIL, FN_QUANT_ROWS, FN_CAST_FP8, FN_VIS_LOSS_FWD, FN_VIS_LOSS_BWD,
 FN_NCE_NEG, FN_TEXT_END_ROWS, FN_VLBERT_OBJ_IDS, FN_VLBERT_POSITIONS, FN_HOLD, FN_GATE, FN_BUMP) = range(1, 34)


class AttnArgs(C.Structure):
    _fields_ = [("q", c_p * 2), ("k", c_p * 2), ("v", c_p * 2), ("ld", i32 * 2), ("L", i32 * 2), ("mask", c_p * 2),
                ("ctx", c_p * 2), ("ldo", i32 * 2), ("lse", c_p * 2), ("B", i32), ("nh", i32), ("gate", (i32 * 2) * 2),
                ("drop", (Dropout * 2) * 2), ("scale", C.c_float), ("dh", i32), ("probs", (c_p * 2) * 2)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("dctx", c_p * 2), ("dq", c_p * 2), ("dk", c_p * 2), ("dv", c_p * 2), ("ldg", i32 * 2)]


NT, NN, TN = 0, 1, 2
EPI_BF16, EPI_GELU, EPI_MULR, EPI_ADDR, EPI_F32, EPI_RELU, EPI_F32_ACC = range(7)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("vk_version", C.c_int)
_sig("vk_device_arch", C.c_char_p)
_sig("vk_last_error", C.c_char_p)
_sig("vk_set_seed", C.c_int, c_p, C.c_uint64, c_p)
_sig("vk_cast_f32_bf16", C.c_int, c_p, c_p, C.c_int64, c_p)
_sig("vk_gemm_grouped", C.c_int, C.c_int, C.c_int, C.POINTER(GemmProblem), C.c_int, c_p)
_sig("vk_gemm_fp8_grouped", C.c_int, C.c_int, C.POINTER(GemmFp8Problem), C.c_int, C.c_int, c_p)
_sig("vk_quant_rows_fp8", C.c_int, c_p, C.c_int, C.c_int64, c_p, C.c_int64, c_p, C.c_int, C.c_int, c_p, c_p)
_sig("vk_cast_bf16_fp8", C.c_int, c_p, c_p, C.c_int64, C.c_float, c_p)
_sig("vk_gemm_grouped_ex", C.c_int, C.c_int, C.c_int, C.POINTER(GemmProblem), C.c_int, C.c_int, c_p)
_sig("vk_gemm_chain", C.c_int, C.c_int, C.c_int, C.POINTER(GemmProblem), C.c_int, C.c_int, C.POINTER(GemmProblem), C.c_int, c_p)
_sig("vk_gemm_split_workspace_bytes", C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int))
_sig("vk_gated_attn_fwd", C.c_int, C.POINTER(AttnArgs), c_p)
_sig("vk_gated_attn_bwd", C.c_int, C.POINTER(AttnArgs), C.POINTER(AttnBwdArgs), c_p)
_sig("vk_gated_attn_lds_bytes", C.c_size_t, C.POINTER(AttnArgs), C.c_int)
_sig("vk_ln_fwd", C.c_int, C.POINTER(LnArgs), c_p)
_sig("vk_ln_bwd_partial_rows", C.c_int, C.c_int)
_sig("vk_ln_bwd", C.c_int, C.POINTER(LnBwdArgs), c_p)
_sig("vk_embed_sum_fwd", C.c_int, C.POINTER(EmbedArgs), c_p)
_sig("vk_embed_sum_bwd", C.c_int, C.POINTER(EmbedBwdArgs), c_p)
_sig("vk_rows32", C.c_int, C.c_int)
_sig("vk_loc_linear_fwd", C.c_int, c_p, c_p, c_p, c_p, C.c_int, C.c_int, C.c_int, c_p)
_sig("vk_loc_linear_bwd", C.c_int, c_p, c_p, c_p, c_p, c_p, C.c_int, C.c_int, C.c_int, c_p)
_sig("vk_add_dropout", C.c_int, c_p, c_p, c_p, C.c_int, C.c_int, C.c_float, Dropout, C.c_int, c_p)
_sig("vk_colsum_bf16", C.c_int, c_p, c_p, c_p, C.c_int, C.c_int, C.c_int, c_p)
_sig("vk_vlbert_prep_fwd", C.c_int, c_p, C.c_int, c_p, c_p, c_p, c_p, C.c_int, C.c_int, C.c_int, Dropout, c_p)
_sig("vk_vlbert_maskgrad", C.c_int, c_p, C.c_int, C.c_int, c_p, c_p, c_p, C.c_int, C.c_int, Dropout, c_p)
_sig("vk_rowgroup_sum_bf16", C.c_int, c_p, c_p, C.c_int, C.c_int, C.c_int, c_p)
_sig("vk_relu_bwd_bf16", C.c_int, c_p, c_p, c_p, C.c_int64, c_p)
_sig("vk_copy_async", C.c_int, c_p, c_p, C.c_int64, c_p)
_sig("vk_select_rows", C.c_int, c_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_p, c_p, c_p, c_p)
_sig("vk_gather_rows", C.c_int, c_p, c_p, c_p, c_p, C.c_int, C.c_int, c_p)
_sig("vk_scatter_rows_add", C.c_int, c_p, c_p, c_p, c_p, C.c_int, C.c_int, c_p)
_sig("vk_xent_fwd", C.c_int, C.POINTER(XentArgs), c_p)
_sig("vk_xent_bwd", C.c_int, C.POINTER(XentArgs), c_p, C.c_int, c_p, c_p)
_sig("vk_kl_fwd", C.c_int, C.POINTER(KlArgs), c_p)
_sig("vk_kl_bwd", C.c_int, C.POINTER(KlArgs), c_p, C.c_int, c_p, c_p)
_sig("vk_loss_finalize", C.c_int, c_p, c_p, c_p, C.c_int, C.c_float, c_p, c_p)
_sig("vk_pool_mul_fwd", C.c_int, c_p, c_p, c_p, C.c_int, C.c_int, Dropout, c_p)
_sig("vk_pool_fuse_fwd", C.c_int, c_p, c_p, c_p, C.c_int, C.c_int, C.c_int, Dropout, c_p)
_sig("vk_pool_fuse_bwd", C.c_int, c_p, C.c_int, c_p, c_p, c_p, c_p, C.c_int, C.c_int, C.c_int, Dropout, c_p)
_sig("vk_text_end_rows", C.c_int, c_p, C.c_int, C.c_int, c_p, c_p, c_p)
_sig("vk_vlbert_obj_ids", C.c_int, c_p, c_p, C.c_int, C.c_int, c_p)
_sig("vk_lmdb_open", C.c_int, C.c_char_p, C.POINTER(c_p))
_sig("vk_lmdb_close", None, c_p)
_sig("vk_lmdb_entries", C.c_int64, c_p)
_sig("vk_lmdb_first", C.c_int, c_p)
_sig("vk_lmdb_next", C.c_int, c_p, C.POINTER(c_p), C.POINTER(C.c_size_t), C.POINTER(c_p), C.POINTER(C.c_size_t))
_sig("vk_lmdb_get", C.c_int, c_p, C.c_char_p, C.c_size_t, C.POINTER(c_p), C.POINTER(C.c_size_t))
_sig("vk_concap_record_decode", C.c_int, c_p, C.c_size_t, C.POINTER(ConcapRecord))
_sig("vk_concap_records_decode", C.c_int, C.POINTER(c_p), C.POINTER(C.c_size_t), C.POINTER(ConcapRecord), C.c_int, C.c_int, C.POINTER(C.c_int))
_sig("vk_wordpiece_open", C.c_int, C.c_char_p, C.c_int, C.POINTER(c_p))
_sig("vk_wordpiece_close", None, c_p)
_sig("vk_wordpiece_vocab_size", C.c_int, c_p)
_sig("vk_wordpiece_token_id", C.c_int, c_p, C.c_char_p)
_sig("vk_wordpiece_encode", C.c_int, c_p, C.c_char_p, C.c_size_t, c_p, C.c_int)
_sig("vk_wordpiece_encode_batch", C.c_int, c_p, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_int, c_p, C.c_int, c_p, C.c_int)
_sig("vk_b64_decode", C.c_int, C.c_char_p, C.c_size_t, c_p, C.c_size_t, C.POINTER(C.c_size_t))
_sig("vk_vlbert_positions", C.c_int, c_p, C.c_int, C.c_int, C.c_int, c_p, c_p, c_p)
_sig("vk_vis_loss_fwd", C.c_int, C.POINTER(VisLossArgs), c_p)
_sig("vk_vis_loss_bwd", C.c_int, C.POINTER(VisLossArgs), c_p, C.c_int, c_p, c_p)
_sig("vk_nce_negatives", C.c_int, Dropout, C.c_int, C.c_int, c_p, c_p)
_sig("vk_pool_mul_bwd", C.c_int, c_p, C.c_int, c_p, c_p, c_p, c_p, C.c_int, C.c_int, Dropout, c_p)
_sig("vk_mask_prep", C.c_int, c_p, c_p, C.c_int, c_p)
_sig("vk_mul_bf16", C.c_int, c_p, c_p, c_p, C.c_int64, c_p, C.c_int, c_p)
_sig("vk_grad_norm_workspace_floats", C.c_int)
_sig("vk_grad_norm_clip", C.c_int, c_p, C.c_int64, C.c_float, C.c_float, c_p, c_p, c_p)
_sig("vk_adamw_step", C.c_int, C.POINTER(AdamwArgs), c_p)
_sig("vk_adamw_step_on", C.c_int, C.POINTER(AdamwArgs), C.c_int, c_p)
_sig("vk_grad_norm_clip_masked", C.c_int, c_p, C.c_int64, c_p, C.c_float, C.c_float, c_p, c_p, c_p)
_sig("vk_grad_sqnorm_chunks", C.c_int, c_p, C.c_int64, C.c_int64, c_p, c_p, c_p)
_sig("vk_grad_norm_from_chunks", C.c_int, c_p, C.c_int64, C.c_float, C.c_float, c_p, c_p)
_sig("vk_axpy_f32", C.c_int, c_p, c_p, C.c_float, C.c_int64, c_p)
_sig("vk_sum_slabs_f32", C.c_int, c_p, c_p, C.c_int64, C.c_int, C.c_int64, c_p)
_sig("vk_sum_slabs_bf16", C.c_int, c_p, c_p, C.c_int64, C.c_int, C.c_int64, c_p, C.c_int, c_p)
_sig("vk_memset_async", C.c_int, c_p, C.c_int, C.c_int64, c_p)
_sig("vk_hold_cus", C.c_int, C.c_int, C.c_int, C.c_int, c_p)
_sig("vk_gate_wait", C.c_int, c_p, c_p, C.c_int, c_p, c_p)
_sig("vk_bump_u64", C.c_int, c_p, c_p)
_sig("vk_store_u64", C.c_int, c_p, C.c_uint64, c_p)
_sig("vk_gate_value", C.c_int, c_p, C.c_uint64, C.c_int, c_p, c_p)
_sig("vk_comm_standin", C.c_int, c_p, c_p, C.c_int64, C.c_int, C.c_int, c_p, c_p)
_sig("vk_gemm_reserve_cus", C.c_int, C.c_int)
_sig("vk_side_tail", C.c_int, C.POINTER(TailJob), C.c_int, c_p)
_sig("vk_run_ops", C.c_int, C.POINTER(Op), C.c_int, c_p)
_sig("vk_run_ops_timed", C.c_int, C.POINTER(Op), C.c_int, c_p, C.POINTER(C.c_float))
_sig("vk_ln_bwd_finalize", C.c_int, C.POINTER(LnBwdArgs), c_p)
_sig("vk_ln_fwd_pair", C.c_int, C.POINTER(LnArgs), C.POINTER(LnArgs), c_p)
_sig("vk_ln_bwd_pair", C.c_int, C.POINTER(LnBwdArgs), C.POINTER(LnBwdArgs), c_p)
_sig("vk_concap_batch", C.c_int, C.POINTER(ConcapArgs), c_p)
_sig("vk_side_join", C.c_int, c_p)
_sig("vk_side_join_from", C.c_int, c_p, c_p)
_sig("vk_side_enable", None, C.c_int)
_sig("vk_side_stream", c_p, c_p)

EXPORTS = ["vk_version", "vk_device_arch", "vk_last_error", "vk_set_seed", "vk_cast_f32_bf16", "vk_gemm_grouped", "vk_gemm_grouped_ex", "vk_gemm_chain", "vk_gemm_split_workspace_bytes", "vk_gemm_fp8_grouped", "vk_quant_rows_fp8", "vk_cast_bf16_fp8",
           "vk_ln_fwd", "vk_ln_fwd_pair", "vk_ln_bwd_partial_rows", "vk_ln_bwd", "vk_ln_bwd_pair", "vk_ln_bwd_finalize", "vk_gated_attn_fwd", "vk_gated_attn_bwd", "vk_gated_attn_lds_bytes",
           "vk_embed_sum_fwd", "vk_embed_sum_bwd", "vk_rows32", "vk_loc_linear_fwd", "vk_loc_linear_bwd",
           "vk_add_dropout", "vk_colsum_bf16", "vk_vlbert_prep_fwd", "vk_vlbert_maskgrad", "vk_rowgroup_sum_bf16",
           "vk_relu_bwd_bf16", "vk_copy_async", "vk_select_rows", "vk_gather_rows", "vk_scatter_rows_add", "vk_xent_fwd",
           "vk_xent_bwd", "vk_kl_fwd", "vk_kl_bwd", "vk_loss_finalize", "vk_pool_mul_fwd", "vk_pool_mul_bwd",
           "vk_pool_fuse_fwd", "vk_pool_fuse_bwd", "vk_text_end_rows", "vk_vlbert_obj_ids", "vk_vlbert_positions", "vk_vis_loss_fwd", "vk_vis_loss_bwd", "vk_nce_negatives",
           "vk_mask_prep", "vk_mul_bf16", "vk_grad_norm_workspace_floats", "vk_grad_norm_clip", "vk_grad_norm_clip_masked", "vk_grad_sqnorm_chunks", "vk_grad_norm_from_chunks", "vk_adamw_step", "vk_adamw_step_on",
           "vk_axpy_f32", "vk_sum_slabs_f32", "vk_sum_slabs_bf16", "vk_memset_async", "vk_hold_cus", "vk_gate_wait", "vk_bump_u64", "vk_store_u64", "vk_gate_value", "vk_comm_standin", "vk_gemm_reserve_cus", "vk_side_tail", "vk_run_ops", "vk_run_ops_timed", "vk_side_join", "vk_side_join_from", "vk_side_stream", "vk_side_enable", "vk_concap_batch",
           "vk_lmdb_open", "vk_lmdb_close", "vk_lmdb_entries", "vk_lmdb_first", "vk_lmdb_next", "vk_lmdb_get", "vk_concap_record_decode", "vk_concap_records_decode", "vk_b64_decode",
           "vk_wordpiece_open", "vk_wordpiece_close", "vk_wordpiece_vocab_size", "vk_wordpiece_token_id", "vk_wordpiece_encode", "vk_wordpiece_encode_batch"]


def check(rc):
    if rc != 0:
        raise VoltaHipError(lib.vk_last_error().decode())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None
