"""ctypes binding of ``libbiomedkg_hip.so`` -- the C-ABI boundary declared in
``include/biomedkg_hip.h``.  There is no fallback: if the library is missing, or a call returns
non-zero, this raises.  torch only supplies device memory (``tensor.data_ptr()``) and the stream.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int32, c_int64, c_uint32, c_uint64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BKG_HIP_LIB: another build of the same library -- tuning variants compiled with other -D flags, scripts/dev_*)
LIB_PATH = os.environ.get("BKG_HIP_LIB") or os.path.join(_HERE, "csrc", "libbiomedkg_hip.so")

P = c_void_p
I32, I64, U32, U64, F32, F64 = c_int32, c_int64, c_uint32, c_uint64, c_float, c_double

class RowSplitStruct(ctypes.Structure):
    """``bkg_rowsplit`` of include/biomedkg_hip.h (device pointers + upper-bound sizes)."""
    _fields_ = [(n, c_void_p) for n in ("item_row", "item_beg", "item_end", "item_slot", "long_row", "long_slot0",
                                        "long_cnt", "short_desc", "short_col", "totals")] + [(n, c_int64) for n in
                                                                                ("n_items_max", "n_slots_max", "n_long_max",
                                                                                 "n_short_max")]


# name -> (restype, argtypes); mirrors include/biomedkg_hip.h one to one.
SIGNATURES = {
    "bkg_abi_version": (c_int32, []),
    "bkg_last_error": (ctypes.c_char_p, []),
    "bkg_set_deterministic_workspace": (I32, [P, I64]),
    "bkg_set_rng_epoch": (I32, [P]),
    # graph.hip
    "bkg_sort_workspace_bytes": (I64, [I64]),
    "bkg_sort_edges": (I32, [P, P, I64, I64, I64, P, P, P, P, P, I64, P]),
    "bkg_keys_rowptr_presorted": (I32, [P, P, I64, I64, I64, P, P, I64, U32, P, P]),
    "bkg_sort_edges_rowptr": (I32, [P, P, I64, I64, I64, P, P, P, P, I64, U32, P, P, I64, P]),
    "bkg_lower_bound_u32": (I32, [P, I64, I64, U32, P, P]),
    "bkg_gather_i64_to_i32": (I32, [P, P, P, I64, P]),
    "bkg_gather_i32": (I32, [P, P, P, I64, P]),
    "bkg_gather_f32": (I32, [P, P, P, I64, P]),
    "bkg_csr_compact_workspace_bytes": (I64, [I64]),
    "bkg_csr_compact": (I32, [P, P, P, I64, I64, P, P, P, P, P, P, P, I64, P]),
    "bkg_gcn_view_workspace_bytes": (I64, [I64]),
    "bkg_gcn_view": (I32, [P, P, P, P, P, P, P, P, P, I64, I64, P, P, P, P, P, P, P, P, P, P, P, P, I64, P]),
    "bkg_gcn_norm": (I32, [P, P, P, P, P, P, I64, I64, P, P, P, P]),
    "bkg_rgcn_segments": (I32, [P, I64, I64, P, P, P, P, P, P, P, P, P, I64, P]),
    "bkg_rgcn_segments_static": (I32, [P, I64, I64, I32, P, P, P, P, P, P, P, P, P, I64, P]),
    "bkg_rgcn_edge_order": (I32, [P, P, P, I64, P, P, P]),
    # spmm.hip
    "bkg_rowsplit_build": (I32, [P, I64, I64, I32, I32, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I64, P]),
    "bkg_cast_f32_bf16": (I32, [P, P, F32, P, I64, P]),
    "bkg_mask_cast_bf16": (I32, [P, I32, F32, U64, U32, P, I64, P]),
    "bkg_gate_cast_bf16": (I32, [P, I32, P, I32, F32, P, I64, P]),
    "bkg_spmm_csr_f32": (I32, [P, P, P, P, I32, I64, P, I32, I64, I64, I32, P, P, I32, I64, P, I32, P, F32, U64, U32, P, I64, F32,
                                 P, P, I32, P]),
    # gemm_f32.hip
    "bkg_gemm_f32": (I32, [I32, I32, I64, I64, I64, F32, P, I64, P, I64, F32, P, I64, P, I32, P, I64, F32, I32, P]),
    "bkg_gemm_grouped_rows_f32": (I32, [I32, I64, I64, I64, P, I64, P, I64, I64, P, I64, P, I32, P, I64, F32, I32, P]),
    "bkg_gemm_grouped_k_chunk": (I32, [I64, I64, I64, I32]),
    "bkg_gemm_grouped_k_f32": (I32, [I64, I64, I64, P, I64, P, I64, P, I64, I64, P, I32, I32, P]),
    "bkg_gemm_tn_bf16": (I32, [I64, I64, I64, P, I32, I64, P, I32, I64, P, I64, P]),
    "bkg_gemm_grouped_k_bf16": (I32, [I64, I64, I64, P, I32, I64, P, I32, I64, P, P, I64, I64, P, I32, P]),
    # gemm_xw.hip
    "bkg_weight_to_bf16": (I32, [P, I64, I64, I64, I64, I32, P, I64, I64, P, I64, I64, P]),
    "bkg_gemm_xw": (I32, [I64, I64, I64, F32, P, I32, I64, P, P, I64, I64, P, I32, F32, P, I32, I64, P, I32, P, I64, F32, P]),
    "bkg_gemm_xw8_gated_workspace_floats": (I64, [I64]),
    "bkg_gemm_xw8_gated": (I32, [I64, P, I64, P, I64, P, I64, P, I64, F32, P, P, P]),
    # kge.hip
    "bkg_kge_score_f32": (I32, [I32, P, I64, I32, P, P, P, I64, P, I64, P, P]),
    "bkg_row_l1norm_f32": (I32, [P, I64, I64, I32, P, P]),
    "bkg_kge_incidence_streams": (I32, [P, I64, I64, P, P, P, P, P, P, P, P, P]),
    "bkg_kge_bwd_stream_f32": (I32, [P, I32, I64, I64, I32, P, I64, I64, P, P, P, P, I64, P, P, I32, P, P, I64, F32, P, I64, P]),
    "bkg_kge_onepass_batch": (I32, [I32, I32]),
    "bkg_kge_onepass_blocks": (I32, [I64]),
    "bkg_kge_onepass_smem_bytes": (I64, [I32, I32, I64]),
    "bkg_kge_role_incidence": (I32, [P, P, P, I64, I64, P, P, P, P, P, P, I64, P]),
    "bkg_kge_onepass_mx_ok": (I32, [I32, I32, I64]),
    "bkg_kge_onepass_mx_blocks": (I32, [I64]),
    "bkg_kge_onepass_mx_f32": (I32, [P, I64, I64, P, I64, I64, P, P, P, I64, F32, F32, P, P, P, I64, P, I32, P]),
    "bkg_kge_onepass_f32": (I32, [P, I32, I64, I64, I32, P, I64, I64, P, P, P, P, I64, F32, F32, P, P, P, P, I64, P, P, P]),
    "bkg_kge_bwd_nodes_f32": (I32, [I32, P, I32, I64, I64, I32, P, P, P, I64, P, I64, I64, P, P, P, P, P, P, I64, P, P, I32, I64, P]),
    "bkg_kge_fused_chunk": (I32, []),
    "bkg_kge_fused_bce_waves": (I64, [I64, I32]),
    "bkg_kge_fused_bce_f32": (I32, [I32, P, I32, I64, I32, P, P, I64, F32, F32, P, I64, I32, P, P, P, P, P, P, P, I64, P]),
    "bkg_kge_fused_bce_dev_f32": (I32, [I32, P, I32, I64, I32, P, P, I64, F32, F32, P, P, I64, I32, P, P, P, P, P, P, P, I64, P]),
    "bkg_kge_loss_finish_scratch_bytes": (I64, []),
    "bkg_kge_loss_finish_f32": (I32, [P, I32, F32, P, P, I64, I32, I32, P, F64, P, I64, I32, F32, P, P, P]),
    "bkg_scale_add_rows_f32": (I32, [P, I64, P, I64, I32, I32, P, F32, P, P]),
    "bkg_scale_add_rows_out_f32": (I32, [P, I64, P, I64, I32, I32, P, F32, P, I64, P]),
    "bkg_kge_dz_finish_f32": (I32, [P, I64, P, I64, P, I64, I64, I32, P, F32, P, I64, P]),
    "bkg_relation_wave_ptr": (I32, [P, I32, I32, P, P]),
    "bkg_kge_bwd_rel_f32": (I32, [I32, P, I64, I32, P, P, P, I64, P, I64, P, I32, P, I64, P, P, P]),
    # ntxent.hip
    "bkg_ntxent_hpad": (I32, [I32, I32]),
    "bkg_ntxent_bf16_scale": (F32, [F32]),
    "bkg_ntxent_prepare": (I32, [P, I64, P, I64, I64, I32, I32, F32, P, P, P, P, P]),
    "bkg_ntxent_forward_workspace_floats": (I64, [I64, I32, I32]),
    "bkg_ntxent_forward": (I32, [P, I32, I64, I32, F32, P, P, P, P, P, P, P]),
    "bkg_ntxent_backward_workspace_floats": (I64, [I64, I32, I32]),
    "bkg_ntxent_backward": (I32, [P, I32, P, P, P, P, I64, I32, F32, P, P, I64, P, I64, P]),
    "bkg_ntxent_rows_forward_workspace_floats": (I64, [I64, I64, I32, I32]),
    "bkg_ntxent_rows_forward": (I32, [P, I32, I64, I32, F32, P, I64, P, P, P]),
    "bkg_ntxent_rows_backward_workspace_floats": (I64, [I64, I64, I32, I32]),
    "bkg_ntxent_rows_backward": (I32, [P, I32, P, P, P, P, I64, I32, F32, P, I64, I64, F32, P, P, I64, P, I64, P]),
    # sample.hip
    "bkg_random_walk": (I32, [P, P, P, I64, I32, U64, P, P, P]),
    "bkg_sample_neighbors": (I32, [P, P, I64, I32, U64, U32, P, P, P]),
    "bkg_neighbor_workspace_bytes": (I64, [I64, I32]),
    "bkg_neighbor_begin": (I32, [P, I64, I64, P, P, P, P]),
    "bkg_neighbor_hop": (I32, [P, P, P, I64, I32, U64, U32, P, P, P, P, P, P, P, P, P, I64, P, P, P, P, I64, P]),
    "bkg_neighbor_finish": (I32, [I64, P, P, P, I64, P, P, P]),
    "bkg_saint_batch_workspace_bytes": (I64, [I64, I64]),
    "bkg_saint_batch": (I32, [P, P, P, I64, I32, U64, P, P, P, P, P, P, I64, I64, P, P, P, P, P, I64, I64, I64, I64, P, P, P, P, P,
                              P, P, I64, P]),
    # rank.hip
    "bkg_rank_true_score_f32": (I32, [P, I64, P, I64, P, I64, I32, P, P]),
    "bkg_rank_count_f32": (I32, [P, I64, I64, P, I64, I64, I32, P, P, I32, P, P, P]),
    "bkg_rank_filter_f32": (I32, [P, I64, P, I64, I32, P, P, P, P, I64, P, P, P]),
    "bkg_rank_l1_true_f32": (I32, [P, I64, P, I64, P, I64, I32, P, P]),
    "bkg_rank_l1_count_f32": (I32, [P, I64, I64, P, I64, I64, I32, P, P, I32, P, P, P]),
    "bkg_rank_l1_filter_f32": (I32, [P, I64, P, I64, I32, P, P, P, P, I64, P, P, P]),
    # pair.hip
    "bkg_pair_rows_bf16": (I32, [P, P, P, I64, I32, P]),
    "bkg_unpair_rows_bf16": (I32, [P, P, P, I64, I32, P]),
    # colsum.hip
    "bkg_colsum_workspace_floats": (I64, [I64, I32]),
    "bkg_gate_f32": (I32, [P, P, F32, I64, P, P]),
    "bkg_colsum_f32": (I32, [P, I32, I64, I64, I32, P, P, P]),
    "bkg_colsum_weighted_f32": (I32, [P, I32, I64, I64, I32, P, P, P, P]),
    "bkg_gate_cast_colsum_bf16": (I32, [P, I32, P, I32, F32, P, I64, I32, P, P, P]),
    "bkg_outer_gate_colsum_bf16": (I32, [P, P, P, F32, P, I64, I32, P, P, P]),
    "bkg_rowdot_f32": (I32, [P, I32, I64, I64, I32, P, P, P]),
    "bkg_scaled_rows": (I32, [P, P, I64, I32, P, I32, I64, P]),
    # negsample.hip
    "bkg_hashset_insert_offdiag": (I32, [P, P, I64, I64, P, I64, P]),
    "bkg_negative_sample_workspace_bytes": (I64, [I64]),
    "bkg_negative_sample": (I32, [I64, I64, I64, U64, P, I64, P, P, P, P, P, I64, I64, P, P, I64, P]),
    "bkg_negative_blocks_struct_bytes": (I64, []),
    "bkg_negative_sample_blocks": (I32, [P, I32, I64, I64, U64, I64, P, I64, P, I64, P, P, P, P, P, P, P, P, I64, P]),
    "bkg_permute_gather_i64": (I32, [P, I64, U64, P, P]),
    "bkg_permute_gather_repeat_i64": (I32, [P, I64, I64, U64, P, P]),
    "bkg_narrow_i64_u8": (I32, [P, I64, P, P]),
    "bkg_permute_gather_repeat_u8": (I32, [P, I64, I64, U64, P, P]),
    "bkg_randperm_dev": (I32, [P, I64, U64, P, P, P]),
    "bkg_negative_sample_dev_candidates": (I64, [I64]),
    "bkg_negative_sample_dev": (I32, [P, I64, P, P, P, I64, U64, P, P, I64, P, P, P, I64, I64, P, P, P, P, P, I64, P]),
    "bkg_padded_loss_inputs": (I32, [P, P, P, P, P, I64, I64, I64, P, P, P, P, P]),
    "bkg_padded_gcl_batch": (I32, [P, P, P, P, I64, I64, I64, I64, P, P, P, P, P]),
    # optim.hip
    "bkg_clip_adam_max_tensors": (I64, []),
    "bkg_clip_adam_chunks": (I64, [P, I32]),
    "bkg_clip_adam_f32": (I32, [I32, P, P, P, P, P, F32, F32, P, F32, F32, F32, F32, P, P, I64, P, P]),
    "bkg_clip_adam_images_f32": (I32, [I32, P, P, P, P, P, P, P, P, P, F32, F32, P, F32, F32, F32, F32, P, P, I64, P, P]),
    # fusion.hip
    "bkg_attn_fuse_f32": (I32, [P, I64, I64, I32, I32, P, I64, P]),
    "bkg_attn_scores_combine_bf16": (I32, [P, I64, P, I64, I64, I32, P, P, F32, P, I64, P]),
    "bkg_redaf_combine_f32": (I32, [P, I64, I64, I32, I32, P, P, F32, U64, U32, I64, P, I64, P]),
    "bkg_ggd_bce_partials": (I64, []),
    "bkg_ggd_bce_f32": (I32, [P, I64, P, I64, I64, P, P, P, P]),
    "bkg_ggd_bce_bwd_f32": (I32, [P, I64, P, I64, I64, P, P, P, I64, P, I64, P]),
    "bkg_redaf_fused_xw": (I32, [I64, I64, I64, P, I32, I64, P, I64, P, P, P, F32, U64, U32, P, I64, P]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(biomedkg_amd has no CPU / eager fallback)"
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the .so is stale
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def sref(struct):
    """Host pointer to a ctypes struct (None -> NULL)."""
    return None if struct is None else ctypes.cast(ctypes.byref(struct), c_void_p)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """Raw handle of torch's CURRENT stream on the current device (every launch of the library goes there)."""
    if _raw_stream is not None:          # ~0.3 us instead of ~2 us for the Stream object: it is paid on every launch
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise HipLibraryError(f"{name} failed ({rc}): {lib().bkg_last_error().decode()}")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipLibraryError(
                "biomedkg_amd runs on MI355X only: got a CPU tensor (there is no CPU fallback; "
                "the CPU restatement lives in oracle/ and is test infrastructure)"
            )
