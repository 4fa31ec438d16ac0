"""ctypes binding of librecstudio_amd.so (the C ABI in include/recstudio_amd.h).

The library is built in-tree (``recstudio_amd/librecstudio_amd.so``) by
``recstudio_amd/csrc/Makefile`` (hipcc --offload-arch=gfx950).  There is no
fallback: if the library is missing or a symbol is absent, loading raises.
"""
import ctypes
import os
import subprocess
from ctypes import (POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_uint32, c_uint64, c_void_p)

HERE = os.path.dirname(os.path.abspath(__file__))
# RSA_LIB: load another build of the same sources (tools/build_variant.sh writes librecstudio_amd_<name>.so next to the
# default one for A/B measurements of compile-time switches); the ABI check below still applies
LIB_PATH = os.environ.get('RSA_LIB') or os.path.join(HERE, 'librecstudio_amd.so')
CSRC = os.path.join(HERE, 'csrc')

RSA_OK = 0
SCORE_IP, SCORE_COS, SCORE_EUC = 0, 1, 2
SAMPLER_GIVEN, SAMPLER_UNIFORM, SAMPLER_POPULAR = 0, 1, 2
LOSS_BPR, LOSS_SSM, LOSS_BCE = 0, 1, 2
LOSS_WBPR, LOSS_WBCE, LOSS_HINGE, LOSS_NCE, LOSS_CCL = 3, 4, 5, 6, 7


class NativeError(RuntimeError):
    """A C-ABI call returned a negative status."""


class FusedArgs(Structure):
    """struct rsa_fused_args (include/recstudio_amd.h)."""
    _fields_ = [
        ('item_table', c_void_p), ('n_items', c_int64), ('dim', c_int32), ('score_mode', c_int32),
        ('query', c_void_p), ('query_index', c_void_p), ('n_query_rows', c_int64),
        ('pos_ids', c_void_p), ('n_queries', c_int64),
        ('num_neg', c_int32), ('sampler', c_int32), ('mask_pad_pos', c_int32), ('guide_log2', c_int32),
        ('seed', c_uint64), ('offset', c_uint64), ('grid_threads', c_uint32), ('_pad', c_uint32),
        ('table', c_void_p), ('pop_prob', c_void_p), ('guide', c_void_p),
        ('neg_ids', c_void_p), ('neg_logp', c_void_p), ('pos_logp', c_void_p),
        ('pos_score', c_void_p), ('neg_score', c_void_p), ('table_prob', c_void_p),
        ('fused_loss', c_int32), ('_pad2', c_int32), ('row_loss', c_void_p), ('loss_out', c_void_p),
        ('dpos', c_void_p), ('dneg', c_void_p), ('cdf_lut', c_void_p), ('query_grad', c_void_p), ('packed_keys', c_void_p), ('offset_dev', c_void_p),
        ('elem_base', c_uint64), ('reduce_scratch', c_void_p), ('cdf_lines', c_void_p), ('lines_log2', c_int32), ('_pad3', c_int32),
        ('solo_flags', c_void_p), ('upd_scale', c_void_p), ('n_batches', c_int32), ('_pad4', c_int32),
        ('batch_offset_step', c_uint64),
    ]


class BackwardArgs(Structure):
    """struct rsa_backward_args (include/recstudio_amd.h)."""
    _fields_ = [
        ('item_table', c_void_p), ('n_items', c_int64), ('dim', c_int32), ('num_neg', c_int32),
        ('query', c_void_p), ('query_index', c_void_p), ('n_query_rows', c_int64),
        ('pos_ids', c_void_p), ('neg_ids', c_void_p), ('n_queries', c_int64),
        ('dpos', c_void_p), ('dneg', c_void_p), ('upstream', c_void_p),
        ('item_grad', c_void_p), ('item_grad_rows', c_void_p), ('query_grad', c_void_p),
        ('query_table_grad', c_void_p), ('query_table_pad_row', c_int32), ('item_pad_row', c_int32),
        ('score_mode', c_int32), ('_pad', c_int32),
    ]


class ShardRouteArgs(Structure):
    """struct rsa_shard_route_args (include/recstudio_amd.h)."""
    _fields_ = [
        ('pos_ids', c_void_p), ('neg_ids', c_void_p), ('neg_logp', c_void_p), ('pos_logp', c_void_p),
        ('n_queries', c_int64), ('num_neg', c_int32), ('sampler', c_int32), ('n_slices', c_int32), ('n_shards', c_int32),
        ('n_banks', c_int32), ('_pad0', c_int32), ('rows_per_shard', c_int64), ('query_base', c_int64), ('capacity', c_int64), ('n_items', c_int64),
        ('seed', c_uint64), ('offset', c_uint64), ('grid_threads', c_uint32), ('_pad', c_uint32), ('elem_base', c_uint64),
        ('table', c_void_p), ('pop_prob', c_void_p), ('guide', c_void_p), ('table_prob', c_void_p), ('cdf_lut', c_void_p),
        ('cdf_lines', c_void_p), ('guide_log2', c_int32), ('lines_log2', c_int32),
        ('send_keys', c_void_p), ('slot_of', c_void_p), ('cursors', c_void_p), ('counts_out', c_void_p),
        ('skip_pos', c_int32), ('group_by_query', c_int32),
        ('deterministic', c_int32), ('_pad1', c_int32), ('wg_scratch', c_void_p), ('wg_scratch_ints', c_int64),
        ('extra_dropped', c_void_p),
    ]


class ShardHomeArgs(Structure):
    """struct rsa_shard_home_args (include/recstudio_amd.h)."""
    _fields_ = [
        ('scores', c_void_p), ('slot_of', c_void_p), ('n_queries', c_int64), ('num_neg', c_int32), ('loss', c_int32),
        ('pos_logp', c_void_p), ('neg_logp', c_void_p), ('mean_den', c_int64),
        ('pos_score', c_void_p), ('neg_score', c_void_p), ('row_loss', c_void_p), ('loss_out', c_void_p),
        ('dpos', c_void_p), ('dneg', c_void_p), ('d_send', c_void_p), ('reduce_scratch', c_void_p),
    ]


class ShardBackwardArgs(Structure):
    """struct rsa_shard_backward_args (include/recstudio_amd.h)."""
    _fields_ = [
        ('item_local', c_void_p), ('n_rows', c_int64), ('dim', c_int32), ('q_all', c_void_p), ('n_query_rows', c_int64),
        ('keys', c_void_p), ('n_segments', c_int64), ('stride', c_int64), ('d_owner', c_void_p), ('item_target', c_void_p),
        ('item_scale', c_void_p), ('step_dropped', c_void_p), ('scale_out', c_void_p), ('qgrad_all', c_void_p),
        ('item_pad_row', c_int64), ('workspace', c_void_p), ('workspace_bytes', c_int64),
    ]


class ShardOwnerBprArgs(Structure):
    """struct rsa_shard_owner_bpr_args (include/recstudio_amd.h)."""
    _fields_ = [
        ('item_local', c_void_p), ('n_rows', c_int64), ('dim', c_int32), ('num_neg', c_int32), ('q_all', c_void_p),
        ('n_query_rows', c_int64), ('keys', c_void_p), ('n_segments', c_int64), ('stride', c_int64), ('pos_rows', c_void_p),
        ('pos_score', c_void_p), ('mean_den', c_int64), ('item_target', c_void_p), ('item_scale', c_void_p),
        ('step_dropped', c_void_p), ('overflow_sticky', c_void_p), ('scale_out', c_void_p), ('qgrad_all', c_void_p),
        ('d_slots', c_void_p), ('dsum_part', c_void_p), ('loss_part', c_void_p), ('reduce_scratch', c_void_p),
        ('item_pad_row', c_int64), ('workspace', c_void_p), ('workspace_bytes', c_int64), ('keys_grouped', c_int32), ('finish_parts', c_int32), ('forward_parts', c_int32),
        ('logq_rows', c_void_p), ('run_max', c_void_p), ('run_sum', c_void_p), ('run_acc', c_void_p),
    ]


class _Sized(Structure):
    """A versioned argument block (include/recstudio_amd.h, "Versioned argument blocks"): ``size`` is filled in here."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.size = ctypes.sizeof(type(self))


class PopularArgs(_Sized):
    """struct rsa_popular_args."""
    _fields_ = [
        ('size', c_int64), ('table', c_void_p), ('pop_prob', c_void_p), ('guide', c_void_p), ('n_items', c_int64),
        ('guide_log2', c_int32), ('lines_log2', c_int32), ('cdf_lut', c_void_p), ('cdf_lines', c_void_p), ('u_in', c_void_p),
        ('ids', c_void_p), ('logp', c_void_p), ('u_out', c_void_p), ('numel', c_int64), ('seed', c_uint64), ('offset', c_uint64),
        ('grid_threads', c_uint32), ('_pad', c_uint32), ('elem_base', c_uint64),
    ]


class LossArgs(_Sized):
    """struct rsa_loss_args."""
    _fields_ = [
        ('size', c_int64), ('loss_kind', c_int32), ('n_pos', c_int32), ('pos_score', c_void_p), ('neg_score', c_void_p),
        ('pos_logp', c_void_p), ('neg_logp', c_void_p), ('n_rows', c_int64), ('num_neg', c_int32), ('_pad', c_int32),
        ('param0', c_float), ('param1', c_float), ('row_loss', c_void_p), ('loss_out', c_void_p), ('dpos', c_void_p),
        ('dneg', c_void_p), ('scratch', c_void_p),
    ]


class RowsUpdateArgs(_Sized):
    """struct rsa_rows_update_args."""
    _fields_ = [
        ('size', c_int64), ('query', c_void_p), ('query_index', c_void_p), ('n_query_rows', c_int64), ('dim', c_int32),
        ('has_pos', c_int32), ('pos_ids', c_void_p), ('neg_ids', c_void_p), ('n_queries', c_int64), ('num_neg', c_int32),
        ('_pad', c_int32), ('dpos', c_void_p), ('dneg', c_void_p), ('upstream', c_void_p), ('n_items', c_int64),
        ('pad_row', c_int64), ('target', c_void_p), ('exp_avg', c_void_p), ('exp_avg_sq', c_void_p), ('lr', c_float),
        ('beta1', c_float), ('beta2', c_float), ('eps', c_float), ('step', c_int64), ('solo', c_void_p), ('workspace', c_void_p),
        ('workspace_bytes', c_int64),
    ]


class BprSgdArgs(_Sized):
    """struct rsa_bpr_sgd_args."""
    _fields_ = [
        ('size', c_int64), ('item_table', c_void_p), ('n_items', c_int64), ('user_table', c_void_p), ('n_users', c_int64),
        ('dim', c_int32), ('num_neg', c_int32), ('user_ids', c_void_p), ('pos_ids', c_void_p), ('n_queries', c_int64),
        ('sampler', c_int32), ('_pad', c_int32), ('pop', POINTER(PopularArgs)), ('seed', c_uint64), ('offset', c_uint64),
        ('grid_threads', c_uint32), ('_pad2', c_uint32), ('elem_base', c_uint64), ('step_scale', c_void_p), ('neg_ids', c_void_p),
        ('solo', c_void_p), ('item_workspace', c_void_p), ('item_workspace_bytes', c_int64), ('user_workspace', c_void_p),
        ('user_workspace_bytes', c_int64), ('pos_score', c_void_p), ('neg_score', c_void_p), ('row_loss', c_void_p),
        ('dpos', c_void_p), ('dneg', c_void_p), ('query_grad', c_void_p), ('ones', c_void_p), ('loss_out', c_void_p),
        ('reduce_scratch', c_void_p), ('uniform_high', c_int64),
    ]


class SegGatherArgs(_Sized):
    """struct rsa_seg_gather_args."""
    _fields_ = [
        ('size', c_int64), ('item_table', c_void_p), ('n_items', c_int64), ('dim', c_int32), ('max_len', c_int32),
        ('flat_item_ids', c_void_p), ('n_flat', c_int64), ('seg_start', c_void_p), ('seg_end', c_void_p), ('n_seg', c_int64),
        ('out_ids', c_void_p), ('out_rows', c_void_p), ('out_len', c_void_p),
    ]


class FullscoreArgs(_Sized):
    """struct rsa_fullscore_args."""
    _fields_ = [
        ('size', c_int64), ('item_table', c_void_p), ('n_items', c_int64), ('dim', c_int32), ('score_mode', c_int32),
        ('query', c_void_p), ('n_query', c_int64), ('scores', c_void_p), ('lse', c_void_p), ('topk_val', c_void_p),
        ('topk_idx', c_void_p), ('k', c_int32), ('_pad', c_int32), ('item_aux', c_void_p), ('query_aux', c_void_p),
        ('workspace', c_void_p), ('workspace_bytes', c_int64),
    ]


# every `typedef struct rsa_* {...}` of the header and its ctypes mirror (tests/test_native_abi.py checks the list against the
# header and every field offset against the C compiler)
STRUCTS = {
    'rsa_fused_args': FusedArgs, 'rsa_backward_args': BackwardArgs, 'rsa_shard_route_args': ShardRouteArgs,
    'rsa_shard_home_args': ShardHomeArgs, 'rsa_shard_backward_args': ShardBackwardArgs,
    'rsa_shard_owner_bpr_args': ShardOwnerBprArgs, 'rsa_popular_args': PopularArgs, 'rsa_loss_args': LossArgs,
    'rsa_rows_update_args': RowsUpdateArgs, 'rsa_bpr_sgd_args': BprSgdArgs, 'rsa_seg_gather_args': SegGatherArgs,
    'rsa_fullscore_args': FullscoreArgs,
}

# name -> (restype, argtypes); must list every symbol the header declares.
SIGNATURES = {
    'rsa_last_error': (c_char_p, []),
    'rsa_abi_version': (c_int, []),
    'rsa_scratch_bytes': (c_int64, []),
    'rsa_device_info': (c_int, [c_int, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    'rsa_sample_uniform': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_uint64, c_uint64, c_uint32, c_uint64, c_void_p]),
    'rsa_sample_masked_uniform': (c_int, [c_void_p, c_int64, c_int32, c_int64, c_int32, c_void_p, c_uint64, c_uint64,
                                          c_uint32, c_uint64, c_void_p]),
    'rsa_sample_popular': (c_int, [POINTER(PopularArgs), c_void_p]),
    'rsa_popular_lookup': (c_int, [POINTER(PopularArgs), c_void_p]),
    'rsa_item_logp': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    'rsa_embedding_gather': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p]),
    'rsa_fused_sample_gather_score': (c_int, [POINTER(FusedArgs), c_void_p]),
    'rsa_pairwise_loss': (c_int, [POINTER(LossArgs), c_void_p]),
    'rsa_ssm_shared_loss': (c_int, [POINTER(LossArgs), c_void_p]),
    'rsa_mean_rows': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'rsa_row_lse': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_float, c_void_p]),
    'rsa_fused_backward': (c_int, [POINTER(BackwardArgs), c_void_p]),
    'rsa_sort_step_elements': (c_int, [POINTER(RowsUpdateArgs), c_void_p]),
    'rsa_rows_update_presorted': (c_int, [POINTER(RowsUpdateArgs), c_void_p]),
    'rsa_scatter_add_rows': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p]),
    'rsa_seg_gather': (c_int, [POINTER(SegGatherArgs), c_void_p]),
    'rsa_fullscore_softmax': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    'rsa_fullscore_softmax_dq_workspace_bytes': (c_int64, [c_int64, c_int64, c_int32]),
    'rsa_fullscore_softmax_dq': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int64, c_void_p]),
    'rsa_scatter_rows_sorted_workspace_bytes': (c_int64, [c_int64, c_int32, c_int64]),
    'rsa_rows_update_sorted': (c_int, [POINTER(RowsUpdateArgs), c_void_p]),
    'rsa_bpr_sgd_prepare': (c_int, [POINTER(BprSgdArgs), c_void_p]),
    'rsa_bpr_sgd_apply': (c_int, [POINTER(BprSgdArgs), c_void_p]),
    'rsa_probs_t_query': (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    'rsa_fullscore_lse_grad_workspace_bytes': (c_int64, [c_int64, c_int64, c_int32]),
    'rsa_fullscore_lse_grad': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'rsa_fullscore_softmax_dw': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'rsa_rng_advance': (c_int, [c_void_p, c_uint64, c_void_p]),
    'rsa_placement_probe': (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_uint32, c_void_p]),
    'rsa_row_topk': (c_int, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    'rsa_topk_mask_history': (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int32, c_void_p,
                                      c_void_p, c_void_p]),
    'rsa_shard_count': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int32, c_void_p, c_void_p]),
    'rsa_shard_route': (c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int64, c_int32, c_int64, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    'rsa_shard_segment_stride': (c_int64, [c_int64]),
    'rsa_shard_sample_route': (c_int, [POINTER(ShardRouteArgs), c_void_p]),
    'rsa_shard_route_workgroups': (c_int64, [POINTER(ShardRouteArgs)]),
    'rsa_shard_route_query_groups': (c_int32, [c_int32, c_uint32, c_uint64, c_int32, c_int32]),
    'rsa_shard_score_segments': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p,
                                         c_void_p, c_void_p, c_void_p]),
    'rsa_shard_home': (c_int, [POINTER(ShardHomeArgs), c_void_p]),
    'rsa_shard_scatter_slots': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    'rsa_shard_unpack_segments': (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          c_void_p]),
    'rsa_shard_backward_workspace_bytes': (c_int64, [c_int64, c_int64, c_int64]),
    'rsa_shard_backward_segments': (c_int, [POINTER(ShardBackwardArgs), c_void_p]),
    'rsa_shard_pos_score': (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                    c_int32, c_void_p]),
    'rsa_shard_owner_bpr_forward': (c_int, [POINTER(ShardOwnerBprArgs), c_void_p]),
    'rsa_shard_owner_bpr_finish': (c_int, [POINTER(ShardOwnerBprArgs), c_void_p, c_void_p]),
    'rsa_shard_owner_ssm_forward': (c_int, [POINTER(ShardOwnerBprArgs), c_void_p]),
    'rsa_shard_owner_ssm_finish': (c_int, [POINTER(ShardOwnerBprArgs), c_void_p, c_void_p]),
    'rsa_shard_unpack': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    'rsa_scatter_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'rsa_gather_f32': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    'rsa_fullscore_workspace_bytes': (c_int64, [c_int64, c_int64, c_int32]),
    'rsa_fullscore': (c_int, [POINTER(FullscoreArgs), c_void_p]),
    'rsa_row_sqnorm': (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
}

_lib = None


def build(verbose=False):
    """Compile the HIP sources for gfx950 (cross-compiles without a GPU)."""
    out = subprocess.run(['make', '-C', CSRC, '-j8'], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout[-4000:])
        print(out.stderr[-4000:])
    if out.returncode != 0:
        raise RuntimeError('recstudio_amd: building librecstudio_amd.so failed')
    global _lib
    _lib = None
    return LIB_PATH


ABI_VERSION = 11     # RSA_ABI_VERSION of include/recstudio_amd.h this binding was written against


def lib():
    """The loaded library with typed entry points.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'recstudio_amd: {LIB_PATH} is missing -- the HIP extension is required (no CPU fallback). '
                f'Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C {CSRC}`.')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.rsa_abi_version() != ABI_VERSION:
            raise RuntimeError('recstudio_amd: ABI version mismatch between header and library')
        _lib = handle
    return _lib


def check(rc, what):
    if rc != RSA_OK:
        msg = lib().rsa_last_error()
        raise NativeError(f'{what} failed (status {rc}): {msg.decode() if msg else ""}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())
