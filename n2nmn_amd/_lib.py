"""ctypes binding of include/n2nmn.h (the C-ABI boundary).  No torch types cross it: tensors are
passed as raw device pointers (`tensor.data_ptr()`), streams as `hipStream_t` handles.

The library is REQUIRED: there is no CPU or PyTorch fallback.  If the shared object is missing and
cannot be built, importing this module's `lib()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_LIB = None


class Dims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'H', 'W', 'D', 'map_dim', 'embed_dim_txt', 'embed_dim_nmn', 'lstm_dim', 'num_layers',
        'num_vocab_txt', 'num_vocab_nmn', 'num_choices', 'T_encoder', 'T_decoder', 'N',
        'kernel_size', 'variant', 'qpn_hidden')]


class Seq2SeqIO(C.Structure):
    _fields_ = [
        ('input_seq', C.c_void_p), ('seq_length', C.c_void_p),
        ('T_enc', C.c_int32), ('N', C.c_int32), ('T_dec', C.c_int32),
        ('use_gt_layout', C.c_int32),
        ('gt_layout', C.c_void_p), ('sample_uniforms', C.c_void_p), ('forced_tokens', C.c_void_p),
        ('predicted_tokens', C.c_void_p), ('token_probs', C.c_void_p), ('neg_entropy', C.c_void_p),
        ('atts', C.c_void_p), ('word_vecs', C.c_void_p), ('token_scores', C.c_void_p),
        ('encoder_outputs', C.c_void_p), ('encoder_h_transformed', C.c_void_p),
        ('encoder_states', C.c_void_p), ('log_seq_prob', C.c_void_p), ('flags', C.c_int32),
        ('image_feat', C.c_void_p), ('drop_enc0', C.c_void_p), ('drop_dec0', C.c_void_p),
        ('seq_length_host', C.c_void_p), ('gt_length_host', C.c_void_p)]


class TrainIO(C.Structure):
    _fields_ = [
        ('input_seq', C.c_void_p), ('seq_length', C.c_void_p),
        ('T_enc', C.c_int32), ('N', C.c_int32), ('T_dec', C.c_int32),
        ('gt_layout', C.c_void_p), ('image_feat', C.c_void_p), ('answer_labels', C.c_void_p),
        ('weight_decay', C.c_float),
        ('scores', C.c_void_p), ('losses', C.c_void_p), ('grads', C.c_void_p),
        ('objective', C.c_int32), ('expr_validity', C.c_void_p),
        ('invalid_expr_loss', C.c_float), ('lambda_entropy', C.c_float),
        ('baseline_decay', C.c_float), ('baseline', C.c_void_p),
        ('drop_enc0', C.c_void_p), ('drop_dec0', C.c_void_p), ('drop_qpn_h', C.c_void_p),
        ('drop_qpn_fc1', C.c_void_p)]


class Node(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        'op', 'time_idx', 'batch_idx', 'in0', 'in1', 'level', 'out_row', 'reserved')]


class WalkBatch(C.Structure):
    """n2nmn_walk_batch (include/n2nmn.h section 4b)"""
    _fields_ = [('ctx', C.c_void_p), ('tokens', C.c_void_p), ('image_feat', C.c_void_p),
                ('word_vecs', C.c_void_p), ('scores', C.c_void_p), ('validity', C.c_void_p),
                ('atts', C.c_void_p), ('input_seq', C.c_void_p), ('seq_length', C.c_void_p)]


ERRORS = {-1: 'N2NMN_EINVAL', -2: 'N2NMN_EHIP', -3: 'N2NMN_ENOWEIGHT', -4: 'N2NMN_ECAPACITY',
          -5: 'N2NMN_EKEY'}

# every symbol include/n2nmn.h declares: (name, restype, argtypes)
_P = C.c_void_p
_I = C.c_int
SYMBOLS = [
    ('n2nmn_last_error', C.c_char_p, []),
    ('n2nmn_version', C.c_char_p, []),
    ('n2nmn_ctx_create', _I, [C.POINTER(Dims), _I, C.POINTER(_P)]),
    ('n2nmn_ctx_fork', _I, [_P, C.POINTER(_P)]),
    ('n2nmn_ctx_destroy', _I, [_P]),
    ('n2nmn_ctx_dims', _I, [_P, C.POINTER(Dims)]),
    ('n2nmn_ctx_set_mode', _I, [_P, _I]),
    ('n2nmn_set_weight', _I, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), _I]),
    ('n2nmn_commit_weights', _I, [_P, _P]),
    ('n2nmn_set_validity_tables', _I, [_P, _P, _P, _P]),
    ('n2nmn_num_variables', _I, [_P]),
    ('n2nmn_variable_info', _I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                                 C.POINTER(_I)]),
    ('n2nmn_encoder_forward', _I, [_P, C.POINTER(Seq2SeqIO), _P]),
    ('n2nmn_decoder_forward', _I, [_P, C.POINTER(Seq2SeqIO), _P]),
    ('n2nmn_seq2seq_forward', _I, [_P, C.POINTER(Seq2SeqIO), _P]),
    ('n2nmn_program_create', _I, [C.POINTER(_P)]),
    ('n2nmn_program_destroy', _I, [_P]),
    ('n2nmn_assemble', _I, [_P, _P, _I, _I, _P, _I, _P]),
    ('n2nmn_program_from_nodes', _I, [_P, C.POINTER(Node), _I, _I]),
    ('n2nmn_program_num_nodes', _I, [_P]),
    ('n2nmn_program_num_rows', _I, [_P]),
    ('n2nmn_program_num_levels', _I, [_P]),
    ('n2nmn_program_get_nodes', _I, [_P, C.POINTER(Node), _I]),
    ('n2nmn_program_status', _I, [_P, _I, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32)]),
    ('n2nmn_program_num_launches', _I, [_P]),
    ('n2nmn_execute_program', _I, [_P, _P, _P, _P, _I, _P, _P]),
    ('n2nmn_module_forward', _I, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _P, _P]),
    ('n2nmn_set_token_ops', _I, [_P, _P, _I]),
    ('n2nmn_walk_supported', _I, [_P]),
    ('n2nmn_walk_set_defer_pool', _I, [_P, _I]),
    ('n2nmn_walk_set_front_end', _I, [_P, _I]),
    ('n2nmn_walk_set_staged', _I, [_P, _I]),
    ('n2nmn_walk_set_levels', _I, [_P, _I]),
    ('n2nmn_walk_set_nesting_bound', _I, [_P, _I]),
    ('n2nmn_walk_set_conv_inline', _I, [_P, _I]),
    ('n2nmn_conv_image', _I, [_P, _P, _I, _I, _P, _I, _P]),
    ('n2nmn_walk_layouts', _I, [_P, C.POINTER(WalkBatch), _I, _I, _I, _I, _P]),
    ('n2nmn_execute_tokens', _I, [_P, _P, _I, _I, _P, _P, _P, _P, _P]),
    ('n2nmn_set_tokens_via_levels', _I, [_P, _I]),
    ('n2nmn_add_coords', _I, [_P, _P, _I, _I, _P, _P]),
    ('n2nmn_question_prior_add', _I, [_P, _I, _P, _P]),
    ('n2nmn_train_enable', _I, [_P]),
    ('n2nmn_grad_numel', C.c_int64, [_P]),
    ('n2nmn_grad_split', C.c_int64, [_P]),
    ('n2nmn_grad_layout', _I, [_P, _I, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    ('n2nmn_train_forward', _I, [_P, C.POINTER(TrainIO), _P, _P]),
    ('n2nmn_train_backward', _I, [_P, C.POINTER(TrainIO), _P, _I, _P]),
    ('n2nmn_train_join', _I, [_P, _P]),
    ('n2nmn_adam_step', _I, [_P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                             C.c_float, C.c_int64, _P]),
    ('n2nmn_dropout_multipliers', _I, [_P, C.c_int64, C.c_float, C.c_uint64, C.c_uint64, _P]),
    ('n2nmn_train_reset_optimizer', _I, [_P, _P]),
    ('n2nmn_get_weight', _I, [_P, C.c_char_p, _P, _P]),
    ('n2nmn_train_debug_tensor', C.c_int64, [_P, C.c_char_p, _P, C.c_int64, _P]),
    ('n2nmn_comm_unique_id', _I, [_P]),
    ('n2nmn_comm_create', _I, [_P, _I, _I, _I, C.POINTER(_P)]),
    ('n2nmn_comm_world', _I, [_P]),
    ('n2nmn_allreduce_grads', _I, [_P, _P, _I, _P, _P]),
    ('n2nmn_allreduce_wait', _I, [_P, _P]),
    ('n2nmn_comm_destroy', _I, [_P]),
    ('n2nmn_profile_begin', _I, [_P]),
    ('n2nmn_profile_end', _I, [_P, _P]),
    ('n2nmn_profile_num_families', _I, []),
    ('n2nmn_profile_get', _I, [_P, _I, C.POINTER(C.c_char_p), C.POINTER(C.c_int64),
                               C.POINTER(C.c_double), C.POINTER(C.c_double),
                               C.POINTER(C.c_double)]),
    ('n2nmn_debug_walk_stats', _I, [_P, C.POINTER(C.c_uint64)]),
    ('n2nmn_debug_lstm_bench', _I, [_P, _I, _I, _I, _I, _I, C.POINTER(C.c_double), _P]),
    ('n2nmn_debug_gemm_tn', _I, [_P, _P, _I, _I, _P, _I, _I, _I, _P, _I, _P, _P, _P, _I, _P]),
    ('n2nmn_debug_colsum', _I, [_P, _P, _I, _I, _I, _P, _I, _P, _P]),
    ('n2nmn_debug_event_overhead', _I, [_P, _I, C.POINTER(C.c_double), _P]),
    ('n2nmn_debug_walk_replay', _I, [_P, _I, _I, C.POINTER(C.c_double), _P]),
    ('n2nmn_debug_walk_timeline', _I, [_P, _P]),
    ('n2nmn_debug_gemm', _I, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    ('n2nmn_fc_forward', _I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    ('n2nmn_crc32c', C.c_uint32, [C.c_uint32, C.c_char_p, C.c_size_t]),
    ('n2nmn_debug_set', _I, [_P, C.c_char_p, C.c_char_p]),
    ('n2nmn_automaton_forces_eos', _I, [_P, _P, _P, _P, _I, _I]),
]


def lib_path() -> str:
    return _build.LIB


def lib():
    """Load (building in-tree first if the sources are newer) libn2nmn_hip.so."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if _build.is_stale():
        try:
            _build.build(verbose=False)
        except Exception as e:  # no silent fallback: the HIP extension is the product
            if not os.path.exists(_build.LIB):
                raise RuntimeError(
                    'n2nmn_amd: the HIP extension %s is missing and could not be built (%s). '
                    'There is no CPU fallback.' % (_build.LIB, e))
            # an older binary exists but csrc/ is newer and does not compile: running it would test
            # kernels and ctypes structs that no longer match the sources
            if os.environ.get('N2NMN_ALLOW_STALE_LIB') != '1':
                raise RuntimeError(
                    'n2nmn_amd: %s is older than csrc/ and the rebuild failed (%s); fix the build '
                    'or set N2NMN_ALLOW_STALE_LIB=1 to load the stale library anyway'
                    % (_build.LIB, e))
            import warnings
            warnings.warn('n2nmn_amd: loading a STALE %s (rebuild failed: %s)' % (_build.LIB, e))
    # PyTorch's bundled HIP runtime must be the process's: loaded after this library's own dependency
    # (the system libamdhip64) it would be a second runtime, and this library's would see no device
    # (build() followed by smoke() in one process did that)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_build.LIB)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)       # AttributeError if the .so does not export the symbol
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


class N2nmnError(RuntimeError):
    pass


def check(rc: int):
    """Map the C error conventions onto the reference's exception types:
    bad arguments / capacity -> ValueError, unknown names -> KeyError, the rest RuntimeError."""
    if rc >= 0:
        return rc
    msg = lib().n2nmn_last_error().decode(errors='replace')
    text = '%s: %s' % (ERRORS.get(rc, str(rc)), msg)
    if rc in (-1, -4):
        raise ValueError(text)
    if rc == -5:
        raise KeyError(text)
    raise N2nmnError(text)
