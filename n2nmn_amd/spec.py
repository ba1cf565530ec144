"""Static description of the N2NMN CLEVR hot path: dimensions, module tables and
the reference's variable names.

Everything here restates *facts* of the reference (not its code):

* module arity / output type tables ........ models_clevr/nmn3_assembler.py:9-41
* layout vocabulary (token order) ........... exp_clevr/data/vocabulary_layout.txt
* eval-time dimensions ....................... exp_clevr/eval_clevr.py:27-37
* variable names (TF 1.0.0 scoping) .......... models_clevr/nmn3_model.py:22-49,
  models_clevr/nmn3_netgen_att.py:82-86,102-103,139-156, models_clevr/nmn3_modules.py
  (scope= arguments of every module), util/cnn.py:19-25,104-109,
  util/empty_safe_conv.py:24-27.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, List, Tuple

# --- layout vocabulary ------------------------------------------------------------------
# exp_clevr/data/vocabulary_layout.txt (token index == line number)
CLEVR_MODULE_NAMES: Tuple[str, ...] = (
    '_Scene', '_Find', '_Filter', '_FindSameProperty', '_Transform', '_And', '_Or',
    '_Exist', '_Count', '_EqualNum', '_MoreNum', '_LessNum', '_SameProperty',
    '_Describe', '<eos>')

# models_clevr/nmn3_assembler.py:9-24 -- number of attention inputs of each module
MODULE_INPUT_NUM: Dict[str, int] = {
    '_Scene': 0, '_Find': 0, '_Filter': 1, '_FindSameProperty': 1, '_Transform': 1,
    '_And': 2, '_Or': 2, '_Count': 1, '_Exist': 1, '_EqualNum': 2, '_MoreNum': 2,
    '_LessNum': 2, '_SameProperty': 2, '_Describe': 1}

# models_clevr/nmn3_assembler.py:26-41 -- output type of each module
MODULE_OUTPUT_TYPE: Dict[str, str] = {
    '_Scene': 'att', '_Find': 'att', '_Filter': 'att', '_FindSameProperty': 'att',
    '_Transform': 'att', '_And': 'att', '_Or': 'att', '_Count': 'ans', '_Exist': 'ans',
    '_EqualNum': 'ans', '_MoreNum': 'ans', '_LessNum': 'ans', '_SameProperty': 'ans',
    '_Describe': 'ans'}

INVALID_EXPR = 'INVALID_EXPR'   # models_clevr/nmn3_assembler.py:43

# Operator codes shared with the C-ABI (include/n2nmn.h, enum n2nmn_op).  The numeric value
# is the CLEVR layout-token index of the module, so a token *is* its op code.
OP_CODE: Dict[str, int] = {name: i for i, name in enumerate(CLEVR_MODULE_NAMES[:-1])}
OP_INVALID = -1


@dataclass(frozen=True)
class Dims:
    """Tensor dimensions of one model instance (defaults = exp_clevr/eval_clevr.py:27-37)."""
    H: int = 10
    W: int = 15
    D: int = 512              # image feature channels (VGG pool5)
    map_dim: int = 250        # models_clevr/nmn3_modules.py:74 (map_dim=250 everywhere)
    embed_dim_txt: int = 300
    embed_dim_nmn: int = 300
    lstm_dim: int = 512
    num_layers: int = 2
    num_vocab_txt: int = 82   # exp_clevr/data/vocabulary_clevr.txt
    num_vocab_nmn: int = 15   # exp_clevr/data/vocabulary_layout.txt
    num_choices: int = 28     # exp_clevr/data/answers_clevr.txt
    T_encoder: int = 45
    T_decoder: int = 20
    N: int = 64
    kernel_size: int = 5      # models_clevr/nmn3_modules.py:185 (TransformModule)
    variant: int = 0          # 0: models_clevr, 1: models_vqa (n2nmn_amd/vqa.py)
    qpn_hidden: int = 0       # models_vqa/question_prior_net.py hidden width (500); 0 = no QPN

    @property
    def HW(self) -> int:
        return self.H * self.W

    def asdict(self):
        return asdict(self)


PREFIX = 'neural_module_network/'
_ENC = PREFIX + 'layout_generation/encoder_decoder/encoder/'
_DEC = PREFIX + 'layout_generation/encoder_decoder/decoder/'
_MOD = PREFIX + 'layout_execution/module_variables/'


def lstm_var(enc_or_dec: str, layer: int, kind: str) -> str:
    """TF 1.0.0 name of a BasicLSTMCell variable inside MultiRNNCell."""
    base = _ENC if enc_or_dec == 'encoder' else _DEC
    return '%slstm/multi_rnn_cell/cell_%d/basic_lstm_cell/%s' % (base, layer, kind)


def variable_shapes(d: Dims) -> Dict[str, Tuple[int, ...]]:
    """name -> shape of every trainable variable on the path, in the reference's layout
    (fc / 1x1 weights [in,out]; conv weights [kh,kw,in,out]; LSTM [in+hidden, 4*hidden])."""
    L, E, En, M, C, D, HW = (d.lstm_dim, d.embed_dim_txt, d.embed_dim_nmn, d.map_dim,
                             d.num_choices, d.D, d.HW)
    if d.num_layers != 2:
        raise ValueError('the hot path is built for num_layers == 2 (reference eval config)')
    s: Dict[str, Tuple[int, ...]] = {}
    # --- encoder (models_clevr/nmn3_netgen_att.py:73-113)
    s[_ENC + 'embedding_mat'] = (d.num_vocab_txt, E)
    s[lstm_var('encoder', 0, 'weights')] = (E + L, 4 * L)
    s[lstm_var('encoder', 0, 'biases')] = (4 * L,)
    s[lstm_var('encoder', 1, 'weights')] = (2 * L, 4 * L)
    s[lstm_var('encoder', 1, 'biases')] = (4 * L,)
    s[_ENC + 'encoder_h_transform/weights'] = (L, L)
    s[_ENC + 'encoder_h_transform/biases'] = (L,)
    # --- decoder (models_clevr/nmn3_netgen_att.py:139-156,303)
    s[_DEC + 'embedding_mat'] = (d.num_vocab_nmn, En)
    s[_DEC + 'go_embedding'] = (1, En)
    s[_DEC + 'att_prediction/v'] = (L,)
    s[_DEC + 'att_prediction/weights'] = (L, L)
    s[_DEC + 'att_prediction/biases'] = (L,)
    s[_DEC + 'token_prediction/weights'] = (2 * L, d.num_vocab_nmn)
    s[_DEC + 'token_prediction/biases'] = (d.num_vocab_nmn,)
    s[lstm_var('decoder', 0, 'weights')] = (En + L, 4 * L)
    s[lstm_var('decoder', 0, 'biases')] = (4 * L,)
    s[lstm_var('decoder', 1, 'weights')] = (2 * L, 4 * L)
    s[lstm_var('decoder', 1, 'biases')] = (4 * L,)
    # --- modules (models_clevr/nmn3_modules.py)

    def layer(scope, name, shape):
        s[_MOD + scope + '/' + name + '/weights'] = shape
        s[_MOD + scope + '/' + name + '/biases'] = (shape[-1],)

    layer('FindModule', 'conv_image', (D, M))
    layer('FindModule', 'fc_text', (E, M))
    layer('FindModule', 'conv_eltwise', (M, 1))
    layer('FindSamePropertyModule', 'conv_image', (D, M))
    layer('FindSamePropertyModule', 'fc_text', (E, M))
    layer('FindSamePropertyModule', 'fc_att', (D, M))
    layer('FindSamePropertyModule', 'conv_eltwise', (M, 1))
    layer('TransformModule', 'conv_maps', (d.kernel_size, d.kernel_size, 1, M))
    layer('TransformModule', 'text_fc', (E, M))
    layer('TransformModule', 'conv_eltwise', (M, 1))
    layer('ExistModule', 'fc_scores', (3, C))
    layer('CountModule', 'fc_scores', (HW + 2, C))
    layer('EqualNumModule', 'fc_scores', (2 * HW + 4, C))
    layer('MoreNumModule', 'fc_scores', (2 * HW + 4, C))
    layer('LessNumModule', 'fc_scores', (2 * HW + 4, C))
    layer('SamePropertyModule', 'fc_text', (E, M))
    layer('SamePropertyModule', 'fc_att_0', (D, M))
    layer('SamePropertyModule', 'fc_att_1', (D, M))
    layer('SamePropertyModule', 'fc_eltwise', (M, C))
    layer('DescribeModule', 'fc_text', (E, M))
    layer('DescribeModule', 'fc_att', (D, M))
    layer('DescribeModule', 'fc_eltwise', (M, C))
    return s


def num_parameters(d: Dims) -> int:
    n = 0
    for shp in variable_shapes(d).values():
        k = 1
        for v in shp:
            k *= v
        n += k
    return n


# config-2 layout templates (cycled n mod 10); derived from
# exp_clevr/data/get_ground_truth_layout.py:4-37,90-96 + util/clevr_train/data_reader.py:65-71.
CLEVR_LAYOUT_TEMPLATES: Tuple[Tuple[str, ...], ...] = (
    ('_Find', '_Count'),
    ('_Find', '_Exist'),
    ('_Find', '_Describe'),
    ('_Find', '_Transform', '_Filter', '_Describe'),
    ('_Find', '_FindSameProperty', '_Count'),
    ('_Find', '_Find', '_EqualNum'),
    ('_Find', '_Find', '_MoreNum'),
    ('_Find', '_Find', '_SameProperty'),
    ('_Find', '_Transform', '_Find', '_Transform', '_And', '_Filter', '_Count'),
    ('_Find', '_Find', '_Or', '_Exist'),
)
