"""Layout assembler with the reference's interface (models_clevr/nmn3_assembler.py:121-222).

`Assembler(vocab_file)` exposes `.module_names .EOS_idx .name2idx_dict .num_vocab_nmn .P .W .b`,
`.module_list2tokens(list, T)` and `.assemble(tokens[T,N]) -> (expr_list, validity[N] bool)` with the
same expr-dict schema ({'module','output_type','time_idx','batch_idx','input_0','input_1'} /
{'module': 'INVALID_EXPR','expr_str','error'}) and the same error conventions: invalid layouts are
data, never exceptions; `module_list2tokens` raises ValueError('Not enough time steps to add <eos>').

Unlike the reference (a per-example Python stack machine), the Reverse-Polish decoding itself runs
in the C++ scheduler behind the C-ABI (`n2nmn_assemble`, csrc/schedule.cpp), which also produces the
packed, level-scheduled program that phase 2 executes.  `assemble()` rebuilds the reference's nested
dicts from the packed nodes for API compatibility; `assemble_packed()` skips the dicts (hot path).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import numpy as np

from . import _lib
from .spec import MODULE_INPUT_NUM, MODULE_OUTPUT_TYPE, OP_CODE, INVALID_EXPR

_ASM_ERRORS = {
    1: lambda names, op, k: 'cannot find <eos>',
    2: lambda names, op, k: 'not enough input for ' + names[op],
    3: lambda names, op, k: 'input incompatible for ' + names[op],
    4: lambda names, op, k: 'final stack size not equal to 1 (%d remains)' % k,
    5: lambda names, op, k: 'result type must be ans, not att',
}


def build_validity_mats(module_names: Sequence[str], input_num=None, output_type=None):
    """P [V,3], W [3,V,4], b [V,4] of the decoding automaton: token s is allowed in state
    x = (#att, #ans, T_remain) iff all_c(x . W[:,s,c] >= b[s,c]); emitting s adds P[s] to x
    (semantics of models_clevr/nmn3_assembler.py:50-119, vectorised)."""
    input_num = MODULE_INPUT_NUM if input_num is None else input_num
    output_type = MODULE_OUTPUT_TYPE if output_type is None else output_type
    names = list(module_names)
    V = len(names)
    is_eos = np.array([s == '<eos>' for s in names])
    arity = np.array([0 if e else input_num[s] for s, e in zip(names, is_eos)])
    att_o = np.array([0 if e else int(output_type[s] == 'att') for s, e in zip(names, is_eos)])
    ans_o = np.array([0 if e else int(output_type[s] == 'ans') for s, e in zip(names, is_eos)])
    absorb = arity - att_o
    mana = int((absorb * (ans_o == 0)).max())      # most attentions a non-answer module absorbs
    maa = int((absorb * (ans_o != 0)).max())       # ... an answer module absorbs
    P = np.stack([att_o - arity, ans_o, -np.ones(V, int)], axis=1).astype(np.int32)
    W = np.zeros((3, V, 4), np.int32)
    b = np.zeros((V, 4), np.int32)
    mod = ~is_eos
    ans = mod & (ans_o == 1)
    non = mod & (ans_o == 0)
    # c0: enough attentions on the stack            #att >= arity
    W[0, mod, 0] = 1
    b[mod, 0] = arity[mod]
    # c1: answer modules leave no attention behind  -#att >= -arity ; others need T_remain >= 3
    W[0, ans, 1] = -1
    b[ans, 1] = -arity[ans]
    W[2, non, 1] = 1
    b[non, 1] = 3
    # c2: nothing after an answer but <eos>         -#ans >= 0
    W[1, mod, 2] = -1
    # c3: enough steps left to consume every attention, answer and emit <eos>
    W[0, non, 3] = -1
    W[2, non, 3] = mana
    b[non, 3] = 3 * mana - maa - absorb[non]
    # <eos>: an answer must be on the stack         #ans >= 1
    W[1, is_eos, 0] = 1
    b[is_eos, 0] = 1
    return P, W, b


class PackedLayouts:
    """Handle of a packed, level-scheduled batch of layout trees (an `n2nmn_program`)."""

    def __init__(self):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().n2nmn_program_create(C.byref(self._h)))

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            try:
                _lib.lib().n2nmn_program_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h

    @property
    def num_nodes(self) -> int:
        return _lib.check(_lib.lib().n2nmn_program_num_nodes(self._h))

    @property
    def num_rows(self) -> int:
        return _lib.check(_lib.lib().n2nmn_program_num_rows(self._h))

    @property
    def num_levels(self) -> int:
        return _lib.check(_lib.lib().n2nmn_program_num_levels(self._h))

    @property
    def num_launches(self) -> int:
        return _lib.check(_lib.lib().n2nmn_program_num_launches(self._h))

    def nodes(self) -> np.ndarray:
        """structured array with fields op,time_idx,batch_idx,in0,in1,level,out_row,reserved"""
        n = self.num_nodes
        buf = (_lib.Node * max(n, 1))()
        _lib.check(_lib.lib().n2nmn_program_get_nodes(self._h, buf, n))
        arr = np.frombuffer(buf, dtype=np.int32).reshape(-1, 8)[:n].copy()
        return arr.view([(f, np.int32) for f, _ in _lib.Node._fields_]).reshape(n)

    def status(self, example: int):
        k, o, r = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib().n2nmn_program_status(self._h, example, C.byref(k), C.byref(o),
                                                   C.byref(r)))
        return k.value, o.value, r.value

    @classmethod
    def from_nodes(cls, nodes: Sequence[Sequence[int]], num_rows: int) -> 'PackedLayouts':
        """nodes: rows of (op, time_idx, batch_idx, in0, in1, out_row), topologically ordered."""
        self = cls()
        buf = (_lib.Node * max(len(nodes), 1))()
        for i, (op, t, n, i0, i1, row) in enumerate(nodes):
            buf[i] = _lib.Node(op, t, n, i0, i1, 0, row, 0)
        _lib.check(_lib.lib().n2nmn_program_from_nodes(self._h, buf, len(nodes), num_rows))
        return self


class ExprList(list):
    """list of expr dicts that remembers the packed program it was decoded from."""
    packed: PackedLayouts = None


class Assembler:
    def __init__(self, module_vocab_file, op_code=None, input_num=None, output_type=None):
        """op_code: module name -> C-ABI operator code; default = the models_clevr vocabulary
        (a token is its op code).  models_vqa / models_shapes pass their own maps (and, for
        modules models_clevr does not have, their arity / output-type tables)."""
        op_code = OP_CODE if op_code is None else op_code
        self._op_code = op_code
        self._input_num = MODULE_INPUT_NUM if input_num is None else input_num
        self._output_type = MODULE_OUTPUT_TYPE if output_type is None else output_type
        if isinstance(module_vocab_file, (list, tuple)):
            self.module_names = list(module_vocab_file)
        else:
            with open(module_vocab_file) as f:
                self.module_names = [line.strip() for line in f.readlines()]
        self.EOS_idx = self.module_names.index('<eos>')
        self.name2idx_dict = {name: i for i, name in enumerate(self.module_names)}
        self.num_vocab_nmn = len(self.module_names)
        self.P, self.W, self.b = build_validity_mats(self.module_names, self._input_num,
                                                     self._output_type)
        # op code of each token for the C-ABI (-1 = <eos>); KeyError on modules we do not know
        self._token_op = np.array(
            [-1 if s == '<eos>' else op_code[s] for s in self.module_names], np.int32)
        self._op_name = {op_code[s]: s for s in self.module_names if s != '<eos>'}

    # -- token helpers ----------------------------------------------------------------------
    def module_list2tokens(self, module_list, T=None):
        tokens = [self.name2idx_dict[name] for name in module_list]
        if T is not None:
            if len(module_list) >= T:
                raise ValueError('Not enough time steps to add <eos>')
            tokens += [self.EOS_idx] * (T - len(module_list))
        return tokens

    def _layout_tokens2str(self, layout_tokens):
        return ' '.join(self.module_names[int(i)] for i in layout_tokens)

    # -- assembly -----------------------------------------------------------------------------
    def assemble_packed(self, layout_tokens_batch):
        """tokens [T, N] -> (PackedLayouts, validity[N] bool); no Python objects per node."""
        toks = np.ascontiguousarray(layout_tokens_batch, dtype=np.int32)
        if toks.ndim != 2:
            raise ValueError('layout_tokens_batch must have shape [T, N]')
        T, N = toks.shape
        packed = PackedLayouts()
        validity = np.zeros(N, np.uint8)
        _lib.check(_lib.lib().n2nmn_assemble(
            packed.handle, toks.ctypes.data, T, N, self._token_op.ctypes.data,
            self.num_vocab_nmn, validity.ctypes.data))
        return packed, validity.astype(bool)

    def assemble(self, layout_tokens_batch):
        toks = np.asarray(layout_tokens_batch)
        packed, validity = self.assemble_packed(toks)
        nodes = packed.nodes()
        built = [None] * len(nodes)
        roots = {}
        for i, nd in enumerate(nodes):      # topological order: inputs precede consumers
            name = self._op_name[int(nd['op'])]
            e = {'module': name, 'output_type': self._output_type[name],
                 'time_idx': int(nd['time_idx']), 'batch_idx': int(nd['batch_idx'])}
            if nd['in0'] >= 0:
                e['input_0'] = built[int(nd['in0'])]
            if nd['in1'] >= 0:
                e['input_1'] = built[int(nd['in1'])]
            built[i] = e
            if nd['out_row'] >= 0:
                roots[int(nd['out_row'])] = e
        expr_list = ExprList()
        for n in range(toks.shape[1]):
            if validity[n]:
                expr_list.append(roots[n])
            else:
                kind, op, remains = packed.status(n)
                names = self._op_name
                expr_list.append({'module': INVALID_EXPR,
                                  'expr_str': self._layout_tokens2str(toks[:, n]),
                                  'error': _ASM_ERRORS[kind](names, op, remains)})
        expr_list.packed = packed
        return expr_list, validity

    # -- dict walk (the build_feed_dict path for hand-made expression lists) ------------------
    def pack_expr_list(self, expr_list) -> PackedLayouts:
        """nested expr dicts -> PackedLayouts (what td.Compiler.build_feed_dict did with Loom)."""
        packed = getattr(expr_list, 'packed', None)
        if packed is not None:
            return packed
        rows: List[tuple] = []

        def walk(e, out_row):
            ins = [walk(e[k], -1) for k in ('input_0', 'input_1') if k in e]
            ins += [-1] * (2 - len(ins))
            name = e['module']
            if len([k for k in ('input_0', 'input_1') if k in e]) != self._input_num[name]:
                raise ValueError('wrong number of inputs for ' + name)
            rows.append((self._op_code[name], int(e['time_idx']), int(e['batch_idx']), ins[0], ins[1],
                         out_row))
            return len(rows) - 1

        for i, e in enumerate(expr_list):
            if e['module'] != INVALID_EXPR:
                walk(e, i)
        return PackedLayouts.from_nodes(rows, len(expr_list))
