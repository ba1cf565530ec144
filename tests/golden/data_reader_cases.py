"""Deterministic synthetic imdbs in the reference's format (util/clevr_train/data_reader.py,
util/vqa_train/data_reader.py) for the data-plane tests: tiny feature files, vocabularies with and
without <unk>, Filter chains for prune_filter_module, "how many" questions, several valid answers."""
import os

import numpy as np

CLEVR_LAYOUTS = (['_Scene', '_Count'], ['_Find', '_Filter', '_Filter', '_Exist'],
                 ['_Find', '_Filter', '_Find', '_Filter', '_Filter', '_EqualNum'],
                 ['_Find', '_Transform', '_Filter', '_Describe'],
                 ['_Find', '_FindSameProperty', '_Filter', '_Filter', '_Filter', '_Count'])
VQA_LAYOUTS = (['_Find', '_Describe'], ['_Find', '_Find', '_And', '_Describe'],
               ['_Find', '_Transform', '_Describe'])
WORDS = ['<unk>', 'what', 'color', 'is', 'the', 'cube', 'how', 'many', 'things', 'are', 'there', 'left', 'of']
ANSWERS = ['<unk>', 'red', 'blue', '2', '3', 'yes', 'no']


def _write_lines(path, lines):
    with open(path, 'w') as f:
        f.write('\n'.join(lines) + '\n')


def build(root, variant, n=23, H=3, W=4, D=8, seed=0):
    """Writes vocab files + feature files under `root`, returns (imdb list of dicts, data_params
    without the assembler)."""
    rng = np.random.RandomState(seed + (0 if variant == 'clevr' else 100))
    os.makedirs(os.path.join(root, 'feat'), exist_ok=True)
    vq, va = os.path.join(root, 'vocab_q.txt'), os.path.join(root, 'vocab_a.txt')
    _write_lines(vq, WORDS)
    _write_lines(va, ANSWERS)
    imdb = []
    for i in range(n):
        fp = os.path.join(root, 'feat', '%s_%03d.npy' % (variant, i))
        np.save(fp, rng.standard_normal((1, H, W, D)).astype(np.float32))
        L = int(rng.randint(2, 9))
        toks = [WORDS[int(rng.randint(1, len(WORDS)))] for _ in range(L)]
        if i % 5 == 0:
            toks[0] = 'zebra'                      # not in the vocabulary -> <unk>
        info = dict(image_path='img_%03d.png' % i, feature_path=fp, question_tokens=toks)
        if variant == 'clevr':
            info['answer'] = ANSWERS[int(rng.randint(1, len(ANSWERS)))] if i % 7 else 'purple'
            info['gt_layout_tokens'] = list(CLEVR_LAYOUTS[i % len(CLEVR_LAYOUTS)])
        else:
            k = int(rng.randint(1, 4))
            info['valid_answers'] = [ANSWERS[int(rng.randint(1, len(ANSWERS)))] for _ in range(k)]
            info['question_id'] = 1000 + i
            info['question_str'] = ('How many ' if i % 3 == 0 else 'What ') + ' '.join(toks) + '?'
            info['gt_layout_tokens'] = list(VQA_LAYOUTS[i % len(VQA_LAYOUTS)])
            info['gt_txtatt'] = [(0, min(2, L)) if t % 2 == 0 else None
                                 for t in range(len(info['gt_layout_tokens']))]
        imdb.append(info)
    params = dict(batch_size=5, T_encoder=10, T_decoder=8, vocab_question_file=vq, vocab_answer_file=va)
    return imdb, params


class TokenTable:
    """Stand-in for `assembler` where the reference's own models_vqa Assembler cannot be built: the
    loaders only call module_list2tokens (models_vqa/nmn3_assembler.py:117-124), and the shipped
    module tables have no '_Count', which `use_count_module` substitutes for '_Describe'."""

    def __init__(self, names):
        self.names = list(names)
        self.eos = self.names.index('<eos>')

    def module_list2tokens(self, module_list, T=None):
        tokens = [self.names.index(m) for m in module_list]
        if T is not None:
            if len(module_list) >= T:
                raise ValueError('Not enough time steps to add <eos>')
            tokens += [self.eos] * (T - len(module_list))
        return tokens


COUNT_NAMES = ('_Find', '_Transform', '_And', '_Describe', '_Count', '<eos>')

# the cases both the golden generator and the test run: (name, variant, reader kwargs, extra params)
CASES = (
    ('clevr_plain', 'clevr', dict(shuffle=False, one_pass=True), dict()),
    ('clevr_prune_shuffle', 'clevr', dict(shuffle=True, one_pass=False), dict(prune_filter_module=True)),
    ('vqa_plain', 'vqa', dict(shuffle=False, one_pass=True), dict(load_binary_labels=True)),
    ('vqa_count_shuffle', 'vqa', dict(shuffle=True, one_pass=False),
     dict(use_count_module=True, load_gt_txtatt=True)),
    ('vqa_override', 'vqa', dict(shuffle=False, one_pass=True),
     dict(overriding_layout=['_Find', '_Describe'])),
)
NUM_BATCHES = 12          # > 2 epochs of 23 samples in batches of 5 for the endless readers
ARRAY_KEYS = ('input_seq_batch', 'seq_length_batch', 'image_feat_batch', 'answer_label_batch',
              'gt_layout_batch', 'answer_binarylabel_batch', 'gt_txtatt_batch')
LIST_KEYS = ('image_path_list', 'qid_list', 'qstr_list', 'valid_answers_list', 'all_answers_list')


def digest(batches):
    """A canonical description of a sequence of batches: key sets, shapes, dtypes and sha256 of the
    array bytes / repr of the lists."""
    import hashlib
    out = []
    for b in batches:
        d = {}
        for k in sorted(b):
            v = b[k]
            if isinstance(v, np.ndarray):
                a = np.ascontiguousarray(v)
                dt = 'bool' if a.dtype == bool else str(a.dtype)
                d[k] = [list(a.shape), dt, hashlib.sha256(a.tobytes()).hexdigest()[:24]]
            else:
                d[k] = hashlib.sha256(repr([str(x) if not isinstance(x, list) else [str(y) for y in x]
                                            for x in v]).encode()).hexdigest()[:24]
        out.append(d)
    return out
