"""Data plane either side of the hot path: question / feature / layout
batches in the reference's format, and their way into the GPU's super-bucket slots.

Host half -- what the reference's readers produce, key for key:
    util/text_processing.py                 tokenize, VocabDict (<unk> fallback)
    util/clevr_train/data_reader.py:10-82   BatchLoaderClevr.load_one_batch (prune_filter_module)
    util/vqa_train/data_reader.py:11-155    BatchLoaderVqa.load_one_batch (random valid answer,
                                            binary labels, overriding_layout, use_count_module,
                                            gt_txtatt)
    .../data_reader.py DataReader / _run_prefetch   epoch shuffling, short last batch, one_pass
The imdb is the reference's: a .npy array of dicts with `question_tokens`, `feature_path`,
`image_path` (+ `answer` | `valid_answers`, `gt_layout_tokens`, `question_id`, `question_str`,
`gt_txtatt`).  Randomness (epoch permutation, the sampled valid answer) comes from one
`numpy.random.RandomState`-like object in the reference's order of draws; the default is numpy's
global generator, which the reference uses, so `np.random.seed(s)` reproduces its batches.
Differences on purpose: the generator ends with `return` (the reference's `raise StopIteration()`
inside a generator is a RuntimeError since Python 3.7), and a reader can be closed.

Device half (MI355X): `DeviceFeeder` moves each batch through pinned host staging into a slot of a
`SuperBucket` on a copy stream of its own, so the H2D copy of batch i+1 (19.7 MB of CLEVR features
= 0.3 ms on PCIe Gen5) runs under the compute of batch i; the consumer orders its stream after the
slot's copy event and never blocks the host.
"""
from __future__ import annotations

import queue
import re
import threading
from typing import Dict, Iterable, Iterator, List, Optional

import numpy as np

_SPLIT = re.compile(r'(\W+)')


def tokenize(sentence: str) -> List[str]:
    """util/text_processing.py:3-7"""
    return [t.strip() for t in _SPLIT.split(sentence.lower()) if t.strip()]


class VocabDict:
    """util/text_processing.py:15-35: one word per line; unknown words map to <unk> when the
    vocabulary has it, otherwise ValueError."""

    def __init__(self, vocab_file: str):
        with open(vocab_file) as f:
            self.word_list = [l.strip() for l in f.readlines()]
        self.word2idx_dict = {w: i for i, w in enumerate(self.word_list)}
        self.num_vocab = len(self.word_list)
        self.UNK_idx = self.word2idx_dict.get('<unk>')

    def idx2word(self, n_w: int) -> str:
        return self.word_list[n_w]

    def word2idx(self, w: str) -> int:
        i = self.word2idx_dict.get(w, self.UNK_idx)
        if i is None:
            raise ValueError('word %s not in dictionary (while dictionary does not contain <unk>)' % w)
        return i

    def tokenize_and_index(self, sentence: str) -> List[int]:
        return [self.word2idx(w) for w in tokenize(sentence)]


def prune_filter_modules(tokens: Iterable[str]) -> List[str]:
    """util/clevr_train/data_reader.py:64-70: of consecutive _Find/_Filter ... _Filter runs only the
    first module stays (scanned from the end, like the reference)."""
    t = list(tokens)
    for i in range(len(t) - 1, 0, -1):
        if t[i - 1] in ('_Filter', '_Find') and t[i] == '_Filter':
            t[i] = None
    return [x for x in t if x]


class _Loader:
    """What both reference loaders share: question indices, lengths, features, paths."""
    answer_key = 'answer'

    def __init__(self, imdb, data_params: Dict, rng=None):
        self.imdb = imdb
        self.data_params = data_params
        self.rng = rng if rng is not None else np.random
        self.vocab_dict = VocabDict(data_params['vocab_question_file'])
        self.answer_dict = VocabDict(data_params['vocab_answer_file'])     # always loaded
        self.T_encoder = data_params['T_encoder']
        first = imdb[0]
        self.load_answer = first.get(self.answer_key) is not None
        self.load_gt_layout = data_params.get('load_gt_layout', first.get('gt_layout_tokens') is not None)
        self.feat_H, self.feat_W, self.feat_D = np.load(first['feature_path']).shape[1:]

    # hooks ------------------------------------------------------------------------------------
    def _layout_tokens(self, iminfo) -> List[str]:
        raise NotImplementedError

    def _extra(self, batch, n, iminfo):
        pass

    def _finish(self, batch):
        pass

    def load_one_batch(self, sample_ids, out: Optional[Dict[str, np.ndarray]] = None):
        """out (optional): preallocated arrays (e.g. pinned) for input_seq_batch / seq_length_batch /
        image_feat_batch (/ answer_label_batch / gt_layout_batch) of at least this batch size; the
        returned batch then holds views of them."""
        nb = len(sample_ids)

        def arr(key, shape, dtype):
            if out is not None and key in out:
                a = out[key]
                a = a[..., :nb] if key in ('input_seq_batch', 'gt_layout_batch') else a[:nb]
                if key != 'image_feat_batch':    # (every image row is overwritten below)
                    a[...] = 0
                return a
            return np.zeros(shape, dtype)

        batch = dict(
            input_seq_batch=arr('input_seq_batch', (self.T_encoder, nb), np.int32),
            seq_length_batch=arr('seq_length_batch', (nb,), np.int32),
            image_feat_batch=arr('image_feat_batch', (nb, self.feat_H, self.feat_W, self.feat_D),
                                 np.float32),
            image_path_list=[None] * nb)
        if self.load_answer:
            batch['answer_label_batch'] = arr('answer_label_batch', (nb,), np.int32)
        if self.load_gt_layout:
            batch['gt_layout_batch'] = arr('gt_layout_batch', (self.T_decoder, nb), np.int32)
        self._begin(batch, nb)
        for n, sid in enumerate(sample_ids):
            iminfo = self.imdb[sid]
            inds = [self.vocab_dict.word2idx(w) for w in iminfo['question_tokens']]
            batch['input_seq_batch'][:len(inds), n] = inds
            batch['seq_length_batch'][n] = len(inds)
            batch['image_feat_batch'][n:n + 1] = np.load(iminfo['feature_path'])
            batch['image_path_list'][n] = iminfo['image_path']
            self._extra(batch, n, iminfo)
            if self.load_gt_layout:
                batch['gt_layout_batch'][:, n] = self.assembler.module_list2tokens(
                    self._layout_tokens(iminfo), self.T_decoder)
        self._finish(batch)
        return batch

    def _begin(self, batch, nb):
        pass


class BatchLoaderClevr(_Loader):
    """util/clevr_train/data_reader.py:10-82"""

    def __init__(self, imdb, data_params, rng=None):
        super().__init__(imdb, data_params, rng)
        if self.load_gt_layout:
            self.T_decoder = data_params['T_decoder']
            self.assembler = data_params['assembler']
            self.prune_filter_module = data_params.get('prune_filter_module', False)

    def _layout_tokens(self, iminfo):
        t = iminfo['gt_layout_tokens']
        return prune_filter_modules(t) if self.prune_filter_module else t

    def _extra(self, batch, n, iminfo):
        if self.load_answer:
            batch['answer_label_batch'][n] = self.answer_dict.word2idx(iminfo['answer'])


class BatchLoaderVqa(_Loader):
    """util/vqa_train/data_reader.py:11-155"""
    answer_key = 'valid_answers'

    def __init__(self, imdb, data_params, rng=None):
        super().__init__(imdb, data_params, rng)
        first = imdb[0]
        self.load_gt_txtatt = data_params.get('load_gt_txtatt', first.get('gt_txtatt') is not None)
        self.num_choices = self.answer_dict.num_vocab
        self.load_binary_labels = bool(self.load_answer and data_params.get('load_binary_labels'))
        self.overriding_layout = data_params.get('overriding_layout')
        if self.overriding_layout is not None:       # :45-50
            self.load_gt_layout = True
            self.load_gt_txtatt = False
        if self.load_gt_layout:
            self.T_decoder = data_params['T_decoder']
            self.assembler = data_params['assembler']
        self.use_count_module = bool(data_params.get('use_count_module'))

    def _begin(self, batch, nb):
        batch['qid_list'] = [None] * nb
        batch['qstr_list'] = [None] * nb
        if self.load_answer:
            batch['valid_answers_list'] = [None] * nb
            batch['all_answers_list'] = [None] * nb
            if self.load_binary_labels:
                batch['answer_binarylabel_batch'] = np.zeros((nb, self.num_choices), np.float32)
        if self.load_gt_txtatt:
            batch['gt_txtatt_batch'] = np.zeros((self.T_decoder, self.T_encoder, nb, 1), bool)

    def _layout_tokens(self, iminfo):
        if self.overriding_layout is not None:
            return self.overriding_layout
        t = list(iminfo['gt_layout_tokens'])
        if self.use_count_module and 'how many' in iminfo['question_str'].lower():
            assert t[-1] == '_Describe'              # :121-125
            t[-1] = '_Count'
        return t

    def _extra(self, batch, n, iminfo):
        batch['qid_list'][n] = iminfo['question_id']
        batch['qstr_list'][n] = iminfo['question_str']
        if self.load_answer:
            valid = iminfo['valid_answers']
            batch['valid_answers_list'][n] = valid
            batch['all_answers_list'][n] = valid
            answer = self.rng.choice(valid)          # :107: one of the valid answers, at random
            batch['answer_label_batch'][n] = self.answer_dict.word2idx(answer)
            if self.load_binary_labels:
                batch['answer_binarylabel_batch'][n, [self.answer_dict.word2idx(a) for a in valid]] = 1.
        if self.load_gt_txtatt:
            for t_dec, ind in enumerate(iminfo['gt_txtatt']):
                if ind is not None:
                    batch['gt_txtatt_batch'][t_dec, ind[0]:ind[1], n, 0] = True

    def _finish(self, batch):
        # key order of the reference's dict (:139-153) does not matter; nothing to do
        pass


class DataReader:
    """DataReader(imdb_file, shuffle=True, one_pass=False, prefetch_num=8, **data_params) of both
    reference readers; `variant` picks the loader ('clevr' | 'vqa')."""

    def __init__(self, imdb_file, shuffle=True, one_pass=False, prefetch_num=8, variant='clevr',
                 rng=None, imdb=None, **kwargs):
        if imdb is None:
            if not str(imdb_file).endswith('.npy'):
                raise TypeError('unknown imdb format.')
            imdb = np.load(imdb_file, allow_pickle=True)
        self.imdb = imdb
        self.shuffle, self.one_pass, self.prefetch_num = shuffle, one_pass, prefetch_num
        self.data_params = kwargs
        self.rng = rng if rng is not None else np.random
        cls = {'clevr': BatchLoaderClevr, 'vqa': BatchLoaderVqa}[variant]
        self.batch_loader = cls(self.imdb, self.data_params, self.rng)
        self._stop = threading.Event()
        self._staging: Optional['queue.Queue'] = None     # free staging sets (use_staging)
        self.prefetch_queue: 'queue.Queue' = queue.Queue(maxsize=prefetch_num)
        self.prefetch_thread = threading.Thread(target=self._run_prefetch, daemon=True)
        self.prefetch_thread.start()

    def sample_order(self) -> Iterator[np.ndarray]:
        """The reference's _run_prefetch schedule: a fresh permutation per epoch when shuffling,
        batches of batch_size with a short last one, None after each epoch when one_pass."""
        n, bs = len(self.imdb), self.data_params['batch_size']
        while True:
            order = self.rng.permutation(n) if self.shuffle else np.arange(n)
            for i in range(0, n, bs):
                yield order[i:i + bs]
            if self.one_pass:
                yield None

    def _put(self, item) -> bool:
        while not self._stop.is_set():
            try:
                self.prefetch_queue.put(item, timeout=0.1)
                return True
            except queue.Full:
                continue
        return False

    def use_staging(self, sets) -> 'queue.Queue':
        """From now on the prefetch thread builds every batch directly inside one of `sets` (dicts of
        preallocated -- e.g. pinned -- arrays for load_one_batch(out=...)), marks the batch with
        batch['_staging'] = that dict, and waits for a free one when all are out.  The consumer hands
        a set back with the returned queue's put() once it no longer reads it.  Needs more sets than
        prefetch_num + whatever the consumer holds at a time."""
        q: 'queue.Queue' = queue.Queue()
        for st in sets:
            q.put(st)
        self._staging = q
        return q

    def _take_staging(self):
        q = self._staging
        while q is not None and not self._stop.is_set():
            try:
                return q.get(timeout=0.1)
            except queue.Empty:
                continue
        return None

    def _run_prefetch(self):
        try:
            for ids in self.sample_order():
                item = None
                if ids is not None:
                    st = self._take_staging()
                    item = self.batch_loader.load_one_batch(ids, out=st)
                    if st is not None:
                        item['_staging'] = st
                if not self._put(item):
                    return
        except Exception as e:           # surface loader errors in the consumer, not in a dead thread
            self._put(e)

    def batches(self):
        while True:
            batch = self.prefetch_queue.get(block=True)
            if isinstance(batch, Exception):
                raise batch
            if batch is None:
                return
            yield batch

    def close(self, timeout: float = 5.0):
        """Stop the prefetch thread (it may be in the middle of a batch: with the default global
        numpy generator it keeps drawing random numbers until it has finished that batch)."""
        self._stop.set()
        if self.prefetch_thread is not threading.current_thread():
            self.prefetch_thread.join(timeout)


class DeviceFeeder:
    """Batches of a DataReader -> slots of SuperBuckets, through pinned staging and a copy stream.

        feeder = DeviceFeeder(reader, [bucket_a, bucket_b])
        for group in feeder.groups():          # up to K batches, their copies already enqueued
            feeder.wait(group)                 # order the compute stream after the group's copies
            group.bucket.run(...)
            for k, batch in enumerate(group.batches): scores_k = group.bucket.result(k)[0]

    The buckets take turns: while the GPU works on group g in one bucket, the host fills the pinned
    staging set of group g+1 and its copies run on the copy stream into the other bucket.  When the
    generator is resumed it records an event on the consumer's stream (the work the consumer
    enqueued for the group it just had), and the copies that next overwrite that bucket wait for
    it: neither side ever blocks the host except to reuse a staging set whose copies are still in
    flight."""

    class Group:
        def __init__(self, batches, event, bucket, index):
            self.batches, self.event, self.bucket, self.index = batches, event, bucket, index

    def __init__(self, reader: DataReader, buckets, use_gt_layout: bool = False):
        import torch
        self.torch = torch
        self.reader, self.use_gt = reader, use_gt_layout
        self.buckets = list(buckets) if isinstance(buckets, (list, tuple)) else [buckets]
        b0 = self.buckets[0]
        self.device = b0.engine.device
        self.copy_stream = torch.cuda.Stream(device=self.device)
        d, K, Nb = b0.dims, b0.K, b0.Nb
        bl = reader.batch_loader

        def pinned(shape, dtype):
            return torch.empty(shape, dtype=dtype).pin_memory()

        def staging_set():
            t = dict(input_seq_batch=pinned((d.T_encoder, Nb), torch.int32),
                     seq_length_batch=pinned((Nb,), torch.int32),
                     image_feat_batch=pinned((Nb, bl.feat_H, bl.feat_W, bl.feat_D), torch.float32),
                     gt_layout_batch=pinned((d.T_decoder, Nb), torch.int32))
            if not (use_gt_layout and getattr(bl, 'load_gt_layout', False)):
                del t['gt_layout_batch']
            st = {k: v.numpy() for k, v in t.items()}     # numpy views of the pinned storage
            st['_pinned'] = t
            return st
        # The reader's prefetch thread builds each batch INSIDE a pinned staging set
        # (load_one_batch(out=...)): the only host copy of the 20 MB of image features is np.load's,
        # and the consumer thread only enqueues the H2D copies.  A set is out while it sits in the
        # prefetch queue or belongs to a group whose copies may still be running.
        nsets = len(self.buckets) * K + reader.prefetch_num + 2
        self._free = reader.use_staging([staging_set() for _ in range(nsets)])
        self._held = [[] for _ in self.buckets]                      # sets of each bucket's last group
        # batches that were prefetched before use_staging (pageable): one staging set per slot
        self.stage = [[None] * K for _ in self.buckets]
        self._staging_set = staging_set
        self.copied = [torch.cuda.Event() for _ in self.buckets]    # staging sets free again
        self.released = [torch.cuda.Event() for _ in self.buckets]  # bucket free again
        self._used = [False] * len(self.buckets)

    @staticmethod
    def _rows(t, key, nb):
        return t[..., :nb] if key in ('input_seq_batch', 'gt_layout_batch') else t[:nb]

    def groups(self):
        torch = self.torch
        it = self.reader.batches()
        keys = ('input_seq_batch', 'seq_length_batch', 'image_feat_batch') + \
            (('gt_layout_batch',) if self.use_gt else ())
        g = 0
        while True:
            s = g % len(self.buckets)
            bucket = self.buckets[s]
            if self._used[s]:
                self.copied[s].synchronize()     # the copies reading this bucket's staging sets are done
                for st in self._held[s]:
                    self._free.put(st)
                self._held[s] = []
                self.copy_stream.wait_event(self.released[s])    # the bucket's last pass is done
            batches = []
            with torch.cuda.stream(self.copy_stream):
                for k in range(bucket.K):
                    b = next(it, None)
                    if b is None:
                        break
                    nb = b['seq_length_batch'].shape[0]
                    slot = bucket.slot(k)
                    st = b.pop('_staging', None)
                    if st is not None:
                        self._held[s].append(st)
                    else:                        # prefetched before use_staging: copy it into a set
                        if self.stage[s][k] is None:
                            self.stage[s][k] = self._staging_set()
                        st = self.stage[s][k]
                        for key in keys:
                            self._rows(st['_pinned'][key], key, nb).copy_(
                                torch.from_numpy(np.ascontiguousarray(b[key])))
                    for key in keys:
                        src = self._rows(st['_pinned'][key], key, nb)
                        self._rows(slot[key], key, nb).copy_(src, non_blocking=True)
                    if nb < bucket.Nb:           # short last batch: the rest of the slot is padding
                        slot['seq_length_batch'][nb:].fill_(1)
                    # Lifetime of what the consumer sees in group.batches: a batch built inside a
                    # pinned staging set holds numpy VIEWS of it, and the set goes back to the prefetch
                    # thread when this bucket is reused (len(buckets) groups later).  The small arrays
                    # (tokens, lengths, layouts: a few KB) are therefore copied out here, so a consumer
                    # may keep them for logging or accumulation; only the 20 MB `image_feat_batch`
                    # stays an alias, valid until the group after next is requested.
                    for key in ('input_seq_batch', 'seq_length_batch', 'gt_layout_batch'):
                        if key in b and isinstance(b[key], np.ndarray) and not b[key].flags.owndata:
                            b[key] = np.array(b[key])
                    batches.append(b)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
                self.copied[s].record(self.copy_stream)
                self._used[s] = True
            if not batches:
                return
            yield DeviceFeeder.Group(batches, ev, bucket, g)
            # resumed: whatever the consumer enqueued for this group is on its current stream
            self.released[s].record(torch.cuda.current_stream(self.device))
            g += 1

    def wait(self, group):
        """Order the current (compute) stream after the group's copies; no host wait."""
        self.torch.cuda.current_stream(self.device).wait_event(group.event)
