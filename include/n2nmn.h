/*
 * n2nmn.h -- C ABI of the MI355X-native N2NMN (CLEVR) forward hot path.
 *
 * This is the drop-in boundary.  The reference is pure Python on top of
 * TensorFlow 1.0.0 + TensorFlow-Fold 0.0.1; the "FFI" it would bind for this path is the set of
 * TF/Fold entry points listed below.  Each export names the reference interface it replaces
 * (file:line relative to ronghanghu/n2nmn).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns 0 on success or a negative N2NMN_E* code; no C++ exception crosses
 *     the boundary; n2nmn_last_error() returns a thread-local message for the last failure.
 *   - all tensor arguments are plain DEVICE pointers (fp32 / int32, layouts stated per argument)
 *     unless the name ends in _host.  The caller owns every buffer it passes in; the library
 *     only allocates inside n2nmn_ctx_create / n2nmn_program_create and frees in *_destroy.
 *   - every launching function takes the HIP stream explicitly and never calls
 *     hipDeviceSynchronize / hipStreamSynchronize (the only host sync on the path is the caller's
 *     fetch of predicted_tokens between the two phases, exp_clevr/eval_clevr.py:111-125).
 *   - one context may be used from one thread at a time; different contexts are independent.
 *   - data layouts follow the reference: features NHWC, text time-major [T,N], attention maps
 *     [Nb,H,W,1] row-major, fc/1x1 weights [in,out], conv weights [kh,kw,in,out], LSTM weights
 *     [in+hidden, 4*hidden] with gate order i,j,f,o (TF BasicLSTMCell, forget_bias = 1).
 */
#ifndef N2NMN_H_
#define N2NMN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define N2NMN_OK            0
#define N2NMN_EINVAL       -1   /* bad argument (shape / null / out of range)            */
#define N2NMN_EHIP         -2   /* a HIP runtime call failed (message has the HIP error) */
#define N2NMN_ENOWEIGHT    -3   /* a required variable was never registered / committed  */
#define N2NMN_ECAPACITY    -4   /* batch / program larger than the context was built for */
#define N2NMN_EKEY         -5   /* unknown variable or module name                       */

typedef void *n2nmn_stream;              /* a hipStream_t */
typedef struct n2nmn_ctx n2nmn_ctx;      /* weights + workspace of one model on one device */
typedef struct n2nmn_program n2nmn_program; /* packed, level-scheduled batch of layout trees */

/* Module operator codes == CLEVR layout-token indices (exp_clevr/data/vocabulary_layout.txt;
 * arity / output type tables: models_clevr/nmn3_assembler.py:9-41). */
typedef enum {
  N2NMN_OP_SCENE = 0, N2NMN_OP_FIND = 1, N2NMN_OP_FILTER = 2, N2NMN_OP_FIND_SAME_PROPERTY = 3,
  N2NMN_OP_TRANSFORM = 4, N2NMN_OP_AND = 5, N2NMN_OP_OR = 6, N2NMN_OP_EXIST = 7,
  N2NMN_OP_COUNT = 8, N2NMN_OP_EQUAL_NUM = 9, N2NMN_OP_MORE_NUM = 10, N2NMN_OP_LESS_NUM = 11,
  N2NMN_OP_SAME_PROPERTY = 12, N2NMN_OP_DESCRIBE = 13, N2NMN_NUM_OPS = 14
} n2nmn_op;

/* Model dimensions (defaults of exp_clevr/eval_clevr.py:27-37 in comments). */
typedef struct {
  int32_t H, W, D;            /* 10, 15, 512   image feature grid                          */
  int32_t map_dim;            /* 250           models_clevr/nmn3_modules.py map_dim         */
  int32_t embed_dim_txt;      /* 300 */
  int32_t embed_dim_nmn;      /* 300 */
  int32_t lstm_dim;           /* 512 */
  int32_t num_layers;         /* 2   (only 2 is supported) */
  int32_t num_vocab_txt;      /* 82  */
  int32_t num_vocab_nmn;      /* 15  (last index = <eos>) */
  int32_t num_choices;        /* 28  */
  int32_t T_encoder;          /* 45  maximum encoder length */
  int32_t T_decoder;          /* 20  maximum decoder length */
  int32_t N;                  /* 64  maximum batch size */
  int32_t kernel_size;        /* 5   TransformModule conv kernel */
  int32_t variant;            /* N2NMN_VARIANT_CLEVR (models_clevr) or N2NMN_VARIANT_VQA (models_vqa) */
  int32_t qpn_hidden;         /* VQA: hidden width of question_prior_net (500), 0 = use_qpn False */
} n2nmn_dims;

/* Model variants.  models_vqa (exp_vqa/eval_vqa2.py:27-39, models_vqa/nmn3_modules.py) has four
 * modules over the feature grid WITH its two coordinate channels appended by the caller
 * (add_spatial_coordinate_map, nmn3_modules.py:11-31): _Find (= FindModule), _Transform (the
 * attention-pooled three-way product: arithmetic of models_clevr's FindSamePropertyModule, run
 * under N2NMN_OP_FIND_SAME_PROPERTY with the variables of scope TransformModule), _And, _Describe,
 * plus the question prior network.  Other operator codes are rejected (N2NMN_EKEY). */
#define N2NMN_VARIANT_CLEVR 0
#define N2NMN_VARIANT_VQA   1

const char *n2nmn_last_error(void);
const char *n2nmn_version(void);

/* ------------------------------------------------------------------------------------------
 * (1) context, weights.   Replaces: tf.Session + tf.get_variable + tf.train.Saver.restore
 *     (exp_clevr/eval_clevr.py:18-20,90-91; models_clevr/nmn3_modules.py:11-47).
 * ---------------------------------------------------------------------------------------- */
int n2nmn_ctx_create(const n2nmn_dims *dims, int device, n2nmn_ctx **out);
/* A second context on the same device that SHARES the weight store of `parent` (committed through
 * the parent) and owns only its workspace: several batches in flight on different streams / host
 * threads without duplicating the weights in HBM and L2.  The parent must outlive its forks. */
int n2nmn_ctx_fork(n2nmn_ctx *parent, n2nmn_ctx **out);
int n2nmn_ctx_destroy(n2nmn_ctx *ctx);
int n2nmn_ctx_dims(const n2nmn_ctx *ctx, n2nmn_dims *out);
/* Scheduling hint for the recurrent step kernels of THIS context (forks have their own):
 *   N2NMN_MODE_LATENCY    (default) 64-row x 16-column workgroup tiles: shortest single-batch step
 *   N2NMN_MODE_THROUGHPUT for passes of >= 128 rows (super-bucketed batches): 64-row x 64-gate-column
 *                         tiles whose operands are staged through LDS by LDS-DMA (the weight tile is
 *                         shared by the workgroup's four waves: 2.5x less L2 operand traffic than the
 *                         K-split tiles); smaller passes fall back to the K-split 32 x 32 tile
 *   N2NMN_MODE_THROUGHPUT_KSPLIT  the K-split 32 x 32 tile at every size (round 2's throughput mode;
 *                         kept for A/B measurements) */
#define N2NMN_MODE_LATENCY    0
#define N2NMN_MODE_THROUGHPUT 1
#define N2NMN_MODE_THROUGHPUT_KSPLIT 2
/* N2NMN_MODE_THROUGHPUT_BF16X3  (opt-in, experimental) N2NMN_MODE_THROUGHPUT with the recurrent contraction of passes of
 *   >= 128 rows on the bf16 matrix cores over three-way split operands: w = wh + wm + wl, h = hh + hm + hl
 *   (bf16 each, exact sums), six cross products per 16 x 16 x 32 block with fp32 accumulation
 *   (csrc/kernels_lstm_tile3.hip).  The dense contractions (encoder_h_transform, W_a, conv_image) stay on the
 *   exact-fp32 kernels: their split-operand form is not part of this library (a pass whose conv_image launch ran
 *   on it returned wrong logits while another stream ran passes; DESIGN.md 2.1, profiles/r06_notes.md section 1).
 *   The terms dropped are <= 2^-26 relative: same error class as the fp32
 *   MFMA's own rounding -- every parity test runs unchanged at 1e-4 in this mode -- but NOT the same bits
 *   as the fp32 kernels, so it is a mode of its own and never the default.  Training forwards keep the
 *   fp32 kernels.  Needs lstm_dim % 128 == 0 and lstm_dim >= 256 (N2NMN_EINVAL otherwise); the first call
 *   packs the split weights (may wait for the device). */
#define N2NMN_MODE_THROUGHPUT_BF16X3 3
int n2nmn_ctx_set_mode(n2nmn_ctx *ctx, int mode);

/* Register one variable by its reference (TF 1.0.0) name, e.g.
 * "neural_module_network/layout_execution/module_variables/FindModule/conv_image/weights".
 * `data` is a device pointer to contiguous fp32 in the reference's layout; it is only read
 * during n2nmn_commit_weights.  shape is checked against the model dimensions. */
int n2nmn_set_weight(n2nmn_ctx *ctx, const char *name, const float *data,
                     const int64_t *shape, int ndim);
/* Re-pack every registered variable into the kernels' HBM layouts (k-interleaved MFMA operand
 * tiles, gate-interleaved LSTM column tiles).  Must be called after the
 * variables change and before any forward.  Asynchronous on `stream`. */
int n2nmn_commit_weights(n2nmn_ctx *ctx, n2nmn_stream stream);
/* Validity automaton of the layout vocabulary: Assembler.P [V,3], Assembler.W [3,V,4],
 * Assembler.b [V,4] (models_clevr/nmn3_assembler.py:50-119), which the reference hands to the
 * decoder at models_clevr/nmn3_netgen_att.py:59-62.  HOST int32 pointers, copied synchronously. */
int n2nmn_set_validity_tables(n2nmn_ctx *ctx, const int32_t *P_host, const int32_t *W_host,
                              const int32_t *b_host);
/* Number of expected variables and the i-th expected name/shape (introspection for loaders). */
/* Host only (no device, no context): 1 if the automaton P [V,3], W [3,V,4], b [V,4] (models_clevr/nmn3_assembler.py:
 * 50-135; token s valid in state x iff all_c (x . W[:, s, c] - b[s, c] >= 0), x += P[token]) allows nothing but <eos>
 * behind <eos> and behind every answer operator (token_op [V]: n2nmn_op codes, < 0 = <eos>), in every state reachable
 * from (0, 0, T), T = 1 .. T_decoder; 0 otherwise.  n2nmn_set_validity_tables / n2nmn_set_token_ops evaluate this for
 * the installed tables: N2NMN_S2S_EOS_RETIRE retires decoder-chosen layouts only if it is 1. */
int n2nmn_automaton_forces_eos(const int32_t *P_host, const int32_t *W_host, const int32_t *b_host,
                               const int32_t *token_op_host, int V, int T_decoder);
int n2nmn_num_variables(const n2nmn_ctx *ctx);
int n2nmn_variable_info(const n2nmn_ctx *ctx, int i, const char **name, int64_t shape[4],
                        int *ndim);

/* ------------------------------------------------------------------------------------------
 * (2) phase 1: layout generator.   Replaces AttentionSeq2Seq's graph
 *     (models_clevr/nmn3_netgen_att.py:46-322: tf.nn.embedding_lookup, tf.nn.dynamic_rnn,
 *     tf.nn.raw_rnn + loop_fn, util/cnn.py:87-119 fc) as run by the first sess.partial_run
 *     (exp_clevr/eval_clevr.py:111-114).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  /* inputs */
  const int32_t *input_seq;        /* [T_enc, N] time-major word indices, zero padded       */
  const int32_t *seq_length;       /* [N]                                                   */
  int32_t T_enc, N, T_dec;
  int32_t use_gt_layout;           /* != 0: teacher forcing, all tokens valid (:204-207,239-241) */
  const int32_t *gt_layout;        /* [T_dec, N] or NULL                                    */
  const float *sample_uniforms;    /* [T_dec, N] in [0,1) -> decoder_sampling=True (:212-232,
                                      inverse-CDF stand-in for tf.multinomial); NULL = greedy */
  const int32_t *forced_tokens;    /* [T_dec, N] or NULL: parity hook, overrides the chosen token
                                      without touching validity / probabilities            */
  /* outputs (NULL = not wanted) */
  int32_t *predicted_tokens;       /* [T_dec, N]                                            */
  float *token_probs;              /* [T_dec, N]                                            */
  float *neg_entropy;              /* [N]                                                   */
  float *atts;                     /* [T_dec, T_enc, N] (the reference's trailing 1 dropped) */
  float *word_vecs;                /* [T_dec, N, embed_dim_txt]                             */
  float *token_scores;             /* [T_dec, N, num_vocab_nmn] pre-mask logits (debug/parity) */
  float *encoder_outputs;          /* [T_enc, N, lstm_dim]                                  */
  float *encoder_h_transformed;    /* [T_enc, N, lstm_dim]                                  */
  float *encoder_states;           /* [2 layers][c,h][N, lstm_dim]                          */
  float *log_seq_prob;             /* [N] sum_t log token_probs (models_clevr/nmn3_model.py:46) */
  int32_t flags;                   /* N2NMN_S2S_* */
  /* optional input: [N, H, W, D] image features of the same batch.  When given, the hoisted
   * conv_image GEMMs (n2nmn_conv_image with FIND | FSP, the second gated by gt_layout under teacher
   * forcing, by the chosen tokens otherwise) are issued by this call -- under teacher forcing in ONE
   * launch with encoder_h_transform and q -- and n2nmn_walk_layouts can follow directly.  Only
   * n2nmn_decoder_forward / n2nmn_seq2seq_forward read it. */
  const float *image_feat;
  /* optional inputs: dropout multipliers (0 or 1 / keep_prob, see n2nmn_dropout_multipliers) on
   * the output of LSTM layer 0 of the encoder [T_enc, N, lstm_dim] / the decoder [T_dec, N, lstm_dim]
   * (encoder_dropout / decoder_dropout = True, models_vqa/nmn3_netgen_att.py:17-44).  The policy-
   * gradient scripts sample the layout from the network WITH dropout
   * (exp_vqa/train_vqa_rl_gt_layout.py:34-35); the same buffers then go into n2nmn_train_io. */
  const float *drop_enc0, *drop_dec0;
  /* optional HOST copy of seq_length (the reference feeds lengths from the host,
   * util/clevr_train/data_reader.py:74-82): with it the launcher knows how many rows are still
   * active at every encoder step and picks the recurrent-step tile per step in N2NMN_MODE_THROUGHPUT
   * (LDS-staged 64 x 64 tiles while >= 4 blocks of 64 rows are active, K-split tiles for the tail).
   * NULL: one tile shape for the whole pass.  Read during the call only. */
  const int32_t *seq_length_host;
  /* optional HOST array [N] for N2NMN_S2S_EOS_RETIRE: tokens of gt_layout[:, n] in front of its first
   * <eos> (the reference's data reader holds the layouts on the host, util/clevr_train/data_reader.py:
   * 60-72).  With it the decoder issues only as many step launches as the longest layout needs and
   * picks the step tile from the rows still alive; NULL: T_dec + 1 launches whose workgroups find the
   * live-row counts on the device.  Read during the call only. */
  const int32_t *gt_length_host;
} n2nmn_seq2seq_io;
/* skip word_vecs / neg_entropy / log_seq_prob (one launch): for inference through
 * n2nmn_walk_layouts with attention maps, which derives the text maps from atts directly */
#define N2NMN_S2S_NO_WORD_VECS 1
/* Inference option for teacher-forced passes (use_gt_layout, >= 128 rows, a throughput mode): retire a row
 * from the decoder at its layout's first <eos>.  Decoder step
 * t >= len(layout) of a row feeds nothing exp_clevr/eval_clevr.py:103-135 fetches -- the layout is read
 * up to its first <eos> (models_clevr/nmn3_assembler.py:153-170) and a module's text attention is the
 * step of its own token (models_clevr/nmn3_modules.py:53-57) -- so the LSTM cells, q, and the attention
 * run only over the (row, step) pairs in front of it: rows are ordered by layout length and step t
 * covers the row blocks still alive (the encoder's length trick).  predicted_tokens are complete;
 * atts / token_probs / word_vecs hold the live (row, step) pairs only, neg_entropy / log_seq_prob (sums
 * over all steps) are not meaningful.  The full outputs (all T_dec steps; training
 * and the debug fetches need them) come from a call without the flag -- n2nmn_decoder_forward on the
 * same context recomputes them from the encoder results it holds.  Where the preconditions do not hold
 * the flag is ignored.
 * Layouts the decoder chooses itself (greedy / sampled decoding; >= 128 and <= 1024 rows, a throughput
 * mode): a row is finished once it has emitted <eos> OR an answer operator (Exist .. Describe) -- behind either
 * the reference's automaton allows nothing but <eos> (models_clevr/nmn3_assembler.py:94-117) -- and the
 * finished rows are compacted out of the state after every step; they get their <eos> tokens (probability 1)
 * without a recurrent step.  This is exact only for an automaton with that property, so it is PROVEN when the
 * tables are installed: n2nmn_set_validity_tables / n2nmn_set_token_ops search every state reachable from
 * (0, 0, T), T = 1 .. T_decoder, and the flag is ignored on this path unless no finished row can ever emit
 * another token (all-zero tables -- every token always valid, models_shapes -- do not qualify).  Same contract:
 * predicted_tokens complete and identical to the full decoder's, atts / token_probs for live (row, step) pairs only. */
#define N2NMN_S2S_EOS_RETIRE 2

int n2nmn_encoder_forward(n2nmn_ctx *ctx, const n2nmn_seq2seq_io *io, n2nmn_stream stream);
/* decoder uses the encoder results held in the context by the preceding encoder call */
int n2nmn_decoder_forward(n2nmn_ctx *ctx, const n2nmn_seq2seq_io *io, n2nmn_stream stream);
/* encoder + decoder in one call */
int n2nmn_seq2seq_forward(n2nmn_ctx *ctx, const n2nmn_seq2seq_io *io, n2nmn_stream stream);

/* ------------------------------------------------------------------------------------------
 * (3) host: RPN tokens -> packed program.   Replaces Assembler.assemble
 *     (models_clevr/nmn3_assembler.py:153-222) + td.Compiler.build_feed_dict / Loom's
 *     serialisation and per-depth batching (models_clevr/nmn3_model.py:55-159;
 *     exp_clevr/eval_clevr.py:125-128).  Pure host code, no GPU needed.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t op;          /* n2nmn_op */
  int32_t time_idx;    /* RPN position t: text parameter row t*N_full + batch_idx            */
  int32_t batch_idx;   /* question / image index                                             */
  int32_t in0, in1;    /* node ids of input_0 / input_1 (input_1 = most recently pushed), -1 */
  int32_t level;       /* dependency level assigned by the scheduler                          */
  int32_t out_row;     /* answer nodes: row of `scores`; attention nodes: -1                  */
  int32_t reserved;
} n2nmn_node;

/* assembly error kinds, in the order the reference checks them (nmn3_assembler.py:172-211) */
#define N2NMN_ASM_OK            0
#define N2NMN_ASM_NO_EOS        1  /* 'cannot find <eos>'                                  */
#define N2NMN_ASM_NOT_ENOUGH    2  /* 'not enough input for <module>'                      */
#define N2NMN_ASM_INCOMPATIBLE  3  /* 'input incompatible for <module>'                    */
#define N2NMN_ASM_STACK_SIZE    4  /* 'final stack size not equal to 1 (<k> remains)'      */
#define N2NMN_ASM_NOT_ANS       5  /* 'result type must be ans, not att'                   */

int n2nmn_program_create(n2nmn_program **out);
int n2nmn_program_destroy(n2nmn_program *p);
/* tokens_host [T,N] int32 (host).  token_op_host[V]: op code of each layout token, or -1 for
 * <eos>.  validity_host[N] receives 1/0 like expr_validity_array.  Never fails on invalid
 * layouts -- they are data (INVALID_EXPR -> zero logits, nmn3_model.py:146,155). */
int n2nmn_assemble(n2nmn_program *p, const int32_t *tokens_host, int T, int N,
                   const int32_t *token_op_host, int V, uint8_t *validity_host);
/* Build a program from explicit nodes (the dict-walking build_feed_dict path).  `nodes_host`
 * must be topologically ordered (inputs before consumers); level / reserved are ignored and
 * recomputed.  num_rows = number of score rows (len(expr_list)). */
int n2nmn_program_from_nodes(n2nmn_program *p, const n2nmn_node *nodes_host, int num_nodes,
                             int num_rows);
int n2nmn_program_num_nodes(const n2nmn_program *p);
int n2nmn_program_num_rows(const n2nmn_program *p);
int n2nmn_program_num_levels(const n2nmn_program *p);
int n2nmn_program_get_nodes(const n2nmn_program *p, n2nmn_node *out_host, int capacity);
/* per-example assembly status: kind (N2NMN_ASM_*), offending op (or -1), remaining stack size */
int n2nmn_program_status(const n2nmn_program *p, int example, int32_t *kind, int32_t *op,
                         int32_t *remains);
/* number of kernel launches execute_program will issue for this program (scheduler metric) */
int n2nmn_program_num_launches(const n2nmn_program *p);

/* ------------------------------------------------------------------------------------------
 * (4) phase 2: module-network execution.   Replaces compiler.loom_input_tensor ->
 *     compiler.output_tensors[0] (models_clevr/nmn3_model.py:158-159), i.e. Loom's depth-wise
 *     dynamic batching of the Modules.* operators, as run by the second sess.partial_run
 *     (exp_clevr/eval_clevr.py:132).
 *     image_feat [N_full,H,W,D]; word_vecs [T_dec,N_full,E]; scores [num_rows, num_choices].
 * ---------------------------------------------------------------------------------------- */
int n2nmn_execute_program(n2nmn_ctx *ctx, n2nmn_program *p, const float *image_feat,
                          const float *word_vecs, int N_full, float *scores,
                          n2nmn_stream stream);

/* ------------------------------------------------------------------------------------------
 * (4b) phase 2 WITHOUT the host hop: layouts are decoded and executed on the device.
 *     Replaces, for inference, the host block between the two partial runs of
 *     exp_clevr/eval_clevr.py:111-132 -- the predicted_tokens fetch, Assembler.assemble
 *     (models_clevr/nmn3_assembler.py:153-222), compiler.build_feed_dict and Loom's per-depth
 *     batching (models_clevr/nmn3_model.py:55-159) -- by ONE kernel in which a workgroup decodes
 *     its question's RPN tokens (same five validity checks; invalid -> zero logits, validity 0)
 *     and runs the whole tree.  The Python Assembler / n2nmn_assemble stay
 *     as the compatibility path (expression dicts, error strings, explicit programs, training).
 * ---------------------------------------------------------------------------------------- */
/* op code of every layout token (HOST int32 [V], -1 for <eos>): what Assembler passes to
 * n2nmn_assemble per call, kept on the device for the walker.  Root context only. */
int n2nmn_set_token_ops(n2nmn_ctx *ctx, const int32_t *token_op_host, int V);
/* 1 if the context's dimensions fit the walker's tiling (models_clevr dimensions do;
 * models_vqa runs through n2nmn_execute_program). */
int n2nmn_walk_supported(const n2nmn_ctx *ctx);

#define N2NMN_CONV_FIND 1   /* FindModule conv_image (shared by _Find and _Filter, nmn3_modules.py:129) */
#define N2NMN_CONV_FSP  2   /* FindSamePropertyModule conv_image */
/* The hoisted, text-independent 1x1 convolution image_feat . W_img + b (nmn3_modules.py:98-99,
 * 158-159 via util/empty_safe_conv.py:8-32) of all N images into the context's workspace; needs only
 * the features, so it can run beside phase 1.  tokens (device [T_dec, N]) or NULL: when given, the
 * FindSameProperty map is computed only for images whose layout contains that token. */
int n2nmn_conv_image(n2nmn_ctx *ctx, const float *image_feat, int N, int which,
                     const int32_t *tokens, int T_dec, n2nmn_stream stream);

typedef struct {
  n2nmn_ctx *ctx;               /* context (root or fork) whose workspace holds THIS batch's conv_image
                                   maps (n2nmn_conv_image was called on it); NULL = the calling one */
  const int32_t *tokens;        /* [T_dec, N] layout tokens (predicted_tokens or gt_layout), device */
  const float *image_feat;      /* [N, H, W, D]                                               */
  const float *word_vecs;       /* [T_dec, N, E], or NULL when atts is given                  */
  float *scores;                /* [N, num_choices] out                                       */
  int32_t *validity;            /* [N] out: 1 / 0 like expr_validity_array, or NULL          */
  /* Optional (all batches or none): the decoder's attention maps instead of word_vecs.  A text
   * parameter is word_vecs[t, n] = sum_tau atts[t, tau, n] * embedding_mat[input_seq[tau, n]]
   * (nmn3_netgen_att.py:312) and every use of it is fc_text(word_vec) (nmn3_modules.py:101,161,209,
   * 424,479), so the walker evaluates  b + sum_tau atts * (embedding_mat . W_txt)[input_seq]  from
   * a [num_vocab_txt, map_dim] table per weight set built at commit time: no word_vecs launch and
   * no text-map launch.  Needs num_vocab_txt <= 4096. */
  const float *atts;            /* [T_dec, T_enc, N]                                          */
  const int32_t *input_seq;     /* [T_enc, N]                                                 */
  const int32_t *seq_length;    /* [N]                                                        */
} n2nmn_walk_batch;
/* One launch for the questions of K in-flight batches ("super-bucket", 1 <= K <= 16; every batch
 * has N questions and T_dec steps).  The caller orders `stream` after each batch's phase 1 and
 * n2nmn_conv_image (events). */
int n2nmn_walk_layouts(n2nmn_ctx *ctx, const n2nmn_walk_batch *batches, int K, int T_dec,
                       int T_enc /* only with atts */, int N, n2nmn_stream stream);
/* Where the answer operators that pool image features (_Describe, _SameProperty: always the root of a
 * layout) run.  0: inside the walker (lowest single-batch latency).  1: the walker only computes
 * their soft-max weights, and two chip-wide launches follow: the attention-weighted feature sums
 * of all such questions (8 workgroups per question: the HBM-bound kernel of the attention-module
 * path), then fc_att + answer head.  -1 (default): 1 when a launch carries >= 128 questions. */
int n2nmn_walk_set_defer_pool(n2nmn_ctx *ctx, int mode);
/* Where the tree-independent work of n2nmn_walk_layouts runs when the text maps come from the
 * attention tables: mode -1 (default) chip-wide launches ahead of the walker for passes of >= 128
 * questions (walk_tmap_kernel: text maps of all nodes; walk_find_kernel: the Find / Filter epilogues,
 * 4 workgroups per question streaming the conv_image map), 0 always inside the walker, 1 always
 * chip-wide. */
int n2nmn_walk_set_front_end(n2nmn_ctx *ctx, int mode);
/* Staged walker.  When a launch runs both of the above chip-wide (front end AND deferred pooling: passes
 * of >= 128 questions by default), the tree-dependent rest leaves the one-workgroup-per-question walker
 * as well: walk_tmap_kernel decodes every layout once (nmn3_assembler.py:153-222) and lists the
 * _Transform / _FindSameProperty nodes by nesting level (level = such nodes below it in its own subtree);
 * per level, walk_heavy_kernel runs every _Transform node as two jobs (pixel halves, four workgroups per CU,
 * the 5x5 convolution on the matrix cores) and every _FindSameProperty node as 8 channel parts of its soft-max
 * pooling + fc_att share, and walk_fspepi_kernel finishes the _FindSameProperty maps (8 row parts per node
 * over the operator's conv_image map); walk_light_kernel (one small workgroup per question) evaluates the
 * remaining And / Or / Filter / Scene nodes and the answer operator (nmn3_modules.py:60-72,113-132,218-400)
 * or hands the root to the deferred pooling.  How many levels a pass launches: see n2nmn_walk_set_levels.
 * mode -1 (default): on; 0: off (the walker serves every question). */
int n2nmn_walk_set_staged(n2nmn_ctx *ctx, int mode);
/* Nesting levels of _Transform / _FindSameProperty the staged walker launches per pass.  A question is served
 * either by the level launches or -- nested deeper than the pass launches -- by the one-workgroup walker: same
 * operators, other summation order in the answer head (logits within 1e-5 of each other).
 * levels = 0 (default): every level a layout of T_dec tokens can reach (T_dec - 1 launches pairs; the level kernels
 * are persistent grids that leave at once when their list is empty) and no fall-back launch: the route of a question
 * depends on nothing but its own layout, repeated passes return the same bits whatever ran before
 * (Fold's result does not depend on batching).  A pass whose caller promised a nesting bound
 * (n2nmn_walk_set_nesting_bound: ground-truth layouts) launches exactly that many instead.
 * levels >= 1: exactly that many levels in every pass (1 .. 24), deeper layouts on the fall-back walker.
 * levels = -1: adaptive -- as deep as the previous two passes of this context went (a host-mapped word the GPU
 * writes and the host reads without waiting).  Saves the empty launches of the default (~0.1 ms of a 1024-question
 * pass of decoder-chosen layouts); the logits of NESTED layouts are then reproducible to 1e-5, not bit for bit. */
int n2nmn_walk_set_levels(n2nmn_ctx *ctx, int levels);
/* A promise for the NEXT n2nmn_walk_layouts / n2nmn_execute_tokens call of this context only: no layout of
 * that pass nests _Transform / _FindSameProperty deeper than `bound` levels (0: none has such a node).  A
 * caller that holds the layouts on the host -- ground-truth layouts from the data reader,
 * util/clevr_train/data_reader.py:74-82 -- knows this before the pass; the staged walker then launches
 * exactly max(bound, 1) levels and NO fall-back walker (an empty launch of it is 4 - 5 us of a 1024-question
 * pass), whatever n2nmn_walk_set_levels says.  A layout that breaks the promise is reported INVALID
 * (validity 0, zero logits), never evaluated wrongly.  bound = -1 withdraws the promise. */
int n2nmn_walk_set_nesting_bound(n2nmn_ctx *ctx, int bound);
/* on != 0: the NEXT n2nmn_walk_layouts call of this context computes the hoisted conv_image maps of its
 * batches itself (what n2nmn_conv_image(FIND | FSP, tokens) would have written, same GEMM kernels, same bits)
 * -- behind the text maps and right in front of walk_find, FindSameProperty's maps first and Find's last, so
 * the 154 KB per image that walk_find streams is the pass's most recently written data and is served by the
 * Infinity Cache (walk_find 36.5 -> 29.5 us per 1024 questions).  The caller then leaves image_feat out of
 * n2nmn_seq2seq_forward and does not call n2nmn_conv_image.  n2nmn_execute_tokens does this by itself. */
int n2nmn_walk_set_conv_inline(n2nmn_ctx *ctx, int on);
/* Phase 2 straight from DEVICE tokens (no token fetch, no host assembly): replaces Assembler.assemble +
 * td.Compiler.build_feed_dict + the second partial_run (exp_clevr/eval_clevr.py:121-132,
 * exp_vqa/eval_vqa2.py:103-137).  n2nmn_conv_image(FIND | FSP gated by tokens), then
 *   - dimensions the layout walker covers (n2nmn_walk_supported): n2nmn_walk_layouts(K = 1);
 *   - any other (models_vqa): the level path of n2nmn_execute_program with the program assembled and
 *     level-scheduled ON THE DEVICE (sched_kernel): every level of the capacity T_dec gets its three
 *     persistent-grid launches, which read their work tables and lengths from HBM; nothing is read back.
 *     (n2nmn_set_tokens_via_levels(ctx, 1) forces this form for every variant: the scheduler's cross-check
 *     against the walker and the host assembler.)
 * scores [N][num_choices] (INVALID_EXPR rows zero), validity [N] (expr_validity_array) or NULL. */
int n2nmn_execute_tokens(n2nmn_ctx *ctx, const int32_t *tokens, int T_dec, int N,
                         const float *image_feat, const float *word_vecs, float *scores,
                         int32_t *validity, n2nmn_stream stream);
int n2nmn_set_tokens_via_levels(n2nmn_ctx *ctx, int on);

/* models_vqa: out[n,h,w,:] = [feat[n,h,w,0:D0], x(w), y(h), 0...]  with x = linspace(-1,1,W)[w],
 * y = linspace(-1,1,H)[h]  (add_spatial_coordinate_map, models_vqa/nmn3_modules.py:11-31).
 * feat [N,H,W,D0]; out [N,H,W,D] with D = the context's (padded) feature depth >= D0 + 2. */
int n2nmn_add_coords(n2nmn_ctx *ctx, const float *feat, int N, int D0, float *out,
                     n2nmn_stream stream);

/* VQA only: scores[n, :] += fc2(relu(fc1(concat_layers(encoder h))))   (models_vqa/
 * question_prior_net.py:10-28 and `self.scores = self.scores_nmn + self.scores_qpn`,
 * models_vqa/nmn3_model.py:106-114), from the final encoder state held by the context after
 * n2nmn_encoder_forward / n2nmn_seq2seq_forward.  scores [N, num_choices]. */
int n2nmn_question_prior_add(n2nmn_ctx *ctx, int N, float *scores, n2nmn_stream stream);

/* ------------------------------------------------------------------------------------------
 * (5) one direct entry per module operator, mirroring Modules.<X>Module(input_0[, input_1],
 *     time_idx, batch_idx) (models_clevr/nmn3_modules.py:60-495; used directly by
 *     exp_shapes/visualize_shapes.ipynb).  input_0 / input_1: [Nb,H,W] attention logits or NULL
 *     according to the op's arity; time_idx_host / batch_idx_host: [Nb] int32 on the HOST;
 *     out: [Nb,H,W] (attention modules) or [Nb,num_choices] (answer modules).
 * ---------------------------------------------------------------------------------------- */
int n2nmn_module_forward(n2nmn_ctx *ctx, int op, int Nb, const float *input_0,
                         const float *input_1, const int32_t *time_idx_host,
                         const int32_t *batch_idx_host, const float *image_feat,
                         const float *word_vecs, int N_full, float *out, n2nmn_stream stream);

/* ------------------------------------------------------------------------------------------
 * (6) training step.   Replaces the graph built by exp_clevr/train_clevr_gt_layout.py:104-130:
 *     tf.nn.sparse_softmax_cross_entropy_with_logits + seq_likelihood_loss + weight_decay*l2_reg
 *     (models_clevr/nmn3_model.py:161-166), tf.train.AdamOptimizer().compute_gradients,
 *     tf.clip_by_norm(g, max_grad_l2_norm) per tensor, apply_gradients.  Teacher-forced layouts
 *     (use_gt_layout = True).  Data parallelism is NEW relative to the reference:
 *     the gradient lives in ONE caller-owned flat fp32 buffer so that a single all-reduce per
 *     bucket (RCCL through torch.distributed, or any other collective) can run between
 *     n2nmn_train_backward and n2nmn_adam_step.
 *
 *     Flat layout: variable i (order of n2nmn_variable_info) occupies
 *     [offset_i, offset_i + numel_i) in the reference's own element order, contiguous, no padding.
 *     Encoder variables come first, so the buffer splits into two buckets:
 *       bucket "late"  = [n2nmn_grad_split(), n2nmn_grad_numel())  decoder + module variables,
 *                        final after phase 0 of n2nmn_train_backward
 *       bucket "early" = [0, n2nmn_grad_split())                   encoder variables, final after phase 1
 *     which lets the all-reduce of the late bucket overlap the encoder's backward pass.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  /* inputs (device pointers) */
  const int32_t *input_seq;        /* [T_enc, N]                                             */
  const int32_t *seq_length;       /* [N]                                                    */
  int32_t T_enc, N, T_dec;
  const int32_t *gt_layout;        /* [T_dec, N] ground-truth layout tokens                  */
  const float *image_feat;         /* [N, H, W, D]                                           */
  const int32_t *answer_labels;    /* [N]                                                    */
  float weight_decay;              /* 5e-6 (train_clevr_gt_layout.py:39)                     */
  /* outputs */
  float *scores;                   /* [N, num_choices] answer logits                         */
  float *losses;                   /* [8]: avg_sample_loss, seq_likelihood_loss (cloning) or
                                      policy_gradient_loss (policy gradient), l2_reg, total_loss,
                                      entropy_reg, 3 reserved  (l2_reg / total_loss are complete
                                      after backward phase 1)                                */
  float *grads;                    /* flat [n2nmn_grad_numel]: d total_loss / d variable     */
  /* objective (zero-initialised = behavioural cloning, the fields below are then ignored).
   * N2NMN_OBJ_POLICY_GRADIENT = exp_clevr/train_clevr_rl_gt_layout.py:107-129:
   *   final_loss = expr_validity ? CE : invalid_expr_loss;  avg_sample_loss = mean(final_loss)
   *   policy_gradient_loss = mean(stop_gradient(final_loss - baseline) * log_seq_prob)
   *   total_loss = policy_gradient_loss + avg_sample_loss + lambda_entropy * entropy_reg
   *                + weight_decay * l2_reg;   baseline += (1 - baseline_decay)(avg - baseline)
   * gt_layout then holds the tokens the decoder SAMPLED for this batch (n2nmn_seq2seq_forward
   * with sample_uniforms); token validity comes from the automaton as it did when they were drawn
   * (nmn3_netgen_att.py:200-260), `p` is the program assembled from them.                   */
  int32_t objective;
  const int32_t *expr_validity;    /* [N] device: 1 = the layout assembled (n2nmn_assemble)  */
  float invalid_expr_loss;         /* 0.5           (train_clevr_rl_gt_layout.py:40)         */
  float lambda_entropy;            /* 0.005         (:41)                                    */
  float baseline_decay;            /* 0.99          (:42)                                    */
  float *baseline;                 /* device float[1]: read for this step's loss, then updated */
  /* dropout of the models_vqa training graph (encoder_dropout / decoder_dropout / qpn_dropout,
   * exp_vqa/train_vqa_gt_layout.py:33-39; models_vqa/nmn3_netgen_att.py:17-44: DropoutWrapper on the
   * OUTPUT of every LSTM layer but the last; models_vqa/question_prior_net.py:22-26).  TF draws the
   * masks from its own RNG stream; here they are inputs: float MULTIPLIERS, 0 for a dropped element
   * and 1 / keep_prob (= 2) for a kept one (n2nmn_dropout_multipliers fills such a buffer from a
   * counter-based generator).  NULL = that dropout is off.  Row order: the batch's own. */
  const float *drop_enc0;          /* [T_enc, N, lstm_dim]: encoder LSTM layer 0 -> layer 1   */
  const float *drop_dec0;          /* [T_dec, N, lstm_dim]: decoder LSTM layer 0 -> layer 1   */
  const float *drop_qpn_h;         /* [N, 2 * lstm_dim]: question_prior_net input h_concat    */
  const float *drop_qpn_fc1;       /* [N, qpn_hidden]: after fc1 + ReLU                       */
} n2nmn_train_io;
#define N2NMN_OBJ_CLONING 0
#define N2NMN_OBJ_POLICY_GRADIENT 1

/* Allocates the training workspace of a ROOT context (saved activations, backward scratch, Adam
 * moments: ~0.5 GB at CLEVR dimensions).  Idempotent. */
int n2nmn_train_enable(n2nmn_ctx *ctx);
int64_t n2nmn_grad_numel(const n2nmn_ctx *ctx);
int64_t n2nmn_grad_split(const n2nmn_ctx *ctx);
int n2nmn_grad_layout(const n2nmn_ctx *ctx, int variable, int64_t *offset, int64_t *numel);
/* forward of the objective; `p` = program assembled from the SAME gt_layout (n2nmn_assemble);
 * keeps every activation the backward pass needs inside the context */
int n2nmn_train_forward(n2nmn_ctx *ctx, const n2nmn_train_io *io, n2nmn_program *p,
                        n2nmn_stream stream);
/* phase 0: module network + decoder (zeroes io->grads first); phase 1: encoder.  Must follow
 * n2nmn_train_forward with the same io / program, in this order, on the same stream.
 * The weight-gradient GEMMs are leaves of the backward graph and run on a library-owned side stream
 * under the latency-bound chains (attention backward, both reverse-time recurrences).  By default
 * each phase ends with `stream` waiting for them, so the bucket that phase completes is final in
 * stream order.  phase = 0 | N2NMN_BWD_DEFER_JOIN skips that wait: the encoder's backward (phase 1)
 * then starts while the decoder's weight gradients and the finish of the late bucket are still
 * running; the late bucket is final for a stream only after n2nmn_train_join(ctx, that stream) --
 * n2nmn_allreduce_grads does this itself -- or after phase 1, which always joins. */
#define N2NMN_BWD_DEFER_JOIN 0x10
int n2nmn_train_backward(n2nmn_ctx *ctx, const n2nmn_train_io *io, n2nmn_program *p, int phase,
                         n2nmn_stream stream);
/* `stream` waits for everything n2nmn_train_backward has put on its side stream so far */
int n2nmn_train_join(n2nmn_ctx *ctx, n2nmn_stream stream);
/* out[i] = u_i < keep_prob ? 1 / keep_prob : 0 for i in [0, n), u_i the element offset + i of the
 * counter-based stream `seed` (tf.nn.dropout / DropoutWrapper draw from TF's RNG,
 * models_vqa/question_prior_net.py:22-26, models_vqa/nmn3_netgen_att.py:27): the multiplier buffers
 * n2nmn_train_io.drop_* expect.  Stateless: any slice can be regenerated from (seed, offset). */
int n2nmn_dropout_multipliers(float *out, int64_t n, float keep_prob, uint64_t seed, uint64_t offset,
                              n2nmn_stream stream);
/* g = grads * grad_scale (1/world_size after a sum all-reduce); per-tensor tf.clip_by_norm(g,
 * max_grad_l2_norm) -- max_grad_l2_norm <= 0: no clipping, as exp_vqa/train_vqa_gt_layout.py:119-123
 * trains --; Adam update (TF 1.0.0: lr_t = lr*sqrt(1-b2^step)/(1-b1^step)) of the
 * registered variables in place; re-packs the weights (n2nmn_commit_weights).  step = 1 first. */
int n2nmn_adam_step(n2nmn_ctx *ctx, const float *grads, float grad_scale, float lr, float beta1,
                    float beta2, float eps, float max_grad_l2_norm, int64_t step,
                    n2nmn_stream stream);
/* zero the Adam moments (a fresh tf.train.AdamOptimizer) */
int n2nmn_train_reset_optimizer(n2nmn_ctx *ctx, n2nmn_stream stream);
/* copies variable `name` (current value, reference layout) into `out` (device) */
int n2nmn_get_weight(n2nmn_ctx *ctx, const char *name, float *out, n2nmn_stream stream);
/* debug / parity: copy an internal gradient tensor ("d_word_vecs" [T_dec,N,E], "d_token_scores"
 * [T_dec,N,16], "d_encoder_outputs" [T_enc,N,L], "d_encoder_h_transformed" [T_enc,N,L],
 * "d_scores" [N,C]) into `out` (device, capacity in floats); returns the element count */
int64_t n2nmn_train_debug_tensor(n2nmn_ctx *ctx, const char *name, float *out, int64_t capacity,
                                 n2nmn_stream stream);

/* ------------------------------------------------------------------------------------------
 * (6b) gradient all-reduce over RCCL / xGMI for the data-parallel training step (SURVEY.md 8(b) item
 *     6, 8(e)).  The reference is single-GPU (no collective exists in it); this is the exchange step
 *     of exp_clevr/train_clevr_gt_layout.py:112-120 run on several GPUs: every rank computes the
 *     gradient of its own batch (n2nmn_train_forward / n2nmn_train_backward), the flat gradient
 *     vector is summed over ranks in two buckets, and n2nmn_adam_step applies scale = 1 / world.
 *     One process per GPU.  librccl is bound at run time.
 * ---------------------------------------------------------------------------------------- */
typedef struct n2nmn_comm n2nmn_comm;
#define N2NMN_COMM_ID_BYTES 128
/* rank 0: create the 128-byte id (ncclGetUniqueId) and hand it to the other ranks by any channel */
int n2nmn_comm_unique_id(void *id_out_128);
/* every rank: join the communicator (ncclCommInitRank) on `device`; creates the library-owned side
 * stream the collectives run on */
int n2nmn_comm_create(const void *unique_id_128, int rank, int world, int device, n2nmn_comm **out);
int n2nmn_comm_world(const n2nmn_comm *comm);
/* In-place sum over ranks of one bucket of the flat gradient vector `grads` (n2nmn_grad_numel
 * floats): bucket 0 = [split, numel) (decoder + module variables, final after backward phase 0),
 * bucket 1 = [0, split) (encoder variables, final after phase 1), split = n2nmn_grad_split.
 * Asynchronous: forked from `stream` with an event (the bucket's gradients must have been written
 * by work already enqueued on `stream`), runs on the communicator's side stream -- so bucket 0
 * overlaps backward phase 1. */
int n2nmn_allreduce_grads(n2nmn_ctx *ctx, n2nmn_comm *comm, int bucket, float *grads,
                          n2nmn_stream stream);
/* make `stream` wait for the buckets issued so far (call before n2nmn_adam_step) */
int n2nmn_allreduce_wait(n2nmn_comm *comm, n2nmn_stream stream);
int n2nmn_comm_destroy(n2nmn_comm *comm);

/* ------------------------------------------------------------------------------------------
 * (7) introspection used by the roofline report: algorithmic bytes / flops of one launch of a
 *     kernel family (SURVEY.md section 8d figures), and a plain GEMM entry for unit parity.
 * ---------------------------------------------------------------------------------------- */
/* Per-kernel-family profiler: between profile_begin and profile_end every kernel launch of the
 * context is bracketed by HIP events on its launch stream; profile_end synchronises the stream and
 * accumulates, per family, launches / total duration / ALGORITHMIC flops and bytes (the figures
 * of SURVEY.md section 8(d), stated per kernel in DESIGN.md section 4).  Returns the number of
 * launches recorded. */
int n2nmn_profile_begin(n2nmn_ctx *ctx);
int n2nmn_profile_end(n2nmn_ctx *ctx, n2nmn_stream stream);
int n2nmn_profile_num_families(void);
int n2nmn_profile_get(const n2nmn_ctx *ctx, int family, const char **name, int64_t *launches,
                      double *total_ms, double *flops, double *bytes);
/* Node counters of the walker launches since n2nmn_profile_begin (valid after n2nmn_profile_end), the
 * device-side source of the `walk(...)` family's algorithmic bytes / flops (the walker decodes its
 * layouts on the device, so the host has no other way to know what it executed).  out[10]:
 * [0] conv_image map passes inside the walker, [1] pooled attention inputs, [2] pooling nodes,
 * [3] nodes with a text parameter, [4] Transform nodes, [5] valid questions, [6] pooling jobs handed to
 * walk_pool_kernel, [7] their inputs, [8] map passes of walk_find_kernel, [9] reserved. */
int n2nmn_debug_walk_stats(n2nmn_ctx *ctx, uint64_t *out10);

/* Kernel-variant microbenchmark of the fused LSTM step (see csrc/capi.cpp); debugging aid. */
int n2nmn_debug_lstm_bench(n2nmn_ctx *ctx, int variant, int rows_per_wg, int njobs, int N,
                           int iters, double *us, n2nmn_stream stream);

/* C[m][n] += sum_r A[row(r)][m] * B[r][n]  -- the weight-gradient GEMM of the training step, for
 * unit parity.  a_row_idx (device, may be NULL): source row of r; a_onehot (device, may be NULL):
 * A[r][m] = (a_onehot[r] == m) and A is ignored; b_sel (device, may be NULL): only rows with
 * b_sel[r] == b_sel_val contribute.  M % 4 == 0 unless a_onehot; C must be initialised. */
int n2nmn_debug_gemm_tn(n2nmn_ctx *ctx, const float *A, int lda, int M, const float *B, int ldb,
                        int N, int R, float *C, int ldc, const int32_t *a_row_idx,
                        const int32_t *a_onehot, const int32_t *b_sel, int b_sel_val,
                        n2nmn_stream stream);
/* dst[c] += sum_r src[r*ld + c] over rows with sel[r] == sel_val (sel may be NULL) */
int n2nmn_debug_colsum(n2nmn_ctx *ctx, const float *src, int R, int ncols, int ld,
                       const int32_t *sel, int sel_val, float *dst, n2nmn_stream stream);

/* C[M,N] = A[M,K] . B[K,N] + bias[N]   (row-major fp32; B is packed internally) */
/* Mean HIP-event-pair time (us) around an EMPTY kernel on `stream`: the fixed cost every entry of the
 * n2nmn_profile_* table carries on top of its kernel's duration. */
int n2nmn_debug_event_overhead(n2nmn_ctx *ctx, int iters, double *us_pair, n2nmn_stream stream);
/* Re-launch one kernel of the LAST n2nmn_walk_layouts call `iters` times back to back inside one HIP
 * event pair (which: 0 walker = every launch between walk_find and the deferred pooling, 1 deferred pooling
 * kernel, 2 heads kernel, 3 walk_find_kernel, 4 walk_tmap_kernel; staged passes also 5 walk_heavy_kernel,
 * 6 walk_fspepi_kernel, 7 walk_light_kernel, 8 the fall-back walk_kernel, level 0 each; | 0x10: one event
 * pair PER
 * launch instead, which measures what a pair adds to this kernel) and return the average
 * microseconds per launch: the live duration the roofline of short kernels is computed from (an
 * event pair around a single ~5 us launch reads ~4 us too much).  Inputs must still be alive. */
int n2nmn_debug_walk_replay(n2nmn_ctx *ctx, int which, int iters, double *us_avg, n2nmn_stream stream);
/* Debugging: when timeline_dev != NULL every following walker launch of this context stamps the
 * shader clock of thread 0 at the phase boundaries of each node into
 * timeline_dev[question][32][4] (int64: start, after text map, after pooling + fc_att, end). */
int n2nmn_debug_walk_timeline(n2nmn_ctx *ctx, long long *timeline_dev);
int n2nmn_debug_gemm(n2nmn_ctx *ctx, const float *A, const float *B, const float *bias,
                     float *C, int M, int N, int K, n2nmn_stream stream);
/* A/B switches of ONE context (a fork without its own entry uses its parent's); value NULL removes the entry.
 * The library reads no environment variable: these are the only run-time switches, they are per context, and
 * none of them changes what is computed -- only which launch schedule / tile computes it (the tests run every
 * setting against the oracle).  Unknown key: N2NMN_EKEY.  Keys (values are decimal strings):
 *   "tile_min_rows"  encoder steps with at most this many live rows use the K-split tiles (default 192)
 *   "eht_rows"       0: encoder_h_transform over all T N rows instead of the rows inside their length (default 1)
 *   "profile_walk_stats" 0: profiled passes keep their event pairs but the walker kernels do not count nodes (default 1:
 *                    the counts are the families' algorithmic bytes; the atomics slow the tree-dependent launches 3 x)
 *   "debug_gemm_b3"  n2nmn_debug_gemm: n < 0 = -n launches per call (timing loops); n > 0 (n launches on the
 *                    split-operand bf16 GEMM) is refused: that kernel exists in the diagnostic library only
 *   before n2nmn_train_enable -- where the weight-gradient GEMMs of a training step run:
 *   "train_overlap"  0: no side stream (default 1)      "train_schedule" 0: every leaf after its recurrence
 *   "train_bg_wgs"   cap on background workgroups, 0 = unbounded     "train_chunks" "p0,p1,p2" time-chunk split */
int n2nmn_debug_set(n2nmn_ctx *ctx, const char *key, const char *value);
/* out[M,N] = (relu ? max(0, .) : .)(A[M,K] . W[K,N] + bias[N]), row-major fp32, K % 4 == 0:
 * util/cnn.py:87-126 (fc_layer / fc_relu_layer) and -- on im2col rows -- the VALID strided convolutions of
 * models_shapes/shapes_convnet.py:8-17 (BASELINE.json configs[0]; not on the CLEVR hot path).  Packs W per
 * call and synchronises `stream`. */
int n2nmn_fc_forward(n2nmn_ctx *ctx, const float *A, const float *W, const float *bias, float *out,
                     int M, int N, int K, int relu, n2nmn_stream stream);

/* ------------------------------------------------------------------------------------------
 * (8) snapshot I/O helper (host): CRC-32C (Castagnoli, polynomial 0x1EDC6F41 reflected) of `n` bytes,
 *     continuing from `crc` (0 to start) -- the checksum of TensorFlow's tensor-bundle checkpoints
 *     (tf.train.Saver.restore / .save: exp_clevr/eval_clevr.py:88-91, train_clevr_gt_layout.py:221-223;
 *     tensorflow/core/lib/hash/crc32c.h, restated).  n2nmn_amd/tf_checkpoint.py checksums tensors with it.
 * ---------------------------------------------------------------------------------------- */
uint32_t n2nmn_crc32c(uint32_t crc, const void *data, size_t n);

#ifdef __cplusplus
}
#endif
#endif /* N2NMN_H_ */
