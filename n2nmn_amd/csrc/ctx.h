// Internal definition of the context shared by capi.cpp (forward path) and capi_train.cpp
// (training step).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.h"
#include "program.h"

namespace n2nmn {

void set_last_error(const std::string& s);

#define N2_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
      return N2NMN_EHIP;                                                                 \
    }                                                                                    \
  } while (0)

#define N2_REQUIRE(cond, code, msg)                                                      \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      set_last_error(msg);                                                               \
      return code;                                                                       \
    }                                                                                    \
  } while (0)

struct Var {
  std::string name;
  std::vector<int64_t> shape;
  size_t numel = 0;
  float* mirror = nullptr;   // context-owned copy in the reference layout
  bool set = false;
  bool present = true;       // false: the variable does not exist in this model variant (numel 0)
};

// Host-side recorder of the pack jobs of one commit; uploaded once, launched on every commit.
struct PackBatch {
  std::vector<PackJob> jobs;
  PackJob* dev = nullptr;          // device copy (inside the weight store / training workspace)
  int capacity = 0;
  int blocks = 0;
  bool uploaded = false;
  void add(int kind, const float* src, float* dst, size_t total, int p0 = 0, int p1 = 0, int p2 = 0,
           int p3 = 0, int p4 = 0) {
    PackJob j{};
    j.kind = kind; j.src = src; j.dst = dst; j.total = (uint32_t)total;
    j.p[0] = p0; j.p[1] = p1; j.p[2] = p2; j.p[3] = p3; j.p[4] = p4;
    j.block0 = (uint32_t)blocks;
    blocks += (int)((total + PACK_ELEMS_PER_BLOCK - 1) / PACK_ELEMS_PER_BLOCK);
    jobs.push_back(j);
  }
  void pk(const float* src, int ld, int K, int N, float* dst, int Kp, int Np) {
    add(PJ_PK, src, dst, (size_t)Kp * Np, ld, K, N, Kp, Np);
  }
  void pk_t(const float* src, int ld, int K, int N, float* dst, int Kp, int Np) {
    add(PJ_PK_T, src, dst, (size_t)Kp * Np, ld, K, N, Kp, Np);
  }
  void pk_gates(const float* src, int ld, int K, int L, float* dst, int Kp) {
    add(PJ_PK_GATES, src, dst, (size_t)Kp * 4 * L, ld, K, L);
  }
  void vec_gates(const float* src, int L, float* dst) { add(PJ_VEC_GATES, src, dst, (size_t)4 * L, L); }
  void tiles(const float* W, int ld, int row0, int K, int ntiles, int gate_L, float* dst) {
    add(PJ_TILES, W, dst, (size_t)ntiles * K * 16, ld, row0, K, gate_L);
  }
  void tiles64(const float* W, int ld, int row0, int K, int L, float* dst) {
    add(PJ_TILES64, W, dst, (size_t)K * 4 * L, ld, row0, K, L);
  }
  void tiles_t(const float* W, int ld, int row0, int L, float* dst, int Ktot, int k_off) {
    add(PJ_TILES_T, W, dst, (size_t)(L / 16) * L * 64, ld, row0, L, Ktot, k_off);
  }
  void pad(const float* src, int R, int M, float* dst, int Mp) {
    add(PJ_PAD, src, dst, (size_t)R * Mp, R, M, Mp);
  }
};
constexpr int kMaxPackJobs = 112;

// indices into Ctx::vars (order of build_vars)
enum VarId {
  V_ENC_EMB, V_ENC_W0, V_ENC_B0, V_ENC_W1, V_ENC_B1, V_EHT_W, V_EHT_B,
  V_DEC_EMB, V_DEC_GO, V_ATT_V, V_ATT_W, V_ATT_B, V_TOK_W, V_TOK_B,
  V_DEC_W0, V_DEC_B0, V_DEC_W1, V_DEC_B1,
  V_FIND_IMG_W, V_FIND_IMG_B, V_FIND_TXT_W, V_FIND_TXT_B, V_FIND_E_W, V_FIND_E_B,
  V_FSP_IMG_W, V_FSP_IMG_B, V_FSP_TXT_W, V_FSP_TXT_B, V_FSP_ATT_W, V_FSP_ATT_B, V_FSP_E_W,
  V_FSP_E_B,
  V_TR_MAPS_W, V_TR_MAPS_B, V_TR_TXT_W, V_TR_TXT_B, V_TR_E_W, V_TR_E_B,
  V_EXIST_W, V_EXIST_B, V_COUNT_W, V_COUNT_B, V_EQ_W, V_EQ_B, V_MORE_W, V_MORE_B, V_LESS_W,
  V_LESS_B,
  V_SP_TXT_W, V_SP_TXT_B, V_SP_ATT0_W, V_SP_ATT0_B, V_SP_ATT1_W, V_SP_ATT1_B, V_SP_E_W, V_SP_E_B,
  V_DE_TXT_W, V_DE_TXT_B, V_DE_ATT_W, V_DE_ATT_B, V_DE_E_W, V_DE_E_B,
  V_QPN_W1, V_QPN_B1, V_QPN_W2, V_QPN_B2,      // models_vqa/question_prior_net.py (VQA variant only)
  V_COUNT_
};

}  // namespace n2nmn

using namespace n2nmn;

namespace n2nmn {
// Where the forward pass keeps the activations the backward pass needs (set while a training
// forward runs; see capi_train.cpp).  Strides use the ACTUAL batch size N of the call.
struct TrainRec {
  float4 *eg0 = nullptr, *eg1 = nullptr, *dg0 = nullptr, *dg1 = nullptr;   // gates [T][N][L]
  float *ec0s = nullptr, *ec1s = nullptr, *eh0s = nullptr, *eh1s = nullptr; // [(T+1)][N][L]
  float *dc0s = nullptr, *dc1s = nullptr, *dh0s = nullptr, *dh1s = nullptr; // [(Td+1)][N][L]
  float *ctx = nullptr;        // [Td][N][L]
  float *tscores = nullptr;    // [Td][N][V]
  float *lsp = nullptr;        // [N] log_seq_prob
  int32_t *valid_bits = nullptr;   // [Td][N] token validity of the forward (policy gradient)
  float *pooled = nullptr;     // [max_pool][2][D]
  // dropout on the output of LSTM layer 0 (models_vqa training): what the forward keeps is the
  // dropped h0 the layer above consumed (operand of the W1 gradient)
  float *eh0d = nullptr, *dh0d = nullptr;                   // [T][N][L], [Td][N][L] row-major
};
struct TrainState;
}  // namespace n2nmn

struct n2nmn_ctx {
  n2nmn_dims d{};
  int device = 0;
  std::vector<Var> vars;                   // indexed by VarId (absent ones have numel 0)
  std::vector<int> pub;                    // VarIds of the variables of this variant, in order
  std::unordered_map<std::string, int> index;
  // n2nmn_debug_set(ctx, key, value): A/B switches of this context (a fork without its own entry asks its
  // parent); the keys are listed in include/n2nmn.h section 7.  Never read from the environment.
  std::unordered_map<std::string, std::string> knobs;
  bool committed = false;
  bool have_tables = false;
  int mode = 0;                            // N2NMN_MODE_*: tile shape of the recurrent step kernels

  char* base = nullptr;        // weight store (owned by the root context only)
  size_t bytes = 0;
  char* ws_base = nullptr;     // workspace (every context owns its own)
  size_t ws_bytes = 0;
  n2nmn_ctx* parent = nullptr; // forked contexts share the parent's weight store
  // pinned staging ring for the program upload (nodes + tables)
  static constexpr int kStage = 4;
  char* stage[kStage] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t stage_ev[kStage] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes = 0;
  int stage_next = 0;

  int Mp = 0, HWp = 0, KpE = 0, KpL = 0, KpD = 0;
  int max_nodes = 0, max_text = 0, max_pool = 0;

  // packed weights / derived tables
  float *enc_W0x_p = nullptr, *dec_W0x_p = nullptr, *enc_xtab = nullptr, *dec_xtab = nullptr;
  float *enc_b0_t = nullptr, *dec_b0_t = nullptr;    // layer-0 biases in the x-table's tile column order
  float *enc_b1_t = nullptr, *dec_b1_t = nullptr;    // layer-1 biases, same order (LstmJob::bias)
  float *enc_W0h_t = nullptr, *enc_W1_t = nullptr, *dec_W0h_t = nullptr, *dec_W1_t = nullptr;
  // the same four matrices as 16-unit tiles for lstm_tile_kernel (nullptr: lstm_dim % 128 != 0)
  float *enc_W0h_64 = nullptr, *enc_W1_64 = nullptr, *dec_W0h_64 = nullptr, *dec_W1_64 = nullptr;
  // ... and as three bf16 planes per tile for lstm_tile3_kernel (N2NMN_MODE_THROUGHPUT_BF16X3; packed by
  // a commit once the mode has been requested on the root or a fork: b3_on)
  uint16_t *enc_W0h_b3 = nullptr, *enc_W1_b3 = nullptr, *dec_W0h_b3 = nullptr, *dec_W1_b3 = nullptr;
  // ... and of the PK-packed GEMM weights (gemm_dma3_kernel): encoder_h_transform, W_a, the two conv_image sets
  uint16_t *eht_W_b3 = nullptr, *att_W_b3 = nullptr, *find_img_b3 = nullptr, *fsp_img_b3 = nullptr;
  bool b3_on = false;
  float *eht_W_p = nullptr, *att_W_t = nullptr, *att_W_p = nullptr, *find_img_p = nullptr, *fsp_img_p = nullptr;
  float* dec_emb_cat = nullptr;
  PackBatch packs;                                   // every re-pack of a commit, one launch
  PackBatch packs_rest;                              // ... (packs: what the ENCODER reads; packs_rest: the
                                                     // decoder's and the module network's operands)
  PackBatch packs_infer;                             // ... except what only inference reads (64-column tiles)
  float *qpn_W1_p = nullptr, *qpn_W2_p = nullptr;    // PK packs of question_prior_net fc1 / fc2
  float *wans_sp_p = nullptr, *wans_de_p = nullptr;  // PK packs of fc_eltwise (large num_choices only)
  // question vocabularies beyond 4096 words (models_vqa: 17742): the layer-0 input projection is a
  // per-batch GEMM over the batch's own words (xproj [T][N][4L], rows addressed through iota)
  // instead of a [num_vocab_txt][4L] table rebuilt at every weight commit
  float *ehd[2] = {nullptr, nullptr}, *dhd[2] = {nullptr, nullptr};   // dropped layer-0 outputs
  // split-operand bf16 mode: bf16 planes [3][L/8][N][8] of every fp32 state buffer a recurrent step reads
  // as an MFMA operand.  Block A = the 10 N L floats from eh0[0] (encoder states + final states), block B =
  // ehd / dhd / decoder states (10 N L floats from ehd[0]); planes of float offset o of a block start at
  // uint16 offset 3 o of its plane block (hb_A directly behind block A: one clear covers both)
  float *st_A = nullptr, *st_B = nullptr;
  uint16_t *hb_A = nullptr, *hb_B = nullptr;
                                                                      // (packed state layout)
  bool big_vocab = false;
  float* xproj = nullptr;
  int32_t* iota = nullptr;
  bool big_heads = false;                            // map_dim * num_choices beyond the fused head
  float* wtxt_pad[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // walker text maps as a table: ew[ws][v] = encoder embedding_mat[v] . W_txt[ws]  ([V_txt][Mp]),
  // so  fc_text(sum_tau att * emb[seq]) = b + sum_tau att * ew[seq]  (small vocabularies only)
  float* tr_At = nullptr;                  // k-major Transform taps + bias (ModuleWeights::trA)
  float* wtxt_pk[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* ew[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* btxt_pad[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  float* watt_pad[4] = {nullptr, nullptr, nullptr, nullptr};
  float* we_pad[3] = {nullptr, nullptr, nullptr};
  float* batt_pad[4] = {nullptr, nullptr, nullptr, nullptr};
  int32_t *P = nullptr, *Wv = nullptr, *bv = nullptr;
  int32_t* token_op = nullptr;             // [V] op code per layout token (-1: <eos>), device
  bool have_token_ops = false;
  int eos_token = -1;                      // the token whose op code is < 0 (<eos>), host copy
  // host copies of the automaton and the token -> op table, and what they prove (refresh_retire_ok, capi.cpp):
  // from every state the automaton can reach, a row that has emitted <eos> or an answer operator may only emit
  // <eos> from then on.  Only then may the sequential decoder retire such rows (N2NMN_S2S_EOS_RETIRE).
  std::vector<int32_t> P_host, W_host, b_host, token_op_host;
  bool retire_ok = false;

  // seq2seq workspace
  float *eh0[2] = {nullptr, nullptr}, *eh1[2] = {nullptr, nullptr}, *ec0 = nullptr, *ec1 = nullptr;
  float *dh0[2] = {nullptr, nullptr}, *dh1[2] = {nullptr, nullptr}, *dc0 = nullptr, *dc1 = nullptr;
  float *fc0 = nullptr, *fh0 = nullptr, *fc1 = nullptr, *fh1 = nullptr;
  int32_t *perm = nullptr, *nact = nullptr;
  int32_t *enc_rows = nullptr, *enc_rows_n = nullptr;   // rows (t, n) inside their length, and how many
  // eos_retire (N2NMN_S2S_EOS_RETIRE): layout lengths, rows ranked by them, live rows per decoder step,
  // the live (step, row) pairs and their number
  bool dec_retired = false;     // the last decoder call retired rows (its atts / token_probs are partial)
  int32_t *dlen = nullptr, *dperm = nullptr, *dnact = nullptr, *drows = nullptr, *drows_n = nullptr;
  float *qpn_h = nullptr, *qpn_hid = nullptr;        // [N][2L] concat of final h, [N][qpn_hidden]
  float *enc_out = nullptr, *eht = nullptr, *qbuf = nullptr, *dec_h1_all = nullptr, *ent_t = nullptr, *dh1_rm = nullptr;
  // `eht` rows (tau, n) with tau >= len[n] were NOT computed by the last encoder pass (listed-row GEMM,
  // encoder_impl): they hold whatever an earlier pass left.  Every reader must take the
  // encoder_h_transform bias for them (DecStepArgs::eht_bias) or refuse a partial matrix.
  bool eht_partial = false;
  long eht_listed_rows = -1;   // rows of that listed-row GEMM when the host knows the lengths (profile figures)
  int32_t *state = nullptr, *next_idx = nullptr, *tokens = nullptr;
  float *tprobs = nullptr, *negent = nullptr, *atts = nullptr, *word_vecs = nullptr;
  int enc_T = 0, enc_N = 0;            // shape of the encoder results currently held
  const int32_t* enc_seq = nullptr;    // input_seq of the last encoder call (for word_vecs)
  const int32_t* enc_len = nullptr;

  // module workspace
  float* wtmap = nullptr;                  // [T_dec][N][Mp] text maps of the walker path
  float* watt = nullptr;                   // [N][T_dec][HWp] Find / Filter logits (walk_find_kernel)
  // deferred pooling of the walker path: job code, soft-max weights, text map, pooled features
  int32_t* wpjob = nullptr; float *wpw = nullptr, *wptm = nullptr, *wpooled = nullptr, *wpfc = nullptr;
  // staged walker (kernels.h WalkArgs::staged): decoded layouts of this context's questions, and -- for
  // the launches this context issues -- the job lists and the two counter sets (used alternately)
  WalkProg* wprog = nullptr;
  float* wfpart = nullptr;   // [N][T_decoder][WALK_POOL_PARTS][Mp]: FindSameProperty fc_att shares (staged walker)
  int32_t *whjobs = nullptr, *wfblist = nullptr, *wcnt = nullptr, *wplist = nullptr;
  int whoff[WALK_HLEVELS + 1] = {0}, walk_parity = 0;
  int32_t* walk_hint_host = nullptr;   // host-mapped word the staged walker reports its deepest nesting in
  int32_t* walk_hint_dev = nullptr;
  int walk_hint_prev = 0;
  int walk_levels = 0;                        // n2nmn_walk_set_levels: 0 adaptive, >= 1 fixed
  bool walk_conv_inline = false;              // n2nmn_walk_set_conv_inline: the NEXT walk_layouts call computes the conv_image maps
  int walk_nesting_bound = -1;                // n2nmn_walk_set_nesting_bound: promise for the NEXT walk_layouts call
  int walk_staged = -1;                       // -1 auto (with the chip-wide front end + deferred pooling), 0 off
  float *arena = nullptr, *tmap = nullptr, *pfc = nullptr, *mfind = nullptr, *mfsp = nullptr;
  float* ev_out = nullptr;
  int32_t* ev_rows = nullptr;
  DevNode* dev_nodes = nullptr;
  int32_t* dev_tab = nullptr;
  bool tokens_via_levels = false;   // n2nmn_set_tokens_via_levels
  int32_t* dsched = nullptr;   // device-scheduled level path: (offset, count) per launch slot, then the overflow flag
  int max_tab = 0;

  n2nmn_program* scratch_prog = nullptr;   // used by n2nmn_module_forward

  // training step (capi_train.cpp)
  uint64_t commit_epoch = 0;               // bumped by every n2nmn_commit_weights
  const TrainRec* rec = nullptr;           // != nullptr while a training forward records
  TrainState* train = nullptr;             // owned; created by n2nmn_train_enable

  // per-kernel-family HIP-event profiler (n2nmn_profile_*)
  bool prof_on = false;
  std::vector<hipEvent_t> prof_events;      // pairs
  struct ProfRec { int fam; double flops, bytes; };
  std::vector<ProfRec> prof_recs;
  double prof_ms[24] = {0}, prof_flops[24] = {0}, prof_bytes[24] = {0};
  long prof_launches[24] = {0};
  WalkArgs last_walk{};                       // arguments of the last walker launch (debug replay)
  bool have_last_walk = false;
  int last_walk_T_enc = 0;                    // T_enc of the last chip-wide front end (debug replay)
  int walk_pre_find = -1;                     // -1 auto (>= 128 questions), 0 in the walker, 1 chip-wide
  int walk_defer_pool = -1;                   // -1 auto (>= 128 questions per launch), 0 never, 1 always
  double walk_jobs_deferred = 0;              // pooling jobs of the last profiled passes
  long long* walk_timeline = nullptr;         // n2nmn_debug_walk_timeline (caller-owned)
  unsigned long long* walk_stats = nullptr;   // device [WALK_STATS]: node counts of profiled walks
};


namespace n2nmn {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }



struct Carver {
  char* base;
  size_t off = 0;
  explicit Carver(char* b) : base(b) {}
  template <typename T>
  T* take(size_t count) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

enum Family {
  F_LSTM_ENC = 0, F_LSTM_DEC0, F_LSTM_DEC1, F_LINEAR_Q, F_DEC_STEP, F_GEMM_EHT, F_WORD_VECS,
  F_TEXTMAP, F_CONV_IMAGE, F_ATT_OPS, F_POOL, F_HEADS,
  F_LSTM_BWD, F_GEMM_TN, F_BWD_MISC, F_OPTIMISER, F_WALK, F_GEMM_MULTI, F_WALK_FIND, F_WALK_TMAP,
  F_SCHED, F_COUNT
};
extern const char* kFamilyNames[F_COUNT];

// Brackets one launch with HIP events on the launch stream when profiling is enabled.
struct ProfScope {
  n2nmn_ctx* c; hipStream_t s; bool on;
  ProfScope(n2nmn_ctx* c_, int fam, double flops, double bytes, hipStream_t s_)
      : c(c_), s(s_), on(c_->prof_on) {
    if (!on) return;
    const size_t i = c->prof_recs.size();
    while (c->prof_events.size() < 2 * (i + 1)) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
      c->prof_events.push_back(e);
    }
    c->prof_recs.push_back({fam, flops, bytes});
    (void)hipEventRecord(c->prof_events[2 * i], s);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(c->prof_events[2 * (c->prof_recs.size() - 1) + 1], s);
  }
};

const n2nmn_ctx* root(const n2nmn_ctx* c);
// switches of n2nmn_debug_set (capi.cpp): this context's entry, else its parent's, else nullptr / the default
const char* knob_str(const n2nmn_ctx* c, const char* key);
int knob_int(const n2nmn_ctx* c, const char* key, int dflt);
bool is_committed(const n2nmn_ctx* c);
bool has_tables(const n2nmn_ctx* c);
hipStream_t S(n2nmn_stream s);
int check_launch(const char* what);
ModuleWeights module_weights(const n2nmn_ctx* c);
void packed_state(const n2nmn_ctx* c, LstmJob& j);
void rowmajor_a(const n2nmn_ctx* c, LstmJob& j);
int qpn_forward(n2nmn_ctx* c, int N, float* scores, const float* drop_h, const float* drop_fc1,
                hipStream_t s);
int encoder_impl(n2nmn_ctx* c, const n2nmn_seq2seq_io* io, hipStream_t s, GemmArgs* defer_eht = nullptr);
int decoder_impl(n2nmn_ctx* c, const n2nmn_seq2seq_io* io, hipStream_t s, const GemmArgs* pre = nullptr,
                 int npre = 0);
void train_state_destroy(TrainState* t);
// `waiter` waits for the training step's side stream (weight-gradient GEMMs, late-bucket finish)
void train_side_join(n2nmn_ctx* c, hipStream_t waiter);
// inference-only operand refresh of a commit (capi.cpp: n2nmn_commit_weights): the stream it runs on
// (the training step's side stream, ordered after `s`; `s` itself when the context does not train),
// the marker after it, and the wait of whoever reads those operands
hipStream_t train_infer_fork(n2nmn_ctx* c, hipStream_t s);
void train_infer_done(n2nmn_ctx* c, hipStream_t side);
void train_infer_wait(const n2nmn_ctx* root, hipStream_t s);
void train_infer_host_wait(const n2nmn_ctx* root);   // same, blocking the host (null-stream copies)
// n2nmn_commit_weights with the decoder's / module network's / inference-only operands on `rest`
// (== s: everything in stream order, the public entry point)
int commit_weights_on(n2nmn_ctx* c, hipStream_t s, hipStream_t rest);
enum { RP_PREP = 1, RP_CONV = 2, RP_REST = 4, RP_ALL = 7 };
int run_program(n2nmn_ctx* c, Program& p, const float* feat, const float* word_vecs, int N_full,
                float* scores, const float* ext0, const float* ext1, float* att_out,
                int att_out_first, int att_out_count, hipStream_t s, int stages = RP_ALL);

}  // namespace n2nmn
