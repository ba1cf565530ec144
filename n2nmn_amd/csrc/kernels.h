// Host-callable launchers of the gfx950 kernels (implemented in the .hip files).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "program.h"

namespace n2nmn {

// ---------------------------------------------------------------------------------------------
// packed operand layouts
//   PK layout  [Kp/4][Np][4]           : k-interleaved B operand of gemm_pk (Kp % 32 == 0, Np % 64 == 0)
//   LSTM tiles [L/4 tiles][K/4][16][4] : per 4-hidden-unit column tile, columns ordered
//                                        gate-major (i,j,f,o) x 4 units, k-interleaved
// ---------------------------------------------------------------------------------------------
inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct GemmArgs {
  const float* A; int lda; int M; int K;
  const int32_t* group_idx; int group_size;   // optional row-group gather: logical row r reads
                                              // source row group_idx[r/gs]*gs + r%gs
  int src_rows;                               // with group_idx: rows of the SOURCE table A (the gathered
                                              // row is not bounded by M); 0 = unknown (kernels that
                                              // need the bound -- 32-bit offsets -- then refuse)
  const float* Bp; int Np; int Kp;
  const uint16_t* Bp3;                        // optional: Bp as three bf16 planes (launch_pack_pk_b3): the
                                              // launch may run as split-operand bf16 (gemm_dma3_kernel)
  const float* bias; int N;
  float* C; int ldc; int n_store;
  int accumulate;                             // != 0: C += A.B (+ bias) instead of C = ...
  int ksplit;                                 // > 1: K split over blockIdx.z, atomic C += (needs accumulate)
  int relu;                                   // != 0: C = max(0, A.B + bias)
  const int32_t* c_row_idx;                   // optional: result row r goes to C row c_row_idx[r]
                                              // (negative: the row is not stored)
  // optional gate (hoisted conv_image of an operator only some layouts use): a row tile is computed
  // only if a layout of one of its row groups (group = gate_rows consecutive rows = one image)
  // contains a token whose op code is gate_op.  gate_tokens [gate_T][gate_N] device.
  const int32_t* gate_tokens; const int32_t* gate_token_op;
  int gate_T, gate_N, gate_V, gate_op, gate_rows;
  // optional row count on the device: only rows [0, min(M, *m_dev)) exist (M bounds the launch
  // geometry; row tiles past the count return at once).  With group_idx / c_row_idx = a compacted row
  // list this is "the GEMM over the listed rows": encoder_h_transform over the rows inside their
  // question's length.
  const int32_t* m_dev;
};
void launch_gemm_pk(const GemmArgs& a, hipStream_t s);
// two problems (no split-K) in one launch
void launch_gemm_pk2(const GemmArgs& a0, const GemmArgs& a1, hipStream_t s);
// up to four problems (no split-K) as one flat, XCD-ordered tile list
struct GemmBatch { GemmArgs a[4]; int start[5]; };
void launch_gemm_pkn(const GemmArgs* a, int n, hipStream_t s);
// LDS-DMA-staged 128 x 128 tiles (kernels_gemm_dma.hip) for problems gemm_dma_supported() accepts;
// launch_gemm_pkn / launch_gemm_pk route there when every problem of the launch qualifies
bool gemm_dma_supported(const GemmArgs& a);
void launch_gemm_dma(const GemmArgs* a, int n, hipStream_t s);
// the same contraction on bf16 MFMAs over three-way split operands (kernels_gemm_dma3.hip; opt-in mode
// N2NMN_MODE_THROUGHPUT_BF16X3): problems that carry Bp3.  dst: 3 * Kp * Np bf16.
bool gemm_dma3_supported(const GemmArgs& a);
void launch_gemm_dma3(const GemmArgs* a, int n, hipStream_t s);
void launch_pack_pk_b3(const float* Bp, int Kp, int Np, uint16_t* dst, hipStream_t s);

// generic packer: dst PK layout <- src[k*ld + n] (k < K, n < N), zero padded
void launch_pack_pk(const float* src, int ld, int K, int N, float* dst, int Kp, int Np,
                    hipStream_t s);
// dst[r][0..Mp) = src[r][0..M) zero padded
void launch_pad_rows(const float* src, int R, int M, float* dst, int Mp, hipStream_t s);
// column-tile packer: src rows [row0, row0+K) of a [*, ld] matrix -> [tile][K/4][16][4].
// gate_L > 0: LSTM gate interleave, tile j column c <- source column (c>>2)*gate_L + 4j + (c&3)
// gate_L == 0: plain, tile j column c <- source column 16j + c
void launch_pack_tiles(const float* W, int ld, int row0, int K, int ntiles, int gate_L, float* dst,
                       hipStream_t s);

// All operand re-packs of one weight commit in ONE launch: a job table (built once per context,
// the pointers never change) replaces ~40 tiny pack / pad / copy launches per optimiser step.
enum PackKind : int32_t { PJ_PK = 0, PJ_PK_T, PJ_TILES, PJ_TILES_T, PJ_PAD, PJ_PK_GATES, PJ_VEC_GATES,
                          PJ_TILES64 };
struct PackJob {
  int32_t kind;
  int32_t p[7];            // PK / PK_T: ld, K, N, Kp, Np    TILES: ld, row0, K, gate_L
                           // TILES_T: ld, row0, L, Ktot, k_off    PAD: R, M, Mp
                           // TILES64: ld, row0, K, L  (dst [L/16][K/4][64][4], column c = 16 gate + unit)
  const float* src;
  float* dst;
  uint32_t total;          // elements this job iterates over
  uint32_t block0;         // first workgroup of this job
  int32_t pad_[2];
};
static_assert(sizeof(PackJob) == 64, "PackJob is uploaded verbatim");
constexpr int PACK_ELEMS_PER_BLOCK = 4096;
void launch_pack_jobs(const PackJob* jobs_dev, int njobs, int total_blocks, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// seq2seq
// ---------------------------------------------------------------------------------------------
struct LstmJob {
  const float* A0;        // [N][L] source of k in [0, L)
  const float* A1;        // [N][L] source of k in [L, 2L) (layer-1 jobs), else nullptr
  int a_rs, a_ks;         // A addressing in floats: element (row, k) at row*a_rs + (k/4)*a_ks + k%4
                          //   row-major [rows][L]: a_rs = L, a_ks = 4
                          //   k-interleaved state layout [L/4][R][4]: a_rs = 4, a_ks = 4*R
  int hp_R;               // > 0: h_old / h_new use the k-interleaved layout with R rows per k4
  int K;                  // L or 2L
  const float* Wp;        // packed tiles for this job
  const float* Wp64;      // the same weights as 16-unit tiles [L/16][K/4][64][4] (column c = 16 * gate +
                          // unit) for lstm_tile_kernel, or nullptr
  int ntiles;             // number of 16-column tiles (LSTM: L/4; linear: Ncols/16)
  int mode;               // 0: LSTM cell epilogue; 1: plain linear  out = z + bias
  const float* xtab;      // [V][4L] input-projection table incl. bias (layer 0) or nullptr; columns in
                          // TILE order: column 16 * tile + 4 * gate + unit holds gate `gate` of hidden
                          // unit 4 * tile + unit, so the 16 values a 4-unit tile adds are one 64-byte line
  const int32_t* xidx;    // [N] row of xtab per batch row (layer 0) or nullptr
  int xidx_const;         // used when xidx == nullptr && xtab != nullptr (go embedding row)
  const float* bias;      // [4L] (layer 1; LSTM cell jobs: in the tile column order of xtab) or nullptr
  const float* c_in;      // [N][L] previous cell state
  float* c_out;           // [N][L] new cell state (may alias c_in: each element has one owner)
  const float* h_old;     // [N][L] previous hidden state of THIS layer (copied when masked)
  float* h_new;           // [N][L]  (linear mode: output [N][ldo])
  int ldo;                // linear mode: row stride of the output
  float* out_seq;         // [N][L] slice of encoder_outputs (zeros when masked) or nullptr
  const int32_t* seq_len; // [N] or nullptr (no masking); indexed by ORIGINAL row
  int t;                  // time step compared against seq_len
  // length-sorted encoder: state row r holds original row perm[r]; rows >= *n_active are past
  // their length (whole 16-row tiles beyond it skip their loads and MFMAs)
  const int32_t* perm;    // [N] or nullptr (identity)
  const int32_t* n_active;// device scalar for this step or nullptr (= N)
  float* fin_c;           // [N][L] row-major, ORIGINAL order: c of a row at the step it finishes
  float* fin_h;           // k-interleaved [L/4][R][4], ORIGINAL order (or nullptr)
  int active;             // 0: skip this job entirely (pipeline fill / drain)
  // training: activations kept for the backward pass (ORIGINAL row order), or nullptr
  float4* save_gates;     // [N][L] (i, j, f, o) after their nonlinearities, this step
  float* save_c;          // [N][L] cell state after this step
  float* save_h;          // [N][L] hidden state after this step (row-major)
  // DropoutWrapper on this layer's OUTPUT (models_vqa/nmn3_netgen_att.py:17-44): the layer above
  // reads h * drop, the recurrent state stays h.  drop: multipliers [N][L], ORIGINAL row order.
  const float* drop;      // nullptr: no dropout
  float* h_drop;          // h * drop in the layout of h_new (the next layer's A0)
  float* save_hd;         // [N][L] row-major copy of it (training: operand of the W1 gradient)
  // split-operand bf16 mode (lstm_tile3_kernel, kernels_lstm_tile3.hip): every fp32 state buffer above
  // that a recurrent step READS as an MFMA operand has a companion of three bf16 planes
  // [3][L/8][R][8] (x = hi + mid + lo exactly); the step that writes a state writes its planes too.
  // All nullptr: the job runs on the exact-fp32 kernels.
  const uint16_t* A0b;    // planes of A0 / A1
  const uint16_t* A1b;
  uint16_t* h_new_b;      // planes of h_new, fin_h, h_drop (nullptr where the fp32 pointer is)
  uint16_t* fin_h_b;
  uint16_t* h_drop_b;
  const uint16_t* Wb3;    // the job's weights as three bf16 planes per 16-unit tile:
                          // [L/16][K/32][3 planes][4 gates][64 lanes][8]
};
// rows_per_wg: 64 (4 M-tiles per workgroup) or 32 (2 M-tiles; doubles the workgroups of a launch)
// wide != 0: 32-row x 32-column workgroup tiles for LSTM cell jobs (throughput mode)
// wide == 2: LDS-staged 64-row x 64-gate-column tiles (lstm_tile_kernel) where the jobs allow it
void launch_lstm_step(const LstmJob* jobs, int njobs, int N, int L, int rows_per_wg,
                      hipStream_t s, int wide = 0);
// lstm_tile_kernel (kernels_lstm_tile.hip): LSTM cell jobs on the packed state layout, K in {L, 2L}
bool lstm_tile_supported(const LstmJob* jobs, int njobs, int L);
void launch_lstm_tile(const LstmJob* jobs, int njobs, int N, int L, int stages, hipStream_t s);
// lstm_tile3_kernel (kernels_lstm_tile3.hip): the same step on bf16 MFMAs over three-way split operands
// (6 cross products, fp32 accumulate): opt-in mode N2NMN_MODE_THROUGHPUT_BF16X3
bool lstm_tile3_supported(const LstmJob* jobs, int njobs, int L);
void launch_lstm_tile3(const LstmJob* jobs, int njobs, int N, int L, hipStream_t s, int variant = 0);
// dst[L/16][K/32][3][4][64][8] bf16 planes of W[row0 + k][gate * L + unit] (k < K), see LstmJob::Wb3
void launch_pack_tiles64_b3(const float* W, int ld, int row0, int K, int L, uint16_t* dst, hipStream_t s);
// planes [3][L/8][R][8] of a k-interleaved fp32 state buffer [L/4][R][4] (tests / debug entry points)
void launch_split_state_b3(const float* h, int L, int R, uint16_t* planes, hipStream_t s);

// Arguments of dec_attn_kernel.  Every per-step pointer is the slice of the FIRST step of the
// launch; workgroup (n, ts) addresses element ts*N + n of it.
// k-interleaved state [L/4][R][4] -> row-major [N][L]
void launch_unpack_h(const float* src, float* dst, int N, int L, int R, hipStream_t s);
// dst[n][0:L) = unpack(a)[n], dst[n][L:2L) = unpack(b)[n]
void launch_unpack_h2(const float* a, const float* b, float* dst, int N, int L, int R,
                      hipStream_t s);
void launch_lstm_step_dbg(const LstmJob* jobs, int njobs, int N, int L, int rows_per_wg,
                          int variant, hipStream_t s);

struct DecStepArgs {
  // inputs
  const float* q;          // [steps][N][L]   out . W_a + b_a
  const float* out;        // [steps][N][L]   top-layer h
  const float* eht;        // [T][N][L]
  const float* eht_bias;   // [L] bias of encoder_h_transform, or nullptr.  Rows past a question's
                           // length are fc(0) = this bias exactly (dynamic_rnn emits zero rows,
                           // Appendix A.2), so the batched attention evaluates them once
  const float* eout;       // [T][N][L]
  const int32_t* seq_len;  // [N]
  const int32_t* order;    // [N] questions by decreasing length (enc_prepare's perm) or nullptr: the
                           // per-question kernel takes its workgroups' questions in this order, so the
                           // dispatcher hands the longest questions out first (LPT schedule)
  const float* v;          // [L]
  const float* Wy;         // [2L][V]
  const float* by;         // [V]
  const int32_t* P;        // [V][3]
  const int32_t* Wv;       // [3][V][4]
  const int32_t* bv;       // [V][4]
  const int32_t* gt;       // [steps][N] teacher-forcing tokens or nullptr
  const float* uni;        // [steps][N] uniforms or nullptr
  const int32_t* forced;   // [steps][N] or nullptr
  int use_gt;              // 0 free-running (state), 1 ground truth (all tokens valid), 2 given
                           // tokens with the automaton's validity (state = prefix sum of P)
  int T, N, L, V;
  int Td;                  // use_gt == 2: total decoder steps (initial automaton state)
  int32_t* valid_bits;     // optional [steps][N]: bit v = token v valid at that step
  // state / outputs
  int32_t* state;          // [N][3]   (sequential decoding only)
  int32_t* tokens;         // [steps][N]
  float* tprobs;           // [steps][N]
  float* ent_t;            // [steps][N] per-step entropy terms (summed by word_vecs_kernel)
  float* atts;             // [steps][T][N]
  float* scores;           // [steps][N][V] or nullptr
  int32_t* next_idx;       // [N] row of the decoder x-table for the next step (= token) or nullptr
  float* ctx_out;          // [steps][N][L] context vectors kept for the backward pass, or nullptr
  // eos_retire (teacher-forced passes: dec_attn_question_kernel / dec_attn_multi_kernel): live decoder steps per question;
  // steps at or past it get their token from `gt` and nothing else.  nullptr: every step of every question
  const int32_t* dec_len;  // [N]
  // eos_retire, sequential decoding (dec_attn_seq_kernel; dec_compact_kernel builds these after every step):
  // workgroup j serves original row live_perm[j]; j >= *live_n: the row has emitted <eos> and gets eos_token.
  // q is read at state row j, every other operand by original row.  nullptr: all rows live, identity
  const int32_t* live_perm;
  const int32_t* live_n;
  int eos_token;
};
// can launch_dec_attn serve this launch with dec_attn_question_kernel (the kernel that honours dec_len)?
bool dec_question_supported(const DecStepArgs& a, int nsteps);
bool dec_len_supported(const DecStepArgs& a, int nsteps);
// eos_retire helpers (kernels_seq2seq.hip): layout lengths from the tokens; state rows gathered by `perm`
bool dec_seq_retire_supported(const DecStepArgs& a);
void launch_dec_compact(const int32_t* tokens, const int32_t* token_op, int V, const int32_t* perm_old,
                        const int32_t* n_old, int32_t* perm_new, int32_t* n_new, const float* const src[4],
                        float* const dst[4], const uint16_t* const srcb[2], uint16_t* const dstb[2], int N, int L,
                        int R, hipStream_t s);
void launch_dec_len(const int32_t* tokens, const int32_t* token_op, int V, int T_dec, int N,
                    int32_t* dec_len, hipStream_t s);
void launch_gather_state(const float* const src[4], float* const dst[4], const uint16_t* const srcb[2],
                         uint16_t* const dstb[2], const int32_t* perm, int N, int L, int R, hipStream_t s);
// nsteps == 1: one sequential step (1024-thread workgroups); nsteps > 1: all steps in one launch
void launch_dec_attn(const DecStepArgs& a, int nsteps, hipStream_t s);

void launch_dec_init(int32_t* state, int N, int T_dec, hipStream_t s);
// perm = rows sorted by decreasing length (stable); n_active[t] = #{n : seq_len[n] > t}, t < T
// also zeroes `zero_floats` floats at `zero` (multiple of 4; the recurrent state block) in the same launch
// rows[0 .. *count) = { t*N + n : t < seq_len[n] } in any order (*count must be 0 on entry:
// launch_enc_prepare clears it)
void launch_enc_rows(const int32_t* seq_len, int T, int N, int32_t* rows, int32_t* count,
                     hipStream_t s);
void launch_enc_prepare(const int32_t* seq_len, int N, int T, int32_t* perm, int32_t* n_active,
                        float* zero, size_t zero_floats, hipStream_t s, int32_t* zero_int = nullptr);

// word_vecs[t][n][:] = sum_tau atts[t][tau][n] * emb[seq[tau][n]][:];  log_seq_prob
void launch_word_vecs(const float* atts, const int32_t* seq, const float* emb, int T_dec,
                      int T_enc, int N, int E, float* word_vecs, const float* tprobs,
                      const float* ent_t, float* neg_entropy, float* log_seq_prob, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// module network
// ---------------------------------------------------------------------------------------------
struct ModuleWeights {     // device pointers into the context's packed weight store
  // text maps: zero-padded [E][Mp] + bias [Mp]; ws = Find, FSP, Transform, SameProperty, Describe
  const float* Wtxt[5]; const float* btxt[5];
  // conv_eltwise of Find / FSP / Transform: w [M], b scalar (device)
  const float* we[3]; const float* be[3];
  // Transform conv_maps [k*k][M], bias [M]
  const float* Kt; const float* bt;
  // the same, k-major and zero padded for the walker's MFMA Transform: [(k*k + 1 + 3) & ~3][Mq],
  // rows 0..k*k-1 = taps, row k*k = bias, Mq = round_up(M, 16)
  const float* trA;
  // fc_att of FSP, SameProperty(0,1), Describe: zero-padded [D][Mp] + bias [Mp]
  const float* Watt[4]; const float* batt[4];
  // answer FCs: Exist [3][C], Count [HW+2][C], Equal/More/Less [2HW+4][C], SameProp/Describe [M][C]
  const float* Wans[7]; const float* bans[7];
};

struct ModuleBuffers {
  const DevNode* nodes;   // device copy of Program::dev_nodes
  const int32_t* tab;     // device copy of Program::tab
  float* arena;           // [max_nodes][HWp]
  float* tmap;            // [max_text][Mp]
  float* pfc;             // [max_pool][2][POOL_PARTS][Mp] partial fc_att outputs
  float* mfind;           // [N][HW][Mp]
  float* mfsp;            // [N][HW][Mp]
  const float* feat;      // [N_full][HW][D]
  const float* word_vecs; // [T_dec][N_full][E]
  float* scores;          // [rows][C]
  int N_full, H, W, D, M, Mp, E, C, HWp, ksize;
  int wl_cap;             // floats of LDS the answer heads may use to stage fc weights
  float* pooled;          // [max_pool][2][D] attention-pooled features kept for backward, or nullptr
  int vqa;                // models_vqa: no conv Transform / raw-map answer heads (no LDS for them)
  // large answer vocabularies (map_dim * num_choices too big for the fused head): the head kernel
  // only writes the normalised vectors and the fc_eltwise runs as one gemm_pk per launch
  float* ev_out;          // [count][Mp] or nullptr (fused fc)
  int32_t* ev_rows;       // [2][max_pool]: result row of entry i for Describe / SameProperty, or -1
  int ev_stride;          // max_pool
  // device-scheduled launches (sched_kernel): (offset, count) per launch slot, or nullptr
  const int32_t* dsched;
  int ev_by_q;            // ev_out row = the question (one fc_eltwise GEMM after the last level)
};

// ---- layout walker (kernels_walk.hip): one workgroup runs one question's whole module network
// straight from its RPN tokens.  A launch covers the questions of up to WALK_MAX_BATCHES in-flight
// batches ("super-bucket"): workgroup q handles question q % N of batch q / N.
constexpr int WALK_THREADS = 512;
constexpr int WALK_POOL_ROWS = 38;   // feature rows a thread keeps in flight (H*W / (512 / (D/4)))
constexpr int WALK_MAX_T = 32;
constexpr int WALK_MAX_BATCHES = 16;
constexpr int WALK_HLEVELS = 24;     // nesting levels of Transform / FindSameProperty the staged walker lists
constexpr int WALK_CNT = 64;         // counters per set (WalkArgs::cnt)
constexpr int WALK_POOL_PARTS = 8;    // channel parts of a deferred pooling job (walk_pool_kernel)
constexpr int WALK_POOLK_ROWS = 10;   // feature rows a walk_pool_kernel thread keeps in flight
constexpr int WALK_MAX_PIXEL_GROUPS = 3;   // H*W <= 192 (64-pixel groups a Transform wave holds)
struct WalkBatch {
  const int32_t* tokens;   // [T][N] layout tokens (decoder output or ground truth), device
  const float* feat;       // [N][HW][D]
  const float* word_vecs;  // [T][N][E]
  float* scores;           // [N][C]
  int32_t* validity;       // [N] 1 / 0 (expr_validity_array) or nullptr
  const float* mfind;      // [N][HW][Mp] conv_image maps, FindModule weights
  const float* mfsp;       // [N][HW][Mp] conv_image maps, FindSamePropertyModule weights (only the
                           // images whose layout has a _FindSameProperty token are filled in)
  float* tmap;             // [T][N][Mp] text maps (walk_textmap_kernel / walk_tmap_kernel fill the rows that
                           // are read)
  float* watt;             // [N][T][HWp] Find / Filter logits written by walk_find_kernel (pre_find)
  // deferred pooling (WalkArgs::defer_pool): per question job code (0 none / op), soft-max weights
  // [N][2][HWp], text map [N][Mp], pooled features [N][2][D]
  int32_t* pjob; float* pw; float* ptm; float* pooled;
  float* pfc;              // [N][2][WALK_POOL_PARTS][Mp] partial fc_att rows of the deferred jobs
  // attention-table text maps (T_enc > 0): word_vecs / tmap unused
  const float* atts;       // [T][T_enc][N]
  const int32_t* seq;      // [T_enc][N]
  const int32_t* seq_len;  // [N]
  // staged walker (WalkArgs::staged): the decoded layout of every question, written by
  // walk_tmap_kernel (which becomes the pass's "plan" step), read by walk_heavy / walk_light
  struct WalkProg* prog;   // [N]
  // FindSameProperty as chip-wide stages: the WALK_POOL_PARTS shares of fc_att(pooled features) of node
  // (n, t), written by stage A (items of walk_heavy_kernel), summed by stage B (walk_fspepi_kernel)
  float* fpart;            // [N][T][WALK_POOL_PARTS][Mp]
};
// One question's decoded layout (nmn3_assembler.py:153-222 on the device).  op: n2nmn_op of node t
// | 0x80 for answer-type nodes; in0 / in1: input nodes or -1; hd: "heavy depth" = the largest number of
// Transform / FindSameProperty nodes on a path from a leaf up to and including the node; lo: first
// node of the node's subtree (a subtree is a contiguous token range in Reverse-Polish order).
struct WalkProg {
  int32_t nn, valid, fallback, nfind;
  uint8_t op[WALK_MAX_T];
  int8_t in0[WALK_MAX_T], in1[WALK_MAX_T];
  uint8_t hd[WALK_MAX_T], lo[WALK_MAX_T];
  uint8_t flist[WALK_MAX_T];   // the nfind nodes that read the FindModule conv_image map (Find, Filter)
};
static_assert(sizeof(WalkProg) == 16 + 6 * WALK_MAX_T && sizeof(WalkProg) % 16 == 0, "WalkProg layout");
struct WalkArgs {
  WalkBatch b[WALK_MAX_BATCHES];
  int K, N, T, V;
  const int32_t* token_op; // [V] device: op code of each layout token, -1 for <eos>
  int H, W, D, M, Mp, HWp, E, C, ksize;
  int defer_pool;          // root Describe / SameProperty -> walk_pool_kernel + walk_heads_kernel
  int pre_find;            // Find / Filter logits come from walk_find_kernel (watt), text maps from tmap
  int T_enc, V_txt;        // T_enc > 0: text maps from ew[ws][seq] weighted by atts
  const float* ew[5];      // [V_txt][Mp] embedding_mat . W_txt[ws]
  // staged walker (passes of many questions, throughput mode): the tree-dependent work leaves the
  // one-workgroup-per-question chain.  walk_tmap_kernel decodes every layout (prog), lists the
  // Transform / FindSameProperty nodes whose input subtree holds no other such node (hjobs: they run
  // chip-wide in walk_heavy_kernel, one workgroup per node) and the questions with deeper nesting
  // (fblist: the one-workgroup walker serves those as before); walk_light_kernel finishes the others.
  // cnt[0] = Transform jobs of level 0, cnt[1] = fallback questions, cnt[2] = FindSameProperty jobs of
  // level 0, cnt[3], cnt[4] = pooled roots (Describe / SameProperty), cnt[5] = deepest nesting seen in the
  // pass, cnt[6 + 2 lv + kind] = jobs of level lv >= 1 (kind 0 Transform, 1 FindSameProperty).
  // A node's level = number of Transform / FindSameProperty nodes below it in its own subtree; level lv
  // runs in the lv-th walk_heavy launch (its inputs of lower levels are in `watt` by then).  Nodes of one
  // level have disjoint subtrees of >= lv + 2 tokens, so a level holds at most T / (lv + 2) of them per
  // question: hjobs[hoff[lv] .. hoff[lv + 1]) is level lv, first half Transform, second half
  // FindSameProperty.
  int staged;
  int32_t* hjobs;          // (question << 8) | node
  int32_t* fblist;         // [K * N] flat question indices
  int32_t* cnt;            // [WALK_CNT] this pass's counters (cnt[3] / cnt[4]: deferred Describe / SameProperty jobs)
  int32_t* cnt_next;       // [WALK_CNT] the other set: walk_fcatt_kernel zeroes it for the next pass (this pass's
                           // set stays readable for n2nmn_debug_walk_replay)
  int hoff[WALK_HLEVELS + 1];
  int hlevels, hlevel;     // levels listed by this pass's plan (deeper nesting: fall-back list); level of this launch
  int32_t* hint;           // host-mapped word: deepest nesting of the pass (written once, by walk_fcatt_kernel)
  int no_fallback;         // the host vouches that no layout nests deeper than hlevels: no fall-back launch follows
  // deferred pooling (defer_pool): the questions whose root pools, listed per operator by whoever
  // writes pjob (walk_light_kernel / walk_kernel): plist[0 .. pcap) Describe, plist[pcap ..) SameProperty
  int32_t* plist;
  int pcap;
  // profiling only: [0] conv_image map reads (one per <= 4 Find / Filter nodes of a question, one per
  // FindSameProperty node), [1] pooled inputs, [2] pooling nodes, [3] text maps,
  // [4] Transform nodes, [5] valid questions, [6] deferred pooling jobs, [7] their inputs,
  // [8] map passes done by walk_find_kernel instead of the walker (atomic adds by thread 0)
  unsigned long long* stats;
  // debugging only: [question][WALK_MAX_T][4] shader-clock stamps of thread 0 per node
  // (start, after text map, after pooling + fc_att, end)
  long long* timeline;
};
constexpr int WALK_STATS = 10;
constexpr int WALK_FIND_PARTS = 8;   // workgroups of walk_find_kernel per question (rows of the map)
int walk_supported(int H, int W, int D, int M, int Mp, int HWp, int E, int C, int T, int ksize,
                   int T_enc);
void launch_walk_textmap(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
// chip-wide front end (table text maps, Find / Filter epilogues): see kernels_walk.hip
void launch_walk_tmap(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk_find(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk_heavy(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk_fspepi(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk_light(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk_pool(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
void launch_walk_heads(const ModuleWeights& w, const WalkArgs& a, hipStream_t s);
int walk_pool_supported(int H, int W, int D);

// out[n,h,w,:] = [feat[n,h,w,:D0], linspace(-1,1,W)[w], linspace(-1,1,H)[h], 0 ...]
void launch_add_coords(const float* feat, int N, int H, int W, int D0, int D, float* out,
                       hipStream_t s);
// dl < 0: host-scheduled launch, `count` work items at `tab_off` (one workgroup each).  dl >= 0:
// device-scheduled, `count` is the size of a persistent grid and the items are
// b.tab[b.dsched[2 dl] ...), b.dsched[2 dl + 1] of them.
void launch_textmap(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                    hipStream_t s, int dl = -1);
void launch_att_ops(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                    hipStream_t s, int dl = -1);
void launch_pool(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                 hipStream_t s, int dl = -1);
void launch_heads(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                  hipStream_t s, int dl = -1);

// Layout assembler + level scheduler ON THE DEVICE (the host's assemble_tokens + schedule of
// schedule.cpp; nmn3_assembler.py:153-222): tokens [T][N] -> nodes (node id = n * T + t = its text-map
// row, pooling slot and attention-map row; conv_image row = n), the work tables of every stage of
// every level and their (offset, count) pairs.  Launch slots: 0 = text maps, 1 + 3 L + {0, 1, 2} =
// stage A / B / C of level L < levels.
struct SchedArgs {
  const int32_t* tokens;     // [T][N]
  const int32_t* token_op;   // [V] n2nmn_op of a token, < 0 = <eos>
  int T, N, V, levels;
  DevNode* nodes;            // [N * T]
  int32_t* tab;              // work tables
  int tab_cap;
  int32_t* dsched;           // [(1 + 3 levels)][2]
  int32_t* validity;         // [N] or nullptr
  int32_t* ev_rows;          // [2][ev_stride]: answer row of question n for Describe / SameProperty roots (or -1)
  int ev_stride;
  int32_t* overflow;         // set to 1 when the tables did not fit (nothing is launched wrongly: counts are 0)
};
constexpr int SCHED_MAX_T = 32;
void launch_sched(const SchedArgs& a, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// training step (exp_clevr/train_clevr_gt_layout.py:104-130): kernels_train.hip
// Every gradient writer ACCUMULATES into the zeroed flat gradient buffer.
// ---------------------------------------------------------------------------------------------
// C[m][n] += sum_r A[row(r)][m] * B[r][n]   (weight gradients: both operands have the reduction
// index as their slow axis).  fp32 MFMA, split over r across blockIdx.z, atomicAdd epilogue.
struct GemmTnArgs {
  const float* A; int lda; int M;              // m < M; M % 4 == 0; rows readable up to M
  const int32_t* a_group_idx; int a_group_size; // source row of r = group_idx[r/gs]*gs + r%gs
  const int32_t* a_onehot;                     // != nullptr: A[r][m] = (a_onehot[r] == m), A unused
  const float* B; int ldb; int N;              // n < N; rows readable up to round_up(N, 4)
  const int32_t* b_sel; int b_sel_val;         // optional: row r contributes iff b_sel[r] == val
  int R;
  // optional compaction of the reduction: row r of BOTH operands (and of a_onehot) is row_idx[r],
  // and the number of rows is read from the device (*r_dev <= R; R bounds the launch geometry)
  const int32_t* row_idx;
  const int32_t* r_dev;
  float* C; int ldc;
  float* colsum;                               // optional: colsum[n] += sum_r B[r][n] (the bias
                                               // gradient that goes with dW = X^T . dY), same rows
  // several problems of the same shape in ONE launch (the launch fills the chip with fewer splits
  // of the reduction, hence fewer atomic adds, and pays its fixed latencies once): problem p in
  // [0, nprob) uses A_p / B_p / C_p / colsum_p / bsel_p[p] instead of A / B / C / colsum /
  // b_sel_val; everything else is shared.  The same operands with bsel_p[p] = p is "several row
  // selections" (e.g. the five fc_text weight sets); distinct operands is a batch of GEMMs (e.g.
  // the three recurrent weight gradients of an LSTM stack).  nprob = 0: one problem.
  int nprob;
  const float* A_p[6];
  const float* B_p[6];
  float* C_p[6];
  float* colsum_p[6];
  int bsel_p[6];
  // set by the launcher: a launch that covers z slices [z_off, z_off + gridDim.z) of z_total
  int z_off, z_total;
};
// max_resident > 0: a background GEMM (side stream, beside a latency-bound chain on the caller's
// stream): issued as consecutive launches of at most that many workgroups, so that the chain's
// kernels always find room on every CU instead of queueing behind 1024 resident GEMM workgroups
void launch_gemm_tn(const GemmTnArgs& a, hipStream_t s, int max_resident = 0);
// rows[0 .. *count) = { t*N + n : t < seq_len[n] } in any order; count must be zero on entry
void launch_active_rows(const int32_t* seq_len, int T, int N, int32_t* rows, int32_t* count,
                        int32_t* rows_ch, const int* chunk_start, hipStream_t s);
// dst[c] += sum_r src[r*ld + c] (rows filtered by sel[r] == sel_val when sel != nullptr)
void launch_colsum(const float* src, int R, int ncols, int ld, const int32_t* sel, int sel_val,
                   float* dst, hipStream_t s);
// PK layout of the TRANSPOSE: dst <- B[k][n] = src[n*ld + k]
void launch_pack_pk_t(const float* src, int ld, int K, int N, float* dst, int Kp, int Np,
                      hipStream_t s);
// backward recurrent operand: tile j, k = 4u+g (unit u, gate g) -> W[(row0 + 16j + c)*ld + g*L + u]
// written at k offset k_off of tiles holding Ktot k's: dst [L/16][Ktot/4][16][4]
void launch_pack_tiles_t(const float* W, int ld, int row0, int L, float* dst, int Ktot, int k_off,
                         hipStream_t s);

// One reverse-time step of one LSTM layer: rec = dz_next . W^T (MFMA, K = 4L or 8L), then the cell
// backward of step t in the epilogue (Appendix A.1 differentiated; masking as dynamic_rnn A.2).
struct LstmBwdJob {
  int active;
  const float* A0;        // dz operand, k-interleaved [L][R][4] (k = 4u+g), first 4L k's
  const float* A1;        // second 4L k's (layer 0: its own dz of step t+1) or nullptr
  int K, R;
  const float* Wt;        // packed tiles [L/16][K/4][16][4]
  int gemm;               // 0: dz operands are all zero (first launch), skip the contraction
  int cell;               // 0: only produce dH (gradient of the initial state), no cell backward
  int t, T;
  const int32_t* seq_len; // [N] or nullptr
  const float4* gates;    // [N][L] saved (i,j,f,o) of step t
  const float* c_new;     // [N][L] cell state after step t
  const float* c_prev;    // [N][L] cell state before step t
  const float* dout;      // [N][L] gradient arriving at this layer's output of step t, or nullptr
  float* dH;              // [N][L] carried gradient of the hidden state (in/out)
  float* dC;              // [N][L] carried gradient of the cell state (in/out)
  float* dz_k;            // out: k-interleaved [L][R][4]
  float* dz_rm;           // out: row-major [N][4L], reference column order g*L+u
  const float* drop;      // [N][L] dropout multipliers of THIS layer's output at step t (or nullptr):
                          // the gradient arriving through A0 (the layer above) is scaled by them
  // length-sorted rows (encoder): GEMM row / dz_k row r is question perm[r] (everything indexed [N][..]
  // above is indexed by question), and only rows r < *n_act are inside their length at step t -- a row
  // block past that returns at once: its dz is zero (dz_k cleared at the start of the pass) and its
  // carried dH / dC stay as they are.  nullptr: r is the question, every row block runs.
  const int32_t* perm;
  const int32_t* n_act;
};
void launch_lstm_bwd_step(const LstmBwdJob* jobs, int njobs, int N, int L, hipStream_t s);

// idx[t*N+n] = t == 0 ? go_row : gt[(t-1)*N+n]
void launch_dec_xidx(const int32_t* gt, int Td, int N, int go_row, int32_t* idx, hipStream_t s);

struct DecBwdArgs {
  const float* scores;     // [Td][N][V] token logits of the forward pass
  const int32_t* gt;       // [Td][N]
  const float* q;          // [Td][N][L]
  const float* eht;        // [T][N][L]
  const float* eout;       // [T][N][L]
  const float* atts;       // [Td][T][N]
  const float* datts_wv;   // [Td][T][N] gradient arriving through word_vecs
  const int32_t* seq_len;  // [N]
  const int32_t* order;    // [N] questions by decreasing length (enc_prepare's perm) or nullptr: the
                           // per-question kernel takes its workgroups' questions in this order, so the
                           // dispatcher hands the longest questions out first (LPT schedule)
  const float* v;          // [L]
  const float* Wy;         // [2L][V]
  int T, N, L, V, Td;
  float inv_n;             // 1/N of the batch mean
  // d loss / d log p(chosen token) per question (nullptr: -1/N, behavioural cloning), validity
  // bits of the forward (nullptr: all valid), weight of d neg_entropy (lambda_entropy / N)
  const float* coef; const int32_t* valid_bits; float ent_coef;
  float* dsc;              // [Td][N][16] d token logits (zero padded)
  float* dout;             // [Td][N][L] direct part of d(top-layer h)
  float* dctx;             // [Td][N][L]
  float* de;               // [Td][T][N]
  float* dq;               // [Td][N][L]
  float* dvp;              // [Td*N][L] per-(t,n) partial of d v
  float* deht;             // [T][N][L]
  float* deout;            // [T][N][L]
};
void launch_dec_bwd_a(const DecBwdArgs& a, hipStream_t s);   // per (n, t)
void launch_dec_bwd_b(const DecBwdArgs& a, hipStream_t s);   // per (tau, n)

// word_vecs = sum_tau atts * emb[seq]: datts_wv [Td][T][N] and dE [T][N][E] = gradient of the
// embedded question (rows tau < len only; the embedding gradient is a one-hot gemm_tn over them)
void launch_word_vecs_bwd(const float* dwv, const float* atts, const int32_t* seq,
                          const int32_t* seq_len, const float* emb, int T_dec, int T_enc, int N,
                          int E, float* datts_wv, float* dE, hipStream_t s);

// losses[0] = mean CE(scores, labels), losses[1] = mean(-log_seq_prob); dscores = (p - onehot)/N
// policy-gradient objective (train_clevr_rl_gt_layout.py:107-129); coef[n] = (final_loss - baseline)/N
struct LossRlArgs {
  const float* scores; const int32_t* labels; const float* log_seq_prob; const float* neg_entropy;
  const int32_t* expr_validity; int N, C; float invalid_expr_loss, baseline_decay;
  float* baseline; float* dscores; float* losses; float* coef;
};
void launch_loss_rl(const LossRlArgs& a, hipStream_t s);
void launch_loss(const float* scores, const int32_t* labels, const float* log_seq_prob, int N,
                 int C, float* dscores, float* losses, hipStream_t s, float* ds_pad = nullptr,
                 int Cp = 0);      // ds_pad [N][Cp]: zero-padded copy of dscores (GEMM operand)
void launch_embed_scatter(const float* src, const int32_t* idx, const int32_t* rows,
                          const int32_t* count, int max_rows, int ncols, float* dst, hipStream_t s);
void launch_ew_mul(float* x, const float* m, size_t n, hipStream_t s);
void launch_dropout_mult(float* out, size_t n, float keep_prob, unsigned long long seed,
                         unsigned long long offset, hipStream_t s);
void launch_qpn_dpre(float* dad, const float* ad, const float* m1, size_t n, hipStream_t s);
void launch_qpn_dh_add(const float* dh, const float* mh, float* dH0, float* dH1, int N, int L,
                       hipStream_t s);

void launch_loss_total(float* losses, float wd, float lambda_entropy, hipStream_t s);
// zero up to 6 byte ranges (4-byte aligned starts and sizes) in ONE launch: a memset node costs ~5 us
// on the stream whatever its size, and the backward pass starts with four of them
struct ZeroRanges { void* ptr[6]; size_t bytes[6]; int n; };
void launch_zero_ranges(const ZeroRanges& z, hipStream_t s);

struct ModuleGrads {
  float* garena;          // [max_nodes][HWp]   d loss / d attention map of a node
  float* dtmap;           // [max_text][Mp]     (zeroed; Find-type parts add atomically)
  float* dpfc;            // [max_pool][2][Mp]  d fc_att output (zeroed)
  float* gda;             // [max_pool][2][HWq] d softmax weights of the pools (zeroed; parts add)
  float* dmfind;          // [N][HW][Mp]        d conv_image map, FindModule weights (zeroed)
  float* dmfsp;
  const float* dscores;   // [rows][C]
  float* dwv;             // [T_dec][N_full][E] (zeroed)
  // destinations inside the flat gradient buffer (reference layouts)
  float* gwe[3]; float* gbe[3];        // conv_eltwise of Find / FSP / Transform
  float* gKt; float* gbt;              // Transform conv_maps
  float* gbatt[4];                     // fc_att biases: FSP, SameProperty(0,1), Describe
  float* gWans[7]; float* gbans[7];    // answer FCs (order of ModuleWeights::Wans)
  // large answer vocabulary (map_dim * num_choices beyond the fused head, models_vqa): fc_eltwise's
  // backward runs as GEMMs over the whole batch.  hb_den [rows][Mp] = dscores . W_e^T is computed
  // before the level loop; heads_bwd reads its row, writes the normalised product it recomputed
  // into hb_en [rows][Mp] and marks the row in hb_sel; dW_e = hb_en^T . dscores follows the loop.
  const float* hb_den; float* hb_en; int32_t* hb_sel;
};
void launch_heads_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                      int tab_off, int count, hipStream_t s);
// tab entries of a node are `stride` int32 apart (forward pooling table: 2*POOL_PARTS)
void launch_pool_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                     int tab_off, int count, int stride, hipStream_t s);
void launch_att_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                    int tab_off, int count, hipStream_t s);
void launch_textmap_bwd(const ModuleWeights& w, const ModuleBuffers& b, const ModuleGrads& g,
                        int tab_off, int count, hipStream_t s);

// Optimiser.  Segment table: seg[i] = {var, begin, end} over the flat parameter vector.
struct ParamSeg { int32_t var; int32_t pad; int64_t begin, end; };
// g[i] = g[i]*scale + wd*w[i] (only where decay[var] != 0); l2 += 0.5*w^2 over decayed vars
void launch_grad_finish(float* grads, const float* const* mirrors, const int64_t* var_off,
                        const int32_t* decay, const ParamSeg* segs, int nsegs, float scale, float wd,
                        float* l2_out, hipStream_t s);
void launch_grad_sqnorm(const float* grads, const ParamSeg* segs, int nsegs, float scale,
                        float* norm2, hipStream_t s);
// per-tensor tf.clip_by_norm then Adam (TF 1.0.0 formula) on the mirrors
void launch_adam(const float* grads, float* const* mirrors, const int64_t* var_off,
                 const ParamSeg* segs, int nsegs, const float* norm2, float scale, float clip,
                 float lr_t, float beta1, float beta2, float eps, float* m, float* v,
                 hipStream_t s);

}  // namespace n2nmn
