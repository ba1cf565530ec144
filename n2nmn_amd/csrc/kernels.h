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
  const float* Bp; int Np; int Kp;
  const float* bias; int N;
  float* C; int ldc; int n_store;
};
void launch_gemm_pk(const GemmArgs& a, hipStream_t s);

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
  int ntiles;             // number of 16-column tiles (LSTM: L/4; linear: Ncols/16)
  int mode;               // 0: LSTM cell epilogue; 1: plain linear  out = z + bias
  const float* xtab;      // [V][4L] input-projection table incl. bias (layer 0) or nullptr
  const int32_t* xidx;    // [N] row of xtab per batch row (layer 0) or nullptr
  int xidx_const;         // used when xidx == nullptr && xtab != nullptr (go embedding row)
  const float* bias;      // [4L] (layer 1) or nullptr
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
};
// rows_per_wg: 64 (4 M-tiles per workgroup) or 32 (2 M-tiles; doubles the workgroups of a launch)
void launch_lstm_step(const LstmJob* jobs, int njobs, int N, int L, int rows_per_wg,
                      hipStream_t s);

// Arguments of dec_attn_kernel.  Every per-step pointer is the slice of the FIRST step of the
// launch; workgroup (n, ts) addresses element ts*N + n of it.
// k-interleaved state [L/4][R][4] -> row-major [N][L]
void launch_unpack_h(const float* src, float* dst, int N, int L, int R, hipStream_t s);
void launch_lstm_step_dbg(const LstmJob* jobs, int njobs, int N, int L, int rows_per_wg,
                          int variant, hipStream_t s);

struct DecStepArgs {
  // inputs
  const float* q;          // [steps][N][L]   out . W_a + b_a
  const float* out;        // [steps][N][L]   top-layer h
  const float* eht;        // [T][N][L]
  const float* eout;       // [T][N][L]
  const int32_t* seq_len;  // [N]
  const float* v;          // [L]
  const float* Wy;         // [2L][V]
  const float* by;         // [V]
  const int32_t* P;        // [V][3]
  const int32_t* Wv;       // [3][V][4]
  const int32_t* bv;       // [V][4]
  const int32_t* gt;       // [steps][N] teacher-forcing tokens or nullptr
  const float* uni;        // [steps][N] uniforms or nullptr
  const int32_t* forced;   // [steps][N] or nullptr
  int use_gt;
  int T, N, L, V;
  // state / outputs
  int32_t* state;          // [N][3]   (sequential decoding only)
  int32_t* tokens;         // [steps][N]
  float* tprobs;           // [steps][N]
  float* ent_t;            // [steps][N] per-step entropy terms (summed by word_vecs_kernel)
  float* atts;             // [steps][T][N]
  float* scores;           // [steps][N][V] or nullptr
  int32_t* next_idx;       // [N] row of the decoder x-table for the next step (= token) or nullptr
};
// nsteps == 1: one sequential step (1024-thread workgroups); nsteps > 1: all steps in one launch
void launch_dec_attn(const DecStepArgs& a, int nsteps, hipStream_t s);

void launch_dec_init(int32_t* state, int N, int T_dec, hipStream_t s);
// perm = rows sorted by decreasing length (stable); n_active[t] = #{n : seq_len[n] > t}, t < T
void launch_enc_prepare(const int32_t* seq_len, int N, int T, int32_t* perm, int32_t* n_active,
                        hipStream_t s);

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
};

void launch_textmap(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                    hipStream_t s);
void launch_att_ops(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                    hipStream_t s);
void launch_pool(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                 hipStream_t s);
void launch_heads(const ModuleWeights& w, const ModuleBuffers& b, int tab_off, int count,
                  hipStream_t s);

}  // namespace n2nmn
