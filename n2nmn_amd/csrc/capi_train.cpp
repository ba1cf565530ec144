// C ABI of the training step (include/n2nmn.h section 6): forward of the behavioural-cloning
// objective with activations kept, backward in two phases (module network + decoder, then encoder)
// into ONE caller-owned flat gradient buffer, per-tensor clip + Adam.
// Reference: exp_clevr/train_clevr_gt_layout.py:104-130, models_clevr/nmn3_model.py:46,161-166.
// Host code only; kernels live in kernels_train.hip / kernels_train_modules.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "ctx.h"

namespace n2nmn {

struct TrainState {
  char* base = nullptr;
  size_t bytes = 0;
  TrainRec rec;
  // forward results kept inside the context
  float* scores = nullptr;            // [N][C] (when the caller passes no buffer)
  // backward scratch
  float *rl_coef = nullptr;           // policy gradient: d total_loss / d log_seq_prob per question
  float *dscores = nullptr, *dsc = nullptr, *garena = nullptr, *dtmap = nullptr, *dpfc = nullptr, *gda = nullptr,
        *dmfind = nullptr, *dmfsp = nullptr, *dwv = nullptr, *datts_wv = nullptr, *dE = nullptr, *de = nullptr,
        *dctx = nullptr, *dq = nullptr, *dout = nullptr, *dvp = nullptr, *deht = nullptr,
        *denc_out = nullptr;
  float *dz0_all = nullptr, *dz1_all = nullptr;     // encoder: dz of every step, row-major
  // the decoder's own (its weight-gradient GEMMs read them on the side stream while the encoder's
  // reverse-time pass is already writing dz0_all / dz1_all)
  float *ddz0_all = nullptr, *ddz1_all = nullptr;
  // models_vqa: large answer head and question prior net as batch GEMMs
  int Cp = 0;                         // num_choices rounded up to 4 (row stride of ds_pad)
  float *ds_pad = nullptr;            // [N][Cp] zero-padded dscores
  float *hb_den = nullptr, *hb_en = nullptr;   // [N][Mp]
  int32_t* hb_sel = nullptr;          // [N] 1 where the row has a Describe head
  float *qpn_dad = nullptr, *qpn_dh = nullptr; // [N][Hq], [N][2L]
  float *wde_T_p = nullptr, *qpn_W2T_p = nullptr, *qpn_W1T_p = nullptr;   // transposed packs
  float *dzk0[2] = {nullptr, nullptr}, *dzk1[2] = {nullptr, nullptr};
  float *dH0 = nullptr, *dH1 = nullptr, *dC0 = nullptr, *dC1 = nullptr;
  char *zero_begin = nullptr, *zero_end = nullptr;   // dtmap..act_count: one memset per step
  float *dxtab_enc = nullptr, *dxtab_dec = nullptr;
  int32_t* dec_xidx = nullptr;
  int32_t *act_rows = nullptr, *act_count = nullptr;   // encoder rows (t, n) with t < len[n]
  // the same rows per time chunk (list ci at act_rows_ch + chunk_start[ci]*N, length act_count[1+ci]):
  // the encoder's weight-gradient GEMMs follow its reverse-time recurrence chunk by chunk
  int32_t* act_rows_ch = nullptr;
  int chunk_start[4] = {0, 0, 0, 0};
  bool deferred = false;              // phase 0 left its leaves on the side stream (n2nmn_train_join)
  // transposed operand packs (rebuilt after every weight commit)
  float *enc_Wt1 = nullptr, *enc_Wt0 = nullptr, *dec_Wt1 = nullptr, *dec_Wt0 = nullptr;
  float *eht_WT_p = nullptr, *att_WT_p = nullptr, *enc_W0xT_p = nullptr, *dec_W0xT_p = nullptr;
  uint64_t pack_epoch = 0;
  PackBatch packs;                    // the transposed packs as one launch
  // per-program tables (text slot -> word_vecs row / weight set; pooling slot -> fc_att set)
  int32_t *tslot_row = nullptr, *tslot_ws = nullptr, *pool_sel = nullptr;
  int32_t* tab_host = nullptr;        // pinned staging for the three tables
  hipEvent_t tab_ev = nullptr;
  // optimiser
  float *m = nullptr, *v = nullptr, *norm2 = nullptr;
  float** mirrors_dev = nullptr;
  int64_t* var_off_dev = nullptr;
  int32_t* decay_dev = nullptr;
  ParamSeg* segs_dev = nullptr;
  int nsegs = 0, nsegs_early = 0;     // segments of the encoder variables come first
  std::vector<int64_t> var_off;
  int64_t total = 0, split = 0;
  int KpL4 = 0;                       // round_up(4L, 32)
  int last_N = 0, last_T = 0, last_Td = 0;   // shape of the forward currently held
  // Weight-gradient GEMMs are leaves of the backward graph: they run on a low-priority side
  // stream while the latency-bound chains (BPTT, attention backward) continue on the caller's.
  hipStream_t side = nullptr;
  static constexpr int kForkEvents = 16;
  hipEvent_t ev_fork[kForkEvents] = {};
  hipEvent_t ev_join = nullptr;
  hipEvent_t infer_ev = nullptr;      // after the last commit's inference-only operand refresh
  bool infer_pending = false;
  int fork_i = 0;
  bool overlap = true;
  // side stream runs everything enqueued on `main` so far before its next kernel
  hipStream_t fork(hipStream_t main) {
    if (!overlap || !side) return main;
    hipEvent_t e = ev_fork[fork_i];
    fork_i = (fork_i + 1) % kForkEvents;
    (void)hipEventRecord(e, main);
    (void)hipStreamWaitEvent(side, e, 0);
    return side;
  }
  // workgroups a background GEMM may keep resident (launch_gemm_tn's max_resident): 3 per CU leave
  // room for a whole workgroup of whatever the latency-bound chain launches next
  int bg_wgs = 768;
  // 1: the recurrences' weight gradients follow them on the side stream (decoder: after its last
  // step, under the encoder's pass; encoder: chunk by chunk); 0: on the caller's stream after each
  // recurrence (round 2's schedule).  n2nmn_debug_set "train_schedule"
  int schedule = 1;
  int chunk_pct[3] = {33, 0, 0};
  int bg(hipStream_t s) const { return overlap && side && s == side ? bg_wgs : 0; }
  // `main` waits for everything enqueued on the side stream so far
  void join(hipStream_t main) {
    if (!overlap || !side) return;
    (void)hipEventRecord(ev_join, side);
    (void)hipStreamWaitEvent(main, ev_join, 0);
  }
};

void train_side_join(n2nmn_ctx* c, hipStream_t waiter) {
  if (c && c->train) c->train->join(waiter);
}

hipStream_t train_infer_fork(n2nmn_ctx* c, hipStream_t s) {
  return c->train && c->train->infer_ev ? c->train->fork(s) : s;
}

void train_infer_done(n2nmn_ctx* c, hipStream_t side) {
  TrainState* t = c->train;
  if (!t || !t->infer_ev || !t->overlap || side != t->side) return;
  (void)hipEventRecord(t->infer_ev, side);
  t->infer_pending = true;
}

void train_infer_wait(const n2nmn_ctx* root, hipStream_t s) {
  const TrainState* t = root->train;
  if (t && t->infer_pending) (void)hipStreamWaitEvent(s, t->infer_ev, 0);
}

void train_infer_host_wait(const n2nmn_ctx* root) {
  const TrainState* t = root->train;
  if (t && t->infer_pending) (void)hipEventSynchronize(t->infer_ev);
}

void train_state_destroy(TrainState* t) {
  if (!t) return;
  if (t->base) (void)hipFree(t->base);
  if (t->tab_host) (void)hipHostFree(t->tab_host);
  if (t->tab_ev) (void)hipEventDestroy(t->tab_ev);
  for (hipEvent_t e : t->ev_fork) if (e) (void)hipEventDestroy(e);
  if (t->ev_join) (void)hipEventDestroy(t->ev_join);
  if (t->infer_ev) (void)hipEventDestroy(t->infer_ev);
  if (t->side) (void)hipStreamDestroy(t->side);
  delete t;
}

namespace {

constexpr int SEG_ELEMS = 16384;

size_t carve_train(n2nmn_ctx* c, TrainState* t, char* base) {
  const n2nmn_dims& d = c->d;
  const size_t L = d.lstm_dim, E = d.embed_dim_txt, N = d.N, T = d.T_encoder, Td = d.T_decoder,
               V = d.num_vocab_nmn, Vt = d.num_vocab_txt, D = d.D, HW = (size_t)d.H * d.W,
               C = d.num_choices;
  const size_t Mp = c->Mp, HWp = c->HWp;
  Carver k(base);
  TrainRec& r = t->rec;
  r.eg0 = k.take<float4>(T * N * L); r.eg1 = k.take<float4>(T * N * L);
  r.dg0 = k.take<float4>(Td * N * L); r.dg1 = k.take<float4>(Td * N * L);
  r.ec0s = k.take<float>((T + 1) * N * L); r.ec1s = k.take<float>((T + 1) * N * L);
  r.eh0s = k.take<float>((T + 1) * N * L); r.eh1s = k.take<float>((T + 1) * N * L);
  r.dc0s = k.take<float>((Td + 1) * N * L); r.dc1s = k.take<float>((Td + 1) * N * L);
  r.dh0s = k.take<float>((Td + 1) * N * L); r.dh1s = k.take<float>((Td + 1) * N * L);
  r.ctx = k.take<float>(Td * N * L);
  if (d.variant == N2NMN_VARIANT_VQA) {      // dropout on LSTM layer 0's output (models_vqa training)
    r.eh0d = k.take<float>(T * N * L);
    r.dh0d = k.take<float>(Td * N * L);
  }
  r.tscores = k.take<float>(Td * N * V);
  r.lsp = k.take<float>(N);
  r.pooled = k.take<float>((size_t)c->max_pool * 2 * D);
  t->scores = k.take<float>(N * C);
  t->dscores = k.take<float>(N * C + 4);
  t->rl_coef = k.take<float>(N);
  r.valid_bits = k.take<int32_t>(Td * N);
  t->dsc = k.take<float>(Td * N * 16);
  // [zero block: cleared by ONE memset at the start of backward phase 0]
  t->zero_begin = reinterpret_cast<char*>(k.take<float>(0));
  t->garena = k.take<float>((size_t)c->max_nodes * HWp);     // Transform adds into its input's row
  t->dtmap = k.take<float>((size_t)c->max_text * Mp);
  t->dpfc = k.take<float>((size_t)c->max_pool * 2 * Mp);
  t->gda = k.take<float>((size_t)c->max_pool * 2 * ((HW + 3) & ~3));
  t->dwv = k.take<float>(Td * N * E);
  t->dH0 = k.take<float>(N * L); t->dH1 = k.take<float>(N * L);
  t->dC0 = k.take<float>(N * L); t->dC1 = k.take<float>(N * L);
  t->dxtab_enc = c->big_vocab ? nullptr : k.take<float>(Vt * 4 * L);
  t->dxtab_dec = k.take<float>((V + 1) * 4 * L);
  t->act_count = k.take<int32_t>(8);
  if (c->big_heads) {
    t->hb_en = k.take<float>(N * Mp);
    t->hb_sel = k.take<int32_t>(N);
  }
  t->zero_end = reinterpret_cast<char*>(k.take<float>(0));
  t->Cp = (int)((C + 3) & ~(size_t)3);
  if (c->big_heads) {
    t->ds_pad = k.take<float>(N * (size_t)t->Cp);
    t->hb_den = k.take<float>(N * Mp);
    t->wde_T_p = k.take<float>((size_t)round_up((int)C, 32) * round_up(d.map_dim, 64));
  }
  if (c->qpn_h) {
    const size_t Hq = d.qpn_hidden;
    if (!t->ds_pad) t->ds_pad = k.take<float>(N * (size_t)t->Cp);
    t->qpn_dad = k.take<float>(N * Hq);
    t->qpn_dh = k.take<float>(N * 2 * L);
    t->qpn_W2T_p = k.take<float>((size_t)round_up((int)C, 32) * round_up((int)Hq, 64));
    t->qpn_W1T_p = k.take<float>((size_t)round_up((int)Hq, 32) * round_up((int)(2 * L), 64));
  }
  t->dmfind = k.take<float>(N * HW * Mp);
  t->dmfsp = k.take<float>(N * HW * Mp);
  t->datts_wv = k.take<float>(Td * T * N);
  t->dE = k.take<float>(T * N * E);
  t->de = k.take<float>(Td * T * N);
  t->dctx = k.take<float>(Td * N * L);
  t->dq = k.take<float>(Td * N * L);
  t->dout = k.take<float>(Td * N * L);
  t->dvp = k.take<float>(Td * N * L);
  t->deht = k.take<float>(T * N * L);
  t->denc_out = k.take<float>(T * N * L);
  t->dz0_all = k.take<float>(T * N * 4 * L);
  t->dz1_all = k.take<float>(T * N * 4 * L);
  t->ddz0_all = k.take<float>(Td * N * 4 * L);
  t->ddz1_all = k.take<float>(Td * N * 4 * L);
  for (int i = 0; i < 2; ++i) {
    t->dzk0[i] = k.take<float>(4 * L * N);
    t->dzk1[i] = k.take<float>(4 * L * N);
  }
  t->dec_xidx = k.take<int32_t>(Td * N);
  t->act_rows = k.take<int32_t>(T * N);
  t->act_rows_ch = k.take<int32_t>(T * N);
  t->enc_Wt1 = k.take<float>(L * 4 * L); t->enc_Wt0 = k.take<float>(L * 8 * L);
  t->dec_Wt1 = k.take<float>(L * 4 * L); t->dec_Wt0 = k.take<float>(L * 8 * L);
  t->eht_WT_p = k.take<float>((size_t)c->KpL * L);
  t->att_WT_p = k.take<float>((size_t)c->KpL * L);
  const size_t Ep = round_up((int)E, 64);
  t->enc_W0xT_p = k.take<float>((size_t)t->KpL4 * Ep);
  t->dec_W0xT_p = k.take<float>((size_t)t->KpL4 * Ep);
  t->tslot_row = k.take<int32_t>(2 * (size_t)c->max_text + 2 * (size_t)c->max_pool);   // one block,
  t->tslot_ws = t->tslot_row + c->max_text;                  // laid out like the pinned tab_host
  t->pool_sel = t->tslot_ws + c->max_text;
  t->m = k.take<float>((size_t)t->total);
  t->v = k.take<float>((size_t)t->total);
  t->norm2 = k.take<float>(V_COUNT_);
  t->mirrors_dev = k.take<float*>(V_COUNT_);
  t->var_off_dev = k.take<int64_t>(V_COUNT_);
  t->decay_dev = k.take<int32_t>(V_COUNT_);
  t->segs_dev = k.take<ParamSeg>(t->nsegs);
  t->packs.dev = k.take<PackJob>(16);
  return align_up(k.off, 256);
}

bool ends_with(const std::string& s, const char* suf) {
  const size_t n = std::strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// operand packs of the backward GEMMs: W^T views of the committed weights
int repack_transposed(n2nmn_ctx* c, hipStream_t s) {
  TrainState* t = c->train;
  const n2nmn_dims& d = c->d;
  const int L = d.lstm_dim, E = d.embed_dim_txt;
  auto m = [&](int id) { return c->vars[id].mirror; };
  if (!t->packs.uploaded) {
    PackBatch& pb = t->packs;
    // layer 1: rec = dz1 . W1[L:2L, :]^T ; layer 0: rec = [dz1 ; dz0] . [W1[0:L, :] ; W0[E:E+L, :]]^T
    pb.tiles_t(m(V_ENC_W1), 4 * L, L, L, t->enc_Wt1, 4 * L, 0);
    pb.tiles_t(m(V_ENC_W1), 4 * L, 0, L, t->enc_Wt0, 8 * L, 0);
    pb.tiles_t(m(V_ENC_W0), 4 * L, E, L, t->enc_Wt0, 8 * L, 4 * L);
    pb.tiles_t(m(V_DEC_W1), 4 * L, L, L, t->dec_Wt1, 4 * L, 0);
    pb.tiles_t(m(V_DEC_W1), 4 * L, 0, L, t->dec_Wt0, 8 * L, 0);
    pb.tiles_t(m(V_DEC_W0), 4 * L, E, L, t->dec_Wt0, 8 * L, 4 * L);
    pb.pk_t(m(V_EHT_W), L, L, L, t->eht_WT_p, c->KpL, L);
    pb.pk_t(m(V_ATT_W), L, L, L, t->att_WT_p, c->KpL, L);
    const int Cc = d.num_choices, Hq = d.qpn_hidden;
    if (t->wde_T_p)        // den = dscores . W_e^T: B'[k = class][n = map channel]
      pb.pk_t(m(V_DE_E_W), Cc, Cc, d.map_dim, t->wde_T_p, round_up(Cc, 32), round_up(d.map_dim, 64));
    if (t->qpn_W2T_p) {
      pb.pk_t(m(V_QPN_W2), Cc, Cc, Hq, t->qpn_W2T_p, round_up(Cc, 32), round_up(Hq, 64));
      pb.pk_t(m(V_QPN_W1), Hq, Hq, 2 * L, t->qpn_W1T_p, round_up(Hq, 32), round_up(2 * L, 64));
    }
    const int Ep = round_up(E, 64);
    // B[k][n] = W0[n][k], k < 4L (gate pre-activations), n < E (embedding dims)
    pb.pk_t(m(V_ENC_W0), 4 * L, 4 * L, E, t->enc_W0xT_p, t->KpL4, Ep);
    pb.pk_t(m(V_DEC_W0), 4 * L, 4 * L, E, t->dec_W0xT_p, t->KpL4, Ep);
    N2_HIP(hipMemcpy(pb.dev, pb.jobs.data(), sizeof(PackJob) * pb.jobs.size(), hipMemcpyHostToDevice));
    pb.uploaded = true;
  }
  launch_pack_jobs(t->packs.dev, (int)t->packs.jobs.size(), t->packs.blocks, s);
  t->pack_epoch = c->commit_epoch;
  return check_launch("train: repack_transposed");
}

float* gptr(const n2nmn_ctx* c, const n2nmn_train_io* io, int var) {
  return io->grads + c->train->var_off[var];
}

int gemm_tn(n2nmn_ctx* c, hipStream_t s, const float* A, int lda, int M, const float* B, int ldb,
            int N, int R, float* C, int ldc, const int32_t* a_idx = nullptr, int a_gs = 1,
            const int32_t* b_sel = nullptr, int b_val = 0, const int32_t* row_idx = nullptr,
            const int32_t* r_dev = nullptr, float* colsum_dst = nullptr) {
  if (R <= 0) return N2NMN_OK;
  GemmTnArgs g{};
  g.row_idx = row_idx; g.r_dev = r_dev; g.colsum = colsum_dst;
  g.A = A; g.lda = lda; g.M = M; g.a_group_idx = a_idx; g.a_group_size = a_gs;
  g.B = B; g.ldb = ldb; g.N = N; g.b_sel = b_sel; g.b_sel_val = b_val; g.R = R; g.C = C; g.ldc = ldc;
  ProfScope ps(c, F_GEMM_TN, 2.0 * M * N * R, 4.0 * ((double)R * (M + N) + (double)M * N), s);
  launch_gemm_tn(g, s, c->train->bg(s));
  return N2NMN_OK;
}

// nprob weight-gradient GEMMs of one shape (M, N, R, strides, row list) in one launch
struct TnProblem { const float* A; const float* B; float* C; float* colsum; };
void gemm_tn_batch(n2nmn_ctx* c, hipStream_t s, int nprob, const TnProblem* pr, int lda, int M,
                   int ldb, int N, int R, int ldc, const int32_t* row_idx = nullptr,
                   const int32_t* r_dev = nullptr) {
  if (R <= 0 || nprob <= 0) return;
  GemmTnArgs g{};
  g.row_idx = row_idx; g.r_dev = r_dev;
  g.lda = lda; g.M = M; g.a_group_size = 1; g.ldb = ldb; g.N = N; g.R = R; g.ldc = ldc;
  g.nprob = nprob;
  for (int i = 0; i < nprob; ++i) {
    g.A_p[i] = pr[i].A; g.B_p[i] = pr[i].B; g.C_p[i] = pr[i].C; g.colsum_p[i] = pr[i].colsum;
    g.bsel_p[i] = 0;
  }
  g.A = pr[0].A; g.B = pr[0].B;
  ProfScope ps(c, F_GEMM_TN, 2.0 * M * N * R * nprob,
               4.0 * nprob * ((double)R * (M + N) + (double)M * N), s);
  launch_gemm_tn(g, s, c->train->bg(s));
}

// C (+)= A[M,K] . Bp   (NT GEMMs of the backward pass go through the forward gemm_pk kernel)
void gemm_nt(n2nmn_ctx* c, hipStream_t s, const float* A, int lda, int M, int K, const float* Bp,
             int Np, int Kp, int N, float* C, int ldc, bool accumulate) {
  if (M <= 0) return;
  GemmArgs g{};
  g.A = A; g.lda = lda; g.M = M; g.K = K; g.group_size = 1; g.Bp = Bp; g.Np = Np; g.Kp = Kp;
  g.bias = nullptr; g.N = N; g.C = C; g.ldc = ldc; g.n_store = N; g.accumulate = accumulate ? 1 : 0;
  // skinny outputs (a few 64x64 tiles) with a long K: split K so the launch covers the chip
  const int tiles = ((N + 63) / 64) * ((M + 63) / 64), nkt = Kp / 32;
  if (accumulate && tiles < 64 && nkt >= 16) g.ksplit = std::min(nkt / 4, std::max(1, 256 / tiles));
  ProfScope ps(c, F_BWD_MISC, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)K * N + 2.0 * M * N), s);
  launch_gemm_pk(g, s);
}

void colsum(n2nmn_ctx* c, hipStream_t s, const float* src, int R, int ncols, int ld, float* dst,
            const int32_t* sel = nullptr, int val = 0) {
  if (R <= 0) return;
  ProfScope ps(c, F_BWD_MISC, (double)R * ncols, 4.0 * R * ncols, s);
  launch_colsum(src, R, ncols, ld, sel, val, dst, s);
}

// reverse-time recurrence of a 2-layer LSTM stack; layer 1 runs one launch ahead of layer 0
struct BpttArgs {
  int T, N;
  bool want_init_grad;          // also produce dH of the initial state (decoder)
  const int32_t* seq_len;
  const float4 *g0, *g1;        // gates [T][N][L]
  const float *c0s, *c1s;       // [(T+1)][N][L]
  const float* dout;            // [T][N][L] gradient arriving at the top layer's outputs
  const float *Wt0, *Wt1;
  const float* drop0;           // [T][N][L] dropout multipliers of layer 0's output, or nullptr
  float *dz0_all, *dz1_all;     // [T][N][4L] out: dz of every step (operands of the weight gradients)
  // length-sorted pass (encoder): the forward's row ranking and per-step active-row counts
  const int32_t *perm, *nact;   // [N], [T], or nullptr
};

// after_step(t0): called after the launch that completes step t0 of BOTH layers (dz0 rows of t >= t0
// and dz1 rows of t >= t0 - 1 are then enqueued on s)
int run_bptt(n2nmn_ctx* c, const BpttArgs& a, hipStream_t s,
             const std::function<void(int)>* after_step = nullptr) {
  TrainState* t = c->train;
  const int L = c->d.lstm_dim, R = c->d.N, N = a.N;
  const size_t nl = (size_t)N * L;
  if (a.perm) {
    // row blocks outside their lengths never write their dz: all four ping-pong buffers start at zero
    // (they are carved back to back: dzk0[0], dzk1[0], dzk0[1], dzk1[1])
    char* b = reinterpret_cast<char*>(t->dzk0[0]);
    char* e = reinterpret_cast<char*>(t->dzk1[1] + 4 * (size_t)L * R);
    N2_HIP(hipMemsetAsync(b, 0, (size_t)(e - b), s));
  } else {
    N2_HIP(hipMemsetAsync(t->dzk0[0], 0, sizeof(float) * 4 * (size_t)L * R, s));
  }
  const int last = a.T + (a.want_init_grad ? 1 : 0);
  for (int k = 0; k <= last; ++k) {
    LstmBwdJob jobs[2];
    const int t1 = a.T - 1 - k, t0 = t1 + 1;
    LstmBwdJob& j1 = jobs[0];
    j1 = LstmBwdJob{};
    j1.active = (t1 >= 0) || (t1 == -1 && a.want_init_grad);
    j1.A0 = t->dzk1[(k + 1) & 1]; j1.A1 = nullptr; j1.K = 4 * L; j1.R = R; j1.Wt = a.Wt1;
    j1.gemm = k > 0; j1.cell = t1 >= 0; j1.t = t1; j1.T = a.T; j1.seq_len = a.seq_len;
    j1.dH = t->dH1; j1.dC = t->dC1; j1.dz_k = t->dzk1[k & 1];
    if (t1 >= 0) {
      j1.gates = a.g1 + (size_t)t1 * nl; j1.c_new = a.c1s + (size_t)(t1 + 1) * nl;
      j1.c_prev = a.c1s + (size_t)t1 * nl; j1.dout = a.dout + (size_t)t1 * nl;
      j1.dz_rm = a.dz1_all + (size_t)t1 * N * 4 * L;
      if (a.perm) { j1.perm = a.perm; j1.n_act = a.nact + t1; }
    }
    LstmBwdJob& j0 = jobs[1];
    j0 = LstmBwdJob{};
    j0.active = k >= 1 && ((t0 >= 0) || (t0 == -1 && a.want_init_grad));
    j0.A0 = t->dzk1[(k + 1) & 1]; j0.A1 = t->dzk0[(k + 1) & 1]; j0.K = 8 * L; j0.R = R;
    j0.Wt = a.Wt0; j0.gemm = 1; j0.cell = t0 >= 0; j0.t = t0; j0.T = a.T; j0.seq_len = a.seq_len;
    j0.dH = t->dH0; j0.dC = t->dC0; j0.dz_k = t->dzk0[k & 1];
    if (t0 >= 0 && t0 < a.T) {
      j0.gates = a.g0 + (size_t)t0 * nl; j0.c_new = a.c0s + (size_t)(t0 + 1) * nl;
      j0.c_prev = a.c0s + (size_t)t0 * nl; j0.dout = nullptr;
      j0.dz_rm = a.dz0_all + (size_t)t0 * N * 4 * L;
      if (a.drop0) j0.drop = a.drop0 + (size_t)t0 * nl;
      if (a.perm) { j0.perm = a.perm; j0.n_act = a.nact + t0; }
    }
    const double fl = 2.0 * N * L * ((j1.active && j1.gemm ? 4.0 * L : 0) + (j0.active ? 8.0 * L : 0));
    const double by = 4.0 * ((j1.active && j1.gemm ? 4.0 * L * L + 4.0 * N * L : 0) +
                             (j0.active ? 8.0 * L * L + 8.0 * N * L : 0) + 20.0 * N * L);
    {
      ProfScope ps(c, F_LSTM_BWD, fl, by, s);
      // (the contraction split over workgroups as well was measured and rejected in round 6: 15.7 us per launch
      // against 15.0, tools/rejected/lstm_bwd_step_ksplit.hip.txt)
      launch_lstm_bwd_step(jobs, 2, N, L, s);
    }
    if (after_step && k >= 1 && t0 >= 0 && t0 < a.T) (*after_step)(t0);
  }
  return check_launch("train: bptt");
}

}  // namespace
}  // namespace n2nmn

using namespace n2nmn;

extern "C" {

int n2nmn_train_enable(n2nmn_ctx* c) {
  N2_REQUIRE(c, N2NMN_EINVAL, "train_enable: null context");
  N2_REQUIRE(!c->parent, N2NMN_EINVAL, "train_enable: train on the root context");
  const bool vqa_variant = c->d.variant == N2NMN_VARIANT_VQA;
  N2_REQUIRE(vqa_variant || (c->d.H * c->d.W + TRANSFORM_PARTS - 1) / TRANSFORM_PARTS <= 64, N2NMN_EINVAL,
             "train_enable: the Transform backward handles at most 64 pixels per part (H*W <= 192)");
  N2_REQUIRE(!c->big_heads || (vqa_variant && c->d.map_dim % 4 == 0), N2NMN_EINVAL,
             "train_enable: map_dim * num_choices beyond the fused answer head is built for models_vqa "
             "(map_dim a multiple of 4)");
  N2_REQUIRE(!c->qpn_h || (c->d.qpn_hidden % 4 == 0 && c->d.lstm_dim % 2 == 0), N2NMN_EINVAL,
             "train_enable: the question prior net's weight-gradient GEMMs need qpn_hidden % 4 == 0");
  if (c->train) return N2NMN_OK;
  N2_REQUIRE(c->d.num_vocab_nmn <= 15, N2NMN_EINVAL,
             "train_enable: num_vocab_nmn + <go> must fit 16 x-table rows");
  N2_HIP(hipSetDevice(c->device));
  TrainState* t = new (std::nothrow) TrainState();
  N2_REQUIRE(t, N2NMN_EINVAL, "train_enable: out of host memory");
  t->KpL4 = round_up(4 * c->d.lstm_dim, 32);
  // flat parameter vector: variables in registration order, contiguous
  t->var_off.resize(c->vars.size());
  int64_t off = 0;
  std::vector<ParamSeg> segs;
  for (size_t i = 0; i < c->vars.size(); ++i) {
    if ((int)i == V_DEC_EMB) { t->split = off; t->nsegs_early = (int)segs.size(); }
    t->var_off[i] = off;
    const int64_t n = (int64_t)c->vars[i].numel;
    for (int64_t b = 0; b < n; b += SEG_ELEMS) {
      ParamSeg sg{};
      sg.var = (int32_t)i; sg.begin = off + b; sg.end = off + std::min<int64_t>(n, b + SEG_ELEMS);
      segs.push_back(sg);
    }
    off += n;
  }
  t->total = off;
  t->nsegs = (int)segs.size();
  c->train = t;
  t->bytes = carve_train(c, t, nullptr);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->base), t->bytes);
  if (e != hipSuccess) {
    set_last_error(std::string("train_enable: hipMalloc of ") + std::to_string(t->bytes) +
                   " bytes failed: " + hipGetErrorString(e));
    c->train = nullptr;
    delete t;
    return N2NMN_EHIP;
  }
  carve_train(c, t, t->base);
  N2_HIP(hipMemset(t->base, 0, t->bytes));
  std::vector<float*> mir(c->vars.size());
  std::vector<int32_t> decay(c->vars.size());
  for (size_t i = 0; i < c->vars.size(); ++i) {
    mir[i] = c->vars[i].mirror;
    decay[i] = ends_with(c->vars[i].name, "weights") ? 1 : 0;      // nmn3_model.py:163-165
  }
  N2_HIP(hipMemcpy(t->mirrors_dev, mir.data(), sizeof(float*) * mir.size(), hipMemcpyHostToDevice));
  N2_HIP(hipMemcpy(t->var_off_dev, t->var_off.data(), sizeof(int64_t) * t->var_off.size(),
                   hipMemcpyHostToDevice));
  N2_HIP(hipMemcpy(t->decay_dev, decay.data(), sizeof(int32_t) * decay.size(), hipMemcpyHostToDevice));
  N2_HIP(hipMemcpy(t->segs_dev, segs.data(), sizeof(ParamSeg) * segs.size(), hipMemcpyHostToDevice));
  const size_t tabn = 2 * (size_t)c->max_text + 2 * (size_t)c->max_pool;
  N2_HIP(hipHostMalloc(reinterpret_cast<void**>(&t->tab_host), sizeof(int32_t) * tabn, 0));
  N2_HIP(hipEventCreateWithFlags(&t->tab_ev, hipEventDisableTiming));
  {
    int lo = 0, hi = 0;
    N2_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));      // lo = lowest priority (largest number)
    // (measured and rejected, round 5: a CU-masked side stream -- hipExtStreamCreateWithCUMask with 32 .. 192 of
    // the 256 CUs -- doubles the step, 2.58 -> 5.16 ms, whatever the mask: profiles/r05_notes.md)
    N2_HIP(hipStreamCreateWithPriority(&t->side, hipStreamNonBlocking, lo));
    for (int i = 0; i < TrainState::kForkEvents; ++i)
      N2_HIP(hipEventCreateWithFlags(&t->ev_fork[i], hipEventDisableTiming));
    N2_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
    N2_HIP(hipEventCreateWithFlags(&t->infer_ev, hipEventDisableTiming));
    // schedule switches (n2nmn_debug_set before n2nmn_train_enable; tests/test_gpu_train.py runs every schedule
    // against the oracle): "train_overlap", "train_bg_wgs", "train_schedule", "train_chunks"
    t->overlap = knob_int(c, "train_overlap", 1) != 0;
    // models_vqa (lstm_dim 1024, 32-row backward tiles, GEMMs several times the size): its step
    // measured 7.93 ms with unbounded background launches against 8.14 bounded (same box)
    if (vqa_variant) t->bg_wgs = 0;
    if (const char* e = knob_str(c, "train_bg_wgs")) t->bg_wgs = std::max(0, atoi(e));
    if (const char* e = knob_str(c, "train_schedule")) t->schedule = atoi(e) != 0;
    if (const char* e = knob_str(c, "train_chunks")) {
      int v[3] = {0, 0, 0};
      const int got = sscanf(e, "%d,%d,%d", &v[0], &v[1], &v[2]);
      for (int i = 0; i < 3; ++i) {
        const int hi = i == 0 ? 100 : t->chunk_pct[i - 1];
        t->chunk_pct[i] = i < got ? std::min(std::max(v[i], 0), hi) : 0;
      }
    }
  }
  return N2NMN_OK;
}

int64_t n2nmn_grad_numel(const n2nmn_ctx* c) { return (c && c->train) ? c->train->total : N2NMN_EINVAL; }
int64_t n2nmn_grad_split(const n2nmn_ctx* c) { return (c && c->train) ? c->train->split : N2NMN_EINVAL; }

int n2nmn_grad_layout(const n2nmn_ctx* c, int variable, int64_t* offset, int64_t* numel) {
  N2_REQUIRE(c && c->train, N2NMN_EINVAL, "grad_layout: training not enabled");
  N2_REQUIRE(variable >= 0 && variable < (int)c->pub.size(), N2NMN_EINVAL, "grad_layout: bad index");
  const int id = c->pub[variable];           // index space of n2nmn_variable_info
  if (offset) *offset = c->train->var_off[id];
  if (numel) *numel = (int64_t)c->vars[id].numel;
  return N2NMN_OK;
}

int n2nmn_get_weight(n2nmn_ctx* c, const char* name, float* out, n2nmn_stream stream) {
  N2_REQUIRE(c && name && out, N2NMN_EINVAL, "get_weight: null argument");
  const n2nmn_ctx* r = root(c);
  auto it = r->index.find(name);
  if (it == r->index.end()) {
    set_last_error(std::string("get_weight: unknown variable '") + name + "'");
    return N2NMN_EKEY;
  }
  const Var& v = r->vars[it->second];
  train_infer_wait(r, S(stream));      // (the optimiser's second half runs on the side stream)
  N2_HIP(hipMemcpyAsync(out, v.mirror, sizeof(float) * v.numel, hipMemcpyDeviceToDevice, S(stream)));
  return N2NMN_OK;
}

static int check_train_io(const n2nmn_ctx* c, const n2nmn_train_io* io, const n2nmn_program* p,
                          const char* what) {
  N2_REQUIRE(c && io && p, N2NMN_EINVAL, std::string(what) + ": null argument");
  N2_REQUIRE(c->train, N2NMN_EINVAL, std::string(what) + ": call n2nmn_train_enable first");
  N2_REQUIRE(io->input_seq && io->seq_length && io->gt_layout && io->image_feat &&
                 io->answer_labels && io->losses && io->grads,
             N2NMN_EINVAL, std::string(what) + ": null tensor in n2nmn_train_io");
  N2_REQUIRE(io->N >= 1 && io->N <= c->d.N && io->T_enc >= 1 && io->T_enc <= c->d.T_encoder &&
                 io->T_dec >= 1 && io->T_dec <= c->d.T_decoder,
             N2NMN_ECAPACITY, std::string(what) + ": N / T_enc / T_dec exceed the context capacity");
  N2_REQUIRE(p->prog.num_rows == io->N, N2NMN_EINVAL,
             std::string(what) + ": program was not assembled from this batch");
  N2_REQUIRE(io->objective == N2NMN_OBJ_CLONING || io->objective == N2NMN_OBJ_POLICY_GRADIENT,
             N2NMN_EINVAL, std::string(what) + ": unknown objective");
  if (io->objective == N2NMN_OBJ_POLICY_GRADIENT) {
    N2_REQUIRE(io->expr_validity && io->baseline, N2NMN_EINVAL,
               std::string(what) + ": policy gradient needs expr_validity and baseline");
    N2_REQUIRE(io->baseline_decay >= 0.f && io->baseline_decay <= 1.f, N2NMN_EINVAL,
               std::string(what) + ": baseline_decay outside [0, 1]");
    N2_REQUIRE(c->d.num_vocab_nmn <= 32, N2NMN_ECAPACITY,
               std::string(what) + ": policy gradient keeps token validity in 32-bit masks");
  }
  return N2NMN_OK;
}

int n2nmn_train_forward(n2nmn_ctx* c, const n2nmn_train_io* io, n2nmn_program* p,
                        n2nmn_stream stream) {
  int rc = check_train_io(c, io, p, "train_forward");
  if (rc != N2NMN_OK) return rc;
  N2_REQUIRE(is_committed(c), N2NMN_ENOWEIGHT, "train_forward: weights not committed");
  TrainState* t = c->train;
  hipStream_t s = S(stream);
  const n2nmn_dims& d = c->d;
  const int N = io->N, L = d.lstm_dim;
  const size_t nl = (size_t)N * L;
  if (t->pack_epoch != c->commit_epoch) {
    // only the backward pass reads the transposed packs: side stream (joined below, before the
    // module network), off the forward's critical path
    rc = repack_transposed(c, t->fork(s));
    if (rc != N2NMN_OK) return rc;
  }
  // slot 0 of the kept encoder sequences is the zero initial state.  Slot strides follow the actual
  // batch size, so a previous call with another N has written through what is slot 0 now: clear it
  // whenever the shape changes (otherwise it is never written)
  if (t->last_N != N) {
    N2_HIP(hipMemsetAsync(t->rec.ec0s, 0, sizeof(float) * nl, s));
    N2_HIP(hipMemsetAsync(t->rec.ec1s, 0, sizeof(float) * nl, s));
    N2_HIP(hipMemsetAsync(t->rec.eh0s, 0, sizeof(float) * nl, s));
    N2_HIP(hipMemsetAsync(t->rec.eh1s, 0, sizeof(float) * nl, s));
  }
  n2nmn_seq2seq_io sio{};
  sio.input_seq = io->input_seq; sio.seq_length = io->seq_length; sio.T_enc = io->T_enc; sio.N = N;
  // policy gradient: the given tokens are the decoder's own samples -> automaton validity (mode 2)
  const bool rl = io->objective == N2NMN_OBJ_POLICY_GRADIENT;
  sio.T_dec = io->T_dec; sio.use_gt_layout = rl ? 2 : 1; sio.gt_layout = io->gt_layout;
  t->last_N = N; t->last_T = io->T_enc; t->last_Td = io->T_dec;
  const bool vqa = d.variant == N2NMN_VARIANT_VQA;
  N2_REQUIRE(vqa || !(io->drop_enc0 || io->drop_dec0 || io->drop_qpn_h || io->drop_qpn_fc1),
             N2NMN_EINVAL, "train_forward: dropout belongs to the models_vqa variant");
  sio.drop_enc0 = io->drop_enc0; sio.drop_dec0 = io->drop_dec0;
  c->rec = &t->rec;
  float* scores = io->scores ? io->scores : t->scores;
  // the hoisted conv_image GEMMs of the module network need only the image features: they run on
  // the side stream beside the (strictly sequential) encoder
  rc = run_program(c, p->prog, io->image_feat, c->word_vecs, N, scores, nullptr, nullptr, nullptr,
                   0, 0, s, RP_PREP);
  if (rc == N2NMN_OK)
    rc = run_program(c, p->prog, io->image_feat, c->word_vecs, N, scores, nullptr, nullptr,
                     nullptr, 0, 0, t->fork(s), RP_CONV);
  if (rc == N2NMN_OK) rc = encoder_impl(c, &sio, s);
  if (rc == N2NMN_OK) {
    // the decoder starts from the encoder's final state (nmn3_netgen_att.py:177)
    launch_unpack_h(c->fc0, t->rec.dc0s, N, L, d.N, s);     // (the cell state is packed like h)
    launch_unpack_h(c->fc1, t->rec.dc1s, N, L, d.N, s);
    launch_unpack_h(c->fh0, t->rec.dh0s, N, L, d.N, s);
    launch_unpack_h(c->fh1, t->rec.dh1s, N, L, d.N, s);
    rc = decoder_impl(c, &sio, s);
  }
  t->join(s);
  if (rc == N2NMN_OK)
    rc = run_program(c, p->prog, io->image_feat, c->word_vecs, N, scores, nullptr, nullptr, nullptr,
                     0, 0, s, RP_REST);
  c->rec = nullptr;
  if (rc != N2NMN_OK) return rc;
  if (c->qpn_h) {                    // scores = scores_nmn + scores_qpn (models_vqa/nmn3_model.py:106-112)
    rc = qpn_forward(c, N, scores, io->drop_qpn_h, io->drop_qpn_fc1, s);
    if (rc != N2NMN_OK) return rc;
  }
  if (scores != t->scores)
    N2_HIP(hipMemcpyAsync(t->scores, scores, sizeof(float) * (size_t)N * d.num_choices,
                          hipMemcpyDeviceToDevice, s));
  N2_HIP(hipMemsetAsync(io->losses, 0, sizeof(float) * 8, s));
  {
    ProfScope ps(c, F_BWD_MISC, 6.0 * N * d.num_choices, 4.0 * 2 * N * d.num_choices, s);
    if (rl) {
      LossRlArgs la{};
      la.scores = t->scores; la.labels = io->answer_labels; la.log_seq_prob = t->rec.lsp;
      la.neg_entropy = c->negent; la.expr_validity = io->expr_validity; la.N = N;
      la.C = d.num_choices; la.invalid_expr_loss = io->invalid_expr_loss;
      la.baseline_decay = io->baseline_decay; la.baseline = io->baseline; la.dscores = t->dscores;
      la.losses = io->losses; la.coef = t->rl_coef;
      launch_loss_rl(la, s);
      if (t->ds_pad) {                 // padded copy for the batch GEMMs of the models_vqa heads
        N2_HIP(hipMemsetAsync(t->ds_pad, 0, sizeof(float) * (size_t)N * t->Cp, s));
        N2_HIP(hipMemcpy2DAsync(t->ds_pad, sizeof(float) * t->Cp, t->dscores,
                                sizeof(float) * d.num_choices, sizeof(float) * d.num_choices, N,
                                hipMemcpyDeviceToDevice, s));
      }
    } else {
      launch_loss(t->scores, io->answer_labels, t->rec.lsp, N, d.num_choices, t->dscores, io->losses, s,
                  t->ds_pad, t->Cp);
    }
  }
  return check_launch("train_forward");
}

int n2nmn_train_backward(n2nmn_ctx* c, const n2nmn_train_io* io, n2nmn_program* pp, int phase,
                         n2nmn_stream stream) {
  int rc = check_train_io(c, io, pp, "train_backward");
  if (rc != N2NMN_OK) return rc;
  const bool defer_join = (phase & N2NMN_BWD_DEFER_JOIN) != 0;
  phase &= ~N2NMN_BWD_DEFER_JOIN;
  N2_REQUIRE(phase == 0 || phase == 1, N2NMN_EINVAL, "train_backward: phase must be 0 or 1");
  TrainState* t = c->train;
  Program& p = pp->prog;
  hipStream_t s = S(stream);
  const n2nmn_dims& d = c->d;
  const int N = io->N, L = d.lstm_dim, E = d.embed_dim_txt, T = io->T_enc, Td = io->T_dec,
            V = d.num_vocab_nmn, Vt = d.num_vocab_txt, M = d.map_dim, Mp = c->Mp, D = d.D,
            HW = d.H * d.W, C = d.num_choices;
  const int Ep = round_up(E, 64);
  auto G = [&](int var) { return gptr(c, io, var); };
  auto mir = [&](int var) { return (const float*)c->vars[var].mirror; };

  if (phase == 0) {
    // ------------------------------- module network ---------------------------------------
    const int nn = (int)p.dev_nodes.size();
    {
      // one launch clears the flat gradient, the per-step block (garena, dtmap, dpfc, gda, dwv, the
      // carried dH/dC, both x-table gradients, the active-row counter) and the used part of the
      // two d conv_image slabs
      ZeroRanges z{};
      auto add = [&](void* ptr, size_t bytes) { if (bytes) { z.ptr[z.n] = ptr; z.bytes[z.n] = bytes; ++z.n; } };
      add(io->grads, sizeof(float) * (size_t)t->total);
      add(t->zero_begin, (size_t)(t->zero_end - t->zero_begin));
      if (nn > 0) {
        add(t->dmfind, sizeof(float) * (size_t)p.num_find_img * HW * Mp);
        add(t->dmfsp, sizeof(float) * (size_t)p.num_fsp_img * HW * Mp);
      }
      launch_zero_ranges(z, s);
    }
    if (nn > 0) {
      // tables: text slot -> (word_vecs row, weight set); pooling slot/input -> fc_att weight set
      {
        N2_HIP(hipEventSynchronize(t->tab_ev));
        int32_t* row = t->tab_host;
        int32_t* ws = row + c->max_text;
        int32_t* sel = ws + c->max_text;
        for (int i = 0; i < 2 * p.num_pool; ++i) sel[i] = -1;
        for (const DevNode& nd : p.dev_nodes) {
          if (nd.tslot >= 0) {
            row[nd.tslot] = nd.t * N + nd.n;
            int w5;
            switch (nd.op) {
              case N2NMN_OP_FIND: case N2NMN_OP_FILTER: w5 = 0; break;
              case N2NMN_OP_FIND_SAME_PROPERTY: w5 = 1; break;
              case N2NMN_OP_TRANSFORM: w5 = 2; break;
              case N2NMN_OP_SAME_PROPERTY: w5 = 3; break;
              default: w5 = 4; break;
            }
            ws[nd.tslot] = w5;
          }
          if (nd.pslot >= 0) {
            if (nd.op == N2NMN_OP_FIND_SAME_PROPERTY) sel[2 * nd.pslot] = 0;
            else if (nd.op == N2NMN_OP_SAME_PROPERTY) { sel[2 * nd.pslot] = 1; sel[2 * nd.pslot + 1] = 2; }
            else sel[2 * nd.pslot] = 3;
          }
        }
        // the three tables are one block on both sides: one upload
        N2_HIP(hipMemcpyAsync(t->tslot_row, row,
                              sizeof(int32_t) * (2 * (size_t)c->max_text + 2 * (size_t)p.num_pool),
                              hipMemcpyHostToDevice, s));
        N2_HIP(hipEventRecord(t->tab_ev, s));
      }
      ModuleWeights w = module_weights(c);
      ModuleBuffers b{};
      b.nodes = c->dev_nodes; b.tab = c->dev_tab; b.arena = c->arena; b.tmap = c->tmap;
      b.pfc = c->pfc; b.mfind = c->mfind; b.mfsp = c->mfsp; b.feat = io->image_feat;
      b.word_vecs = c->word_vecs; b.scores = nullptr; b.N_full = N; b.H = d.H; b.W = d.W; b.D = D;
      b.M = M; b.Mp = Mp; b.wl_cap = 0; b.E = E; b.C = C; b.HWp = c->HWp; b.ksize = d.kernel_size;
      b.pooled = t->rec.pooled;
      ModuleGrads g{};
      g.garena = t->garena; g.dtmap = t->dtmap; g.dpfc = t->dpfc; g.gda = t->gda; g.dmfind = t->dmfind;
      g.dmfsp = t->dmfsp; g.dscores = t->dscores; g.dwv = t->dwv;
      g.gwe[0] = G(V_FIND_E_W); g.gbe[0] = G(V_FIND_E_B);
      g.gwe[1] = G(V_FSP_E_W); g.gbe[1] = G(V_FSP_E_B);
      g.gwe[2] = G(V_TR_E_W); g.gbe[2] = G(V_TR_E_B);
      g.gKt = G(V_TR_MAPS_W); g.gbt = G(V_TR_MAPS_B);
      const int attb[4] = {V_FSP_ATT_B, V_SP_ATT0_B, V_SP_ATT1_B, V_DE_ATT_B};
      for (int i = 0; i < 4; ++i) g.gbatt[i] = G(attb[i]);
      const int answ[7] = {V_EXIST_W, V_COUNT_W, V_EQ_W, V_MORE_W, V_LESS_W, V_SP_E_W, V_DE_E_W};
      for (int i = 0; i < 7; ++i) { g.gWans[i] = G(answ[i]); g.gbans[i] = G(answ[i] + 1); }
      if (c->big_heads) {              // den[n] = dscores[n] . W_e^T for the whole batch
        gemm_nt(c, s, t->ds_pad, t->Cp, N, t->Cp, t->wde_T_p, round_up(M, 64), round_up(C, 32), M,
                t->hb_den, Mp, false);
        g.hb_den = t->hb_den; g.hb_en = t->hb_en; g.hb_sel = t->hb_sel;
      }
      bool att_done = false;
      hipStream_t sd = nullptr;        // side stream, forked once every level launch is enqueued
      for (int li = (int)p.launches.size() - 1; li >= 0; --li) {
        const Launch& l = p.launches[li];
        switch (l.kind) {
          case LK_HEAD: {
            ProfScope ps(c, F_BWD_MISC, l.count * 4.0 * M * C, 4.0 * l.count * (8.0 * Mp + C), s);
            launch_heads_bwd(w, b, g, l.offset, l.count, s);
            break;
          }
          case LK_POOL: {
            const int jobs = l.count / POOL_PARTS;
            ProfScope ps(c, F_BWD_MISC, jobs * (2.0 * HW * D + 2.0 * D * M),
                         4.0 * (jobs * ((double)HW * D + 2.0 * HW) + (double)D * M), s);
            launch_pool_bwd(w, b, g, l.offset, jobs, 2 * POOL_PARTS, s);
            break;
          }
          case LK_ATT: {
            ProfScope ps(c, F_BWD_MISC, l.count * 8.0 * HW * M / FIND_PARTS,
                         4.0 * l.count * 2.0 * HW * Mp / FIND_PARTS, s);
            launch_att_bwd(w, b, g, l.offset, l.count, s);
            break;
          }
          case LK_CONV_FIND:
          case LK_CONV_FSP:
          case LK_TEXTMAP: {
            // every level launch is enqueued: dM / dpfc / dtmap are final.  The weight-gradient
            // GEMMs below only read them -> side stream; the chain textmap_bwd -> word_vecs_bwd ->
            // attention backward continues on the caller's stream.
            if (!sd) sd = t->fork(s);
            if (!att_done) {            // fc_att weight gradients from the kept pooled features
              att_done = true;
              const int attw[4] = {V_FSP_ATT_W, V_SP_ATT0_W, V_SP_ATT1_W, V_DE_ATT_W};
              if (p.num_pool > 0) {     // the four fc_att weight sets in one launch
                GemmTnArgs ga{};
                ga.A = t->rec.pooled; ga.lda = D; ga.M = D; ga.a_group_size = 1;
                ga.B = t->dpfc; ga.ldb = Mp; ga.N = M; ga.b_sel = t->pool_sel; ga.R = 2 * p.num_pool;
                ga.ldc = M; ga.nprob = 0;
                // weight sets the variant does not have (models_vqa: SameProperty) are left out:
                // their gradient slots have no storage of their own
                for (int i = 0; i < 4; ++i) {
                  if (c->vars[attw[i]].numel == 0) continue;
                  const int q = ga.nprob++;
                  ga.A_p[q] = ga.A; ga.B_p[q] = ga.B; ga.bsel_p[q] = i;
                  ga.C_p[q] = G(attw[i]); ga.colsum_p[q] = nullptr;
                }
                ProfScope ps(c, F_GEMM_TN, 2.0 * D * M * 2.0 * p.num_pool,
                             4.0 * (2.0 * p.num_pool * (D + Mp) + 4.0 * D * M), sd);
                launch_gemm_tn(ga, sd, t->bg(sd));
              }
            }
            if (l.kind == LK_TEXTMAP) {
              {
                ProfScope ps(c, F_BWD_MISC, 2.0 * p.num_text * E * M,
                             4.0 * (l.count * (double)E * Mp + p.num_text * (double)(E + Mp)), s);
                launch_textmap_bwd(w, b, g, l.offset, l.count, s);
              }
              const int txw[5] = {V_FIND_TXT_W, V_FSP_TXT_W, V_TR_TXT_W, V_SP_TXT_W, V_DE_TXT_W};
              if (p.num_text > 0) {     // the five fc_text weight sets (+ biases) in one launch
                GemmTnArgs ga{};
                ga.A = c->word_vecs; ga.lda = E; ga.M = E; ga.a_group_idx = t->tslot_row;
                ga.a_group_size = 1; ga.B = t->dtmap; ga.ldb = Mp; ga.N = M; ga.b_sel = t->tslot_ws;
                ga.R = p.num_text; ga.ldc = M; ga.nprob = 0;
                for (int i = 0; i < 5; ++i) {
                  if (c->vars[txw[i]].numel == 0) continue;
                  const int q = ga.nprob++;
                  ga.A_p[q] = ga.A; ga.B_p[q] = ga.B; ga.bsel_p[q] = i;
                  ga.C_p[q] = G(txw[i]); ga.colsum_p[q] = G(txw[i] + 1);
                }
                ProfScope ps(c, F_GEMM_TN, 2.0 * E * M * (double)p.num_text,
                             4.0 * (p.num_text * (double)(E + Mp) + 5.0 * E * M), sd);
                launch_gemm_tn(ga, sd, t->bg(sd));
              }
            } else {
              const bool fsp = l.kind == LK_CONV_FSP;
              gemm_tn(c, sd, io->image_feat, D, D, fsp ? t->dmfsp : t->dmfind, Mp, M, l.count * HW,
                      G(fsp ? V_FSP_IMG_W : V_FIND_IMG_W), M, c->dev_tab + l.offset, HW, nullptr, 0,
                      nullptr, nullptr, G(fsp ? V_FSP_IMG_B : V_FIND_IMG_B));
            }
            break;
          }
          default: break;
        }
      }
      if (c->big_heads) {              // dW_e = en^T . dscores (+ db_e) over the rows that have a head
        hipStream_t sh = t->fork(s);
        gemm_tn(c, sh, t->hb_en, Mp, M, t->ds_pad, t->Cp, C, N, G(V_DE_E_W), C, nullptr, 1,
                t->hb_sel, 1, nullptr, nullptr, G(V_DE_E_B));
      }
    }
    if (c->qpn_h) {
      // question prior net backward (models_vqa/question_prior_net.py:10-28): scores_qpn shares
      // dscores with the module network's answer
      const int Hq = d.qpn_hidden;
      hipStream_t sq = t->fork(s);
      gemm_tn(c, sq, c->qpn_hid, Hq, Hq, t->ds_pad, t->Cp, C, N, G(V_QPN_W2), C, nullptr, 1, nullptr, 0,
              nullptr, nullptr, G(V_QPN_B2));
      gemm_nt(c, s, t->ds_pad, t->Cp, N, t->Cp, t->qpn_W2T_p, round_up(Hq, 64), round_up(C, 32), Hq,
              t->qpn_dad, Hq, false);
      launch_qpn_dpre(t->qpn_dad, c->qpn_hid, io->drop_qpn_fc1, (size_t)N * Hq, s);
      sq = t->fork(s);
      gemm_tn(c, sq, c->qpn_h, 2 * L, 2 * L, t->qpn_dad, Hq, Hq, N, G(V_QPN_W1), Hq, nullptr, 1, nullptr,
              0, nullptr, nullptr, G(V_QPN_B1));
      gemm_nt(c, s, t->qpn_dad, Hq, N, Hq, t->qpn_W1T_p, round_up(2 * L, 64), round_up(Hq, 32), 2 * L,
              t->qpn_dh, 2 * L, false);
    }
    // ------------------------------- decoder ----------------------------------------------
    // rows (tau, n) inside the question's length: reduction index of every encoder-side weight
    // gradient (here the embedding gradient through word_vecs; in phase 1 the LSTM's)
    // chunk starts of the encoder's reverse-time pass (descending).  Every chunk is a launch that
    // updates all of dW (fixed cost ~70 us whatever its rows), so there are few: by default
    // [T/3, T) under the last third of the recurrence and [0, T/3) after it (n2nmn_debug_set "train_chunks" =
    // up to three descending percentages of T)
    for (int i = 0; i < 4; ++i) t->chunk_start[i] = i < 3 ? T * t->chunk_pct[i] / 100 : 0;
    launch_active_rows(io->seq_length, T, N, t->act_rows, t->act_count, t->act_rows_ch,
                       t->chunk_start, s);
    {
      ProfScope ps(c, F_BWD_MISC, 4.0 * Td * T * N * E, 4.0 * N * (double)(2 * T * E + 2 * Td * E + 2 * Td * T), s);
      launch_word_vecs_bwd(t->dwv, c->atts, io->input_seq, io->seq_length, mir(V_ENC_EMB), Td, T, N,
                           E, t->datts_wv, t->dE, s);
    }
    if (!c->big_vocab) {
      // d embedding_mat (through word_vecs) = onehot(word)^T . dE over the active rows; it lands in
      // the encoder bucket, which is only finished in phase 1 -> side stream, joined with the rest
      // (large vocabularies: dE is scattered in phase 1 together with the LSTM input's share)
      hipStream_t sd = t->fork(s);
      GemmTnArgs g1{};
      g1.A = nullptr; g1.lda = 0; g1.M = Vt; g1.a_onehot = io->input_seq;
      g1.B = t->dE; g1.ldb = E; g1.N = E; g1.R = T * N; g1.C = G(V_ENC_EMB); g1.ldc = E;
      g1.row_idx = t->act_rows; g1.r_dev = t->act_count;
      ProfScope ps(c, F_GEMM_TN, 2.0 * Vt * (double)E * T * N, 4.0 * ((double)T * N * E + Vt * (double)E), sd);
      launch_gemm_tn(g1, sd, t->bg(sd));
    }
    N2_REQUIRE(!c->eht_partial, N2NMN_EINVAL,
               "train_backward: the forward left a partial encoder_h_transform (listed-row GEMM)");
    DecBwdArgs a{};
    a.scores = t->rec.tscores; a.gt = io->gt_layout; a.q = c->qbuf; a.eht = c->eht;
    a.eout = c->enc_out; a.atts = c->atts; a.datts_wv = t->datts_wv; a.seq_len = io->seq_length;
    a.v = mir(V_ATT_V); a.Wy = mir(V_TOK_W); a.T = T; a.N = N; a.L = L; a.V = V; a.Td = Td;
    a.inv_n = 1.0f / (float)N;
    if (io->objective == N2NMN_OBJ_POLICY_GRADIENT) {
      a.coef = t->rl_coef; a.valid_bits = t->rec.valid_bits;
      a.ent_coef = io->lambda_entropy / (float)N;
    }
    a.dsc = t->dsc; a.dout = t->dout; a.dctx = t->dctx; a.de = t->de; a.dq = t->dq; a.dvp = t->dvp;
    a.deht = t->deht; a.deout = t->denc_out;
    {
      ProfScope ps(c, F_BWD_MISC, (double)Td * N * (8.0 * T * L + 4.0 * L * V),
                   4.0 * Td * N * (2.0 * T * L + 4.0 * L), s);
      launch_dec_bwd_a(a, s);
    }
    {
      ProfScope ps(c, F_BWD_MISC, (double)Td * T * N * 8.0 * L,
                   4.0 * ((double)T * N * 3 * L + (double)Td * N * 2 * L), s);
      launch_dec_bwd_b(a, s);
    }
    const int RT = Td * N;
    {
      hipStream_t sd = t->fork(s);     // token / attention projection gradients: leaves
      const TnProblem tok[2] = {{c->dec_h1_all, t->dsc, G(V_TOK_W), G(V_TOK_B)},
                                {t->rec.ctx, t->dsc, G(V_TOK_W) + (size_t)L * V, nullptr}};
      gemm_tn_batch(c, sd, 2, tok, L, L, 16, V, RT, V);
      colsum(c, sd, t->dvp, RT, L, L, G(V_ATT_V));
      gemm_tn(c, sd, c->dec_h1_all, L, L, t->dq, L, L, RT, G(V_ATT_W), L, nullptr, 1, nullptr, 0,
              nullptr, nullptr, G(V_ATT_B));
    }
    gemm_nt(c, s, t->dq, L, RT, L, t->att_WT_p, L, c->KpL, L, t->dout, L, true);
    // BPTT through the decoder LSTM stack; its initial state is the encoder's final state
    BpttArgs ba{};
    ba.T = Td; ba.N = N; ba.want_init_grad = true; ba.seq_len = nullptr;
    ba.g0 = t->rec.dg0; ba.g1 = t->rec.dg1; ba.c0s = t->rec.dc0s; ba.c1s = t->rec.dc1s;
    ba.dout = t->dout; ba.Wt0 = t->dec_Wt0; ba.Wt1 = t->dec_Wt1; ba.drop0 = io->drop_dec0;
    ba.dz0_all = t->ddz0_all; ba.dz1_all = t->ddz1_all;
    rc = run_bptt(c, ba, s);
    if (rc != N2NMN_OK) return rc;
    if (c->qpn_h)                    // dH now holds the decoder's gradient of the encoder's final h
      launch_qpn_dh_add(t->qpn_dh, io->drop_qpn_h, t->dH0, t->dH1, N, L, s);
    // Everything below is a leaf of the backward graph (input-table, embedding and recurrent weight
    // gradients of the decoder, then the finish of the late bucket): side stream.  With
    // N2NMN_BWD_DEFER_JOIN the caller's stream does not wait for it here -- the encoder's backward
    // (phase 1) starts at once and the late bucket is final after n2nmn_train_join / phase 1.
    hipStream_t sl = t->schedule ? t->fork(s) : s;
    launch_dec_xidx(io->gt_layout, Td, N, V, t->dec_xidx, sl);
    // gradient of the input-projection table: dxtab = onehot(idx)^T . dz0  (one-hot gemm_tn)
    {
      GemmTnArgs g1{};
      g1.A = nullptr; g1.lda = 0; g1.M = V + 1; g1.a_onehot = t->dec_xidx;
      g1.B = t->ddz0_all; g1.ldb = 4 * L; g1.N = 4 * L; g1.R = RT; g1.C = t->dxtab_dec; g1.ldc = 4 * L;
      ProfScope ps(c, F_GEMM_TN, 2.0 * (V + 1) * 4.0 * L * RT, 4.0 * ((double)RT * 4 * L + (V + 1) * 4.0 * L), sl);
      launch_gemm_tn(g1, sl, t->bg(sl));
    }
    gemm_tn(c, sl, c->dec_emb_cat, E, E, t->dxtab_dec, 4 * L, 4 * L, V + 1, G(V_DEC_W0), 4 * L,
            nullptr, 1, nullptr, 0, nullptr, nullptr, G(V_DEC_B0));
    gemm_nt(c, sl, t->dxtab_dec, 4 * L, V, 4 * L, t->dec_W0xT_p, Ep, t->KpL4, E, G(V_DEC_EMB), E, true);
    gemm_nt(c, sl, t->dxtab_dec + (size_t)V * 4 * L, 4 * L, 1, 4 * L, t->dec_W0xT_p, Ep, t->KpL4, E,
            G(V_DEC_GO), E, true);
    {
      // the three recurrent weight gradients of the stack, one launch: W0 (h part) = h0(t-1)^T dz0,
      // W1 = [h0(t) ; h1(t-1)]^T dz1 (+ b1)
      const TnProblem dw[3] = {
          {t->rec.dh0s, t->ddz0_all, G(V_DEC_W0) + (size_t)E * 4 * L, nullptr},
          // layer 1 saw layer 0's output through the dropout multipliers
          {io->drop_dec0 ? t->rec.dh0d : t->rec.dh0s + (size_t)N * L, t->ddz1_all, G(V_DEC_W1), G(V_DEC_B1)},
          {t->rec.dh1s, t->ddz1_all, G(V_DEC_W1) + (size_t)L * 4 * L, nullptr}};
      gemm_tn_batch(c, sl, 3, dw, L, L, 4 * L, 4 * L, RT, 4 * L);
    }
    if (!t->schedule) { t->join(s); }  // (module / projection gradients ran on the side stream)
    {                                  // decoder + module gradients complete: finish the late bucket
      ProfScope ps(c, F_OPTIMISER, 3.0 * (t->total - t->split), 4.0 * 3 * (t->total - t->split), sl);
      launch_grad_finish(io->grads, (const float* const*)t->mirrors_dev, t->var_off_dev, t->decay_dev,
                         t->segs_dev + t->nsegs_early, t->nsegs - t->nsegs_early, 1.0f,
                         io->weight_decay, io->losses + 2, sl);
    }
    t->deferred = defer_join && t->schedule;
    if (!t->deferred) t->join(s);
    return check_launch("train_backward(0)");
  }

  // ------------------------------- phase 1: encoder -----------------------------------------
  const int RT = T * N;
  {
    hipStream_t sd = t->fork(s);
    gemm_tn(c, sd, c->enc_out, L, L, t->deht, L, L, RT, G(V_EHT_W), L, nullptr, 1, nullptr, 0,
            nullptr, nullptr, G(V_EHT_B));
  }
  // d encoder_outputs = (through the context vectors, already in denc_out) + deht . W_eht^T
  gemm_nt(c, s, t->deht, L, RT, L, t->eht_WT_p, L, c->KpL, L, t->denc_out, L, true);
  // rows (t, n) past the question's length have dz = 0: the weight-gradient GEMMs run over the
  // compacted list of active rows built in phase 0 (about 56 % of T*N with lengths in [5, 45])
  BpttArgs ba{};
  ba.T = T; ba.N = N; ba.want_init_grad = false; ba.seq_len = io->seq_length;
  ba.g0 = t->rec.eg0; ba.g1 = t->rec.eg1; ba.c0s = t->rec.ec0s; ba.c1s = t->rec.ec1s;
  ba.dout = t->denc_out; ba.Wt0 = t->enc_Wt0; ba.Wt1 = t->enc_Wt1; ba.drop0 = io->drop_enc0;
  ba.dz0_all = t->dz0_all; ba.dz1_all = t->dz1_all;
  ba.perm = c->perm; ba.nact = c->nact;
  // The weight-gradient GEMMs follow the recurrence chunk by chunk on the side stream: when the
  // step that starts chunk ci has been enqueued, dz0 / dz1 of the chunk's rows are complete, and its
  // GEMMs (reduction over the chunk's own list of active rows; C += with atomics or, single split,
  // in stream order) run under the remaining steps.  Only the last, shortest chunk is left for after
  // the recurrence.  (Round 2 ran all of it after the last step: 240 us of a 2.75 ms step.)
  auto chunk_grads = [&](int ci) {
    const int beg = t->chunk_start[ci], end = ci == 0 ? T : t->chunk_start[ci - 1];
    if (end <= beg) return;
    hipStream_t sd = t->schedule ? t->fork(s) : s;
    const int32_t* rows = t->act_rows_ch + (size_t)beg * N;
    const int32_t* cnt = t->act_count + 1 + ci;
    const int Rc = (end - beg) * N;
    if (c->big_vocab) {
      // the layer-0 input weights straight from the batch's rows: dW0[0:E] = emb[word_r]^T . dz0_r (+ db0)
      gemm_tn(c, sd, mir(V_ENC_EMB), E, E, t->dz0_all, 4 * L, 4 * L, Rc, G(V_ENC_W0), 4 * L,
              io->input_seq, 1, nullptr, 0, rows, cnt, G(V_ENC_B0));
    } else {
      GemmTnArgs g1{};
      g1.A = nullptr; g1.lda = 0; g1.M = Vt; g1.a_onehot = io->input_seq;
      g1.B = t->dz0_all; g1.ldb = 4 * L; g1.N = 4 * L; g1.R = Rc; g1.C = t->dxtab_enc; g1.ldc = 4 * L;
      g1.row_idx = rows; g1.r_dev = cnt;
      ProfScope ps(c, F_GEMM_TN, 2.0 * Vt * 4.0 * L * Rc, 4.0 * ((double)Rc * 4 * L + Vt * 4.0 * L), sd);
      launch_gemm_tn(g1, sd, t->bg(sd));
    }
    const TnProblem dw[3] = {
        {t->rec.eh0s, t->dz0_all, G(V_ENC_W0) + (size_t)E * 4 * L, nullptr},
        {io->drop_enc0 ? t->rec.eh0d : t->rec.eh0s + (size_t)N * L, t->dz1_all, G(V_ENC_W1), G(V_ENC_B1)},
        {t->rec.eh1s, t->dz1_all, G(V_ENC_W1) + (size_t)L * 4 * L, nullptr}};
    gemm_tn_batch(c, sd, 3, dw, L, L, 4 * L, 4 * L, Rc, 4 * L, rows, cnt);
  };
  const std::function<void(int)> after_step = [&](int t0) {
    for (int ci = 0; ci < 4; ++ci)
      if (t0 == t->chunk_start[ci] && (ci == 0 || t->chunk_start[ci] != t->chunk_start[ci - 1]))
        chunk_grads(ci);
  };
  rc = run_bptt(c, ba, s, t->schedule ? &after_step : nullptr);
  if (rc != N2NMN_OK) return rc;
  if (!t->schedule)
    for (int ci = 0; ci < 4; ++ci)
      if (ci == 0 || t->chunk_start[ci] != t->chunk_start[ci - 1]) chunk_grads(ci);
  {
    hipStream_t sd = t->schedule ? t->fork(s) : s;       // (in stream order after the chunks)
    if (c->big_vocab) {
      // dE_r += dz0_r . W0[0:E]^T;  demb[word_r] += dE_r   (large vocabularies: scattered)
      gemm_nt(c, sd, t->dz0_all, 4 * L, RT, 4 * L, t->enc_W0xT_p, Ep, t->KpL4, E, t->dE, E, true);
      ProfScope ps(c, F_BWD_MISC, (double)RT * E, 4.0 * 2 * RT * E, sd);
      launch_embed_scatter(t->dE, io->input_seq, t->act_rows, t->act_count, RT, E, G(V_ENC_EMB), sd);
    } else {
      gemm_tn(c, sd, mir(V_ENC_EMB), E, E, t->dxtab_enc, 4 * L, 4 * L, Vt, G(V_ENC_W0), 4 * L,
              nullptr, 1, nullptr, 0, nullptr, nullptr, G(V_ENC_B0));
      gemm_nt(c, sd, t->dxtab_enc, 4 * L, Vt, 4 * L, t->enc_W0xT_p, Ep, t->KpL4, E, G(V_ENC_EMB), E, true);
    }
  }
  t->join(s);                          // every weight gradient (side stream) done
  t->deferred = false;
  {
    ProfScope ps(c, F_OPTIMISER, 3.0 * t->split, 4.0 * 3 * t->split, s);
    launch_grad_finish(io->grads, (const float* const*)t->mirrors_dev, t->var_off_dev, t->decay_dev,
                       t->segs_dev, t->nsegs_early, 1.0f, io->weight_decay, io->losses + 2, s);
  }
  launch_loss_total(io->losses, io->weight_decay,
                    io->objective == N2NMN_OBJ_POLICY_GRADIENT ? io->lambda_entropy : 0.f, s);
  return check_launch("train_backward(1)");
}

int n2nmn_train_join(n2nmn_ctx* c, n2nmn_stream stream) {
  N2_REQUIRE(c && c->train, N2NMN_EINVAL, "train_join: call n2nmn_train_enable first");
  c->train->join(S(stream));
  return N2NMN_OK;
}

int n2nmn_dropout_multipliers(float* out, int64_t n, float keep_prob, uint64_t seed, uint64_t offset,
                              n2nmn_stream stream) {
  N2_REQUIRE(out && n >= 0, N2NMN_EINVAL, "dropout_multipliers: bad argument");
  N2_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, N2NMN_EINVAL,
             "dropout_multipliers: keep_prob must be in (0, 1]");
  if (n == 0) return N2NMN_OK;
  launch_dropout_mult(out, (size_t)n, keep_prob, seed, offset, S(stream));
  return check_launch("dropout_multipliers");
}

int n2nmn_adam_step(n2nmn_ctx* c, const float* grads, float grad_scale, float lr, float beta1,
                    float beta2, float eps, float max_grad_l2_norm, int64_t step,
                    n2nmn_stream stream) {
  N2_REQUIRE(c && grads, N2NMN_EINVAL, "adam_step: null argument");
  N2_REQUIRE(c->train, N2NMN_EINVAL, "adam_step: call n2nmn_train_enable first");
  N2_REQUIRE(step >= 1, N2NMN_EINVAL, "adam_step: step counts from 1");
  TrainState* t = c->train;
  hipStream_t s = S(stream);
  t->join(s);                          // (a previous step's second half may still be on the side stream)
  N2_HIP(hipMemsetAsync(t->norm2, 0, sizeof(float) * V_COUNT_, s));
  const double lr_t = (double)lr * std::sqrt(1.0 - std::pow((double)beta2, (double)step)) /
                      (1.0 - std::pow((double)beta1, (double)step));
  {
    ProfScope ps(c, F_OPTIMISER, 2.0 * t->total, 4.0 * t->total, s);
    launch_grad_sqnorm(grads, t->segs_dev, t->nsegs, grad_scale, t->norm2, s);
  }
  // The update and the re-pack in two halves: the ENCODER's variables (they come first in the flat
  // layout) and the operands made from them stay on the caller's stream -- the next forward pass
  // starts with the encoder and needs nothing else for its first ~0.4 ms -- while the decoder's and the
  // module network's update, their operand packs and the inference-only tables run on the side
  // stream (the decoder / module network / walker entry points wait for them).  Round 3: the
  // encoder's first step starts ~75 us after the optimiser instead of ~175.
  hipStream_t sr = s;
  if (t->schedule && t->nsegs_early > 0 && t->nsegs_early < t->nsegs) sr = t->fork(s);
  // <= 0: no clipping (exp_vqa/train_vqa_gt_layout.py:119-123): clip / max(|g|, clip) = 1
  const float clip = max_grad_l2_norm > 0.f ? max_grad_l2_norm : 3.0e38f;
  const int n0 = sr == s ? t->nsegs : t->nsegs_early;
  {
    ProfScope ps(c, F_OPTIMISER, 12.0 * t->total, 4.0 * 7 * t->total, s);
    launch_adam(grads, t->mirrors_dev, t->var_off_dev, t->segs_dev, n0, t->norm2, grad_scale, clip,
                (float)lr_t, beta1, beta2, eps, t->m, t->v, s);
  }
  if (n0 < t->nsegs)
    launch_adam(grads, t->mirrors_dev, t->var_off_dev, t->segs_dev + n0, t->nsegs - n0, t->norm2,
                grad_scale, clip, (float)lr_t, beta1, beta2, eps, t->m, t->v, sr);
  int rc = check_launch("adam_step");
  if (rc != N2NMN_OK) return rc;
  return commit_weights_on(c, s, sr);
}

int n2nmn_train_reset_optimizer(n2nmn_ctx* c, n2nmn_stream stream) {
  N2_REQUIRE(c && c->train, N2NMN_EINVAL, "train_reset_optimizer: training not enabled");
  // the late half of the previous Adam update (decoder / module variables with their m, v) may still
  // be running on the library's side stream: the memsets below are ordered behind it
  c->train->join(S(stream));
  N2_HIP(hipMemsetAsync(c->train->m, 0, sizeof(float) * (size_t)c->train->total, S(stream)));
  N2_HIP(hipMemsetAsync(c->train->v, 0, sizeof(float) * (size_t)c->train->total, S(stream)));
  return N2NMN_OK;
}

int n2nmn_debug_gemm_tn(n2nmn_ctx* ctx, const float* A, int lda, int M, const float* B, int ldb,
                        int N, int R, float* C, int ldc, const int32_t* a_row_idx,
                        const int32_t* a_onehot, const int32_t* b_sel, int b_sel_val,
                        n2nmn_stream stream) {
  N2_REQUIRE(ctx && B && C && (A || a_onehot), N2NMN_EINVAL, "debug_gemm_tn: null argument");
  N2_REQUIRE(M > 0 && N > 0 && R >= 0 && (a_onehot || (M % 4 == 0 && lda % 4 == 0)) && ldb % 4 == 0,
             N2NMN_EINVAL, "debug_gemm_tn: M, lda, ldb must be multiples of 4");
  GemmTnArgs g{};
  g.A = A; g.lda = lda; g.M = M; g.a_group_idx = a_row_idx; g.a_group_size = 1; g.a_onehot = a_onehot;
  g.B = B; g.ldb = ldb; g.N = N; g.b_sel = b_sel; g.b_sel_val = b_sel_val; g.R = R; g.C = C; g.ldc = ldc;
  launch_gemm_tn(g, S(stream));
  return check_launch("debug_gemm_tn");
}

int n2nmn_debug_colsum(n2nmn_ctx* ctx, const float* src, int R, int ncols, int ld,
                       const int32_t* sel, int sel_val, float* dst, n2nmn_stream stream) {
  N2_REQUIRE(ctx && src && dst && R >= 0 && ncols > 0 && ld >= ncols, N2NMN_EINVAL,
             "debug_colsum: bad argument");
  launch_colsum(src, R, ncols, ld, sel, sel_val, dst, S(stream));
  return check_launch("debug_colsum");
}

int64_t n2nmn_train_debug_tensor(n2nmn_ctx* c, const char* name, float* out, int64_t capacity,
                                 n2nmn_stream stream) {
  N2_REQUIRE(c && c->train && name && out, N2NMN_EINVAL, "train_debug_tensor: bad argument");
  TrainState* t = c->train;
  const n2nmn_dims& d = c->d;
  const int64_t N = t->last_N, T = t->last_T, Td = t->last_Td, L = d.lstm_dim;
  const std::string k(name);
  const float* src = nullptr;
  int64_t n = 0;
  if (k == "d_word_vecs") { src = t->dwv; n = Td * N * d.embed_dim_txt; }
  else if (k == "d_token_scores") { src = t->dsc; n = Td * N * 16; }
  else if (k == "d_encoder_outputs") { src = t->denc_out; n = T * N * L; }
  else if (k == "d_encoder_h_transformed") { src = t->deht; n = T * N * L; }
  else if (k == "d_scores") { src = t->dscores; n = N * d.num_choices; }
  else if (k == "d_dec_out") { src = t->dout; n = Td * N * L; }
  else if (k == "d_state") { src = t->dH0; n = 4 * (int64_t)d.N * L; }
  else {
    set_last_error("train_debug_tensor: unknown tensor '" + k + "'");
    return N2NMN_EKEY;
  }
  N2_REQUIRE(n <= capacity, N2NMN_ECAPACITY, "train_debug_tensor: capacity too small");
  N2_HIP(hipMemcpyAsync(out, src, sizeof(float) * n, hipMemcpyDeviceToDevice, S(stream)));
  return n;
}

}  // extern "C"
