// Layout-generator kernels: fused LSTM layer step, attentional decoder step, word vectors.
//
// Reference semantics: models_clevr/nmn3_netgen_att.py:73-113 (encoder), :115-322 (decoder);
// TF 1.0.0 BasicLSTMCell / dynamic_rnn / raw_rnn semantics per SURVEY.md Appendix A.1-A.3.
#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

// ---------------------------------------------------------------------------------------------
// lstm_step_kernel: one LSTM layer step (or two independent ones: grid.y = job).
//
//   z[n, :] = [x, h][n, :] . W + b ;  i, j, f, o = split(z) ;
//   c' = c * sig(f + 1) + sig(i) * tanh(j) ;  h' = tanh(c') * sig(o)        (Appendix A.1)
//
// The input projection x . W_x + b of layer 0 is a table lookup (the vocabulary is tiny:
// xtab[v] = emb[v] . W_x + b is computed once per weight commit), so the kernel only contracts
// over the recurrent K = L (layer 0) or K = 2L (layer 1: [h_below, h_own]).
//
// CDNA4 mapping: a workgroup owns 4 hidden units = 16 gate columns (one 16x16x4 fp32 MFMA
// N-tile, columns ordered gate-major in the packed weights) for 64 batch rows (4 M-tiles).
// Its 8 waves split K; each wave streams its K-slice of the weight tile (contiguous float4s of
// the k-interleaved pack) and of h straight from L2 into MFMA operand registers -- no LDS staging,
// because nothing is reused inside the workgroup.  The 8 partial 64x16 tiles are reduced through
// LDS and the gate nonlinearities + state update run in the same kernel, so z never exists in HBM.
// grid = (L/4 column tiles, jobs, row blocks of 64): 128..256 workgroups per launch.
// ---------------------------------------------------------------------------------------------
constexpr int LSTM_WAVES = 8;
constexpr int LSTM_THREADS = LSTM_WAVES * 64;

struct LstmJobs {
  LstmJob j[2];
};

__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_kernel(LstmJobs jobs, int N, int L) {
  const LstmJob& jb = jobs.j[blockIdx.y];
  if (!jb.active) return;
  __shared__ float part[LSTM_WAVES][64][17];

  const int tile = blockIdx.x;
  if (tile >= jb.ntiles) return;
  const int row0 = blockIdx.z * 64;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = lane & 15, kg = lane >> 4;
  const int K = jb.K;
  const int kslice = K / LSTM_WAVES;
  const int kbeg = w * kslice;
  const float* Asrc = (kbeg < L) ? jb.A0 : jb.A1;
  const int kloc = (kbeg < L) ? kbeg : kbeg - L;
  const float4* Wp4 = reinterpret_cast<const float4*>(jb.Wp) + (size_t)tile * (K / 4) * 16;

  const float* arow[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int r = row0 + 16 * m + ci;
    r = r < N ? r : N - 1;
    arow[m] = Asrc + (size_t)r * L + kloc + 4 * kg;
  }

  f32x4 acc[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nchunk = kslice / 16;
#pragma unroll 4
  for (int kc = 0; kc < nchunk; ++kc) {
    const float4 bq = Wp4[(size_t)((kbeg >> 2) + 4 * kc + kg) * 16 + ci];
    float4 aq[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) aq[m] = *reinterpret_cast<const float4*>(arow[m] + 16 * kc);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].x, bq.x, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].y, bq.y, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].z, bq.z, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[m].w, bq.w, acc[m], 0, 0, 0);
    }
  }
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][16 * m + 4 * kg + r][ci] = acc[m][r];
  __syncthreads();

  if (tid < 256) {
    const int row = tid >> 2, ul = tid & 3;
    const int gr = row0 + row;
    if (gr < N) {
      float z[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float s = 0.f;
#pragma unroll
        for (int ww = 0; ww < LSTM_WAVES; ++ww) s += part[ww][row][g * 4 + ul];
        z[g] = s;
      }
      if (jb.mode == 1) {                       // plain linear: out = z + bias
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int col = 16 * tile + g * 4 + ul;
          jb.h_new[(size_t)gr * jb.ldo + col] = z[g] + (jb.bias ? jb.bias[col] : 0.f);
        }
        return;
      }
      const int u = 4 * tile + ul;
      if (jb.xtab) {
        const int xi = jb.xidx ? jb.xidx[gr] : jb.xidx_const;
        const float* xr = jb.xtab + (size_t)xi * 4 * L + u;
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] += xr[g * L];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] += jb.bias[g * L + u];
      }
      const size_t idx = (size_t)gr * L + u;
      const float c_old = jb.c_in[idx];
      float c_new = c_old * sigmoidf_(z[2] + 1.0f) + sigmoidf_(z[0]) * tanhf(z[1]);
      float h_new = tanhf(c_new) * sigmoidf_(z[3]);
      float o = h_new;
      if (jb.seq_len && jb.t >= jb.seq_len[gr]) {   // dynamic_rnn past the length (A.2)
        c_new = c_old;
        h_new = jb.h_old[idx];
        o = 0.f;
      }
      jb.c_out[idx] = c_new;
      jb.h_new[idx] = h_new;
      if (jb.out_seq) jb.out_seq[idx] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dec_step_kernel: everything of one decoder step after the LSTM cell, one workgroup per
// question (nmn3_netgen_att.py:184-268):
//   additive attention over the encoder steps, masked renormalised softmax, context vector,
//   token logits, validity automaton (int32), greedy / sampled / teacher-forced choice,
//   token probability, entropy term, automaton update.
// q = out . W_a + b_a comes from gemm_pk.  eht / eout rows of this question (2 x T x L fp32) are
// streamed with float4 loads; v, q, out, the attention row and the context live in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int DEC_THREADS = 256;
constexpr int MAXV = 16;

__global__ __launch_bounds__(DEC_THREADS) void dec_step_kernel(DecStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int L = a.L, T = a.T, N = a.N, V = a.V;
  float* qs = smem;             // [L]
  float* outs = qs + L;         // [L]
  float* ctx = outs + L;        // [L]
  float* ctxp = ctx + L;        // [2][L] partial contexts
  float* es = ctxp + 2 * L;     // [T] logits -> attention
  float* red = es + ((T + 3) & ~3);   // [4][MAXV] + scratch
  const int n = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int len = a.seq_len[n];

  for (int k = tid; k < L; k += DEC_THREADS) {
    qs[k] = a.q[(size_t)n * L + k];
    outs[k] = a.out[(size_t)n * L + k];
  }
  __syncthreads();

  // ---- e[tau] = sum_k v_k tanh(q_k + eht[tau, n, k])                               (:184-187)
  for (int tau = w; tau < T; tau += DEC_THREADS / 64) {
    const float* er = a.eht + ((size_t)tau * N + n) * L;
    float s = 0.f;
    for (int k = 4 * lane; k < L; k += 256) {
      const float4 e4 = *reinterpret_cast<const float4*>(er + k);
      const float4 v4 = *reinterpret_cast<const float4*>(a.v + k);
      s += v4.x * tanhf(qs[k] + e4.x) + v4.y * tanhf(qs[k + 1] + e4.y) +
           v4.z * tanhf(qs[k + 2] + e4.z) + v4.w * tanhf(qs[k + 3] + e4.w);
    }
    s = wave_sum(s);
    if (lane == 0) es[tau] = s;
  }
  __syncthreads();

  // ---- softmax over ALL T rows, mask finished rows, renormalise                    (:190-191)
  if (w == 0) {
    float m = -INFINITY;
    for (int tau = lane; tau < T; tau += 64) m = fmaxf(m, es[tau]);
    m = wave_max(m);
    float s = 0.f;
    for (int tau = lane; tau < T; tau += 64) s += expf(es[tau] - m);
    s = wave_sum(s);
    float s2 = 0.f;
    for (int tau = lane; tau < T; tau += 64) {
      float p = expf(es[tau] - m) / s;
      p = tau < len ? p : 0.f;
      es[tau] = p;
      s2 += p;
    }
    s2 = wave_sum(s2);
    for (int tau = lane; tau < T; tau += 64) {
      const float att = es[tau] / s2;
      es[tau] = att;
      a.atts[(size_t)tau * N + n] = att;
    }
  }
  __syncthreads();

  // ---- ctx = sum_tau att[tau] * eout[tau, n, :]                                     (:193)
  {
    const int ncol = L / 4;
    const int nsplit = (2 * ncol <= DEC_THREADS) ? 2 : 1;
    for (int c = tid; c < ncol * nsplit; c += DEC_THREADS) {
      const int col = c % ncol, sp = c / ncol;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int tau = sp; tau < len; tau += nsplit) {
        const float at = es[tau];
        const float4 o4 =
            *reinterpret_cast<const float4*>(a.eout + ((size_t)tau * N + n) * L + 4 * col);
        acc.x += at * o4.x; acc.y += at * o4.y; acc.z += at * o4.z; acc.w += at * o4.w;
      }
      *reinterpret_cast<float4*>(ctxp + sp * L + 4 * col) = acc;
    }
    __syncthreads();
    for (int k = tid; k < L; k += DEC_THREADS)
      ctx[k] = (nsplit == 2) ? ctxp[k] + ctxp[L + k] : ctxp[k];
    __syncthreads();
  }

  // ---- token logits = [out, ctx] . W_y + b_y                                        (:196-198)
  {
    float ps[MAXV];
#pragma unroll
    for (int s = 0; s < MAXV; ++s) ps[s] = 0.f;
    for (int k = tid; k < 2 * L; k += DEC_THREADS) {
      const float x = k < L ? outs[k] : ctx[k - L];
      const float* wr = a.Wy + (size_t)k * V;
#pragma unroll
      for (int s = 0; s < MAXV; ++s)
        if (s < V) ps[s] += x * wr[s];
    }
#pragma unroll
    for (int s = 0; s < MAXV; ++s) {
      const float r = wave_sum(ps[s]);
      if (lane == 0) red[w * MAXV + s] = r;
    }
  }
  __syncthreads();

  // ---- validity, choice, probabilities: lanes 0..V-1 of wave 0                      (:200-268)
  if (w == 0) {
    const bool on = lane < V;
    float sc = -INFINITY;
    if (on) {
      sc = a.by[lane];
      for (int ww = 0; ww < DEC_THREADS / 64; ++ww) sc += red[ww * MAXV + lane];
      if (a.scores) a.scores[(size_t)n * V + lane] = sc;
    }
    const int x0 = a.state[n * 3 + 0], x1 = a.state[n * 3 + 1], x2 = a.state[n * 3 + 2];
    bool valid = false;
    if (on) {
      valid = true;
      for (int c = 0; c < 4; ++c) {          // all_c( X . W[:, s, c] - b[s, c] >= 0 )   (:8-11)
        const int val = x0 * a.Wv[(0 * V + lane) * 4 + c] + x1 * a.Wv[(1 * V + lane) * 4 + c] +
                        x2 * a.Wv[(2 * V + lane) * 4 + c] - a.bv[lane * 4 + c];
        valid = valid && (val >= 0);
      }
      if (a.use_gt) valid = true;            // logical_or(valid, use_gt_layout)         (:204-207)
    }
    // greedy: first index of the maximum over valid tokens (invalid ones sit at min-1)  (:234-238)
    const float key = (on && valid) ? sc : -INFINITY;
    const float kmax = wave_max(key);
    const unsigned long long hit = __ballot(on && valid && key == kmax);
    int tok = hit ? (int)__builtin_ctzll(hit) : 0;
    if (a.uni) {                             // sampling with a caller-supplied uniform   (:212-232)
      const float sv = on ? sc - (valid ? 0.f : 50.f) : -INFINITY;
      const float mx = wave_max(sv);
      const float ex = on ? expf(sv - mx) : 0.f;
      const float den = wave_sum(ex);
      const float ps = ex / den;
      // inclusive scan over V <= 16 lanes, sequential order
      float cdf = 0.f, tot = 0.f;
      for (int s = 0; s < V; ++s) {
        const float v = __shfl(ps, s, 64);
        tot += v;
        if (s == lane) cdf = tot;
      }
      const float thr = a.uni[n] * tot;
      const unsigned long long le = __ballot(on && cdf <= thr);
      int samp = __builtin_popcountll(le);
      samp = samp < V - 1 ? samp : V - 1;
      const bool ok = (__ballot(on && valid) >> samp) & 1ull;
      tok = ok ? samp : tok;
    }
    if (a.use_gt && a.gt) tok = a.gt[n];     // (:239-241)
    if (a.forced) tok = a.forced[n];
    // robust softmax restricted to valid tokens                                        (:245-260)
    const float mx = wave_max(sc);
    const float ex = on ? expf(sc - mx) : 0.f;
    const float den = wave_sum(ex);
    float p = (on && valid) ? ex / den : 0.f;
    const float psum = wave_sum(p);
    p = p / psum;
    const float tp = __shfl(p, tok, 64);
    float ent = on ? p * logf(fmaxf(1e-5f, p + (valid ? 0.f : 1.f))) : 0.f;
    ent = wave_sum(ent);
    if (lane == 0) {
      a.tokens[n] = tok;
      a.tprobs[n] = tp;
      a.neg_entropy[n] += ent;
      a.next_idx[n] = tok;
      a.state[n * 3 + 0] = x0 + a.P[tok * 3 + 0];          // X += P[token]            (:13-15)
      a.state[n * 3 + 1] = x1 + a.P[tok * 3 + 1];
      a.state[n * 3 + 2] = x2 + a.P[tok * 3 + 2];
    }
  }
}

__global__ void dec_init_kernel(int32_t* state, float* neg_entropy, int N, int T_dec) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) {
    state[n * 3 + 0] = 0; state[n * 3 + 1] = 0; state[n * 3 + 2] = T_dec;     // (:284)
    neg_entropy[n] = 0.f;
  }
}

// word_vecs[t, n, :] = sum_tau atts[t, tau, n] * emb[seq[tau, n], :]   (nmn3_netgen_att.py:312)
// one workgroup per question: the question's T_enc embedding rows are staged once in LDS.
__global__ __launch_bounds__(256) void word_vecs_kernel(const float* __restrict__ atts,
                                                        const int32_t* __restrict__ seq,
                                                        const float* __restrict__ emb, int T_dec,
                                                        int T_enc, int N, int E,
                                                        float* __restrict__ wv,
                                                        const float* __restrict__ tprobs,
                                                        float* __restrict__ log_seq_prob) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* rows = smem;                       // [T_enc][E]
  float* at = rows + (size_t)T_enc * E;     // [T_dec][T_enc]
  const int n = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < T_enc * E; i += 256) {
    const int tau = i / E, e = i - tau * E;
    rows[i] = emb[(size_t)seq[tau * N + n] * E + e];
  }
  for (int i = tid; i < T_dec * T_enc; i += 256) {
    const int t = i / T_enc, tau = i - t * T_enc;
    at[i] = atts[((size_t)t * T_enc + tau) * N + n];
  }
  __syncthreads();
  for (int i = tid; i < T_dec * E; i += 256) {
    const int t = i / E, e = i - t * E;
    float s = 0.f;
    for (int tau = 0; tau < T_enc; ++tau) s += at[t * T_enc + tau] * rows[tau * E + e];
    wv[((size_t)t * N + n) * E + e] = s;
  }
  if (log_seq_prob && tid == 0) {           // models_clevr/nmn3_model.py:46
    float s = 0.f;
    for (int t = 0; t < T_dec; ++t) s += logf(tprobs[t * N + n]);
    log_seq_prob[n] = s;
  }
}

}  // namespace

void launch_lstm_step(const LstmJob* jobs, int njobs, int N, int L, hipStream_t s) {
  LstmJobs js;
  for (int i = 0; i < 2; ++i) {
    if (i < njobs) js.j[i] = jobs[i];
    else { js.j[i] = LstmJob{}; js.j[i].active = 0; }
  }
  int nt = 0;
  for (int i = 0; i < njobs; ++i) nt = jobs[i].ntiles > nt ? jobs[i].ntiles : nt;
  dim3 grid(nt, njobs, (N + 63) / 64);
  hipLaunchKernelGGL(lstm_step_kernel, grid, dim3(LSTM_THREADS), 0, s, js, N, L);
}

void launch_dec_step(const DecStepArgs& a, hipStream_t s) {
  const size_t smem = sizeof(float) * (5 * (size_t)a.L + ((a.T + 3) & ~3) + 4 * MAXV + 16);
  hipLaunchKernelGGL(dec_step_kernel, dim3(a.N), dim3(DEC_THREADS), smem, s, a);
}

void launch_dec_init(int32_t* state, float* neg_entropy, int N, int T_dec, hipStream_t s) {
  hipLaunchKernelGGL(dec_init_kernel, dim3((N + 63) / 64), dim3(64), 0, s, state, neg_entropy, N,
                     T_dec);
}

void launch_word_vecs(const float* atts, const int32_t* seq, const float* emb, int T_dec,
                      int T_enc, int N, int E, float* word_vecs, const float* tprobs,
                      float* log_seq_prob, hipStream_t s) {
  const size_t smem = sizeof(float) * ((size_t)T_enc * E + (size_t)T_dec * T_enc);
  hipLaunchKernelGGL(word_vecs_kernel, dim3(N), dim3(256), smem, s, atts, seq, emb, T_dec, T_enc,
                     N, E, word_vecs, tprobs, log_seq_prob);
}

}  // namespace n2nmn
