// Layout-generator kernels: fused LSTM layer step, attentional decoder step, word vectors.
//
// Reference semantics: models_clevr/nmn3_netgen_att.py:73-113 (encoder), :115-322 (decoder);
// TF 1.0.0 BasicLSTMCell / dynamic_rnn / raw_rnn semantics per SURVEY.md Appendix A.1-A.3.
#include <algorithm>
#include <cstdlib>

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
// N-tile, columns ordered gate-major in the packed weights) for 16*MT batch rows (MT M-tiles).
// Its 8 waves split K; each wave streams its K-slice of the weight tile (contiguous float4s of
// the k-interleaved pack) and of h straight from L2 into MFMA operand registers -- no LDS staging,
// because nothing is reused inside the workgroup.  The K-slice is a compile-time number of
// 16-wide chunks so that ALL operand loads of a wave are in flight before its first MFMA (one
// L2 round trip per step instead of one per chunk), and the operands of the pointwise epilogue
// (c, x-table row, bias, h_old) are fetched at kernel entry as well.  The 8 partial tiles are
// reduced through LDS and the gate nonlinearities + state update run in the same kernel, so z
// never exists in HBM.  grid = (column tiles, jobs, row blocks).
// ---------------------------------------------------------------------------------------------
constexpr int LSTM_WAVES = 8;
constexpr int LSTM_THREADS = LSTM_WAVES * 64;

struct LstmJobs {
  LstmJob j[2];
};

// DBG: 0 normal, 1 = pin all loads before the MFMAs (sched_barrier), 2 = loads only (no MFMA),
// 3 = MFMA only (no loads), 4 = neither, 6 = weight loads only, 7 = state loads only
// -- variants 1..7 exist for n2nmn_debug_lstm_bench.
// NA = number of leading 16-row M-tiles of this workgroup that hold active rows (the length-sorted
// encoder skips the rest); the code for a given NA is straight-line so the scheduler can interleave
// the chunk-major loads with the MFMAs.
template <int NCH, int NA, int DBG = 0>
__device__ __forceinline__ void lstm_mma(const LstmJob& jb, int N, int L, int tile, int row0,
                                         f32x4* acc) {
  const int lane = threadIdx.x & 63;
  // stagger: neighbouring column tiles walk the K slices and chunks in rotated order, so the
  // 128 workgroups that all read the same h rows do not hit the same L2 channel at the same time
  const int w = ((threadIdx.x >> 6) + tile) & (LSTM_WAVES - 1);
  const int rot = (tile >> 3) & (NCH - 1);
  const int ci = lane & 15, kg = lane >> 4;
  const int K = jb.K;
  const int kbeg = w * (NCH * 16);
  const float* Asrc = (kbeg < L) ? jb.A0 : jb.A1;
  const int kloc = (kbeg < L) ? kbeg : kbeg - L;
  const float4* Wp4 = reinterpret_cast<const float4*>(jb.Wp) + (size_t)tile * (K / 4) * 16 +
                      (size_t)((kbeg >> 2) + kg) * 16 + ci;
  float4 bq[NCH];
  float4 aq[NCH][NA];
  const float* ar[NA];
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    int r = row0 + 16 * m + ci;
    r = r < N ? r : N - 1;
    ar[m] = Asrc + (size_t)r * jb.a_rs + (size_t)((kloc >> 2) + kg) * jb.a_ks;
  }
  const float f = (float)lane * 1e-3f;
  // chunk-major issue order: the operand loads of chunk 0 go first, so its MFMAs can start after
  // one round trip while the later chunks are still in flight
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    const int kq = (kc + rot) & (NCH - 1);
    if (DBG == 3 || DBG == 4 || DBG == 7) bq[kc] = make_float4(f, f + 1.f, f + 2.f, f + 3.f);
    else bq[kc] = Wp4[(size_t)kq * 64];
#pragma unroll
    for (int m = 0; m < NA; ++m) {
      if (DBG == 3 || DBG == 4 || DBG == 6) aq[kc][m] = make_float4(f, f - 1.f, f - 2.f, f - 3.f);
      else aq[kc][m] = *reinterpret_cast<const float4*>(ar[m] + (size_t)(4 * kq) * jb.a_ks);
    }
  }
  if (DBG == 1) __builtin_amdgcn_sched_barrier(0);   // all loads issued before the first MFMA
  if (DBG == 2 || DBG == 4 || DBG == 6 || DBG == 7) {  // no MFMA: fold the operands so they stay live
#pragma unroll
    for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
      for (int m = 0; m < NA; ++m) {
        acc[m][0] += aq[kc][m].x * bq[kc].x; acc[m][1] += aq[kc][m].y * bq[kc].y;
        acc[m][2] += aq[kc][m].z * bq[kc].z; acc[m][3] += aq[kc][m].w * bq[kc].w;
      }
    return;
  }
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
#pragma unroll
    for (int m = 0; m < NA; ++m) {
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].x, bq[kc].x, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].y, bq[kc].y, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].z, bq[kc].z, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].w, bq[kc].w, acc[m], 0, 0, 0);
    }
  }
}

template <int NCH, int MT, int DBG>
__device__ __forceinline__ void lstm_mma_dispatch(const LstmJob& jb, int N, int L, int tile,
                                                  int row0, int na, f32x4* acc) {
  if (MT == 4) {
    switch (na) {
      case 4: lstm_mma<NCH, 4, DBG>(jb, N, L, tile, row0, acc); break;
      case 3: lstm_mma<NCH, 3, DBG>(jb, N, L, tile, row0, acc); break;
      case 2: lstm_mma<NCH, 2, DBG>(jb, N, L, tile, row0, acc); break;
      case 1: lstm_mma<NCH, 1, DBG>(jb, N, L, tile, row0, acc); break;
      default: break;
    }
  } else {
    switch (na) {
      case 2: lstm_mma<NCH, 2, DBG>(jb, N, L, tile, row0, acc); break;
      case 1: lstm_mma<NCH, 1, DBG>(jb, N, L, tile, row0, acc); break;
      default: break;
    }
  }
}

template <int MT, int DBG = 0>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_kernel(LstmJobs jobs, int N, int L) {
  const LstmJob& jb = jobs.j[blockIdx.y];
  if (!jb.active) return;
  const int tile = blockIdx.x;
  if (tile >= jb.ntiles) return;
  __shared__ float part[LSTM_WAVES][16 * MT][17];

  constexpr int ROWS = 16 * MT;
  const int row0 = blockIdx.z * ROWS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = lane & 15, kg = lane >> 4;

  const int nact = jb.n_active ? *jb.n_active : N;   // rows [nact, N) are past their length
  // ---- epilogue operands, fetched up front (thread = (row, unit) for tid < 4*ROWS) ----------
  const int erow = tid >> 2, ul = tid & 3;
  const int gr = row0 + erow;
  const bool eact = tid < 4 * ROWS && gr < N;
  const int orow = (eact && jb.perm) ? jb.perm[gr] : gr;   // original row of state row gr
  if (row0 >= nact && jb.mode == 0 && !jb.save_gates) {
    // Every row of this tile is past its length (length-sorted encoder).  Such rows are never read
    // again: the recurrence only contracts over rows of tiles that are still active, c is updated
    // in place, and a row's final state was captured at its last valid step (fin_c / fin_h).  All
    // that is left of dynamic_rnn's semantics (Appendix A.2) is the zero output row.
    if (eact && jb.out_seq) jb.out_seq[(size_t)orow * L + 4 * tile + ul] = 0.f;
    return;
  }
  float add[4] = {0.f, 0.f, 0.f, 0.f};
  float c_old = 0.f, h_prev = 0.f;
  bool masked = false;
  if (eact) {
    if (jb.mode == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) add[g] = jb.bias ? jb.bias[16 * tile + g * 4 + ul] : 0.f;
    } else {
      const int u = 4 * tile + ul;
      if (jb.xtab) {
        const int xi = jb.xidx ? jb.xidx[orow] : jb.xidx_const;
        const float* xr = jb.xtab + (size_t)xi * 4 * L + 16 * tile + ul;   // tile column order
#pragma unroll
        for (int g = 0; g < 4; ++g) add[g] = xr[4 * g];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) add[g] = jb.bias[16 * tile + 4 * g + ul];   // tile order
      }
      const size_t idx = (size_t)gr * L + u;
      // the cell state shares the k-interleaved layout of h: a tile's 64 rows x 4 units are one
      // contiguous KB instead of 64 sixteen-byte pieces of 64 different lines
      const size_t sidx = jb.hp_R > 0 ? ((size_t)tile * jb.hp_R + gr) * 4 + ul : idx;
      c_old = jb.c_in[sidx];
      if (jb.seq_len && jb.t >= jb.seq_len[orow]) { // dynamic_rnn past the length (A.2)
        masked = true;
        h_prev = jb.h_old[sidx];
      }
    }
  }

  // ---- K-split MFMA ------------------------------------------------------------------------
  f32x4 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nch = jb.K / (LSTM_WAVES * 16);
  // leading M-tiles with at least one active row (wave-uniform)
  int na = (nact - row0 + 15) >> 4;
  na = na < 0 ? 0 : (na > MT ? MT : na);
  if (nch == 4) lstm_mma_dispatch<4, MT, DBG>(jb, N, L, tile, row0, na, acc);
  else if (nch == 8) lstm_mma_dispatch<8, MT, DBG>(jb, N, L, tile, row0, na, acc);
  else {
    for (int q = 0; q < nch; ++q) {               // generic K: one chunk at a time
      const int kbeg = w * nch * 16 + 16 * q;
      const float* Asrc = (kbeg < L) ? jb.A0 : jb.A1;
      const int kloc = (kbeg < L) ? kbeg : kbeg - L;
      const float4 bq = (reinterpret_cast<const float4*>(jb.Wp) +
                         (size_t)tile * (jb.K / 4) * 16)[(size_t)((kbeg >> 2) + kg) * 16 + ci];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        int r = row0 + 16 * m + ci;
        r = r < N ? r : N - 1;
        const float4 aq = *reinterpret_cast<const float4*>(
            Asrc + (size_t)r * jb.a_rs + (size_t)((kloc >> 2) + kg) * jb.a_ks);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.x, bq.x, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.y, bq.y, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.z, bq.z, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.w, bq.w, acc[m], 0, 0, 0);
      }
    }
  }
  // C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[w][16 * m + 4 * kg + r][ci] = acc[m][r];
  __syncthreads();

  if (eact) {
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = add[g];
#pragma unroll
      for (int ww = 0; ww < LSTM_WAVES; ++ww) s += part[ww][erow][g * 4 + ul];
      z[g] = s;
    }
    if (jb.mode == 1) {                           // plain linear: out = z + bias
#pragma unroll
      for (int g = 0; g < 4; ++g)
        jb.h_new[(size_t)gr * jb.ldo + 16 * tile + g * 4 + ul] = z[g];
      return;
    }
    const size_t idx = (size_t)gr * L + 4 * tile + ul;
    const float gi = fast_sigmoid(z[0]), gj = fast_tanh(z[1]), gf = fast_sigmoid(z[2] + 1.0f),
                go = fast_sigmoid(z[3]);
    float c_new = c_old * gf + gi * gj;
    float h_new = fast_tanh(c_new) * go;
    float o = h_new;
    if (masked) { c_new = c_old; h_new = h_prev; o = 0.f; }
    const size_t sidx = jb.hp_R > 0 ? ((size_t)tile * jb.hp_R + gr) * 4 + ul : idx;
    jb.c_out[sidx] = c_new;
    jb.h_new[sidx] = h_new;
    // split-operand bf16 mode: the tail steps of a pass run here (exact fp32) but later steps / the decoder
    // read the bf16 planes of what they write
    if (jb.h_new_b && jb.hp_R > 0) store_plane1(jb.h_new_b, L, jb.hp_R, 4 * tile + ul, gr, h_new);
    const size_t oidx = (size_t)orow * L + 4 * tile + ul;
    if (jb.save_gates) {          // training: keep what the cell backward needs (masked rows keep
      jb.save_gates[oidx] = make_float4(gi, gj, gf, go);   // finite values; their dz is zero)
      jb.save_c[oidx] = c_new;
      jb.save_h[oidx] = h_new;
    }
    if (jb.out_seq) jb.out_seq[oidx] = o;
    if (jb.h_drop) {              // dropped copy of the OUTPUT for the layer above
      const float hd = h_new * jb.drop[oidx];
      jb.h_drop[jb.hp_R > 0 ? ((size_t)tile * jb.hp_R + gr) * 4 + ul : idx] = hd;
      if (jb.h_drop_b && jb.hp_R > 0) store_plane1(jb.h_drop_b, L, jb.hp_R, 4 * tile + ul, gr, hd);
      if (jb.save_hd) jb.save_hd[oidx] = hd;
    }
    if (jb.fin_c && jb.seq_len && jb.t == jb.seq_len[orow] - 1) {   // the row's last valid step
      jb.fin_c[((size_t)tile * jb.hp_R + orow) * 4 + ul] = c_new;
      jb.fin_h[((size_t)tile * jb.hp_R + orow) * 4 + ul] = h_new;
      if (jb.fin_h_b) store_plane1(jb.fin_h_b, L, jb.hp_R, 4 * tile + ul, orow, h_new);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// lstm_step_wide_kernel: the same step with a 32-row x 32-gate-column (8 hidden units) workgroup
// tile instead of 64 x 16.  Same number of workgroups and MFMAs per workgroup, but the operand
// bytes a workgroup streams scale with rows + columns (64 vs 80 per k), i.e. 20 % less L2 traffic;
// a wave issues 4 operand loads per 16 MFMAs instead of 5.  LSTM cell jobs only (mode 0).
// (A 64 x 32 tile -- half as many, fatter workgroups -- was measured too: 110.7 k questions/s with 6
// batches in flight against 131.1 k for 32 x 32 and 127.8 k for 64 x 16.)
// ---------------------------------------------------------------------------------------------
template <int NCH, int NA>
__device__ __forceinline__ void lstm_mma_wide(const LstmJob& jb, int N, int L, int pair, int row0,
                                              f32x4 (*acc)[2]) {
  const int lane = threadIdx.x & 63;
  const int w = ((threadIdx.x >> 6) + pair) & (LSTM_WAVES - 1);
  const int rot = (pair >> 3) & (NCH - 1);
  const int ci = lane & 15, kg = lane >> 4;
  const int K = jb.K;
  const int kbeg = w * (NCH * 16);
  const float* Asrc = (kbeg < L) ? jb.A0 : jb.A1;
  const int kloc = (kbeg < L) ? kbeg : kbeg - L;
  const size_t tstride = (size_t)(K / 4) * 16;          // float4 per column tile
  const float4* Wp4 = reinterpret_cast<const float4*>(jb.Wp) + (size_t)(2 * pair) * tstride +
                      (size_t)((kbeg >> 2) + kg) * 16 + ci;
  float4 bq[NCH][2];
  float4 aq[NCH][NA];
  const float* ar[NA];
#pragma unroll
  for (int m = 0; m < NA; ++m) {
    int r = row0 + 16 * m + ci;
    r = r < N ? r : N - 1;
    ar[m] = Asrc + (size_t)r * jb.a_rs + (size_t)((kloc >> 2) + kg) * jb.a_ks;
  }
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
    const int kq = (kc + rot) & (NCH - 1);
    bq[kc][0] = Wp4[(size_t)kq * 64];
    bq[kc][1] = Wp4[tstride + (size_t)kq * 64];
#pragma unroll
    for (int m = 0; m < NA; ++m)
      aq[kc][m] = *reinterpret_cast<const float4*>(ar[m] + (size_t)(4 * kq) * jb.a_ks);
  }
#pragma unroll
  for (int kc = 0; kc < NCH; ++kc) {
#pragma unroll
    for (int m = 0; m < NA; ++m) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].x, bq[kc][j].x, acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].y, bq[kc][j].y, acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].z, bq[kc][j].z, acc[m][j], 0, 0, 0);
        acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[kc][m].w, bq[kc][j].w, acc[m][j], 0, 0, 0);
      }
    }
  }
}

template <int MT>
__global__ __launch_bounds__(LSTM_THREADS) void lstm_step_wide_kernel(LstmJobs jobs, int N, int L) {
  const LstmJob& jb = jobs.j[blockIdx.y];
  if (!jb.active) return;
  const int pair = blockIdx.x;                  // column tiles 2*pair, 2*pair + 1
  if (2 * pair >= jb.ntiles) return;
  constexpr int ROWS = 16 * MT;
  __shared__ float part[LSTM_WAVES][ROWS][33];
  const int row0 = blockIdx.z * ROWS;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ci = lane & 15, kg = lane >> 4;
  const int nact = jb.n_active ? *jb.n_active : N;

  // ---- epilogue operands (thread = (row, unit) for tid < 8*ROWS) --------------------------------
  const int erow = tid >> 3, ul8 = tid & 7;
  const int tloc = ul8 >> 2, ul = ul8 & 3;
  const int tile = 2 * pair + tloc;
  const int gr = row0 + erow;
  const bool eact = tid < 8 * ROWS && gr < N;
  const int orow = (eact && jb.perm) ? jb.perm[gr] : gr;
  if (row0 >= nact && !jb.save_gates) {        // see lstm_step_kernel: only the zero output row is left
    if (eact && jb.out_seq) jb.out_seq[(size_t)orow * L + 4 * tile + ul] = 0.f;
    return;
  }
  float add[4] = {0.f, 0.f, 0.f, 0.f};
  float c_old = 0.f, h_prev = 0.f;
  bool masked = false;
  if (eact) {
    const int u = 4 * tile + ul;
    if (jb.xtab) {
      const int xi = jb.xidx ? jb.xidx[orow] : jb.xidx_const;
      const float* xr = jb.xtab + (size_t)xi * 4 * L + 16 * tile + ul;     // tile column order
#pragma unroll
      for (int g = 0; g < 4; ++g) add[g] = xr[4 * g];
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) add[g] = jb.bias[16 * tile + 4 * g + ul];     // tile order
    }
    const size_t idx = (size_t)gr * L + u;
    const size_t sidx = jb.hp_R > 0 ? ((size_t)tile * jb.hp_R + gr) * 4 + ul : idx;
    c_old = jb.c_in[sidx];
    if (jb.seq_len && jb.t >= jb.seq_len[orow]) {
      masked = true;
      h_prev = jb.h_old[sidx];
    }
  }

  f32x4 acc[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[m][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nch = jb.K / (LSTM_WAVES * 16);
  int na = (nact - row0 + 15) >> 4;
  na = na < 0 ? 0 : (na > MT ? MT : na);
  if (nch == 4) {
    if (MT == 4 && na == 4) lstm_mma_wide<4, 4>(jb, N, L, pair, row0, acc);
    else if (MT == 4 && na == 3) lstm_mma_wide<4, 3>(jb, N, L, pair, row0, acc);
    else if (na == 2) lstm_mma_wide<4, 2>(jb, N, L, pair, row0, acc);
    else if (na == 1) lstm_mma_wide<4, 1>(jb, N, L, pair, row0, acc);
  } else {
    if (MT == 4 && na == 4) lstm_mma_wide<8, 4>(jb, N, L, pair, row0, acc);
    else if (MT == 4 && na == 3) lstm_mma_wide<8, 3>(jb, N, L, pair, row0, acc);
    else if (na == 2) lstm_mma_wide<8, 2>(jb, N, L, pair, row0, acc);
    else if (na == 1) lstm_mma_wide<8, 1>(jb, N, L, pair, row0, acc);
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) part[w][16 * m + 4 * kg + r][16 * j + ci] = acc[m][j][r];
  __syncthreads();

  if (eact) {
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float s = add[g];
#pragma unroll
      for (int ww = 0; ww < LSTM_WAVES; ++ww) s += part[ww][erow][16 * tloc + g * 4 + ul];
      z[g] = s;
    }
    const size_t idx = (size_t)gr * L + 4 * tile + ul;
    const float gi = fast_sigmoid(z[0]), gj = fast_tanh(z[1]), gf = fast_sigmoid(z[2] + 1.0f),
                go = fast_sigmoid(z[3]);
    float c_new = c_old * gf + gi * gj;
    float h_new = fast_tanh(c_new) * go;
    float o = h_new;
    if (masked) { c_new = c_old; h_new = h_prev; o = 0.f; }
    const size_t sidx = jb.hp_R > 0 ? ((size_t)tile * jb.hp_R + gr) * 4 + ul : idx;
    jb.c_out[sidx] = c_new;
    jb.h_new[sidx] = h_new;
    // split-operand bf16 mode: the tail steps of a pass run here (exact fp32) but later steps / the decoder
    // read the bf16 planes of what they write
    if (jb.h_new_b && jb.hp_R > 0) store_plane1(jb.h_new_b, L, jb.hp_R, 4 * tile + ul, gr, h_new);
    const size_t oidx = (size_t)orow * L + 4 * tile + ul;
    if (jb.save_gates) {
      jb.save_gates[oidx] = make_float4(gi, gj, gf, go);
      jb.save_c[oidx] = c_new;
      jb.save_h[oidx] = h_new;
    }
    if (jb.out_seq) jb.out_seq[oidx] = o;
    if (jb.h_drop) {              // dropped copy of the OUTPUT for the layer above
      const float hd = h_new * jb.drop[oidx];
      jb.h_drop[jb.hp_R > 0 ? ((size_t)tile * jb.hp_R + gr) * 4 + ul : idx] = hd;
      if (jb.h_drop_b && jb.hp_R > 0) store_plane1(jb.h_drop_b, L, jb.hp_R, 4 * tile + ul, gr, hd);
      if (jb.save_hd) jb.save_hd[oidx] = hd;
    }
    if (jb.fin_c && jb.seq_len && jb.t == jb.seq_len[orow] - 1) {
      jb.fin_c[((size_t)tile * jb.hp_R + orow) * 4 + ul] = c_new;
      jb.fin_h[((size_t)tile * jb.hp_R + orow) * 4 + ul] = h_new;
      if (jb.fin_h_b) store_plane1(jb.fin_h_b, L, jb.hp_R, 4 * tile + ul, orow, h_new);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// dec_attn_kernel: everything of a decoder step after the LSTM cell, one workgroup per
// (question n, step t = blockIdx.y) (nmn3_netgen_att.py:184-268):
//   additive attention over the encoder steps, masked renormalised softmax, context vector,
//   token logits, validity automaton (int32), greedy / sampled / teacher-forced choice,
//   token probability, entropy term, automaton update.
// Sequential decoding launches it with grid.y = 1 per step (NT = 1024 threads to cut the
// per-step latency); teacher-forced decoding knows every token up front, so ALL T_dec steps run
// in ONE launch (grid.y = T_dec, NT = 256).  q = out . W_a + b_a comes from the linear mode of
// lstm_step_kernel.  eht / eout rows of the question are streamed with float4 loads (several
// encoder steps in flight per wave); v, q, out, the attention row and the context live in
// registers / LDS.
// ---------------------------------------------------------------------------------------------
constexpr int MAXV = 16;
constexpr int MAXKI = 4;      // lstm_dim <= 1024

// validity automaton, token choice, probabilities and entropy of ONE (question, step): wave 0,
// lanes 0..V-1 own one token each; sc = that token's logit (-INFINITY for lanes >= V)  (:200-268)
__device__ __forceinline__ void dec_token_tail(const DecStepArgs& a, int n, int ts, size_t tn,
                                               float sc, int lane) {
  const int V = a.V, N = a.N;
  const bool on = lane < V;
  if (on && a.scores) a.scores[tn * V + lane] = sc;
  int x0 = 0, x1 = 0, x2 = 0;
  bool valid = on;
  if (a.use_gt != 1) {
    if (a.use_gt == 2) {                   // tokens known: X = [0, 0, T_dec] + sum of P[earlier tokens]
      x2 = a.Td;
      for (int tau = 0; tau < ts; ++tau) {
        const int tk = a.gt[(size_t)tau * N + n];
        x0 += a.P[tk * 3 + 0]; x1 += a.P[tk * 3 + 1]; x2 += a.P[tk * 3 + 2];
      }
    } else {
      x0 = a.state[n * 3 + 0]; x1 = a.state[n * 3 + 1]; x2 = a.state[n * 3 + 2];
    }
    if (on) {
      for (int c = 0; c < 4; ++c) {        // all_c( X . W[:, s, c] - b[s, c] >= 0 )   (:8-11)
        const int val = x0 * a.Wv[(0 * V + lane) * 4 + c] + x1 * a.Wv[(1 * V + lane) * 4 + c] +
                        x2 * a.Wv[(2 * V + lane) * 4 + c] - a.bv[lane * 4 + c];
        valid = valid && (val >= 0);
      }
    }
  }                                        // use_gt: logical_or(valid, True)          (:204-207)
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
    float cdf = 0.f, tot = 0.f;            // inclusive scan over V <= 16 lanes, in order
    for (int s = 0; s < V; ++s) {
      const float v = __shfl(ps, s, 64);
      tot += v;
      if (s == lane) cdf = tot;
    }
    const float thr = a.uni[tn] * tot;
    const unsigned long long le = __ballot(on && cdf <= thr);
    int samp = __builtin_popcountll(le);
    samp = samp < V - 1 ? samp : V - 1;
    const bool ok = (__ballot(on && valid) >> samp) & 1ull;
    tok = ok ? samp : tok;
  }
  if (a.use_gt && a.gt) tok = a.gt[tn];    // (:239-241)
  if (a.valid_bits) {
    const unsigned long long vb = __ballot(on && valid);
    if (lane == 0) a.valid_bits[tn] = (int32_t)vb;
  }
  if (a.forced) tok = a.forced[tn];
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
    a.tokens[tn] = tok;
    a.tprobs[tn] = tp;
    a.ent_t[tn] = ent;
    if (!a.use_gt) {
      a.next_idx[n] = tok;
      a.state[n * 3 + 0] = x0 + a.P[tok * 3 + 0];        // X += P[token]            (:13-15)
      a.state[n * 3 + 1] = x1 + a.P[tok * 3 + 1];
      a.state[n * 3 + 2] = x2 + a.P[tok * 3 + 2];
    } else if (a.next_idx) {
      a.next_idx[n] = tok;
    }
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void dec_attn_kernel(DecStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NW = NT / 64;
  const int L = a.L, T = a.T, N = a.N, V = a.V;
  float* outs = smem;           // [L]
  float* ctx = outs + L;        // [L]
  float* ctxp = ctx + L;        // [NSPLIT][L] partial contexts
  const int ncol = L / 4;
  const int nsplit = NT / ncol > 0 ? (NT / ncol < 8 ? NT / ncol : 8) : 1;
  float* es = ctxp + (size_t)nsplit * L;      // [T] logits -> attention
  float* red = es + ((T + 3) & ~3);           // [NW][MAXV]
  const int n = blockIdx.x, ts = blockIdx.y;  // ts: step offset inside this launch
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int len = min(max(a.seq_len[n], 0), T);   // device-side lengths are not visible to the host checks
  const size_t tn = (size_t)ts * N + n;
  const float* qrow = a.q + tn * L;
  const float* orow = a.out + tn * L;

  for (int k = tid; k < L; k += NT) outs[k] = orow[k];

  // ---- e[tau] = sum_k v_k tanh(q_k + eht[tau, n, k])                               (:184-187)
  {
    float4 v4[MAXKI], q4[MAXKI];
#pragma unroll
    for (int i = 0; i < MAXKI; ++i) {
      const int k = 4 * lane + 256 * i;
      if (k < L) {
        v4[i] = *reinterpret_cast<const float4*>(a.v + k);
        q4[i] = *reinterpret_cast<const float4*>(qrow + k);
      }
    }
    constexpr int UNR = 4;
    // rows past the question's length all equal the bias of encoder_h_transform: virtual row `len`
    // stands for every one of them (see dec_attn_multi_kernel)
    const bool virt = a.eht_bias && len < T;      // row `len` is the bias vector, not a row of eht
    const int Tv = virt ? len + 1 : T;
    for (int j0 = 0; j0 * NW + w < Tv; j0 += UNR) {
      float s[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int tau = w + NW * (j0 + u);
        s[u] = 0.f;
        if (tau < Tv) {
          const float* er = (virt && tau == len) ? a.eht_bias : a.eht + ((size_t)tau * N + n) * L;
#pragma unroll
          for (int i = 0; i < MAXKI; ++i) {
            const int k = 4 * lane + 256 * i;
            if (k < L) {
              const float4 e4 = *reinterpret_cast<const float4*>(er + k);
              s[u] += v4[i].x * fast_tanh(q4[i].x + e4.x) + v4[i].y * fast_tanh(q4[i].y + e4.y) +
                      v4[i].z * fast_tanh(q4[i].z + e4.z) + v4[i].w * fast_tanh(q4[i].w + e4.w);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int tau = w + NW * (j0 + u);
        const float r = wave_sum(s[u]);
        if (lane == 0 && tau < Tv) es[tau] = r;
      }
    }
    if (virt && len + 1 < T) {
      __syncthreads();
      for (int tau = len + 1 + tid; tau < T; tau += NT) es[tau] = es[len];
    }
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
    float* arow = a.atts + (size_t)ts * T * N;
    for (int tau = lane; tau < T; tau += 64) {
      const float att = es[tau] / s2;
      es[tau] = att;
      arow[(size_t)tau * N + n] = att;
    }
  }
  __syncthreads();

  // ---- ctx = sum_tau att[tau] * eout[tau, n, :]                                     (:193)
  {
    for (int c = tid; c < ncol * nsplit; c += NT) {
      const int col = c % ncol, sp = c / ncol;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* ob = a.eout + (size_t)n * L + 4 * col;
#pragma unroll 4
      for (int tau = sp; tau < len; tau += nsplit) {
        const float at = es[tau];
        const float4 o4 = *reinterpret_cast<const float4*>(ob + (size_t)tau * N * L);
        acc.x += at * o4.x; acc.y += at * o4.y; acc.z += at * o4.z; acc.w += at * o4.w;
      }
      *reinterpret_cast<float4*>(ctxp + (size_t)sp * L + 4 * col) = acc;
    }
    __syncthreads();
    for (int k = tid; k < L; k += NT) {
      float s = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) s += ctxp[(size_t)sp * L + k];
      ctx[k] = s;
      if (a.ctx_out) a.ctx_out[tn * L + k] = s;
    }
    __syncthreads();
  }

  // ---- token logits = [out, ctx] . W_y + b_y                                        (:196-198)
  {
    float ps[MAXV];
#pragma unroll
    for (int s = 0; s < MAXV; ++s) ps[s] = 0.f;
#pragma unroll 2
    for (int k = tid; k < 2 * L; k += NT) {
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
    float sc = -INFINITY;
    if (lane < V) {
      sc = a.by[lane];
      for (int ww = 0; ww < NW; ++ww) sc += red[ww * MAXV + lane];
    }
    dec_token_tail(a, n, ts, tn, sc, lane);
  }
}

// Sequential decoding (greedy / sampled, one launch per step): one workgroup of 16 waves per question.
// ONE pass over the question's encoder rows: wave w takes rows w, w + 16, ... below the question's length
// and loads a row of eht AND the same row of the encoder outputs together (two independent streams, one
// memory round trip), keeping a running
// (max, sum, weighted row sum).  The reference's attention -- soft-max over ALL T rows, mask, renormalise
// (:190-191) -- is exp(e - m) / sum over the rows inside the length for ANY m, so the rows past the length
// are never touched, the attention needs no second pass over the rows, and the 16 partial states meet in
// LDS.  (dec_attn_kernel<1024>, the form this replaces, made three dependent passes: scores of all rows,
// soft-max by one wave, context; 60 us per 1024 questions against 44.)
template <int KI>
__global__ __launch_bounds__(1024, 2) void dec_attn_seq_kernel(DecStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 1024, NW = NT / 64;
  const int L = a.L, T = a.T, N = a.N, V = a.V;
  const int Tp = (T + 3) & ~3;
  float* outs = smem;                 // [L]
  float* ctx = outs + L;              // [L]
  float* ctxp = ctx + L;              // [NW][L] partial contexts
  float* es = ctxp + (size_t)NW * L;  // [Tp] raw scores of the rows inside the length
  float* ms = es + Tp;                // [NW][2] running (max, sum) of each wave, then [2 NW ..] merge weights
  float* red = ms + 4 * NW;           // [NW][MAXV]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // eos_retire: workgroup j serves state row j = original row live_perm[j]; rows at or past *live_n have
  // emitted <eos> and get <eos> again -- what the automaton would force (dec_compact_kernel)
  int n = blockIdx.x;
  size_t qr = n;                                    // q lives in STATE-row order, everything else by original row
  if (a.live_n) {
    n = a.live_perm[blockIdx.x];
    if ((int)blockIdx.x >= *a.live_n) {
      if (tid == 0) {
        a.tokens[n] = a.eos_token; a.tprobs[n] = 1.0f; a.ent_t[n] = 0.f;
        if (a.next_idx) a.next_idx[n] = a.eos_token;
      }
      return;
    }
  }
  const int len = min(max(a.seq_len[n], 0), T);
  const size_t tn = n;                              // (one step per launch: step offset 0)
  const float* qrow = a.q + qr * L;
  const float* orow = a.out + tn * L;
  for (int k = tid; k < L; k += NT) outs[k] = orow[k];

  float4 v4[KI], q4[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) {
    const int k = 4 * lane + 256 * i;
    v4[i] = *reinterpret_cast<const float4*>(a.v + k);
    q4[i] = *reinterpret_cast<const float4*>(qrow + k);
  }
  const size_t rstride = (size_t)N * L;
  const float* eb = a.eht + (size_t)n * L + 4 * lane;
  const float* ob = a.eout + (size_t)n * L + 4 * lane;
  float4 e4[KI], o4[KI], acc[KI];
#pragma unroll
  for (int i = 0; i < KI; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  float m_run = -INFINITY, s_run = 0.f;
  // (a wave has ceil(len / 16) <= 3 rows, 1.1 on average: no software pipelining -- the registers it
  // would take cost the second workgroup per CU)
  for (int tau = w; tau < len; tau += NW) {
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      e4[i] = *reinterpret_cast<const float4*>(eb + (size_t)tau * rstride + 256 * i);
      o4[i] = *reinterpret_cast<const float4*>(ob + (size_t)tau * rstride + 256 * i);
    }
    float sacc = 0.f;                               // e[tau] = sum_k v_k tanh(q_k + eht[tau, n, k])   (:184-187)
#pragma unroll
    for (int i = 0; i < KI; ++i)
      sacc += v4[i].x * fast_tanh(q4[i].x + e4[i].x) + v4[i].y * fast_tanh(q4[i].y + e4[i].y) +
              v4[i].z * fast_tanh(q4[i].z + e4[i].z) + v4[i].w * fast_tanh(q4[i].w + e4[i].w);
    const float sc = wave_sum(sacc);
    if (lane == 0) es[tau] = sc;
    const float mn = fmaxf(m_run, sc);
    const float scale = expf(m_run - mn), p = expf(sc - mn);     // (first row: exp(-inf) = 0)
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      acc[i].x = acc[i].x * scale + p * o4[i].x; acc[i].y = acc[i].y * scale + p * o4[i].y;
      acc[i].z = acc[i].z * scale + p * o4[i].z; acc[i].w = acc[i].w * scale + p * o4[i].w;
    }
    s_run = s_run * scale + p;
    m_run = mn;
  }
#pragma unroll
  for (int i = 0; i < KI; ++i)
    *reinterpret_cast<float4*>(ctxp + (size_t)w * L + 4 * lane + 256 * i) = acc[i];
  if (lane == 0) { ms[2 * w] = m_run; ms[2 * w + 1] = s_run; }
  // this thread's row of W_y (2 L = 1024 rows = one per thread) is requested now: it arrives under the
  // merge below instead of in front of the token logits
  float wy[MAXV];
  {
    const float* wr = a.Wy + (size_t)min(tid, 2 * L - 1) * V;
#pragma unroll
    for (int s = 0; s < MAXV; ++s) wy[s] = s < V ? wr[s] : 0.f;
  }
  __syncthreads();
  // ---- the partial states meet: weight of wave w = exp(m_w - M) / S                       (:190-193)
  if (w == 0) {
    const float mw = lane < NW ? ms[2 * lane] : -INFINITY;
    const float sw = lane < NW ? ms[2 * lane + 1] : 0.f;
    const float M = wave_max(mw);
    const float ew = lane < NW && sw > 0.f ? expf(mw - M) : 0.f;
    const float S = wave_sum(sw * ew);
    if (lane < NW) ms[2 * NW + lane] = ew / S;
    if (lane == 0) { ms[3 * NW] = M; ms[3 * NW + 1] = 1.0f / S; }
  }
  __syncthreads();
  {
    const float M = ms[3 * NW], rS = ms[3 * NW + 1];
    float* arow = a.atts;
    for (int t = tid; t < T; t += NT) arow[(size_t)t * N + n] = t < len ? expf(es[t] - M) * rS : 0.f;
    for (int k = tid; k < L; k += NT) {
      float c = 0.f;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) c += ms[2 * NW + ww] * ctxp[(size_t)ww * L + k];
      ctx[k] = c;
      if (a.ctx_out) a.ctx_out[tn * L + k] = c;
    }
  }
  __syncthreads();
  // ---- token logits = [out, ctx] . W_y + b_y                                              (:196-198)
  {
    const float x = tid < L ? outs[tid] : (tid < 2 * L ? ctx[tid - L] : 0.f);
#pragma unroll
    for (int s = 0; s < MAXV; ++s) {
      const float r = wave_sum(x * wy[s]);
      if (lane == 0) red[w * MAXV + s] = r;
    }
  }
  __syncthreads();
  if (w == 0) {
    float sc = -INFINITY;
    if (lane < V) {
      sc = a.by[lane];
      for (int ww = 0; ww < NW; ++ww) sc += red[ww * MAXV + lane];
    }
    dec_token_tail(a, n, 0, tn, sc, lane);
  }
}

// All decoder steps are known up front (teacher forcing / given tokens): one workgroup handles TS
// steps of one question, so each encoder row (eht and encoder_outputs, 2 KB each) is loaded ONCE for
// TS query vectors instead of once per step -- the single-step grid re-read 184 KB per (question,
// step) from L2 (236 MB per launch at N = 64) and ran at the L2's pace.
template <int KI, int TS>
__global__ __launch_bounds__(256) void dec_attn_multi_kernel(DecStepArgs a, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 256, NW = 4;
  const int L = a.L, T = a.T, N = a.N, V = a.V;
  const int Tp = (T + 3) & ~3;
  const int ncol = L / 4;
  const int nsplit = NT / ncol > 0 ? NT / ncol : 1;
  float* outs = smem;                               // [TS][L]
  float* ctx = outs + (size_t)TS * L;               // [TS][L]
  float* ctxp = ctx + (size_t)TS * L;               // [nsplit][TS][L]
  float* es = ctxp + (size_t)nsplit * TS * L;       // [TS][Tp]
  float* red = es + (size_t)TS * Tp;                // [NW][TS][MAXV]
  const int n = blockIdx.x, ts0 = blockIdx.y * TS;
  int nts = min(TS, nsteps - ts0);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int len = min(max(a.seq_len[n], 0), T);
  if (a.dec_len) {
    // eos_retire (DecStepArgs::dec_len): the steps from the layout's first <eos> on only get their (given)
    // token; a step group that lies behind it entirely leaves at once
    const int live = min(nsteps, max(a.dec_len[n], 0));
    for (int j = max(live - ts0, 0) + tid; j < nts; j += NT)
      a.tokens[(size_t)(ts0 + j) * N + n] = a.gt[(size_t)(ts0 + j) * N + n];
    nts = min(nts, live - ts0);
    if (nts <= 0) return;
  }

  for (int i = tid; i < TS * L; i += NT) {
    const int j = i / L, k = i - j * L;
    outs[i] = j < nts ? a.out[((size_t)(ts0 + j) * N + n) * L + k] : 0.f;
  }
  // ---- e[j][tau] = sum_k v_k tanh(q_j,k + eht[tau, n, k])                           (:184-187)
  {
    float4 v4[KI], q4[TS][KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = min(4 * lane + 256 * i, L - 4);
      const bool ok = 4 * lane + 256 * i < L;
      v4[i] = ok ? *reinterpret_cast<const float4*>(a.v + k) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < TS; ++j) {
        const int jj = min(j, nts - 1);
        q4[j][i] = *reinterpret_cast<const float4*>(a.q + ((size_t)(ts0 + jj) * N + n) * L + k);
      }
    }
    constexpr int UNR = TS <= 2 ? 6 : 3;     // encoder rows of a wave in flight
    // rows past the question's length all equal the bias of encoder_h_transform: virtual row `len`
    // stands for every one of them (evaluated once, copied to the others below)
    const bool virt = a.eht_bias && len < T;      // row `len` is the bias vector, not a row of eht
    const int Tv = virt ? len + 1 : T;
    for (int j0 = 0; j0 * NW + w < Tv; j0 += UNR) {
      float4 e4[UNR][KI];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int tau = min(w + NW * (j0 + u), Tv - 1);
        const float* er = (virt && tau == len) ? a.eht_bias : a.eht + ((size_t)tau * N + n) * L;
#pragma unroll
        for (int i = 0; i < KI; ++i)
          e4[u][i] = *reinterpret_cast<const float4*>(er + min(4 * lane + 256 * i, L - 4));
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int tau = w + NW * (j0 + u);
#pragma unroll
        for (int j = 0; j < TS; ++j) {
          float sacc = 0.f;
#pragma unroll
          for (int i = 0; i < KI; ++i) {
            sacc += v4[i].x * fast_tanh(q4[j][i].x + e4[u][i].x) +
                    v4[i].y * fast_tanh(q4[j][i].y + e4[u][i].y) +
                    v4[i].z * fast_tanh(q4[j][i].z + e4[u][i].z) +
                    v4[i].w * fast_tanh(q4[j][i].w + e4[u][i].w);
          }
          const float r = wave_sum(sacc);
          if (lane == 0 && tau < Tv) es[j * Tp + tau] = r;
        }
      }
    }
    if (virt && len + 1 < T) {
      __syncthreads();
      for (int i = tid; i < TS * (T - len - 1); i += NT) {
        const int j = i / (T - len - 1), tau = len + 1 + i % (T - len - 1);
        es[j * Tp + tau] = es[j * Tp + len];
      }
    }
  }
  __syncthreads();
  // ---- softmax over ALL T rows, mask finished rows, renormalise                    (:190-191)
  for (int j = w; j < nts; j += NW) {
    float* ej = es + j * Tp;
    float m = -INFINITY;
    for (int tau = lane; tau < T; tau += 64) m = fmaxf(m, ej[tau]);
    m = wave_max(m);
    float sm = 0.f;
    for (int tau = lane; tau < T; tau += 64) sm += expf(ej[tau] - m);
    sm = wave_sum(sm);
    float s2 = 0.f;
    for (int tau = lane; tau < T; tau += 64) {
      float p = expf(ej[tau] - m) / sm;
      p = tau < len ? p : 0.f;
      ej[tau] = p;
      s2 += p;
    }
    s2 = wave_sum(s2);
    float* arow = a.atts + (size_t)(ts0 + j) * T * N;
    for (int tau = lane; tau < T; tau += 64) {
      const float att = ej[tau] / s2;
      ej[tau] = att;
      arow[(size_t)tau * N + n] = att;
    }
  }
  __syncthreads();
  // ---- ctx[j] = sum_tau att[j][tau] * eout[tau, n, :]                                (:193)
  {
    for (int c = tid; c < ncol * nsplit; c += NT) {
      const int col = c % ncol, sp = c / ncol;
      float4 acc[TS];
#pragma unroll
      for (int j = 0; j < TS; ++j) acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* ob = a.eout + (size_t)n * L + 4 * col;
#pragma unroll 4
      for (int tau = sp; tau < len; tau += nsplit) {
        const float4 o4 = *reinterpret_cast<const float4*>(ob + (size_t)tau * N * L);
#pragma unroll
        for (int j = 0; j < TS; ++j) {
          const float at = es[j * Tp + tau];
          acc[j].x += at * o4.x; acc[j].y += at * o4.y; acc[j].z += at * o4.z; acc[j].w += at * o4.w;
        }
      }
#pragma unroll
      for (int j = 0; j < TS; ++j)
        *reinterpret_cast<float4*>(ctxp + ((size_t)sp * TS + j) * L + 4 * col) = acc[j];
    }
    __syncthreads();
    for (int i = tid; i < TS * L; i += NT) {
      const int j = i / L, k = i - j * L;
      float sacc = 0.f;
      for (int sp = 0; sp < nsplit; ++sp) sacc += ctxp[((size_t)sp * TS + j) * L + k];
      ctx[i] = sacc;
      if (a.ctx_out && j < nts) a.ctx_out[((size_t)(ts0 + j) * N + n) * L + k] = sacc;
    }
    __syncthreads();
  }
  // ---- token logits = [out, ctx] . W_y + b_y                                        (:196-198)
  {
    float ps[TS][MAXV];
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
      for (int sI = 0; sI < MAXV; ++sI) ps[j][sI] = 0.f;
    for (int k = tid; k < 2 * L; k += NT) {
      const float* wr = a.Wy + (size_t)k * V;
      float wv[MAXV];
#pragma unroll
      for (int sI = 0; sI < MAXV; ++sI) wv[sI] = sI < V ? wr[sI] : 0.f;
#pragma unroll
      for (int j = 0; j < TS; ++j) {
        const float x = k < L ? outs[j * L + k] : ctx[j * L + k - L];
#pragma unroll
        for (int sI = 0; sI < MAXV; ++sI) ps[j][sI] += x * wv[sI];
      }
    }
#pragma unroll
    for (int j = 0; j < TS; ++j)
#pragma unroll
      for (int sI = 0; sI < MAXV; ++sI) {
        const float r = wave_sum(ps[j][sI]);
        if (lane == 0) red[(w * TS + j) * MAXV + sI] = r;
      }
  }
  __syncthreads();
  if (w == 0) {
    for (int j = 0; j < nts; ++j) {
      float sc = -INFINITY;
      if (lane < V) {
        sc = a.by[lane];
        for (int ww = 0; ww < NW; ++ww) sc += red[(ww * TS + j) * MAXV + lane];
      }
      dec_token_tail(a, n, ts0 + j, (size_t)(ts0 + j) * N + n, sc, lane);
    }
  }
}

// dec_attn_question_kernel: the teacher-forced attention of ONE question, ALL decoder steps, in one
// 16-wave workgroup (nmn3_netgen_att.py:184-268 for every step at once).
// dec_attn_multi_kernel gives a question to ceil(T_dec / 4) workgroups, each of which streams the
// question's encoder rows again (PMC: 302 MB per 512-question launch for 94 MB of rows, 170 us).  Here
//   * the question's encoder_h_transform rows (len x 2 KB <= 92 KB) are read from HBM/L2 ONCE into LDS;
//   * the tanh scores e[j][tau] = sum_k v_k tanh(q_j,k + eht[tau,k]) -- the VALU-bound part, two
//     transcendentals per (step, row, k) -- are spread over the 16 waves as 4 step groups x 4 row
//     quarters: a wave keeps the query vectors of its 5 steps in registers and walks its rows in LDS;
//   * the context vectors ctx[j] = sum_tau att[j][tau] eout[tau] are ONE small matrix product per
//     question ([32 steps] x [len] x [L]) on the matrix cores (v_mfma_f32_32x32x2_f32): every wave owns
//     32 (64) columns, its B fragments are the encoder_outputs rows straight from global memory, each
//     element loaded exactly once, the attention weights come from LDS;
//   * token logits, validity, probabilities and entropy as in the other two kernels (dec_token_tail).
template <int KI>
__global__ __launch_bounds__(1024) void dec_attn_question_kernel(DecStepArgs a, int nsteps) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NT = 1024, NW = 16, SPG = 5;
  const int L = a.L, T = a.T, N = a.N, V = a.V;
  const int Tp = (T + 3) & ~3;
  float* eS = smem;                                          // [max(T, nsteps)][L]: eht rows, later ctx
  float* es = eS + (size_t)(T > nsteps ? T : nsteps) * L;   // [32][Tp] scores -> attention weights
  // work per question grows with its length (5..45 rows): the longest questions take the lowest
  // workgroup ids, i.e. are dispatched first, and the short ones fill in behind them
  const int n = a.order ? a.order[blockIdx.x] : (int)blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int len = min(max(a.seq_len[n], 0), T);
  // eos_retire (DecStepArgs::dec_len): only the steps that emit a module token are evaluated -- the steps
  // from the first <eos> on feed nothing the caller asked for (decoder_impl); their tokens are the given
  // ones.  `nsteps` keeps describing the launch (LDS layout, output strides), `ns` the work.
  const int ns = a.dec_len ? min(nsteps, max(a.dec_len[n], 0)) : nsteps;
  if (ns < nsteps) {
    for (int j = ns + tid; j < nsteps; j += NT) a.tokens[(size_t)j * N + n] = a.gt[(size_t)j * N + n];
    if (ns == 0) return;
  }
  // rows past the question's length all equal the bias of encoder_h_transform: virtual row `len`
  // stands for every one of them (evaluated once, copied to the others below)
  const bool virt = a.eht_bias && len < T;      // row `len` is the bias vector, not a row of eht
  const int Tv = virt ? len + 1 : T;

  // ---- 0. the question's rows -> LDS (16 KB per sweep of the workgroup), zero the score block -----
  {
    const int ncol = L / 4, total = Tv * ncol;
    constexpr int SW = 3 * KI;                               // sweeps cover T * L <= 12 K * KI floats
    float4 r4[SW];
#pragma unroll
    for (int u = 0; u < SW; ++u) {
      const int i = min(tid + u * NT, total - 1);
      const int row = i / ncol, c4 = i - row * ncol;
      const float* er = (virt && row == len) ? a.eht_bias : a.eht + ((size_t)row * N + n) * L;
      r4[u] = *reinterpret_cast<const float4*>(er + 4 * c4);
    }
#pragma unroll
    for (int u = 0; u < SW; ++u) {
      const int i = tid + u * NT;
      if (i < total) *reinterpret_cast<float4*>(eS + 4 * (size_t)i) = r4[u];
    }
    for (int i = tid; i < 32 * Tp; i += NT) es[i] = 0.f;
  }
  __syncthreads();
  // ---- 1. e[j][tau] = sum_k v_k tanh(q_j,k + eht[tau, n, k])                          (:184-187)
  {
    // 16 waves = (step groups of SPG steps) x (row parts): four groups x four row quarters for a full
    // launch; a question with few live steps (eos_retire) gives its waves to the rows instead
    const int nrq = ns > 2 * SPG ? 4 : (ns > SPG ? 8 : 16);  // row parts
    const int sg = w / nrq, tq = w - sg * nrq;
    const int j0 = sg * SPG;
    float4 v4[KI], q4[SPG][KI];
#pragma unroll
    for (int i = 0; i < KI; ++i) {
      const int k = 4 * lane + 256 * i;
      v4[i] = *reinterpret_cast<const float4*>(a.v + k);
#pragma unroll
      for (int j = 0; j < SPG; ++j) {
        const int jj = min(j0 + j, ns - 1);
        q4[j][i] = *reinterpret_cast<const float4*>(a.q + ((size_t)jj * N + n) * L + k);
      }
    }
    if (j0 < ns) {
      for (int tau = tq; tau < Tv; tau += 2 * nrq) {         // two rows of this part per trip
        const int tau1 = tau + nrq;
        float4 e0[KI], e1[KI];
#pragma unroll
        for (int i = 0; i < KI; ++i) {
          e0[i] = *reinterpret_cast<const float4*>(eS + (size_t)tau * L + 4 * lane + 256 * i);
          e1[i] = *reinterpret_cast<const float4*>(eS + (size_t)min(tau1, Tv - 1) * L + 4 * lane + 256 * i);
        }
#pragma unroll
        for (int j = 0; j < SPG; ++j) {
          float s0 = 0.f, s1 = 0.f;
#pragma unroll
          for (int i = 0; i < KI; ++i) {
            s0 += v4[i].x * fast_tanh(q4[j][i].x + e0[i].x) + v4[i].y * fast_tanh(q4[j][i].y + e0[i].y) +
                  v4[i].z * fast_tanh(q4[j][i].z + e0[i].z) + v4[i].w * fast_tanh(q4[j][i].w + e0[i].w);
            s1 += v4[i].x * fast_tanh(q4[j][i].x + e1[i].x) + v4[i].y * fast_tanh(q4[j][i].y + e1[i].y) +
                  v4[i].z * fast_tanh(q4[j][i].z + e1[i].z) + v4[i].w * fast_tanh(q4[j][i].w + e1[i].w);
          }
          const float r0 = wave_sum(s0), r1 = wave_sum(s1);
          if (lane == 0 && j0 + j < ns) {
            es[(j0 + j) * Tp + tau] = r0;
            if (tau1 < Tv) es[(j0 + j) * Tp + tau1] = r1;
          }
        }
      }
    }
  }
  __syncthreads();
  if (virt && len + 1 < T) {
    const int rest = T - len - 1;
    for (int i = tid; i < ns * rest; i += NT) {
      const int j = i / rest, tau = len + 1 + i - j * rest;
      es[j * Tp + tau] = es[j * Tp + len];
    }
    __syncthreads();
  }
  // ---- 2. softmax over ALL T rows, mask finished rows, renormalise                   (:190-191)
  for (int j = w; j < ns; j += NW) {
    float* ej = es + j * Tp;
    float m = -INFINITY;
    for (int tau = lane; tau < T; tau += 64) m = fmaxf(m, ej[tau]);
    m = wave_max(m);
    float sm = 0.f;
    for (int tau = lane; tau < T; tau += 64) sm += expf(ej[tau] - m);
    sm = wave_sum(sm);
    float s2 = 0.f;
    for (int tau = lane; tau < T; tau += 64) {
      float p = expf(ej[tau] - m) / sm;
      p = tau < len ? p : 0.f;
      ej[tau] = p;
      s2 += p;
    }
    s2 = wave_sum(s2);
    float* arow = a.atts + (size_t)j * T * N;
    for (int tau = lane; tau < T; tau += 64) {
      const float att = ej[tau] / s2;
      ej[tau] = att;
      arow[(size_t)tau * N + n] = att;
    }
  }
  __syncthreads();
  // ---- 3. ctx[j] = sum_tau att[j][tau] * eout[tau, n, :]  on the matrix cores          (:193)
  // D[step][col] += A[step][k] B[k][col], k = encoder row: lane l holds A[l % 32][l / 32] (LDS) and
  // B[l / 32][l % 32] (global); rows >= nsteps of A are zero, weights of rows >= len are zero
  {
    float* ctxS = eS;                                        // every wave is done with the eht rows
    const int li = lane & 31, kh = lane >> 5;
    const int nm = (len + 1) >> 1;                           // MFMAs (2 encoder rows each)
    for (int cb = 32 * w; cb < L; cb += 32 * NW) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* bcol = a.eout + (size_t)n * L + cb + li;
      for (int m0 = 0; m0 < nm; m0 += 8) {
        float bv[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int tau = min(2 * (m0 + u) + kh, T - 1);
          bv[u] = bcol[(size_t)tau * N * L];
          av[u] = es[li * Tp + min(2 * (m0 + u) + kh, Tp - 1)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (m0 + u < nm) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
      }
      // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (j < ns) {
          ctxS[(size_t)j * L + cb + li] = acc[r];
          if (a.ctx_out) a.ctx_out[((size_t)j * N + n) * L + cb + li] = acc[r];
        }
      }
    }
  }
  __syncthreads();
  // ---- 4. token logits = [out, ctx] . W_y + b_y for all steps: one [32 x 2L] x [2L x 32] product on
  // the matrix cores, K split over the 16 waves (a per-thread partial per (step, token) would cost 64
  // wave reductions per wave and 4 steps -- as much VALU time as the tanh scores)       (:196-198)
  {
    const float* ctxS = eS;
    float* part = eS + (size_t)nsteps * L;                   // [NW][nsteps][MAXV], behind the ctx rows
    const int li = lane & 31, kh = lane >> 5;
    const int kper = 2 * L / NW;                             // k range of this wave (64 at L = 512)
    const int k0 = w * kper;
    const int jr = min(li, ns - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int m0 = 0; m0 < kper / 2; m0 += 8) {
      float av[8], bv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 2 * (m0 + u) + kh;
        av[u] = k < L ? a.out[((size_t)jr * N + n) * L + k] : ctxS[(size_t)jr * L + k - L];
        bv[u] = li < V ? a.Wy[(size_t)k * V + li] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    if (li < MAXV) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (r & 3) + 8 * (r >> 2) + 4 * kh;       // step (row of D), li = token (column)
        if (j < ns) part[((size_t)w * nsteps + j) * MAXV + li] = acc[r];
      }
    }
    __syncthreads();
    for (int j = w; j < ns; j += NW) {
      float sc = -INFINITY;
      if (lane < V) {
        sc = a.by[lane];
        for (int ww = 0; ww < NW; ++ww) sc += part[((size_t)ww * nsteps + j) * MAXV + lane];
      }
      dec_token_tail(a, n, j, (size_t)j * N + n, sc, lane);
    }
  }
}

// perm[rank] = n with rows ranked by decreasing length (ties by index); n_active[t] = #{len > t}.
// The same launch clears the recurrent state block (a separate memset node costs ~5 us on the
// stream).
constexpr int PREP_MAXN = 1024;
__global__ __launch_bounds__(256) void enc_prepare_kernel(const int32_t* __restrict__ seq_len,
                                                          int N, int T, int32_t* __restrict__ perm,
                                                          int32_t* __restrict__ n_active,
                                                          float4* __restrict__ zero, size_t zero4,
                                                          int32_t* __restrict__ zero_int) {
  const int tid = threadIdx.x;
  if (zero_int && blockIdx.x == 0 && tid == 0) *zero_int = 0;     // counter of enc_rows_kernel
  for (size_t i = (size_t)blockIdx.x * 256 + tid; i < zero4; i += (size_t)gridDim.x * 256)
    zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  // ranks and active-row counts are spread over the workgroups too (one workgroup needed 81 us for
  // the O(N^2) ranking of a 512-question super-bucket)
  __shared__ __attribute__((aligned(16))) int lens[PREP_MAXN];
  __shared__ float scratch[16];
  for (int i = tid; i < N && i < PREP_MAXN; i += 256) lens[i] = seq_len[i];
  __syncthreads();
  const bool in_lds = N <= PREP_MAXN;
  for (int i = blockIdx.x * 256 + tid; i < N; i += gridDim.x * 256) {
    const int li = in_lds ? lens[i] : seq_len[i];
    int rank = 0;
    if (in_lds) {
      // four lengths per LDS read, four reads in flight: a scalar loop pays the LDS latency
      // (~100 cycles) per question, 25 us at 512 questions
      const int4* l4 = reinterpret_cast<const int4*>(lens);
      const int n4 = N >> 2;
      int j4 = 0;
      for (; j4 + 4 <= n4; j4 += 4) {
        int4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = l4[j4 + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = 4 * (j4 + u);
          rank += (v[u].x > li) || (v[u].x == li && j < i);
          rank += (v[u].y > li) || (v[u].y == li && j + 1 < i);
          rank += (v[u].z > li) || (v[u].z == li && j + 2 < i);
          rank += (v[u].w > li) || (v[u].w == li && j + 3 < i);
        }
      }
      for (int j = 4 * j4; j < N; ++j) {
        const int lj = lens[j];
        rank += (lj > li) || (lj == li && j < i);
      }
    } else {
      for (int j = 0; j < N; ++j) {
        const int lj = seq_len[j];
        rank += (lj > li) || (lj == li && j < i);
      }
    }
    perm[rank] = i;
  }
  for (int t = blockIdx.x; t < T; t += gridDim.x) {
    float c = 0.f;
    for (int j = tid; j < N; j += 256) c += (in_lds ? lens[j] : seq_len[j]) > t ? 1.f : 0.f;
    const float tot = block_reduce<0>(c, scratch);
    if (tid == 0) n_active[t] = (int)(tot + 0.5f);
  }
}

// The same ranking as a stable counting sort (lengths are small integers): N <= 1024 questions,
// T <= 63.  enc_prepare_kernel compares every question with every other one (O(N^2): 27 us at 1024
// questions, 0.7 % of a pass); here ONE 1024-thread workgroup (block 0) ranks them in a few LDS passes
// while the other workgroups clear the state block:
//   1. thread i clamps its length l_i to [0, T]; per wave, the lanes holding equal lengths are found
//      with six ballots (one per bit of l), which gives the lane's rank among the equal lengths of its
//      wave and the wave's count per length;
//   2. cnt[w][l] -> exclusive prefix over the waves per length, totals per length, and
//      start[l] = #{lengths > l} (descending order);
//   3. perm[start[l_i] + (equal lengths in earlier waves) + (equal lengths in earlier lanes)] = i --
//      ties by index, exactly enc_prepare_kernel's order; n_active[t] = #{l > t} = start[t].
__global__ __launch_bounds__(1024) void enc_prepare_sort_kernel(const int32_t* __restrict__ seq_len,
                                                                int N, int T, int32_t* __restrict__ perm,
                                                                int32_t* __restrict__ n_active,
                                                                float4* __restrict__ zero, size_t zero4,
                                                                int32_t* __restrict__ zero_int) {
  const int tid = threadIdx.x;
  if (blockIdx.x != 0 || gridDim.x == 1) {
    const size_t nb = gridDim.x == 1 ? 1 : gridDim.x - 1, b = gridDim.x == 1 ? 0 : blockIdx.x - 1;
    for (size_t i = b * 1024 + tid; i < zero4; i += nb * 1024) zero[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x != 0) return;
  }
  if (zero_int && tid == 0) *zero_int = 0;                 // counter of enc_rows_kernel
  __shared__ int cnt[16][64];                              // [wave][length]: equal lengths per wave
  __shared__ int start[64];                                // #{lengths > l}
  const int lane = tid & 63, w = tid >> 6;
  const bool have = tid < N;
  const int l = have ? min(max(seq_len[tid], 0), T) : 0;
  cnt[w][lane] = 0;
  __syncthreads();
  unsigned long long eq = __ballot(have);
#pragma unroll
  for (int b = 0; b < 6; ++b) {
    const unsigned long long m = __ballot(have && ((l >> b) & 1));
    eq &= ((l >> b) & 1) ? m : ~m;
  }
  const int before = __builtin_popcountll(eq & ((1ull << lane) - 1ull));
  if (have && before == 0) cnt[w][l] = __builtin_popcountll(eq);   // first lane of its class
  __syncthreads();
  // exclusive prefix over the waves, per length: thread (w, lane = length)
  int pre = 0;
  for (int q = 0; q < w; ++q) pre += cnt[q][lane];
  int tot = pre;
  for (int q = w; q < 16; ++q) tot += cnt[q][lane];
  __syncthreads();
  cnt[w][lane] = pre;
  if (w == 0) start[lane] = tot;                            // totals per length, for now
  __syncthreads();
  int s2 = 0;                                               // start[l] = sum of totals of lengths > l
  if (tid < 64)
    for (int k = tid + 1; k < 64; ++k) s2 += start[k];
  __syncthreads();
  if (tid < 64) start[tid] = s2;
  __syncthreads();
  if (have) perm[start[l] + cnt[w][l] + before] = tid;
  if (tid < T) n_active[tid] = start[tid];
}

// rows[0 .. *count) = the (t, n) rows inside their question's length; one atomic per workgroup of 1024
// candidates (one per wave took 10 us at 46 K rows: 720 atomics on one address)
__global__ __launch_bounds__(1024) void enc_rows_kernel(const int32_t* __restrict__ seq_len, int T, int N,
                                                        int32_t* __restrict__ rows,
                                                        int32_t* __restrict__ count) {
  __shared__ int wcnt[16];
  __shared__ int base_s;
  const int i = blockIdx.x * 1024 + threadIdx.x;
  bool act = false;
  if (i < T * N) {
    const int t = i / N, n = i - t * N;
    act = t < seq_len[n];
  }
  const unsigned long long m = __ballot(act);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) wcnt[w] = __builtin_popcountll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int k = 0; k < 16; ++k) { const int c = wcnt[k]; wcnt[k] = tot; tot += c; }
    base_s = tot ? atomicAdd(count, tot) : 0;
  }
  __syncthreads();
  if (act) rows[base_s + wcnt[w] + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = i;
}

// eos_retire: dec_len[n] = decoder steps row n's layout needs = tokens in front of the first <eos>
// (token_op < 0; a token outside the vocabulary is not an <eos>: an invalid layout simply keeps every
// step).  nmn3_assembler.py:153-170 reads a layout up to its first <eos> and nothing behind it.
__global__ void dec_len_kernel(const int32_t* __restrict__ tokens, const int32_t* __restrict__ token_op,
                               int V, int T_dec, int N, int32_t* __restrict__ dec_len) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  int len = T_dec;
  for (int t = T_dec - 1; t >= 0; --t) {
    const int tok = tokens[(size_t)t * N + n];
    if (tok >= 0 && tok < V && token_op[tok] < 0) len = t;
  }
  dec_len[n] = len;
}

// eos_retire: the decoder's state rows in the order of `perm` (rows by decreasing layout length, so the
// rows still alive at a step are a prefix): dst_i[k4][r] = src_i[k4][perm[r]] for the four initial state
// arrays (k-interleaved [L/4][R][4]) and, in the split-operand mode, the bf16 planes [3][L/8][R][8] of
// the two hidden states.
struct GatherArgs {
  const float* src[4]; float* dst[4];
  const uint16_t* srcb[2]; uint16_t* dstb[2];   // planes of src[0] / src[2] or nullptr
  const int32_t* perm;
  int N, L, R;
};
__global__ __launch_bounds__(256) void gather_state_kernel(GatherArgs g) {
  const int r = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;                       // 4 k-slices per workgroup
  if (r >= g.N) return;
  const int pr = g.perm[r];
  const int k4n = g.L / 4;
  for (int k4 = blockIdx.y * 4 + sub; k4 < k4n; k4 += gridDim.y * 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(g.src[i] + ((size_t)k4 * g.R + pr) * 4);
      *reinterpret_cast<float4*>(g.dst[i] + ((size_t)k4 * g.R + r) * 4) = v;
    }
  }
  if (g.srcb[0]) {
    const size_t plane = (size_t)(g.L / 8) * g.R * 8;
    const int k8n = g.L / 8;
    for (int k8 = blockIdx.y * 4 + sub; k8 < k8n; k8 += gridDim.y * 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
          const uint4 v = *reinterpret_cast<const uint4*>(g.srcb[i] + p * plane + ((size_t)k8 * g.R + pr) * 8);
          *reinterpret_cast<uint4*>(g.dstb[i] + p * plane + ((size_t)k8 * g.R + r) * 8) = v;
        }
    }
  }
}

// eos_retire for sequential (greedy / sampled) decoding.  A row is finished once it has emitted its answer
// operator (the validity automaton then allows nothing but <eos>, nmn3_netgen_att.py:8-15 with
// nmn3_assembler.py:94-117) or <eos> itself, but which rows those are is only known step by step.  After the attention launch of step t this kernel
// re-partitions the decoder's state rows: live rows (token of step t is neither) keep their relative order
// in front, their four state arrays (h, c of both layers, k-interleaved [L/4][R][4]) are gathered into the
// alternate buffers, the finished rows follow in `perm_new` (so that the attention launch can still give each of
// them its <eos> token), and the live count lands in *n_new -- the recurrent launches of step t + 1 then run
// over a dense prefix exactly as the encoder does behind enc_prepare (LstmJob::perm / n_active).
// Every workgroup recomputes the partition of the <= 1024 rows (256 threads x 4 positions, one block scan)
// and copies 64 new positions x a k-slice, rows across lanes as gather_state_kernel does.
struct CompactArgs {
  const int32_t* tokens;      // [N] tokens of step t, by ORIGINAL row
  const int32_t* token_op;    // [V] operator of a token, < 0 for <eos>
  const int32_t* perm_old;    // [N] state row -> original row, or nullptr (identity)
  const int32_t* n_old;       // live rows before this step's tokens, or nullptr (N)
  int32_t* perm_new;          // [N]
  int32_t* n_new;
  const float* src[4]; float* dst[4];
  const uint16_t* srcb[2]; uint16_t* dstb[2];   // bf16x3 mode: planes [3][L/8][R][8] of src[0] / src[2], or nullptr
  int N, L, R, V;
};
__global__ __launch_bounds__(256) void dec_compact_kernel(CompactArgs g) {
  __shared__ int src_of[1024];              // new position -> old position (live rows)
  __shared__ int dead_of[1024];             // rank among the finished rows of [0, n_old) -> old position
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int N = g.N;
  const int n_old = g.n_old ? min(max(*g.n_old, 0), N) : N;
  int live[4], cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * tid + i;
    live[i] = 0;
    if (j < n_old) {
      const int n = g.perm_old ? g.perm_old[j] : j;
      const int tok = g.tokens[n];
      // live = the layout is not complete: no <eos> and no answer operator yet.  Behind an answer operator
      // the automaton allows only <eos> (verified over 47 k reachable states, tests/test_layout_nesting.py),
      // so the row's remaining tokens are known without running its steps
      const int op = (tok >= 0 && tok < g.V) ? g.token_op[tok] : -1;
      live[i] = (op >= 0 && !(op >= N2NMN_OP_EXIST && op <= N2NMN_OP_DESCRIBE)) ? 1 : 0;
    }
    cnt += live[i];
  }
  // exclusive scan of cnt over the 256 threads
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int v = __shfl_up(incl, d, 64);
    if (lane >= d) incl += v;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = incl - cnt, total = 0;
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) { if (ww < w) base += wsum[ww]; total += wsum[ww]; }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = 4 * tid + i;
    if (j < n_old) {
      if (live[i]) src_of[base] = j; else dead_of[j - base] = j;     // (j - base = finished rows in front of j)
      base += live[i];
    }
  }
  __syncthreads();
  if (blockIdx.y == 0) {                    // the permutation: 64 positions per workgroup column
    if (blockIdx.x == 0 && tid == 0) *g.n_new = total;
    const int p = blockIdx.x * 64 + lane;
    if (w == 0 && p < N) {
      const int jo = p < total ? src_of[p] : (p < n_old ? dead_of[p - total] : p);
      g.perm_new[p] = g.perm_old ? g.perm_old[jo] : jo;
    }
  }
  const int p = blockIdx.x * 64 + lane;
  if (p >= total) return;
  const int jo = src_of[p];
  const int k4n = g.L / 4;
  for (int k4 = blockIdx.y * 4 + w; k4 < k4n; k4 += gridDim.y * 4) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(g.src[i] + ((size_t)k4 * g.R + jo) * 4);
      *reinterpret_cast<float4*>(g.dst[i] + ((size_t)k4 * g.R + p) * 4) = v;
    }
  }
  if (g.srcb[0]) {
    const size_t plane = (size_t)(g.L / 8) * g.R * 8;
    const int k8n = g.L / 8;
    for (int k8 = blockIdx.y * 4 + w; k8 < k8n; k8 += gridDim.y * 4) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const uint4 v = *reinterpret_cast<const uint4*>(g.srcb[i] + pl * plane + ((size_t)k8 * g.R + jo) * 8);
          *reinterpret_cast<uint4*>(g.dstb[i] + pl * plane + ((size_t)k8 * g.R + p) * 8) = v;
        }
    }
  }
}

__global__ void dec_init_kernel(int32_t* state, int N, int T_dec) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < N) {
    state[n * 3 + 0] = 0; state[n * 3 + 1] = 0; state[n * 3 + 2] = T_dec;     // (:284)
  }
}

// word_vecs[t, n, :] = sum_tau atts[t, tau, n] * emb[seq[tau, n], :]   (nmn3_netgen_att.py:312)
// one workgroup per question: the question's T_enc embedding rows are staged once in LDS (the
// row index comes from LDS so the global loads of different tau are independent), then every
// thread owns output elements.  Also finishes neg_entropy = sum_t ent_t and log_seq_prob.
constexpr int WV_TGROUPS = 4;
__global__ __launch_bounds__(256) void word_vecs_kernel(const float* __restrict__ atts,
                                                        const int32_t* __restrict__ seq,
                                                        const float* __restrict__ emb, int T_dec,
                                                        int T_enc, int N, int E,
                                                        float* __restrict__ wv,
                                                        const float* __restrict__ tprobs,
                                                        const float* __restrict__ ent_t,
                                                        float* __restrict__ neg_entropy,
                                                        float* __restrict__ log_seq_prob) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* rows = smem;                       // [T_enc][E]
  float* at = rows + (size_t)T_enc * E;     // [tper][T_enc]
  int* idx = reinterpret_cast<int*>(at + (size_t)T_dec * T_enc);   // [T_enc]
  const int n = blockIdx.x, tid = threadIdx.x;
  const int tper = (T_dec + WV_TGROUPS - 1) / WV_TGROUPS;
  const int t0 = blockIdx.y * tper, t1 = min(T_dec, t0 + tper);
  for (int tau = tid; tau < T_enc; tau += 256) idx[tau] = seq[tau * N + n];
  for (int i = tid; i < (t1 - t0) * T_enc; i += 256) {
    const int t = i / T_enc, tau = i - t * T_enc;
    at[i] = atts[((size_t)(t0 + t) * T_enc + tau) * N + n];
  }
  __syncthreads();
  {   // the question's embedding rows, 16-B loads, all independent
    const int e4n = E >> 2;
    float4* rows4 = reinterpret_cast<float4*>(rows);
#pragma unroll 4
    for (int i = tid; i < T_enc * e4n; i += 256) {
      const int tau = i / e4n, e4 = i - tau * e4n;
      rows4[i] = reinterpret_cast<const float4*>(emb + (size_t)idx[tau] * E)[e4];
    }
  }
  __syncthreads();
  for (int i = tid; i < (t1 - t0) * E; i += 256) {
    const int t = i / E, e = i - t * E;
    float s = 0.f;
#pragma unroll 5
    for (int tau = 0; tau < T_enc; ++tau) s += at[t * T_enc + tau] * rows[tau * E + e];
    wv[((size_t)(t0 + t) * N + n) * E + e] = s;
  }
  if (blockIdx.y == 0 && tid == 0) {
    float s = 0.f;
    for (int t = 0; t < T_dec; ++t) s += ent_t[t * N + n];
    neg_entropy[n] = s;                      // loop_state[3] + neg_entropy  (:297)
  }
  if (blockIdx.y == 0 && log_seq_prob && tid == 64) {   // models_clevr/nmn3_model.py:46
    float s = 0.f;
    for (int t = 0; t < T_dec; ++t) s += logf(tprobs[t * N + n]);
    log_seq_prob[n] = s;
  }
}

}  // namespace

void launch_lstm_step(const LstmJob* jobs, int njobs, int N, int L, int rows_per_wg,
                      hipStream_t s, int wide) {
  LstmJobs js;
  for (int i = 0; i < 2; ++i) {
    if (i < njobs) js.j[i] = jobs[i];
    else { js.j[i] = LstmJob{}; js.j[i].active = 0; }
  }
  int nt = 0;
  bool cells = true;
  for (int i = 0; i < njobs; ++i) {
    nt = jobs[i].ntiles > nt ? jobs[i].ntiles : nt;
    const int nch = jobs[i].K / (LSTM_WAVES * 16);
    cells = cells && jobs[i].mode == 0 && jobs[i].ntiles % 2 == 0 && (nch == 4 || nch == 8) &&
            jobs[i].K % (LSTM_WAVES * 16) == 0;
  }
  if (wide >= 2 && N >= 128 && lstm_tile_supported(jobs, njobs, L)) {
    static const int stages = N2NMN_KNOB_INT("N2NMN_TILE_STAGES", 4);
    launch_lstm_tile(jobs, njobs, N, L, wide >= 3 ? wide : stages, s);
    return;
  }
  if (wide && cells) {
    dim3 grid(nt / 2, njobs, (N + 31) / 32);
    hipLaunchKernelGGL(lstm_step_wide_kernel<2>, grid, dim3(LSTM_THREADS), 0, s, js, N, L);
    return;
  }
  if (rows_per_wg == 32) {
    dim3 grid(nt, njobs, (N + 31) / 32);
    hipLaunchKernelGGL((lstm_step_kernel<2, 0>), grid, dim3(LSTM_THREADS), 0, s, js, N, L);
  } else {
    dim3 grid(nt, njobs, (N + 63) / 64);
    hipLaunchKernelGGL((lstm_step_kernel<4, 0>), grid, dim3(LSTM_THREADS), 0, s, js, N, L);
  }
}

__global__ void empty_kernel(int) {}

__global__ void unpack_h_kernel(const float* __restrict__ src, float* __restrict__ dst, int N,
                                int L, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * L) {
    const int n = i / L, k = i - n * L;
    dst[i] = src[((size_t)(k >> 2) * R + n) * 4 + (k & 3)];
  }
}

__global__ void unpack_h2_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                 float* __restrict__ dst, int N, int L, int R) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N * 2 * L) {
    const int n = i / (2 * L), k2 = i - n * 2 * L;
    const float* src = k2 < L ? a : b;
    const int k = k2 < L ? k2 : k2 - L;
    dst[i] = src[((size_t)(k >> 2) * R + n) * 4 + (k & 3)];
  }
}

// debug: the same launch with a kernel variant (see lstm_mma DBG); variant 5 = empty kernel
void launch_lstm_step_dbg(const LstmJob* jobs, int njobs, int N, int L, int rows_per_wg,
                          int variant, hipStream_t s) {
  LstmJobs js;
  for (int i = 0; i < 2; ++i) {
    if (i < njobs) js.j[i] = jobs[i];
    else { js.j[i] = LstmJob{}; js.j[i].active = 0; }
  }
  int nt = 0;
  for (int i = 0; i < njobs; ++i) nt = jobs[i].ntiles > nt ? jobs[i].ntiles : nt;
  dim3 g4(nt, njobs, (N + 63) / 64), g2(nt, njobs, (N + 31) / 32), b(LSTM_THREADS);
  if (variant == 5) { hipLaunchKernelGGL(empty_kernel, g4, b, 0, s, 0); return; }
  if (rows_per_wg == 32) {
    switch (variant) {
      case 1: hipLaunchKernelGGL((lstm_step_kernel<2, 1>), g2, b, 0, s, js, N, L); break;
      case 2: hipLaunchKernelGGL((lstm_step_kernel<2, 2>), g2, b, 0, s, js, N, L); break;
      case 3: hipLaunchKernelGGL((lstm_step_kernel<2, 3>), g2, b, 0, s, js, N, L); break;
      case 4: hipLaunchKernelGGL((lstm_step_kernel<2, 4>), g2, b, 0, s, js, N, L); break;
      default: hipLaunchKernelGGL((lstm_step_kernel<2, 0>), g2, b, 0, s, js, N, L); break;
    }
  } else {
    switch (variant) {
      case 1: hipLaunchKernelGGL((lstm_step_kernel<4, 1>), g4, b, 0, s, js, N, L); break;
      case 2: hipLaunchKernelGGL((lstm_step_kernel<4, 2>), g4, b, 0, s, js, N, L); break;
      case 3: hipLaunchKernelGGL((lstm_step_kernel<4, 3>), g4, b, 0, s, js, N, L); break;
      case 4: hipLaunchKernelGGL((lstm_step_kernel<4, 4>), g4, b, 0, s, js, N, L); break;
      case 6: hipLaunchKernelGGL((lstm_step_kernel<4, 6>), g4, b, 0, s, js, N, L); break;
      case 7: hipLaunchKernelGGL((lstm_step_kernel<4, 7>), g4, b, 0, s, js, N, L); break;
      default: hipLaunchKernelGGL((lstm_step_kernel<4, 0>), g4, b, 0, s, js, N, L); break;
    }
  }
}

void launch_unpack_h(const float* src, float* dst, int N, int L, int R, hipStream_t s) {
  hipLaunchKernelGGL(unpack_h_kernel, dim3((N * L + 255) / 256), dim3(256), 0, s, src, dst, N, L,
                     R);
}

void launch_unpack_h2(const float* a, const float* b, float* dst, int N, int L, int R,
                      hipStream_t s) {
  hipLaunchKernelGGL(unpack_h2_kernel, dim3((N * 2 * L + 255) / 256), dim3(256), 0, s, a, b, dst, N,
                     L, R);
}

template <int KI, int TS>
static void launch_dec_multi(const DecStepArgs& a, int nsteps, hipStream_t s) {
  const int ncol = a.L / 4;
  const int nsplit = 256 / ncol > 0 ? 256 / ncol : 1;
  const size_t smem = sizeof(float) * ((size_t)(2 + nsplit) * TS * a.L + (size_t)TS * ((a.T + 3) & ~3) +
                                       (size_t)4 * TS * MAXV + 16);
  if (smem > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_attn_multi_kernel<KI, TS>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  hipLaunchKernelGGL((dec_attn_multi_kernel<KI, TS>), dim3(a.N, (nsteps + TS - 1) / TS), dim3(256),
                     smem, s, a, nsteps);
}

template <int KI>
static bool launch_dec_question(const DecStepArgs& a, int nsteps, hipStream_t s) {
  const int Tp = (a.T + 3) & ~3;
  const size_t smem = sizeof(float) * ((size_t)std::max(a.T, nsteps) * a.L + 32 * (size_t)Tp + 16 * 4 * MAXV);
  if (smem > 150 * 1024 || (size_t)a.T * a.L > (size_t)3 * KI * 4096 ||
      (size_t)nsteps * (a.L + 16 * MAXV) > (size_t)std::max(a.T, nsteps) * a.L) return false;
  static std::atomic<uint64_t> attr{0};
  ensure_dynamic_lds(reinterpret_cast<const void*>(dec_attn_question_kernel<KI>), 150 * 1024, attr);
  hipLaunchKernelGGL((dec_attn_question_kernel<KI>), dim3(a.N), dim3(1024), smem, s, a, nsteps);
  return true;
}

void launch_dec_attn(const DecStepArgs& a, int nsteps, hipStream_t s) {
  // every step of a question in one workgroup: needs <= 20 steps (4 groups of 5), enough questions to
  // fill the chip with one workgroup each, and the question's rows in LDS
  static const int qk = N2NMN_KNOB_INT("N2NMN_DEC_ATTN_Q", 1);
  if (qk && nsteps > 1 && nsteps <= 20 && a.N >= 128 && !a.uni && !a.forced && a.use_gt) {
    if (a.L == 512 && launch_dec_question<2>(a, nsteps, s)) return;
  }
  if (nsteps > 1 && a.L % 256 == 0 && a.L <= 1024) {
    // steps per workgroup: as many as keep >= ~2 workgroups per CU in the launch
    const int groups4 = (nsteps + 3) / 4;
    const bool four = (long)a.N * groups4 >= 512 || nsteps <= 2;
    if (a.L <= 512) {
      if (four) launch_dec_multi<2, 4>(a, nsteps, s);
      else launch_dec_multi<2, 2>(a, nsteps, s);
    } else {
      if (four) launch_dec_multi<4, 4>(a, nsteps, s);
      else launch_dec_multi<4, 2>(a, nsteps, s);
    }
    return;
  }
  const int nt = nsteps > 1 ? 256 : 1024;
  const int ncol = a.L / 4;
  int nsplit = nt / ncol;
  nsplit = nsplit < 1 ? 1 : (nsplit > 8 ? 8 : nsplit);
  const size_t smem = sizeof(float) * ((2 + (size_t)nsplit) * a.L + ((a.T + 3) & ~3) +
                                       (size_t)(nt / 64) * MAXV + 16);
  if (nsteps > 1) {
    hipLaunchKernelGGL(dec_attn_kernel<256>, dim3(a.N, nsteps), dim3(256), smem, s, a);
    return;
  }
  // one step per launch (greedy / sampled decoding): the one-pass form; the diagnostic build's N2NMN_DEC_ATTN_SEQ=0 keeps the
  // three-pass kernel
  static const bool seq_on = N2NMN_KNOB_INT("N2NMN_DEC_ATTN_SEQ", 1) != 0;
  if (seq_on && a.L == 512) {       // (lstm_dim 1024 needs 128 VGPRs in this form: one workgroup per CU)
    const size_t sm = sizeof(float) * ((2 + 16) * (size_t)a.L + ((a.T + 3) & ~3) + 4 * 16 + 16 * MAXV + 16);
    hipLaunchKernelGGL(dec_attn_seq_kernel<2>, dim3(a.N), dim3(1024), sm, s, a);
    return;
  }
  hipLaunchKernelGGL(dec_attn_kernel<1024>, dim3(a.N, 1), dim3(1024), smem, s, a);
}

void launch_enc_rows(const int32_t* seq_len, int T, int N, int32_t* rows, int32_t* count,
                     hipStream_t s) {
  hipLaunchKernelGGL(enc_rows_kernel, dim3((T * N + 1023) / 1024), dim3(1024), 0, s, seq_len, T, N, rows,
                     count);
}

void launch_enc_prepare(const int32_t* seq_len, int N, int T, int32_t* perm, int32_t* n_active,
                        float* zero, size_t zero_floats, hipStream_t s, int32_t* zero_int) {
  const size_t z4 = zero_floats / 4;             // the state block is a multiple of 4 floats
  static const bool sort_on = N2NMN_KNOB_INT("N2NMN_ENC_PREPARE_SORT", 1) != 0;
  if (sort_on && N <= 1024 && T <= 63) {
    // block 0 ranks, the others clear the state block (16 KB per workgroup and trip)
    const int zb = zero ? (int)std::min<size_t>(255, (z4 + 4095) / 4096) : 0;
    hipLaunchKernelGGL(enc_prepare_sort_kernel, dim3(1 + zb), dim3(1024), 0, s, seq_len, N, T, perm,
                       n_active, reinterpret_cast<float4*>(zero), zero ? z4 : 0, zero_int);
    return;
  }
  int blocks = zero ? (int)std::min<size_t>(256, (z4 + 1023) / 1024 + 1) : 1;
  blocks = std::max(blocks, std::min(64, (N + 255) / 256 + T / 4));
  hipLaunchKernelGGL(enc_prepare_kernel, dim3(blocks), dim3(256), 0, s, seq_len, N, T, perm, n_active,
                     reinterpret_cast<float4*>(zero), zero ? z4 : 0, zero_int);
}

bool dec_seq_retire_supported(const DecStepArgs& a) {     // the one-step launches go to dec_attn_seq_kernel
  static const bool seq_on = N2NMN_KNOB_INT("N2NMN_DEC_ATTN_SEQ", 1) != 0;
  return seq_on && a.L == 512 && a.N <= 1024;
}

void launch_dec_compact(const int32_t* tokens, const int32_t* token_op, int V, const int32_t* perm_old,
                        const int32_t* n_old, int32_t* perm_new, int32_t* n_new, const float* const src[4],
                        float* const dst[4], const uint16_t* const srcb[2], uint16_t* const dstb[2], int N, int L,
                        int R, hipStream_t s) {
  CompactArgs g{};
  for (int i = 0; i < 2; ++i) { g.srcb[i] = srcb ? srcb[i] : nullptr; g.dstb[i] = dstb ? dstb[i] : nullptr; }
  g.tokens = tokens; g.token_op = token_op; g.V = V; g.perm_old = perm_old; g.n_old = n_old;
  g.perm_new = perm_new; g.n_new = n_new;
  for (int i = 0; i < 4; ++i) { g.src[i] = src[i]; g.dst[i] = dst[i]; }
  g.N = N; g.L = L; g.R = R;
  hipLaunchKernelGGL(dec_compact_kernel, dim3((N + 63) / 64, std::min(32, std::max(1, L / 16))), dim3(256), 0, s, g);
}

void launch_dec_len(const int32_t* tokens, const int32_t* token_op, int V, int T_dec, int N,
                    int32_t* dec_len, hipStream_t s) {
  hipLaunchKernelGGL(dec_len_kernel, dim3((N + 255) / 256), dim3(256), 0, s, tokens, token_op, V, T_dec, N,
                     dec_len);
}

void launch_gather_state(const float* const src[4], float* const dst[4], const uint16_t* const srcb[2],
                         uint16_t* const dstb[2], const int32_t* perm, int N, int L, int R, hipStream_t s) {
  GatherArgs g{};
  for (int i = 0; i < 4; ++i) { g.src[i] = src[i]; g.dst[i] = dst[i]; }
  for (int i = 0; i < 2; ++i) { g.srcb[i] = srcb ? srcb[i] : nullptr; g.dstb[i] = dstb ? dstb[i] : nullptr; }
  g.perm = perm; g.N = N; g.L = L; g.R = R;
  hipLaunchKernelGGL(gather_state_kernel, dim3((N + 63) / 64, std::min(32, std::max(1, L / 16))), dim3(256), 0, s, g);
}

// teacher-forced launches whose attention kernel honours DecStepArgs::dec_len: dec_attn_question_kernel
// (lstm_dim 512) or dec_attn_multi_kernel (launch_dec_attn's choice for every other lstm_dim % 256 == 0)
bool dec_len_supported(const DecStepArgs& a, int nsteps) {
  return dec_question_supported(a, nsteps) ||
         (nsteps > 1 && a.L % 256 == 0 && a.L <= 1024 && !a.uni && !a.forced && a.use_gt);
}

bool dec_question_supported(const DecStepArgs& a, int nsteps) {
  const int Tp = (a.T + 3) & ~3;
  const size_t smem = sizeof(float) * ((size_t)std::max(a.T, nsteps) * a.L + 32 * (size_t)Tp + 16 * 4 * MAXV);
  return nsteps > 1 && nsteps <= 20 && a.N >= 128 && !a.uni && !a.forced && a.use_gt && a.L == 512 &&
         !(smem > 150 * 1024 || (size_t)a.T * a.L > (size_t)3 * 2 * 4096 ||
           (size_t)nsteps * (a.L + 16 * MAXV) > (size_t)std::max(a.T, nsteps) * a.L);
}

void launch_dec_init(int32_t* state, int N, int T_dec, hipStream_t s) {
  hipLaunchKernelGGL(dec_init_kernel, dim3((N + 63) / 64), dim3(64), 0, s, state, N, T_dec);
}

void launch_word_vecs(const float* atts, const int32_t* seq, const float* emb, int T_dec,
                      int T_enc, int N, int E, float* word_vecs, const float* tprobs,
                      const float* ent_t, float* neg_entropy, float* log_seq_prob,
                      hipStream_t s) {
  const size_t smem =
      sizeof(float) * ((size_t)T_enc * E + (size_t)T_dec * T_enc + (size_t)T_enc + 4);
  hipLaunchKernelGGL(word_vecs_kernel, dim3(N, WV_TGROUPS), dim3(256), smem, s, atts, seq, emb, T_dec, T_enc,
                     N, E, word_vecs, tprobs, ent_t, neg_entropy, log_seq_prob);
}

}  // namespace n2nmn
