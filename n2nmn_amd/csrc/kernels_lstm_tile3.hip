// lstm_tile3_kernel: the recurrent LSTM step of kernels_lstm_tile.hip on the bf16 matrix cores, at fp32
// accuracy, by splitting both operands three ways (opt-in: N2NMN_MODE_THROUGHPUT_BF16X3).
//
//   z[n, :] = [x, h][n, :] . W + b ;  i, j, f, o = split(z) ;
//   c' = c * sig(f + 1) + sig(i) * tanh(j) ;  h' = tanh(c') * sig(o)
// (models_clevr/nmn3_netgen_att.py:17-44, 91-96; TF 1.0.0 BasicLSTMCell / dynamic_rnn semantics per
// SURVEY.md Appendix A.1-A.2) -- same job description (LstmJob), same state layouts, same epilogue.
//
// Why: the pass is 88 % fp32 MFMA at 0.6-0.8 of a 157 TFLOP/s peak; gfx950 has no xf32 / TF32 form and
// bf16 MFMA runs 16x faster.  An fp32 value is EXACTLY the sum of three bf16 values
//     x = hi + mid + lo,   hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid)
// (round to nearest: |mid| <= 2^-9 |x|, |lo| <= 2^-18 |x|; the subtractions are exact in fp32), so
//     w . h = (wh + wm + wl)(hh + hm + hl)
//           = wh hh + wh hm + wm hh + wh hl + wl hh + wm hm   (six bf16 MFMAs, fp32 accumulate)
//             + wm hl + wl hm + wl hl                          (dropped: <= 2^-26 |w h| each)
// Every bf16 x bf16 product is exact in fp32 and the accumulator is fp32, so what is lost is of the
// order of one fp32 rounding per product -- the same class as the fp32 MFMA's own fma chain.  Six
// `v_mfma_f32_16x16x32_bf16` (16 cycles, k = 32) replace eight `v_mfma_f32_16x16x4_f32` (32 cycles, k = 4)
// per 16 x 16 x 32 block: 96 instead of 256 matrix-pipe cycles.
//
// Operands.  The weights are split once per commit (pack_tiles64_b3_kernel).  h is split by the step
// that PRODUCES it (the cell-update epilogue below writes three bf16 planes beside the fp32 state),
// not by the 64 workgroups that consume each element.  Both live in HBM in MFMA fragment order: a lane
// of `v_mfma_f32_16x16x32_bf16` holds 8 consecutive k of one row / unit = 16 bytes, and
//   state planes   [3][L/8][R][8]                 (k8 group, row): 16 lanes x 16 B = 256 contiguous bytes
//   weight planes  [L/16][K/32][3][4 gates][64 lanes][8]           a stage of a tile is 12 contiguous KiB
// so one LDS-DMA instruction (global_load_lds_dwordx4, 1 KiB per wave) moves one fragment of all 64
// lanes and the LDS image is read back conflict-free as it lies (ds_read_b128, 256 contiguous bytes per
// 16 lanes).
//
// Workgroup = 64 rows x 64 gate columns (16 hidden units x i, j, f, o) over the whole K, 8 waves:
// 4 row groups x 2 gate pairs (waves 0-3: i, j; waves 4-7: f, o -- every wave then needs the row group's
// three h planes and only its own two gates' weight planes: 9 ds_read_b128 per 12 MFMAs).  Stage = 32 k
// = 12 KiB of h planes + 12 KiB of weight planes, ring of NS stages by LDS-DMA with counted vmcnt and
// one raw s_barrier per stage, as in lstm_tile_kernel.  The gate pairs meet in LDS after the loop and
// waves 0-3 run the register-local cell update.
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int T3_UNITS = 16, T3_BK = 32;
constexpr int T3_W_IMAGE = 3 * 4 * 1024;            // planes x gates x 1 KiB
// RG = row groups of 16 rows per workgroup (4: 64 rows, 8 waves; 8: 128 rows, 16 waves -- the weight
// planes of a stage then serve twice the rows: 36 KiB per 128 rows instead of 2 x 24 KiB)
template <int RG> struct T3 {
  static constexpr int ROWS = 16 * RG, WAVES = 2 * RG, THREADS = WAVES * 64;
  static constexpr int H_IMAGE = 3 * RG * 1024;     // planes x row groups x 1 KiB
  static constexpr int STAGE = H_IMAGE + T3_W_IMAGE;
  static constexpr int NPIECE = 3 * RG + 12;        // LDS-DMA instructions per stage
  static constexpr int PBASE = NPIECE / WAVES, PEXTRA = NPIECE % WAVES;   // per wave; waves < PEXTRA one more
};

struct LstmJobs3 {
  LstmJob j[2];
};

__device__ __forceinline__ void glds16b(const void* base, uint32_t voff, uint32_t lds) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(base), "s"(lds)
      : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vm3() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// VAR (n2nmn_debug_lstm_bench only): 1 = no DMA, 2 = no MFMA, 3 = DMA + barriers only
template <int NS, int RG, int VAR = 0>
__global__ __launch_bounds__(T3<RG>::THREADS, RG == 4 ? 2 : 1) void lstm_tile3_kernel(LstmJobs3 jobs, int N,
                                                                                      int L, int nrb,
                                                                                      int njobs) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int T3_ROWS = T3<RG>::ROWS, T3_H_IMAGE = T3<RG>::H_IMAGE, T3_STAGE = T3<RG>::STAGE;
  constexpr int PMAX = T3<RG>::PBASE + (T3<RG>::PEXTRA ? 1 : 0);
  const int ntile = L / T3_UNITS;
  // ---- id -> (job, row block, column tile): as lstm_tile_kernel (dense over the active row blocks,
  // the K = 2L job first, column tile fastest so that tile % 8 is the workgroup's XCD)
  int nab[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const LstmJob& jq = jobs.j[j];
    int n = 0;
    if (j < njobs && jq.active) {
      const int na = jq.n_active ? *jq.n_active : N;
      n = (min(max(na, 0), N) + T3_ROWS - 1) / T3_ROWS;
    }
    nab[j] = n;
  }
  int rem = blockIdx.x / ntile;
  const int ct = blockIdx.x - rem * ntile;
  int jsel = -1, rb = 0;
  bool zero_fill = false;
#pragma unroll
  for (int j = 1; j >= 0; --j)
    if (jsel < 0) { if (rem < nab[j]) { jsel = j; rb = rem; } else rem -= nab[j]; }
#pragma unroll
  for (int j = 1; j >= 0; --j)
    if (jsel < 0 && j < njobs && jobs.j[j].active && jobs.j[j].out_seq) {
      if (rem < nrb - nab[j]) { jsel = j; rb = nab[j] + rem; zero_fill = true; } else rem -= nrb - nab[j];
    }
  if (jsel < 0) return;
  const LstmJob& jb = jobs.j[jsel];

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = w % RG, gp = w / RG;               // row group of the wave, and its gate pair
  const int row0 = rb * T3_ROWS;
  const int R = jb.hp_R;
  const int nact = jb.n_active ? *jb.n_active : N;

  // ---- the lane's place in the epilogue: row lr of the wave, units 16 ct + 4 q .. + 3 ----------
  const int lr = lane & 15, q = lane >> 4;
  const int gr = row0 + 16 * wr + lr;
  const bool eact = gr < N;
  const int grc = eact ? gr : N - 1;
  const int t4 = 4 * ct + q;                         // 4-unit group = float4 element of the state
  if (zero_fill) {
    if (gp == 0 && eact) {
      const int zr = jb.perm ? jb.perm[grc] : grc;
      *reinterpret_cast<float4*>(jb.out_seq + (size_t)zr * L + 4 * t4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }

  // ---- operand stream ----------------------------------------------------------------------------
  const int K = jb.K, nst = K / T3_BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t plane_bytes = (uint32_t)(L / 8) * (uint32_t)R * 16u;      // one plane of a state buffer
  // this wave's pieces of every stage: piece ids pc = w, w + WAVES, ... < NPIECE
  //   pc < 3 RG: h plane p = pc / RG of row group pc % RG;   pc >= 3 RG: weight plane (pc - 3 RG) / 4, gate % 4
  uint32_t pbase[PMAX], pstep[PMAX], plds[PMAX];
  bool pish[PMAX];
  const uint32_t wtile = (uint32_t)ct * (uint32_t)nst * (uint32_t)T3_W_IMAGE;
#pragma unroll
  for (int i = 0; i < PMAX; ++i) {
    const int pc = min(w + T3<RG>::WAVES * i, T3<RG>::NPIECE - 1);
    pish[i] = pc < 3 * RG;
    if (pc < 3 * RG) {
      const int p = pc / RG, rg = pc % RG;
      const int arow = min(row0 + 16 * rg + (lane & 15), N - 1);
      // k8 group (lane >> 4) of the stage, row arow: 16 bytes
      pbase[i] = (uint32_t)p * plane_bytes + ((uint32_t)(lane >> 4) * (uint32_t)R + (uint32_t)arow) * 16u;
      pstep[i] = 4u * (uint32_t)R * 16u;                        // four k8 groups per stage
      plds[i] = (uint32_t)pc * 1024u;
    } else {
      const int idx = pc - 3 * RG;
      pbase[i] = wtile + (uint32_t)idx * 1024u + (uint32_t)lane * 16u;
      pstep[i] = (uint32_t)T3_W_IMAGE;
      plds[i] = (uint32_t)T3_H_IMAGE + (uint32_t)idx * 1024u;
    }
  }
  const uint16_t* const A0p = jb.A0b;
  const uint16_t* const A1p = jb.A1b;
  const uint16_t* const Wp = jb.Wb3;
  const int sL = L / T3_BK;                           // stages that read A0 (the rest read A1)
  auto issue = [&](auto np_tag, int s) {
    constexpr int NP = decltype(np_tag)::value;
    const uint32_t slot = lds0 + (uint32_t)(s % NS) * T3_STAGE;
    const bool lo = s < sL;
    const uint32_t sh = (uint32_t)(lo ? s : s - sL);
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      if (pish[i]) glds16b(lo ? A0p : A1p, pbase[i] + sh * pstep[i], slot + plds[i]);
      else glds16b(Wp, pbase[i] + (uint32_t)s * pstep[i], slot + plds[i]);
    }
  };
  // waves below PEXTRA issue one piece more per stage than the others: the counted waits differ, so
  // the stage loop exists per piece count (wave-uniform choice)
  const bool extra = w < T3<RG>::PEXTRA;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) {
    if (VAR != 1) {
      if (extra) issue(std::integral_constant<int, PMAX>{}, s);
      else issue(std::integral_constant<int, T3<RG>::PBASE>{}, s);
    }
  }

  // ---- epilogue operands (waves of gate pair 0 own the cell update): fetched under the DMA prologue
  int orow = grc;
  float4 add[4];
  float4 c_old = make_float4(0.f, 0.f, 0.f, 0.f), h_prev = c_old;
  bool masked = false;
  const size_t sidx = ((size_t)t4 * R + grc) * 4;
  if (gp == 0) {
    if (jb.perm) orow = jb.perm[grc];                // original row of state row gr
    const float* ar;
    if (jb.xtab) {
      const int xi = jb.xidx ? jb.xidx[orow] : jb.xidx_const;
      ar = jb.xtab + (size_t)xi * 4 * L + 16 * t4;    // tile column order: [4-unit group][gate][unit]
    } else {
      ar = jb.bias + 16 * t4;
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) add[g] = *reinterpret_cast<const float4*>(ar + 4 * g);
    c_old = *reinterpret_cast<const float4*>(jb.c_in + sidx);
    masked = jb.seq_len && jb.t >= jb.seq_len[orow];      // dynamic_rnn past the length (A.2)
    if (masked) h_prev = *reinterpret_cast<const float4*>(jb.h_old + sidx);
  }

  f32x4 acc[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool wact = row0 + 16 * wr < nact;          // wave-uniform: any active row in this wave?

  // operands of one stage in registers: three h planes, two gates x three weight planes (36 VGPRs)
  struct Grp { uint4 h[3]; uint4 wq[2][3]; };
  const uint4* const S0 = reinterpret_cast<const uint4*>(smem) + lane;
  auto fetch = [&](Grp& g, int slot) {
    const uint4* st = S0 + slot * (T3_STAGE / 16);
#pragma unroll
    for (int p = 0; p < 3; ++p) g.h[p] = st[(p * RG + wr) * 64];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int p = 0; p < 3; ++p) g.wq[gi][p] = st[(T3_H_IMAGE / 16) + (p * 4 + 2 * gp + gi) * 64];
  };
  // the six products of one gate: small terms first, the leading term last
  auto mma_gate = [&](const Grp& g, int gi) {
    if (VAR == 2) {
      asm volatile("" ::"v"(g.wq[gi][0].x), "v"(g.wq[gi][1].y), "v"(g.wq[gi][2].z), "v"(g.h[0].x),
                   "v"(g.h[1].y), "v"(g.h[2].w));
      return;
    }
    const bf16x8 wh = __builtin_bit_cast(bf16x8, g.wq[gi][0]), wm = __builtin_bit_cast(bf16x8, g.wq[gi][1]),
                 wl = __builtin_bit_cast(bf16x8, g.wq[gi][2]);
    const bf16x8 hh = __builtin_bit_cast(bf16x8, g.h[0]), hm = __builtin_bit_cast(bf16x8, g.h[1]),
                 hl = __builtin_bit_cast(bf16x8, g.h[2]);
    f32x4 c = acc[gi];
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, hh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, hm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, hh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, hh, c, 0, 0, 0);
    acc[gi] = c;
  };
#define N3_PIN() __builtin_amdgcn_sched_barrier(0)
  auto stages = [&](auto work_tag, auto np_tag) {
    constexpr bool WORK = decltype(work_tag)::value;
    constexpr int NP = decltype(np_tag)::value;
    auto sync_stage = [&](int next) {
      const int behind = nst - 1 - next;             // stages after `next`
      if (behind >= NS - 3) wait_vm3<NP * (NS - 3)>();
      else wait_vm3<0>();
      __builtin_amdgcn_s_barrier();
    };
    // One step: the 12 MFMAs of stage `cur` (in registers); woven in: the refill of the slot behind and
    // the LDS reads of stage `next`
    auto step = [&](const Grp& cur, Grp& nxt, int next, bool more) {
      const int refill = next + NS - 2;
      const bool dma = more && refill < nst && VAR != 1;
      if (more) sync_stage(next);
      if (WORK) { mma_gate(cur, 0); N3_PIN(); }
      if (dma) issue(np_tag, refill);
      if (WORK) {
        N3_PIN();
        if (more) fetch(nxt, next % NS);
        N3_PIN();
        mma_gate(cur, 1);
      }
    };
    wait_vm3<NP * (NS - 2)>();                     // stage 0 (the oldest of the NS - 1 in flight)
    __builtin_amdgcn_s_barrier();
    Grp P{}, Q{};
    if (WORK) fetch(P, 0);
    int i = 0;
    for (; i + 2 < nst; i += 2) {                  // nst is even: two stages per trip, P / Q static
      step(P, Q, i + 1, true);
      step(Q, P, i + 2, true);
    }
    step(P, Q, i + 1, true);                       // the last two stages
    step(Q, P, 0, false);
  };
#undef N3_PIN
  const bool work = wact && VAR != 3;
  if (T3<RG>::PEXTRA && extra) {
    if (work) stages(std::true_type{}, std::integral_constant<int, PMAX>{});
    else stages(std::false_type{}, std::integral_constant<int, PMAX>{});
  } else {
    if (work) stages(std::true_type{}, std::integral_constant<int, T3<RG>::PBASE>{});
    else stages(std::false_type{}, std::integral_constant<int, T3<RG>::PBASE>{});
  }

  // ---- the gate pairs meet: waves 4-7 park f, o in LDS (every DMA has landed, every stage has been
  // read: the ring is free), waves 0-3 take them ------------------------------------------------------
  __builtin_amdgcn_s_barrier();
  float4* red = reinterpret_cast<float4*>(smem) + (size_t)wr * 2 * 64 + lane;
  if (gp == 1) {
#pragma unroll
    for (int t = 0; t < 2; ++t) red[t * 64] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
  __syncthreads();
  if (gp == 1 || !eact) return;
  const float4 zf4 = red[0], zo4 = red[64];
  const float zfv[4] = {zf4.x, zf4.y, zf4.z, zf4.w}, zov[4] = {zo4.x, zo4.y, zo4.z, zo4.w};

  // ---- cell update: lane = (row, 4 units), acc[g][r] = z of gate g, unit 4q + r ------------------
  float cn[4], hn[4];
  const float co[4] = {c_old.x, c_old.y, c_old.z, c_old.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float zi = acc[0][r] + (r == 0 ? add[0].x : r == 1 ? add[0].y : r == 2 ? add[0].z : add[0].w);
    const float zj = acc[1][r] + (r == 0 ? add[1].x : r == 1 ? add[1].y : r == 2 ? add[1].z : add[1].w);
    const float zf = zfv[r] + (r == 0 ? add[2].x : r == 1 ? add[2].y : r == 2 ? add[2].z : add[2].w);
    const float zo = zov[r] + (r == 0 ? add[3].x : r == 1 ? add[3].y : r == 2 ? add[3].z : add[3].w);
    const float gi = fast_sigmoid(zi), gj = fast_tanh(zj), gf = fast_sigmoid(zf + 1.0f), go = fast_sigmoid(zo);
    cn[r] = co[r] * gf + gi * gj;
    hn[r] = fast_tanh(cn[r]) * go;
  }
  float4 c4 = make_float4(cn[0], cn[1], cn[2], cn[3]);
  float4 h4 = make_float4(hn[0], hn[1], hn[2], hn[3]);
  float4 o4 = h4;
  if (masked) { c4 = c_old; h4 = h_prev; o4 = make_float4(0.f, 0.f, 0.f, 0.f); }
  *reinterpret_cast<float4*>(jb.c_out + sidx) = c4;
  *reinterpret_cast<float4*>(jb.h_new + sidx) = h4;
  // the planes the next step's MFMAs read: [3][L/8][R][8], units 4 t4 .. 4 t4 + 3 = half of a k8 group
  const size_t plane_elems = (size_t)(L / 8) * R * 8;
  const size_t poff = ((size_t)(t4 >> 1) * R + grc) * 8 + (size_t)(t4 & 1) * 4;
  if (jb.h_new_b) store_planes(jb.h_new_b, plane_elems, poff, h4);
  const size_t oidx = (size_t)orow * L + 4 * t4;
  if (jb.out_seq) *reinterpret_cast<float4*>(jb.out_seq + oidx) = o4;
  if (jb.h_drop) {                // dropped copy of the OUTPUT for the layer above
    const float4 dm = *reinterpret_cast<const float4*>(jb.drop + oidx);
    const float4 hd = make_float4(h4.x * dm.x, h4.y * dm.y, h4.z * dm.z, h4.w * dm.w);
    *reinterpret_cast<float4*>(jb.h_drop + sidx) = hd;
    if (jb.h_drop_b) store_planes(jb.h_drop_b, plane_elems, poff, hd);
  }
  if (jb.fin_c && jb.seq_len && jb.t == jb.seq_len[orow] - 1) {   // the row's last valid step
    const size_t fidx = ((size_t)t4 * R + orow) * 4;
    *reinterpret_cast<float4*>(jb.fin_c + fidx) = c4;
    *reinterpret_cast<float4*>(jb.fin_h + fidx) = h4;
    if (jb.fin_h_b)
      store_planes(jb.fin_h_b, plane_elems, ((size_t)(t4 >> 1) * R + orow) * 8 + (size_t)(t4 & 1) * 4, h4);
  }
}

// weights -> three bf16 planes in fragment order (see the header): one thread per 16-byte fragment
__global__ __launch_bounds__(256) void pack_tiles64_b3_kernel(const float* __restrict__ W, int ld, int row0,
                                                              int K, int L, uint16_t* __restrict__ dst) {
  const size_t total = (size_t)(L / 16) * (K / 32) * 4 * 64;        // fragments per plane
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63), g = (int)((i >> 6) & 3);
    const size_t r = i >> 8;
    const int s = (int)(r % (K / 32)), ct = (int)(r / (K / 32));
    const int unit = 16 * ct + (lane & 15), k0 = 32 * s + 8 * (lane >> 4);
    uint32_t pl[3][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float x0 = W[(size_t)(row0 + k0 + 2 * e) * ld + (size_t)g * L + unit];
      const float x1 = W[(size_t)(row0 + k0 + 2 * e + 1) * ld + (size_t)g * L + unit];
      split3(x0, x1, pl[0][e], pl[1][e], pl[2][e]);
    }
    // [ct][s][p][g][lane][8]
    const size_t base = ((size_t)ct * (K / 32) + s) * 3;
#pragma unroll
    for (int p = 0; p < 3; ++p)
      *reinterpret_cast<uint4*>(dst + (((base + p) * 4 + g) * 64 + lane) * 8) =
          make_uint4(pl[p][0], pl[p][1], pl[p][2], pl[p][3]);
  }
}

// planes of a whole fp32 state buffer [L/4][R][4] (debug / tests: the product path splits in the
// epilogue of the step that writes a state)
__global__ __launch_bounds__(256) void split_state_b3_kernel(const float* __restrict__ h, int L, int R,
                                                             uint16_t* __restrict__ planes) {
  const size_t total = (size_t)(L / 4) * R;
  const size_t plane_elems = (size_t)(L / 8) * R * 8;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int row = (int)(i % R), t4 = (int)(i / R);
    const float4 v = *reinterpret_cast<const float4*>(h + i * 4);
    store_planes(planes, plane_elems, ((size_t)(t4 >> 1) * R + row) * 8 + (size_t)(t4 & 1) * 4, v);
  }
}

template <int NS, int RG, int VAR = 0>
void launch_tile3(const LstmJobs3& js, int njobs, int N, int L, hipStream_t s) {
  static std::atomic<uint64_t> attr{0};
  const int lds = NS * T3<RG>::STAGE;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&lstm_tile3_kernel<NS, RG, VAR>), lds, attr);
  const int nrb = (N + T3<RG>::ROWS - 1) / T3<RG>::ROWS;
  const int grid = njobs * nrb * (L / T3_UNITS);
  hipLaunchKernelGGL((lstm_tile3_kernel<NS, RG, VAR>), dim3(grid), dim3(T3<RG>::THREADS), lds, s, js, N, L,
                     nrb, njobs);
}

}  // namespace

bool lstm_tile3_supported(const LstmJob* jobs, int njobs, int L) {
  if (njobs < 1 || njobs > 2 || L % (8 * T3_UNITS) != 0) return false;
  for (int i = 0; i < njobs; ++i) {
    const LstmJob& j = jobs[i];
    if (!j.active) continue;
    if (j.mode != 0 || !j.Wb3 || !j.A0b || j.hp_R <= 0 || j.a_rs != 4 || j.K % (2 * T3_BK) != 0 ||
        j.K / T3_BK < 8 || j.ntiles != L / 4 || (j.K != L && j.K != 2 * L) || (j.K == 2 * L && !j.A1b) ||
        j.save_gates)
      return false;
    // 32-bit LDS-DMA offsets
    if ((size_t)6 * L * j.hp_R >= ((size_t)1 << 31) || (size_t)j.K * 4 * L * 6 >= ((size_t)1 << 32)) return false;
  }
  return true;
}

void launch_lstm_tile3(const LstmJob* jobs, int njobs, int N, int L, hipStream_t s, int variant) {
  LstmJobs3 js;
  for (int i = 0; i < 2; ++i) {
    if (i < njobs) js.j[i] = jobs[i];
    else { js.j[i] = LstmJob{}; js.j[i].active = 0; }
  }
  // default: 128-row workgroups (16 waves, 4 stages of 36 KiB) when the launch has at least two
  // 128-row blocks per job; variants (n2nmn_debug_lstm_bench): 4xx = 64-row workgroups
  static const int dflt = N2NMN_KNOB_INT("N2NMN_TILE3_VARIANT", 0);
  // shipped: 128-row x 64-column workgroups of 16 waves (4 x 36 KiB) from 256 rows on, 64-row workgroups
  // below.  Measured and rejected (tools/rejected/lstm_tile3_variants.hip.txt has the kernels and numbers):
  // 128 x 128 workgroups and the register-split form -- both run 8 waves per CU and lose to
  // two-waves-per-SIMD bubbles what their smaller streams gain; s_setprio around the MFMA groups (no
  // change); the two gates' MFMA chains interleaved (slower: the LDS reads no longer sit between them).
  if (variant == 0) variant = dflt ? dflt : (N >= 256 ? 804 : 403);
  switch (variant) {
    case 404: launch_tile3<4, 4>(js, njobs, N, L, s); break;
    case 413: launch_tile3<3, 4, 1>(js, njobs, N, L, s); break;     // debug variants: no DMA
    case 423: launch_tile3<3, 4, 2>(js, njobs, N, L, s); break;     // no MFMA
    case 433: launch_tile3<3, 4, 3>(js, njobs, N, L, s); break;     // DMA + barriers only
    case 803: launch_tile3<3, 8>(js, njobs, N, L, s); break;
    case 804: launch_tile3<4, 8>(js, njobs, N, L, s); break;
    case 814: launch_tile3<4, 8, 1>(js, njobs, N, L, s); break;
    case 824: launch_tile3<4, 8, 2>(js, njobs, N, L, s); break;
    case 834: launch_tile3<4, 8, 3>(js, njobs, N, L, s); break;
    default: launch_tile3<3, 4>(js, njobs, N, L, s); break;         // 403
  }
}

void launch_pack_tiles64_b3(const float* W, int ld, int row0, int K, int L, uint16_t* dst, hipStream_t s) {
  const size_t total = (size_t)(L / 16) * (K / 32) * 4 * 64;
  hipLaunchKernelGGL(pack_tiles64_b3_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048)),
                     dim3(256), 0, s, W, ld, row0, K, L, dst);
}

void launch_split_state_b3(const float* h, int L, int R, uint16_t* planes, hipStream_t s) {
  const size_t total = (size_t)(L / 4) * R;
  hipLaunchKernelGGL(split_state_b3_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048)),
                     dim3(256), 0, s, h, L, R, planes);
}

}  // namespace n2nmn
