// lstm_tile_kernel: the recurrent LSTM step for MANY rows per launch (super-bucketed passes).
//
//   z[n, :] = [x, h][n, :] . W + b ;  i, j, f, o = split(z) ;
//   c' = c * sig(f + 1) + sig(i) * tanh(j) ;  h' = tanh(c') * sig(o)
// (models_clevr/nmn3_netgen_att.py:17-44, 91-96; TF 1.0.0 BasicLSTMCell / dynamic_rnn semantics per
// SURVEY.md Appendix A.1-A.2) -- same arithmetic, same job description (LstmJob) and same state
// layouts as lstm_step_kernel; what differs is how the operands reach the matrix cores.
//
// lstm_step_kernel splits K over the 8 waves of a 64 x 16 tile and streams every operand from L2
// straight into MFMA registers: nothing is reused inside a workgroup, so a 512-row step moves
// 502 MB from L2 to the CUs at 6.4 flop per byte and waits on it (profiles/r02_notes.md).  Here a
// workgroup owns 64 rows x 64 gate columns (16 hidden units x i,j,f,o) over the WHOLE K:
//   * operand tiles are staged through LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
//     instruction, no VGPRs, no ds_write pass) into a ring of NS stages of 32 k (8 KiB of weights +
//     8 KiB of h), NS - 1 stages in flight ahead of the MFMAs; counted s_waitcnt vmcnt + ONE raw
//     s_barrier per stage (a __syncthreads() would drain the DMA queue);
//   * both operands already live in HBM in k-interleaved order ([k/4][row or column][4]), so one
//     DMA instruction copies a contiguous KiB and the LDS image needs no swizzle: a ds_read_b128 of
//     16 consecutive lanes covers 256 contiguous bytes = all 64 banks once;
//   * the weight tile of a stage is read from LDS by all four waves (the reuse the K-split kernel
//     lacks): L2 -> CU traffic per 512-row two-layer step drops from 502 MB to 196 MB;
//   * wave w owns rows 16w .. 16w+15 and all four gates of the 16 units: the MFMA takes the WEIGHTS
//     as its A operand (M = hidden unit) and h as B (N = batch row), so lane (row = l % 16,
//     q = l / 16) ends up holding z of units 4q .. 4q+3 of ONE row for all four gates: the cell
//     update is register-local and c / h are read and written as the float4 elements of the
//     k-interleaved state layout ([L/4][R][4]), 256 contiguous bytes per 16 lanes.
// Grid: flat; id -> (job, row block, column tile) with the column tile fastest, so that tile % 8 is
// the XCD of the workgroup: an XCD's L2 keeps ITS eighth of the weights across the steps of a pass
// and only h (new every step) comes over the fabric.  The layer-1 job (K = 2L) goes first: with two
// workgroups per CU every CU gets one K = 2L and one K = L tile.  (Equal-size work items -- the
// K = 2L tile as two workgroups that meet through L2 -- were built and measured:
// tools/rejected/lstm_tile_equal_items.hip.txt; the hand-off costs what the balance gains.)
#include <type_traits>

#include "device_utils.h"
#include "kernels.h"

namespace n2nmn {

namespace {


constexpr int TILE_ROWS = 64, TILE_UNITS = 16, TILE_BK = 32;
constexpr int TILE_WAVES = 8, TILE_THREADS = TILE_WAVES * 64;
constexpr int STAGE_BYTES = 2 * (TILE_BK / 4) * 64 * 16;      // h image 8 KiB, then W image 8 KiB
constexpr int W_IMAGE = (TILE_BK / 4) * 64 * 16;

struct LstmJobs2 {
  LstmJob j[2];
};

// One LDS-DMA instruction: every lane copies 16 bytes from `base + voff` to LDS[lds + 16 * lane]
// (global_load_lds_dwordx4; M0 carries the LDS address).  M0 is compiler-reserved, so it is saved and
// restored inside the statement (cdna_hip_programming.md 5.7).  hipcc does not count this load: the
// waits are ours.
__device__ __forceinline__ void glds16(const float* base, uint32_t voff, uint32_t lds) {
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
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// DBG (n2nmn_debug_lstm_bench only): 1 = no DMA (LDS holds whatever it held), 2 = no MFMA (operands
// read from LDS and kept live), 3 = DMA + barriers only, 4 = no DMA waits in the k loop (wrong results),
// 5 = every stage re-reads the first stage's (cache-hot) addresses
template <int NS, int DBG = 0>
__global__ __launch_bounds__(TILE_THREADS) void lstm_tile_kernel(LstmJobs2 jobs, int N, int L,
                                                                 int nrb, int njobs) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int ntile = L / TILE_UNITS;
  // ---- id -> (job, row block, column tile) -------------------------------------------------------
  // Only row blocks with an active row do the step (length-sorted encoder: rows [nact, N) are past
  // their length; the training step keeps every block).  Their workgroups take the ids
  // [0, (nab1 + nab0) * ntile), the K = 2L job first, so the dispatcher spreads THEM over all CUs
  // (with the blocks numbered in place, a half-active step ran two workgroups on half of the CUs);
  // the ids behind write dynamic_rnn's zero output rows of the inactive blocks or leave.
  int nab[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const LstmJob& jq = jobs.j[j];
    int n = 0;
    if (j < njobs && jq.active) {
      const int na = (jq.n_active && !jq.save_gates) ? *jq.n_active : N;
      n = (min(max(na, 0), N) + TILE_ROWS - 1) / TILE_ROWS;
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
  const int wr = w & 3, kh = w >> 2;                // row group of the wave, and its half of every stage
  const int row0 = rb * TILE_ROWS;
  const int R = jb.hp_R;
  const int nact = (jb.n_active && !jb.save_gates) ? *jb.n_active : N;

  // ---- the lane's place in the epilogue: row lr of the wave, units 16 ct + 4 q .. + 3 ----------
  const int lr = lane & 15, q = lane >> 4;
  const int gr = row0 + 16 * wr + lr;
  const bool eact = gr < N;
  const int grc = eact ? gr : N - 1;
  const int t4 = 4 * ct + q;                         // 4-unit group = float4 element of the state
  if (zero_fill) {
    // every row of the block is past its length: only dynamic_rnn's zero output row is left of the
    // step (see lstm_step_kernel)
    if (kh == 0 && eact) {
      const int zr = jb.perm ? jb.perm[grc] : grc;
      *reinterpret_cast<float4*>(jb.out_seq + (size_t)zr * L + 4 * t4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }

  // ---- operand stream ----------------------------------------------------------------------------
  const int K = jb.K, nst = K / TILE_BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int arow = row0 + lane < N ? row0 + lane : N - 1;         // DMA lane = row of the block
  // byte offsets of this lane inside a k4 slab of h ([k4][R][4]) and of the weight tile ([k4][64][4])
  const uint32_t hoff = (uint32_t)arow * 16u;
  const uint32_t woff = (uint32_t)lane * 16u;
  const float* Wt = jb.Wp64 + (size_t)ct * (K / 4) * 256;
  const uint32_t hslab = (uint32_t)R * 16u;                        // bytes per k4 of h
  // (copies: `jb` lives in the kernel-argument segment, and every DMA statement clobbers "memory",
  // so a jb.A0 inside the k loop is re-fetched by an s_load per stage -- whose out-of-order return
  // makes the compiler wait lgkmcnt(0) around the LDS reads)
  const float* const A0p = jb.A0;
  const float* const A1p = jb.A1;
  // stage s -> ring slot s % NS; wave w moves k4 slab w of h (issue_h) and of the weights (issue_w)
  auto issue_h = [&](int s) {
    const int k0 = DBG == 5 ? 0 : s * TILE_BK;
    const bool lo = k0 < L;
    glds16(lo ? A0p : A1p, (uint32_t)(((lo ? k0 : k0 - L) >> 2) + w) * hslab + hoff,
           lds0 + (uint32_t)(s % NS) * STAGE_BYTES + (uint32_t)w * 1024u);
  };
  auto issue_w = [&](int s) {
    glds16(Wt, (uint32_t)(((DBG == 5 ? 0 : s) * TILE_BK >> 2) + w) * 1024u + woff,
           lds0 + (uint32_t)(s % NS) * STAGE_BYTES + W_IMAGE + (uint32_t)w * 1024u);
  };
  auto issue = [&](int s) { issue_h(s); issue_w(s); };
  // the first NS - 1 stages go out before anything else (K / 32 >= NS - 1: lstm_tile_supported)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (DBG != 1) issue(s);

  // ---- epilogue operands (waves of the first k half own the cell update): fetched now, under the
  // DMA prologue, used after the last MFMA.  (The compiler waits for its own loads with counts that
  // do not know about the DMAs: in the in-order queue that can only over-wait, here once at the
  // start, never in the k loop.) ---------------------------------------------------------------------
  int orow = grc;
  float4 add[4];
  float4 c_old = make_float4(0.f, 0.f, 0.f, 0.f), h_prev = c_old;
  bool masked = false;
  const size_t sidx = ((size_t)t4 * R + grc) * 4;
  if (kh == 0) {
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

  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool wact = row0 + 16 * wr < nact;          // wave-uniform: any active row in this wave?

  // A stage is two groups of 16 k; waves 0-3 take the first group of EVERY stage, waves 4-7 the
  // second (8 waves = two per SIMD from this workgroup alone, so one covers the other's LDS reads,
  // DMA issue and barrier; the two halves of z meet in LDS after the loop).  A wave keeps the
  // operands of its group of stage i in registers (P) while it runs that group's 16 MFMAs, and
  // fetches its group of stage i + 1 (Q) from LDS first.  One raw s_barrier per stage: before it a
  // wave waits (counted vmcnt -- a __syncthreads() would drain the whole queue) for its own DMA pieces
  // of stage i + 1; after it every wave's pieces of that stage are visible and every wave is done
  // with stage i - 1, whose slot is refilled.
  struct Grp { float4 h; float4 w[4]; };
  const float4* const hS0 = reinterpret_cast<const float4*>(smem) + (4 * kh + q) * 64 + 16 * wr + lr;
  const float4* const wS0 = reinterpret_cast<const float4*>(smem + W_IMAGE) + (4 * kh + q) * 64 + lr;
  auto mma1 = [&](const Grp& g, int c) {           // the four gates' MFMAs of k component c
    if (DBG == 2) {
      asm volatile("" ::"v"(g.w[0].x), "v"(g.w[1].y), "v"(g.w[2].z), "v"(g.w[3].w), "v"(g.h.x), "v"(g.h.w));
      return;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float a = c == 0 ? g.w[t].x : c == 1 ? g.w[t].y : c == 2 ? g.w[t].z : g.w[t].w;
      const float b = c == 0 ? g.h.x : c == 1 ? g.h.y : c == 2 ? g.h.z : g.h.w;
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
  };
  // stage `next` must be visible before its group is read: wait for this wave's pieces (at most the
  // stages issued after it may stay outstanding), then meet the other waves
  auto sync_stage = [&](int next) {
    const int behind = nst - 1 - next;             // stages after `next`
    if (DBG == 4) { if (behind == 0) wait_vm<0>(); }
    else if (behind >= NS - 3) wait_vm<2 * (NS - 3)>();
    else if (NS > 5 && behind == 2) wait_vm<4>();
    else if (NS > 4 && behind == 1) wait_vm<2>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
  };
  // One step of the pipeline: the 16 MFMAs of group `cur` (in registers) with everything else the
  // wave owes at this point woven into the first MFMAs' shadows (an MFMA holds the pipe for 32
  // cycles, the wave may issue ~5 other instructions meanwhile): the refill of the slot behind and
  // the LDS reads of the group of stage `next` into `nxt`.  The pins keep hipcc from sinking the
  // reads below the MFMAs (it would, to reuse cur's registers: no double buffering).
#define N2_PIN() __builtin_amdgcn_sched_barrier(0)
  auto step = [&](auto work_tag, const Grp& cur, Grp& nxt, int next, bool fetch) {
    constexpr bool WORK = decltype(work_tag)::value;
    const int refill = next + NS - 2;
    const bool dma = fetch && refill < nst && DBG != 1;
    const int slot = next % NS;
    if (fetch) sync_stage(next);
    if (WORK) { mma1(cur, 0); N2_PIN(); }
    if (dma) issue_h(refill);
    if (WORK) {
      N2_PIN();
      if (fetch) nxt.h = hS0[slot * (STAGE_BYTES / 16)];
      N2_PIN(); mma1(cur, 1); N2_PIN();
    }
    if (dma) issue_w(refill);
    if (WORK) {
      N2_PIN();
      if (fetch) {
#pragma unroll
        for (int t = 0; t < 4; ++t) nxt.w[t] = wS0[slot * (STAGE_BYTES / 16) + 16 * t];
      }
      N2_PIN(); mma1(cur, 2); mma1(cur, 3);
    }
  };
  // The k loop exists twice -- for waves that hold an active row and for waves that only feed the
  // DMA ring and the barriers -- instead of testing `wact` inside it: with conditional LDS reads in
  // the loop body the compiler's lgkmcnt bookkeeping merges the two paths and ends up waiting for
  // the NEXT group's reads before the MFMAs of the current one.
  auto stages = [&](auto work_tag) {
    constexpr bool WORK = decltype(work_tag)::value;
    wait_vm<2 * (NS - 2)>();                       // stage 0 (the oldest of the NS - 1 in flight)
    __builtin_amdgcn_s_barrier();
    Grp P{}, Q{};
    if (WORK) {
      P.h = hS0[0];
#pragma unroll
      for (int t = 0; t < 4; ++t) P.w[t] = wS0[16 * t];
    }
    int i = 0;
    for (; i + 2 < nst; i += 2) {                  // nst is even: two stages per trip, P / Q static
      step(work_tag, P, Q, i + 1, true);
      step(work_tag, Q, P, i + 2, true);
    }
    step(work_tag, P, Q, i + 1, true);             // the last two stages
    step(work_tag, Q, P, 0, false);
  };
#undef N2_PIN
  if (wact && DBG != 3) stages(std::true_type{});
  else stages(std::false_type{});

  // ---- the two k halves meet: waves 4-7 park their partial tile in LDS (every DMA has landed and
  // every group has been read: the ring is free), waves 0-3 add it to theirs -----------------------
  __builtin_amdgcn_s_barrier();
  float4* red = reinterpret_cast<float4*>(smem) + (size_t)wr * 4 * 64 + lane;
  if (kh == 1) {
#pragma unroll
    for (int t = 0; t < 4; ++t) red[t * 64] = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
  __syncthreads();
  if (kh == 1 || !eact) return;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float4 o = red[t * 64];
    acc[t][0] += o.x; acc[t][1] += o.y; acc[t][2] += o.z; acc[t][3] += o.w;
  }

  // ---- cell update: lane = (row, 4 units), acc[g][r] = z of gate g, unit 4q + r ------------------
  float cn[4], hn[4], gi[4], gj[4], gf[4], go[4];
  const float co[4] = {c_old.x, c_old.y, c_old.z, c_old.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float zi = acc[0][r] + (r == 0 ? add[0].x : r == 1 ? add[0].y : r == 2 ? add[0].z : add[0].w);
    const float zj = acc[1][r] + (r == 0 ? add[1].x : r == 1 ? add[1].y : r == 2 ? add[1].z : add[1].w);
    const float zf = acc[2][r] + (r == 0 ? add[2].x : r == 1 ? add[2].y : r == 2 ? add[2].z : add[2].w);
    const float zo = acc[3][r] + (r == 0 ? add[3].x : r == 1 ? add[3].y : r == 2 ? add[3].z : add[3].w);
    gi[r] = fast_sigmoid(zi); gj[r] = fast_tanh(zj); gf[r] = fast_sigmoid(zf + 1.0f); go[r] = fast_sigmoid(zo);
    cn[r] = co[r] * gf[r] + gi[r] * gj[r];
    hn[r] = fast_tanh(cn[r]) * go[r];
  }
  float4 c4 = make_float4(cn[0], cn[1], cn[2], cn[3]);
  float4 h4 = make_float4(hn[0], hn[1], hn[2], hn[3]);
  float4 o4 = h4;
  if (masked) { c4 = c_old; h4 = h_prev; o4 = make_float4(0.f, 0.f, 0.f, 0.f); }
  *reinterpret_cast<float4*>(jb.c_out + sidx) = c4;
  *reinterpret_cast<float4*>(jb.h_new + sidx) = h4;
  const size_t oidx = (size_t)orow * L + 4 * t4;
  if (jb.save_gates) {            // training: what the cell backward needs (ORIGINAL row order)
#pragma unroll
    for (int r = 0; r < 4; ++r) jb.save_gates[oidx + r] = make_float4(gi[r], gj[r], gf[r], go[r]);
    *reinterpret_cast<float4*>(jb.save_c + oidx) = c4;
    *reinterpret_cast<float4*>(jb.save_h + oidx) = h4;
  }
  if (jb.out_seq) *reinterpret_cast<float4*>(jb.out_seq + oidx) = o4;
  if (jb.h_drop) {                // dropped copy of the OUTPUT for the layer above (models_vqa)
    const float4 dm = *reinterpret_cast<const float4*>(jb.drop + oidx);
    const float4 hd = make_float4(h4.x * dm.x, h4.y * dm.y, h4.z * dm.z, h4.w * dm.w);
    *reinterpret_cast<float4*>(jb.h_drop + sidx) = hd;
    if (jb.save_hd) *reinterpret_cast<float4*>(jb.save_hd + oidx) = hd;
  }
  if (jb.fin_c && jb.seq_len && jb.t == jb.seq_len[orow] - 1) {   // the row's last valid step
    const size_t fidx = ((size_t)t4 * R + orow) * 4;
    *reinterpret_cast<float4*>(jb.fin_c + fidx) = c4;
    *reinterpret_cast<float4*>(jb.fin_h + fidx) = h4;
  }
}

template <int NS, int DBG = 0>
void launch_tile(const LstmJobs2& js, int njobs, int N, int L, hipStream_t s) {
  static std::atomic<uint64_t> attr{0};
  const int lds = NS * STAGE_BYTES;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&lstm_tile_kernel<NS, DBG>), lds, attr);
  const int nrb = (N + TILE_ROWS - 1) / TILE_ROWS;
  const int grid = njobs * nrb * (L / TILE_UNITS);
  hipLaunchKernelGGL((lstm_tile_kernel<NS, DBG>), dim3(grid), dim3(TILE_THREADS), lds, s, js, N, L,
                     nrb, njobs);
}

}  // namespace

bool lstm_tile_supported(const LstmJob* jobs, int njobs, int L) {
  if (njobs < 1 || njobs > 2 || L % (8 * TILE_UNITS) != 0) return false;
  for (int i = 0; i < njobs; ++i) {
    const LstmJob& j = jobs[i];
    if (j.mode != 0 || !j.Wp64 || j.hp_R <= 0 || j.a_rs != 4 || j.K % TILE_BK != 0 ||
        j.K % (2 * TILE_BK) != 0 || j.K / TILE_BK < 8 || j.ntiles != L / 4 || (j.K != L && j.K != 2 * L))
      return false;
  }
  return true;
}

void launch_lstm_tile(const LstmJob* jobs, int njobs, int N, int L, int stages, hipStream_t s) {
  LstmJobs2 js;
  for (int i = 0; i < 2; ++i) {
    if (i < njobs) js.j[i] = jobs[i];
    else { js.j[i] = LstmJob{}; js.j[i].active = 0; }
  }
  switch (stages) {
    case 3: launch_tile<3>(js, njobs, N, L, s); break;
    case 5: launch_tile<5>(js, njobs, N, L, s); break;
    case 6: launch_tile<6>(js, njobs, N, L, s); break;
    case 14: launch_tile<4, 1>(js, njobs, N, L, s); break;     // debug variants of the 4-stage kernel
    case 24: launch_tile<4, 2>(js, njobs, N, L, s); break;
    case 34: launch_tile<4, 3>(js, njobs, N, L, s); break;
    case 44: launch_tile<4, 4>(js, njobs, N, L, s); break;
    case 54: launch_tile<4, 5>(js, njobs, N, L, s); break;
    default: launch_tile<4>(js, njobs, N, L, s); break;
  }
}

}  // namespace n2nmn
