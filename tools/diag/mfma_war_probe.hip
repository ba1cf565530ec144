// Hardware probe for profiles/r06_notes.md section 1: can an LDS read that is issued right BEHIND a chain of
// dependent MFMAs overwrite a source operand (SrcB) of the last MFMA before that MFMA has read it?
//
// gemm_dma3_kernel's compiled K loop does exactly this (kernels_gemm_dma3.hip, ISA of hipcc 7.2):
//     v_mfma_f32_16x16x32_bf16 v[14:17], v[18:21], v[46:49], v[14:17]   ; 5th of 6 dependent MFMAs, SrcB = v[46:49]
//     ds_read_b128 v[46:49], ...                                        ; next column tile's fragment, same VGPRs
// and returned wrong tiles only while ANOTHER stream's kernels shared the CUs.  LLVM inserts wait states for
// the MFMA SrcC write-after-read hazard only; SrcA / SrcB are taken to be read at issue.
//
// Victim wave: acc = 0; N dependent v_mfma_f32_16x16x32_bf16 with A = 1, B = 1 (each adds 32 to every element);
// then ds_read_b128 of a fragment of 2s INTO B's registers; wait; acc must be 32 N.  If the last MFMA saw the
// new fragment acc is 32 N + 32.  Aggressor waves (odd workgroups) keep the SIMD's matrix pipe busy with 16-pass
// v_mfma_f32_32x32x2_f32, so a queued victim MFMA starts late.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/diag/mfma_war_probe tools/diag/mfma_war_probe.hip
//   tools/diag/mfma_war_probe [iterations] [chain length] [aggressor workgroups per victim]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                             \
  do {                                                                                       \
    hipError_t e_ = (x);                                                                     \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); }        \
  } while (0)

template <int CHAIN, int MODE>
__global__ __launch_bounds__(64) void probe(int iters, int every, unsigned* bad, unsigned* hist, float* sink) {
  __shared__ __attribute__((aligned(16))) uint32_t frag[64 * 4];
  const int lane = threadIdx.x;
  if (every > 0 && (blockIdx.x % (every + 1)) != 0) {
    // ---- aggressor: independent 16-pass fp32 MFMAs, back to back -----------------------------------
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    const float x = 1.0f + lane, y = 0.5f;
    for (int i = 0; i < iters * 4; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
    }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = a0[0];
    return;
  }
  // ---- victim ---------------------------------------------------------------------------------------
  for (int j = 0; j < 4; ++j) frag[lane * 4 + j] = 0x40004000u;          // two bf16 2.0
  __syncthreads();
  const uint32_t lds = (uint32_t)(uintptr_t)frag + lane * 16;
  const u32x4 ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};   // eight bf16 1.0
  unsigned nbad = 0, nread = 0, h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < iters; ++i) {
    u32x4 A = ones, B = ones;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // (one asm block: the order below is the order on the chip)
    if (MODE == 0) {
      asm volatile(
          "s_nop 15\n\t"         // (the operand set-up in front is VALU: hipcc cannot see the MFMA in here)
          ".rept %4\n\t"
          "v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0\n\t"
          ".endr\n\t"
          "ds_read_b128 %1, %3\n\t"
          "s_waitcnt lgkmcnt(0)\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
          : "+v"(acc), "+v"(B)
          : "v"(A), "v"(lds), "n"(CHAIN)
          : "memory");
    } else {
      // control: the read is issued only after the chain's result has been consumed
      asm volatile(
          "s_nop 15\n\t"         // (the operand set-up in front is VALU: hipcc cannot see the MFMA in here)
          ".rept %4\n\t"
          "v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0\n\t"
          ".endr\n\t"
          "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
          "ds_read_b128 %1, %3\n\t"
          "s_waitcnt lgkmcnt(0)"
          : "+v"(acc), "+v"(B)
          : "v"(A), "v"(lds), "n"(CHAIN)
          : "memory");
    }
    const float want = 32.0f * CHAIN;
    bool wrong = false;
    for (int r = 0; r < 4; ++r) wrong |= acc[r] != want;
    nread += B[0] == 0x40004000u;                          // (the read itself happened)
    if (wrong) {
      ++nbad;
      const int k = (int)((acc[0] - want) / 32.0f);        // how many MFMAs saw the new fragment
      ++h[k < 0 ? 7 : (k > 6 ? 6 : k)];
    }
  }
  // (one set of atomics per wave, at the end: lane 0's view)
  if (lane == 0) atomicAdd(&hist[8], nread);
  if (lane == 0 && nbad) {
    atomicAdd(bad, nbad);
    for (int k = 0; k < 8; ++k)
      if (h[k]) atomicAdd(&hist[k], h[k]);
  }
}


// ---- second question: how many wait states does the chip need between a VALU write of an MFMA source register and
// the MFMA?  (hipcc pads what it knows of; kernels_gemm_dma3.hip produces its A fragments with v_cvt_pk_bf16_f32
// right in front of the MFMAs.)  One asm block on fixed registers: A dword 0 holds two bf16 2.0, is rewritten to
// two 1.0 by v_cvt_pk_bf16_f32, N x s_nop 0 (and, VIA = 1, one independent MFMA) later the MFMA reads it: 32 if it
// saw the new value, 34 if the old one.
template <int N, int VIA, int W = 0>
__global__ __launch_bounds__(64) void hazard(int iters, int every, unsigned* bad, float* sink) {
  if (every > 0 && (blockIdx.x % (every + 1)) != 0) {
    f32x16 a0 = {0}, a1 = {0};
    const float x = 1.0f + threadIdx.x, y = 0.5f;
    for (int i = 0; i < iters * 2; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    }
    if (a0[0] + a1[1] == 12345.f) sink[0] = a0[0];
    return;
  }
  unsigned nbad = 0;
  const uint32_t ones = 0x3f803f80u;
  const float f1 = 1.0f;
  for (int i = 0; i < iters; ++i) {
    float r;
    asm volatile(
        "v_mov_b32 v100, 0x40004000\n\tv_mov_b32 v101, %1\n\tv_mov_b32 v102, %1\n\tv_mov_b32 v103, %1\n\t"
        "v_mov_b32 v104, %1\n\tv_mov_b32 v105, %1\n\tv_mov_b32 v106, %1\n\tv_mov_b32 v107, %1\n\t"
        "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t"
        "v_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\t"
        "s_nop 15\n\ts_nop 15\n\t"
        "v_cvt_pk_bf16_f32 v100, %2, %2\n\t"
        ".rept %5\n\ts_waitcnt lgkmcnt(0)\n\t.endr\n\t"     // (nothing outstanding: the wait is satisfied at once)
        ".rept %3\n\ts_nop 0\n\t.endr\n\t"
        ".rept %4\n\tv_mfma_f32_16x16x32_bf16 v[112:115], v[104:107], v[104:107], v[112:115]\n\t.endr\n\t"
        "v_mfma_f32_16x16x32_bf16 v[108:111], v[100:103], v[104:107], v[108:111]\n\t"
        "s_nop 15\n\ts_nop 15\n\ts_nop 15\n\t"
        "v_mov_b32 %0, v108"
        : "=v"(r)
        : "v"(ones), "v"(f1), "n"(N), "n"(VIA), "n"(W)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113", "v114", "v115", "memory");
    nbad += r != 32.0f;
  }
  if (threadIdx.x == 0 && nbad) atomicAdd(bad, nbad);
}

template <int N, int VIA, int W = 0>
static void run_hazard(int iters, int every, int grid) {
  unsigned* bad;
  float* sink;
  CHECK(hipMalloc(&bad, 4));
  CHECK(hipMalloc(&sink, 16));
  CHECK(hipMemset(bad, 0, 4));
  hipLaunchKernelGGL((hazard<N, VIA, W>), dim3(grid), dim3(64), 0, 0, iters, every, bad, sink);
  CHECK(hipDeviceSynchronize());
  unsigned hb = 0;
  CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
  const long victims = every > 0 ? (grid + every) / (every + 1) : grid;
  printf("v_cvt_pk_bf16_f32 -> %d x satisfied s_waitcnt + %d x s_nop 0%s -> MFMA reading it, %d aggressor workgroup(s) per victim: %u of %ld used the OLD value\n",
         W, N, VIA ? " + one independent MFMA" : "", every, hb, victims * iters);
  CHECK(hipFree(bad)); CHECK(hipFree(sink));
}

template <int CHAIN>
static void run(int iters, int every, int grid) {
  unsigned *bad, *hist;
  float* sink;
  CHECK(hipMalloc(&bad, 4));
  CHECK(hipMalloc(&hist, 64));
  CHECK(hipMalloc(&sink, 16));
  for (int mode = 0; mode < 2; ++mode) {
    CHECK(hipMemset(bad, 0, 4));
    CHECK(hipMemset(hist, 0, 64));
    if (mode == 0) hipLaunchKernelGGL((probe<CHAIN, 0>), dim3(grid), dim3(64), 0, 0, iters, every, bad, hist, sink);
    else hipLaunchKernelGGL((probe<CHAIN, 1>), dim3(grid), dim3(64), 0, 0, iters, every, bad, hist, sink);
    CHECK(hipDeviceSynchronize());
    unsigned hb = 0, hh[16];
    CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hh, hist, 64, hipMemcpyDeviceToHost));
    const long victims = every > 0 ? (grid + every) / (every + 1) : grid;
    printf("chain %d, %d aggressor workgroup(s) per victim, %s: %u wrong of %ld chains; extra products per wrong chain:",
           CHAIN, every, mode == 0 ? "read issued right behind the chain" : "control (read behind the result)", hb,
           victims * iters);
    printf(" (reads seen %u)", hh[8]);
    for (int k = 0; k < 8; ++k)
      if (hh[k]) printf(" [%d]=%u", k, hh[k]);
    printf("\n");
  }
  CHECK(hipFree(bad)); CHECK(hipFree(hist)); CHECK(hipFree(sink));
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  const int grid = argc > 2 ? atoi(argv[2]) : 256 * 16;
  for (int every : {0, 1, 3, 7}) {
    run<1>(iters, every, grid);
    run<2>(iters, every, grid);
    run<6>(iters, every, grid);
    run<12>(iters, every, grid);
  }
  for (int every : {0, 3}) {
    run_hazard<0, 0>(iters, every, grid);
    run_hazard<1, 0>(iters, every, grid);
    run_hazard<2, 0>(iters, every, grid);
    run_hazard<3, 0>(iters, every, grid);
    run_hazard<4, 0>(iters, every, grid);
    run_hazard<0, 1>(iters, every, grid);
    run_hazard<1, 1>(iters, every, grid);
    run_hazard<0, 0, 1>(iters, every, grid);
    run_hazard<1, 0, 1>(iters, every, grid);
    run_hazard<0, 0, 2>(iters, every, grid);
    run_hazard<0, 0, 3>(iters, every, grid);
    run_hazard<2, 0, 1>(iters, every, grid);
  }
  return 0;
}
