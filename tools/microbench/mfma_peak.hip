// Calibration: what fp32 MFMA rate does this MI355X sustain with no memory traffic at all?
// hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x16 acc[NACC];
  f32x4 acc4[NACC];
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int r = 0; r < 4; ++r) acc4[i][r] = 0.f;
  }
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        else acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) {
    for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int r = 0; r < 4; ++r) s += acc4[i][r];
  }
  if (s == 123.456f) out[0] = s;
}

template <int NACC, int KIND>
void run(int wgs_per_cu, float* d) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(grid), dim3(256), 0, 0, d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, KIND>), dim3(grid), dim3(256), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (KIND == 0 ? 4096.0 : 2048.0) * 8 * NACC * (double)iters * 4 * grid;
  printf("%s  acc/wave %d  wgs/cu %d (waves/simd %d): %7.1f TFLOP/s (%.1f %% of 157.3)\n",
         KIND == 0 ? "32x32x2" : "16x16x4", NACC, wgs_per_cu, wgs_per_cu, fl / ms * 1e-9,
         fl / ms * 1e-9 / 157.3 * 100);
}

int main() {
  float* d; hipMalloc(&d, 4);
  for (int w : {1, 2, 4}) { run<1, 0>(w, d); run<2, 0>(w, d); run<4, 0>(w, d); }
  for (int w : {1, 2, 4}) { run<1, 1>(w, d); run<4, 1>(w, d); }
  return 0;
}
