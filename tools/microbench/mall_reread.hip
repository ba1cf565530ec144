// Does a re-read of a buffer that was just read (or written) come back faster than a cold read from HBM?
// (MI355X: 256 MB memory-side Infinity Cache behind eight 4 MB L2s.)  A streaming float4 read of `bytes` with a
// grid of 2048 x 256 threads, timed with events: cold (after sweeping a 1 GB buffer), re-read at once, re-read after
// another 64 MB went through, and read-after-write.
//   hipcc --offload-arch=gfx950 -O3 -o tools/microbench/mall_reread tools/microbench/mall_reread.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void read_sum(const float4* __restrict__ p, size_t n4, float* out) {
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = p[i];
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  if (a.x + a.y + a.z + a.w == 12345.678f) out[0] = a.x;       // (never true: keeps the loads)
}

__global__ __launch_bounds__(256) void write_val(float4* __restrict__ p, size_t n4, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    p[i] = make_float4(v, v, v, v);
}

static float timed_read(const float4* p, size_t n4, float* out, hipStream_t s) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  CK(hipEventRecord(a, s));
  hipLaunchKernelGGL(read_sum, dim3(2048), dim3(256), 0, s, p, n4, out);
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
  return ms;
}

int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  const size_t flush_bytes = (size_t)1 << 30;
  float4 *flush = nullptr, *other = nullptr;
  float* out = nullptr;
  CK(hipMalloc(&flush, flush_bytes)); CK(hipMalloc(&other, (size_t)64 << 20)); CK(hipMalloc(&out, 64));
  CK(hipMemset(flush, 0, flush_bytes)); CK(hipMemset(other, 0, (size_t)64 << 20));
  for (size_t mb : {24, 48, 96, 160, 320}) {
    const size_t bytes = mb << 20, n4 = bytes / 16;
    float4* buf = nullptr;
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 0, bytes));
    double cold = 0, again = 0, later = 0, raw = 0;
    const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      timed_read(flush, flush_bytes / 16, out, s);                 // sweep the caches
      cold += timed_read(buf, n4, out, s);
      again += timed_read(buf, n4, out, s);
      timed_read(other, ((size_t)64 << 20) / 16, out, s);
      later += timed_read(buf, n4, out, s);
      timed_read(flush, flush_bytes / 16, out, s);
      hipLaunchKernelGGL(write_val, dim3(2048), dim3(256), 0, s, buf, n4, (float)r);
      raw += timed_read(buf, n4, out, s);                          // read right after it was written
    }
    auto gbs = [&](double ms) { return bytes / (ms / reps * 1e-3) / 1e9; };
    printf("%4zu MB: cold %.0f GB/s (%.1f us) | re-read at once %.0f GB/s (%.1f us) | after 64 MB of other traffic %.0f GB/s | "
           "read after write %.0f GB/s\n", mb, gbs(cold), cold / reps * 1e3, gbs(again), again / reps * 1e3, gbs(later), gbs(raw));
    CK(hipFree(buf));
  }
  return 0;
}
