// What would a PERSISTENT recurrent step cost on MI355X?  (measure, do not cite.)
// hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
//
// A persistent encoder keeps its 49 KB weight slice per CU in LDS and replaces the kernel boundary of
// every time step by a grid barrier plus the exchange of the new hidden state (each of the 256
// workgroups produces 1 KB of h per step -- 4 hidden units x 64 rows x 2 layers -- and every workgroup
// reads all of it, 256 KB, before its next MFMA phase).  This microbenchmark times exactly that
// skeleton: 256 workgroups x 512 threads (the step kernel's geometry), XCD-hierarchical barrier
// (per-group counter -> leader -> top counter -> per-group generation word; release fence before the
// arrive, acquire fence after the wait), with
//   variant 0: barrier only
//   variant 1: + every workgroup publishes its 1 KB slice before the barrier
//   variant 2: + every workgroup reads the whole 256 KB state after the barrier (float4, all loads of
//                a thread in flight)
// against the launch-per-step baseline: the same write/read body as its own kernel, launched back to
// back (variant 3).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NWG = 256, NT = 512, NGROUP = 8;
constexpr int STATE_FLOATS = 64 * 1024;          // 256 KB of fp32 hidden state (two layers)
constexpr int SLICE = STATE_FLOATS / NWG;         // 256 floats = 1 KB per workgroup

struct Bar {
  unsigned group_count[NGROUP * 16];   // one 64-B line per group
  unsigned group_gen[NGROUP * 16];
  unsigned top_count[16];
};

__device__ __forceinline__ void grid_barrier(Bar* b, unsigned gen /* barriers passed so far */) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int g = blockIdx.x % NGROUP;            // observed placement: block b runs on XCD b % 8
    const unsigned members = NWG / NGROUP;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(&b->group_count[g * 16], 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    if (old == (gen + 1) * members - 1) {         // last of its group: the group's leader
      __hip_atomic_fetch_add(&b->top_count[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&b->top_count[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
             (gen + 1) * NGROUP)
        __builtin_amdgcn_s_sleep(1);
      __hip_atomic_store(&b->group_gen[g * 16], gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&b->group_gen[g * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <
             gen + 1)
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

__device__ __forceinline__ float body_read(const float* state) {
  // 256 KB / 512 threads = 32 float4 per thread, all in flight
  const float4* s4 = reinterpret_cast<const float4*>(state);
  float4 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = s4[threadIdx.x + i * NT];
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) acc += v[i].x + v[i].y + v[i].z + v[i].w;
  return acc;
}

template <int VARIANT>
__global__ __launch_bounds__(NT) void persistent(Bar* b, float* state0, float* state1, float* sink,
                                                 int iters) {
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    float* wr = (it & 1) ? state1 : state0;
    if (VARIANT >= 1 && threadIdx.x < SLICE) wr[blockIdx.x * SLICE + threadIdx.x] = acc + it;
    grid_barrier(b, (unsigned)it);
    if (VARIANT >= 2) acc += body_read(wr);
  }
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

__global__ __launch_bounds__(NT) void step_kernel(const float* rd, float* wr, float* sink, int it) {
  const float acc = body_read(rd);
  if (threadIdx.x < SLICE) wr[blockIdx.x * SLICE + threadIdx.x] = acc + it;
  if (threadIdx.x == 0) sink[blockIdx.x] = acc;
}

__global__ void empty_kernel() {}

int main() {
  Bar* bar;
  float *s0, *s1, *sink;
  CK(hipMalloc(&bar, sizeof(Bar)));
  CK(hipMalloc(&s0, STATE_FLOATS * 4)); CK(hipMalloc(&s1, STATE_FLOATS * 4)); CK(hipMalloc(&sink, NWG * 4));
  CK(hipMemset(s0, 0, STATE_FLOATS * 4)); CK(hipMemset(s1, 0, STATE_FLOATS * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 400;
  auto run = [&](int variant) -> float {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipMemset(bar, 0, sizeof(Bar));
      (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0, 0);
      if (variant == 0) hipLaunchKernelGGL(persistent<0>, dim3(NWG), dim3(NT), 0, 0, bar, s0, s1, sink, iters);
      if (variant == 1) hipLaunchKernelGGL(persistent<1>, dim3(NWG), dim3(NT), 0, 0, bar, s0, s1, sink, iters);
      if (variant == 2) hipLaunchKernelGGL(persistent<2>, dim3(NWG), dim3(NT), 0, 0, bar, s0, s1, sink, iters);
      if (variant == 3)
        for (int it = 0; it < iters; ++it)
          hipLaunchKernelGGL(step_kernel, dim3(NWG), dim3(NT), 0, 0, (it & 1) ? s1 : s0, (it & 1) ? s0 : s1, sink, it);
      if (variant == 4)
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(empty_kernel, dim3(NWG), dim3(NT), 0, 0);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    return best * 1e3f / iters;
  };
  const char* names[] = {"persistent: grid barrier only",
                         "persistent: publish 1 KB per workgroup + barrier",
                         "persistent: publish + barrier + read 256 KB state per workgroup",
                         "launch per step: read 256 KB + publish 1 KB (one kernel per step)",
                         "launch per step: empty kernel"};
  for (int v = 0; v < 5; ++v) printf("%-70s %7.2f us per step\n", names[v], run(v));
  if (hipGetLastError() != hipSuccess) { printf("HIP error\n"); return 1; }
  return 0;
}
