// Micro-benchmark: float atomic-add throughput on MI355X by memory scope and access pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o /tmp/atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SCOPE>
__global__ void k_atomic(float *buf, const unsigned *idx, int n_per_thread, unsigned mask) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = t * 2654435761u;
  for (int i = 0; i < n_per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    float *p = buf + ((h >> 8) & mask);
    if (SCOPE == 0) unsafeAtomicAdd(p, 1.0f);
    if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (SCOPE == 3) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    if (SCOPE == 4) *p += 1.0f;  // plain RMW (racy) as a bandwidth reference
  }
}

// 12 consecutive floats per "entry" (the trace-bwd flush pattern)
template <int SCOPE>
__global__ void k_atomic12(float *buf, int n_per_thread, unsigned mask) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned h = t * 2654435761u;
  for (int i = 0; i < n_per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    float *p = buf + (size_t)((h >> 8) & mask) * 12;
    for (int j = 0; j < 12; ++j) {
      if (SCOPE == 0) unsafeAtomicAdd(p + j, 1.0f);
      if (SCOPE == 2) __hip_atomic_fetch_add(p + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}

// lane-coalesced: RUN adjacent lanes add to RUN adjacent floats of one random entry
template <int RUN>
__global__ void k_atomic_co(float *buf, int n_per_thread, unsigned mask) {
  unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned e = t / RUN, c = t % RUN;
  unsigned h = e * 2654435761u;
  for (int i = 0; i < n_per_thread; ++i) {
    h = h * 1664525u + 1013904223u;
    unsafeAtomicAdd(buf + (size_t)((h >> 8) & mask) * RUN + c, 1.0f);
  }
}

int main() {
  const unsigned mask = (1u << 20) - 1;  // 1M floats = 4 MB
  float *buf;
  hipMalloc(&buf, sizeof(float) * (size_t)(mask + 1) * 12);
  hipMemset(buf, 0, sizeof(float) * (size_t)(mask + 1) * 12);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int blocks = 2048, threads = 256, npt = 64;
  auto run = [&](const char *name, auto launch, double ops) {
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms/launch  %8.2f Gatomic/s\n", name, ms / 5, ops / (ms / 5 * 1e-3) / 1e9);
  };
  const double ops = (double)blocks * threads * npt;
  run("unsafeAtomicAdd", [&] { hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(threads), 0, 0, buf, nullptr, npt, mask); }, ops);
  run("agent relaxed", [&] { hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, buf, nullptr, npt, mask); }, ops);
  run("workgroup relaxed", [&] { hipLaunchKernelGGL(k_atomic<2>, dim3(blocks), dim3(threads), 0, 0, buf, nullptr, npt, mask); }, ops);
  run("wavefront relaxed", [&] { hipLaunchKernelGGL(k_atomic<3>, dim3(blocks), dim3(threads), 0, 0, buf, nullptr, npt, mask); }, ops);
  run("plain RMW (racy)", [&] { hipLaunchKernelGGL(k_atomic<4>, dim3(blocks), dim3(threads), 0, 0, buf, nullptr, npt, mask); }, ops);
  const unsigned mask12 = (1u << 16) - 1;  // 64k entries x 12 floats = 3 MB
  run("12-run unsafeAtomicAdd", [&] { hipLaunchKernelGGL(k_atomic12<0>, dim3(blocks), dim3(threads), 0, 0, buf, 8, mask12); }, (double)blocks * threads * 8 * 12);
  run("12-run workgroup", [&] { hipLaunchKernelGGL(k_atomic12<2>, dim3(blocks), dim3(threads), 0, 0, buf, 8, mask12); }, (double)blocks * threads * 8 * 12);
  run("coalesced 12-run", [&] { hipLaunchKernelGGL(k_atomic_co<12>, dim3(blocks), dim3(threads), 0, 0, buf, npt, mask12); }, ops);
  run("coalesced 16-run", [&] { hipLaunchKernelGGL(k_atomic_co<16>, dim3(blocks), dim3(threads), 0, 0, buf, npt, mask12); }, ops);
  run("coalesced 4-run", [&] { hipLaunchKernelGGL(k_atomic_co<4>, dim3(blocks), dim3(threads), 0, 0, buf, npt, mask12); }, ops);
  run("coalesced 64-run", [&] { hipLaunchKernelGGL(k_atomic_co<64>, dim3(blocks), dim3(threads), 0, 0, buf, npt, mask12 >> 3); }, ops);
  return 0;
}
