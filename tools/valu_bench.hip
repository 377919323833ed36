// Micro-benchmark: VALU issue cost per wave-instruction on MI355X for the instruction kinds the sweep's inner loop is
// made of (plain / packed fp32 FMA, 32- and 64-bit compares + selects, LDS broadcast reads, LDS writes), at 1 / 2 / 4
// waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/valu_bench.hip -o /tmp/valu_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int kU = 8;       // independent chains
constexpr int kIter = 4096;

template <int KIND>
__global__ void __launch_bounds__(64) k_valu(float *out, const float seed, const int iters) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x;
  float a[kU], b[kU];
  v2f p[kU];
  unsigned long long q[kU];
  for (int u = 0; u < kU; ++u) { a[u] = seed + u + lane; b[u] = seed * 0.5f + u; p[u] = (v2f){a[u], b[u]}; q[u] = (unsigned long long)(lane * 77 + u) << 20; }
  for (int i = lane; i < 4096; i += 64) lds[i] = (float)i;
  __syncthreads();
  const float c0 = seed * 1.0001f, c1 = seed * 0.9999f;
  const v2f pc0 = (v2f){c0, c1}, pc1 = (v2f){c1, c0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int u = 0; u < kU; ++u) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(c0), "v"(c1));      // v_fma_f32 (asm: the compiler SLP-packs plain fmaf)
        if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(pc0), "v"(pc1));  // v_pk_fma_f32
        if (KIND == 2) {                                                           // 64-bit compare + 2 selects
          const unsigned long long k = q[(u + 1) % kU] + (unsigned)it;
          q[u] = (k < q[u]) ? k : q[u] + 3;
        }
        if (KIND == 3) {                                                           // f32 compare + select
          const float k = a[(u + 1) % kU] + c0;
          a[u] = (k < a[u]) ? k : a[u] + c1;
        }
        if (KIND == 4) a[u] += lds[(it * 4 + r * kU + u) & 4095];                  // LDS broadcast read b32 + add
        if (KIND == 5) {                                                           // LDS broadcast read b64 + 2 adds
          const v2f v = *reinterpret_cast<const v2f *>(&lds[((it * 4 + r * kU + u) * 2) & 4094]);
          a[u] += v.x; b[u] += v.y;
        }
        if (KIND == 6) { lds[((it + u) & 63) * 64 + lane] = a[u]; a[u] += c0; }    // LDS write b32 per lane + add
        if (KIND == 7) {                                                           // LDS write b64 per lane + add
          *reinterpret_cast<v2f *>(&lds[(((it + u) & 31) * 64 + lane) * 2]) = (v2f){a[u], b[u]};
          a[u] += c0;
        }
        if (KIND == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[u]) : "v"(c0));                  // v_mul_f32
        if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[u]) : "v"(pc0));              // v_pk_mul_f32
        if (KIND == 10) asm volatile("v_exp_f32 %0, %0" : "+v"(a[u]));                                // v_exp_f32 (transcendental)
        if (KIND == 11) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[u]));                                // v_rcp_f32
        if (KIND == 12) { asm volatile("v_exp_f32 %0, %0" : "+v"(a[u])); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[u]) : "v"(pc0), "v"(pc1));
                          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(u + 1) % kU]) : "v"(pc0), "v"(pc1));
                          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(u + 2) % kU]) : "v"(pc0), "v"(pc1)); }      // 1 exp + 3 pk_fma: do they overlap?
      }
    }
  }
  float s = 0.f;
  for (int u = 0; u < kU; ++u) s += a[u] + b[u] + p[u].x + p[u].y + (float)q[u];
  if (s == 12345.678f) out[blockIdx.x * 64 + lane] = s + lds[lane];
}

template <int KIND>
static void run(const char *name, float *out, int waves_per_simd, double instr_per_iter) {
  const int blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f, 64);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_valu<KIND>, dim3(blocks), dim3(64), 0, 0, out, 1.0f, kIter);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)kIter * instr_per_iter;                // wave-instructions per wave
  printf("%-34s waves/SIMD %d: %8.3f ms  -> %6.2f ns per wave-instr per wave, %6.2f ns per SIMD\n", name, waves_per_simd, ms,
         ms * 1e6 / n, ms * 1e6 / (n * waves_per_simd));
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 4 * 8 * 64 * sizeof(float));
  for (int w : {1, 2, 4}) {
    run<0>("v_fma_f32", out, w, 4.0 * kU);
    run<1>("v_pk_fma_f32", out, w, 4.0 * kU);
    run<8>("v_mul_f32", out, w, 4.0 * kU);
    run<9>("v_pk_mul_f32", out, w, 4.0 * kU);
    run<2>("add_u64 + cmp_u64 + 2 sel (+add)", out, w, 4.0 * kU);
    run<3>("add + cmp_f32 + sel (+add)", out, w, 4.0 * kU);
    run<4>("ds_read_b32 bcast + add", out, w, 4.0 * kU);
    run<5>("ds_read_b64 bcast + 2 add", out, w, 4.0 * kU);
    run<6>("ds_write_b32 + add", out, w, 4.0 * kU);
    run<7>("ds_write_b64 + add", out, w, 4.0 * kU);
    run<10>("v_exp_f32", out, w, 4.0 * kU);
    run<11>("v_rcp_f32", out, w, 4.0 * kU);
    run<12>("v_exp_f32 + 3 v_pk_fma_f32 (per group)", out, w, 4.0 * kU);
  }
  return 0;
}
