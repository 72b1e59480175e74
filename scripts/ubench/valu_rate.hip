// Micro-benchmark: issue rate of fp32 VALU ops on gfx950 (scalar v_fma_f32 / v_add_f32 vs packed v_pk_fma_f32),
// at 1..8 waves per SIMD, with 8 independent accumulators per lane. Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float a, float b) {
  float x[8];
  float2v p[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; p[i] = (float2v){x[i], x[i] + 1.0f}; }
  const float2v av = {a, a}, bv = {b, b};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) x[i] = fmaf(x[i], a, b);
        if (MODE == 1) x[i] = x[i] + a;
        if (MODE == 2) p[i] = __builtin_elementwise_fma(p[i], av, bv);
        if (MODE == 3) p[i] = p[i] + av;
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd) {
  float* d;
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD
  hipMalloc(&d, sizeof(float) * blocks * 256);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 64 * waves_per_simd;  // wave-instructions issued on one SIMD
  const double ns_per_inst = ms * 1e6 / insts_per_simd;
  printf("%-14s waves/SIMD %d: %.3f ms, %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz, %.2f @2.1GHz)\n", name, waves_per_simd, ms,
         ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.1);
  hipFree(d);
}
int main() {
  for (int w : {1, 2, 4, 8}) {
    run<0>("v_fma_f32", w);
    run<1>("v_add_f32", w);
    run<2>("v_pk_fma_f32", w);
    run<3>("v_pk_add_f32", w);
  }
  return 0;
}
