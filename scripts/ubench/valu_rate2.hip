// Micro-benchmark 2: fp32 VALU issue rate with 2 and 3 VGPR source operands, literal constants, and LDS b32/b64
// read/write rates in the FFT kernel's access pattern, at 8 waves per SIMD (4 blocks of 512 threads per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, float b) {
  __shared__ float lds[8704];
  float x[8], y[8], z[8];
  const int t = threadIdx.x;
  for (int i = 0; i < 8; ++i) { x[i] = t * 0.001f + i; y[i] = x[i] * 0.5f + 1.0f; z[i] = y[i] - 3.0f; }
  for (int i = t; i < 8704; i += 512) lds[i] = i;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) x[i] = x[i] + y[i];                 // 2 VGPR sources
        if (MODE == 1) x[i] = fmaf(x[i], y[i], z[i]);      // 3 VGPR sources
        if (MODE == 2) x[i] = x[i] * 0.98078528f;          // literal constant
        if (MODE == 3) x[i] = fmaf(x[i], 0.98078528f, y[i]);
        if (MODE == 4) { lds[17 * t + i + 8 * (u & 1)] = x[i]; }                  // ds_write_b32, 17-word pitch
        if (MODE == 5) { x[i] += lds[((t + 512 * (i + 8 * (u & 1))) * 17) >> 4]; } // ds_read_b32 consecutive-ish
        if (MODE == 6) { reinterpret_cast<float2*>(lds)[(33 * (t & 255) + i + 8 * (u & 1)) & 4095] = make_float2(x[i], y[i]); }  // ds_write_b64
        if (MODE == 7) { float2 v = reinterpret_cast<float2*>(lds)[(t + 512 * i) & 4095]; x[i] += v.x; y[i] += v.y; }          // ds_read_b64
      }
    }
    if (MODE >= 4) __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i] + z[i];
  out[blockIdx.x * blockDim.x + t] = s + lds[t];
}
template <int MODE>
void run(const char* name) {
  float* d;
  const int blocks = 256 * 4;
  hipMalloc(&d, sizeof(float) * blocks * 512);
  const int iters = 500;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, 10, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, iters, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_simd = (double)iters * 64 * 8;  // 8 waves per SIMD
  const double insts_per_cu = insts_per_simd * 4;
  printf("%-28s %.3f ms: %.2f ns per wave-instr per SIMD, %.2f ns per wave-instr per CU\n", name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_cu);
  hipFree(d);
}
int main() {
  run<0>("v_add_f32 v,v,v");
  run<1>("v_fma_f32 v,v,v,v");
  run<2>("v_mul_f32 literal");
  run<3>("v_fma_f32 literal");
  run<4>("ds_write_b32 pitch17");
  run<5>("ds_read_b32");
  run<6>("ds_write_b64 pitch33");
  run<7>("ds_read_b64");
  return 0;
}
