// Micro-benchmark 5: how fast can one launch move the FFT kernel's HBM traffic (64 MiB in, 32 MiB out, 1024 frames)
// with (a) an ideal streaming pattern (16 B/lane contiguous) and (b) the FFT kernel's own access pattern
// (512 threads per frame, 8 B/lane loads 4 KiB apart, 4 B/lane stores 1 KiB apart), with trivial arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void k_ideal(const float4* __restrict__ in, float4* __restrict__ out, size_t n_out4) {
  // each thread: read two float4, write one
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_out4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 a = in[2 * i], b = in[2 * i + 1];
    out[i] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
  }
}
template <int LDSBYTES>
__global__ __launch_bounds__(512) void k_fftlike(const float2* __restrict__ in, float* __restrict__ out) {
  extern __shared__ float lds[];
  const int t = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 8192;
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = in[base + t + 512 * r];
  if (LDSBYTES > 0) { lds[t] = a[0].x; __syncthreads(); a[1].x += lds[(t + 1) & 511]; }
  const int lane = t & 63, h = lane >> 5, j = ((t >> 6) << 5) + (lane & 31);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int kk = k + 8 * h;
    const int bin0 = j + 256 * kk, bin1 = bin0 + 4096;
    out[base + (bin0 ^ 4096)] = a[k].x + a[k + 8].y;
    out[base + (bin1 ^ 4096)] = a[k].y - a[k + 8].x;
  }
}
template <typename F>
float time_it(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 20; ++rep) {
    hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best * 1e3f;
}
int main() {
  const size_t rbytes = 64ull << 20, wbytes = 32ull << 20;
  void *in, *out; hipMalloc(&in, rbytes); hipMalloc(&out, wbytes); hipMemset(in, 0, rbytes);
  const double bytes = (double)(rbytes + wbytes);
  float us = time_it([&] { hipLaunchKernelGGL(k_ideal, dim3(2048), dim3(256), 0, 0, (const float4*)in, (float4*)out, wbytes / 16); });
  printf("ideal 16B streaming            : %6.1f us  %.0f GB/s\n", us, bytes / us * 1e-3);
  us = time_it([&] { hipLaunchKernelGGL(k_fftlike<0>, dim3(1024), dim3(512), 0, 0, (const float2*)in, (float*)out); });
  printf("FFT pattern, no LDS            : %6.1f us  %.0f GB/s\n", us, bytes / us * 1e-3);
  us = time_it([&] { hipLaunchKernelGGL(k_fftlike<34816>, dim3(1024), dim3(512), 34816, 0, (const float2*)in, (float*)out); });
  printf("FFT pattern, 34 KiB LDS (4/CU) : %6.1f us  %.0f GB/s\n", us, bytes / us * 1e-3);
  us = time_it([&] { hipLaunchKernelGGL(k_fftlike<69632>, dim3(1024), dim3(512), 69632, 0, (const float2*)in, (float*)out); });
  printf("FFT pattern, 68 KiB LDS (2/CU) : %6.1f us  %.0f GB/s\n", us, bytes / us * 1e-3);
  return 0;
}
