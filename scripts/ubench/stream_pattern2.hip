// Micro-benchmark 7: which side of the FFT kernel's access pattern costs bandwidth? 1024 frames, 64 MiB in, 32 MiB out,
// 512 threads per frame. Loads: 8 B/lane 4 KiB apart (as now) or 16 B/lane. Stores: 4 B/lane 1 KiB apart (as now) or 16 B/lane contiguous.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int LD16, int ST16>
__global__ __launch_bounds__(512) void k(const float2* __restrict__ in, float* __restrict__ out) {
  const int t = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * 8192;
  float2 a[16];
  if (LD16) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 q = *reinterpret_cast<const float4*>(in + base + 2 * t + 1024 * r);
      a[2 * r] = make_float2(q.x, q.y);
      a[2 * r + 1] = make_float2(q.z, q.w);
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = in[base + t + 512 * r];
  }
  if (ST16) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<float4*>(out + base + 4 * t + 2048 * k) = make_float4(a[4 * k].x + a[4 * k].y, a[4 * k + 1].x - a[4 * k + 1].y, a[4 * k + 2].x, a[4 * k + 3].y);
  } else {
    const int lane = t & 63, h = lane >> 5, j = ((t >> 6) << 5) + (lane & 31);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kk = k + 8 * h;
      const int bin0 = j + 256 * kk, bin1 = bin0 + 4096;
      out[base + (bin0 ^ 4096)] = a[k].x + a[k + 8].y;
      out[base + (bin1 ^ 4096)] = a[k].y - a[k + 8].x;
    }
  }
}
template <typename F>
float time_it(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  float best = 1e9;
  for (int rep = 0; rep < 20; ++rep) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
  return best * 1e3f;
}
int main() {
  const size_t rbytes = 64ull << 20, wbytes = 32ull << 20;
  void *in, *out; hipMalloc(&in, rbytes); hipMalloc(&out, wbytes); hipMemset(in, 0, rbytes);
  const double bytes = (double)(rbytes + wbytes);
  float us;
  us = time_it([&] { hipLaunchKernelGGL((k<0, 0>), dim3(1024), dim3(512), 0, 0, (const float2*)in, (float*)out); }); printf("load 8B,  store 4B : %5.1f us %.0f GB/s\n", us, bytes / us * 1e-3);
  us = time_it([&] { hipLaunchKernelGGL((k<1, 0>), dim3(1024), dim3(512), 0, 0, (const float2*)in, (float*)out); }); printf("load 16B, store 4B : %5.1f us %.0f GB/s\n", us, bytes / us * 1e-3);
  us = time_it([&] { hipLaunchKernelGGL((k<0, 1>), dim3(1024), dim3(512), 0, 0, (const float2*)in, (float*)out); }); printf("load 8B,  store 16B: %5.1f us %.0f GB/s\n", us, bytes / us * 1e-3);
  us = time_it([&] { hipLaunchKernelGGL((k<1, 1>), dim3(1024), dim3(512), 0, 0, (const float2*)in, (float*)out); }); printf("load 16B, store 16B: %5.1f us %.0f GB/s\n", us, bytes / us * 1e-3);
  return 0;
}
