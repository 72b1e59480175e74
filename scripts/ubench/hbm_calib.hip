// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the FFT kernel uses
// (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by 2x; other widths must be calibrated).
//   k_read8 : reads `bytes` with 8-byte-per-lane loads (global_load_dwordx2), writes 4 bytes per workgroup
//   k_read16: reads `bytes` with 16-byte-per-lane loads (global_load_dwordx4)
//   k_write4: writes `bytes` with 4-byte-per-lane stores (global_store_dword), 256 B per wave instruction
// Known byte counts: 64 MiB read, 32 MiB written — the same as one launch of the 1024 x 8192 FFT kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_read8(const float2* in, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float2 v = in[i]; acc += v.x + v.y; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_read16(const float4* in, size_t n, float* out) {
  float acc = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}
__global__ void k_write4(float* out, size_t n, float v) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = v + (float)(i & 7);
}
int main() {
  const size_t rbytes = 64ull << 20, wbytes = 32ull << 20;
  void *in, *out;
  hipMalloc(&in, rbytes); hipMalloc(&out, wbytes);
  hipMemset(in, 1, rbytes); hipMemset(out, 0, wbytes);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 5; ++rep) {
    hipLaunchKernelGGL(k_read8, dim3(2048), dim3(256), 0, 0, (const float2*)in, rbytes / 8, (float*)out);
    hipLaunchKernelGGL(k_read16, dim3(2048), dim3(256), 0, 0, (const float4*)in, rbytes / 16, (float*)out);
    hipLaunchKernelGGL(k_write4, dim3(2048), dim3(256), 0, 0, (float*)out, wbytes / 4, 1.0f);
  }
  hipDeviceSynchronize();
  printf("calibration kernels: read8 %zu B, read16 %zu B, write4 %zu B per launch\n", rbytes, rbytes, wbytes);
  return 0;
}
