// Micro-benchmark 4: throughput of a loop whose body is B straight-line fp32 FMAs (8-byte VOP3 encodings with literals),
// total work fixed at 8192 FMAs per wave, 8 waves per SIMD on every CU. Shows where the code footprint stops fitting.
#include <hip/hip_runtime.h>
#include <cstdio>
#define F8(i) x[0] = fmaf(x[0], a, b + i); x[1] = fmaf(x[1], a, b); x[2] = fmaf(x[2], b, a); x[3] = fmaf(x[3], a, b); \
              x[4] = fmaf(x[4], b, a); x[5] = fmaf(x[5], a, b); x[6] = fmaf(x[6], b, a); x[7] = fmaf(x[7], a, b);
#define F64(i) F8(i) F8(i+1) F8(i+2) F8(i+3) F8(i+4) F8(i+5) F8(i+6) F8(i+7)
#define F256(i) F64(i) F64(i+8) F64(i+16) F64(i+24)
#define F1024(i) F256(i) F256(i+32) F256(i+64) F256(i+96)
template <int B>
__global__ __launch_bounds__(512) void k(float* out, int trips, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int r = 0; r < trips; ++r) {
    if (B == 64) { F64(1) }
    if (B == 128) { F64(1) F64(9) }
    if (B == 256) { F256(1) }
    if (B == 512) { F256(1) F256(33) }
    if (B == 1024) { F1024(1) }
    if (B == 2048) { F1024(1) F1024(129) }
    if (B == 4096) { F1024(1) F1024(129) F1024(257) F1024(385) }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int B>
void run() {
  float* d;
  const int blocks = 1024;
  hipMalloc(&d, sizeof(float) * blocks * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int trips = 8192 / B;
  hipLaunchKernelGGL(k<B>, dim3(blocks), dim3(512), 0, 0, d, trips, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<B>, dim3(blocks), dim3(512), 0, 0, d, trips, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("body %4d instrs (~%2d KiB): %7.1f us  %.2f ns per wave-instr per SIMD\n", B, B * 8 / 1024, ms * 1e3, ms * 1e6 / (8192.0 * 8));
  hipFree(d);
}
int main() {
  run<64>(); run<128>(); run<256>(); run<512>(); run<1024>(); run<2048>(); run<4096>();
  return 0;
}
