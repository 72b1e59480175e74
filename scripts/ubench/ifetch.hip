// Micro-benchmark 3: is straight-line (fully unrolled) code limited by instruction fetch? Same 2048 dependent-free
// fp32 FMAs per wave either as a 64-instruction loop body x 32 trips, or as 2048 instructions of straight-line code,
// run by 8192 waves (like the FFT kernel's 1024 workgroups x 8 waves), each wave executing the body REPS times.
#include <hip/hip_runtime.h>
#include <cstdio>
#define F8(i) x[0] = fmaf(x[0], a, b + i); x[1] = fmaf(x[1], a, b); x[2] = fmaf(x[2], b, a); x[3] = fmaf(x[3], a, b); \
              x[4] = fmaf(x[4], b, a); x[5] = fmaf(x[5], a, b); x[6] = fmaf(x[6], b, a); x[7] = fmaf(x[7], a, b);
#define F64(i) F8(i) F8(i+1) F8(i+2) F8(i+3) F8(i+4) F8(i+5) F8(i+6) F8(i+7)
#define F512(i) F64(i) F64(i+8) F64(i+16) F64(i+24) F64(i+32) F64(i+40) F64(i+48) F64(i+56)
#define F2048(i) F512(i) F512(i+64) F512(i+128) F512(i+192)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int reps, float a, float b) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.001f + i;
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) {
      for (int it = 0; it < 32; ++it) { F64(it) }
    } else {
      F2048(1)
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int reps) {
  float* d;
  const int blocks = 1024;
  hipMalloc(&d, sizeof(float) * blocks * 512);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, reps, 1.0001f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(512), 0, 0, d, reps, 1.0001f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-22s reps %d: %.1f us  (%.2f ns per wave-instr per SIMD)\n", name, reps, ms * 1e3, ms * 1e6 / (2048.0 * reps * 8));
  hipFree(d);
}
int main() {
  for (int reps : {1, 4}) {
    run<0>("loop 64 x 32", reps);
    run<1>("straight-line 2048", reps);
  }
  return 0;
}
