// mfma_fold_lab — can the fold's W_8 butterflies (csrc/fft65536_dif8.h) ride on the matrix pipe? (round 5's review, item 3)
// The fold of a 65536-point frame is, per point n, a 16-vector (re, im of the eight windowed samples x[n + 8192 q]) times the 16 x 16 real
// form of DFT_8. As v_mfma_f32_16x16x4_f32: 16 points per instruction, K = 16 in four steps — four MFMAs per 16 points give all EIGHT
// residues (16 real outputs per point). A workgroup of k_scan_step KIND 8 needs TWO residues (4 of the 16 output columns): the other
// twelve are computed and thrown away, because the eight residues' 8192-point transforms (64 KiB of points each) do not fit one
// workgroup's registers or LDS, and a work buffer for them is what the fold exists to avoid.
// This lab measures what the argument needs from the hardware:
//   1  the rate of v_mfma_f32_16x16x4_f32 per SIMD (cycles per instruction at 1, 2, 4 waves per SIMD),
//   2  the rate of v_fma_f32 beside it (same waves, interleaved 1 MFMA : k FMAs) — do the two pipes overlap, and what does a wave-64 FMA cost,
//   3  from those: time on the matrix pipe for the fold of one frame by one workgroup (two residues wanted) against the VALU instructions the
//      butterfly form spends on the same sums (8-10 per point, dif8_bfly2) and the accumulating form (32 per point).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_fold_lab mfma_fold_lab.hip    run: gpurun -- scripts/ubench/mfma_fold_lab
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

typedef float float4v __attribute__((ext_vector_type(4)));

// MODE 0: MFMA only (4 independent accumulators); 1: FMA only (16 independent chains); 2: 1 MFMA : 8 FMAs interleaved; 3: 1 MFMA : 16 FMAs
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float4v acc[4];
  float x[16];
  for (int i = 0; i < 4; ++i) acc[i] = (float4v){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  const float av = a + threadIdx.x * 1e-6f, bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0 || MODE >= 2) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[u], 0, 0, 0);
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) x[(8 * u + i) & 15] = fmaf(x[(8 * u + i) & 15], a, b);
      }
      if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], a, b);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(int waves_per_simd, int iters) {  // ms
  float* d;
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 per SIMD of a CU
  CK(hipMalloc(&d, sizeof(float) * blocks * 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10, 1.0001f, 0.5f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipFree(d));
  return ms;
}

int main() {
  const int iters = 4000;
  double mfma_ns = 0, fma_ns = 0;
  for (int w : {1, 2, 4}) {
    const double m0 = run<0>(w, iters), m1 = run<1>(w, iters), m2 = run<2>(w, iters), m3 = run<3>(w, iters);
    const double n_mfma = (double)iters * 4 * w, n_fma8 = (double)iters * 32 * w, n_fma16 = (double)iters * 64 * w;  // wave-instructions per SIMD
    printf("waves/SIMD %d: MFMA 16x16x4 f32 alone %.2f ns per instr per SIMD | FMA alone %.2f ns | 1 MFMA + 8 FMA: %.2f ns per group (sum of the two alone: %.2f) | "
           "1 MFMA + 16 FMA: %.2f ns per group (sum alone: %.2f)\n",
           w, m0 * 1e6 / n_mfma, m1 * 1e6 / n_fma8, m2 * 1e6 / n_mfma, m0 * 1e6 / n_mfma + 8 * m1 * 1e6 / n_fma8, m3 * 1e6 / n_mfma, m0 * 1e6 / n_mfma + 16 * m1 * 1e6 / n_fma8);
    (void)n_fma16;
    if (w == 4) {
      mfma_ns = m0 * 1e6 / n_mfma;
      fma_ns = m1 * 1e6 / n_fma8;
    }
  }
  // the fold of one frame by one workgroup: 8192 points, all in one workgroup's eight waves (two per SIMD)
  const double mfma_per_wg = 8192.0 / 16.0 * 4.0;          // MFMAs: four per sixteen points
  const double per_simd = mfma_per_wg / 4.0;                // a workgroup's waves spread over the four SIMDs
  printf("fold of one frame's 8192 points for one workgroup on the matrix pipe: %.0f MFMAs = %.0f per SIMD = %.1f us at the measured rate (the workgroup's whole life today: ~20-30 us,\n"
         "  of which the sums the MFMAs would replace are 8-10 of the butterfly form's 57 vector instructions per point: %.1f us of the VALU at the measured FMA rate;\n"
         "  the accumulating form spent 32 per point: %.1f us)\n",
         mfma_per_wg, per_simd, per_simd * mfma_ns * 1e-3, 8192.0 * 9.0 / 64.0 / 4.0 * fma_ns * 1e-3, 8192.0 * 32.0 / 64.0 / 4.0 * fma_ns * 1e-3);
  return 0;
}
