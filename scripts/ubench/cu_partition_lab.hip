// cu_partition_lab — would the stages overlap better on DISJOINT sets of CUs? The FFT + dB kernel and the detect kernel of the
// 8192-point chain on two streams created with CU masks (hipExtStreamCreateWithCUMask), alone and side by side, for several
// splits of the 256 CUs. Per-workgroup stamps of k_scan_step (profiles/r02) show a detect workgroup spending 7.6 of its
// 10 us waiting for its 36 rows on a CU that also streams frames; a CU's vector-memory pipeline returns in order.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o cu_partition_lab cu_partition_lab.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../rtl-sdr-scanner-cpp_amd/csrc/detect_fused.h"
#include "../../rtl-sdr-scanner-cpp_amd/csrc/fft8192_v2.h"

#define CK(x)                                                                           \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                          \
    }                                                                                   \
  } while (0)

__global__ void k_fillr(float* p, size_t n, unsigned seed, float scale, float bias) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    p[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * scale + bias;
  }
}

static hipStream_t masked_stream(int first_cu_per_xcd, int n_cu_per_xcd) {
  // mask bit layout: CU i of the device; CUs are numbered round-robin over the 8 XCDs (bit = cu_in_xcd * 8 + xcd) on gfx942/950
  std::vector<uint32_t> mask(8, 0u);
  for (int x = 0; x < 8; ++x)
    for (int c = first_cu_per_xcd; c < first_cu_per_xcd + n_cu_per_xcd; ++c) {
      const int bit = c * 8 + x;
      mask[bit >> 5] |= 1u << (bit & 31);
    }
  hipStream_t s;
  CK(hipExtStreamCreateWithCUMask(&s, 8, mask.data()));
  return s;
}

int main() {
  const int n = 8192, frames = 1024, nsets = 7;
  // FFT side
  std::vector<float> win(n);
  for (int i = 0; i < n; ++i) win[i] = (float)(0.54 - 0.46 * cos((2.0 * M_PI * i) / (float)(n - 1)));
  std::vector<float2> tw2(256), lane(384), wave(96);
  ss::fft8192_v2_host_tables(tw2.data(), lane.data(), wave.data());
  float* d_win;
  float2 *d_tw2, *d_lane, *d_wave;
  CK(hipMalloc(&d_win, n * 4));
  CK(hipMalloc(&d_tw2, 256 * 8));
  CK(hipMalloc(&d_lane, 384 * 8));
  CK(hipMalloc(&d_wave, 96 * 8));
  CK(hipMemcpy(d_win, win.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw2, tw2.data(), 256 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_lane, lane.data(), 384 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wave, wave.data(), 96 * 8, hipMemcpyHostToDevice));
  const ss::Fft8192V2Tables tabs{d_tw2, d_lane, d_wave, nullptr, nullptr};
  std::vector<float*> d_iq(nsets), d_psd(nsets);
  for (int k = 0; k < nsets; ++k) {
    CK(hipMalloc((void**)&d_iq[k], (size_t)frames * n * 8));
    CK(hipMalloc((void**)&d_psd[k], (size_t)frames * n * 4));
    hipLaunchKernelGGL(k_fillr, dim3(4096), dim3(256), 0, 0, d_iq[k], (size_t)frames * n * 2, 7u + k, 0.1f, 0.0f);
    hipLaunchKernelGGL(k_fillr, dim3(4096), dim3(256), 0, 0, d_psd[k], (size_t)frames * n, 77u + k, 5.0f, -60.0f);
  }
  // detect side
  float *d_thr, *d_hist, *d_avg;
  uint8_t* d_pass;
  uint32_t* d_mask;
  int* d_counts;
  CK(hipMalloc(&d_thr, n * 4));
  CK(hipMalloc(&d_hist, (size_t)128 * n * 4));
  CK(hipMalloc(&d_avg, (size_t)frames * n * 4));
  CK(hipMalloc(&d_pass, n));
  CK(hipMalloc(&d_mask, (size_t)frames * n / 8));
  CK(hipMalloc(&d_counts, frames * 4));
  hipLaunchKernelGGL(k_fillr, dim3(64), dim3(256), 0, 0, d_thr, (size_t)n, 5u, 1.0f, -62.0f);
  CK(hipMemset(d_hist, 0, (size_t)128 * n * 4));
  CK(hipMemset(d_pass, 1, n));
  CK(hipMemset(d_counts, 0, frames * 4));
  CK(hipDeviceSynchronize());
  auto det_args = [&](int k) {
    ss::DetectArgs a{};
    a.psd = d_psd[k];
    a.thr = d_thr;
    a.hist_in = d_hist;
    a.hist_out = d_hist + (size_t)64 * n;
    a.n = n;
    a.nframes = frames;
    a.n_learn = 0;
    a.pushed_before = 21;
    a.shift = 0;
    a.start_level = 8.0f;
    a.pass = d_pass;
    a.maskbits = d_mask;
    a.counts = d_counts;
    a.avg_sparse = d_avg;
    return a;
  };
  const int tiles = (frames / 16) * (n / 256);

  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreate(&f0));
  CK(hipEventCreate(&f1));
  const int iters = 40;
  {  // reference point: the detect kernel on an ordinary stream
    hipStream_t s0;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s0));
      for (int k = 0; k < iters; ++k) hipLaunchKernelGGL((ss::k_detect_fused<21, 21, 16, 256, false>), dim3(tiles), dim3(256), 0, s0, det_args(k % nsets));
      CK(hipEventRecord(e1, s0));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("detect on an ordinary stream: %.2f us per launch\n", ms / iters * 1e3);
    }
    CK(hipEventRecord(e0, s0));
    for (int k = 0; k < iters; ++k) {
      const ss::Fft8192Args g{d_iq[k % nsets], (long long)n, d_win, tabs, 63.1f, 1.0f, d_psd[(k + 3) % nsets]};
      hipLaunchKernelGGL((ss::k_fft8192_psd_v2<ss::FMT_CF32, 2, true>), dim3(frames), dim3(512), ss::kFft8192V2LdsBytes, s0, g);
    }
    CK(hipEventRecord(e1, s0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fft on an ordinary stream: %.2f us per launch\n", ms / iters * 1e3);
  }
  printf("%-44s %10s %10s %10s\n", "CUs per XCD: FFT | detect", "fft us", "detect us", "both us");
  for (int det_cus : {0, 4, 5, 6, 7, 8, 10, 12, 16}) {
    hipStream_t sf = det_cus ? masked_stream(det_cus, 32 - det_cus) : masked_stream(0, 32);
    hipStream_t sd = det_cus ? masked_stream(0, det_cus) : masked_stream(0, 32);
    auto fft = [&](int k) {
      const ss::Fft8192Args g{d_iq[k % nsets], (long long)n, d_win, tabs, 63.1f, 1.0f, d_psd[(k + 3) % nsets]};
      hipLaunchKernelGGL((ss::k_fft8192_psd_v2<ss::FMT_CF32, 2, true>), dim3(frames), dim3(512), ss::kFft8192V2LdsBytes, sf, g);
    };
    auto det = [&](int k) {
      hipLaunchKernelGGL((ss::k_detect_fused<21, 21, 16, 256, false>), dim3(tiles), dim3(256), 0, sd, det_args(k % nsets));
    };
    float ms_f = 0, ms_d = 0, ms_b = 0;
    for (int k = 0; k < 4; ++k) fft(k), det(k);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sf));
    for (int k = 0; k < iters; ++k) fft(k);
    CK(hipEventRecord(e1, sf));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_f, e0, e1));
    CK(hipEventRecord(e0, sd));
    for (int k = 0; k < iters; ++k) det(k);
    CK(hipEventRecord(e1, sd));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_d, e0, e1));
    // side by side: wall time until both streams have done `iters` launches each
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sf));
    CK(hipEventRecord(f0, sd));
    for (int k = 0; k < iters; ++k) {
      fft(k);
      det(k);
    }
    CK(hipEventRecord(e1, sf));
    CK(hipEventRecord(f1, sd));
    CK(hipEventSynchronize(e1));
    CK(hipEventSynchronize(f1));
    float a, b;
    CK(hipEventElapsedTime(&a, e0, e1));
    CK(hipEventElapsedTime(&b, f0, f1));
    ms_b = a > b ? a : b;
    char label[64];
    snprintf(label, sizeof label, det_cus ? "%d | %d" : "32 | 32 (no masks)", 32 - det_cus, det_cus);
    printf("%-44s %10.2f %10.2f %10.2f   (side by side: fft %.2f, detect %.2f)\n", label, ms_f / iters * 1e3, ms_d / iters * 1e3, ms_b / iters * 1e3, a / iters * 1e3,
           b / iters * 1e3);
    fflush(stdout);
    CK(hipStreamDestroy(sf));
    CK(hipStreamDestroy(sd));
  }
  return 0;
}
