// launch_overlap_lab — can the tail of one launch be filled by the head of the next? The FFT + dB kernel and the detect kernel of
// the 8192-point chain, 1024 frames per launch, enqueued (a) on one stream (every dispatch packet carries the barrier bit),
// (b) on one stream with hipExtAnyOrderLaunch (barrier bit cleared), (c) alternating over two / three streams. No launch
// depends on another here (separate buffers); wall time per launch from the host clock around 200 launches.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o launch_overlap_lab launch_overlap_lab.hip
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
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

int main() {
  const int n = 8192, frames = 1024, nsets = 7;
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
  std::vector<float*> d_iq(nsets), d_psd(nsets), d_psd_in(nsets);
  for (int k = 0; k < nsets; ++k) {
    CK(hipMalloc((void**)&d_iq[k], (size_t)frames * n * 8));
    CK(hipMalloc((void**)&d_psd[k], (size_t)frames * n * 4));
    CK(hipMalloc((void**)&d_psd_in[k], (size_t)frames * n * 4));
    hipLaunchKernelGGL(k_fillr, dim3(4096), dim3(256), 0, 0, d_iq[k], (size_t)frames * n * 2, 7u + k, 0.1f, 0.0f);
    hipLaunchKernelGGL(k_fillr, dim3(4096), dim3(256), 0, 0, d_psd_in[k], (size_t)frames * n, 77u + k, 5.0f, -60.0f);
  }
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
  hipLaunchKernelGGL(k_fillr, dim3(64), dim3(256), 0, 0, d_thr, (size_t)n, 5u, 1.0f, 40.0f);  // above every plane: no candidates anywhere (the detect stage costs more where it has hits to record)
  CK(hipMemset(d_hist, 0, (size_t)128 * n * 4));
  CK(hipMemset(d_pass, 1, n));
  CK(hipMemset(d_counts, 0, frames * 4));
  CK(hipDeviceSynchronize());
  auto det_args = [&](int k) {
    ss::DetectArgs a{};
    a.psd = d_psd_in[k];
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
  hipStream_t st[3];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int iters = 210;
  auto fft = [&](int k, hipStream_t s, int flags) {
    const ss::Fft8192Args g{d_iq[k % nsets], (long long)n, d_win, tabs, 63.1f, 1.0f, d_psd[k % nsets]};
    hipExtLaunchKernelGGL((ss::k_fft8192_psd_v2<ss::FMT_CF32, 2, true>), dim3(frames), dim3(512), ss::kFft8192V2LdsBytes, s, nullptr, nullptr, flags, g);
  };
  std::vector<float*> d_seg(nsets);
  for (int k = 0; k < nsets; ++k) CK(hipMalloc((void**)&d_seg[k], (size_t)(frames * 32 + 64) * 4));
  auto fft_seg = [&](int k, hipStream_t s, int flags) {  // with the per-segment maxima for the detect stage's tile culling
    const ss::Fft8192Args g{d_iq[k % nsets], (long long)n, d_win, tabs, 63.1f, 1.0f, d_psd[k % nsets], d_seg[k % nsets], frames};
    hipExtLaunchKernelGGL((ss::k_fft8192_psd_v2<ss::FMT_CF32, 2, true>), dim3(frames), dim3(512), ss::kFft8192V2LdsBytes, s, nullptr, nullptr, flags, g);
  };
  auto det = [&](int k, hipStream_t s, int flags) {
    hipExtLaunchKernelGGL((ss::k_detect_fused<21, 21, 16, 256, false>), dim3(tiles), dim3(256), 0, s, nullptr, nullptr, flags, det_args(k % nsets));
  };
  auto timed = [&](const char* name, auto&& body) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < iters; ++k) body(k);
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (rep) printf("%-64s %8.2f us per iteration\n", name, us / iters);
    }
    fflush(stdout);
  };
  for (int k = 0; k < 2000; ++k) fft(k, st[0], 0);  // clocks up
  timed("fft, one stream", [&](int k) { fft(k, st[0], 0); });
  timed("fft, one stream, any-order launches", [&](int k) { fft(k, st[0], hipExtAnyOrderLaunch); });
  timed("fft, two streams alternating", [&](int k) { fft(k, st[k & 1], 0); });
  timed("fft, three streams alternating", [&](int k) { fft(k, st[k % 3], 0); });
  timed("fft + segment maxima, one stream", [&](int k) { fft_seg(k, st[0], 0); });
  timed("fft + segment maxima, two streams alternating", [&](int k) { fft_seg(k, st[k & 1], 0); });
  if (getenv("LAB_FFT_ONLY")) return 0;
  timed("detect, one stream", [&](int k) { det(k, st[0], 0); });
  timed("detect, one stream, any-order launches", [&](int k) { det(k, st[0], hipExtAnyOrderLaunch); });
  timed("fft + detect, one stream", [&](int k) { fft(k, st[0], 0); det(k, st[0], 0); });
  timed("fft + detect, one stream, any-order launches", [&](int k) { fft(k, st[0], hipExtAnyOrderLaunch); det(k, st[0], hipExtAnyOrderLaunch); });
  timed("fft + detect, fft on stream 0, detect on stream 1", [&](int k) { fft(k, st[0], 0); det(k, st[1], 0); });
  timed("fft + detect, pairs alternate over two streams", [&](int k) { fft(k, st[k & 1], 0); det(k, st[k & 1], 0); });
  timed("fft + detect, both any-order, pairs alternate over two streams", [&](int k) { fft(k, st[k & 1], hipExtAnyOrderLaunch); det(k, st[k & 1], hipExtAnyOrderLaunch); });
  // with the chain's real dependency: detect(k) reads the plane FFT(k) wrote, through an event
  hipStream_t s_det, s_main;
  CK(hipStreamCreateWithFlags(&s_det, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s_main, hipStreamNonBlocking));
  hipEvent_t ev[8], evin[8];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  for (auto& e : evin) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  auto det_of = [&](int k, hipStream_t s) {  // detect over the plane FFT(k) writes
    ss::DetectArgs a = det_args(k % nsets);
    a.psd = d_psd[k % nsets];
    hipLaunchKernelGGL((ss::k_detect_fused<21, 21, 16, 256, false>), dim3(tiles), dim3(256), 0, s, a);
  };
  timed("dependent: fft(k); detect(k) on ONE stream", [&](int k) { fft(k, st[0], 0); det_of(k, st[0]); });
  timed("dependent: fft on stream 0, detect on s_det behind an event", [&](int k) {
    fft(k, st[0], 0);
    CK(hipEventRecord(ev[k & 7], st[0]));
    CK(hipStreamWaitEvent(s_det, ev[k & 7], 0));
    det_of(k, s_det);
  });
  timed("dependent: fft alternating over two streams, detect on s_det behind an event", [&](int k) {
    fft(k, st[k & 1], 0);
    CK(hipEventRecord(ev[k & 7], st[k & 1]));
    CK(hipStreamWaitEvent(s_det, ev[k & 7], 0));
    det_of(k, s_det);
  });
  timed("  + every fft forked from an idle public stream by an event", [&](int k) {
    CK(hipEventRecord(evin[k & 7], s_main));
    CK(hipStreamWaitEvent(st[k & 1], evin[k & 7], 0));
    fft(k, st[k & 1], 0);
    CK(hipEventRecord(ev[k & 7], st[k & 1]));
    CK(hipStreamWaitEvent(s_det, ev[k & 7], 0));
    det_of(k, s_det);
  });
  timed("dependent: fft alternating over three streams, detect on s_det behind an event", [&](int k) {
    fft(k, st[k % 3], 0);
    CK(hipEventRecord(ev[k & 7], st[k % 3]));
    CK(hipStreamWaitEvent(s_det, ev[k & 7], 0));
    det_of(k, s_det);
  });
  return 0;
}
