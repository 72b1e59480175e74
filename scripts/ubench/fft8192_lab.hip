// fft8192_lab — the 8192-point front-end kernels side by side on a working set that does not fit the 256 MiB Infinity
// Cache: every launch reads a different 64 MiB batch and writes a different 32 MiB plane (>= 640 MiB of input in rotation).
// Prints, per variant and batch size, the kernel time from start/stop events attached to each launch, the algorithmic
// bandwidth (12 B/sample) and the largest difference of the dB plane from the first-generation kernel.
//   build: make -C scripts/ubench fft8192_lab      run: gpurun -- scripts/ubench/fft8192_lab [frames ...]
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "fft8192_round1.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed;
    x ^= x >> 15;
    x *= 2246822519u;
    x ^= x >> 13;
    p[i] = ((float)(x & 0xffffff) / 8388608.0f - 1.0f) * 0.1f;
  }
}

struct Variant {
  std::string name;
  std::function<void(const void*, float*, int, hipStream_t, hipEvent_t, hipEvent_t)> launch;
  bool check;
};

int main(int argc, char** argv) {
  std::vector<int> frame_counts;
  for (int i = 1; i < argc; ++i) frame_counts.push_back(atoi(argv[i]));
  if (frame_counts.empty()) frame_counts = {1024, 2048, 4096};
  int max_frames = 0;
  for (int f : frame_counts) max_frames = f > max_frames ? f : max_frames;

  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs\n", prop.name, prop.multiProcessorCount);

  // tables
  const int n = 8192;
  std::vector<float> win(n);
  for (int i = 0; i < n; ++i) win[i] = (float)(0.54 - 0.46 * cos((2.0 * M_PI * i) / (float)(n - 1)));
  std::vector<float2> t8(256 + 1024 + 2048), tw2(256), lane(384), wave(96);
  auto W = [](double num, double den) {
    const double ang = -2.0 * M_PI * num / den;
    return make_float2((float)cos(ang), (float)sin(ang));
  };
  for (int r = 0; r < 16; ++r)
    for (int m = 0; m < 16; ++m) t8[r * 16 + m] = W((double)m * r, 256.0);
  for (int r1 = 0; r1 < 4; ++r1)
    for (int t = 0; t < 256; ++t) t8[256 + r1 * 256 + t] = W((double)t * r1, 8192.0);
  for (int r2 = 0; r2 < 8; ++r2)
    for (int t = 0; t < 256; ++t) t8[256 + 1024 + r2 * 256 + t] = W((double)t * r2, 2048.0);
  ss::fft8192_v2_host_tables(tw2.data(), lane.data(), wave.data());
  float *d_win;
  float2 *d_t8, *d_tw2, *d_lane, *d_wave;
  CK(hipMalloc(&d_win, n * 4));
  CK(hipMalloc(&d_t8, t8.size() * 8));
  CK(hipMalloc(&d_tw2, 256 * 8));
  CK(hipMalloc(&d_lane, 384 * 8));
  CK(hipMalloc(&d_wave, 96 * 8));
  CK(hipMemcpy(d_win, win.data(), n * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_t8, t8.data(), t8.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_tw2, tw2.data(), 256 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_lane, lane.data(), 384 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wave, wave.data(), 96 * 8, hipMemcpyHostToDevice));
  const ss::Fft8192Tables tabs1{d_t8, d_t8 + 256, d_t8 + 256 + 1024, nullptr};
  const ss::Fft8192V2Tables tabs2{d_tw2, d_lane, d_wave, d_t8 + 256, d_t8 + 256 + 1024};
  const ss::Fft8192Second none{nullptr, 0, nullptr, 0};
  const float db_off = (float)(10.0 * log10(2048000.0));

  // rotating working set: one pool, cut into as many sets as the batch size allows (>= 3 sets, >= 768 MiB of input)
  size_t pool_in = (size_t)768 << 20;
  if (pool_in < (size_t)3 * max_frames * n * 8) pool_in = (size_t)3 * max_frames * n * 8;
  const size_t pool_out = pool_in / 2;
  char *pool_in_d, *pool_out_d;
  CK(hipMalloc((void**)&pool_in_d, pool_in));
  CK(hipMalloc((void**)&pool_out_d, pool_out));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, (float*)pool_in_d, pool_in / 4, 12345u);
  CK(hipDeviceSynchronize());
  int nsets = 0;
  std::vector<void*> d_in;
  std::vector<float*> d_out;

  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

#define W8(...)                                                                                                                          \
  [&](const void* in, float* out, int frames, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {                                             \
    hipExtLaunchKernelGGL((ss::k_fft8192_psd_w8<__VA_ARGS__>), dim3(frames), dim3(512), ss::kFft8192W8LdsBytes, s, e0, e1, 0, in,       \
                          (long long)8192, (const float*)d_win, tabs1, db_off, 1.0f, out, none);                                         \
  }
#define V2L(LDSB, ...)                                                                                                                   \
  [&](const void* in, float* out, int frames, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {                                             \
    const ss::Fft8192Args g{in, (long long)8192, (const float*)d_win, tabs2, db_off, 1.0f, out};                                         \
    hipExtLaunchKernelGGL((ss::k_fft8192_psd_v2<__VA_ARGS__>), dim3(frames), dim3(512), LDSB, s, e0, e1, 0, g);                         \
  }
#define V2(...)                                                                                                                          \
  [&](const void* in, float* out, int frames, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {                                             \
    const ss::Fft8192Args g{in, (long long)8192, (const float*)d_win, tabs2, db_off, 1.0f, out};                                         \
    hipExtLaunchKernelGGL((ss::k_fft8192_psd_v2<__VA_ARGS__>), dim3(frames), dim3(512), ss::kFft8192V2LdsBytes, s, e0, e1, 0, g);       \
  }
  std::vector<Variant> variants = {
      {"w8 (first generation)", W8(ss::FMT_CF32, 8), true},
      {"w8 memory traffic only", W8(ss::FMT_CF32, 8, false, 1), false},
      {"w8 memory traffic 16B", W8(ss::FMT_CF32, 8, false, 4), false},
      {"w8 transform only", W8(ss::FMT_CF32, 8, false, 2), false},
      {"v2 TW0 (tables global)", V2(ss::FMT_CF32, 0), true},
      {"v2 TW1 (pass-2 table LDS)", V2(ss::FMT_CF32, 1), true},
      {"v2 TW2 (all tables LDS/SGPR)", V2(ss::FMT_CF32, 2), true},
      {"v2 TW3 (no tables, bound)", V2(ss::FMT_CF32, 3), false},
      {"v2 TW0 + swizzled exchange", V2(ss::FMT_CF32, 0, true), true},
      {"v2 TW2 + swizzled exchange", V2(ss::FMT_CF32, 2, true), true},
      {"v2 TW2+swz, 3 workgroups per CU", V2L(53 * 1024, ss::FMT_CF32, 2, true), true},
      {"v2 TW2+swz, 2 workgroups per CU", V2L(80 * 1024, ss::FMT_CF32, 2, true), true},
      {"v2 TW2, no window (bound)", V2(ss::FMT_CF32, 2, false, true), false},
      {"v2 TW3, no window (bound)", V2(ss::FMT_CF32, 3, false, true), false},
  };

  const int iters = 48;
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  hipEvent_t w0, w1;
  CK(hipEventCreate(&w0));
  CK(hipEventCreate(&w1));
  std::vector<float> ref((size_t)1024 * n), got((size_t)1024 * n);

  for (int frames : frame_counts) {
    nsets = (int)(pool_in / ((size_t)frames * n * 8));
    d_in.assign(nsets, nullptr);
    d_out.assign(nsets, nullptr);
    for (int k = 0; k < nsets; ++k) {
      d_in[k] = pool_in_d + (size_t)k * frames * n * 8;
      d_out[k] = reinterpret_cast<float*>(pool_out_d + (size_t)k * frames * n * 4);
    }
    printf("\n# %d sets of %d MiB in + %d MiB out in rotation\n", nsets, frames * n * 8 >> 20, frames * n * 4 >> 20);
    printf("\n## %d frames per launch (%.1f MB algorithmic)\n", frames, 12.0 * frames * n / 1e6);
    printf("%-32s %9s %9s %9s %7s %9s  %s\n", "variant", "kern us", "min us", "loop us", "GB/s", "% 8TB/s", "max |dB diff| vs w8");
    bool have_ref = false;
    for (auto& v : variants) {
      for (int k = 0; k < 4; ++k) v.launch(d_in[k % nsets], d_out[k % nsets], frames, st, nullptr, nullptr);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(w0, st));
      for (int k = 0; k < iters; ++k) v.launch(d_in[k % nsets], d_out[k % nsets], frames, st, ev[2 * k], ev[2 * k + 1]);
      CK(hipEventRecord(w1, st));
      CK(hipStreamSynchronize(st));
      CK(hipGetLastError());
      double sum = 0, mn = 1e9;
      for (int k = 0; k < iters; ++k) {
        float ms;
        CK(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        sum += ms;
        if (ms < mn) mn = ms;
      }
      float loop_ms;
      CK(hipEventElapsedTime(&loop_ms, w0, w1));
      const double us = sum / iters * 1e3;
      const double gbs = 12.0 * frames * n / (us * 1e-6) / 1e9;
      char diff[64] = "-";
      if (v.check) {
        // same input (set 0) through this variant
        v.launch(d_in[0], d_out[0], frames, st, nullptr, nullptr);
        CK(hipStreamSynchronize(st));
        const size_t cnt = (size_t)(frames < 1024 ? frames : 1024) * n;
        CK(hipMemcpy(got.data(), d_out[0], cnt * 4, hipMemcpyDeviceToHost));
        if (!have_ref) {
          ref = got;
          have_ref = true;
          snprintf(diff, sizeof diff, "(reference)");
        } else {
          double md = 0;
          size_t bad = 0;
          for (size_t i = 0; i < cnt; ++i) {
            const double d = fabs((double)got[i] - (double)ref[i]);
            if (!(d <= 1e30)) ++bad;
            else if (d > md) md = d;
          }
          snprintf(diff, sizeof diff, "%.3g%s", md, bad ? " (+non-finite mismatches)" : "");
        }
      }
      printf("%-32s %9.2f %9.2f %9.2f %7.0f %8.1f%%  %s\n", v.name.c_str(), us, mn * 1e3, loop_ms / iters * 1e3, gbs, gbs / 80.0, diff);
      fflush(stdout);
    }
  }
  return 0;
}
