// dif8_lab — 65536-point frames WITHOUT a work buffer: the radix-8 decimation-in-frequency fold in the load stage of the 8192-point
// transform (rtl-sdr-scanner-cpp_amd/csrc/fft65536_dif8.h), eight workgroups per frame, against what round 4 ships for BASELINE
// config 3 (two halves of a four-step transform through a CF32 work buffer: 40-41 us per 128-frame call, 25.9 B/sample over the fabric).
// The kernel here is the real thing — load + fold + transform + dB + noise-relative rows in residue-major order + the per-column
// maxima — not a traffic model: frame 0 and one more are checked against an fp64 FFT on the host.
//   variants: how the samples reach the threads (two-byte loads / LDS-DMA pieces), whether a frame's eight workgroups share an XCD,
//   with and without the ceiling subtraction; batches of 128 / 256 / 512 frames; a working set of 12 input and 12 output sets.
//   build: make -C scripts/ubench dif8_lab      run: gpurun -- scripts/ubench/dif8_lab [frames ...]   (DIF8_ONLY=<variant index>: that one only, for PMC passes)
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../rtl-sdr-scanner-cpp_amd/csrc/fft8192_v2.h"

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// XMAP 1: block b -> XCD b mod 8 (round-robin dispatch); the eight residues of frame f are blocks 64 (f / 8) + 8 r + (f mod 8): one XCD,
// 64 consecutive block numbers. XMAP 0: b = 8 f + r — a frame's residues on eight different XCDs.
template <int FMT, int FRONT, int XMAP>
__global__ __launch_bounds__(512, FRONT >= 3 ? 4 : 8) void k_dif8_lab(ss::Fft8192Args g, ss::Dif8Front d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int b = (int)blockIdx.x;
  int f, r;
  if (FRONT == 4 || FRONT == 6) {  // 131072 points, radix 16: eight workgroups per frame (residues r and r + 8 each; 6: the fold as a butterfly per point)
    ss::dif8_item<8>(b, d.nframes, &f, &r);
    int hdr;
    ss::fft8192_v2_frame<FMT, 2, true, false, FRONT>(g, (size_t)(16 * f + r), smem_raw, (int)threadIdx.x, &hdr, &d, (size_t)f, r);
    return;
  }
  if (FRONT == 3 || FRONT == 5) {  // four workgroups per frame (residues r and r + 4 each; 5: the fold as a butterfly per point)
    ss::dif8_item<4>(b, d.nframes, &f, &r);
  } else if (XMAP) {
    ss::dif8_item<8>(b, d.nframes, &f, &r);
  } else {
    r = b & 7;
    f = b >> 3;
  }
  int hdr;
  ss::fft8192_v2_frame<FMT, 2, true, false, FRONT>(g, (size_t)(8 * f + r), smem_raw, (int)threadIdx.x, &hdr, &d, (size_t)f, r);
}

static void fft_inplace(std::vector<std::complex<double>>& x) {  // radix-2, forward
  const size_t n = x.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(x[i], x[j]);
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const double ang = -2.0 * M_PI / (double)len;
    const std::complex<double> wl(cos(ang), sin(ang));
    for (size_t i = 0; i < n; i += len) {
      std::complex<double> w(1.0, 0.0);
      for (size_t k = 0; k < len / 2; ++k) {
        const auto u = x[i + k], v = x[i + k + len / 2] * w;
        x[i + k] = u + v;
        x[i + k + len / 2] = u - v;
        if ((k & 63) == 63) w = std::complex<double>(cos(ang * (double)(k + 1)), sin(ang * (double)(k + 1)));
        else w *= wl;
      }
    }
  }
}

struct Variant {
  const char* name;
  int front, xmap;
};

int main(int argc, char** argv) {
  std::vector<int> frame_counts;
  for (int i = 1; i < argc; ++i) frame_counts.push_back(atoi(argv[i]));
  if (frame_counts.empty()) frame_counts = {128, 256, 512, 64};
  int max_frames = 0;
  for (int f : frame_counts) max_frames = f > max_frames ? f : max_frames;
  const char* only_s = getenv("DIF8_ONLY");
  const int only = only_s ? atoi(only_s) : -1;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("# device %s, %d CUs; SS_AUX_DIF_IQ=%d (the rows leave as dB values)\n", prop.name, prop.multiProcessorCount, SS_AUX_DIF_IQ);

  const int Q = getenv("DIF_Q") ? atoi(getenv("DIF_Q")) : 8, logq = Q == 16 ? 4 : 3;  // DIF_Q=16: 131072-point frames, radix 16 (the two-residue form only)
  const int N = 8192 * Q;
  const double fs = 20e6, scale = 1.0 / 128.0;
  std::vector<float2> tw2(256), lane(384), wave(96), dt(ss::dif_table_float2(Q));
  ss::fft8192_v2_host_tables(tw2.data(), lane.data(), wave.data());
  ss::dif8_host_tables(dt.data(), scale, Q);
  float2 *d_tw2, *d_lane, *d_wave, *d_dt;
  CK(hipMalloc(&d_tw2, 256 * 8));
  CK(hipMalloc(&d_lane, 384 * 8));
  CK(hipMalloc(&d_wave, 96 * 8));
  CK(hipMalloc(&d_dt, dt.size() * 8));
  CK(hipMemcpy(d_tw2, tw2.data(), 256 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_lane, lane.data(), 384 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_wave, wave.data(), 96 * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_dt, dt.data(), dt.size() * 8, hipMemcpyHostToDevice));
  const ss::Fft8192V2Tables tabs{d_tw2, d_lane, d_wave, nullptr, nullptr};
  const float db_off = (float)(10.0 * log10(fs));

  // noise ceiling, residue-major like the rows
  std::vector<float> thr(N), thr_perm(N);
  for (int i = 0; i < N; ++i) thr[i] = -70.0f + 3.0f * (float)sin(0.001 * i);
  for (int i = 0; i < N; ++i) thr_perm[ss::dif_bin_offset(i, logq)] = thr[i];
  float* d_thr;
  CK(hipMalloc(&d_thr, N * 4));
  CK(hipMemcpy(d_thr, thr_perm.data(), N * 4, hipMemcpyHostToDevice));

  // input: int8 noise + tones; 12 sets
  const int nsets = 12;
  const size_t in_bytes = (size_t)max_frames * N * 2, out_bytes = (size_t)max_frames * N * 4, seg_floats = (size_t)4096 * 1024;
  std::vector<signed char> h_in(in_bytes);
  {
    unsigned x = 12345u;
    const auto rnd = [&]() {
      x = x * 1664525u + 1013904223u;
      return (int)((x >> 16) & 0xff) - 128;
    };
    for (int f = 0; f < max_frames; ++f)
      for (int n = 0; n < N; ++n) {
        const double ph1 = 2.0 * M_PI * (double)(12345 + 7 * f) * n / 65536.0, ph2 = 2.0 * M_PI * (double)(40001) * n / 65536.0;
        double re = 20.0 * cos(ph1) + 9.0 * cos(ph2) + 0.12 * (rnd() + rnd() + rnd() + rnd()) * 0.25;
        double im = 20.0 * sin(ph1) + 9.0 * sin(ph2) + 0.12 * (rnd() + rnd() + rnd() + rnd()) * 0.25;
        h_in[((size_t)f * N + n) * 2] = (signed char)lrint(re);
        h_in[((size_t)f * N + n) * 2 + 1] = (signed char)lrint(im);
      }
  }
  std::vector<char*> d_in(nsets);
  std::vector<float*> d_out(nsets), d_seg(nsets);
  for (int s = 0; s < nsets; ++s) {
    CK(hipMalloc((void**)&d_in[s], in_bytes));
    CK(hipMalloc((void**)&d_out[s], out_bytes));
    CK(hipMalloc((void**)&d_seg[s], seg_floats * 4));
    CK(hipMemcpy(d_in[s], h_in.data(), in_bytes, hipMemcpyHostToDevice));
  }
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));

  const Variant variants[] = {
      {"two-byte loads, a frame's residues on ONE XCD", 1, 1},
      {"LDS-DMA pieces, a frame's residues on ONE XCD", 2, 1},
      {"two-byte loads, residues over eight XCDs", 1, 0},
      {"LDS-DMA pieces, residues over eight XCDs", 2, 0},
      {"LDS-DMA pieces, TWO residues per workgroup (128 VGPRs), ONE XCD", 3, 1},
      {"131072 points, radix 16, TWO residues per workgroup, ONE XCD", 4, 1},
      {"LDS-DMA pieces, TWO residues per workgroup, radix-8 BUTTERFLY per point", 5, 1},
      {"131072 points, radix 16, TWO residues per workgroup, BUTTERFLY per point", 6, 1},
  };
  const int nvariants = 8;
  const auto launch = [&](const Variant& v, int set, int frames, hipEvent_t e0, hipEvent_t e1) {
    ss::Fft8192Args g{};
    g.tabs = tabs;
    g.db_off = db_off;
    g.scale = (float)scale;
    g.psd = d_out[set];
    ss::Dif8Front d = ss::dif8_front_of(d_in[set], (long long)N, d_dt, Q);
    d.smax = d_seg[set];
    d.smax_mask = 1023;
    d.nframes = frames;
    const dim3 grid((v.front == 3 || v.front == 5 ? 4 : 8) * frames), block(512);  // (radix 16, two residues each: eight per frame too)
#define GO(FRONT, XMAP) hipExtLaunchKernelGGL((k_dif8_lab<ss::FMT_CS8, FRONT, XMAP>), grid, block, ss::kFft8192V2LdsBytes, st, e0, e1, 0, g, d)
    if (v.front == 4) GO(4, 1);
    else if (v.front == 6) GO(6, 1);
    else if (v.front == 5) GO(5, 1);
    else if (v.front == 3) GO(3, 1);
    else if (v.front == 1 && v.xmap == 1) GO(1, 1);
    else if (v.front == 2 && v.xmap == 1) GO(2, 1);
    else if (v.front == 1 && v.xmap == 0) GO(1, 0);
    else GO(2, 0);
#undef GO
  };

  // ---- correctness: frames 0 and 77 against an fp64 transform ----
  {
    std::vector<std::vector<double>> ref;
    const int check_frames[2] = {0, max_frames > 77 ? 77 : max_frames - 1};
    for (int cf : check_frames) {
      std::vector<std::complex<double>> x(N);
      for (int n = 0; n < N; ++n) {
        const double w = 0.54 - 0.46 * cos(2.0 * M_PI * n / (double)(N - 1));
        x[n] = std::complex<double>(h_in[((size_t)cf * N + n) * 2] * scale * w, h_in[((size_t)cf * N + n) * 2 + 1] * scale * w);
      }
      fft_inplace(x);
      std::vector<double> db(N);
      for (int i = 0; i < N; ++i) db[i] = 10.0 * log10(std::norm(x[(i + N / 2) % N]) / fs);
      ref.push_back(db);
    }
    for (int vi = 0; vi < nvariants; ++vi) {
      if (only >= 0 && vi != only) continue;
      if ((Q == 16) != (variants[vi].front == 4 || variants[vi].front == 6)) continue;
      CK(hipMemset(d_out[0], 0xff, out_bytes));
      launch(variants[vi], 0, max_frames, nullptr, nullptr);
      CK(hipStreamSynchronize(st));
      std::vector<float> row(N);
      for (int c = 0; c < 2; ++c) {
        CK(hipMemcpy(row.data(), d_out[0] + (size_t)check_frames[c] * N, N * 4, hipMemcpyDeviceToHost));
        double worst = 0.0, worst_strong = 0.0, mean = 0.0;
        for (int i = 0; i < N; ++i) mean += ref[c][i];
        mean /= N;
        int worst_i = 0;
        for (int i = 0; i < N; ++i) {
          const double got = (double)row[ss::dif_bin_offset(i, logq)] ;
          const double e = fabs(got - ref[c][i]);
          if (e > worst) worst = e, worst_i = i;
          if (ref[c][i] > mean - 10.0 && e > worst_strong) worst_strong = e;
        }
        printf("check  variant %d frame %3d: max |dB - fp64| = %.3e (bin %d, ref %.2f dB), over bins above mean - 10 dB: %.3e\n", vi, check_frames[c], worst, worst_i,
               ref[c][worst_i], worst_strong);
      }
    }
  }

  // ---- timing ----
  const int iters = 48;
  std::vector<hipEvent_t> ev(2 * iters);
  for (auto& e : ev) CK(hipEventCreate(&e));
  for (int frames : frame_counts) {
    for (int vi = 0; vi < nvariants; ++vi) {
      if (only >= 0 && vi != only) continue;
      if ((Q == 16) != (variants[vi].front == 4 || variants[vi].front == 6)) continue;
      for (int k = 0; k < 12; ++k) launch(variants[vi], k % nsets, frames, nullptr, nullptr);
      CK(hipStreamSynchronize(st));
      hipEvent_t w0, w1;
      CK(hipEventCreate(&w0));
      CK(hipEventCreate(&w1));
      CK(hipEventRecord(w0, st));
      for (int k = 0; k < iters; ++k) launch(variants[vi], k % nsets, frames, ev[2 * k], ev[2 * k + 1]);
      CK(hipEventRecord(w1, st));
      CK(hipStreamSynchronize(st));
      float wall;
      CK(hipEventElapsedTime(&wall, w0, w1));
      double sum = 0.0, mn = 1e9;
      for (int k = 0; k < iters; ++k) {
        float ms;
        CK(hipEventElapsedTime(&ms, ev[2 * k], ev[2 * k + 1]));
        sum += ms;
        mn = ms < mn ? ms : mn;
      }
      const double us = 1e3 * sum / iters, samples = (double)frames * N;
      printf("%4d frames  %-50s  kernel %7.2f us (min %7.2f)  back-to-back %7.2f us per launch  %7.1f GS/s  2 B/sample: %.3f TB/s\n", frames, variants[vi].name, us, 1e3 * mn,
             1e3 * wall / iters, samples / (1e3 * wall / iters) * 1e-3, 2.0 * samples / (1e3 * wall / iters) * 1e-6);
    }
  }
  return 0;
}
