// fft1024_kernels.h — TWO-pass front end for N = 2^20 = 1024 x 1024 (BASELINE.json config 5: 1M-point FFT at 61.44 MS/s).
//
// Same contract as the kernels of fft256_kernels.h (Decimator + fft_v(Hamming, forward, shift) + PSD::work: reference
// sources/radio/blocks/decimator.h:15-22, sources/radio/sdr_device.cpp:164, sources/radio/blocks/psd.cpp:18-20). Until round 3 a
// 2^20-point frame took three passes over a 128 MiB work buffer (256-point columns, a radix-16 step, 256-point rows: 32 B per sample
// of work-buffer traffic on top of the 8 + 4 that are the job). Here it takes two:
//   n = n1 * 1024 + n2,  k = k1 + 1024 * k2
//   step A (k_fft_cols1024):      for every n2 the 1024-point FFT over n1, times W_N^(n2 k1)   -> work[k1 * 1024 + n2]
//   step B (k_fft_rows1024_psd):  for every k1 the 1024-point FFT over n2 -> X[k1 + 1024 k2]   -> dB
// What a workgroup of 8192 points can do about coalescing is bounded — C columns in and R rows out with C * R = 64 — so both
// kernels sit in the middle: the column tiles are 8 columns wide (64-byte runs of CF32 in, 64-byte runs of the work buffer out),
// the row tiles 8 rows (32-byte runs of dB values out), and the block index is decoded so that the tiles that share a 128-byte
// line run on the same XCD at about the same time (block b runs on XCD b mod 8 — observed, not promised: it only matters for speed)
// and meet in that XCD's L2.
//
// A 1024-point FFT in registers is four interleaved 256-point FFTs (fft256_passes: two radix-16 passes, one exchange) and a
// radix-4 step across them — n = 4 m + q: X[k' + 256 kap] = sum_q W_4^(q kap) W_1024^(q k') Z_q[k'] — with a second exchange
// through LDS in between, the shape of k_fft256xR_psd<., 2>. 16 points per thread, 512 threads, <= 61 VGPRs, 39-40 KiB of LDS (the
// W_256 and W_1024 tables sit there too: a table read from global memory waits behind the CU's streaming loads, fft8192_v2.h):
// four workgroups per CU, and the column tile is the FFT role of k_scan_step (KIND 3) like the 256-point column tiles of the
// other long transforms, the deferred detect / emit stages of earlier calls riding on its launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft256_kernels.h"

namespace ss {

constexpr int kFft1024Pitch2 = 261;  // second exchange of the column tiles, [sub][k']: 5 mod 32 — writers (lanes along sub) conflict-free, readers (8 columns x 4 k') two-way at worst
constexpr int kFft1024ColsPlaneBytes = 32 * kFft256PitchCols * 4;                 // 34 944 (>= 32 x 261 words)
constexpr int kFft1024RowsPlaneBytes = 1024 * 9 * 4;                              // 36 864: the read-out tile [k2][row], pitch 9 (>= 32 x 273 words)
constexpr int kFft1024TableBytes = (256 + 256) * 8;                               // W_256 [16][16], then W_1024^k' [256]
constexpr int kFft1024ColsLdsBytes = kFft1024ColsPlaneBytes + kFft1024TableBytes; // 39 040: fits k_scan_step's 39 936 (the column tiles are its FFT role, KIND 3)
// the row tiles' tables sit behind the EXCHANGE plane; the read-out tile, which is larger, overlays the front of them once the
// transform is done with them: 39 040 bytes, and the row tile fits k_scan_step's 39 936 (it is its FFT role, KIND 4)
constexpr int kFft1024RowsLdsBytes = kFft1024ColsPlaneBytes + kFft1024TableBytes;
static_assert(kFft1024RowsPlaneBytes <= kFft1024RowsLdsBytes, "the read-out tile fits");
static_assert(32 * kFft1024Pitch2 * 4 <= kFft1024ColsPlaneBytes && 32 * kFft256PitchCols * 4 <= kFft1024RowsPlaneBytes, "exchange planes");
constexpr int kFft1024TwA = 0, kFft1024TwB = 64 * 1024, kFft1024Tw1024 = 66 * 1024, kFft1024TableEntries = 66 * 1024 + 256;  // offsets into ColsArgs::twc

// Host side: the tables of both kernels in one block of kFft1024TableEntries entries (double precision, rounded once).
//   [kFft1024TwA]    [64][1024] W_N^(n2 h)                  step-A twiddle W_N^(n2 k1), k1 = h + 64 u + 256 kap, factored as
//   [kFft1024TwB]    [2][1024]  W_N^(64 n2), W_N^(256 n2)   W_N^(n2 h) * (W_N^(64 n2))^u * (W_N^(256 n2))^kap
//   [kFft1024Tw1024] [256]      W_1024^k'                   (the radix-4 step's W_1024^(q k') are its first three powers)
inline void fft1024_host_tables(float2* tab) {
  const auto W = [](double num, double den) {
    const double ang = -2.0 * 3.14159265358979323846 * num / den;
    return make_float2((float)cos(ang), (float)sin(ang));
  };
  const double n = 1048576.0;
  for (int h = 0; h < 64; ++h)
    for (int n2 = 0; n2 < 1024; ++n2) tab[kFft1024TwA + h * 1024 + n2] = W((double)n2 * h, n);
  for (int n2 = 0; n2 < 1024; ++n2) {
    tab[kFft1024TwB + n2] = W(64.0 * n2, n);
    tab[kFft1024TwB + 1024 + n2] = W(256.0 * n2, n);
  }
  for (int k = 0; k < 256; ++k) tab[kFft1024Tw1024 + k] = W((double)k, 1024.0);
}

// which columns a column tile takes: block = frame * tiles + w -> frame, first column (8-column tiles: XCD (w mod 8) takes the 16
// neighbouring tiles, so that the two tiles that share a 128-byte line are consecutive blocks of one XCD; 16-column tiles take whole lines)
__host__ __device__ inline void cols1024_block(int block, int logc, int* f, int* tile) {
  const int tiles = 1024 >> logc, w = block % tiles;
  *f = block / tiles;
  *tile = logc == 3 ? ((w & 7) << 4) | (w >> 3) : w;
}

// Host side: the window taps in the column tiles' own order — out[(tile * threads + t) * 16 + r] = the tap of the sample thread t
// of tile `tile` loads as its r-th (sample n1 = 4 (j + 16 r) + q of column tile * COLS + c; sub = t mod 4 COLS = c + COLS q,
// j = t div 4 COLS). logc = 3 (8 columns, 512 threads) or 4 (16 columns, 1024 threads).
inline void fft1024_window_order(const float* win, float* out, int logc) {
  const int cols = 1 << logc, nsub = 4 * cols, threads = 16 * nsub, tiles = 1024 / cols;
  for (int tile = 0; tile < tiles; ++tile)
    for (int t = 0; t < threads; ++t) {
      const int sub = t & (nsub - 1), j = t / nsub, c = sub & (cols - 1), q = sub >> logc;
      for (int r = 0; r < 16; ++r) out[((size_t)tile * threads + t) * 16 + r] = win[((size_t)(4 * (j + 16 * r) + q) << 10) + tile * cols + c];
    }
}

// One column tile: 8 columns x 1024 rows; `block` = frame * 128 + w. XCD (w mod 8) takes the 16 neighbouring tiles
// [16 (w mod 8), 16 (w mod 8) + 16): two tiles that share the 128-byte lines of the frame and of the work buffer are consecutive
// blocks of one XCD. `g`: ColsArgs of fft256_kernels.h with twc = the block of fft1024_host_tables (logn2 is 10).
// LOGC = 3: 8 columns by 512 threads (above). LOGC = 4: 16 columns by 1024 threads — whole 128-byte lines on either side, twice
// the LDS (two workgroups per CU: the same threads per CU); too many threads for a role of k_scan_step, so a launch of its own.
// WCALC: the Hamming taps are not loaded but formed (ss_create: only for the default window). A thread's sixteen samples are
// n = m + 65536 r with m = 1024 (4 j + q) + column < 65536, so
//   w[n] = 0.54 - 0.46 cos(theta_m + phi_r) = 0.54 + (-0.46 cos phi_r) cos theta_m + (0.46 sin phi_r) sin theta_m,   theta_m = 2 pi m / (N - 1),  phi_r = 2 pi 65536 r / (N - 1):
// one 8-byte load of (cos theta_m, sin theta_m) per thread (ColsArgs::wtab, 512 KiB) and two FMAs per sample with sixteen pairs
// of constants, instead of 64 bytes of taps per thread — 4 B/sample through L2 and the vector-memory pipe, a fifth of the tile's
// loads (12 of the column launch's 81 us per 16-frame call in the 8-column form, profiles/r04/s4_summary.txt). The taps formed
// this way differ from (float)(0.54 - 0.46 cos(2 pi n / (N - 1))) by 1.2e-7 at most (3.4e-8 rms: the size of the rounding of the
// exact tap itself), the dB plane by < 1e-5 dB but for the deepest nulls (tests/test_window_rotation.py).
__device__ constexpr float kWin1024C[16] = {-0.460000008f, -0.424984515f, -0.325268865f, -0.176033899f, 6.89093611e-07f, 0.176035181f, 0.325269848f, 0.424985051f,
                                            0.460000008f, 0.424983978f, 0.325267911f, 0.176032633f, -2.06728078e-06f, -0.176036447f, -0.325270832f, -0.424985588f};
__device__ constexpr float kWin1024S[16] = {0.0f, 0.17603454f, 0.325269371f, 0.424984783f, 0.460000008f, 0.424984246f, 0.325268388f, 0.176033258f,
                                            -1.37818722e-06f, -0.176035807f, -0.325270325f, -0.424985319f, -0.460000008f, -0.42498374f, -0.325267404f, -0.176031992f};
// Host side: ColsArgs::wtab — (cos, sin)(2 pi m / (N - 1)), m < 65536 (double precision, rounded once).
inline void fft1024_window_rotation_table(float2* out) {
  for (int m = 0; m < 65536; ++m) {
    const double th = 2.0 * 3.14159265358979323846 * (double)m / 1048575.0;
    out[m] = make_float2((float)cos(th), (float)sin(th));
  }
}

template <int FMT, int LOGC = 3, bool WCALC = false>
__device__ __forceinline__ void fft_cols1024_tile(const ColsArgs& g, int block, unsigned char* __restrict__ smem_raw, int t) {
  constexpr int COLS = 1 << LOGC, NSUB = 4 * COLS, THREADS = 16 * NSUB, TILES = 1024 / COLS;
  float* s = reinterpret_cast<float*>(smem_raw);
  float2* tw_lds = reinterpret_cast<float2*>(smem_raw + NSUB * kFft256PitchCols * 4);
  float2* tw1024_lds = tw_lds + 256;
  if (t < 512) tw_lds[t] = t < 256 ? g.tw256[t] : g.twc[kFft1024Tw1024 + t - 256];  // (tw_lds and tw1024_lds are contiguous: entries 0..511)
  int f, tile;
  cols1024_block(block, LOGC, &f, &tile);
  const int c0 = tile << LOGC;
  // tile culling: the run maxima of a frame are gathered by atomic maxima in the rows kernel, so the frame's row of the ring
  // starts from zero — cleared here, one launch earlier
  if (g.smax && t < 32768 / TILES) g.smax[((size_t)((g.abs0 + f) & g.smax_mask) << 15) + tile * (32768 / TILES) + t] = 0u;
  // after the second exchange thread t owns column c2 and the outputs k1 = h + 64 u + 256 kap: its step-A twiddles are asked for
  // first, ahead of the frame's own loads in the vector-memory queue
  const int c2 = t & (COLS - 1), h = t >> LOGC;
  const int n2b = c0 + c2;
  float2 pu = g.twc[kFft1024TwA + (h << 10) + n2b];
  const float2 t64 = g.twc[kFft1024TwB + n2b], g1 = g.twc[kFft1024TwB + 1024 + n2b];
  // first half: sub-sequence q of column c, butterfly j of its 256-point FFT — sample n1 = 4 (j + 16 r) + q of column c0 + c
  const int sub = t & (NSUB - 1), j = t >> (LOGC + 2);
  const int c = sub & (COLS - 1), q = sub >> LOGC;
  constexpr int kIn = FMT == FMT_CF32 ? 8 : 2;
  const uint32_t tn = ((uint32_t)(4 * j + q) << 10) + (uint32_t)(c0 + c);
  const __amdgpu_buffer_rsrc_t rin = buffer_of(reinterpret_cast<const char*>(g.iq) + (size_t)f * (size_t)g.item_stride * kIn, (1 << 20) * kIn);
  // The window taps come in THIS kernel's order (fft1024_window_order, built once per context from whatever taps the context has):
  // a thread's sixteen taps are 64 consecutive bytes — four 16-byte loads, whole lines per wave — where the frame's own layout
  // gives sixteen 4-byte loads in 32-byte runs (12 of the column launch's 81 us per 16-frame call, profiles/r04/s4_summary.txt).
  float wv16[16];
  float2 wt = make_float2(0.0f, 0.0f);
  if constexpr (WCALC) {
    wt = buffer_load_f2<0>(buffer_of(g.wtab, 65536 * 8), (int)(tn * 8u), 0);
  } else {
    const __amdgpu_buffer_rsrc_t rwin = buffer_of(g.win, (1 << 20) * 4);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      const auto w4 = __builtin_amdgcn_raw_buffer_load_b128(rwin, (tile * THREADS + t) * 64, 16 * r4, 0);
      wv16[4 * r4] = __uint_as_float(w4[0]);
      wv16[4 * r4 + 1] = __uint_as_float(w4[1]);
      wv16[4 * r4 + 2] = __uint_as_float(w4[2]);
      wv16[4 * r4 + 3] = __uint_as_float(w4[3]);
    }
  }
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float2 x;
    if constexpr (FMT == FMT_CF32) {
      x = buffer_load_f2<2>(rin, (int)(tn * 8u), r * (65536 * 8));
    } else {
      const unsigned short raw = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, (int)(tn * 2u), r * (65536 * 2), 2);
      if constexpr (FMT == FMT_CS8) x = make_float2((float)(signed char)(raw & 0xff) * g.scale, (float)(signed char)(raw >> 8) * g.scale);
      else x = make_float2(((float)(raw & 0xff) - 127.5f) * g.scale, ((float)(raw >> 8) - 127.5f) * g.scale);
    }
#ifndef SS_C1024_ABL  // (A/B builds, scripts/build_ab.py — garbage results, the column tiles' time without: 1 = the window loads, 2 = the work-buffer stores, 4 = the frame loads)
#define SS_C1024_ABL 0
#endif
    if (SS_C1024_ABL & 4) x = make_float2(__int_as_float(0x3f800000 + (int)tn + r), 0.5f);
    const float wv = (SS_C1024_ABL & 1) ? 1.0f : WCALC ? fmaf(wt.x, kWin1024C[r], fmaf(wt.y, kWin1024S[r], 0.54f)) : wv16[r];
    a[r] = make_float2(x.x * wv, x.y * wv);  // volk_32fc_32f_multiply_32fc
  }
  float2 cc[16];
  fft256_passes<kFft256PitchCols>(a, cc, s, tw_lds, sub, j);  // Z_q[j + 16 k] of column c in cc[slot16(k)]
  __syncthreads();
  // second exchange: Z_q[k'] of column c to word (c + 8 q) * 261 + k'; thread (c2, h) reads its four q's for k' = h + 64 u
  float* zp = s + sub * kFft1024Pitch2 + j;
  const float* zr = s + c2 * kFft1024Pitch2 + h;
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = cc[slot16(k)].x;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) a[4 * u + qq].x = zr[COLS * qq * kFft1024Pitch2 + 64 * u];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = cc[slot16(k)].y;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const float2 w1 = tw1024_lds[h + 64 * u], w2 = cmul(w1, w1), w3 = cmul(w2, w1);  // W_1024^(q k'), q = 1, 2, 3
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const float2 v = make_float2(a[4 * u + qq].x, zr[COLS * qq * kFft1024Pitch2 + 64 * u]);
      a[4 * u + qq] = qq == 0 ? v : cmul(v, qq == 1 ? w1 : qq == 2 ? w2 : w3);
    }
    dft4(a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]);  // Y[h + 64 u + 256 kap] in a[4 u + kap]
  }
  // step-A twiddle W_N^(n2 k1) and out: work[k1 * 1024 + n2], lanes along the 8 columns (64-byte runs)
  const float2 g2 = cmul(g1, g1), g3 = cmul(g2, g1);
  const __amdgpu_buffer_rsrc_t rw = buffer_of(g.work + ((size_t)f << 20), (1 << 20) * 8);
  const int voff = ((h << 10) + n2b) * 8;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
#pragma unroll
    for (int kap = 0; kap < 4; ++kap) {
      const float2 tw = kap == 0 ? pu : cmul(pu, kap == 1 ? g1 : kap == 2 ? g2 : g3);
      const float2 y = cmul(a[4 * u + kap], tw);
      if ((SS_C1024_ABL & 2) && y.x != 12345.678f) continue;
      __builtin_amdgcn_raw_buffer_store_b64(*reinterpret_cast<const __attribute__((ext_vector_type(2))) unsigned*>(&y), rw, voff, ((64 * u + 256 * kap) << 10) * 8, SS_AUX_WORK);
    }
    if (u < 3) pu = cmul(pu, t64);
  }
}

// Stand-alone launch (contexts without the step kernel; and the 16-column form).
template <int FMT, int LOGC = 3, bool WCALC = false>
__global__ __launch_bounds__(64 << LOGC, 8) void k_fft_cols1024(ColsArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fft_cols1024_tile<FMT, LOGC, WCALC>(g, (int)blockIdx.x, smem_raw, (int)threadIdx.x);
}
constexpr int fft1024_cols_lds_bytes(int logc) { return (4 << logc) * kFft256PitchCols * 4 + kFft1024TableBytes; }
static_assert(fft1024_cols_lds_bytes(3) == kFft1024ColsLdsBytes && (4 << 4) * kFft1024Pitch2 <= (4 << 4) * kFft256PitchCols, "column tile LDS");

// An order-preserving key for atomic maxima of dB values: larger float <-> larger unsigned; 0 = nothing seen (below -inf),
// 0xffffffff = NaN (wins: "cannot be bounded").
__host__ __device__ __forceinline__ unsigned max_key(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  return x != x ? 0xffffffffu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}
__host__ __device__ __forceinline__ float max_key_value(unsigned key) {
  if (key == 0u) return -__builtin_inff();
  if (key == 0xffffffffu) return __builtin_nanf("");
  return __builtin_bit_cast(float, (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}
// which rows a row tile takes: block = ((f * 16) + v) * 8 + x -> frame f, first row r0 (XCD x takes rows [128 x, 128 x + 128) of
// every frame, v its 16 tiles in turn: the four tiles that fill the 128-byte lines of the dB plane are consecutive blocks of one XCD)
// (LOGN1 = 8 — 262144 points as 256 columns x 1024-point rows, round 6: 32 tiles per frame, XCD x takes rows [32 x, 32 x + 32), its four
// tiles — the ones that fill the 128-byte lines — in turn)
template <int LOGN1 = 10>
__host__ __device__ inline void rows1024_block(int block, int* f, int* r0) {
  static_assert(LOGN1 == 10 || LOGN1 == 8, "1024 or 256 rows of 1024 points");
  const int xc = block & 7, v = block >> 3;
  *f = v >> (LOGN1 - 6);
  *r0 = (xc << (LOGN1 - 3)) + ((v & ((1 << (LOGN1 - 6)) - 1)) << 3);
}
// 262144 points: where the maximum of 32-bin run `run` (0..7: rows k1 in [32 run, 32 run + 32)) of tile column `col` (output order:
// k2 ^ 512, DC in the middle) lies in a frame's 8192 words of the ring — [run][col], max_key values gathered by atomic maxima (four
// row tiles share a run; the column tiles clear the frame's words one launch earlier, fft256_kernels.h). PlanLongArgs::layout 3.
__host__ __device__ inline int rows1024x256_smax_index(int run, int col) { return (run << 10) | col; }
// where the maximum of 32-bin run R (= bin >> 5, output order: DC in the middle) lies in a frame's row of the ring: the rows
// kernel's lanes run along k2, so [k1 group][k2] makes its atomics contiguous
__host__ __device__ inline int rows1024_smax_index(int run) { return ((run & 31) << 10) | (run >> 5); }

struct Rows1024Args {
  const float2* work;
  const float2* tw256;
  const float2* tw1024;  // W_1024^k' [256]
  float db_off;
  float* psd;  // null: no dB plane wanted (detect mode with calls shorter than the averager ring: the ring rows are all there is)
  RowsExtra x; // (smax holds max_key values here)
};

// One row tile: 8 rows k1 x 1024 points. block = ((f * 16) + v) * 8 + x: XCD x takes rows [128 x, 128 x + 128) of every
// frame, v its 16 tiles in turn — the four tiles that fill the 128-byte lines of the dB plane (32 consecutive k1 for every k2)
// are consecutive blocks of one XCD.
// LOGN1: 10 = a 2^20-point frame (1024 rows); 8 = a 262144-point frame behind 256-point column tiles (fft256_kernels.h): 256 rows of
// 1024 points, bin k1 + 256 k2 — the size getFft picks at 61.44 MS/s (utils/radio_utils.cpp:98-104), which until round 6 went through
// k_fft_rows256xR_psd: no run maxima, no ring rows, a dB plane always written, every averaging tile evaluated.
template <int LOGN1 = 10>
__device__ __forceinline__ void fft_rows1024_tile(const Rows1024Args& g, int block, unsigned char* __restrict__ smem_raw, int t) {
  constexpr int LOGN = LOGN1 + 10;
  float* s = reinterpret_cast<float*>(smem_raw);
  float2* tw_lds = reinterpret_cast<float2*>(smem_raw + kFft1024ColsPlaneBytes);
  float2* tw1024_lds = tw_lds + 256;
  tw_lds[t] = t < 256 ? g.tw256[t] : g.tw1024[t - 256];
  const RowsExtra& x = g.x;
  if (x.zero_word && block == 0 && t == 0) *x.zero_word = 0;
  int f, r0;
  rows1024_block<LOGN1>(block, &f, &r0);
  const int fl = t >> 6, tt = t & 63;
  const int q = tt & 3, j = tt >> 2;
  const float2* row = g.work + ((size_t)f << LOGN) + ((size_t)(r0 + fl) << 10);
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = row[tt + 64 * r];
  float2 c[16];
  const int sub = (fl << 2) | q;
  fft256_passes<kFft256PitchCols>(a, c, s, tw_lds, sub, j);
  __syncthreads();
  float* zp = s + sub * 257 + j;
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = c[slot16(k)].x;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = t + 512 * u;
    const float* zr = s + ((p >> 8) << 2) * 257 + (p & 255);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) a[u * 4 + qq].x = zr[qq * 257];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = c[slot16(k)].y;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = t + 512 * u;
    const int kp = p & 255;
    const float* zr = s + ((p >> 8) << 2) * 257 + kp;
    const float2 w1 = tw1024_lds[kp], w2 = cmul(w1, w1), w3 = cmul(w2, w1);  // W_1024^(q k'), q = 1, 2, 3
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const float2 vv = make_float2(a[u * 4 + qq].x, zr[qq * 257]);
      a[u * 4 + qq] = qq == 0 ? vv : cmul(vv, qq == 1 ? w1 : qq == 2 ? w2 : w3);
    }
    dft4(a[4 * u], a[4 * u + 1], a[4 * u + 2], a[4 * u + 3]);  // X_row[kp + 256 kap] in a[4 u + kap]
  }
  float dbv[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) dbv[i] = psd_db(a[i], g.db_off);
  __syncthreads();  // every Z read is done before the plane is reused for the read-out
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int p = t + 512 * u;
#pragma unroll
    for (int kap = 0; kap < 4; ++kap) s[((p & 255) + 256 * kap) * 9 + (p >> 8)] = dbv[u * 4 + kap];
  }
  __syncthreads();
  constexpr int half = 1 << (LOGN - 1);
  if (x.smax) {  // the largest dB value of this tile's 8 rows for every k2: a quarter of the 32-bin run (r0 / 32, k2)
    unsigned* srow = reinterpret_cast<unsigned*>(x.smax) + ((size_t)((x.abs0 + f) & x.smax_mask) << (LOGN - 5));  // (a word per 32-bin run)
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int k2 = t + 512 * e;
      float m = s[k2 * 9];
      bool bad = m != m;
#pragma unroll
      for (int r = 1; r < 8; ++r) {
        const float vv = s[k2 * 9 + r];
        bad = bad || (vv != vv);
        m = fmaxf(m, vv);
      }
      if constexpr (LOGN1 == 10) atomicMax(&srow[rows1024_smax_index((((r0 + (k2 << 10)) ^ half) >> 5))], bad ? 0xffffffffu : max_key(m));
      else atomicMax(&srow[rows1024x256_smax_index(r0 >> 5, k2 ^ 512)], bad ? 0xffffffffu : max_key(m));
    }
  }
  float* out = g.psd ? g.psd + ((size_t)f << LOGN) : nullptr;
  float* hrow = (x.hist_out && f >= x.first_hist) ? x.hist_out + ((size_t)(f - x.first_hist) << LOGN) : nullptr;  // (workgroup-uniform)
  const int rr = t & 7, kb = t >> 3;
  // bin of output i: ((r0 + rr) + ((kb + 64 i) << LOGN1)) ^ half — fft_v shift=true: X[k] lands at k ^ (N/2)
  const int bin0 = ((r0 + rr) + (kb << LOGN1)) ^ half;
  constexpr int ISH = LOGN1 + 6;  // (64 k2 further on: 64 << LOGN1 bins)
  if (out) {
#pragma unroll
    for (int i = 0; i < 16; ++i) store_f1_policy<SS_AUX_ROWS>(&out[bin0 ^ (i << ISH)], s[(kb + 64 * i) * 9 + rr]);
  }
  if (hrow) {
    // The sixteen ceiling values FIRST, all in flight together, then the sixteen stores. (Until session 16 of round 4 the loop was
    // load, subtract, store per output — and since nothing tells the compiler that the ceiling and the ring are different memory it
    // kept that order: every store waited for its own load AND, vmcnt counting both, for the store before it: sixteen memory round
    // trips one after the other at the end of every workgroup, ~20 of its ~25 us.)
    // (round 5: detect-mode calls hand in no ceiling — x.thr null, workgroup-uniform — and the rows leave as dB values: the sixteen ceiling
    // loads per thread, 4 B/sample through L2, cost the launch 4-5 of its 45 us, profiles/r05/s18_summary.txt; the tiles that are
    // evaluated subtract the ceiling, DetectArgs::ring_db_from)
    if (!x.thr) {
#pragma unroll
      for (int i = 0; i < 16; ++i) store_f1_policy<SS_AUX_ROWS>(&hrow[bin0 ^ (i << ISH)], s[(kb + 64 * i) * 9 + rr]);
    } else {
      float th[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) th[i] = x.thr[bin0 ^ (i << ISH)];
#pragma unroll
      for (int i = 0; i < 16; ++i) store_f1_policy<SS_AUX_ROWS>(&hrow[bin0 ^ (i << ISH)], s[(kb + 64 * i) * 9 + rr] - th[i]);  // noise_learner.cpp:55, as detect_tile forms it
    }
  }
}

// Stand-alone launch (contexts without the step kernel; otherwise the row tiles run as a role of k_scan_step, scan_step.h).
template <int LOGN1 = 10>
__global__ __launch_bounds__(512, 8) void k_fft_rows1024_psd(Rows1024Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fft_rows1024_tile<LOGN1>(g, (int)blockIdx.x, smem_raw, (int)threadIdx.x);
}

}  // namespace ss
