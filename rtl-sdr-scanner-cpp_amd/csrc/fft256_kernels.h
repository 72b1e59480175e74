// fft256_kernels.h — four-step front end for N >= 65536 built on 256-point FFTs held in registers
// (BASELINE.json configs 3 and 5: 65536 points at 20 MS/s, 2^20 points at 61.44 MS/s).
//
// Same contract as k_fft_cols / k_fft_rows_psd in fft_kernels.h (Decimator + fft_v(Hamming, forward, shift) + PSD::work:
// reference sources/radio/blocks/decimator.h:15-22, sources/radio/sdr_device.cpp:164, sources/radio/blocks/psd.cpp:18-20):
//   N = 256 * N2,  n = n1 * N2 + n2,  k = k1 + 256 * k2
//   step A (k_fft_cols256):  for every n2 the 256-point FFT over n1, times W_N^(n2 k1)  -> work[k1 * N2 + n2]
//   step B:  for every k1 the N2-point FFT over n2 -> X[k1 + 256 k2] -> dB
//            N2 = 256: k_fft_rows256_psd below;  other N2: the generic k_fft_rows_psd<8, log2 N2>
//
// A 256-point FFT is two radix-16 passes (the 16-point DFT of fft8192_kernel.h) with ONE exchange, done one fp32 plane
// at a time through a 17-word pitch per thread exactly like the first exchange of the 8192-point kernel: the generic
// kernels make four radix-4 trips of the whole tile through LDS with two barriers each.
//
// k_fft_cols256: workgroup = 512 threads = 32 columns x 256 rows = 8192 points. Thread (q = t % 32, j = t / 32): column
// n2 = c0 + q, rows n1 = j + 16 r. Lanes run along n2, so every global load and every store of a wave is two 256-byte
// runs; no transposition is needed on the way out (work[k1 * N2 + n2] is n2-fastest too).
//   pass 1: Y[16 j + k] = DFT16_r( x[j + 16 r] )                    -> LDS word q * 273 + 17 j + k
//   pass 2: X[j + 16 k] = DFT16_r'( Y[j + 16 r'] W_256^(r' j) )     <- LDS word q * 273 + 17 r' + j
// k_fft_rows256_psd: workgroup = 512 threads = 32 rows (k1) x 256 points. Thread (rho = t / 16, j = t % 16): lanes run
// along n2 inside a row (work is n2-fastest). The dB values go through LDS once more so that the store runs along k1,
// the fastest index of the output bin k1 + 256 k2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft8192_kernel.h"
#include "fft_kernels.h"

namespace ss {

// words per 256-point FFT in the exchange plane (16 threads x 17-word pitch = 272): the columns kernel, whose lanes run
// over 32 different FFTs, adds one so that the FFTs start on different banks; the rows kernel, whose 16-lane groups
// belong to 4 FFTs, keeps 272 = 16 mod 32 so that consecutive FFTs use complementary bank halves
constexpr int kFft256PitchCols = 273, kFft256PitchRows = 272;
constexpr int kFft256LdsBytes = 32 * kFft256PitchCols * 4;  // one fp32 plane of 32 FFTs: 34 944 bytes
constexpr int kFft256ColsLdsBytes = kFft256LdsBytes + 256 * 8;  // the column tiles keep W_256 behind the plane (below)
constexpr int kFft256RowsLdsBytes = 256 * 33 * 4;           // rows kernel read-out plane [k2][rho], 33-word pitch: 33 792 bytes

// the two register passes of 32 independent 256-point FFTs; `fft` = which of the 32, `j` = butterfly 0..15
template <int PITCH>
__device__ __forceinline__ void fft256_passes(float2 (&a)[16], float2 (&c)[16], float* __restrict__ s, const float2* __restrict__ tw256, int fft,
                                              int j) {
  dft16(a);
  float* plane = s + fft * PITCH;
#pragma unroll
  for (int k = 0; k < 16; ++k) plane[17 * j + k] = a[slot16(k)].x;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r].x = plane[17 * r + j];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) plane[17 * j + k] = a[slot16(k)].y;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float2 v = make_float2(c[r].x, plane[17 * r + j]);
    c[r] = r == 0 ? v : cmul(v, tw256[r * 16 + j]);  // W_256^(r j), applied as the second plane arrives
  }
  dft16(c);  // X[j + 16 k] in c[slot16(k)]
}

// Cache policy of the long transforms' big stores — the column tiles' work buffer (SS_AUX_WORK) and the row tiles' dB / ring rows
// (SS_AUX_ROWS): 0 = the default write-back policy (ships), 16 = sc1, write-through, as the 8192-point kernel's dB rows (fft8192_v2.h
// SS_AUX_PSD), 2 = nt. Measured in round 6 (profiles/r06/s5_summary.txt, alternating runs): write-through LOSES here — the row tiles
// store 32-byte runs that four workgroups of one XCD complete to 128-byte lines in their L2, and written through every piece goes to
// memory by itself: row launch 22.6 -> 32.6 us per 32 frames of 262144 points, 43 -> 60 us per 16 frames of 2^20; the work buffer
// (whole lines) a wash: column launch 35.0 -> 33.6 us at 262144 points, 57.7 -> 59.4 at 2^20. Build-time switches (scripts/build_ab.py).
#ifndef SS_AUX_WORK
#define SS_AUX_WORK 0
#endif
#ifndef SS_AUX_ROWS
#define SS_AUX_ROWS 0
#endif
template <int AUX>
__device__ __forceinline__ void store_f1_policy(float* p, float v) {
  if constexpr (AUX == 16) asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  else if constexpr (AUX == 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}
template <int AUX>
__device__ __forceinline__ void store_f2_policy(float2* p, float2 v) {
  typedef float f2v __attribute__((ext_vector_type(2)));
  const f2v vv = {v.x, v.y};
  if constexpr (AUX == 16) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(vv) : "memory");
  else if constexpr (AUX == 2) __builtin_nontemporal_store(vv, reinterpret_cast<f2v*>(p));
  else *p = v;
}

struct ColsArgs {
  const void* iq;
  long long item_stride;
  const float* win;
  const float2* tw256;
  const float2* twc;
  float scale;
  float2* work;
  int logn2;
  // N = 1024 x 1024 (fft1024_kernels.h): twc is the block of fft1024_host_tables; with tile culling the column tiles clear the
  // frame's row of the run-maxima ring (null: nothing to clear)
  unsigned* smax;
  int smax_mask, abs0;
  const float2* wtab;  // 1024-point column tiles that form their Hamming taps (fft1024_kernels.h, WCALC): (cos, sin)(2 pi m / (N - 1)), m < 65536
};

// N = 65536, default window: the column tiles form their Hamming taps instead of loading them (the 2^20-point tiles' way,
// fft1024_kernels.h): a thread's sixteen samples are n = m + 4096 r with m = 256 j + n2 < 4096, so
//   w[n] = 0.54 + (-0.46 cos phi_r) cos theta_m + (0.46 sin phi_r) sin theta_m,   theta_m = 2 pi m / 65535,  phi_r = 2 pi 4096 r / 65535
// — one 8-byte table entry per thread (ColsArgs::wtab: 4096 entries) and two FMAs per sample in the place of sixteen 4-byte loads
// from a 256 KiB table: with int8 IQ the taps were two thirds of the bytes the tile loads. Within 1.2e-7 of the table's taps
// (3.4e-8 rms), checked on the CPU (tests/host/index_check.cpp).
__device__ constexpr float kWin65536C[16] = {-0.460000008f, -0.424983531f, -0.325265229f, -0.176026732f, 1.10256551e-05f, 0.176047117f, 0.325280815f, 0.424991965f,
                                             0.460000008f, 0.424975097f, 0.325249642f, 0.176006362f, -3.30769653e-05f, -0.176067486f, -0.325296402f, -0.425000399f};
__device__ constexpr float kWin65536S[16] = {0.0f, 0.176036924f, 0.325273007f, 0.424987763f, 0.460000008f, 0.424979299f, 0.325257421f, 0.176016554f,
                                             -2.20513102e-05f, -0.176057294f, -0.325288624f, -0.424996197f, -0.460000008f, -0.424970865f, -0.325241834f, -0.175996184f};
inline void fft65536_window_rotation_table(float2* out) {  // host side: ColsArgs::wtab for N = 65536
  for (int m = 0; m < 4096; ++m) {
    const double th = 2.0 * 3.14159265358979323846 * (double)m / 65535.0;
    out[m] = make_float2((float)cos(th), (float)sin(th));
  }
}

// One column tile (32 columns x 256 rows) by one workgroup of 512 threads; `block` = frame * (N2 / 32) + tile.
template <int FMT>
__device__ __forceinline__ void fft_cols256_tile(const ColsArgs& g, int block, unsigned char* __restrict__ smem_raw, int t) {
  float* s = reinterpret_cast<float*>(smem_raw);
  // W_256 (2 KiB) into LDS before anything else: read from global memory in the second pass it would wait behind the tile's
  // own streaming loads and those of the CU's other workgroups (in-order vector memory: fft8192_v2.h). The barriers of the
  // first exchange come before its first use.
  float2* tw_lds = reinterpret_cast<float2*>(smem_raw + kFft256LdsBytes);
  if (t < 256) tw_lds[t] = g.tw256[t];
#ifndef SS_COLS_STAGGER  // (A/B builds, scripts/build_ab.py: every other column tile starts SS_COLS_STAGGER x ~3.4 us late — do a launch's load, compute and store phases overlap better out of step?)
#define SS_COLS_STAGGER 0
#endif
  if (SS_COLS_STAGGER && (block & 1)) {
#pragma unroll
    for (int k = 0; k < SS_COLS_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
  }
  const int logn2 = g.logn2;
  const int q = t & 31, j = t >> 5;
  const int n2size = 1 << logn2;
  const int tiles_per_frame = n2size >> 5;
  const int f = block / tiles_per_frame;
  const int n2 = ((block % tiles_per_frame) << 5) + q;
  // tile culling with run maxima gathered by atomic maxima in the rows kernel (262144 points, fft1024_kernels.h: fft_rows1024_tile<8>):
  // the frame's words of the ring start from zero — cleared here, one launch earlier, 256 of its N / 32 words per column tile
  if (g.smax && t < 256) g.smax[(size_t)((g.abs0 + f) & g.smax_mask) * (size_t)(tiles_per_frame << 8) + ((block % tiles_per_frame) << 8) + t] = 0u;
  // sample n = (j + 16 r) N2 + n2: block-uniform part (frame, 16 r N2) in scalar registers, per-thread part one 32-bit offset
  constexpr uint32_t kInBytes = FMT == FMT_CF32 ? 8u : 2u;
  const char* in_frame = reinterpret_cast<const char*>(g.iq) + (size_t)f * (size_t)g.item_stride * kInBytes;
  const char* win_b = reinterpret_cast<const char*>(g.win);
  const uint32_t tn = ((uint32_t)j << logn2) + (uint32_t)n2;
  float2 a[16];
  if (g.wtab && logn2 == 8) {  // (workgroup-uniform; a loop of its own: a branch around every tap load would cost the loads their interleaving)
    const float2 wt = g.wtab[tn];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const size_t un = (size_t)(16 * r) << 8;
      const float2 x = load_iq<FMT>(in_frame + un * kInBytes + tn * kInBytes, 0, g.scale);
      const float w = fmaf(wt.x, kWin65536C[r], fmaf(wt.y, kWin65536S[r], 0.54f));
      a[r] = make_float2(x.x * w, x.y * w);
    }
  } else
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const size_t un = (size_t)(16 * r) << logn2;
    const float2 x = load_iq<FMT>(in_frame + un * kInBytes + tn * kInBytes, 0, g.scale);
#ifndef SS_COLS_ABL  // (A/B builds, scripts/build_ab.py — garbage results, the column tiles' time without: 1 = the step-A twiddle table loads, 2 = the window loads, 3 = both)
#define SS_COLS_ABL 0
#endif
    const float w = (SS_COLS_ABL & 2) ? 1.0f : *reinterpret_cast<const float*>(win_b + un * 4 + tn * 4u);
    a[r] = make_float2(x.x * w, x.y * w);  // volk_32fc_32f_multiply_32fc
  }
  float2 c[16];
  fft256_passes<kFft256PitchCols>(a, c, s, tw_lds, q, j);
  // step-A twiddle W_N^(n2 k1), k1 = j + 16 k, as W_N^(n2 j) * W_N^(16 n2 k) from two tables laid out [j][n2] and [k][n2]:
  // the lanes of a wave (consecutive n2) read consecutive entries (a gather from the N-entry table W_N^m at m = n2 k1
  // costs as much as the whole transform). work[k1 * N2 + n2]: block-uniform base + one 32-bit offset per access.
  char* wf = reinterpret_cast<char*>(g.work + ((size_t)f << (8 + logn2)));
  const char* t1 = reinterpret_cast<const char*>(g.twc);
  const char* t2 = reinterpret_cast<const char*>(g.twc + ((size_t)16 << logn2));
  const float2 tj = *reinterpret_cast<const float2*>(t1 + 8u * (((uint32_t)j << logn2) + (uint32_t)n2));
#ifndef SS_COLS_TW6  // (A/B builds: 0 = one table entry per k, fifteen loads per thread, as until the end of round 3)
#define SS_COLS_TW6 1
#endif
#if SS_COLS_TW6
  // W_N^(16 n2 k) = w^k, w = W_N^(16 n2), from SIX entries of the [k][n2] table — w, w^2, w^3 and w^4, w^8, w^12 — as
  // w^(4 a + b) = w^(4 a) w^b (one more rounding than the table's own entry: inside the fp32 FFT's floor, §5 of DESIGN.md). The
  // fifteen loads per thread stood, like every table read of a streaming workgroup, behind the tile's and its neighbours' frame
  // loads: without them the column launch ran 5 us (65536 x 128) and 14 us (2^20 x 16) shorter (profiles/r03/s50_summary.txt).
  const auto tw_at = [&](int k) { return *reinterpret_cast<const float2*>(t2 + 8u * (((uint32_t)k << logn2) + (uint32_t)n2)); };
  const float2 wb1 = tw_at(1), wb2 = tw_at(2), wb3 = tw_at(3), wa1 = tw_at(4), wa2 = tw_at(8), wa3 = tw_at(12);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t k1 = (uint32_t)(j + 16 * k);
    float2 y = cmul(c[slot16(k)], tj);
    if (k > 0) {
      const int a = k >> 2, b = k & 3;
      const float2 wa = a == 1 ? wa1 : a == 2 ? wa2 : wa3, wb = b == 1 ? wb1 : b == 2 ? wb2 : wb3;
      y = cmul(y, a == 0 ? wb : b == 0 ? wa : cmul(wa, wb));
    }
    store_f2_policy<SS_AUX_WORK>(reinterpret_cast<float2*>(wf + 8u * ((k1 << logn2) + (uint32_t)n2)), y);
  }
#else
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const uint32_t k1 = (uint32_t)(j + 16 * k);
    float2 y = cmul(c[slot16(k)], tj);
    if (k > 0 && !(SS_COLS_ABL & 1)) y = cmul(y, *reinterpret_cast<const float2*>(t2 + 8u * (((uint32_t)k << logn2) + (uint32_t)n2)));
    *reinterpret_cast<float2*>(wf + 8u * ((k1 << logn2) + (uint32_t)n2)) = y;
  }
#endif
}

// Stand-alone launch (learning batches are launched this way too; otherwise the tiles run as a role of k_scan_step).
template <int FMT>
__global__ __launch_bounds__(512, 8) void k_fft_cols256(ColsArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  fft_cols256_tile<FMT>(g, (int)blockIdx.x, smem_raw, (int)threadIdx.x);
}

// What the rows kernel leaves for the detect stage of a long transform besides the dB rows (tile culling for N = 256 x N2,
// detect_fused.h: k_plan_long). Every workgroup owns one 32-bin run of each of 256 tile columns:
//   smax      the largest dB value of each run -> smax[(abs0 + frame) & smax_mask][rows_smax_index(...)], a ring over the frames
//             since the last reset; within a frame the runs lie in the order the workgroups produce them (256 consecutive floats
//             per workgroup: as 4-byte stores scattered over the row in bin order they cost a 32-byte write each, 1 B per
//             sample). Taken in the power domain (the largest re^2 + im^2 of the run through the same dB formula as the
//             bins themselves: the hardware log2 is monotonic to within an ulp, the culling margin is 0.06 dB), by LDS atomics on
//             the bit patterns of the non-negative powers; a NaN wins and is taken for "cannot be bounded" by the plan.
//   hist_out  the averager ring (rel = dB - thr, what detect_tile stores for the newest H frames of a batch): written here when
//             the call has no learning frames — thr is then what the call's detect stage will subtract, the same fp32
//             subtraction on the same values — so that a detect tile that cannot hold a candidate has nothing left to do.
struct RowsExtra {
  float* smax;       // null: no maxima
  int smax_mask;     // ring rows - 1 (a power of two)
  int abs0;          // frames since the last reset before this batch
  const float* thr;  // the noise ceiling (hist_out != null): the rows are written noise-relative; null: as dB values (DetectArgs::ring_db_from)
  float* hist_out;   // null: the detect stage writes the ring
  int first_hist;    // batch frames >= first_hist become ring rows [frame - first_hist]
  int* zero_word;    // set to zero by the first workgroup: the count of the list k_plan_long appends to right behind this launch (or null)
};
constexpr int kFft256RowsPsdLdsBytes = kFft256ColsLdsBytes + 256 * 4;  // + one word per tile column for the run maxima
// Where the maximum of run `run` (0..7) of tile column `col` (256 bins, in output order: DC in the middle) lies in a frame's row
// of RowsExtra::smax: the workgroup (c, r0 = 32 run) of k_fft_rows256_psd holds, for d = 0..255, the run of the column whose
// unshifted number is c + nsub d — [run][c][d], d fastest.
__host__ __device__ inline int rows_smax_index(int col, int run, int logn) {
  const int cols = 1 << (logn - 8), lognsub = logn - 16;
  const int cx = col ^ (cols >> 1);
  return run * cols + ((cx & ((1 << lognsub) - 1)) << 8) + (cx >> lognsub);
}

// Rows of 256 points spaced row_stride apart: N2 = 256 (row_stride 256, nsub 1) directly after the columns pass, or
// N2 = 256 A after k_fft_sub_dft (row_stride N2, nsub = A sub-rows c per row). Output bin of X_row[d] is
// k1 + 256 c + 256 nsub d; a workgroup takes 32 consecutive k1 of one c so that stores run along k1.
struct Rows256Args {
  const float2* work;
  const float2* tw256;
  float db_off;
  float* psd;  // null: a call that keeps no dB plane
  int logn, lognsub;
  RowsExtra x;
};
// One row tile: 32 rows k1 x 256 points of one sub-row c of one frame; `block` = ((f * nsub) + c) * 8 + k1 tile. A launch of its own
// (k_fft_rows256_psd) or the FFT role of k_scan_step (KIND 6, scan_step.h: 65536 points with tile culling — the column half is
// then a launch of its own right before, as at 2^20 points).
__device__ __forceinline__ void fft_rows256_tile(const Rows256Args& g, int block, unsigned char* __restrict__ smem_raw, int t) {
  const float2* __restrict__ work = g.work;
  const float2* __restrict__ tw256 = g.tw256;
  const float db_off = g.db_off;
  float* __restrict__ psd = g.psd;
  const int logn = g.logn, lognsub = g.lognsub;
  const RowsExtra& x = g.x;
  float* s = reinterpret_cast<float*>(smem_raw);
  float2* tw_lds = reinterpret_cast<float2*>(smem_raw + kFft256LdsBytes);  // W_256 in LDS, as in fft_cols256_tile
  unsigned* pmax = reinterpret_cast<unsigned*>(smem_raw + kFft256ColsLdsBytes);
  if (t < 256) tw_lds[t] = tw256[t];
  if (x.smax && t < 256) pmax[t] = 0u;  // (the barriers of the register passes come before the first atomic)
  if (x.zero_word && block == 0 && t == 0) *x.zero_word = 0;
#ifndef SS_ROWS_STAGGER  // (A/B builds: as SS_COLS_STAGGER, for the row tiles)
#define SS_ROWS_STAGGER 0
#endif
  if (SS_ROWS_STAGGER && (block & 8)) {
#pragma unroll
    for (int k = 0; k < SS_ROWS_STAGGER; ++k) __builtin_amdgcn_s_sleep(127);
  }
  const int rho = t >> 4, j = t & 15;
  const int r0 = (block & 7) << 5;
  const int c = (block >> 3) & ((1 << lognsub) - 1);
  const int f = block >> (3 + lognsub);
  const int log_row = 8 + lognsub;  // log2 of the row stride
  const float2* row = work + ((size_t)f << logn) + ((size_t)(r0 + rho) << log_row) + ((size_t)c << 8);
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = row[j + 16 * r];
  float2 cc[16];
  fft256_passes<kFft256PitchRows>(a, cc, s, tw_lds, rho, j);
#ifndef SS_ROWS_ABL  // (A/B builds, scripts/build_ab.py: 1 = no maxima, 2 = no ring rows — garbage results, the kernel's time without them)
#define SS_ROWS_ABL 0
#endif
  if (x.smax && SS_ROWS_ABL != 1) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float2 v = cc[slot16(k)];
      atomicMax(&pmax[j + 16 * k], __float_as_uint(fmaf(v.x, v.x, v.y * v.y)));  // psd_db's own power, >= +0 or NaN
    }
  }
  __syncthreads();  // the exchange plane is reused for the read-out
  // dB values to LDS at [d][rho] (33-word pitch), then out along k1
#pragma unroll
  for (int k = 0; k < 16; ++k) s[(j + 16 * k) * 33 + rho] = psd_db(cc[slot16(k)], db_off);
  __syncthreads();
  float* out = psd + ((size_t)f << logn);
  const int rr = t & 31, kb = t >> 5;
  const int half = 1 << (logn - 1);
  if (x.smax && t < 256) {  // d = t: the run of 32 bins from (r0 + 256 c + (t << log_row)) ^ half on, run r0 / 32 of its tile column
    const float bound = fmaf(__log2f(__uint_as_float(pmax[t])), 3.01029995663981195f, -db_off);
    x.smax[((size_t)((x.abs0 + f) & x.smax_mask) << (logn - 5)) + ((r0 >> 5) << (logn - 8)) + (c << 8) + t] = bound;  // rows_smax_index
  }
  float* hrow = (x.hist_out && f >= x.first_hist && SS_ROWS_ABL != 2) ? x.hist_out + ((size_t)(f - x.first_hist) << logn) : nullptr;  // (workgroup-uniform)
  // fft_v shift=true: X[k] lands at k ^ (N/2)
  const auto bin_of = [&](int i) { return ((r0 + rr) + (c << 8) + ((kb + 16 * i) << log_row)) ^ half; };
  if (psd) {  // (null: a call that keeps no dB plane, specscan.hip run_batch)
#pragma unroll
    for (int i = 0; i < 16; ++i) store_f1_policy<SS_AUX_ROWS>(&out[bin_of(i)], s[(kb + 16 * i) * 33 + rr]);
  }
  if (hrow) {
    // the ceiling values first, all in flight together, then the stores: load, subtract, store per output is what the compiler keeps
    // when written that way (it cannot know that ceiling and ring are different memory), and then every store waits for its own
    // load and for the store before it (fft1024_kernels.h: fft_rows1024_tile)
    if (!x.thr) {  // (workgroup-uniform) the rows leave as dB values: the tiles that are evaluated subtract the ceiling (DetectArgs::ring_db_from)
#pragma unroll
      for (int i = 0; i < 16; ++i) store_f1_policy<SS_AUX_ROWS>(&hrow[bin_of(i)], s[(kb + 16 * i) * 33 + rr]);
    } else {
      float th[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) th[i] = x.thr[bin_of(i)];
#pragma unroll
      for (int i = 0; i < 16; ++i) store_f1_policy<SS_AUX_ROWS>(&hrow[bin_of(i)], s[(kb + 16 * i) * 33 + rr] - th[i]);  // noise_learner.cpp:55, as detect_tile forms it
    }
  }
}
__global__ __launch_bounds__(512, 8) void k_fft_rows256_psd(const float2* __restrict__ work, const float2* __restrict__ tw256, float db_off,
                                                            float* __restrict__ psd, int logn, int lognsub, RowsExtra x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  Rows256Args g;
  g.work = work;
  g.tw256 = tw256;
  g.db_off = db_off;
  g.psd = psd;
  g.logn = logn;
  g.lognsub = lognsub;
  g.x = x;
  fft_rows256_tile(g, (int)blockIdx.x, smem_raw, (int)threadIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// N = 256 R, R = 4, 8, 16 (1024, 2048, 4096 points: what getFft picks for fs up to 1.024 MS/s), 32 / R frames per
// workgroup of 512 threads, in registers:
//   n = R m + q:  X[k' + 256 kap] = sum_q W_R^(q kap) * W_N^(q k') * Z_q[k'],   Z_q = FFT256 over m of x[R m + q]
// Thread t loads x[frame][tt + 16 R r] with tt = t % (16 R) — consecutive threads, consecutive samples — which is element
// m = j + 16 r (j = tt / R) of sub-sequence q = tt % R, runs the two register passes of fft256_passes (32 sub-FFTs per
// workgroup), then the Z values change owner through LDS ([frame, q][k'] planes, 257-word pitch) so that a thread holds
// Z_0..R-1[k'] for 16 / R pairs (frame, k'), applies W_N^(q k') from a [q][k'] table and finishes each pair with an
// R-point DFT. Stores run along k'. For R = 16: three 16-point DFTs and two exchanges, the shape of the 8192-point kernel.
// Compiled for 4 waves per SIMD: with 80 registers (6 waves) the 4096-point instance spills and is 10 % slower.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFft256xRLdsBytes = 32 * 273 * 4 + 256 * 8;  // >= 32 sub-FFTs x 257 words for the second exchange, W_256 behind them

// R-point DFT of a[B .. B+R-1] (compile-time base: the array must stay in registers)
template <int R, int B>
__device__ __forceinline__ void dft_small(float2 (&a)[16]) {
  if constexpr (R == 16) dft16(a);
  else if constexpr (R == 8) dft8(a[B], a[B + 1], a[B + 2], a[B + 3], a[B + 4], a[B + 5], a[B + 6], a[B + 7]);
  else if constexpr (R == 4) dft4(a[B], a[B + 1], a[B + 2], a[B + 3]);
  else dft2(a[B], a[B + 1]);
}
// all 16 / R groups
template <int R>
__device__ __forceinline__ void dft_small_all(float2 (&a)[16]) {
  if constexpr (R == 16) {
    dft_small<16, 0>(a);
  } else if constexpr (R == 8) {
    dft_small<8, 0>(a);
    dft_small<8, 8>(a);
  } else if constexpr (R == 4) {
    dft_small<4, 0>(a);
    dft_small<4, 4>(a);
    dft_small<4, 8>(a);
    dft_small<4, 12>(a);
  } else {
    dft_small<2, 0>(a);
    dft_small<2, 2>(a);
    dft_small<2, 4>(a);
    dft_small<2, 6>(a);
    dft_small<2, 8>(a);
    dft_small<2, 10>(a);
    dft_small<2, 12>(a);
    dft_small<2, 14>(a);
  }
}
template <int R>
__device__ __forceinline__ constexpr int slot_small(int k) {
  return R == 16 ? slot16(k) : R == 8 ? slot8(k) : k;
}

template <int FMT, int LOGR>
__global__ __launch_bounds__(512, LOGR == 4 ? 4 : 6) void k_fft256xR_psd(const void* __restrict__ iq, long long item_stride, int nframes,
                                                         const float* __restrict__ win, const float2* __restrict__ tw256,
                                                         const float2* __restrict__ twn /* [q][k'] W_N^(q k') */, float db_off, float scale,
                                                         float* __restrict__ psd) {
  constexpr int R = 1 << LOGR, N = 256 * R, FPW = 32 / R, U = 16 / R;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s = reinterpret_cast<float*>(smem_raw);
  const int t = threadIdx.x;
  float2* tw_lds = reinterpret_cast<float2*>(smem_raw + 32 * 273 * 4);  // W_256 in LDS, as in fft_cols256_tile
  if (t < 256) tw_lds[t] = tw256[t];
  const int fl = t >> (4 + LOGR), tt = t & (16 * R - 1);
  const int q = tt & (R - 1), j = tt >> LOGR;
  // frames past the end (ragged last workgroup) recompute the last frame and do not store
  const int frame = min((int)blockIdx.x * FPW + fl, nframes - 1);
  const size_t in_base = (size_t)frame * (size_t)item_stride;
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int n = tt + 16 * R * r;
    const float2 x = load_iq<FMT>(iq, in_base + n, scale);
    const float w = win[n];
    a[r] = make_float2(x.x * w, x.y * w);  // volk_32fc_32f_multiply_32fc
  }
  float2 c[16];
  const int sub = (fl << LOGR) | q;  // which of the 32 sub-FFTs
  fft256_passes<kFft256PitchCols>(a, c, s, tw_lds, sub, j);  // Z_q[j + 16 k] in c[slot16(k)]
  __syncthreads();
  // second exchange: Z_q[k'] to word sub * 257 + k'; pair p = t + 512 u = (frame p / 256, k' = p % 256) reads its R q's
  float* zp = s + sub * 257 + j;
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = c[slot16(k)].x;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p = t + 512 * u;
    const float* zr = s + ((p >> 8) << LOGR) * 257 + (p & 255);
#pragma unroll
    for (int qq = 0; qq < R; ++qq) a[u * R + qq].x = zr[qq * 257];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = c[slot16(k)].y;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p = t + 512 * u;
    const int kp = p & 255;
    const float* zr = s + ((p >> 8) << LOGR) * 257 + kp;
#pragma unroll
    for (int qq = 0; qq < R; ++qq) {
      const float2 v = make_float2(a[u * R + qq].x, zr[qq * 257]);
      a[u * R + qq] = qq == 0 ? v : cmul(v, twn[qq * 256 + kp]);  // W_N^(q k')
    }
  }
  dft_small_all<R>(a);  // X[k' + 256 kap] in slot kap of every group
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p = t + 512 * u;
    const int kp = p & 255;
    const int f2 = (int)blockIdx.x * FPW + (p >> 8);
    if (f2 < nframes) {
      float* out = psd + (size_t)f2 * N;
#pragma unroll
      for (int kap = 0; kap < R; ++kap) out[(kp + 256 * kap) ^ (N / 2)] = psd_db(a[u * R + slot_small<R>(kap)], db_off);  // fft_v shift=true
    }
  }
}

// The same decomposition for the ROWS of a four-step with N2 = 256 R, R = 2, 4 (N = 131072, 262144): 32 / R rows k1 per
// workgroup straight from the work buffer (the step-A twiddle is already in it), X_row[k' + 256 kap] -> bin
// k1 + 256 (k' + 256 kap). The dB values go through LDS once more ([k2][row], pitch rows + 1) so that stores run along k1.
// LDS of k_fft_rows256xR_psd: the exchange plane of the register passes (32 sub-sequences x 273) or the read-out tile
// (N2 k2 x (32 / R rows + 1)), whichever is larger, and W_256 behind it
constexpr int fft_rowsR_plane_bytes(int logr) {
  const int n2 = 256 << logr, fpw = 32 >> logr;
  return 4 * (n2 * (fpw + 1) > 32 * kFft256PitchCols ? n2 * (fpw + 1) : 32 * kFft256PitchCols);
}
constexpr int fft_rowsR_lds_bytes(int logr) { return fft_rowsR_plane_bytes(logr) + 256 * 8; }

template <int LOGR>
__global__ __launch_bounds__(512, 4) void k_fft_rows256xR_psd(const float2* __restrict__ work, const float2* __restrict__ tw256,
                                                              const float2* __restrict__ twn /* [q][k'] W_N2^(q k') */, float db_off,
                                                              float* __restrict__ psd, int xcd_map) {
  constexpr int R = 1 << LOGR, N2 = 256 * R, FPW = 32 / R, U = 16 / R, LOGN = 16 + LOGR;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s = reinterpret_cast<float*>(smem_raw);
  const int t = threadIdx.x;
  float2* tw_lds = reinterpret_cast<float2*>(smem_raw + fft_rowsR_plane_bytes(LOGR));  // W_256 in LDS, as in fft_cols256_tile
  if (t < 256) tw_lds[t] = tw256[t];
  const int fl = t >> (4 + LOGR), tt = t & (16 * R - 1);
  const int q = tt & (R - 1), j = tt >> LOGR;
  constexpr int TILES = 256 / FPW;
  // A tile's dB values leave as runs of FPW floats along k1 (one run per k2): 32 / FPW neighbouring tiles share every
  // 128-byte line of the PSD plane. Consecutive blocks go to consecutive XCDs (block b -> XCD b mod 8, each with its own
  // L2), so with the plain order the pieces of a line are written through eight different L2s; xcd_map hands XCD x
  // the rows [32 x, 32 x + 32) of every frame, its tiles in consecutive blocks of that XCD, and the pieces meet in one L2.
  int f, r0;
  if (xcd_map) {
    constexpr int GT = 32 / FPW;  // tiles per 32-row group
    const int x = blockIdx.x & 7, v = blockIdx.x >> 3;
    f = v / GT;
    r0 = (x * GT + v % GT) * FPW;
  } else {
    f = blockIdx.x / TILES;
    r0 = (blockIdx.x % TILES) * FPW;
  }
  const float2* row = work + ((size_t)f << LOGN) + (size_t)(r0 + fl) * N2;
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = row[tt + 16 * R * r];
  float2 c[16];
  const int sub = (fl << LOGR) | q;
  fft256_passes<kFft256PitchCols>(a, c, s, tw_lds, sub, j);
  __syncthreads();
  float* zp = s + sub * 257 + j;
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = c[slot16(k)].x;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p = t + 512 * u;
    const float* zr = s + ((p >> 8) << LOGR) * 257 + (p & 255);
#pragma unroll
    for (int qq = 0; qq < R; ++qq) a[u * R + qq].x = zr[qq * 257];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) zp[16 * k] = c[slot16(k)].y;
  __syncthreads();
  float dbv[16];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p = t + 512 * u;
    const int kp = p & 255;
    const float* zr = s + ((p >> 8) << LOGR) * 257 + kp;
#pragma unroll
    for (int qq = 0; qq < R; ++qq) {
      const float2 v = make_float2(a[u * R + qq].x, zr[qq * 257]);
      a[u * R + qq] = qq == 0 ? v : cmul(v, twn[qq * 256 + kp]);
    }
  }
  dft_small_all<R>(a);
#pragma unroll
  for (int u = 0; u < U; ++u) {
#pragma unroll
    for (int kap = 0; kap < R; ++kap) dbv[u * R + kap] = psd_db(a[u * R + slot_small<R>(kap)], db_off);
  }
  __syncthreads();  // every Z read is done before the plane is reused for the read-out
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int p = t + 512 * u;
#pragma unroll
    for (int kap = 0; kap < R; ++kap) s[((p & 255) + 256 * kap) * (FPW + 1) + (p >> 8)] = dbv[u * R + kap];
  }
  __syncthreads();
  float* out = psd + ((size_t)f << LOGN);
  const int rr = t & (FPW - 1), kb = t / FPW;
  constexpr int half = 1 << (LOGN - 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k2 = kb + (512 / FPW) * i;
    out[((r0 + rr) + (k2 << 8)) ^ half] = s[k2 * (FPW + 1) + rr];  // fft_v shift=true
  }
}

// N2 = 256 A (A = 2, 4, 8, 16): first half of the row FFTs, in place. n2 = 256 a + b, k2 = c + A d:
//   V[k1][c][b] = W_N2^(b c) * sum_a work[k1][256 a + b] W_A^(a c)
// One thread per (frame, k1, b): A loads and A stores at a stride of 256 elements, lanes along b (2 KiB runs). The
// 256-point FFTs over b that finish the job are k_fft_rows256_psd with nsub = A.
template <int A>
__global__ __launch_bounds__(256) void k_fft_sub_dft(float2* __restrict__ work, const float2* __restrict__ twsub /* [c][b] W_N2^(b c) */) {
  const int b = threadIdx.x;
  float2* row = work + (size_t)blockIdx.x * (256 * A) + b;  // blockIdx = f * 256 + k1: rows are contiguous in work
  float2 v[A];
#pragma unroll
  for (int a = 0; a < A; ++a) v[a] = row[256 * a];
  if constexpr (A == 16) dft16(v);
  else if constexpr (A == 8) dft8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  else if constexpr (A == 4) dft4(v[0], v[1], v[2], v[3]);
  else dft2(v[0], v[1]);
#pragma unroll
  for (int c = 0; c < A; ++c) {
    const float2 y = A == 16 ? v[slot16(c)] : A == 8 ? v[slot8(c)] : v[c];
    row[256 * c] = c == 0 ? y : cmul(y, twsub[c * 256 + b]);
  }
}

}  // namespace ss
