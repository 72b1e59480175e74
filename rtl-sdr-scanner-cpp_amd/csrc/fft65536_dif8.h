// fft65536_dif8.h — N = 65536 without a work buffer: a radix-8 decimation-in-frequency step folded into the load stage of the
// 8192-point transform (BASELINE.json config 3: 65536 points, 20 MS/s, int8 IQ).
//
// Same contract as the other front ends (Decimator + fft_v(Hamming, forward, shift) + PSD::work: reference
// sources/radio/blocks/decimator.h:15-22, sources/radio/sdr_device.cpp:164, sources/radio/blocks/psd.cpp:18-20). With
// N = 65536, m = n + 8192 q (n < 8192, q < 8) and k = 8 k' + r:
//
//     X[8 k' + r] = sum_n W_8192^(n k') * y_r[n],      y_r[n] = W_N^(n r) * sum_q w[n + 8192 q] x[n + 8192 q] W_8^(q r)
//
// so EIGHT workgroups per frame — one per residue r — each fold the whole int8 frame (128 KiB, which the other seven read
// from the same XCD's L2 at the same time) into the 8192 points y_r and run the 8192-point transform that ships for 8192-point
// frames on them (fft8192_v2_frame, fft8192_v2.h). What the four-step form pays for (fft256_kernels.h: a CF32 work buffer of
// 8 B/sample written by the column half and read back by the row half, 16 of the 25.9 B/sample a call moved over the fabric
// in round 4) is gone; what this form pays instead is arithmetic: every workgroup converts and windows all 65536 samples of
// its frame, ~10 vector instructions per sample on top of the transform's own.
//
// The workgroup's thread t holds, for the transform's first pass, y_r[t + 512 rho], rho < 16: 128 samples x[t + 512 (rho + 16 q)].
// The fold is a LOOP over q (the code of a 128-sample straight line is 13 KiB, and this chip wants its hot code small:
// profiles/README.md, ifetch2), q and q + 4 together — W_8^((q + 4) r) = (-1)^r W_8^(q r) — sixteen accumulators in registers:
//     acc[rho] += W_8^(q r) * ( w[m] x[m] + (-1)^r w[m'] x[m'] ),      m = t + 512 rho + 8192 q,  m' = m + 32768
// The Hamming taps are formed, not loaded (as the column tiles of the four-step form do, fft256_kernels.h): with
// theta = 2 pi (t + 8192 q) / (N - 1) — one table entry per thread, rotated once per q — and Phi_rho = 2 pi 512 rho / (N - 1)
//     w[t + 512 rho + 8192 q] = 0.54 + (-0.46 cos Phi_rho) cos theta + (0.46 sin Phi_rho) sin theta
// — two FMAs per sample with literal constants. Behind the loop  a[rho] = acc[rho] * W_N^(t r) * W_128^(rho r)  (a per-thread table
// entry and sixteen wave-uniform ones): sixteen vector instructions per pair of samples on top of the transform's own.
//
// The bins of residue r leave the transform in the order k' — X[8 k' + r], k' = 0 .. 8191 — and a frame's row holds them in BLOCKS of
// 32 Q bins: the 32 consecutive k' of every residue side by side, bin Q k' + g (DC in the middle: fft_v's shift) at
// (k' / 32) * 32 Q + 32 g + k' % 32 (dif_bin_offset below; until session 28 of round 5 the rows were residue-major, bin i at
// (i & 7) * 8192 + (i >> 3)). (The half rotation adds N/2 = 0 mod 8 to a bin number: it stays inside the residue and becomes the
// 8192-point transform's own half rotation.) Whoever reads such rows — the detect tiles, ss_read_window — permutes the bin number;
// nothing is transposed. The rows hold dB values (the tiles that are evaluated subtract the noise ceiling: DetectArgs::ring_db_from).
//
// How the samples reach the threads: LOADV = 0 — 128 two-byte loads per thread straight from global memory (each wave-load is
// one 128-byte line); LOADV = 1 — the frame comes through LDS in eight pieces of 16 KiB (eight values of rho x {q, q + 4}: sixteen
// runs of 512 samples), fetched by LDS-DMA in 16-byte lanes (two 1 KiB wave-instructions per wave and piece, against 128
// two-byte ones per thread) into the two halves of the exchange plane, which is idle until the transform's first exchange;
// the piece of one half is fetched while the other half's is folded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft8192_kernel.h"

namespace ss {

struct Dif8Front {
  const void* iq;          // the batch's frames, item_stride samples apart, 2 bytes per sample (CS8 / CU8)
  long long item_stride;
  const float2* wt;        // [8][512]  W_65536^(t r) times the input format's scale
  const float2* wrho;      // [8][16]   W_128^(rho r)                                          (wave-uniform: scalar loads)
  const float2* w8;        // [8][8]    W_8^(q r)                                              (wave-uniform: scalar loads)
  const float2* wthe;      // [512]     (cos, sin)(2 pi t / 65535)
  const float2* wq;        // [8]       (cos, sin)(2 pi 8192 q / 65535)                        (wave-uniform: scalar loads)
  // What a residue leaves for the detect stage besides its row (the transform's own epilogue, fft8192_v2.h):
  //   smax   the largest dB value of every 32-bin run of the residue's row — run j holds the bins 8 k' + r, k' in [32 j, 32 j + 32): the
  //          bins of residue r of the 256-bin tile column j — at smax[(abs0 + frame) & smax_mask][r * 256 + j]: a ring over the frames
  //          since the last reset, like the four-step form's (fft256_kernels.h RowsExtra::smax), in the layout the plan knows as 2
  //          (detect_fused.h PlanLongArgs::layout). Null: no maxima.
  float* smax;
  int smax_mask, abs0;
  int smax_pitch;          // floats per frame of the ring of maxima: 256 Q
  int first_hist;          // frame f's row is row f - first_hist of Fft8192Args::psd (RowsExtra::first_hist; <= 0 here: every frame has a row)
  int nframes;             // frames of the launch (the workgroup -> (frame, residue) map, dif8_item)
  int* zero_word;          // set to zero by the first workgroup: the count of the list the plan behind this launch appends to (or null)
};
// Which (frame, residue) the j-th of a launch's fold workgroups takes — W = 8 of them per frame (one residue each), or W = 4 (residues
// r and r + 4 each, dif8_front2). Workgroups go round the eight XCDs in dispatch order, so — as long as the launch's fold workgroups
// are consecutive blocks — the W workgroups of a frame are given 8 W consecutive numbers of ONE residue class mod 8: one XCD, one L2,
// and the frame crosses the fabric once (measured, scripts/ubench/dif8_lab: 19 MB of fetches per 128-frame launch against 135 MB with
// a frame's residues on eight XCDs).
template <int W>
__host__ __device__ inline void dif8_item(int j, int nframes, int* frame, int* residue) {
  static_assert(W == 4 || W == 8 || W == 16, "workgroups per frame");
  const int full = nframes >> 3;
  if (j < 8 * W * full) {
    *residue = (j >> 3) & (W - 1);
    *frame = ((j / (8 * W)) << 3) + (j & 7);
  } else {  // the frames beyond the last whole group of eight: in turn
    const int jj = j - 8 * W * full;
    *residue = jj & (W - 1);
    *frame = 8 * full + jj / W;
  }
}

// (-0.46 cos Phi_rho, 0.46 sin Phi_rho), Phi_rho = 2 pi 512 rho / 65535, rounded from double
__device__ constexpr float kDif8P[16] = {-0.46000000834465027f, -0.45944589376449585f, -0.4577849209308624f, -0.45502105355262756f, -0.4511609673500061f, -0.446213960647583f,
                                         -0.44019195437431335f, -0.43310946226119995f, -0.42498353123664856f, -0.4158337414264679f, -0.40568214654922485f, -0.39455321431159973f,
                                         -0.38247373700141907f, -0.36947280168533325f, -0.35558176040649414f, -0.3408340513706207f};
__device__ constexpr float kDif8Q[16] = {0.0f, 0.022571474313735962f, 0.045088570564985275f, 0.06749703735113144f, 0.08974289894104004f, 0.1117725521326065f,
                                         0.13353292644023895f, 0.15497159957885742f, 0.17603692412376404f, 0.19667814671993256f, 0.2168455421924591f, 0.23649051785469055f,
                                         0.25556573271751404f, 0.2740252912044525f, 0.2918246388435364f, 0.3089209496974945f};

// ... and for N = 131072 (the radix-16 fold: sixteen residues of 8192 points, Phi_rho = 2 pi 512 rho / 131071)
__device__ constexpr float kDif16P[16] = {-0.46000000834465027f, -0.4598614573478699f, -0.45944589376449585f, -0.4587535858154297f, -0.4577849507331848f, -0.4565405249595642f,
                                          -0.45502111315727234f, -0.45322760939598083f, -0.45116108655929565f, -0.4488227963447571f, -0.4462141692638397f, -0.44333672523498535f,
                                          -0.44019225239753723f, -0.4367825984954834f, -0.433109849691391f, -0.4291762113571167f};
__device__ constexpr float kDif16Q[16] = {0.0f, 0.011289050802588463f, 0.022571302950382233f, 0.03383995592594147f, 0.04508822783827782f, 0.05630933865904808f,
                                          0.06749653071165085f, 0.0786430612206459f, 0.08974222093820572f, 0.10078732669353485f, 0.11177171766757965f, 0.12268878519535065f,
                                          0.13353194296360016f, 0.14429466426372528f, 0.1549704670906067f, 0.16555292904376984f};
template <int Q>
__device__ __forceinline__ constexpr float dif_tap_p(int rho) { return Q == 16 ? kDif16P[rho] : kDif8P[rho]; }
template <int Q>
__device__ __forceinline__ constexpr float dif_tap_q(int rho) { return Q == 16 ? kDif16Q[rho] : kDif8Q[rho]; }

// The fold generalises from radix 8 (N = 65536) to radix Q = 16 (N = 131072 — what getFft picks at 20 MS/s, utils/radio_utils.cpp:98-104):
// Q residues of 8192 points per frame, X[Q k' + r], the same transform behind a fold of Q samples per point; tables below take Q.
constexpr int dif_table_float2(int q) { return q * 512 + q * 16 + q * q + 512 + q; }
constexpr int kDif8TableFloat2 = dif_table_float2(8);
// Host side: the five tables in one block (wt [Q][512], wrho [Q][16], w8 [Q][Q], wthe [512], wq [Q] in this order), double precision
// rounded once; `scale` is what load_iq multiplies the format's integers with; N = 8192 Q.
inline void dif8_host_tables(float2* tab, double scale, int Q = 8) {
  const double two_pi = 2.0 * 3.14159265358979323846;
  const int N = 8192 * Q;
  float2 *wt = tab, *wrho = wt + Q * 512, *w8 = wrho + Q * 16, *wthe = w8 + Q * Q, *wq = wthe + 512;
  for (int r = 0; r < Q; ++r) {
    for (int t = 0; t < 512; ++t) {
      const double ang = -two_pi * (double)(t * r) / (double)N;
      wt[r * 512 + t] = make_float2((float)(scale * cos(ang)), (float)(scale * sin(ang)));
    }
    for (int rho = 0; rho < 16; ++rho) {  // W_N^(512 rho r) = W_(16 Q)^(rho r)
      const double ang = -two_pi * (double)((rho * r) % (16 * Q)) / (double)(16 * Q);
      wrho[r * 16 + rho] = make_float2((float)cos(ang), (float)sin(ang));
    }
    for (int q = 0; q < Q; ++q) {
      // W_Q^(q r): exact values for the multiples of a right angle
      const int k = (q * r) % Q;
      const double ang = -two_pi * (double)k / (double)Q;
      double c = cos(ang), s_ = sin(ang);
      if ((4 * k) % Q == 0) {
        const int quarter = 4 * k / Q;
        c = quarter == 0 ? 1.0 : quarter == 2 ? -1.0 : 0.0;
        s_ = quarter == 1 ? -1.0 : quarter == 3 ? 1.0 : 0.0;
      }
      w8[r * Q + q] = make_float2((float)c, (float)s_);
    }
  }
  for (int t = 0; t < 512; ++t) {
    const double th = two_pi * (double)t / (double)(N - 1);
    wthe[t] = make_float2((float)cos(th), (float)sin(th));
  }
  for (int q = 0; q < Q; ++q) {
    const double ph = two_pi * 8192.0 * (double)q / (double)(N - 1);
    wq[q] = make_float2((float)cos(ph), (float)sin(ph));
  }
}
inline Dif8Front dif8_front_of(const void* iq, long long item_stride, const float2* tab, int Q = 8) {
  Dif8Front d{};
  d.iq = iq;
  d.item_stride = item_stride;
  d.wt = tab;
  d.wrho = tab + Q * 512;
  d.w8 = d.wrho + Q * 16;
  d.wthe = d.w8 + Q * Q;
  d.wq = d.wthe + 512;
  d.smax_pitch = 256 * Q;
  return d;
}

// The fold's rows (the averager ring while calls go through the fold): a row of 8192 Q bins (Q = 1 << logq residues, DC in the middle)
// in blocks of 32 Q bins — the 32 consecutive k' of every residue side by side: bin Q k' + g at (k' / 32) * 32 Q + 32 g + k' % 32. A
// residue's workgroup writes whole 128-byte lines (32 of its outputs); a detect tile's 256 bins are one run of 1 KB (Q = 8) or half of
// a run of 2 KB (Q = 16). (Until session 28 of round 5 the rows were residue-major — bin Q k' + g at 8192 g + k': a tile column's
// lines, one per row and residue, were 32 KB and 256 KB apart, all of them on one memory channel, and a pair of tiles held its
// workgroup for 7-15 us, alone on the chip or not: profiles/r05/s25_*, s27_*.)
__host__ __device__ inline int dif_bin_offset(int i, int logq) {
  const int g = i & ((1 << logq) - 1), k = i >> logq;
  return ((k >> 5) << (5 + logq)) + (g << 5) + (k & 31);
}
// ... and the bin at position `pos` of such a row
__host__ __device__ inline int dif_offset_bin(int pos, int logq) {
  const int k = ((pos >> (5 + logq)) << 5) + (pos & 31), g = (pos >> 5) & ((1 << logq) - 1);
  return (k << logq) | g;
}
__host__ __device__ inline int dif8_bin_offset(int i) { return dif_bin_offset(i, 3); }

// `rows` rows of 8192 Q floats (Q = 1 << logq) from bin order to the fold's order (to_perm8 != 0) or back; in and out are different
// memory. (The averager ring's window when a context changes between the fold and the four-step form.)
__global__ void k_rows_perm8(const float* __restrict__ in, float* __restrict__ out, int rows, int to_perm8, int logq) {
  const int logn = 13 + logq;
  const size_t total = (size_t)rows << logn;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    // i walks the fold's side (whole lines there, short pieces on the other)
    const size_t bin_order = ((i >> logn) << logn) | (size_t)dif_offset_bin((int)(i & ((1u << logn) - 1)), logq);
    if (to_perm8) out[i] = in[bin_order];
    else out[bin_order] = in[i];
  }
}

typedef const __attribute__((address_space(4))) float* dif8_const_fp;

// re, im of one two-byte sample (re in byte 0, im in byte 1) as floats: CS8 two's complement, CU8 offset binary (minus 127.5,
// sdr_device / SoapySDR's convention as load_iq has it). One instruction per component: the byte select and the sign extension
// ride on the conversion (SDWA) — the compiler's own sequence is a bit-field extract or a 16-bit shift first.
template <int FMT>
__device__ __forceinline__ void dif8_convert(unsigned raw, float& re, float& im) {
  if constexpr (FMT == FMT_CS8) {
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0" : "=v"(re) : "v"(raw));
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(im) : "v"(raw));
  } else {
    re = (float)(raw & 0xff) - 127.5f;          // v_cvt_f32_ubyte0
    im = (float)((raw >> 8) & 0xff) - 127.5f;   // v_cvt_f32_ubyte1
  }
}

// The rotated table entries of one trip: (cos, sin)(theta_t + 2 pi 8192 q / 65535) for q and, times sg = (-1)^r, for q + 4
struct Dif8Trip {
  float c1, s1, c2, s2, k2;  // k2 = 0.54 sg
  float w8c, w8s;            // W_8^(q r)
};
__device__ __forceinline__ Dif8Trip dif8_trip(float2 th, dif8_const_fp cq, dif8_const_fp c8, int q, float sg, int half_q = 4) {
  Dif8Trip x;
  const float qc1 = cq[2 * q], qs1 = cq[2 * q + 1], qc2 = cq[2 * (q + half_q)] * sg, qs2 = cq[2 * (q + half_q) + 1] * sg;
  x.c1 = fmaf(th.x, qc1, -(th.y * qs1));
  x.s1 = fmaf(th.y, qc1, th.x * qs1);
  x.c2 = fmaf(th.x, qc2, -(th.y * qs2));
  x.s2 = fmaf(th.y, qc2, th.x * qs2);
  x.k2 = 0.54f * sg;
  x.w8c = c8[2 * q];
  x.w8s = c8[2 * q + 1];
  return x;
}
// two samples of run rho — q and q + 4 — into the run's accumulator
template <int FMT>
__device__ __forceinline__ void dif8_accumulate(float2& acc, unsigned raw1, unsigned raw2, const Dif8Trip& x, int rho) {
  float re1, im1, re2, im2;
  dif8_convert<FMT>(raw1, re1, im1);
  dif8_convert<FMT>(raw2, re2, im2);
  const float t1 = fmaf(x.c1, kDif8P[rho], fmaf(x.s1, kDif8Q[rho], 0.54f));
  const float t2 = fmaf(x.c2, kDif8P[rho], fmaf(x.s2, kDif8Q[rho], x.k2));
  const float ur = fmaf(t2, re2, t1 * re1), ui = fmaf(t2, im2, t1 * im1);  // volk_32fc_32f_multiply_32fc (the format's scale rides on wt)
  acc.x = fmaf(ur, x.w8c, fmaf(-ui, x.w8s, acc.x));
  acc.y = fmaf(ur, x.w8s, fmaf(ui, x.w8c, acc.y));
}

// The load stage: a[rho] = y_r[t + 512 rho] for the workgroup's residue r. `smem_raw`: the transform's exchange plane (32 KiB of it,
// LOADV = 1 only). Every thread of the workgroup calls it (barriers inside for LOADV = 1). `after_first_issue()` is called once the
// first fetches are on their way: the place for the caller's own table loads (they land behind the first pieces, which the fold
// waits for anyway, and ahead of nothing).
template <int FMT, int LOADV, class F>
__device__ __forceinline__ void dif8_front(const Dif8Front& d, size_t frame_in, int residue, unsigned char* __restrict__ smem_raw, int t, float2 (&a)[16],
                                           F&& after_first_issue) {
  static_assert(FMT == FMT_CS8 || FMT == FMT_CU8, "two-byte samples");
  const char* fb = reinterpret_cast<const char*>(d.iq) + frame_in * (size_t)d.item_stride * 2;
  const __amdgpu_buffer_rsrc_t rin = buffer_of(fb, 65536 * 2);
  const dif8_const_fp cq = (dif8_const_fp)(uintptr_t)d.wq;
  const dif8_const_fp c8 = (dif8_const_fp)(uintptr_t)(d.w8 + residue * 8);
  const dif8_const_fp crho = (dif8_const_fp)(uintptr_t)(d.wrho + residue * 16);
  const float sg = (residue & 1) ? -1.0f : 1.0f;
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) a[rho] = make_float2(0.f, 0.f);
  if constexpr (LOADV == 0) {
    after_first_issue();
    const float2 th = d.wthe[t];
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const Dif8Trip x = dif8_trip(th, cq, c8, q, sg);
      const int so = 16384 * q;  // bytes: 8192 q samples
#pragma unroll
      for (int sub = 0; sub < 4; ++sub) {
        unsigned raw1[4], raw2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          raw1[i] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, t * 2 + 1024 * (4 * sub + i), so, SS_AUX_DIF_IQ);
          raw2[i] = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, t * 2 + 1024 * (4 * sub + i) + 65536, so, SS_AUX_DIF_IQ);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) dif8_accumulate<FMT>(a[4 * sub + i], raw1[i], raw2[i], x, 4 * sub + i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  } else {
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63;
    // piece (q, half): rho = 8 half + i, i < 8, of q and of q + 4; run (i, j) (j = 0: q, 1: q + 4) at half * 16 KiB + (2 i + j) KiB;
    // wave w fetches the two runs of i = w
    const auto issue = [&](int q, int half) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem_raw + half * 16384 + (2 * w + j) * 1024), 16, lane * 16,
                                                 1024 * (8 * half + w) + 65536 * j + 16384 * q, 0, SS_AUX_DIF_IQ);
    };
    issue(0, 0);
    issue(0, 1);
    after_first_issue();
    const float2 th = d.wthe[t];
    const unsigned short* mine = reinterpret_cast<const unsigned short*>(smem_raw) + t;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const Dif8Trip x = dif8_trip(th, cq, c8, q, sg);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        // this half's piece has landed for every wave once all have waited for their own fetches and met here — at which point
        // everybody is also done reading the OTHER half's previous piece, whose place the next fetch may take. (The table loads
        // behind the first two pieces make the first wait one for everything.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (!(q == 0 && half == 0) && !(q == 3 && half == 1)) {
          if (half == 0) issue(q, 1);       // (the other half's piece of this trip: its place was read a piece ago)
          else issue(q + 1, 0);
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          unsigned raw1[4], raw2[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            raw1[i] = mine[half * 8192 + 1024 * (4 * sub + i)];
            raw2[i] = mine[half * 8192 + 1024 * (4 * sub + i) + 512];
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float2& acc = a[8 * half + 4 * sub + i];
            dif8_accumulate<FMT>(acc, raw1[i], raw2[i], x, 8 * half + 4 * sub + i);
            // (the sums are formed HERE: left to itself the compiler sinks half of a piece's arithmetic below the next barrier and
            // spills the converted samples it then has to keep)
            asm volatile("" : "+v"(acc.x), "+v"(acc.y));
          }
          __builtin_amdgcn_sched_barrier(0);  // (the next four pairs are read when these are done: sixteen at once spill)
        }
      }
    }
    __syncthreads();  // the plane goes back to the transform
  }
  const float2 wt = d.wt[residue * 512 + t];
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) a[rho] = cmul(a[rho], cmul(wt, make_float2(crho[2 * rho], crho[2 * rho + 1])));
}

// Two residues by one workgroup: r (< 4) and r + 4 share a frame's conversions, taps and pair sums — W_8^(q (r + 4)) = (-1)^q W_8^(q r)
// — so the fold costs twenty vector instructions per pair of samples for BOTH residues instead of sixteen for each; the price is
// a second set of sixteen accumulators (and of points, kept in registers while the first residue goes through the transform):
// 128 registers, four waves per SIMD, two workgroups per CU. LDS-DMA pieces as above. a = y_r, a2 = y_(r + 4).
// Q = 16 (N = 131072): residues r (< 8) and r + 8, eight trips over q and q + 8, W_16 in the place of W_8 — the same code.
template <int FMT, int Q, class F>
__device__ __forceinline__ void dif8_front2(const Dif8Front& d, size_t frame_in, int residue, unsigned char* __restrict__ smem_raw, int t, float2 (&a)[16],
                                            float2 (&a2)[16], F&& after_first_issue) {
  static_assert(FMT == FMT_CS8 || FMT == FMT_CU8, "two-byte samples");
  static_assert(Q == 8 || Q == 16, "radix of the fold");
  constexpr int HQ = Q / 2;
  const char* fb = reinterpret_cast<const char*>(d.iq) + frame_in * (size_t)d.item_stride * 2;
  const __amdgpu_buffer_rsrc_t rin = buffer_of(fb, 8192 * Q * 2);
  const dif8_const_fp cq = (dif8_const_fp)(uintptr_t)d.wq;
  const dif8_const_fp c8 = (dif8_const_fp)(uintptr_t)(d.w8 + residue * Q);
  const dif8_const_fp crho = (dif8_const_fp)(uintptr_t)(d.wrho + residue * 16), crho2 = (dif8_const_fp)(uintptr_t)(d.wrho + (residue + HQ) * 16);
  const float sg = (residue & 1) ? -1.0f : 1.0f;
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) a[rho] = a2[rho] = make_float2(0.f, 0.f);
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
#ifndef SS_DIF_NODMA  // (lab builds: 1 = no fetches and no waits for them — garbage results, the fold's own arithmetic and barriers alone)
#define SS_DIF_NODMA 0
#endif
  const auto issue = [&](int q, int half) {
    if (SS_DIF_NODMA) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem_raw + half * 16384 + (2 * w + j) * 1024), 16, lane * 16,
                                               1024 * (8 * half + w) + (8192 * Q) * j + 16384 * q, 0, SS_AUX_DIF_IQ);
  };
  issue(0, 0);
  issue(0, 1);
  after_first_issue();
  const float2 th = d.wthe[t];
  const unsigned short* mine = reinterpret_cast<const unsigned short*>(smem_raw) + t;
#pragma unroll 1
  for (int q = 0; q < HQ; ++q) {
    const Dif8Trip x = dif8_trip(th, cq, c8, q, sg, HQ);
    const float sq = (q & 1) ? -1.0f : 1.0f;  // W_Q^(q (r + Q/2)) = (-1)^q W_Q^(q r)
    const float w8c2 = x.w8c * sq, w8s2 = x.w8s * sq;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (!SS_DIF_NODMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(q == 0 && half == 0) && !(q == HQ - 1 && half == 1)) {
        if (half == 0) issue(q, 1);
        else issue(q + 1, 0);
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        unsigned raw1[4], raw2[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          raw1[i] = mine[half * 8192 + 1024 * (4 * sub + i)];
          raw2[i] = mine[half * 8192 + 1024 * (4 * sub + i) + 512];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rho = 8 * half + 4 * sub + i;
          float re1, im1, re2, im2;
          dif8_convert<FMT>(raw1[i], re1, im1);
          dif8_convert<FMT>(raw2[i], re2, im2);
          const float t1 = fmaf(x.c1, dif_tap_p<Q>(rho), fmaf(x.s1, dif_tap_q<Q>(rho), 0.54f));
          const float t2 = fmaf(x.c2, dif_tap_p<Q>(rho), fmaf(x.s2, dif_tap_q<Q>(rho), x.k2));
          const float ur = fmaf(t2, re2, t1 * re1), ui = fmaf(t2, im2, t1 * im1);
          float2 &acc = a[rho], &acc2 = a2[rho];
          acc.x = fmaf(ur, x.w8c, fmaf(-ui, x.w8s, acc.x));
          acc.y = fmaf(ur, x.w8s, fmaf(ui, x.w8c, acc.y));
          acc2.x = fmaf(ur, w8c2, fmaf(-ui, w8s2, acc2.x));
          acc2.y = fmaf(ur, w8s2, fmaf(ui, w8c2, acc2.y));
          asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc2.x), "+v"(acc2.y));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();  // the plane goes back to the transform
  const float2 wt = d.wt[residue * 512 + t], wt2 = d.wt[(residue + HQ) * 512 + t];
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) {
    a[rho] = cmul(a[rho], cmul(wt, make_float2(crho[2 * rho], crho[2 * rho + 1])));
    a2[rho] = cmul(a2[rho], cmul(wt2, make_float2(crho2[2 * rho], crho2[2 * rho + 1])));
  }
}

// Round 6: the same two residues, the fold turned inside out — for one run rho ALL EIGHT samples of a point at once, and the sum over q as
// the radix-8 butterfly it is instead of four complex multiply-accumulates per residue:
//     s_q  = w[m] x[m] + (-1)^r w[m'] x[m']                      (q < 4: the pair sums, as before)
//     E    = s_0 + W_8^(2r) s_2,    O = W_8^r s_1 + W_8^(3r) s_3 = W_8^r (s_1 + W_8^(2r) s_3)
//     y_r  = E + O,                 y_(r+4) = E - O
// W_8^(2r) is a power of -i (operands swapped, no arithmetic) and W_8^r is 1, -i or (+-1 - i) / sqrt 2 — with R a template parameter the
// butterfly is 8 (R even) or 10 (R odd) vector instructions per point where the accumulating form spends 32; per point 56-58 instead of 80,
// 912 instead of 1280 per thread and fold. What it needs is all eight taps of a point: the rotating part of the taps for every q,
// (cos, sin)(theta_t + 2 pi 8192 q / 65535), sixteen registers kept through the loop (the accumulating form keeps 64 accumulators
// instead; here a[rho] and a2[rho] are written once) — and straight-line code for the sixteen runs (a loop's counter cannot index
// registers): 9 KiB where the accumulating form's loop has 2.5. The residue is a run-time value (one copy of the code): the four shapes
// of the butterfly sit behind scalar branches. A piece of LDS-DMA is two runs rho x eight q (sixteen runs of 1 KiB, as before: wave w
// fetches q = w of both).

__device__ __forceinline__ void dif8_bfly2(int R, const float2 (&s)[4], float2& y, float2& y2) {  // R: workgroup-uniform (scalar branches)
  constexpr float kH = 0.70710678118654752f;
  if (R == 0) {  // W_8^0 = 1 throughout
    const float2 e = make_float2(s[0].x + s[2].x, s[0].y + s[2].y), o = make_float2(s[1].x + s[3].x, s[1].y + s[3].y);
    y = make_float2(e.x + o.x, e.y + o.y);
    y2 = make_float2(e.x - o.x, e.y - o.y);
  } else if (R == 2) {  // W_8^4 = -1, W_8^2 = -i, W_8^6 = i: E = s0 - s2, O = -i (s1 - s3)
    const float2 e = make_float2(s[0].x - s[2].x, s[0].y - s[2].y), d = make_float2(s[1].x - s[3].x, s[1].y - s[3].y);
    y = make_float2(e.x + d.y, e.y - d.x);
    y2 = make_float2(e.x - d.y, e.y + d.x);
  } else if (R == 1) {  // W_8^2 = -i, W_8 = (1 - i) / sqrt 2: E = s0 - i s2, O = W_8 (s1 - i s3)
    const float2 e = make_float2(s[0].x + s[2].y, s[0].y - s[2].x), d = make_float2(s[1].x + s[3].y, s[1].y - s[3].x);
    const float2 g = make_float2(d.x + d.y, d.y - d.x);  // (1 - i) d
    y = make_float2(fmaf(kH, g.x, e.x), fmaf(kH, g.y, e.y));
    y2 = make_float2(fmaf(-kH, g.x, e.x), fmaf(-kH, g.y, e.y));
  } else {  // R == 3: W_8^6 = i, W_8^3 = (-1 - i) / sqrt 2, W_8^9 = W_8: E = s0 + i s2, O = W_8 (s3 - i s1)
    const float2 e = make_float2(s[0].x - s[2].y, s[0].y + s[2].x), d = make_float2(s[3].x + s[1].y, s[3].y - s[1].x);
    const float2 g = make_float2(d.x + d.y, d.y - d.x);
    y = make_float2(fmaf(kH, g.x, e.x), fmaf(kH, g.y, e.y));
    y2 = make_float2(fmaf(-kH, g.x, e.x), fmaf(-kH, g.y, e.y));
  }
}

template <int FMT, class F>
__device__ __forceinline__ void dif8_front2_bfly(const Dif8Front& d, size_t frame_in, int R, unsigned char* __restrict__ smem_raw, int t, float2 (&a)[16], float2 (&a2)[16],
                                                 F&& after_first_issue) {
  static_assert(FMT == FMT_CS8 || FMT == FMT_CU8, "two-byte samples");
  const char* fb = reinterpret_cast<const char*>(d.iq) + frame_in * (size_t)d.item_stride * 2;
  const __amdgpu_buffer_rsrc_t rin = buffer_of(fb, 65536 * 2);
  const dif8_const_fp cq = (dif8_const_fp)(uintptr_t)d.wq;
  const dif8_const_fp crho = (dif8_const_fp)(uintptr_t)(d.wrho + R * 16), crho2 = (dif8_const_fp)(uintptr_t)(d.wrho + (R + 4) * 16);
  const float sg = (R & 1) ? -1.0f : 1.0f;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  // piece p: runs rho = 2 p + j, j < 2, of every q; run (j, q) at half * 16 KiB + (8 j + q) KiB; wave w fetches q = w of both
  const auto issue = [&](int p, int half) {
    if (SS_DIF_NODMA) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem_raw + half * 16384 + (8 * j + w) * 1024), 16, lane * 16,
                                               1024 * (2 * p + j) + 16384 * w, 0, SS_AUX_DIF_IQ);
  };
  issue(0, 0);
  issue(1, 1);
  after_first_issue();
  const float2 th = d.wthe[t];
  // (cos, sin)(theta_t + phi_q) for the eight q — those of q + 4 times (-1)^R, the sign of their samples in the pair sums
  float tc[8], ts[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float qc = cq[2 * q] * (q >= 4 ? sg : 1.0f), qs = cq[2 * q + 1] * (q >= 4 ? sg : 1.0f);
    tc[q] = fmaf(th.x, qc, -(th.y * qs));
    ts[q] = fmaf(th.y, qc, th.x * qs);
  }
  const float k1 = 0.54f, k2 = 0.54f * sg;
  const unsigned short* mine = reinterpret_cast<const unsigned short*>(smem_raw) + t;
  // (straight-line code, sixteen points: a[] and a2[] are registers, and a loop's counter cannot index registers)
#pragma unroll
  for (int pp = 0; pp < 4; ++pp) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (!SS_DIF_NODMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (!(pp == 0 && half == 0) && !(pp == 3 && half == 1)) issue(2 * pp + half + 1, half ^ 1);  // (the other half's next piece: its place was read a piece ago)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int rho = 4 * pp + 2 * half + j;
        const float tp = kDif8P[rho], tq = kDif8Q[rho];
        unsigned raw[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) raw[q] = mine[half * 8192 + 512 * (8 * j + q)];
        float2 s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float re1, im1, re2, im2;
          dif8_convert<FMT>(raw[q], re1, im1);
          dif8_convert<FMT>(raw[q + 4], re2, im2);
          const float t1 = fmaf(tc[q], tp, fmaf(ts[q], tq, k1));
          const float t2 = fmaf(tc[q + 4], tp, fmaf(ts[q + 4], tq, k2));
          s[q] = make_float2(fmaf(t2, re2, t1 * re1), fmaf(t2, im2, t1 * im1));  // volk_32fc_32f_multiply_32fc (the format's scale rides on wt)
        }
        dif8_bfly2(R, s, a[rho], a2[rho]);
        asm volatile("" : "+v"(a[rho].x), "+v"(a[rho].y), "+v"(a2[rho].x), "+v"(a2[rho].y));  // (formed HERE, not sunk below the next barrier)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();  // the plane goes back to the transform
  const float2 wt = d.wt[R * 512 + t], wt2 = d.wt[(R + 4) * 512 + t];
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) {
    a[rho] = cmul(a[rho], cmul(wt, make_float2(crho[2 * rho], crho[2 * rho + 1])));
    a2[rho] = cmul(a2[rho], cmul(wt2, make_float2(crho2[2 * rho], crho2[2 * rho + 1])));
  }
}

// ... and the radix-16 fold (131072 points, residues R < 8 and R + 8) the same way: all SIXTEEN samples of a point at once,
//     s_q = w[m] x[m] + (-1)^R w[m'] x[m']   (q < 8, m' = m + 65536),
//     A = s_0 + (-i)^R s_4,  B = s_2 + (-i)^R s_6,  C = s_1 + (-i)^R s_5,  D = s_3 + (-i)^R s_7       (four shapes of R mod 4: scalar branches)
//     E = A + W_8^R B,       O = W_16^R (C + W_8^R D),       y_R = E + O,   y_(R+8) = E - O          (two complex constants in scalar registers)
// 24 vector instructions per point for the sums where the accumulating form spends 64 (eight pairs x two residues x four FMAs); with the
// conversions, taps and pair sums (96) 120 instead of 160 per point. The taps of sixteen samples would need thirty-two rotated table
// entries in registers; instead every tap is 0.54 + c cos theta_t + s sin theta_t with (c, s) = (-0.46 cos Phi, 0.46 sin Phi),
// Phi = 2 pi (512 rho + 8192 q) / 131071, as LITERALS (kDif16TapC / kDif16TapS, rounded once from fp64) — two FMAs per sample as before.
// A piece of LDS-DMA is one run rho x sixteen q (wave w fetches q = w and q = w + 8).
__device__ constexpr float kDif16TapC[16][16] = {
    {-0.460000008f, -0.424984068f, -0.325267166f, -0.176030561f, 5.51278572e-06f, 0.176040739f, 0.325274974f, 0.42498827f, 0.460000008f, 0.424979836f, 0.325259387f, 0.176020369f, -1.65383572e-05f, -0.176050931f, -0.325282753f, -0.424992502f},
    {-0.459861457f, -0.420535892f, -0.317186594f, -0.165547773f, 0.0112945624f, 0.186417386f, 0.333159417f, 0.429180205f, 0.459861189f, 0.420531422f, 0.317178607f, 0.165537491f, -0.0113055846f, -0.186427459f, -0.333167017f, -0.429184169f},
    {-0.459445894f, -0.415834397f, -0.308914959f, -0.154965281f, 0.0225768089f, 0.196681723f, 0.340843201f, 0.433113575f, 0.459445357f, 0.415829688f, 0.308906794f, 0.154954895f, -0.0225878209f, -0.196691692f, -0.340850592f, -0.4331173f},
    {-0.458753586f, -0.410882443f, -0.300457239f, -0.144289434f, 0.0338454545f, 0.206827596f, 0.348321646f, 0.436786056f, 0.458752781f, 0.410877496f, 0.300448865f, 0.144278958f, -0.0338564515f, -0.206837445f, -0.348328829f, -0.436789542f},
    {-0.457784951f, -0.405682981f, -0.2918185f, -0.133526668f, 0.0450937152f, 0.21684888f, 0.355590284f, 0.440195441f, 0.457783848f, 0.405677766f, 0.291809976f, 0.133516118f, -0.0451046862f, -0.216858611f, -0.355597258f, -0.44019866f},
    {-0.456540525f, -0.40023911f, -0.283004016f, -0.122683465f, 0.0563148111f, 0.226739541f, 0.362644702f, 0.443339676f, 0.456539184f, 0.400233686f, 0.282995313f, 0.122672841f, -0.0563257523f, -0.226749137f, -0.362651497f, -0.443342626f},
    {-0.455021113f, -0.394554198f, -0.274019063f, -0.111766368f, 0.0675019845f, 0.236493617f, 0.369480699f, 0.446216851f, 0.455019504f, 0.394548506f, 0.274010181f, 0.111755677f, -0.0675128922f, -0.23650308f, -0.369487256f, -0.446219534f},
    {-0.453227609f, -0.388631582f, -0.264869034f, -0.100781947f, 0.0786484927f, 0.246105239f, 0.376094133f, 0.44882524f, 0.453225732f, 0.388625681f, 0.264860004f, 0.100771189f, -0.0786593556f, -0.246114552f, -0.376100481f, -0.448827654f},
    {-0.451161087f, -0.382474869f, -0.255559444f, -0.0897368193f, 0.0897476301f, 0.255568624f, 0.382481009f, 0.451163232f, 0.451158941f, 0.38246876f, 0.255550265f, 0.089726001f, -0.0897584409f, -0.255577773f, -0.382487118f, -0.451165408f},
    {-0.448822796f, -0.376087785f, -0.246095926f, -0.0786376297f, 0.100792706f, 0.264878035f, 0.388637483f, 0.453229487f, 0.448820382f, 0.376081437f, 0.246086612f, 0.0786267668f, -0.100803465f, -0.264887065f, -0.388643384f, -0.453231394f},
    {-0.446214169f, -0.369474143f, -0.236484155f, -0.0674910769f, 0.111777067f, 0.274027914f, 0.39455986f, 0.455022722f, 0.446211487f, 0.369467556f, 0.236474708f, 0.0674801692f, -0.111787759f, -0.274036765f, -0.394565523f, -0.455024362f},
    {-0.443336725f, -0.362637937f, -0.226729944f, -0.0563038662f, 0.122694097f, 0.283012718f, 0.400244564f, 0.456541896f, 0.443333805f, 0.362631142f, 0.226720348f, 0.056292925f, -0.122704722f, -0.28302139f, -0.400249988f, -0.456543237f},
    {-0.440192252f, -0.35558328f, -0.21683915f, -0.0450827405f, 0.133537218f, 0.291827023f, 0.405688167f, 0.457786024f, 0.440189064f, 0.355576277f, 0.216829434f, 0.0450717695f, -0.133547768f, -0.291835546f, -0.405693352f, -0.457787097f},
    {-0.436782598f, -0.348314434f, -0.206817746f, -0.0338344574f, 0.144299895f, 0.300465584f, 0.41088739f, 0.45875439f, 0.436779141f, 0.348307222f, 0.206807896f, 0.0338234641f, -0.14431037f, -0.300473928f, -0.410892367f, -0.458755225f},
    {-0.43310985f, -0.34083578f, -0.196671754f, -0.022565797f, 0.154975653f, 0.308923125f, 0.415839136f, 0.45944643f, 0.433106154f, 0.340828389f, 0.1966618f, 0.0225547832f, -0.154986039f, -0.308931291f, -0.415843844f, -0.459446996f},
    {-0.429176211f, -0.333151817f, -0.186407298f, -0.0112835402f, 0.16555807f, 0.317194581f, 0.420540363f, 0.459861726f, 0.429172248f, 0.333144218f, 0.186397225f, 0.011272518f, -0.165568352f, -0.317202568f, -0.420544833f, -0.459861994f},
};
__device__ constexpr float kDif16TapS[16][16] = {
    {0.0f, 0.176035658f, 0.32527107f, 0.424986154f, 0.460000008f, 0.424981952f, 0.325263262f, 0.176025465f, -1.10255714e-05f, -0.176045835f, -0.325278878f, -0.424990386f, -0.460000008f, -0.42497772f, -0.325255483f, -0.176015273f},
    {0.0112890508f, 0.186412349f, 0.333155632f, 0.429178208f, 0.459861308f, 0.420533657f, 0.3171826f, 0.165542632f, -0.0113000739f, -0.186422423f, -0.333163232f, -0.429182172f, -0.45986104f, -0.420529187f, -0.317174613f, -0.165532351f},
    {0.022571303f, 0.196676746f, 0.340839475f, 0.433111727f, 0.459445626f, 0.415832043f, 0.308910877f, 0.154960081f, -0.0225823149f, -0.196686715f, -0.340846896f, -0.433115423f, -0.459445089f, -0.415827334f, -0.308902681f, -0.15494971f},
    {0.0338399559f, 0.206822678f, 0.34831804f, 0.436784327f, 0.458753198f, 0.41087997f, 0.300453037f, 0.144284189f, -0.033850953f, -0.206832528f, -0.348325253f, -0.436787814f, -0.458752364f, -0.410874993f, -0.300444692f, -0.144273728f},
    {0.0450882278f, 0.216844022f, 0.355586767f, 0.440193862f, 0.457784414f, 0.405680358f, 0.291814238f, 0.133521393f, -0.0450991988f, -0.216853738f, -0.355593771f, -0.440197051f, -0.457783312f, -0.405675173f, -0.291805714f, -0.133510843f},
    {0.0563093387f, 0.226734743f, 0.362641305f, 0.443338215f, 0.456539869f, 0.400236398f, 0.282999665f, 0.122678153f, -0.0563202798f, -0.226744339f, -0.3626481f, -0.443341136f, -0.456538498f, -0.400230974f, -0.282990992f, -0.122667529f},
    {0.0674965307f, 0.236488894f, 0.369477421f, 0.44621551f, 0.455020308f, 0.394551367f, 0.274014622f, 0.111761026f, -0.0675074384f, -0.236498341f, -0.369483978f, -0.446218193f, -0.455018699f, -0.394545674f, -0.274005771f, -0.111750327f},
    {0.0786430612f, 0.246100575f, 0.376090944f, 0.448824018f, 0.453226656f, 0.388628632f, 0.264864504f, 0.100776568f, -0.0786539242f, -0.246109888f, -0.376097292f, -0.448826432f, -0.453224778f, -0.388622731f, -0.264855504f, -0.100765809f},
    {0.0897422209f, 0.255564034f, 0.382477939f, 0.451162159f, 0.451160014f, 0.3824718f, 0.255554855f, 0.0897314101f, -0.0897530392f, -0.255573183f, -0.382484049f, -0.451164335f, -0.451157868f, -0.38246569f, -0.255545706f, -0.0897205994f},
    {0.100787327f, 0.264873534f, 0.388634533f, 0.453228563f, 0.448821604f, 0.376084596f, 0.246091262f, 0.0786321983f, -0.100798085f, -0.264882535f, -0.388640434f, -0.453230441f, -0.44881919f, -0.376078248f, -0.246081948f, -0.0786213353f},
    {0.111771718f, 0.274023473f, 0.394557029f, 0.455021918f, 0.446212828f, 0.369470835f, 0.236479431f, 0.0674856231f, -0.111782417f, -0.274032325f, -0.394562691f, -0.455023557f, -0.446210146f, -0.369464278f, -0.236469969f, -0.0674747154f},
    {0.122688785f, 0.283008367f, 0.400241852f, 0.45654121f, 0.443335265f, 0.36263454f, 0.226725146f, 0.0562983938f, -0.12269941f, -0.283017069f, -0.400247276f, -0.456542552f, -0.443332314f, -0.362627745f, -0.22671555f, -0.0562874526f},
    {0.133531943f, 0.291822761f, 0.405685574f, 0.457785487f, 0.440190643f, 0.355579793f, 0.216834292f, 0.0450772531f, -0.133542493f, -0.291831285f, -0.405690759f, -0.45778656f, -0.440187454f, -0.35557279f, -0.216824576f, -0.0450662822f},
    {0.144294664f, 0.300461411f, 0.410884917f, 0.458754003f, 0.43678087f, 0.348310828f, 0.206812829f, 0.0338289626f, -0.14430514f, -0.300469756f, -0.410889864f, -0.458754808f, -0.436777413f, -0.348303646f, -0.206802979f, -0.0338179655f},
    {0.154970467f, 0.308919042f, 0.415836781f, 0.459446162f, 0.433108002f, 0.340832084f, 0.196666777f, 0.0225602891f, -0.154980853f, -0.308927208f, -0.41584149f, -0.459446698f, -0.433104306f, -0.340824664f, -0.196656808f, -0.0225492772f},
    {0.165552929f, 0.317190588f, 0.420538127f, 0.459861577f, 0.429174244f, 0.333148003f, 0.186402261f, 0.0112780286f, -0.165563211f, -0.317198575f, -0.420542598f, -0.459861875f, -0.429170281f, -0.333140403f, -0.186392188f, -0.0112670064f},
};

template <int FMT, class F>
__device__ __forceinline__ void dif16_front2_bfly(const Dif8Front& d, size_t frame_in, int R, unsigned char* __restrict__ smem_raw, int t, float2 (&a)[16], float2 (&a2)[16],
                                                  F&& after_first_issue) {
  static_assert(FMT == FMT_CS8 || FMT == FMT_CU8, "two-byte samples");
  const char* fb = reinterpret_cast<const char*>(d.iq) + frame_in * (size_t)d.item_stride * 2;
  const __amdgpu_buffer_rsrc_t rin = buffer_of(fb, 131072 * 2);
  const dif8_const_fp c16 = (dif8_const_fp)(uintptr_t)(d.w8 + R * 16);  // W_16^(q R), q = 0 .. 15
  const dif8_const_fp crho = (dif8_const_fp)(uintptr_t)(d.wrho + R * 16), crho2 = (dif8_const_fp)(uintptr_t)(d.wrho + (R + 8) * 16);
  const float sg = (R & 1) ? -1.0f : 1.0f;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = t & 63;
  // piece rho: run q of it at half * 16 KiB + q KiB; wave w fetches q = w and q = w + 8
  const auto issue = [&](int rho, int half) {
    if (SS_DIF_NODMA) return;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rin, (__attribute__((address_space(3))) void*)(smem_raw + half * 16384 + (8 * j + w) * 1024), 16, lane * 16,
                                               1024 * rho + 16384 * (8 * j + w), 0, SS_AUX_DIF_IQ);
  };
  issue(0, 0);
  issue(1, 1);
  after_first_issue();
  const float2 th = d.wthe[t];
  const float thx2 = th.x * sg, thy2 = th.y * sg, k2 = 0.54f * sg;  // (the samples of q + 8 enter the pair sums with the sign (-1)^R)
  const float w8x = c16[4], w8y = c16[5], w16x = c16[2], w16y = c16[3];  // W_8^R = W_16^(2 R), W_16^R
  const int rq = R & 3;
  const unsigned short* mine = reinterpret_cast<const unsigned short*>(smem_raw) + t;
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) {
    const int half = rho & 1;
    if (!SS_DIF_NODMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (rho >= 1 && rho <= 14) issue(rho + 1, half ^ 1);  // (the other half's next piece: its place was read a piece ago)
    float2 s[8];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      unsigned raw[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        raw[i] = mine[half * 8192 + 512 * (4 * sub + i)];
        raw[4 + i] = mine[half * 8192 + 512 * (4 * sub + i + 8)];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = 4 * sub + i;
        float re1, im1, re2, im2;
        dif8_convert<FMT>(raw[i], re1, im1);
        dif8_convert<FMT>(raw[4 + i], re2, im2);
        const float t1 = fmaf(kDif16TapC[rho][q], th.x, fmaf(kDif16TapS[rho][q], th.y, 0.54f));
        const float t2 = fmaf(kDif16TapC[rho][q + 8], thx2, fmaf(kDif16TapS[rho][q + 8], thy2, k2));
        s[q] = make_float2(fmaf(t2, re2, t1 * re1), fmaf(t2, im2, t1 * im1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float2 A, B, C, D;
    if (rq == 0) {
      A = make_float2(s[0].x + s[4].x, s[0].y + s[4].y);
      B = make_float2(s[2].x + s[6].x, s[2].y + s[6].y);
      C = make_float2(s[1].x + s[5].x, s[1].y + s[5].y);
      D = make_float2(s[3].x + s[7].x, s[3].y + s[7].y);
    } else if (rq == 1) {  // times -i: (x, y) -> (y, -x)
      A = make_float2(s[0].x + s[4].y, s[0].y - s[4].x);
      B = make_float2(s[2].x + s[6].y, s[2].y - s[6].x);
      C = make_float2(s[1].x + s[5].y, s[1].y - s[5].x);
      D = make_float2(s[3].x + s[7].y, s[3].y - s[7].x);
    } else if (rq == 2) {
      A = make_float2(s[0].x - s[4].x, s[0].y - s[4].y);
      B = make_float2(s[2].x - s[6].x, s[2].y - s[6].y);
      C = make_float2(s[1].x - s[5].x, s[1].y - s[5].y);
      D = make_float2(s[3].x - s[7].x, s[3].y - s[7].y);
    } else {  // times i: (x, y) -> (-y, x)
      A = make_float2(s[0].x - s[4].y, s[0].y + s[4].x);
      B = make_float2(s[2].x - s[6].y, s[2].y + s[6].x);
      C = make_float2(s[1].x - s[5].y, s[1].y + s[5].x);
      D = make_float2(s[3].x - s[7].y, s[3].y + s[7].x);
    }
    const float2 E = make_float2(fmaf(B.x, w8x, fmaf(-B.y, w8y, A.x)), fmaf(B.x, w8y, fmaf(B.y, w8x, A.y)));
    const float2 Op = make_float2(fmaf(D.x, w8x, fmaf(-D.y, w8y, C.x)), fmaf(D.x, w8y, fmaf(D.y, w8x, C.y)));
    const float2 O = make_float2(fmaf(Op.x, w16x, -(Op.y * w16y)), fmaf(Op.x, w16y, Op.y * w16x));
    a[rho] = make_float2(E.x + O.x, E.y + O.y);
    a2[rho] = make_float2(E.x - O.x, E.y - O.y);
    asm volatile("" : "+v"(a[rho].x), "+v"(a[rho].y), "+v"(a2[rho].x), "+v"(a2[rho].y));  // (formed HERE, not sunk below the next barrier)
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();  // the plane goes back to the transform
  const float2 wt = d.wt[R * 512 + t], wt2 = d.wt[(R + 8) * 512 + t];
#pragma unroll
  for (int rho = 0; rho < 16; ++rho) {
    a[rho] = cmul(a[rho], cmul(wt, make_float2(crho[2 * rho], crho[2 * rho + 1])));
    a2[rho] = cmul(a2[rho], cmul(wt2, make_float2(crho2[2 * rho], crho2[2 * rho + 1])));
  }
}

}  // namespace ss
