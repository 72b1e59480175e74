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
// The bins of residue r leave the transform in the order k' — X[8 k' + r], k' = 0 .. 8191 — and are stored that way: a frame's
// row is RESIDUE-MAJOR, eight runs of 8192 floats, bin i (DC in the middle: fft_v's shift) at (i & 7) * 8192 + (i >> 3). (The
// half rotation adds N/2 = 0 mod 8 to a bin number: it stays inside the residue and becomes the 8192-point transform's own half
// rotation.) Whoever reads such rows — the detect tiles, ss_read_window — permutes the bin number; nothing is transposed.
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

}  // namespace ss
