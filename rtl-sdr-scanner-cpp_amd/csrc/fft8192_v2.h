// fft8192_v2.h — second generation of the N = 8192 front end (BASELINE.json configs 1/2/4).
//
// Same contract and the same three register passes as the round-1 kernel (scripts/ubench/fft8192_round1.h) (Decimator + fft_v(Hamming,
// forward, shift) + PSD::work; reference sources/radio/blocks/decimator.h:15-22, sources/radio/sdr_device.cpp:164,
// sources/radio/blocks/psd.cpp:18-20). What changes is WHERE the twiddle factors come from.
//
// Workgroup time stamps of the first-generation kernel (profiles/README.md, round 2) showed that a workgroup's second and
// third pass take 7.0 and 3.1 us inside a full launch against 2.5 and 1.9 us alone on a CU: its 24 table loads per thread
// (15 x W_256 for pass 2, 9 x W_8192 / W_2048 for pass 3) are ordinary vector-memory loads, and the CU's vector-memory
// pipeline returns data in order — every table load queues behind the 64 KiB frames the other three resident workgroups
// are streaming in from HBM. Here no pass waits on vector memory after its frame has arrived:
//
//   pass 2  W_256^(m r), m = t mod 16: the 256-entry table is copied into LDS once per workgroup (2 KiB).
//   pass 3  W_8192^(j r), j = 32 w + l (w = wave, l = lane mod 32), r = 2 q + h, factored as
//             W_8192^(j (r & 3)) * W_2048^(j (r >> 2))                                  (as before), each factor split
//             W_8192^(j a)  = W_256^(w a)  * W_8192^(l a)      a  = r & 3
//             W_2048^(j q2) = W_64^(w q2)  * W_2048^(l q2)     q2 = r >> 2
//           into a wave-uniform part — scalar loads through the scalar cache into SGPRs, free VALU operands — and a
//           per-lane part that only depends on l: 12 x 32 entries in LDS (3 KiB), read with broadcast ds_read_b64.
//   The table loads are issued before the frame's own loads, so they return first.
//
// LDS: 34 KiB exchange plane + 5 KiB of tables = 39 KiB -> still four workgroups (32 waves) per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft8192_kernel.h"

// Cache policy of a 65536-point frame's samples in the radix-8 fold (fft65536_dif8.h): eight workgroups read every frame,
// so the default policy (0), not the read-once nt of the 8192-point frames
#ifndef SS_AUX_DIF_IQ
#define SS_AUX_DIF_IQ 0
#endif
#include "fft65536_dif8.h"

// Cache policy of the frame loads (read once, never again: nt = 2 measured 2.5 % faster per step than the default policy,
// sc0 / sc1 variants the same as nt) and of the dB row stores (sc1 = 16, write-through: 2.5 % faster per step in a long run, 5 % in a
// 20-step run — the end of a launch no longer has an L2 full of dirty rows to write back before the queue's next launch may start).
// Build-time so that variants can be A/B'd (scripts/build_ab.py), DESIGN.md 4.1.
#ifndef SS_AUX_IQ
#define SS_AUX_IQ 2
#endif
#ifndef SS_AUX_PSD
#define SS_AUX_PSD 16
#endif
// The per-column maxima for the tile culling (Fft8192Args::segsum): 1 = the frame's dB values go through LDS once more and sixteen
// threads per tile column take its maximum from there (16 LDS stores, five 16-byte LDS loads and ~25 vector instructions per
// thread); 0 = the first form, in registers: five v_max_f32_dpp per value and a select to gather them, 96 of the frame path's
// 1082 vector instructions — in a kernel that waits to ISSUE more than it waits for memory (profiles/README.md, SQ counters).
#ifndef SS_SEGMAX_LDS
#define SS_SEGMAX_LDS 1
#endif

namespace ss {

struct Fft8192V2Tables {
  const float2* tw2;   // [16][16]  W_256^(m r) at r*16 + m
  const float2* lane;  // [8][32] W_2048^(l q2) at q2*32 + l, then [4][32] W_8192^(l a) at 256 + a*32 + l
  const float2* wave;  // [8 waves][12]: W_64^(w q2) at q2 = 0..7, then W_256^(w a) at 8 + a, a = 0..3
  const float2* tw3a;  // first-generation tables (TW = 0, 1): [4][256] W_8192^(t r1)
  const float2* tw3b;  //                                      [8][256] W_2048^(t r2)
};

constexpr int kFft8192V2PlaneBytes = (8192 + 512) * 4;
constexpr int kFft8192V2LdsBytes = kFft8192V2PlaneBytes + (256 + 384) * 8;  // 39 936

// Host side: fills the three v2 tables (double precision, rounded once).
inline void fft8192_v2_host_tables(float2* tw2 /*256*/, float2* lane /*384*/, float2* wave /*96*/) {
  const auto W = [](double num, double den) {
    const double ang = -2.0 * 3.14159265358979323846 * num / den;
    return make_float2((float)cos(ang), (float)sin(ang));
  };
  for (int r = 0; r < 16; ++r)
    for (int m = 0; m < 16; ++m) tw2[r * 16 + m] = W((double)m * r, 256.0);
  for (int q2 = 0; q2 < 8; ++q2)
    for (int l = 0; l < 32; ++l) lane[q2 * 32 + l] = W((double)l * q2, 2048.0);
  for (int a = 0; a < 4; ++a)
    for (int l = 0; l < 32; ++l) lane[256 + a * 32 + l] = W((double)l * a, 8192.0);
  for (int w = 0; w < 8; ++w) {
    for (int q2 = 0; q2 < 8; ++q2) wave[w * 12 + q2] = W((double)w * q2, 64.0);
    for (int a = 0; a < 4; ++a) wave[w * 12 + 8 + a] = W((double)w * a, 256.0);
  }
}

struct Fft8192Args {
  const void* iq;         // frames of item_stride samples each; the first 8192 of each are transformed (Decimator)
  long long item_stride;
  const float* win;
  Fft8192V2Tables tabs;
  float db_off, scale;
  float* psd;             // [frames][8192] dB rows, DC at bin 4096
  // Optional summary for the detect stage (detect_fused.h, tile culling): for every 256-bin tile column c of every frame the
  // maximum dB value over bins [256 c - 32, 256 c + 288) — the tile and one 32-bin segment either side, which covers the
  // +-10 bins a tile's frequency means reach — at segsum[c * seg_pitch + frame]: tile-column major, so that the 36 frames a
  // detect tile averages over are 36 consecutive floats (three scalar loads). Null: not wanted.
  float* segsum;
  int seg_pitch;
  // Optional: the header word (DetectArgs::live, detect_fused.h) of the list this workgroup serves for the detect stage that
  // rides on the launch (scan_step.h). It is asked for before the frame's own loads and handed back in `hdr` when they have
  // landed: the answer costs the workgroup nothing.
  const int* live_hint;
  // (FRONT != 0 — a residue of a 65536- / 131072-point frame: `psd` is the averager ring's buffer and the rows are dB values in
  // the fold's order (blocks of 32 Q bins, fft65536_dif8.h); the tiles that are evaluated subtract the noise ceiling, DetectArgs::ring_db_from. Until session 18 of round 5
  // the transform subtracted it itself, from a copy of the ceiling in the rows' order: 32 loads per thread, 8-9 % of the launch.)
#ifdef SS_DIAG
  int hint_nowait;  // timing ablation: do not wait for the header word (garbage result)
#endif
};

// max over the 32 lanes of a half-wave (lanes 0..31 / 32..63) of FOUR values at once, valid in each half's upper 16 lanes:
// four butterfly steps inside each row of 16 (every lane of a row then holds the row's maximum), then lane 15 of rows 0 / 2
// broadcast into rows 1 / 3. v_max_f32 returns the other operand for a (quiet) NaN, so NaN bins are ignored — they cannot
// make a candidate. Written as v_max_f32_dpp by hand (the compiler emits v_mov_b32_dpp + canonicalising v_max triples, 3.5x
// the instructions); four independent chains interleaved keep a DPP read three instructions behind the write it depends on
// (the hardware wants two), the leading s_nop covers the values' producers.
__device__ __forceinline__ void halfwave_max4_hi16(float& a, float& b, float& c, float& d) {
#define SS_DPP4(ctrl)                                  \
  "v_max_f32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n" \
  "v_max_f32_dpp %1, %1, %1 " ctrl " bank_mask:0xf\n" \
  "v_max_f32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n" \
  "v_max_f32_dpp %3, %3, %3 " ctrl " bank_mask:0xf\n"
  asm volatile("s_nop 1\n"
               SS_DPP4("quad_perm:[1,0,3,2] row_mask:0xf")
               SS_DPP4("quad_perm:[2,3,0,1] row_mask:0xf")
               SS_DPP4("row_half_mirror row_mask:0xf")
               SS_DPP4("row_mirror row_mask:0xf")
               SS_DPP4("row_bcast:15 row_mask:0xa")
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef SS_DPP4
}

// Everything behind the load stage: the three register passes on a[16] (the thread's y[t + 512 r], r < 16), dB, the row's stores and
// what the frame leaves for the detect stage. `a` is consumed.
template <int FMT, int TW, bool SWZ, int FRONT>
__device__ __forceinline__ void fft8192_v2_core(float2 (&a)[16], const Fft8192Args& g, size_t frame, unsigned char* __restrict__ smem_raw, int t, int* hdr,
                                                const Dif8Front* dif, size_t frame_in, int residue) {
  float* s = reinterpret_cast<float*>(smem_raw);
  float2* tw2_l = reinterpret_cast<float2*>(smem_raw + kFft8192V2PlaneBytes);
  float2* lane_l = tw2_l + 256;
  const Fft8192V2Tables& tabs = g.tabs;
  const float db_off = g.db_off;
  float* psd = g.psd;
  float* segsum = g.segsum;
  dft16(a);
  float2 c[16];
  if constexpr (SWZ) {
    // exchange 1, unpadded: y[16 t + k] lives at word 16 t + 4 (((k >> 2) + (t >> 1)) & 3) + (k & 3). The 8 lanes of one
    // 16-byte store cycle then hit 8 different bank quads; a reader's 32 consecutive elements stay a permutation of
    // two 16-word rows.
    const int rot = (t >> 1) & 3;
    const int rd = ((t >> 4) << 4) + ((((t >> 2) & 3) + ((t >> 5) & 3)) & 3) * 4 + (t & 3);
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
      *reinterpret_cast<float4*>(&s[16 * t + 4 * ((c4 + rot) & 3)]) =
          make_float4(a[slot16(4 * c4)].x, a[slot16(4 * c4 + 1)].x, a[slot16(4 * c4 + 2)].x, a[slot16(4 * c4 + 3)].x);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r].x = s[rd + 512 * r];
    __syncthreads();
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
      *reinterpret_cast<float4*>(&s[16 * t + 4 * ((c4 + rot) & 3)]) =
          make_float4(a[slot16(4 * c4)].y, a[slot16(4 * c4 + 1)].y, a[slot16(4 * c4 + 2)].y, a[slot16(4 * c4 + 3)].y);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r].y = s[rd + 512 * r];
  } else {
    // exchange 1: y[16 t + k] at word 17 t + k
#pragma unroll
    for (int k = 0; k < 16; ++k) s[17 * t + k] = a[slot16(k)].x;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int e = t + 512 * r;
      c[r].x = s[e + (e >> 4)];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) s[17 * t + k] = a[slot16(k)].y;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int e = t + 512 * r;
      c[r].y = s[e + (e >> 4)];
    }
  }
  // ---------------- pass 2: radix 16, Ns = 16, butterfly j = t ----------------
  {
    const int m = t & 15;
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      float2 w;
      if constexpr (TW == 0) w = tabs.tw2[r * 16 + m];
      else if constexpr (TW == 3) w = make_float2(__int_as_float(0x3f800000 + m * r), 0.25f);
      else w = tw2_l[r * 16 + m];
      c[r] = cmul(c[r], w);
    }
  }
  dft16(c);
  __syncthreads();  // every read of y is done before z overwrites the plane
  // exchange 2: z[(t/16)*256 + t%16 + 16 k]; pass 3 lane (w, l) reads z[j + 256 (2q + h)], j = 32 w + (l & 31), h = l >> 5
  const int zbase = ((t >> 4) << 8) + (t & 15);
  const int lane = t & 63;
  const int h = lane >> 5;
  const int lam = lane & 31;
  const int j = ((t >> 6) << 5) + lam;
  const int rbase = j + 256 * h;
#pragma unroll
  for (int k = 0; k < 16; ++k) s[zbase + 16 * k] = c[slot16(k)].x;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q].x = s[rbase + 512 * q];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 16; ++k) s[zbase + 16 * k] = c[slot16(k)].y;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 16; ++q) a[q].y = s[rbase + 512 * q];
#if SS_SEGMAX_LDS
  __syncthreads();  // the exchange plane takes the frame's dB values at the end (the column maxima): every read of z is done
#endif

  // The list header word this workgroup will want when its frame is done (live_hint): asked for NOW — two thirds of a
  // frame after the workgroup began, so the plan role has had time to publish its counts, and with pass 3 left to hide the
  // round trip behind — by the workgroup's first wave, straight into LDS (LDS-DMA: no register has to hold it through pass 3),
  // with the sc0 sc1 policy bits: it must see what another XCD wrote during this launch. (Asked for before the frame's own
  // loads it stood in front of them in the in-order return queue, and most workgroups were told "not ready yet" and had to
  // ask again; asked for by all eight waves it was eight requests per workgroup to one memory channel.)
  float* hint_slot = s + 8192 + 256;  // (behind the segment maxima; the pad is idle since exchange 1)
  if (g.live_hint && t == 0)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(buffer_of(g.live_hint, 4), (__attribute__((address_space(3))) void*)hint_slot, 4, 0, 0, 0, 17);
  // ---------------- pass 3: radix 32, Ns = 256, butterfly j shared by lanes l and l + 32 ----------------
  // twiddle of input r = 2q + h:  W_8192^(j r) = W_8192^(j (r & 3)) * W_2048^(j (r >> 2)),  r & 3 = 2 (q & 1) + h,  r >> 2 = q >> 1
  if constexpr (TW == 2) {
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    // wave-uniform and never written: the constant address space makes these scalar loads whatever else the kernel does
    typedef const __attribute__((address_space(4))) float* const_fp;
    const_fp swf = (const_fp)(uintptr_t)(tabs.wave + w * 12);
    const auto sw = [swf](int i) { return make_float2(swf[2 * i], swf[2 * i + 1]); };
    // (all four fetched, then selected per lane: a select between the two ADDRESSES would make them vector loads)
    const float2 s8 = sw(8), s9 = sw(9), s10 = sw(10), s11 = sw(11);
    const float2 sa0 = make_float2(h ? s9.x : s8.x, h ? s9.y : s8.y);      // W_256^(w a), a = h
    const float2 sa1 = make_float2(h ? s11.x : s10.x, h ? s11.y : s10.y);  //              a = 2 + h
    const float2 wa0 = cmul(lane_l[256 + h * 32 + lam], sa0);
    const float2 wa1 = cmul(lane_l[256 + (2 + h) * 32 + lam], sa1);
    a[0] = cmul(a[0], wa0);
    a[1] = cmul(a[1], wa1);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) {
      const float2 wb = cmul(lane_l[q2 * 32 + lam], sw(q2));
      a[2 * q2] = cmul(a[2 * q2], cmul(wa0, wb));
      a[2 * q2 + 1] = cmul(a[2 * q2 + 1], cmul(wa1, wb));
    }
  } else {
    float2 wa0, wa1;
    if constexpr (TW == 3) {
      wa0 = make_float2(__int_as_float(0x3f800000 + j + h), 0.5f);
      wa1 = make_float2(__int_as_float(0x3f800000 + 2 * j + h), 0.75f);
    } else {
      wa0 = tabs.tw3a[h * 256 + j];        // r & 3 = h      (h = 0: W^0 = 1)
      wa1 = tabs.tw3a[(2 + h) * 256 + j];  // r & 3 = 2 + h
    }
    a[0] = cmul(a[0], wa0);
    a[1] = cmul(a[1], wa1);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) {
      float2 wb;
      if constexpr (TW == 3) wb = make_float2(__int_as_float(0x3f800000 + j * q2), 0.125f);
      else wb = tabs.tw3b[q2 * 256 + j];
      a[2 * q2] = cmul(a[2 * q2], cmul(wa0, wb));
      a[2 * q2 + 1] = cmul(a[2 * q2 + 1], cmul(wa1, wb));
    }
  }
  dft16(a);  // A_h[k] in slot16(k)
  const bool odd = h != 0;
  float2 u[16];  // u[k] = A_even[k] on the low half-wave, W_32^k A_odd[k] on the high half-wave
  u[0] = a[slot16(0)];
  u[1] = mulw32_if<1>(a[slot16(1)], odd);
  u[2] = mulw32_if<2>(a[slot16(2)], odd);
  u[3] = mulw32_if<3>(a[slot16(3)], odd);
  u[4] = mulw32_if<4>(a[slot16(4)], odd);
  u[5] = mulw32_if<5>(a[slot16(5)], odd);
  u[6] = mulw32_if<6>(a[slot16(6)], odd);
  u[7] = mulw32_if<7>(a[slot16(7)], odd);
  u[8] = mulw32_if<8>(a[slot16(8)], odd);
  u[9] = mulw32_if<9>(a[slot16(9)], odd);
  u[10] = mulw32_if<10>(a[slot16(10)], odd);
  u[11] = mulw32_if<11>(a[slot16(11)], odd);
  u[12] = mulw32_if<12>(a[slot16(12)], odd);
  u[13] = mulw32_if<13>(a[slot16(13)], odd);
  u[14] = mulw32_if<14>(a[slot16(14)], odd);
  u[15] = mulw32_if<15>(a[slot16(15)], odd);
  // FRONT != 0: `frame` = Q * (row of the ring) + residue, and the row is in the fold's layout (fft65536_dif8.h: blocks of 32 Q bins,
  // 32 outputs of every residue side by side) — this thread's output k' = j + 2048 h + 256 kk (+ 4096) at block k' / 32, place j % 32
  constexpr int LOGQ = (FRONT == 4 || FRONT == 6) ? 4 : 3;
  float* out = FRONT != 0 ? psd + (((frame >> LOGQ) << LOGQ) * 8192 + (frame & ((1 << LOGQ) - 1)) * 32) : psd + frame * 8192;
  const __amdgpu_buffer_rsrc_t rout = buffer_of(out, FRONT != 0 ? ((8192 << LOGQ) - 32 * (int)(frame & ((1 << LOGQ) - 1))) * 4 : 8192 * 4);
  const int voff = (j + 2048 * h) * 4;
  constexpr int kOutStep = FRONT != 0 ? (256 << LOGQ) * 4 : 1024;  // bytes between outputs 256 apart: eight blocks / 256 floats
  // Per-segment maxima for the detect stage's tile culling: the 32 lanes of a half-wave hold, for every (k, s), the 32
  // consecutive bins of segment w + 8 k + 64 h + 128 s; lane 16 + i of each half keeps the maximum of value i = 2 k + s.
  // The 256 segment maxima of the frame meet in LDS (the 512 floats behind the exchange plane, idle since exchange 1) and the
  // first 32 threads turn them into the 32 tile-column maxima the detect stage reads.
  *hdr = 0;
#ifdef SS_DIAG
  const bool hint_wait = !g.hint_nowait;
#else
  const bool hint_wait = true;
#endif
  // the first wave sees its header word land (nothing else of it is in flight here: the frame landed long ago, the dB stores
  // come below); the others read it behind the barrier at the bottom
  if (g.live_hint && hint_wait && t < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const bool want_max = segsum != nullptr;  // (workgroup-uniform)
  float mine;
#if SS_SEGMAX_LDS
  float* dbrow = reinterpret_cast<float*>(smem_raw + voff);  // the frame's dB values at their bin numbers (the exchange plane is idle since pass 3 began)
#endif
  // the stores' per-thread offset: voff itself, or for the fold's rows block (j + 2048 h) / 32, place j % 32 — (j + 2048 h) * 4 has
  // j % 32 in bits 2-6, j / 32 in bits 7-9 and h in bit 13
  int gvoff = voff;
  if constexpr (FRONT != 0) gvoff = (voff & 0x7c) | ((voff & 0x2380) << LOGQ);
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) {
    float pv[4];  // pv[2 i + s]: the dB value of bin j + 2048 h + 256 (2 k2 + i) + 4096 s
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int k = 2 * k2 + i;
      // v_permlane32_swap(vdst, src): lanes 32..63 of vdst <-> lanes 0..31 of src. With vdst = u[k], src = u[k+8]:
      //   low half:  (e, o) = (own u[k] = A_even[k],          partner's u[k]   = W^k A_odd[k])
      //   high half: (e, o) = (partner's u[k+8] = A_even[k+8], own u[k+8]      = W^(k+8) A_odd[k+8])
      const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(u[k].x), __float_as_uint(u[k + 8].x), false, false);
      const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(u[k].y), __float_as_uint(u[k + 8].y), false, false);
      const float2 e = make_float2(__uint_as_float(sx[0]), __uint_as_float(sy[0]));
      const float2 o = make_float2(__uint_as_float(sx[1]), __uint_as_float(sy[1]));
      // this lane's output index kk = k + 8 h; bin0 = j + 256 kk < 4096: the half rotation (fft_v shift = true) sends
      // X[kk] to bin0 + 4096 and X[kk + 16] to bin0
      pv[2 * i + 1] = psd_db(cadd(e, o), db_off);
      pv[2 * i] = psd_db(csub(e, o), db_off);
      buffer_store_f1<SS_AUX_PSD>(rout, gvoff, kOutStep * k + 16 * kOutStep, pv[2 * i + 1]);
      buffer_store_f1<SS_AUX_PSD>(rout, gvoff, kOutStep * k, pv[2 * i]);
#if SS_SEGMAX_LDS
      // (whether or not the frame leaves a summary: a branch here costs the dB stores their interleaving and the kernel registers)
      dbrow[256 * k + 4096] = pv[2 * i + 1];
      dbrow[256 * k] = pv[2 * i];
#endif
    }
    if (want_max && !SS_SEGMAX_LDS) {
      halfwave_max4_hi16(pv[0], pv[1], pv[2], pv[3]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // value index 2 k + s = 4 k2 + q goes to lanes 16 + index and 48 + index (a constant lane mask: no compare)
        const unsigned long long keep = (1ull << (16 + 4 * k2 + q)) | (1ull << (48 + 4 * k2 + q));
        if (k2 == 0 && q == 0) mine = pv[0];  // (every lane: no initial value that would have to live through the frame loop)
        else asm("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(mine) : "v"(pv[q]), "s"(keep));
      }
    }
  }
  if constexpr (FRONT != 0) {
    // ---- a residue of a 65536-point frame: the maxima of its 256 runs of 32 bins, two threads per run (fft65536_dif8.h) ----
    static_assert(SS_SEGMAX_LDS, "the residue rows' maxima come out of the LDS image of the row");
    (void)mine;
    (void)want_max;
    __syncthreads();  // the row's dB values are all in the plane
    if (dif->zero_word && frame_in == 0 && residue == 0 && t == 0) *dif->zero_word = 0;
    if (dif->smax) {  // (workgroup-uniform)
      int vv = voff;
      asm volatile("" : "+v"(vv));
      const int tt = ((vv >> 1) & 0x1c0) | ((vv >> 8) & 32) | ((vv >> 2) & 31);  // the thread number, out of the one per-thread value alive here (below)
      const int run = tt >> 1;
      const float4* half_run = reinterpret_cast<const float4*>(s + 16 * tt);  // bins [32 run + 16 (tt & 1), + 16) of the row
      float m = -__builtin_inff();  // (fmaxf ignores a NaN operand like v_max_f32 does: NaN bins cannot make a candidate)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = half_run[(q + run) & 3];  // (rotated by the run number: the lanes of a 16-lane group meet on two banks, not four)
        m = fmaxf(fmaxf(m, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
      }
      asm volatile("s_nop 1\n"
                   "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                   : "+v"(m));
      if ((tt & 1) == 0) {
        const __amdgpu_buffer_rsrc_t rs = buffer_of(dif->smax + (size_t)((dif->abs0 + (int)frame_in) & dif->smax_mask) * (size_t)dif->smax_pitch, dif->smax_pitch * 4);
        buffer_store_f1(rs, run * 4, residue * 1024, m);
      }
    }
    return;
  }
#if SS_SEGMAX_LDS
  (void)mine;
  if (want_max) {
    // Tile column c = t / 16 by sixteen threads: bins [256 c - 32, 256 c + 288) — the column and one 32-bin segment either side,
    // clipped at the band's edges — in sixteen runs of 20 (a run that would start outside the band starts at its edge instead: it
    // reads bins of the same column's range once more). Lanes of a column are 20 words apart: sixteen 16-byte loads hit 64 banks.
    __syncthreads();
    if (g.live_hint) *hdr = __builtin_amdgcn_readfirstlane(__float_as_int(*hint_slot));
    // the thread number comes back out of the PSD stores' own offset (j + 2048 h) * 4, j = 32 w + lane mod 32 — the one per-thread
    // value that is alive here anyway; anything else kept alive through pass 3 for this gets spilled (64 VGPRs)
    int vv = voff;
    asm volatile("" : "+v"(vv));
    const int tt = ((vv >> 1) & 0x1c0) | ((vv >> 8) & 32) | ((vv >> 2) & 31);  // 64 w + 32 h + lane mod 32
    const int c = tt >> 4;
    const int start = min(max(256 * c - 32 + 20 * (tt & 15), 0), 8192 - 20);
    const float4* run = reinterpret_cast<const float4*>(s + start);
    float m = -__builtin_inff();  // (fmaxf ignores a NaN operand like v_max_f32 does: NaN bins cannot make a candidate)
#pragma unroll
    for (int q = 0; q < 5; ++q) {
      const float4 v = run[q];
      m = fmaxf(fmaxf(m, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
    }
    // the column's sixteen lanes are one row of the wave: four butterfly steps leave the row's maximum in every lane
    asm volatile("s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n"
                 : "+v"(m));
    if ((tt & 15) == 0) {
      const __amdgpu_buffer_rsrc_t rseg = buffer_of(segsum, 32 * g.seg_pitch * 4);
      buffer_store_f1(rseg, c * g.seg_pitch * 4, (int)frame * 4, m);
    }
  } else if (g.live_hint) {
    __syncthreads();
    *hdr = __builtin_amdgcn_readfirstlane(__float_as_int(*hint_slot));
  }
#else
  if (want_max) {
    // wave, half and lane come back out of the PSD stores' own offset (j + 2048 h) * 4, j = 32 w + lane mod 32 — the one
    // per-thread value that is alive here anyway; anything else kept alive through pass 3 for this gets spilled (64 VGPRs)
    int vv = voff;
    asm volatile("" : "+v"(vv));
    float* segs = s + 8192;
    const int i = (vv >> 2) & 15;
    if (vv & 64) segs[((vv >> 7) & 7) + 8 * (i >> 1) + 64 * (vv >> 13) + 128 * (i & 1)] = mine;
    __syncthreads();
    if (g.live_hint) *hdr = __builtin_amdgcn_readfirstlane(__float_as_int(*hint_slot));
    if (vv < 128) {  // wave 0, lower half: thread c = tile column
      const int c = vv >> 2;
      float m = segs[max(8 * c - 1, 0)];
#pragma unroll
      for (int q = 0; q < 9; ++q) m = fmaxf(m, segs[min(8 * c + q, 255)]);
      // block-uniform base and frame offset in scalar registers, one per-lane offset (no 64-bit address arithmetic in the vector pipe)
      const __amdgpu_buffer_rsrc_t rseg = buffer_of(segsum, 32 * g.seg_pitch * 4);
      buffer_store_f1(rseg, c * g.seg_pitch * 4, (int)frame * 4, m);
    }
  } else if (g.live_hint) {  // (frames without a summary row — the re-transformed halo frames — still serve a list)
    __syncthreads();
    *hdr = __builtin_amdgcn_readfirstlane(__float_as_int(*hint_slot));
  }
#endif
}


// TW: 0 = every table from global memory (first generation), 1 = pass-2 table in LDS, 2 = all tables in LDS / SGPRs,
//     3 = no tables at all (timing bound only: the output is meaningless).
// SWZ: exchange 1 as four 16-byte LDS stores per plane into an unpadded, quad-rotated image instead of sixteen 4-byte
//      stores into the 17-word pitch.
// One frame by one workgroup of 512 threads; `smem_raw` = kFft8192V2LdsBytes of LDS, `t` = threadIdx.x.
// FRONT: 0 = an 8192-point frame of its own; 1, 2 = residue `residue` of the 65536-point frame `frame_in` of `dif` (fft65536_dif8.h,
//        LOADV = FRONT - 1: the load stage is the radix-8 fold, g.iq / g.win / g.item_stride are not read), `frame` = the row
//        the residue's bins go to, as 8 x (the ring's row) + residue (the bins land in that row's blocks, fft65536_dif8.h); 3 = residues `residue` (< 4)
//        AND residue + 4 by the same workgroup, one fold for both, rows `frame` and `frame` + 4 (twice the registers: four waves per SIMD);
//        4 = the same for a 131072-point frame: radix 16, residues `residue` (< 8) and residue + 8, rows `frame` and `frame` + 8;
//        5 = 3 with the fold as a radix-8 butterfly per point (round 6, fft65536_dif8.h: dif8_front2_bfly); 6 = 4 likewise (dif16_front2_bfly).
template <int FMT, int TW, bool SWZ = false, bool NOWIN = false, int FRONT = 0>
__device__ __forceinline__ void fft8192_v2_frame(const Fft8192Args& g, size_t frame, unsigned char* __restrict__ smem_raw, int t, int* hdr,
                                                 const Dif8Front* dif = nullptr, size_t frame_in = 0, int residue = 0) {
  float2* tw2_l = reinterpret_cast<float2*>(smem_raw + kFft8192V2PlaneBytes);
  float2* lane_l = tw2_l + 256;
  const Fft8192V2Tables& tabs = g.tabs;
  const void* iq = g.iq;
  const float* win = g.win;
  const float scale = g.scale;
  const size_t in_base = frame * (size_t)g.item_stride;

  // ---- tables first: these loads are ahead of the frame's own in the vector-memory queue ----
  float2 tf0 = make_float2(0.f, 0.f), tf1 = make_float2(0.f, 0.f);
  if constexpr (FRONT != 0) {
    static_assert(TW == 2, "the fold keeps the transform's tables in LDS");
  } else if constexpr (TW == 1) {
    if (t < 256) tf0 = tabs.tw2[t];
  } else if constexpr (TW == 2) {
    tf0 = t < 256 ? tabs.tw2[t] : tabs.lane[t - 256];
    if (t < 128) tf1 = tabs.lane[256 + t];
  }

  // ---------------- pass 1: radix 16, Ns = 1, butterfly j = t ----------------
  float2 a[16];
  [[maybe_unused]] float2 a2[16];  // FRONT = 3: the second residue's points
  if constexpr (FRONT != 0) {
    // (the tables go to LDS at once — their place behind the exchange plane is free — instead of through registers that would
    // have to live through the fold)
    const auto tables_to_lds = [&]() {
      tw2_l[t] = t < 256 ? tabs.tw2[t] : tabs.lane[t - 256];
      if (t < 128) lane_l[256 + t] = tabs.lane[256 + t];
    };
    if constexpr (FRONT == 3) dif8_front2<FMT, 8>(*dif, frame_in, residue, smem_raw, t, a, a2, tables_to_lds);
    else if constexpr (FRONT == 5) dif8_front2_bfly<FMT>(*dif, frame_in, residue, smem_raw, t, a, a2, tables_to_lds);
    else if constexpr (FRONT == 6) dif16_front2_bfly<FMT>(*dif, frame_in, residue, smem_raw, t, a, a2, tables_to_lds);
    else if constexpr (FRONT == 4) dif8_front2<FMT, 16>(*dif, frame_in, residue, smem_raw, t, a, a2, tables_to_lds);
    else dif8_front<FMT, FRONT - 1>(*dif, frame_in, residue, smem_raw, t, a, tables_to_lds);
    (void)iq;
    (void)win;
    (void)scale;
    (void)in_base;
  } else {
    constexpr int kSample = FMT == FMT_CF32 ? 8 : 2;  // bytes per IQ sample
    const __amdgpu_buffer_rsrc_t rin = buffer_of(reinterpret_cast<const char*>(iq) + in_base * kSample, 8192 * kSample);
    const __amdgpu_buffer_rsrc_t rwin = buffer_of(win, 8192 * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float2 x;
      if constexpr (FMT == FMT_CF32) {
        x = buffer_load_f2<SS_AUX_IQ>(rin, t * 8, 4096 * r);
      } else {
        const unsigned short raw = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, t * 2, 1024 * r, SS_AUX_IQ);
        if constexpr (FMT == FMT_CS8) x = make_float2((float)(signed char)(raw & 0xff) * scale, (float)(signed char)(raw >> 8) * scale);
        else x = make_float2(((float)(raw & 0xff) - 127.5f) * scale, ((float)(raw >> 8) - 127.5f) * scale);
      }
      if constexpr (NOWIN) {
        a[r] = x;
      } else {
        const float w = buffer_load_f1(rwin, t * 4, 2048 * r);
        a[r] = make_float2(x.x * w, x.y * w);  // volk_32fc_32f_multiply_32fc
      }
    }
  }
  if constexpr (FRONT != 0) {
    // (written by the fold's prologue)
  } else if constexpr (TW == 1) {
    if (t < 256) tw2_l[t] = tf0;
  } else if constexpr (TW == 2) {
    tw2_l[t] = tf0;  // tw2_l and lane_l are contiguous: entries 0..511
    if (t < 128) lane_l[256 + t] = tf1;
  }
  if constexpr (FRONT == 3 || FRONT == 4 || FRONT == 5 || FRONT == 6) {
    // two residues by one workgroup (fft65536_dif8.h, dif8_front2): r's transform, then (r + Q/2)'s from the registers that kept it
    constexpr int HQ = (FRONT == 4 || FRONT == 6) ? 8 : 4;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      if (pass) {
        __syncthreads();  // the first residue's last reads of the plane are done
#pragma unroll
        for (int r = 0; r < 16; ++r) a[r] = a2[r];
      }
      fft8192_v2_core<FMT, TW, SWZ, FRONT>(a, g, frame + HQ * pass, smem_raw, t, hdr, dif, frame_in, residue + HQ * pass);
    }
  } else {
    fft8192_v2_core<FMT, TW, SWZ, FRONT>(a, g, frame, smem_raw, t, hdr, dif, frame_in, residue);
  }
}

}  // namespace ss
