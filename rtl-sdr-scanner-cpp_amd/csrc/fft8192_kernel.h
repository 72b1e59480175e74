// fft8192_kernel.h — butterflies and addressing helpers of the N = 8192 front end, plus the ROUND-1 kernel
// k_fft8192_psd_w8 kept as the reference point of scripts/ubench/fft8192_lab (and its memory-only / transform-only
// ablations, which compute garbage by design). The product does not launch anything from this file: the 8192-point
// transform that ships is fft8192_v2_frame (fft8192_v2.h), a role of k_scan_step (scan_step.h).
//
// Same contract as k_fft_psd_lds in fft_kernels.h (Decimator + fft_v(Hamming, forward, shift) + PSD::work,
// reference sources/radio/blocks/decimator.h:15-22, sources/radio/sdr_device.cpp:164,
// sources/radio/blocks/psd.cpp:18-20).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fft_kernels.h"

namespace ss {

// W_32^k = exp(-2 pi i k / 32), first octant is enough; the rest comes from symmetry in mulw32.
__device__ constexpr float kC32[5] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654757f};
__device__ constexpr float kS32[5] = {0.0f, 0.19509032201612825f, 0.38268343236508978f, 0.55557023301960218f, 0.70710678118654757f};

// x * W_32^K for a compile-time K, with the trivial rotations free.
template <int K>
__device__ __forceinline__ float2 mulw32(float2 x) {
  constexpr int k = ((K % 32) + 32) % 32;
  if constexpr (k == 0) return x;
  else if constexpr (k == 8) return make_float2(x.y, -x.x);
  else if constexpr (k == 16) return make_float2(-x.x, -x.y);
  else if constexpr (k == 24) return make_float2(-x.y, x.x);
  else {
    // W = c - i s with (c, s) folded into the first octant
    constexpr int q = k / 8;       // quadrant
    constexpr int r = k % 8;       // 1..7 inside the quadrant
    constexpr float c0 = r <= 4 ? kC32[r] : kS32[8 - r];
    constexpr float s0 = r <= 4 ? kS32[r] : kC32[8 - r];
    // rotate (c0 - i s0) by (-i)^q
    constexpr float c = q == 0 ? c0 : q == 1 ? -s0 : q == 2 ? -c0 : s0;
    constexpr float s = q == 0 ? -s0 : q == 1 ? -c0 : q == 2 ? s0 : c0;  // imaginary part of W
    return make_float2(fmaf(x.x, c, -(x.y * s)), fmaf(x.x, s, x.y * c));  // explicit, like cmul
  }
}

__device__ __forceinline__ void dft2(float2& a, float2& b) {
  const float2 t = a;
  a = cadd(t, b);
  b = csub(t, b);
}

// natural order in, natural order out
__device__ __forceinline__ void dft4(float2& a0, float2& a1, float2& a2, float2& a3) {
  const float2 s0 = cadd(a0, a2), s1 = csub(a0, a2), s2 = cadd(a1, a3), s3 = cmul_mi(csub(a1, a3));
  a0 = cadd(s0, s2);
  a1 = cadd(s1, s3);
  a2 = csub(s0, s2);
  a3 = csub(s1, s3);
}

// 8-point DFT, inputs x[n] in v0..v7; output X[k] ends up in slot 2*(k&3) + (k>>2)
__device__ __forceinline__ constexpr int slot8(int k) { return 2 * (k & 3) + (k >> 2); }
__device__ __forceinline__ void dft8(float2& v0, float2& v1, float2& v2, float2& v3, float2& v4, float2& v5, float2& v6, float2& v7) {
  dft4(v0, v2, v4, v6);  // n2 = 0: A[0][k1] at v[2 k1]
  dft4(v1, v3, v5, v7);  // n2 = 1: A[1][k1] at v[2 k1 + 1]
  v3 = mulw32<4>(v3);    // W_8^k1
  v5 = mulw32<8>(v5);
  v7 = mulw32<12>(v7);
  dft2(v0, v1);  // X[k1 + 4 k2] at v[2 k1 + k2]
  dft2(v2, v3);
  dft2(v4, v5);
  dft2(v6, v7);
}

// 16-point DFT in registers; X[k] ends up in slot 4*(k&3) + (k>>2)
__device__ __forceinline__ constexpr int slot16(int k) { return 4 * (k & 3) + (k >> 2); }
__device__ __forceinline__ void dft16(float2 (&v)[16]) {
  dft4(v[0], v[4], v[8], v[12]);  // A[n2][k1] at v[n2 + 4 k1]
  dft4(v[1], v[5], v[9], v[13]);
  dft4(v[2], v[6], v[10], v[14]);
  dft4(v[3], v[7], v[11], v[15]);
  // W_16^(n2 k1) = W_32^(2 n2 k1)
  v[5] = mulw32<2>(v[5]);
  v[6] = mulw32<4>(v[6]);
  v[7] = mulw32<6>(v[7]);
  v[9] = mulw32<4>(v[9]);
  v[10] = mulw32<8>(v[10]);
  v[11] = mulw32<12>(v[11]);
  v[13] = mulw32<6>(v[13]);
  v[14] = mulw32<12>(v[14]);
  v[15] = mulw32<18>(v[15]);
  dft4(v[0], v[1], v[2], v[3]);  // X[k1 + 4 k2] at v[4 k1 + k2]
  dft4(v[4], v[5], v[6], v[7]);
  dft4(v[8], v[9], v[10], v[11]);
  dft4(v[12], v[13], v[14], v[15]);
}

// 32-point DFT in registers: n = n2 + 4 n1; X[k] ends up in slot (k>>3) + 4*slot8(k&7)
__device__ __forceinline__ constexpr int slot32(int k) { return (k >> 3) + 4 * slot8(k & 7); }
template <int N2>
__device__ __forceinline__ void dft32_twiddle_row(float2 (&v)[32]) {
  // v[N2 + 4*slot8(k1)] *= W_32^(N2 k1), k1 = 1..7
  v[N2 + 4 * slot8(1)] = mulw32<N2 * 1>(v[N2 + 4 * slot8(1)]);
  v[N2 + 4 * slot8(2)] = mulw32<N2 * 2>(v[N2 + 4 * slot8(2)]);
  v[N2 + 4 * slot8(3)] = mulw32<N2 * 3>(v[N2 + 4 * slot8(3)]);
  v[N2 + 4 * slot8(4)] = mulw32<N2 * 4>(v[N2 + 4 * slot8(4)]);
  v[N2 + 4 * slot8(5)] = mulw32<N2 * 5>(v[N2 + 4 * slot8(5)]);
  v[N2 + 4 * slot8(6)] = mulw32<N2 * 6>(v[N2 + 4 * slot8(6)]);
  v[N2 + 4 * slot8(7)] = mulw32<N2 * 7>(v[N2 + 4 * slot8(7)]);
}
__device__ __forceinline__ void dft32(float2 (&v)[32]) {
  dft8(v[0], v[4], v[8], v[12], v[16], v[20], v[24], v[28]);   // n2 = 0: A[0][k1] at v[0 + 4 slot8(k1)]
  dft8(v[1], v[5], v[9], v[13], v[17], v[21], v[25], v[29]);
  dft8(v[2], v[6], v[10], v[14], v[18], v[22], v[26], v[30]);
  dft8(v[3], v[7], v[11], v[15], v[19], v[23], v[27], v[31]);
  dft32_twiddle_row<1>(v);
  dft32_twiddle_row<2>(v);
  dft32_twiddle_row<3>(v);
#pragma unroll
  for (int s = 0; s < 8; ++s) dft4(v[4 * s], v[4 * s + 1], v[4 * s + 2], v[4 * s + 3]);  // X[k1 + 8 k2] at v[k2 + 4 slot8(k1)]
}

// twiddle tables for this kernel (built on the host in double precision, see specscan.hip):
//   tw2[r*16 + m]   = W_256^(m r)        r = 0..15, m = 0..15
//   tw3a[r1*256 + t] = W_8192^(t r1)     r1 = 0..3
//   tw3b[r2*256 + t] = W_2048^(t r2)     r2 = 0..7
// Raw buffer resources (base in scalar registers, one 32-bit per-thread byte offset, a scalar or immediate offset per
// access): the 16 + 16 loads and 16 stores of a thread then need no address arithmetic in the vector pipe at all — with
// flat global addressing the compiler builds a 64-bit address per access whose offset exceeds the 12-bit immediate.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);  // raw, 32-bit data format (gfx9)
}
// AUX: cache policy bits of the access (gfx94x/gfx950: 1 = sc0, 2 = nt, 16 = sc1); 0 = the default policy.
template <int AUX = 0>
__device__ __forceinline__ float2 buffer_load_f2(__amdgpu_buffer_rsrc_t r, int voffset_bytes, int soffset_bytes) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voffset_bytes, soffset_bytes, AUX);
  return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}
__device__ __forceinline__ float buffer_load_f1(__amdgpu_buffer_rsrc_t r, int voffset_bytes, int soffset_bytes) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voffset_bytes, soffset_bytes, 0));
}
template <int AUX = 0>
__device__ __forceinline__ void buffer_store_f1(__amdgpu_buffer_rsrc_t r, int voffset_bytes, int soffset_bytes, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voffset_bytes, soffset_bytes, AUX);
}

struct Fft8192Tables {
  const float2* tw2;
  const float2* tw3a;
  const float2* tw3b;
  long long* unused;  // (round 1: debug stamps)
};

// =================================================================================================
// Eight-wave kernel (round 1): 512 threads, 16 points per thread. Half the per-thread work of the four-wave
// variant, 7 KiB of code instead of 14 (instruction fetch falls off a cliff between 8 and 16 KiB of hot code,
// scripts/ubench/ifetch2), 64 VGPRs and 34 KiB of LDS: four frames and 32 waves per CU.
//
//   pass 1  radix 16, Ns = 1    thread t: butterfly j = t            <- 16 x 8-byte global loads
//           y[16 j + k]                      -> LDS plane, 17-word pitch per thread (conflict-free)
//   pass 2  radix 16, Ns = 16   thread t: butterfly j = t, twiddle W_256^((t%16) r)
//           z[(t/16) 256 + t%16 + 16 k]      -> LDS plane, linear
//   pass 3  radix 32, Ns = 256  one butterfly j per LANE PAIR (l, l+32) of a wave: lane half h holds the
//           inputs r = 2q + h, runs a 16-point DFT on them, the odd half applies W_32^k, and eight
//           v_permlane32_swap_b32 per component bring A_even[k] and W^k A_odd[k] together:
//           X[k] = A_even[k] + W_32^k A_odd[k],  X[k+16] = A_even[k] - W_32^k A_odd[k]
// LDS exchanges move one fp32 plane at a time (real parts, then imaginary parts): 34 KiB per workgroup.
// =================================================================================================
constexpr int kFft8192W8LdsBytes = (8192 + 512) * 4;

template <int K>
__device__ __forceinline__ float2 mulw32_if(float2 x, bool on) {
  const float2 y = mulw32<K>(x);
  return on ? y : x;
}

// ABLATE (diagnostic, SS_FFT_ABLATE): 1 = memory traffic only (same loads and store count, no transform), 3 / 4 = the same
// bytes with 16-byte stores / 16-byte loads and stores (19.5 / 20.0 / 16.7 us per 1024 frames), 2 = transform
// only (no global loads, stores never execute). Measured at 1024 / 4096 frames per launch: full 26.1 / 90.7 us,
// memory only 19.2 / 72.3 us, transform only 16.7 / 48.6 us — see DESIGN.md.
// TWO (round 1, lab only): frames [0, split) come from `iq` / go to `psd` as usual, frames >= split from a second source to a second
// plane (a lane re-scans the halo it kept and scans the caller's batch in one launch).
struct Fft8192Second {
  const void* iq;
  long long item_stride;
  float* psd;
  int split;
};

template <int FMT, int WAVES_PER_SIMD, bool DBG = false, int ABLATE = 0, bool TWO = false>
__global__ __launch_bounds__(512, WAVES_PER_SIMD) void k_fft8192_psd_w8(const void* __restrict__ iq_a, long long item_stride,
                                                                          const float* __restrict__ win, Fft8192Tables tabs, float db_off,
                                                                          float scale, float* __restrict__ psd_a, Fft8192Second second) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* s = reinterpret_cast<float*>(smem_raw);
  const int t = threadIdx.x;
  size_t frame = blockIdx.x;
  const void* iq = iq_a;
  float* psd = psd_a;
  if constexpr (TWO) {
    if ((int)blockIdx.x >= second.split) {  // block-uniform
      frame = blockIdx.x - (size_t)second.split;
      iq = second.iq;
      item_stride = second.item_stride;
      psd = second.psd;
    }
  }
  const size_t in_base = frame * (size_t)item_stride;

  if constexpr (ABLATE == 3 || ABLATE == 4) {
    // what the same bytes cost with wider accesses: 3 = the kernel's own 8-byte loads + 16-byte stores, 4 = 16-byte loads too
    float acc[16];
    if constexpr (ABLATE == 3) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float2 x = load_iq<FMT>(iq, in_base + t + 512 * r, scale);
        acc[r] = x.x + x.y * win[t + 512 * r];
      }
    } else {
      const float4* src = reinterpret_cast<const float4*>(reinterpret_cast<const float2*>(iq) + in_base);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float4 x = src[t + 512 * r];
        const float2 w = reinterpret_cast<const float2*>(win)[t + 512 * r];
        acc[2 * r] = x.x + x.y * w.x;
        acc[2 * r + 1] = x.z + x.w * w.y;
      }
    }
    float4* dst = reinterpret_cast<float4*>(psd + frame * 8192);
#pragma unroll
    for (int r = 0; r < 4; ++r) dst[t + 512 * r] = make_float4(acc[4 * r], acc[4 * r + 1], acc[4 * r + 2], acc[4 * r + 3]);
    return;
  }
  // ---------------- pass 1: radix 16, Ns = 1, butterfly j = t ----------------
  float2 a[16];
  constexpr bool kBuf = ABLATE == 0;  // buffer addressing (the ablations keep flat loads)
  if constexpr (kBuf) {
    constexpr int kSample = FMT == FMT_CF32 ? 8 : 2;  // bytes per IQ sample
    const __amdgpu_buffer_rsrc_t rin = buffer_of(reinterpret_cast<const char*>(iq) + in_base * kSample, 8192 * kSample);
    const __amdgpu_buffer_rsrc_t rwin = buffer_of(win, 8192 * 4);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float2 x;
      if constexpr (FMT == FMT_CF32) {
        x = buffer_load_f2(rin, t * 8, 4096 * r);
      } else {
        const unsigned short raw = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, t * 2, 1024 * r, 0);
        if constexpr (FMT == FMT_CS8) x = make_float2((float)(signed char)(raw & 0xff) * scale, (float)(signed char)(raw >> 8) * scale);
        else x = make_float2(((float)(raw & 0xff) - 127.5f) * scale, ((float)(raw >> 8) - 127.5f) * scale);
      }
      const float w = buffer_load_f1(rwin, t * 4, 2048 * r);
      a[r] = make_float2(x.x * w, x.y * w);  // volk_32fc_32f_multiply_32fc
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int e = t + 512 * r;
      const float2 x = load_iq<FMT>(iq, in_base + e, scale);
      const float w = win[e];
      a[r] = make_float2(x.x * w, x.y * w);  // volk_32fc_32f_multiply_32fc
      if constexpr (ABLATE == 2) a[r] = make_float2(__int_as_float(0x3f800000 + e), db_off * (float)r);  // no global loads
    }
  }
  if constexpr (ABLATE == 1) {  // memory traffic only: same loads, same number of stores, no transform
#pragma unroll
    for (int r = 0; r < 16; ++r) psd[frame * 8192 + t + 512 * r] = a[r].x + a[r].y;
    return;
  }
  dft16(a);
  float2 c[16];
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
  // ---------------- pass 2: radix 16, Ns = 16, butterfly j = t ----------------
  {
    const int m = t & 15;
#pragma unroll
    for (int r = 1; r < 16; ++r) c[r] = cmul(c[r], tabs.tw2[r * 16 + m]);
  }
  dft16(c);
  __syncthreads();  // every read of y is done before z overwrites the plane
  // exchange 2: z[(t/16)*256 + t%16 + 16 k]; pass 3 lane (w, l) reads z[j + 256 (2q + h)], j = 32 w + (l & 31), h = l >> 5
  const int zbase = ((t >> 4) << 8) + (t & 15);
  const int lane = t & 63;
  const int h = lane >> 5;
  const int j = ((t >> 6) << 5) + (lane & 31);
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

  // ---------------- pass 3: radix 32, Ns = 256, butterfly j shared by lanes l and l + 32 ----------------
  // twiddle of input r = 2q + h:  W_8192^(j r) = W_8192^(j (r & 3)) * W_2048^(j (r >> 2)),  r & 3 = 2 (q & 1) + h,  r >> 2 = q >> 1
  {
    const float2 wa0 = tabs.tw3a[h * 256 + j];        // r & 3 = h      (h = 0: W^0 = 1)
    const float2 wa1 = tabs.tw3a[(2 + h) * 256 + j];  // r & 3 = 2 + h
    a[0] = cmul(a[0], wa0);
    a[1] = cmul(a[1], wa1);
#pragma unroll
    for (int q2 = 1; q2 < 8; ++q2) {
      const float2 wb = tabs.tw3b[q2 * 256 + j];
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
  float* out = psd + frame * 8192;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // v_permlane32_swap(vdst, src): lanes 32..63 of vdst <-> lanes 0..31 of src. With vdst = u[k], src = u[k+8]:
    //   low half:  (e, o) = (own u[k] = A_even[k],          partner's u[k]   = W^k A_odd[k])
    //   high half: (e, o) = (partner's u[k+8] = A_even[k+8], own u[k+8]      = W^(k+8) A_odd[k+8])
    const auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(u[k].x), __float_as_uint(u[k + 8].x), false, false);
    const auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(u[k].y), __float_as_uint(u[k + 8].y), false, false);
    const float2 e = make_float2(__uint_as_float(sx[0]), __uint_as_float(sy[0]));
    const float2 o = make_float2(__uint_as_float(sx[1]), __uint_as_float(sy[1]));
    const int kk = k + 8 * h;                 // this lane's output index k (0..15)
    const int bin0 = j + 256 * kk;            // X[kk]
    const int bin1 = bin0 + 256 * 16;         // X[kk + 16]
    if constexpr (ABLATE == 2) {  // compute only: the stores stay in the program but never execute
      if (scale == 12345.0f) {
        out[bin0 ^ 4096] = psd_db(cadd(e, o), db_off);
        out[bin1 ^ 4096] = psd_db(csub(e, o), db_off);
      }
    } else if constexpr (kBuf) {
      // bin0 < 4096: the half rotation (fft_v shift = true) sends X[kk] to bin0 + 4096 and X[kk + 16] to bin0
      const __amdgpu_buffer_rsrc_t rout = buffer_of(out, 8192 * 4);
      const int voff = (j + 2048 * h) * 4;
      buffer_store_f1(rout, voff, 1024 * k + 16384, psd_db(cadd(e, o), db_off));
      buffer_store_f1(rout, voff, 1024 * k, psd_db(csub(e, o), db_off));
    } else {
      out[bin0 ^ 4096] = psd_db(cadd(e, o), db_off);
      out[bin1 ^ 4096] = psd_db(csub(e, o), db_off);
    }
  }
}


}  // namespace ss
