// fft8192_kernel.h — butterflies and addressing helpers of the N = 8192 front end (16- / 8- / 4-point DFTs held in registers,
// the W_32 rotations, raw buffer loads and stores with cache-policy bits). Nothing in this file is a kernel: the 8192-point
// transform that ships is fft8192_v2_frame (fft8192_v2.h), a role of k_scan_step (scan_step.h), and the 256-point register
// passes of fft256_kernels.h use the same butterflies. (The round-1 kernel k_fft8192_psd_w8 that used to live here is the
// reference point of scripts/ubench/fft8192_lab and lives there now: scripts/ubench/fft8192_round1.h.)
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

template <int K>
__device__ __forceinline__ float2 mulw32_if(float2 x, bool on) {
  const float2 y = mulw32<K>(x);
  return on ? y : x;
}

}  // namespace ss
