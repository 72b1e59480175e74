// fft8192_round1.h — lab only (scripts/ubench/fft8192_lab.hip): the ROUND-1 8192-point kernel k_fft8192_psd_w8 with its
// memory-only / transform-only ablations (which compute garbage by design), kept as the reference point the later generations
// are measured against, and the stand-alone launch of the second-generation frame function (the product runs that function as a
// role of k_scan_step and launches neither). Moved out of the product's headers in round 4.
#pragma once
#include "../../rtl-sdr-scanner-cpp_amd/csrc/fft8192_v2.h"

namespace ss {

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



// Stand-alone launch of fft8192_v2_frame, one frame per workgroup.
template <int FMT, int TW, bool SWZ = false, bool NOWIN = false>
__global__ __launch_bounds__(512, 8) void k_fft8192_psd_v2(Fft8192Args g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int hdr;
  fft8192_v2_frame<FMT, TW, SWZ, NOWIN>(g, blockIdx.x, smem_raw, (int)threadIdx.x, &hdr);
}

}  // namespace ss
