// fft_kernels.h — front end of the scan chain for gfx950: load + int->float + window, Stockham FFT in
// LDS, |X|^2 -> dB, half rotation folded into the store index.
//
// Replaces, fused into one kernel per frame tile (reference file:line):
//   Decimator<gr_complex>::decimate   sources/radio/blocks/decimator.h:15-22   (first N of each N*D item)
//   gr::fft::fft_v<gr_complex,true>   sources/radio/sdr_device.cpp:164          (window, forward FFT, shift)
//   PSD::work                         sources/radio/blocks/psd.cpp:18-20        (10*log10f(|X|^2 / fs))
//
// Data layout: IQ items are interleaved (re,im) in HBM exactly as the SDR delivers them; one frame is
// staged once into LDS as float2 and never leaves the CU until its dB row is written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ss {

constexpr int kFftThreads = 256;

enum { FMT_CF32 = 0, FMT_CS8 = 1, FMT_CU8 = 2 };

// The fused form is spelled out: left to -ffp-contract the compiler picks which of the two products goes into the FMA per
// call site, and two copies of the same butterfly (the unrolled halves of k_fft256xR_psd<3>) then round differently — a
// frame's PSD must not depend on its position in the batch.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward transform rotation)
__device__ __forceinline__ float2 cmul_mi(float2 a) { return make_float2(a.y, -a.x); }

template <int FMT>
__device__ __forceinline__ float2 load_iq(const void* __restrict__ base, size_t idx, float scale) {
  if constexpr (FMT == FMT_CF32) {
    return reinterpret_cast<const float2*>(base)[idx];
  } else if constexpr (FMT == FMT_CS8) {
    const char2 c = reinterpret_cast<const char2*>(base)[idx];
    return make_float2((float)c.x * scale, (float)c.y * scale);
  } else {
    const uchar2 c = reinterpret_cast<const uchar2*>(base)[idx];
    return make_float2(((float)c.x - 127.5f) * scale, ((float)c.y - 127.5f) * scale);
  }
}

// PSD::work, psd.cpp:19:  10*log10f(cabsf(x)^2 / float(fs)), evaluated as
//     (10*log10(2)) * log2(re^2 + im^2)  -  10*log10(fs)
// with the hardware log2 (v_log_f32, 1 ulp) and the constant db_off = 10*log10(fs) rounded from double:
// 4 instructions per bin instead of ~40 for an IEEE division plus the library log10f, which were half
// of the whole kernel's VALU work. Differences from the reference's evaluation order stay below
// 5e-6 dB (the dB values themselves have an ulp of 3.8e-6 around -50 dB); re*re+im*im vs the squared
// hypotf is <= 2 ulp of the power (1e-6 dB). |x| = 0 gives -inf exactly like log10f(0).
__device__ __forceinline__ float psd_db(float2 x, float db_off) {
  const float p = fmaf(x.x, x.x, x.y * x.y);
  return fmaf(__log2f(p), 3.01029995663981195f, -db_off);
}

// One Stockham pass of radix R over `1 << LOGTOT` float2 elements held in LDS as independent
// contiguous sub-FFTs of size M = 1 << LOGM. Ns = product of the radices already done.
// In place: every thread first pulls all its inputs into registers, barrier, then scatters.
//   thread butterfly j:  v[r] = s[j + r*M/R] * W_M^(r*(j mod Ns)*M/(Ns*R));  DFT_R(v);
//                        s[(j/Ns)*Ns*R + (j mod Ns) + r*Ns] = v[r]
// tw = W_TW^k table (TW >= M, power of two), tw_shift = log2(TW / M). BS = element stride between sub-FFTs
// (M, or M + 1 to keep column-wise tile accesses off a single LDS bank).
template <int LOGM, int LOGTOT, int LOGNS, int R, int BS = (1 << LOGM)>
__device__ __forceinline__ void stockham_pass(float2* __restrict__ s, const float2* __restrict__ tw, int tw_shift, int tid) {
  constexpr int M = 1 << LOGM;
  constexpr int TOT = 1 << LOGTOT;
  constexpr int NS = 1 << LOGNS;
  constexpr int LOGR = (R == 4) ? 2 : 1;
  constexpr int BPF = M / R;           // butterflies per sub-FFT
  constexpr int NB = TOT / R;          // butterflies in the tile
  constexpr int Q = (NB + kFftThreads - 1) / kFftThreads;
  float2 v[Q][R];
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int g = tid + q * kFftThreads;
    if (NB >= kFftThreads || g < NB) {
      const int b = g / BPF;
      const int j = g % BPF;
      const int m = j & (NS - 1);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float2 x = s[b * BS + j + r * BPF];
        if (LOGNS > 0 && r > 0) {
          // W_M^(r*m*M/(NS*R)) = W_TW^((r*m) << (LOGM - LOGNS - LOGR + tw_shift))
          const float2 w = tw[(r * m) << (LOGM - LOGNS - LOGR + tw_shift)];
          x = cmul(x, w);
        }
        v[q][r] = x;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int g = tid + q * kFftThreads;
    if (NB >= kFftThreads || g < NB) {
      const int b = g / BPF;
      const int j = g % BPF;
      const int m = j & (NS - 1);
      const int j0 = ((j >> LOGNS) << (LOGNS + LOGR)) + m;
      float2* o = s + b * BS + j0;
      if constexpr (R == 4) {
        const float2 a0 = cadd(v[q][0], v[q][2]);
        const float2 a1 = csub(v[q][0], v[q][2]);
        const float2 a2 = cadd(v[q][1], v[q][3]);
        const float2 a3 = cmul_mi(csub(v[q][1], v[q][3]));
        o[0 * NS] = cadd(a0, a2);
        o[1 * NS] = cadd(a1, a3);
        o[2 * NS] = csub(a0, a2);
        o[3 * NS] = csub(a1, a3);
      } else {
        o[0] = cadd(v[q][0], v[q][1]);
        o[NS] = csub(v[q][0], v[q][1]);
      }
    }
  }
  __syncthreads();
}

// All passes of a size-M FFT: radix 4 while at least two bits remain, then one radix-2 pass.
template <int LOGM, int LOGTOT, int LOGNS = 0, int BS = (1 << LOGM)>
__device__ __forceinline__ void stockham_fft(float2* __restrict__ s, const float2* __restrict__ tw, int tw_shift, int tid) {
  if constexpr (LOGNS < LOGM) {
    if constexpr (LOGM - LOGNS >= 2) {
      stockham_pass<LOGM, LOGTOT, LOGNS, 4, BS>(s, tw, tw_shift, tid);
      stockham_fft<LOGM, LOGTOT, LOGNS + 2, BS>(s, tw, tw_shift, tid);
    } else {
      stockham_pass<LOGM, LOGTOT, LOGNS, 2, BS>(s, tw, tw_shift, tid);
      stockham_fft<LOGM, LOGTOT, LOGNS + 1, BS>(s, tw, tw_shift, tid);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// N <= 8192: whole frames in LDS. One workgroup transforms FPB = TOT/N frames.
//   grid.x = ceil(nframes / FPB)
// ---------------------------------------------------------------------------------------------
template <int LOGN, int LOGTOT, int FMT>
__global__ __launch_bounds__(kFftThreads) void k_fft_psd_lds(const void* __restrict__ iq, long long item_stride /*samples*/,
                                                               int nframes, const float* __restrict__ win,
                                                               const float2* __restrict__ tw, float db_off, float scale,
                                                               float* __restrict__ psd) {
  constexpr int N = 1 << LOGN;
  constexpr int TOT = 1 << LOGTOT;
  constexpr int FPB = TOT / N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* s = reinterpret_cast<float2*>(smem_raw);
  const int tid = threadIdx.x;
  const int f0 = blockIdx.x * FPB;

#pragma unroll 4
  for (int e = tid; e < TOT; e += kFftThreads) {
    const int fl = e >> LOGN;
    const int n = e & (N - 1);
    float2 x = make_float2(0.f, 0.f);
    if (f0 + fl < nframes) {
      x = load_iq<FMT>(iq, (size_t)(f0 + fl) * (size_t)item_stride + n, scale);
      const float w = win[n];
      x.x *= w;  // volk_32fc_32f_multiply_32fc: one rounding per component
      x.y *= w;
    }
    s[e] = x;
  }
  __syncthreads();
  stockham_fft<LOGN, LOGTOT>(s, tw, 0, tid);
#pragma unroll 4
  for (int e = tid; e < TOT; e += kFftThreads) {
    const int fl = e >> LOGN;
    const int k = e & (N - 1);
    if (f0 + fl < nframes) {
      // fft_v shift=true: out[i] = X[(i + N/2) mod N]  ->  X[k] lands at k ^ (N/2)
      psd[(size_t)(f0 + fl) * N + (k ^ (N >> 1))] = psd_db(s[e], db_off);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// N > 8192: four-step, N = N1 * N2, n = n1*N2 + n2, k = k1 + N1*k2.
//   step A (columns): for every n2, FFT over n1 (stride N2), times W_N^(n2*k1)  -> work[f][k1*N2 + n2]
//   step B (rows)   : for every k1, FFT over n2 (contiguous)                    -> X[k1 + N1*k2] -> dB
// Each workgroup keeps a tile of 8192 points in LDS. work is an internal HBM buffer (frame-major).
// ---------------------------------------------------------------------------------------------
template <int LOGN1, int LOGN2, int FMT>
__global__ __launch_bounds__(kFftThreads) void k_fft_cols(const void* __restrict__ iq, long long item_stride, const float* __restrict__ win,
                                                            const float2* __restrict__ tw, float scale, float2* __restrict__ work) {
  constexpr int LOGN = LOGN1 + LOGN2;
  constexpr int N1 = 1 << LOGN1, N2 = 1 << LOGN2;
  constexpr int LOGC = 13 - LOGN1;  // columns per tile
  constexpr int C = 1 << LOGC;
  constexpr int TOT = 1 << 13;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* s = reinterpret_cast<float2*>(smem_raw);
  const int tid = threadIdx.x;
  const int tiles_per_frame = N2 / C;
  const int f = blockIdx.x / tiles_per_frame;
  const int c0 = (blockIdx.x % tiles_per_frame) * C;
  const size_t in_base = (size_t)f * (size_t)item_stride;
  // LDS layout: column c is the sub-FFT s[c*BS + n1], BS = N1 + 1: adjacent lanes hold adjacent columns, and the odd
  // stride spreads them over all banks (with BS = N1 every lane of a wave would hit the same bank)
  constexpr int BS = N1 + 1;
#pragma unroll 4
  for (int e = tid; e < TOT; e += kFftThreads) {
    const int c = e & (C - 1);  // fastest in memory: adjacent lanes read adjacent n2
    const int n1 = e >> LOGC;
    const int n = n1 * N2 + c0 + c;
    float2 x = load_iq<FMT>(iq, in_base + n, scale);
    const float w = win[n];
    x.x *= w;
    x.y *= w;
    s[c * BS + n1] = x;
  }
  __syncthreads();
  stockham_fft<LOGN1, 13, 0, BS>(s, tw, LOGN - LOGN1, tid);
  float2* wf = work + (size_t)f * (1 << LOGN);
#pragma unroll 4
  for (int e = tid; e < TOT; e += kFftThreads) {
    const int c = e & (C - 1);
    const int k1 = e >> LOGC;
    const int n2 = c0 + c;
    const float2 t = tw[(size_t)n2 * k1];  // W_N^(n2*k1), n2*k1 < N
    wf[k1 * N2 + n2] = cmul(s[c * BS + k1], t);
  }
}

template <int LOGN1, int LOGN2>
__global__ __launch_bounds__(kFftThreads) void k_fft_rows_psd(const float2* __restrict__ work, const float2* __restrict__ tw, float db_off,
                                                                float* __restrict__ psd) {
  constexpr int LOGN = LOGN1 + LOGN2;
  constexpr int N = 1 << LOGN, N1 = 1 << LOGN1, N2 = 1 << LOGN2;
  constexpr int LOGRW = 13 - LOGN2;  // rows (k1 values) per tile
  constexpr int RW = 1 << LOGRW;
  constexpr int TOT = 1 << 13;
  constexpr int BS = N2 + 1;  // row r is the sub-FFT s[r*BS + n2]; the odd stride keeps the row-fastest read-out off one bank
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float2* s = reinterpret_cast<float2*>(smem_raw);
  const int tid = threadIdx.x;
  const int tiles_per_frame = N1 / RW;
  const int f = blockIdx.x / tiles_per_frame;
  const int r0 = (blockIdx.x % tiles_per_frame) * RW;
  const float2* wf = work + (size_t)f * N;
#pragma unroll 4
  for (int e = tid; e < TOT; e += kFftThreads) {
    s[(e >> LOGN2) * BS + (e & (N2 - 1))] = wf[(size_t)r0 * N2 + e];  // rows r0..r0+RW-1 are contiguous in work
  }
  __syncthreads();
  stockham_fft<LOGN2, 13, 0, BS>(s, tw, LOGN - LOGN2, tid);
  float* out = psd + (size_t)f * N;
#pragma unroll 4
  for (int e = tid; e < TOT; e += kFftThreads) {
    const int r = e & (RW - 1);  // fastest: adjacent lanes write adjacent k1 -> adjacent output bins
    const int k2 = e >> LOGRW;
    const int k = (r0 + r) + N1 * k2;
    out[k ^ (N >> 1)] = psd_db(s[r * BS + k2], db_off);
  }
}

}  // namespace ss
