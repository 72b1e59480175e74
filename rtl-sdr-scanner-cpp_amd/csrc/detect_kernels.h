// detect_kernels.h — back end of the scan chain for gfx950: noise ceiling, 21-frame and 21-bin means,
// threshold, candidate compaction.
//
// Replaces (reference file:line):
//   NoiseLearner::work / Noise::add   sources/radio/blocks/noise_learner.cpp:11-28,36-67
//   Averager::push / updateAverage    sources/radio/averager.cpp:14-25,52-61
//   average()                         sources/utils/utils.cpp:31-53
//   Transmission::addSignals :90-94   sources/radio/blocks/transmission.cpp (candidate predicate)
//
// Planes are frame-major float rows of N bins. `rel` carries G-1 history rows in front of the batch
// (the averager's ring, oldest first), so frame f of the batch is row (G-1)+f and its time window is
// rows f .. f+G-1. The reference keeps running sums (sum -= oldest; sum += newest); the kernels sum each
// window directly, oldest/lowest index first, in fp32 — same values to ~1e-6 dB, without the reference's
// drift; see DESIGN.md "numerics".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ss {

constexpr float kNoData = -100.0f;  // setNoData, sources/utils/radio_utils.cpp:72-76

// Noise::add for the learning frames of the batch (thr = max(thr, psd), frame order irrelevant for max)
// — noise_learner.cpp:19-21. One thread per bin, coalesced across the row.
__global__ void k_noise_learn(const float* __restrict__ psd, int n, int nlearn, float* __restrict__ thr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float t = thr[i];
  for (int f = 0; f < nlearn; ++f) {
    const float v = psd[(size_t)f * n + i];
    t = (t < v) ? v : t;  // std::max(t, v): keeps t unless t < v (NaN in v is ignored, as in the reference)
  }
  thr[i] = t;
}

// NoiseLearner::work output: -100 while learning (noise_learner.cpp:49), psd - thr afterwards (:55).
// rel_batch points at row G-1 of the rel buffer. Optionally mirrors the rows to the caller's plane.
__global__ void k_noise_apply(const float* __restrict__ psd, const float* __restrict__ thr, int n, int nframes, int nlearn,
                              float* __restrict__ rel_batch, float* __restrict__ rel_out) {
  const size_t total4 = (size_t)nframes * n / 4;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const size_t idx = e * 4;
    const int f = (int)(idx / n);
    const int i = (int)(idx % n);
    float4 r;
    if (f < nlearn) {
      r = make_float4(kNoData, kNoData, kNoData, kNoData);
    } else {
      const float4 p = *reinterpret_cast<const float4*>(psd + idx);
      const float4 t = *reinterpret_cast<const float4*>(thr + i);
      r = make_float4(p.x - t.x, p.y - t.y, p.z - t.z, p.w - t.w);
    }
    *reinterpret_cast<float4*>(rel_batch + idx) = r;
    if (rel_out) *reinterpret_cast<float4*>(rel_out + idx) = r;
  }
}

// Averager::average() after pushing frame f: mean of the newest G rows, or -100 until G frames were
// pushed since the last reset (averager.cpp:52-61). pushed_before = frames pushed before this batch.
__global__ void k_time_mean(const float* __restrict__ rel /*row 0 = oldest history*/, int n, int nframes, int G, int pushed_before,
                            float* __restrict__ avgy) {
  const size_t total4 = (size_t)nframes * n / 4;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total4; e += (size_t)gridDim.x * blockDim.x) {
    const size_t idx = e * 4;
    const int f = (int)(idx / n);
    const int i = (int)(idx % n);
    float4 r;
    if (pushed_before + f + 1 < G) {
      r = make_float4(kNoData, kNoData, kNoData, kNoData);
    } else {
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      const float* p = rel + (size_t)f * n + i;  // rows f .. f+G-1 = frames f-(G-1) .. f
      for (int g = 0; g < G; ++g) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)g * n);
        s.x += v.x;
        s.y += v.y;
        s.z += v.z;
        s.w += v.w;
      }
      const float d = (float)G;  // m_sum[i] / m_groupSize: float / int
      r = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
    }
    *reinterpret_cast<float4*>(avgy + idx) = r;
  }
}

// average(avgY -> avgXY, GROUPING_X) (utils.cpp:31-53: centred window clipped at the edges, divided by
// the number of bins actually inside) + the candidate predicate of transmission.cpp:91:
//   startLevel <= avg[i] && isIndexInRange(i) && !isIndexIgnored(i)   (pass[i] holds the last two)
// One wave-wide ballot per 64 bins becomes two 32-bit mask words. blockDim.x = 256, one block per
// 256 bins of one frame; the row segment plus halo is staged in LDS.
__global__ __launch_bounds__(256) void k_freq_mean_detect(const float* __restrict__ avgy, int n, int nframes, int gx, float start_level,
                                                          const uint8_t* __restrict__ pass, float* __restrict__ avg,
                                                          float* __restrict__ avg_out, uint32_t* __restrict__ maskbits) {
  extern __shared__ float row[];  // 256 + 2*a
  const int a = gx / 2;
  const int blocks_per_row = (n + 255) / 256;
  const int f = blockIdx.x / blocks_per_row;
  const int b0 = (blockIdx.x % blocks_per_row) * 256;
  const float* src = avgy + (size_t)f * n;
  for (int t = threadIdx.x; t < 256 + 2 * a; t += 256) {
    const int i = b0 - a + t;
    row[t] = (i >= 0 && i < n) ? src[i] : 0.0f;
  }
  __syncthreads();
  const int i = b0 + threadIdx.x;
  bool hit = false;
  if (i < n) {
    const int lo = max(0, i - a), hi = min(n - 1, i + a);
    float s = 0.0f;
    for (int k = lo; k <= hi; ++k) s += row[k - b0 + a];
    float v = s / (float)(hi - lo + 1);  // sum / count: float / int
    // average() walks i < size + a - 1 (utils.cpp:39): with a group of one (a = 0) it never reaches the last bin, which
    // keeps the 0.0 its caller initialised it with (transmission.cpp:60)
    if (a == 0 && i == n - 1) v = 0.0f;
    avg[(size_t)f * n + i] = v;
    if (avg_out) avg_out[(size_t)f * n + i] = v;
    hit = (start_level <= v) && pass[i];
  }
  const unsigned long long m = __ballot(hit);
  if ((threadIdx.x & 63) == 0 && i < n) {
    const size_t w = ((size_t)f * n + i) >> 5;
    maskbits[w] = (uint32_t)m;
    if (i + 32 < n) maskbits[w + 1] = (uint32_t)(m >> 32);
  }
}

// Candidate compaction, deterministic (ascending bin per frame, frames in order):
//   k_cand_count: popcount per frame;  k_cand_scan: exclusive scan over frames (one block);
//   k_cand_write: per frame, ranks inside the row by a block scan of word popcounts.
__global__ __launch_bounds__(256) void k_cand_count(const uint32_t* __restrict__ maskbits, int words_per_row, int* __restrict__ counts) {
  __shared__ int part[256];
  const uint32_t* row = maskbits + (size_t)blockIdx.x * words_per_row;
  int c = 0;
  for (int w = threadIdx.x; w < words_per_row; w += 256) c += __popc(row[w]);
  part[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[blockIdx.x] = part[0];
}

__global__ __launch_bounds__(256) void k_cand_scan(const int* __restrict__ counts, int nframes, int* __restrict__ off_int, int* __restrict__ off_out) {
  __shared__ int part[256];
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nframes; base += 256) {
    const int f = base + threadIdx.x;
    const int c = f < nframes ? counts[f] : 0;
    part[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan
      const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    const int excl = carry + part[threadIdx.x] - c;
    if (f < nframes) {
      off_int[f] = excl;
      if (off_out) off_out[f] = excl;
    }
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    off_int[nframes] = carry;
    if (off_out) off_out[nframes] = carry;
  }
}

__global__ __launch_bounds__(256) void k_cand_write(const uint32_t* __restrict__ maskbits, int words_per_row, int n, const int* __restrict__ off,
                                                    const float* __restrict__ avg, int cap, int* __restrict__ cand_idx,
                                                    float* __restrict__ cand_avg) {
  __shared__ int part[256];
  __shared__ int carry;
  const int f = blockIdx.x;
  const int begin = off[f];
  if (off[f + 1] == begin) return;
  const uint32_t* row = maskbits + (size_t)f * words_per_row;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < words_per_row; base += 256) {
    const int w = base + threadIdx.x;
    uint32_t bits = w < words_per_row ? row[w] : 0u;
    const int c = __popc(bits);
    part[threadIdx.x] = c;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int v = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
      __syncthreads();
      part[threadIdx.x] += v;
      __syncthreads();
    }
    int pos = begin + carry + part[threadIdx.x] - c;
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      const int i = w * 32 + b;
      if (pos < cap) {
        cand_idx[pos] = i;
        if (cand_avg) cand_avg[pos] = avg[(size_t)f * n + i];
      }
      ++pos;
    }
    __syncthreads();
    if (threadIdx.x == 255) carry += part[255];
    __syncthreads();
  }
}

// Spectrogram side branch — sources/radio/blocks/spectrogram.cpp:45-60: every frame adds the mean of `m`
// adjacent PSD bins to an accumulator per output bin. Two deterministic steps: a partial sum per chunk of frames
// (frames in order, the m bins in ascending order, then / m as the reference does), then the chunks in order.
__global__ __launch_bounds__(256) void k_spec_partial(const float* __restrict__ psd, int n, int nframes, int m, int out_n, int chunk,
                                                      float* __restrict__ partial) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= out_n) return;
  const int f0 = blockIdx.y * chunk;
  const int f1 = min(nframes, f0 + chunk);
  float acc = 0.0f;
  for (int f = f0; f < f1; ++f) {
    const float* p = psd + (size_t)f * n + (size_t)j * m;
    float s = 0.0f;
    for (int k = 0; k < m; ++k) s += p[k];
    acc += (m == 1) ? s : s / (float)m;  // spectrogram.cpp:48 / :52-57
  }
  partial[(size_t)blockIdx.y * out_n + j] = acc;
}

__global__ __launch_bounds__(256) void k_spec_combine(const float* __restrict__ partial, int nchunks, int out_n, float* __restrict__ sum) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= out_n) return;
  float acc = sum[j];
  for (int c0 = 0; c0 < nchunks; c0 += 16) {  // sixteen loads in flight, the additions one after the other in chunk order
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = partial[(size_t)min(c0 + i, nchunks - 1) * out_n + j];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (c0 + i < nchunks) acc += r[i];
  }
  sum[j] = acc;
}

// Keep the newest G-1 rows of [history ++ batch] as the next history (the averager ring).
__global__ void k_copy_rows(const float* __restrict__ src, float* __restrict__ dst, size_t count4) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < count4; e += (size_t)gridDim.x * blockDim.x) {
    reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[e];
  }
}

__global__ void k_fill(float* __restrict__ dst, size_t count, float v) {
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < count; e += (size_t)gridDim.x * blockDim.x) dst[e] = v;
}

}  // namespace ss
