// channelizer.hip — recorder channeliser for gfx950 behind include/specscan_channelizer.h.
//
// Replaces, for all recording slots of one device at once, the per-slot GNU Radio chain of the reference's Recorder
// (sources/radio/recorder.cpp:22-37): rotator_cc -> rational_resampler<cc,cc,cc> x stages -> complex_to_interleaved_char.
//   stage factors: getResamplersFactors  (sources/utils/radio_utils.cpp:9-35,105-152) — restated below, pinned by the
//                  reference's own known-answer tests (tests/test_radio_utils.cpp:28-69)
//   taps:          rational_resampler::make with no taps -> design_resampler_filter(interp, decim, 0.4) ->
//                  firdes::low_pass(Kaiser, beta 7) — GNU Radio 3.10, restated (un-vendored in the reference)
//   data path:     k_chan_stage: one launch per stage for all active slots. A workgroup owns a tile of output samples
//                  of one slot, stages the input span it needs through LDS (first stage: the shared raw IQ, rotated on
//                  the way in; later stages: the slot's previous stage), laid out polyphase-major so that the 64 lanes
//                  of a wave — 64 consecutive outputs — read consecutive LDS words for every tap, and accumulates
//                  y[m] = sum_j arm[ctr(m)][j] * x[p(m) - j]  (rational_resampler_impl::general_work as a stream).
//   rotator:       phase(n) = phase0 * inc^n evaluated in closed form: the angle of the reference's fp32-rounded
//                  increment (rotator::set_phase_incr) times n, reduced in fp64, then sincospi — no recurrence, so no
//                  drift and no dependence on how the stream is cut into calls (the reference's recurrence renormalises
//                  every 512 samples and at the end of every work() call).
// Bound: fp32 FMA issue and LDS reads (≈33 taps x 2 FMA per input sample per slot); HBM traffic is the input once.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <vector>

#include "../../include/specscan.h"
#include "../../include/specscan_channelizer.h"

namespace {

thread_local char g_sc_create_err[512] = "";

// ---------------------------------------------------------------------------------------------------------------
// design (host): the reference's factor split and GNU Radio's default resampler taps
// ---------------------------------------------------------------------------------------------------------------

int prime_factor_count(int n) {  // getPrimeFactors(n).size(), radio_utils.cpp:105-126
  if (n == 1) return 1;
  int count = 0;
  while (n % 2 == 0) {
    ++count;
    n /= 2;
  }
  for (int i = 3; i <= std::sqrt((double)n); i += 2) {
    while (n % i == 0) {
      ++count;
      n /= i;
    }
  }
  if (n > 2) ++count;
  return count;
}

void split_factors(int value, std::vector<int>& factors, int threshold) {  // split, radio_utils.cpp:9-34
  if (threshold < value && prime_factor_count(value) != 1) {
    int f1 = 1, f2 = value;
    for (int i = (int)std::sqrt((double)value); 1 <= i; --i) {
      if (value % i == 0) {
        f1 = i;
        f2 = value / i;
        break;
      }
    }
    if (threshold < f1) split_factors(f1, factors, threshold);
    else factors.push_back(f1);
    if (threshold < f2) split_factors(f2, factors, threshold);
    else factors.push_back(f2);
  } else {
    factors.push_back(value);
  }
}

std::vector<std::pair<int, int>> resampler_factors(int sample_rate, int bandwidth, int threshold) {  // radio_utils.cpp:128-152
  const int g = std::gcd(sample_rate, bandwidth);
  std::vector<int> left, right;
  split_factors(bandwidth / g, left, threshold);
  split_factors(sample_rate / g, right, threshold);
  while (left.size() < right.size()) left.push_back(1);
  while (right.size() < left.size()) right.push_back(1);
  std::sort(left.begin(), left.end());
  std::sort(right.begin(), right.end());
  std::vector<std::pair<int, int>> r;
  for (size_t i = 0; i < left.size(); ++i) r.push_back({left[i], right[i]});
  return r;
}

double izero(double x) {  // gr-fft window.cc: Izero, IzeroEPSILON = 1e-21
  double sum = 1, u = 1, n = 1;
  const double halfx = x / 2.0;
  double temp;
  do {
    temp = halfx / n;
    n += 1;
    temp *= temp;
    u *= temp;
    sum += u;
  } while (u >= 1e-21 * sum);
  return sum;
}

// gr::filter::firdes::low_pass(gain, fs, cutoff, transition width, WIN_KAISER, beta): the float/double mix is GNU Radio's
std::vector<float> low_pass_kaiser(double gain, double fs, double cutoff, double tw, double beta) {
  const double atten = beta / 0.1102 + 8.7;  // window::max_attenuation(WIN_KAISER, beta)
  int ntaps = (int)(atten * fs / (22.0 * tw));  // firdes::compute_ntaps
  if ((ntaps & 1) == 0) ntaps++;
  std::vector<float> w((size_t)ntaps), taps((size_t)ntaps);
  const double ibeta = 1.0 / izero(beta), inm1 = 1.0 / ((double)(ntaps - 1));
  for (int i = 0; i < ntaps; ++i) {
    const double t = 2 * i * inm1 - 1;
    w[(size_t)i] = (float)(izero(beta * std::sqrt(1.0 - t * t)) * ibeta);
  }
  const int M = (ntaps - 1) / 2;
  const double fwT0 = 2 * M_PI * cutoff / fs;
  for (int n = -M; n <= M; ++n) {
    if (n == 0) taps[(size_t)(n + M)] = (float)(fwT0 / M_PI * w[(size_t)(n + M)]);
    else taps[(size_t)(n + M)] = (float)(std::sin(n * fwT0) / (n * M_PI) * w[(size_t)(n + M)]);
  }
  double fmax = taps[(size_t)M];
  for (int n = 1; n <= M; ++n) fmax += 2 * taps[(size_t)(n + M)];
  gain /= fmax;
  for (auto& t : taps) t = (float)(t * gain);
  return taps;
}

// rational_resampler::make(interp, decim) without taps: gcd-reduced, fractional_bw 0 -> 0.4, design_resampler_filter
std::vector<float> design_resampler_taps(int interp, int decim) {
  const float fractional_bw = 0.4f, beta = 7.0f, halfband = 0.5f;
  const float rate = (float)interp / (float)decim;
  float trans_width, mid;
  if (rate >= 1.0f) {
    trans_width = halfband - fractional_bw;
    mid = (float)(halfband - trans_width / 2.0);
  } else {
    trans_width = rate * (halfband - fractional_bw);
    mid = (float)(rate * halfband - trans_width / 2.0);
  }
  return low_pass_kaiser((double)interp, (double)interp, (double)mid, (double)trans_width, (double)beta);
}

struct Stage {
  int interp = 1, decim = 1, ntaps = 0, nt = 0;  // nt = taps per polyphase arm = block history
  std::vector<float> taps;
  float* d_arm_dec = nullptr;  // fast first stage: h[D t + b] at [t][b], b padded to 64 per pass, zeros beyond D and ntaps
  float* d_arm = nullptr;  // [interp][nt]: arm[i][j] = taps[i + j * interp], zero padded (rational_resampler_impl::install_taps)
  float2* d_buf = nullptr;  // stages > 0: per slot [hist (nt-1) | new samples], stride buf_stride
  long long buf_stride = 0;
  int max_in = 0, max_out = 0;
  int tile = 64;        // outputs per workgroup
  int lds_floats2 = 0;  // LDS elements a tile needs at most
  // first stage with interpolation 1: the polyphase-by-branch kernel (k_chan_dec)
  bool fast = false;
  int logg = 6, passes = 1, waves = 4;
};

struct Slot {
  bool active = false;
  float inc_re = 1.0f, inc_im = 0.0f;  // rotator d_phase_incr (fp32, normalised)
  double f0 = 0.0;                     // phase of the next input sample, in revolutions
  double df = 0.0;                     // angle of the fp32 increment, in revolutions
  int ctr[SC_MAX_STAGES] = {};         // d_ctr of every resampler
  int skip[SC_MAX_STAGES] = {};        // input samples to pass before the next output's window ends
};

}  // namespace

struct sc_ctx {
  sc_config cfg{};
  std::mutex mtx;
  char err[512] = "";
  hipStream_t stream = nullptr;
  std::vector<Stage> stages;
  Slot slots[SC_MAX_CHANNELS];
  float2* d_hist0 = nullptr;  // first stage: per slot the newest nt-1 ROTATED samples
  long long hist0_stride = 0;
  float2* d_ptab = nullptr;  // k_chan_dec: per slot exp(2*pi*i*k*df), rebuilt by sc_start
  long long ptab_stride = 0;
  // host-entry staging
  float2* d_in = nullptr;
  int8_t* d_out_i8 = nullptr;
  float* d_out_cf32 = nullptr;
  int out_cap_alloc = 0;
};

namespace {

int sc_fail(sc_ctx* c, int code, const char* fmt, ...) {
  char* dst = c ? c->err : g_sc_create_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define SC_HIP(ctx, call)                                                                                 \
  do {                                                                                                    \
    hipError_t e_ = (call);                                                                               \
    if (e_ != hipSuccess) return sc_fail(ctx, SS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

// ---------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------

struct ChanArgs {
  const float2* in_raw;  // ROTATE: the device stream, n_in new samples, shared by all slots
  const float2* hist0;   // ROTATE: per slot, the newest nt-1 rotated samples before this call
  long long hist0_stride;
  const float2* in_buf;  // !ROTATE: per slot [hist (nt-1) | new]
  long long in_stride;
  const float* arm;
  const float* arm_dec;  // k_chan_dec: taps by (tap-in-branch t, branch b): [33][64 * passes], zero outside the filter
  const float2* ptab;  // decimating fast path: per slot exp(2*pi*i*k*df), k = 0 .. ptab_stride-1
  long long ptab_stride;
  int interp, decim, nt, tile;
  float2* next_buf;  // outputs for the next stage (written after its history), or null
  long long next_stride;
  int next_hist;
  float2* out_cf32;  // caller's cf32 plane [slot][cap], or null (last stage only)
  int8_t* out_i8;    // caller's int8 plane [slot][cap][2], or null (last stage only)
  int cap;
  float pack_scale;
  int nslots;
  int slot[SC_MAX_CHANNELS];
  int ctr0[SC_MAX_CHANNELS], skip0[SC_MAX_CHANNELS], nin[SC_MAX_CHANNELS], nout[SC_MAX_CHANNELS];
  double f0[SC_MAX_CHANNELS], df[SC_MAX_CHANNELS];
};

// rotator phase of stream sample n of this call: exp(2*pi*i*(f0 + n*df)), the fraction reduced in fp64
__device__ __forceinline__ float2 phase_of(double f0, double df, int n) {
  double f = fma((double)n, df, f0);
  f -= rint(f);  // [-0.5, 0.5]
  float s, c;
  sincospif(2.0f * (float)f, &s, &c);
  return make_float2(c, s);
}

__device__ __forceinline__ float2 rotate(float2 x, float2 p) {  // in * phase (rotator::rotate)
  return make_float2(x.x * p.x - x.y * p.y, x.x * p.y + x.y * p.x);
}

// volk_32f_s32f_convert_8i: r = in * scalar; saturate to [-128, 127]; rintf (round to nearest even)
__device__ __forceinline__ int8_t to_i8(float v, float scale) {
  const float r = v * scale;
  return r > 127.0f ? (int8_t)127 : (r < -128.0f ? (int8_t)-128 : (int8_t)rintf(r));
}

template <bool ROTATE>
__global__ __launch_bounds__(256) void k_chan_stage(ChanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chan_smem[];
  float2* lds = reinterpret_cast<float2*>(chan_smem);
  __shared__ float2 part[4][64];
  const int s = blockIdx.y;
  const int nout = a.nout[s];
  const int m0 = blockIdx.x * a.tile;
  if (m0 >= nout) return;
  const int slot = a.slot[s];
  const int tcount = min(a.tile, nout - m0);
  const int I = a.interp, D = a.decim, nt = a.nt;
  // stream position of output m: total = ctr0 + m*D; window ends at p = skip0 + total / I; arm = total % I
  const long long tot_first = (long long)a.ctr0[s] + (long long)m0 * D;
  const long long tot_last = (long long)a.ctr0[s] + (long long)(m0 + tcount - 1) * D;
  const int p_first = a.skip0[s] + (int)(tot_first / I);
  const int p_last = a.skip0[s] + (int)(tot_last / I);
  const int n_lo = p_first - (nt - 1);  // oldest sample the tile needs; negative = history
  // polyphase-major LDS layout: sample n sits at row (n - base) % D, column (n - base) / D; base is a multiple of D
  const int base = (n_lo >= 0 ? n_lo / D : -((-n_lo + D - 1) / D)) * D;
  const int cols = (p_last - base) / D + 1;
  const int pitch = cols | 1;  // odd: the transposing fill does not pile up on one bank
  const int span = p_last - n_lo + 1;
  const int h = nt - 1;
  for (int idx = threadIdx.x; idx < span; idx += 256) {
    const int n = n_lo + idx;
    float2 v;
    if (ROTATE) {
      if (n < 0) v = a.hist0[(size_t)slot * a.hist0_stride + (h + n)];
      else v = rotate(a.in_raw[n], phase_of(a.f0[s], a.df[s], n));
    } else {
      v = a.in_buf[(size_t)slot * a.in_stride + (h + n)];
    }
    const int r = n - base;
    lds[(r % D) * pitch + r / D] = v;
  }
  __syncthreads();
  const int o = threadIdx.x & 63, q = threadIdx.x >> 6;
  float2 acc = make_float2(0.0f, 0.0f);
  if (o < tcount) {
    const long long tot = tot_first + (long long)o * D;
    const int p = a.skip0[s] + (int)(tot / I);
    const float* arm = a.arm + (size_t)(tot % I) * nt;
    const int chunk = (nt + 3) / 4;
    const int j0 = q * chunk, j1 = min(nt, j0 + chunk);
    int r = p - j0 - base;
    int row = r % D, col = r / D;
    for (int j = j0; j < j1; ++j) {
      const float2 x = lds[row * pitch + col];
      const float t = arm[j];
      acc.x = fmaf(x.x, t, acc.x);
      acc.y = fmaf(x.y, t, acc.y);
      if (--row < 0) {
        row = D - 1;
        --col;
      }
    }
  }
  part[q][o] = acc;
  __syncthreads();
  if (q == 0 && o < tcount) {
    float2 y = part[0][o];
    y.x += part[1][o].x;
    y.y += part[1][o].y;
    y.x += part[2][o].x;
    y.y += part[2][o].y;
    y.x += part[3][o].x;
    y.y += part[3][o].y;
    const int m = m0 + o;
    if (a.next_buf) a.next_buf[(size_t)slot * a.next_stride + a.next_hist + m] = y;
    if (m < a.cap) {
      if (a.out_cf32) a.out_cf32[(size_t)slot * a.cap + m] = y;
      if (a.out_i8) {
        char2 v;
        v.x = to_i8(y.x, a.pack_scale);
        v.y = to_i8(y.y, a.pack_scale);
        reinterpret_cast<char2*>(a.out_i8)[(size_t)slot * a.cap + m] = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// First stage, interpolation 1 (every configuration the reference's defaults produce): polyphase by branch.
//   y[m] = sum_b sum_a h[D a + b] * x[p(m - a) - b],   p(m) = skip + m D,   b = 0..D-1,   a = 0..32
// (GNU Radio's default design always has ceil(ntaps / D) = 33 taps per branch: ntaps = 32.8 D.)
// A lane owns one branch b and R = 16 consecutive outputs: it pulls the 48 samples x[p(m0 + k - 32) - b] of its
// branch out of LDS once (3 LDS reads per output instead of 33), holds its 33 taps in registers, and runs 16 x 33
// complex-by-real multiply-adds; the D branch sums of every output are then added across the lanes of a group with a
// halving butterfly (32 shuffles for the 32 partial values of a lane, not 32 x log2 D). Groups of 8/16/32/64 lanes
// cover D <= 64; for D up to 128 a lane takes branches b and b + 64. The rotator is applied while the samples are
// staged: phase(n) = P(n_lo) * T[n - n_lo] with T[k] = exp(2 pi i k df) tabulated per slot when the shift is set
// (two roundings per sample, independent of where tiles and calls fall).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDecA = 33, kDecR = 16;

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

__global__ void k_chan_ptab(float2* __restrict__ tab, int n, double df) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) tab[k] = phase_of(0.0, df, k);
}

// Two instantiations share the tiles: FULL takes the tiles whose whole span lies in this call's input and that
// produce a full set of outputs — straight-line staging and LDS reads at fixed offsets, nothing clamped; the other one
// takes the few that start in the slot's history (tiles 0..2 at most: three tiles cover more than the filter) or end
// ragged (the last one).
template <int LOGG, int PASSES, bool FULL>
__device__ __forceinline__ void chan_dec_tile(const ChanArgs& a, float2* __restrict__ lds, int block_x) {
  constexpr int GL = 1 << LOGG, G = 64 / GL, A = kDecA, R = kDecR, W = A - 1 + R;
  const int s = blockIdx.y;
  const int nout = a.nout[s];
  const int tile_out = (int)(blockDim.x >> 6) * G * R;
  int tile_index = block_x;
  if (!FULL && block_x == 3) {
    tile_index = (nout - 1) / tile_out;  // the last tile, unless it is one of the first three
    if (tile_index < 3) return;
  }
  const int m0 = tile_index * tile_out;
  if (m0 >= nout) return;
  const int slot = a.slot[s];
  const int D = a.decim, h = a.nt - 1;
  const int tcount = min(tile_out, nout - m0);
  const int p_first = a.skip0[s] + m0 * D;  // interpolation 1: the arm counter is always 0
  const int p_last = p_first + (tcount - 1) * D;
  const int n_lo = p_first - (A * D - 1);  // oldest sample any lane touches (taps beyond ntaps are zero)
  const int span = p_last - n_lo + 1;
  if (FULL != (n_lo >= 0 && tcount == tile_out)) return;  // block-uniform: the other instantiation has this tile
  const float2 P0 = phase_of(a.f0[s], a.df[s], n_lo);
  const float2* ptab = a.ptab + (size_t)slot * a.ptab_stride;
  // this lane's taps of the first pass: requested before the staging so that their latency hides behind it
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lg = lane & (GL - 1), g = lane >> LOGG;
  constexpr bool kTapsFirst = LOGG == 6 && PASSES == 1;
  constexpr int kTapPitch = 64 * PASSES;
  float H0[A];
  if constexpr (kTapsFirst) {
#pragma unroll
    for (int t = 0; t < A; ++t) H0[t] = a.arm_dec[t * kTapPitch + lg];
  }
  {
    // Up to 24 independent, unconditional loads per thread in flight (the whole span of a 64-output tile at D = 64), all
    // on legal addresses — a load, its arithmetic and its store per loop trip would serialise on the memory latency.
    // Phase of sample idx = TPB u + tid: (P0 T[tid]) T[TPB u], the second factor block-uniform (scalar loads), so the
    // table costs one vector load per thread instead of one per sample. Tiles that begin before this call's input
    // (edge instantiation only) take those samples, already rotated, from the slot's history; before that, zeros.
    constexpr int U = LOGG == 6 ? 24 : 12;  // smaller tiles keep four waves per SIMD: fewer registers
    const int TPB = (int)blockDim.x;  // 64, 128 or 256: as many waves as the span leaves LDS for
    const float2* src = a.in_raw + n_lo;
    const float2* hist = a.hist0 + (size_t)slot * a.hist0_stride + h + n_lo;  // hist[idx] = stream sample n_lo + idx < 0
    const float2 q = cmulf(P0, ptab[threadIdx.x]);
    if (FULL && span == U * TPB) {
      // the span of a full tile at D = 64 is exactly 24 samples per thread: no clamps, no predicates
      float2 xv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) xv[u] = src[u * TPB + (int)threadIdx.x];
#pragma unroll
      for (int u = 0; u < U; ++u) lds[u * TPB + (int)threadIdx.x] = rotate(xv[u], cmulf(q, ptab[u * TPB]));
    } else {
      for (int base = 0; base < span; base += U * TPB) {
        float2 xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = min(base + u * TPB + (int)threadIdx.x, span - 1);
          const float2* p = src + idx;
          if (!FULL && n_lo + idx < 0) p = hist + max(idx, -(h + n_lo));
          xv[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * TPB + (int)threadIdx.x;
          const float2 th = ptab[min(base + u * TPB, span - 1)];
          float2 v = rotate(xv[u], cmulf(q, th));
          if (!FULL && n_lo + idx < 0) v = n_lo + idx >= -h ? xv[u] : make_float2(0.0f, 0.0f);
          if (idx < span) lds[idx] = v;
        }
      }
    }
  }
  __syncthreads();
  const int blk = wave * G + g;  // this group's block of R outputs inside the tile
  float v[2 * R];
#pragma unroll
  for (int i = 0; i < 2 * R; ++i) v[i] = 0.0f;
  if (FULL || blk * R < tcount) {
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int b = lg + 64 * pass;
      const int bc = min(b, D - 1);  // lanes beyond the last branch read a legal sample and multiply it by zero taps
      float H[A];
#pragma unroll
      for (int t = 0; t < A; ++t) H[t] = (kTapsFirst && pass == 0) ? H0[t] : a.arm_dec[t * kTapPitch + b];
      // w[k] = x[p(m_blk + k - 32) - b]: LDS index of k = 0 is (p_first - n_lo) + blk*R*D - 32 D - b, then steps of D
      const int i0 = (A * D - 1) + blk * R * D - (A - 1) * D - bc;
      float2 w[W];
#pragma unroll
      for (int k = 0; k < W; ++k) w[k] = lds[FULL ? i0 + k * D : min(i0 + k * D, span - 1)];
#pragma unroll
      for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int t = 0; t < A; ++t) {
          v[2 * r] = fmaf(w[A - 1 + r - t].x, H[t], v[2 * r]);
          v[2 * r + 1] = fmaf(w[A - 1 + r - t].y, H[t], v[2 * r + 1]);
        }
      }
    }
  }
  // add the branch sums across the GL lanes of the group: halve the value set while doubling what each value has seen
  constexpr int S = LOGG - 1;
  int base = 0;
#pragma unroll
  for (int st = 0; st < S; ++st) {
    const int cnt = R >> st;      // values kept after this stage (of 2R, R, ...)
    const int bit = GL >> (st + 1);
    const bool hi = (lane & bit) != 0;
    // bitwise selects (one v_bfi_b32 each): a ?: on two array elements gets turned into a runtime index into the
    // register array, i.e. a compare/select chain over all 32 values per access
    const uint32_t mh = hi ? 0xffffffffu : 0u;
#pragma unroll
    for (int i = 0; i < cnt; ++i) {
      const uint32_t lo_v = __float_as_uint(v[i]), hi_v = __float_as_uint(v[i + cnt]);
      const float send = __uint_as_float((lo_v & mh) | (hi_v & ~mh));
      const float keep = __uint_as_float((hi_v & mh) | (lo_v & ~mh));
      v[i] = keep + __shfl_xor(send, bit);
    }
    base += hi ? cnt : 0;
  }
  constexpr int CNT = (2 * R) >> S;  // values a lane holds now; lanes l and l^1 hold the same ones
#pragma unroll
  for (int i = 0; i < CNT; ++i) v[i] += __shfl_xor(v[i], 1);
  if ((lane & 1) == 0) {
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
      const int idx = base + i;  // 2 r + component
      const int r = idx >> 1, comp = idx & 1;
      const int m = m0 + blk * R + r;
      if (FULL || blk * R + r < tcount) {
        const float y = v[i];
        if (a.next_buf) reinterpret_cast<float*>(a.next_buf + (size_t)slot * a.next_stride + a.next_hist + m)[comp] = y;
        if (m < a.cap) {
          if (a.out_cf32) reinterpret_cast<float*>(a.out_cf32 + (size_t)slot * a.cap + m)[comp] = y;
          if (a.out_i8) a.out_i8[((size_t)slot * a.cap + m) * 2 + comp] = to_i8(y, a.pack_scale);
        }
      }
    }
  }
}

// One launch: blocks [0, gridDim.x - 4) offer every tile to the full-tile code, the last four blocks offer tiles 0, 1, 2 and
// the last one to the edge code (block-uniform branch; each side returns at once when the tile is of the other kind).
template <int LOGG, int PASSES>
__global__ __launch_bounds__(256) void k_chan_dec(ChanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chan_smem[];
  float2* lds = reinterpret_cast<float2*>(chan_smem);
  const int ntiles = (int)gridDim.x - 4;
  if ((int)blockIdx.x < ntiles) chan_dec_tile<LOGG, PASSES, true>(a, lds, (int)blockIdx.x);
  else chan_dec_tile<LOGG, PASSES, false>(a, lds, (int)blockIdx.x - ntiles);
}

// The same as two launches (edge tiles first: grid.x = 4): where the merged kernel's register count — the larger of the
// two sides' — would cost a wave per SIMD (measured: D <= 32 and the two-pass D > 64 form are faster split, D = 33..64 merged).
template <int LOGG, int PASSES, bool FULL>
__global__ __launch_bounds__(256) void k_chan_dec_split(ChanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chan_smem[];
  chan_dec_tile<LOGG, PASSES, FULL>(a, reinterpret_cast<float2*>(chan_smem), (int)blockIdx.x);
}

// After a call: the newest h samples of [history | new samples] become the history. One workgroup per slot, staged
// through LDS because source and destination overlap when fewer than h samples arrived.
template <bool ROTATE>
__global__ __launch_bounds__(256) void k_chan_keep(ChanArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char chan_smem[];
  float2* lds = reinterpret_cast<float2*>(chan_smem);
  const int s = blockIdx.x;
  const int slot = a.slot[s];
  const int h = a.nt - 1, n_in = a.nin[s];
  for (int i = threadIdx.x; i < h; i += 256) {
    const int n = n_in - h + i;  // stream index relative to this call's first new sample
    float2 v;
    if (ROTATE) {
      if (n < 0) v = a.hist0[(size_t)slot * a.hist0_stride + (h + n)];
      else v = rotate(a.in_raw[n], phase_of(a.f0[s], a.df[s], n));
    } else {
      v = a.in_buf[(size_t)slot * a.in_stride + (h + n)];
    }
    lds[i] = v;
  }
  __syncthreads();
  float2* dst = ROTATE ? const_cast<float2*>(a.hist0) + (size_t)slot * a.hist0_stride : const_cast<float2*>(a.in_buf) + (size_t)slot * a.in_stride;
  for (int i = threadIdx.x; i < h; i += 256) dst[i] = lds[i];
}

// outputs of one resampler for n_in new samples given its (ctr, skip), and the state after them
int stage_outputs(int interp, int decim, int ctr, int skip, int n_in, int* ctr_after, int* skip_after) {
  long long nout = 0;
  if (n_in > skip) {
    const long long need = (long long)(n_in - skip) * interp - ctr;  // smallest m with ctr + m*D >= (n_in - skip) * I
    nout = need <= 0 ? 0 : (need + decim - 1) / decim;
  }
  const long long total = (long long)ctr + nout * decim;
  *ctr_after = (int)(total % interp);
  *skip_after = (int)((long long)skip + total / interp - n_in);
  return (int)nout;
}

void free_sc(sc_ctx* c) {
  if (!c) return;
  for (auto& st : c->stages) {
    (void)hipFree(st.d_arm);
    (void)hipFree(st.d_arm_dec);
    (void)hipFree(st.d_buf);
  }
  (void)hipFree(c->d_hist0);
  (void)hipFree(c->d_ptab);
  (void)hipFree(c->d_in);
  (void)hipFree(c->d_out_i8);
  (void)hipFree(c->d_out_cf32);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int run_stages(sc_ctx* c, const float2* d_iq, int nsamples, int8_t* d_out_i8, float* d_out_cf32, int32_t* counts, int cap) {
  const int nst = (int)c->stages.size();
  ChanArgs a{};
  a.nslots = 0;
  for (int ch = 0; ch < c->cfg.channels; ++ch) {
    if (counts) counts[ch] = 0;
    if (c->slots[ch].active) a.slot[a.nslots++] = ch;
  }
  if (a.nslots == 0 || nsamples == 0) return SS_OK;
  int nin[SC_MAX_CHANNELS];
  for (int s = 0; s < a.nslots; ++s) nin[s] = nsamples;
  for (int k = 0; k < nst; ++k) {
    Stage& st = c->stages[(size_t)k];
    const bool last = k == nst - 1;
    a.in_raw = d_iq;
    a.hist0 = c->d_hist0;
    a.hist0_stride = c->hist0_stride;
    a.in_buf = st.d_buf;
    a.in_stride = st.buf_stride;
    a.arm = st.d_arm;
    a.arm_dec = st.d_arm_dec;
    a.ptab = c->d_ptab;
    a.ptab_stride = c->ptab_stride;
    a.interp = st.interp;
    a.decim = st.decim;
    a.nt = st.nt;
    a.tile = st.tile;
    a.next_buf = last ? nullptr : c->stages[(size_t)k + 1].d_buf;
    a.next_stride = last ? 0 : c->stages[(size_t)k + 1].buf_stride;
    a.next_hist = last ? 0 : c->stages[(size_t)k + 1].nt - 1;
    a.out_cf32 = last ? reinterpret_cast<float2*>(d_out_cf32) : nullptr;
    a.out_i8 = last ? d_out_i8 : nullptr;
    a.cap = cap;
    a.pack_scale = c->cfg.pack_scale;
    int max_out = 0;
    int ctr_after[SC_MAX_CHANNELS], skip_after[SC_MAX_CHANNELS];
    for (int s = 0; s < a.nslots; ++s) {
      Slot& sl = c->slots[a.slot[s]];
      a.ctr0[s] = sl.ctr[k];
      a.skip0[s] = sl.skip[k];
      a.nin[s] = nin[s];
      a.nout[s] = stage_outputs(st.interp, st.decim, sl.ctr[k], sl.skip[k], nin[s], &ctr_after[s], &skip_after[s]);
      a.f0[s] = sl.f0;
      a.df[s] = sl.df;
      max_out = a.nout[s] > max_out ? a.nout[s] : max_out;
    }
    const size_t lds_bytes = sizeof(float2) * (size_t)st.lds_floats2;
    if (max_out > 0 && k == 0 && st.fast) {
      // every tile is offered to the full-tile code, tiles 0, 1, 2 and the last one also to the edge code (four more blocks,
      // or a launch of their own)
      const unsigned tiles = (unsigned)((max_out + st.tile - 1) / st.tile);
      const dim3 merged(tiles + 4u, (unsigned)a.nslots), grid(tiles, (unsigned)a.nslots), edge(4, (unsigned)a.nslots), block((unsigned)(64 * st.waves));
#define SC_LAUNCH_SPLIT(LOGG_, PASSES_)                                                                                          \
  do {                                                                                                                           \
    hipLaunchKernelGGL((k_chan_dec_split<LOGG_, PASSES_, false>), edge, block, lds_bytes, c->stream, a);                         \
    if (tiles > 1) hipLaunchKernelGGL((k_chan_dec_split<LOGG_, PASSES_, true>), grid, block, lds_bytes, c->stream, a);           \
  } while (0)
      switch (st.logg * 2 + (st.passes - 1)) {
        case 6: SC_LAUNCH_SPLIT(3, 1); break;
        case 8: SC_LAUNCH_SPLIT(4, 1); break;
        case 10: SC_LAUNCH_SPLIT(5, 1); break;
        case 12: hipLaunchKernelGGL((k_chan_dec<6, 1>), merged, block, lds_bytes, c->stream, a); break;
        default: SC_LAUNCH_SPLIT(6, 2); break;
      }
#undef SC_LAUNCH_SPLIT
    } else if (max_out > 0) {
      const dim3 grid((unsigned)((max_out + st.tile - 1) / st.tile), (unsigned)a.nslots);
      if (k == 0) hipLaunchKernelGGL(k_chan_stage<true>, grid, dim3(256), lds_bytes, c->stream, a);
      else hipLaunchKernelGGL(k_chan_stage<false>, grid, dim3(256), lds_bytes, c->stream, a);
    }
    if (st.nt > 1) {
      const size_t keep_bytes = sizeof(float2) * (size_t)(st.nt - 1);
      if (k == 0) hipLaunchKernelGGL(k_chan_keep<true>, dim3((unsigned)a.nslots), dim3(256), keep_bytes, c->stream, a);
      else hipLaunchKernelGGL(k_chan_keep<false>, dim3((unsigned)a.nslots), dim3(256), keep_bytes, c->stream, a);
    }
    for (int s = 0; s < a.nslots; ++s) {
      Slot& sl = c->slots[a.slot[s]];
      sl.ctr[k] = ctr_after[s];
      sl.skip[k] = skip_after[s];
      nin[s] = a.nout[s];
    }
  }
  for (int s = 0; s < a.nslots; ++s) {
    Slot& sl = c->slots[a.slot[s]];
    double f = sl.f0 + (double)nsamples * sl.df;
    sl.f0 = f - std::floor(f);
    if (counts) counts[a.slot[s]] = nin[s];
  }
  SC_HIP(c, hipGetLastError());
  return SS_OK;
}

}  // namespace

extern "C" {

void sc_default_config(sc_config* cfg, int32_t sample_rate, int32_t bandwidth) {
  if (!cfg) return;
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = SC_ABI_VERSION;
  cfg->sample_rate = sample_rate;
  cfg->bandwidth = bandwidth;
  cfg->threshold = 125;  // RESAMPLER_THRESHOLD, config.h:20
  cfg->channels = 4;
  cfg->max_samples = 1 << 23;
  cfg->pack_scale = 127.0f;  // recorder.cpp:36
  cfg->device_id = 0;
}

const char* sc_last_error(const sc_ctx* c) { return c ? c->err : g_sc_create_err; }

int sc_create(const sc_config* cfg, sc_ctx** out) {
  if (!cfg || !out) return sc_fail(nullptr, SS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != SC_ABI_VERSION) return sc_fail(nullptr, SS_ERR_INVALID, "abi_version %u != %u", cfg->abi_version, SC_ABI_VERSION);
  if (cfg->sample_rate <= 0 || cfg->bandwidth <= 0 || cfg->threshold < 2) return sc_fail(nullptr, SS_ERR_INVALID, "bad sample_rate/bandwidth/threshold");
  if (cfg->channels < 1 || cfg->channels > SC_MAX_CHANNELS) return sc_fail(nullptr, SS_ERR_INVALID, "channels %d not in 1..%d", cfg->channels, SC_MAX_CHANNELS);
  if (cfg->max_samples < 1) return sc_fail(nullptr, SS_ERR_INVALID, "max_samples %d", cfg->max_samples);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return sc_fail(nullptr, SS_ERR_NO_DEVICE, "no HIP device available");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return sc_fail(nullptr, SS_ERR_NO_DEVICE, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
  const auto factors = resampler_factors(cfg->sample_rate, cfg->bandwidth, cfg->threshold);
  if (factors.size() > SC_MAX_STAGES) return sc_fail(nullptr, SS_ERR_INVALID, "%zu resampler stages > %d", factors.size(), SC_MAX_STAGES);
  sc_ctx* c = new sc_ctx;
  c->cfg = *cfg;
#define SC_CREATE_HIP(call)                                                           \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      sc_fail(nullptr, SS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
      free_sc(c);                                                                     \
      return SS_ERR_HIP;                                                              \
    }                                                                                 \
  } while (0)
  SC_CREATE_HIP(hipSetDevice(cfg->device_id));
  SC_CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  long long max_in = cfg->max_samples;
  for (const auto& f : factors) {
    Stage st;
    const int g = std::gcd(f.first, f.second);
    st.interp = f.first / g;
    st.decim = f.second / g;
    st.taps = design_resampler_taps(st.interp, st.decim);
    st.ntaps = (int)st.taps.size();
    st.nt = (st.ntaps + st.interp - 1) / st.interp;
    st.max_in = (int)max_in;
    st.max_out = (int)((max_in * st.interp) / st.decim + 2);
    // tile: as many outputs as keep the staged span within 64 KiB of LDS
    st.tile = 64;
    for (;;) {
      const long long span = (long long)(st.tile - 1) * st.decim / st.interp + 1 + st.nt;  // p_last - p_first + nt
      const long long cols = (span + st.decim) / st.decim + 2;
      const long long elems = (cols | 1) * st.decim;
      if (elems * (long long)sizeof(float2) <= 64 * 1024 || st.tile == 1) {
        st.lds_floats2 = (int)elems;
        break;
      }
      st.tile /= 2;
    }
    // first stage, interpolation 1, at most 33 taps per branch (GNU Radio's design always gives 33): branch kernel
#ifdef SS_DIAG
    const char* force_generic_env = getenv("SC_GENERIC");  // A/B measurements, libspecscan_diag.so only
    const bool force_generic = force_generic_env && force_generic_env[0] == '1';
#else
    const bool force_generic = false;
#endif
    if (c->stages.empty() && st.interp == 1 && st.decim <= 128 && st.ntaps <= kDecA * st.decim && !force_generic) {
      const int generic_tile = st.tile, generic_lds = st.lds_floats2;
      st.fast = true;
      st.passes = st.decim > 64 ? 2 : 1;
      st.logg = 3;
      while ((1 << st.logg) < st.decim && st.logg < 6) ++st.logg;
      const int groups = 64 >> st.logg;
      for (st.waves = 4; st.waves >= 1; st.waves /= 2) {
        st.tile = st.waves * groups * kDecR;
        st.lds_floats2 = (st.tile - 1) * st.decim + kDecA * st.decim;
        if ((long long)st.lds_floats2 * (long long)sizeof(float2) <= 64 * 1024) break;
      }
      if (st.waves < 1) {
        st.fast = false;
        st.waves = 4;
        st.tile = generic_tile;
        st.lds_floats2 = generic_lds;
      } else {
        c->ptab_stride = st.lds_floats2;
      }
    }
    if ((long long)st.lds_floats2 * (long long)sizeof(float2) > 64 * 1024 || (long long)(st.nt - 1) * (long long)sizeof(float2) > 64 * 1024) {
      sc_fail(nullptr, SS_ERR_INVALID, "resampler %d/%d needs %d taps per arm: does not fit the LDS tile", st.interp, st.decim, st.nt);
      free_sc(c);
      return SS_ERR_INVALID;
    }
    std::vector<float> arm((size_t)st.interp * (size_t)st.nt, 0.0f);
    for (int k = 0; k < st.ntaps; ++k) arm[(size_t)(k % st.interp) * st.nt + k / st.interp] = st.taps[(size_t)k];
    SC_CREATE_HIP(hipMalloc(&st.d_arm, sizeof(float) * arm.size()));
    SC_CREATE_HIP(hipMemcpy(st.d_arm, arm.data(), sizeof(float) * arm.size(), hipMemcpyHostToDevice));
    if (st.fast) {
      const int pitch = 64 * st.passes;
      std::vector<float> dec((size_t)kDecA * (size_t)pitch, 0.0f);
      for (int t = 0; t < kDecA; ++t)
        for (int b = 0; b < st.decim; ++b)
          if (st.decim * t + b < st.ntaps) dec[(size_t)t * pitch + b] = st.taps[(size_t)(st.decim * t + b)];
      SC_CREATE_HIP(hipMalloc(&st.d_arm_dec, sizeof(float) * dec.size()));
      SC_CREATE_HIP(hipMemcpy(st.d_arm_dec, dec.data(), sizeof(float) * dec.size(), hipMemcpyHostToDevice));
    }
    if (!c->stages.empty()) {
      st.buf_stride = (long long)(st.nt - 1) + st.max_in;
      SC_CREATE_HIP(hipMalloc(&st.d_buf, sizeof(float2) * (size_t)st.buf_stride * (size_t)cfg->channels));
      SC_CREATE_HIP(hipMemsetAsync(st.d_buf, 0, sizeof(float2) * (size_t)st.buf_stride * (size_t)cfg->channels, c->stream));  // zero history
    }
    max_in = st.max_out;
    c->stages.push_back(st);
  }
  c->hist0_stride = c->stages[0].nt > 1 ? c->stages[0].nt - 1 : 1;
  SC_CREATE_HIP(hipMalloc(&c->d_hist0, sizeof(float2) * (size_t)c->hist0_stride * (size_t)cfg->channels));
  SC_CREATE_HIP(hipMemsetAsync(c->d_hist0, 0, sizeof(float2) * (size_t)c->hist0_stride * (size_t)cfg->channels, c->stream));
  if (c->ptab_stride > 0) {
    SC_CREATE_HIP(hipMalloc(&c->d_ptab, sizeof(float2) * (size_t)c->ptab_stride * (size_t)cfg->channels));
    SC_CREATE_HIP(hipMemsetAsync(c->d_ptab, 0, sizeof(float2) * (size_t)c->ptab_stride * (size_t)cfg->channels, c->stream));
  }
  SC_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chan_stage<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SC_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chan_stage<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SC_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chan_keep<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SC_CREATE_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_chan_keep<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
  SC_CREATE_HIP(hipStreamSynchronize(c->stream));
#undef SC_CREATE_HIP
  *out = c;
  return SS_OK;
}

void sc_destroy(sc_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device_id);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  free_sc(c);
}

int sc_stage_count(const sc_ctx* c) { return c ? (int)c->stages.size() : 0; }

int sc_stage_info(const sc_ctx* c, int32_t stage, int32_t* interpolation, int32_t* decimation, int32_t* ntaps) {
  if (!c || stage < 0 || stage >= (int)c->stages.size()) return SS_ERR_INVALID;
  const Stage& st = c->stages[(size_t)stage];
  if (interpolation) *interpolation = st.interp;
  if (decimation) *decimation = st.decim;
  if (ntaps) *ntaps = st.ntaps;
  return SS_OK;
}

int sc_stage_taps(const sc_ctx* c, int32_t stage, float* taps) {
  if (!c || !taps || stage < 0 || stage >= (int)c->stages.size()) return SS_ERR_INVALID;
  const Stage& st = c->stages[(size_t)stage];
  memcpy(taps, st.taps.data(), sizeof(float) * st.taps.size());
  return SS_OK;
}

int32_t sc_output_capacity(const sc_ctx* c, int32_t nsamples) {
  if (!c || nsamples < 0) return 0;
  long long n = nsamples;
  for (const auto& st : c->stages) n = (n * st.interp) / st.decim + 2;
  return (int32_t)n;
}

int sc_start(sc_ctx* c, int32_t channel, int32_t shift_hz) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (channel < 0 || channel >= c->cfg.channels) return sc_fail(c, SS_ERR_INVALID, "channel %d out of range", channel);
  Slot& sl = c->slots[channel];
  // recorder.cpp:64: set_phase_inc(2.0l * M_PIl * (double(-shift) / float(sampleRate))) -> rotator_cc_impl::set_phase_inc(double):
  // exp(gr_complex(0, phase_inc)) -> the angle becomes a float; rotator::set_phase_incr divides by the magnitude
  const double ratio = (double)(-shift_hz) / (float)c->cfg.sample_rate;
  const double phase_inc = (double)(2.0L * 3.141592653589793238462643383279502884L * (long double)ratio);
  const float ang = (float)phase_inc;
  const float re = cosf(ang), im = sinf(ang);
  const float mag = hypotf(re, im);
  sl.inc_re = re / mag;
  sl.inc_im = im / mag;
  sl.df = atan2((double)sl.inc_im, (double)sl.inc_re) / (2.0 * M_PI);  // what one multiplication by the fp32 increment turns the phase by
  if (c->d_ptab) {
    SC_HIP(c, hipSetDevice(c->cfg.device_id));
    const int n = (int)c->ptab_stride;
    hipLaunchKernelGGL(k_chan_ptab, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_ptab + (size_t)channel * c->ptab_stride, n, sl.df);
    SC_HIP(c, hipGetLastError());
  }
  sl.active = true;
  return SS_OK;
}

int sc_stop(sc_ctx* c, int32_t channel) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (channel < 0 || channel >= c->cfg.channels) return sc_fail(c, SS_ERR_INVALID, "channel %d out of range", channel);
  c->slots[channel].active = false;
  return SS_OK;
}

int sc_is_recording(const sc_ctx* c, int32_t channel) {
  if (!c || channel < 0 || channel >= c->cfg.channels) return 0;
  return c->slots[channel].active ? 1 : 0;
}

int sc_sync(sc_ctx* c) {
  if (!c) return SS_ERR_INVALID;
  SC_HIP(c, hipSetDevice(c->cfg.device_id));
  SC_HIP(c, hipStreamSynchronize(c->stream));
  return SS_OK;
}

int sc_process_device(sc_ctx* c, const void* d_iq, int32_t nsamples, int8_t* d_out_i8, float* d_out_cf32, int32_t* counts, int32_t cap) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (nsamples < 0 || (nsamples > 0 && !d_iq) || cap < 0) return sc_fail(c, SS_ERR_INVALID, "bad iq/nsamples/cap");
  if (nsamples > c->cfg.max_samples) return sc_fail(c, SS_ERR_BATCH, "nsamples %d > max_samples %d", nsamples, c->cfg.max_samples);
  SC_HIP(c, hipSetDevice(c->cfg.device_id));
  return run_stages(c, static_cast<const float2*>(d_iq), nsamples, d_out_i8, d_out_cf32, counts, cap);
}

int sc_process(sc_ctx* c, const void* iq, int32_t nsamples, int8_t* out_i8, float* out_cf32, int32_t* counts, int32_t cap) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (nsamples < 0 || (nsamples > 0 && !iq) || cap < 0 || !counts) return sc_fail(c, SS_ERR_INVALID, "bad iq/nsamples/cap/counts");
  if (nsamples > c->cfg.max_samples) return sc_fail(c, SS_ERR_BATCH, "nsamples %d > max_samples %d", nsamples, c->cfg.max_samples);
  SC_HIP(c, hipSetDevice(c->cfg.device_id));
  if (!c->d_in) SC_HIP(c, hipMalloc(&c->d_in, sizeof(float2) * (size_t)c->cfg.max_samples));
  if (cap > c->out_cap_alloc) {
    (void)hipFree(c->d_out_i8);
    (void)hipFree(c->d_out_cf32);
    c->d_out_i8 = nullptr;
    c->d_out_cf32 = nullptr;
    c->out_cap_alloc = 0;
    SC_HIP(c, hipMalloc(&c->d_out_i8, (size_t)2 * (size_t)cap * (size_t)c->cfg.channels));
    SC_HIP(c, hipMalloc(&c->d_out_cf32, sizeof(float2) * (size_t)cap * (size_t)c->cfg.channels));
    c->out_cap_alloc = cap;
  }
  if (nsamples > 0) SC_HIP(c, hipMemcpyAsync(c->d_in, iq, sizeof(float2) * (size_t)nsamples, hipMemcpyHostToDevice, c->stream));
  const int st = run_stages(c, c->d_in, nsamples, out_i8 ? c->d_out_i8 : nullptr, out_cf32 ? c->d_out_cf32 : nullptr, counts, cap);
  if (st != SS_OK) return st;
  for (int ch = 0; ch < c->cfg.channels; ++ch) {
    const int n = counts[ch] < cap ? counts[ch] : cap;
    if (n <= 0) continue;
    if (out_i8) SC_HIP(c, hipMemcpyAsync(out_i8 + (size_t)2 * (size_t)cap * (size_t)ch, c->d_out_i8 + (size_t)2 * (size_t)cap * (size_t)ch, (size_t)2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (out_cf32) SC_HIP(c, hipMemcpyAsync(out_cf32 + (size_t)2 * (size_t)cap * (size_t)ch, c->d_out_cf32 + (size_t)2 * (size_t)cap * (size_t)ch, sizeof(float2) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
  }
  SC_HIP(c, hipStreamSynchronize(c->stream));
  return SS_OK;
}

int sc_transmission_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* iq_i8, int32_t nsamples, uint8_t* out,
                            int32_t cap) {
  const int header = (int)(sizeof(uint64_t) + 2 * sizeof(int32_t) + sizeof(uint32_t));
  if (nsamples < 0 || (nsamples > 0 && !iq_i8 && out)) return SS_ERR_INVALID;
  const long long total = (long long)header + 2LL * nsamples;
  if (!out) return (int)total;
  if (total > cap) return SS_ERR_INVALID;
  const int32_t start = frequency - sample_rate / 2, stop = frequency + sample_rate / 2;  // data_controller.cpp:28-29
  const uint32_t rate = (uint32_t)sample_rate;
  size_t off = 0;
  memcpy(out + off, &time_ms, sizeof(time_ms));
  off += sizeof(time_ms);
  memcpy(out + off, &start, sizeof(start));
  off += sizeof(start);
  memcpy(out + off, &stop, sizeof(stop));
  off += sizeof(stop);
  memcpy(out + off, &rate, sizeof(rate));
  off += sizeof(rate);
  for (int i = 0; i < 2 * nsamples; ++i) out[off + (size_t)i] = (uint8_t)iq_i8[i] ^ 0x80u;  // data_controller.cpp:38-40
  return (int)total;
}

}  // extern "C"
