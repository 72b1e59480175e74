// reference_nan.h — SS_FLAG_REFERENCE_NAN: what the reference does after a degenerate frame, reproduced instead of repaired.
//
// The reference keeps two kinds of never-re-zeroed fp32 running sums, and a value that is not finite poisons both for good:
//
//   Averager (sources/radio/averager.cpp:14-25,40-50): per bin, sum -= oldest row; sum += new row. A -inf (a bin of exact
//     zeros: log10f(0), psd.cpp:19) makes the sum -inf while the row is inside the 21-frame window and NaN from the frame it
//     leaves (-inf - -inf); a NaN makes it NaN at once. Nothing clears it but Averager::reset (averager.cpp:27-34), which
//     Transmission::resetBuffers calls on every retune (transmission.cpp:42-55).
//   average() (sources/utils/utils.cpp:31-53): ONE running sum walks each row of time means from bin -10 upwards. A NaN at bin p
//     enters at i = p - 10 and never leaves: every output from p - 10 on is NaN. A -inf / +inf at bin p makes the outputs
//     p - 10 .. p + 10 infinite and everything from p + 11 on NaN (inf - inf when it leaves the window); an inf of the other
//     sign entering while it is still inside makes NaN at once.
//
// So the reference's avg row of frame f is: as computed for the bins below bad_from(f), NaN from bad_from(f) on — and no
// candidate there, whatever the signal (`startLevel <= NaN` is false, transmission.cpp:91) —, with
//     bad_from(f) = min( N(f) - 10,  I(f) + 11,  O(f) - 10 if O(f) <= I(f) + 20 )
//     N(f) = the lowest bin whose time sum is NaN at frame f: the lowest NaN bin of ANY frame since the reset, or the lowest
//            non-finite bin of any frame that has left the window (g <= f - 21)
//     I(f) = the lowest bin with an infinite value inside the window (frames f - 20 .. f), O(f) = the lowest one of the other sign.
// Only the FIRST non-finite bins of each frame matter for that, three integers per frame (first NaN, first -inf, first +inf).
// The engine's own tiles restart their sums every 16 frames and every 16 bins (detect_fused.h) and so compute the bins below
// bad_from(f) like the reference (direct sums give the same infinities where the window holds one); what this file adds, between
// the detect and the emit stage of a call, is the poison: mask bits and avg values from bad_from(f) on.
//
// Exact for NaN and -inf — the values degenerate input produces (zeros, NaN samples). +inf (|X|^2 overflowing: samples beyond
// 1e19) is handled through the same first-bin summary, which leaves out one contrived case: infinities of both signs at the SAME
// bin within 21 frames where that bin is not the first infinite bin of both rows.
// Not covered either: a +inf dB value in a LEARNING frame (|X|^2 overflowing while the ceiling is learned). The reference's ceiling
// then is +inf at that bin (std::max, noise_learner.cpp:18-26), every later rel value there -inf and the Averager's sums NaN after
// 21 frames, while the dB rows this scan looks at stay finite: the engine reports -inf averages where the reference has NaN (no
// candidates either way: neither passes `start_level <= avg`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ss {

struct NanState {
  int cum_nan;       // lowest NaN bin of any frame since the last reset (n: none)
  int cum_inf_old;   // lowest infinite bin of any frame that has left the 21-frame window
  int ring[21][2];   // first -inf / +inf bin of the newest 21 frames
  int pos;           // ring slot of the next frame
};

__global__ void k_nan_state_reset(NanState* st, int n) {
  if (threadIdx.x == 0) {
    st->cum_nan = n;
    st->cum_inf_old = n;
    st->pos = 0;
  }
  if (threadIdx.x < 21) st->ring[threadIdx.x][0] = st->ring[threadIdx.x][1] = n;
}

// One workgroup per frame: the first NaN / -inf / +inf bin of the frame's dB row -> nf[4 f + 0 .. 2] (n: none). Learning frames
// count as clean: NoiseLearner hands -100 on for them whatever the PSD holds (noise_learner.cpp:45-51).
__global__ __launch_bounds__(256) void k_nonfinite_scan(const float* __restrict__ psd, int n, int n_learn, int* __restrict__ nf) {
  const int f = blockIdx.x, t = threadIdx.x;
  int first[3] = {n, n, n};
  if (f >= n_learn) {
    const float* row = psd + (size_t)f * n;
    for (int i = t; i < n; i += 256) {
      const float v = row[i];
      if (v != v) first[0] = min(first[0], i);
      else if (v == -__builtin_inff()) first[1] = min(first[1], i);
      else if (v == __builtin_inff()) first[2] = min(first[2], i);
    }
  }
  __shared__ int part[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    int v = first[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d));
    if ((t & 63) == 0) part[t >> 6][k] = v;
  }
  __syncthreads();
  if (t < 3) nf[4 * f + t] = min(min(part[0][t], part[1][t]), min(part[2][t], part[3][t]));
}

// One lane walks the batch's frames in order: bad_from[f] as derived above, the state carried on to the next batch.
// pushed_before = Averager::m_frames before the batch (saturated at 21): until 21 frames have been pushed the Averager hands out
// -100 whatever its sums hold (averager.cpp:53-60), so nothing shows yet — the sums are poisoned all the same.
__global__ void k_nan_plan(NanState* st, const int* __restrict__ nf, int nframes, int n, int pushed_before, int* __restrict__ bad_from) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int cum_nan = st->cum_nan, cum_old = st->cum_inf_old, pos = st->pos;
  for (int f = 0; f < nframes; ++f) {
    const int fn = nf[4 * f], fm = nf[4 * f + 1], fp = nf[4 * f + 2];
    // the row that leaves the window with this push (21 frames ago) turns its infinities into NaN for good
    cum_old = min(cum_old, min(st->ring[pos][0], st->ring[pos][1]));
    st->ring[pos][0] = fm;
    st->ring[pos][1] = fp;
    pos = pos == 20 ? 0 : pos + 1;
    cum_nan = min(cum_nan, fn);
    int wm = n, wp = n;
    for (int k = 0; k < 21; ++k) {
      wm = min(wm, st->ring[k][0]);
      wp = min(wp, st->ring[k][1]);
    }
    const int first_inf = min(wm, wp), other = max(wm, wp);
    const int first_nan = min(cum_nan, cum_old);
    int bad = n;
    if (first_nan < n) bad = min(bad, first_nan - 10);
    if (first_inf < n) bad = min(bad, first_inf + 11);
    if (other < n && other <= first_inf + 20) bad = min(bad, other - 10);
    if (pushed_before + f + 1 < 21) bad = n;
    bad_from[f] = bad < 0 ? 0 : bad;
  }
  st->cum_nan = cum_nan;
  st->cum_inf_old = cum_old;
  st->pos = pos;
}

// One workgroup per frame, between the call's detect and emit stages: mask bits from bad_from[f] on are cleared, the frame's
// candidate count is what is left, and where the call keeps or hands out a full avg plane its values from bad_from[f] on are NaN.
__global__ __launch_bounds__(256) void k_nan_apply(const int* __restrict__ bad_from, int n, uint32_t* __restrict__ maskbits, int* __restrict__ counts, float* __restrict__ avg_full) {
  const int f = blockIdx.x, t = threadIdx.x;
  const int bad = bad_from[f];
  if (bad >= n) return;  // (workgroup-uniform)
  const int words = n >> 5;
  uint32_t* row = maskbits + (size_t)f * words;
  int cnt = 0;
  for (int w = t; w < words; w += 256) {
    uint32_t v = row[w];
    const int lo = 32 * w;
    if (lo + 32 > bad) {
      const uint32_t keep = lo >= bad ? 0u : ((1u << (bad - lo)) - 1u);
      if (v & ~keep) row[w] = v & keep;
      v &= keep;
    }
    cnt += __popc(v);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
  __shared__ int part[4];
  if ((t & 63) == 0) part[t >> 6] = cnt;
  __syncthreads();
  if (t == 0) counts[f] = part[0] + part[1] + part[2] + part[3];
  if (avg_full) {
    float* arow = avg_full + (size_t)f * n;
    const float qnan = __builtin_nanf("");
    for (int i = bad + t; i < n; i += 256) arow[i] = qnan;
  }
}

}  // namespace ss
