// detect_fused.h — fused back end for the reference's own grouping (GROUPING_X = GROUPING_Y = 21,
// sources/config.h:28-29). Other groupings use the unfused kernels of detect_kernels.h.
//
// k_detect_fused reads the PSD plane ONCE and produces candidate mask bits + per-frame counts; nothing
// else is written unless the caller asked for the rel / avg planes:
//   rel(f,i) = psd(f,i) - thr(i)   (-100 for learning frames; ring rows before the batch come from hist_in)
//                                                            sources/radio/blocks/noise_learner.cpp:49,55
//   avgY     = mean over the newest G frames (or -100 during the averager warm-up)   sources/radio/averager.cpp:52-61
//   avgXY    = centred GX-bin mean of avgY, window clipped at the band edges         sources/utils/utils.cpp:31-53
//   hit      = start_level <= avgXY && pass(i)                                       sources/radio/blocks/transmission.cpp:91
//
// Tiling: one workgroup = TF frames x 256 bins. Frame tiles are aligned to the ABSOLUTE frame index (frames
// since the last reset), not to the batch: tile boundaries, and with them every rounding below, do not depend
// on how the frame stream is cut into batches (bit for bit: tests/test_gpu_fullsize.py). A batch that starts
// in the middle of a tile re-reads the tile's earlier rows from the ring, which therefore holds
// H = G-1 + TF-1 rows instead of the Averager's G-1.
//   phase 1: thread = column. The G-1+TF rel values of the column sit in registers. The first time mean of
//            the tile is a fixed-order G-term sum, oldest frame first; the next TF-1 slide it the way the
//            reference's Averager does (sum -= oldest row; sum += new row, averager.cpp:14-25,40-50) —
//            restarted every TF frames, so the drift of the reference's never-re-zeroed sum cannot build up.
//   the avgY tile goes through LDS once and changes owner
//   phase 2: thread = (frame, 16-bin segment). 36 avgY values in registers; the first window of the
//            segment is a fixed-order GX-term sum, the next 15 slide it the way the reference does
//            (sum -= leaving bin; sum += entering bin, utils.cpp:39-48) — restarted every 16 bins, so the
//            drift of the reference's whole-row running sum never builds up.
// Tiles of a steady-state batch (no ring rows, no learning frames, all TF frames inside the batch) take a
// straight-line path; the few others take a general path. The tiles that hold the newest H frames of the
// batch also write them to hist_out, the ring the next batch starts from.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detect_kernels.h"
#include "fft1024_kernels.h"
#include "fft256_kernels.h"
#include "fft8192_v2.h"

// Cache policy of the detect tiles' row loads: 0 = default, 1 = non-temporal (A/B builds)
#ifndef SS_DET_NT
#define SS_DET_NT 0
#endif

// 8192 points, deep pipelining: the averaging tiles whose first rows are halo frames on a straight-line path of their own (1), or on the
// general path like every non-steady tile (0: ships). Measured in round 6 (profiles/r06/s16_summary.txt, alternating runs, 264 GPU tests
// green with it): 23.2-23.6 against 23.0-23.1 us per step in 200 steps, 26.2-26.8 against 26.2-26.7 in 20, 31.3 against 31.2 with every
// tile evaluated — nothing: neither the step nor the drain's detect launch (21.0 / 14.9 us either way) is as long as its slowest tile.
// Build-time so that the two can be timed against each other (scripts/build_ab.py: steadyhalo).
#ifndef SS_STEADY_HALO
#define SS_STEADY_HALO 0
#endif

namespace ss {

__device__ __forceinline__ float load_row_value(const char* p) {
#if SS_DET_NT
  return __builtin_nontemporal_load(reinterpret_cast<const float*>(p));
#else
  return *reinterpret_cast<const float*>(p);
#endif
}

// Tile culling (8192-point frames): a tile that provably holds no candidate need not read its 36 x 276 rel values at all. The
// bound: the FFT role leaves, per frame and 256-bin tile column, the maximum dB value M_g over the column's bins and 32 more on
// either side (Fft8192Args::segsum); with m the minimum of the noise ceiling over the same 320 bins (thr_tilemin, kept up to date
// by k_thr_tilemin) every rel = fl(psd - thr) of frame g is <= fl(M_g - m) because rounding is monotonic, so the 21-bin mean of
// any frame is, and the 21 x 21 mean ending at frame f is at most mean(M_{f-20} .. M_f) - m. The fp32 sums behind a mean of 441
// such values (|rel| < 1000 dB: finite dB values lie in [-450, 400]) and behind the bound are off by < 0.01 dB, so a tile whose
// bound stays below start_level - kCullMargin for each of its 16 frames has no candidate whatever the rounding. -inf and NaN rows need no care: they make no candidates either (a NaN never passes `start_level <= avg`), and
// +inf is not below any level. The decision is exact — culled tiles produce what the full evaluation would have produced:
// zero mask bits, no counts (tests/test_gpu_cull.py: identical candidate lists with SS_FLAG_NO_CULL and against oracle/_ref).
// What a culled tile must not cost is a workgroup: inside the step kernel every detect workgroup holds one of a CU's four
// slots for ~7 us whatever it does (its first loads queue behind the frames the CU streams), and with five tiles in six
// culled that was a third of the step; a launch of their own behind the step costs two kernel boundaries
// per queue instead (measured: worse). So the decision is taken for ALL tiles of a call by a handful of workgroups (the
// plan role of k_scan_step, dispatched first: one lane per tile) which compact the tiles that must be evaluated into a
// list and clear the mask words of the others; the FFT workgroups of the same launch — which live ~14 us before they
// need to know — pick the lists up when their frame is done (scan_step.h). Plan workgroup s owns the tiles of 8 (or 4) tile
// columns and a list of its own; consumer p serves list p mod S, entries 2 (p div S) and the next one, so all it has to
// read is ONE header word, count | ready. The hand-over inside the launch follows MI355X_MICROARCH.md: write-through (sc1) stores
// for the entries, drained (vmcnt 0 — not a release fence: that would write back an L2 full of dB rows being produced),
// then the header word, sc1; consumers read header and entries with sc1 loads. Plan workgroups never wait for anybody
// and are dispatched before every consumer in practice — but HIP promises nothing about dispatch order, so no consumer
// relies on it: a consumer polls its list's header a bounded number of times (StepArgs::wait_limit) and then makes the plan
// of its list ITSELF (plan_tiles again, with the same inputs: the same list, the same count, written to the same words — two
// writers of identical values are harmless) and goes on from there; ss_get_stats counts those workgroups. Nothing in a launch
// waits without bound (scan_step.h; tests/test_gpu_wait_bound.py runs the chain with plan workgroups that never publish,
// and under HSA_CU_MASK with a handful of CUs).
constexpr float kCullMargin = 0.0625f;
// (ss_get_stats) The counters live in kStatShards copies a cache line apart and a workgroup adds to the copy of its block index:
// a thousand plan workgroups adding to ONE word queue up behind each other at 12 ns apiece (measured: k_plan_long went from 7 to
// 20 us per 16-frame call of 2^20 points); the host adds the copies up.
enum { kStatTested = 0, kStatCulled = 1, kStatWaitFallbacks = 2, kStatWords = 4, kStatShards = 64, kStatShardStride = 16 };
__device__ __forceinline__ unsigned long long* stat_word(unsigned long long* stats, int which) { return stats + (blockIdx.x & (kStatShards - 1)) * kStatShardStride + which; }

// s / D for a compile-time integer D, correctly rounded like the reference's `sum / count`
// (float / int -> IEEE division): q0 = s * RN(1/D), one Newton correction with exact residuals (FMA).
// Verified against the hardware division for every finite float by ss_selftest (tests/test_gpu_selftest.py).
// An infinite sum (a -inf dB row inside the window: a frame of zeros) stays the infinity the IEEE division gives — the residual
// inf - inf is not a number, and then the first quotient is the answer (round 4: until then the engine's avg plane held NaN where
// the reference's holds -inf; no candidate either way).
template <int D>
__device__ __forceinline__ float div_const(float s) {
  constexpr float r = 1.0f / (float)D;
  const float q0 = s * r;
  const float e = fmaf(-(float)D, q0, s);
  const float q1 = fmaf(e, r, q0);
  return q1 == q1 ? q1 : q0;
}

template <int G, int GX, int TF, int TB_ = 256>
struct DetectTile {
  static constexpr int A = GX / 2;
  static constexpr int TB = TB_;           // bins per tile = threads per workgroup
  static constexpr int P = TB + 2 * A;     // tile pitch in floats (276 for GX = 21: rows stay 16-byte aligned)
  static constexpr int ROWS = G - 1 + TF;  // rel rows a column needs
  static constexpr int H = G - 1 + TF - 1; // ring rows kept between batches (a tile may start TF-1 frames before the batch)
  static constexpr int SEGW = 16;          // bins per phase-2 thread
  static constexpr int NSEG = TB / SEGW;   // 16
  static constexpr int YW = SEGW + 2 * A;  // 36 avgY values per phase-2 thread
  static_assert((2 * A) % 4 == 0 && P % 4 == 0 && YW % 4 == 0, "tile rows must stay 16-byte aligned");
  static_assert(TF == 16 || TF == 32, "phase 2 pairs lanes l and l+TF inside one wave");
};

struct DetectArgs {
  const float* psd;
  const float* thr;
  const float* hist_in;
  float* hist_out;
  // Rows before the batch straight from the PREVIOUS call's PSD plane instead of the ring (null: the ring). rel = psd - thr is
  // the same fp32 subtraction either way, so the results are the same bits; what changes is the dependence: the ring was
  // written by the previous call's detect stage, the plane by its FFT stage (specscan.hip, deep pipelining). Only between
  // calls without learning frames; halo_rows = frames of that plane (>= H).
  const float* halo_psd;
  int halo_rows;
  // Ring rows that hold dB values, not noise-relative ones (round 5): the FFT stage of a long transform's detect-mode call leaves its
  // rows in the ring's buffer as dB values — the ceiling loads cost its launch 7-9 % — and the tiles that are evaluated subtract the
  // ceiling, the same fp32 subtraction on the same values (noise_learner.cpp:55). Rows of the window from batch-relative frame
  // ring_db_from (<= 0) on are such rows, the ones before it are noise-relative (learning frames' -100, rows a call with planes wrote);
  // 0: none. The host settles the window (subtracts the ceiling in place) before anything else writes to it or the ceiling changes.
  int ring_db_from;
  int n, nframes, n_learn, pushed_before;
  int shift;  // (frames since reset, before this batch) mod TF: tile t covers batch frames [t*TF - shift, t*TF - shift + TF)
  float start_level;
  const uint8_t* pass;
  uint32_t* maskbits;
  int* counts;
  float* rel_out;
  float* avg_out;
  float* avg_sparse;
  // Spectrogram side branch (k_detect_fused<..., SPEC = true> only; spectrogram.cpp:45-60)
  float* spec_partial;             // [frame tile][spec_n]: the tile's frames, bin-decimated and summed in frame order
  const float* spec_prev_partial;  // the previous launch's partial sums, not yet added to their container (or null)
  float* spec_prev_sum;            // [spec_n] Container::m_sum they belong to
  int spec_prev_tiles;
  int spec_m, spec_n;              // m_decimatorFactor (a power of two <= 256), m_outputSize
  // Tile culling (tile_is_culled, 8192 points): per-column maxima of this batch's PSD rows (Fft8192Args::segsum), or null
  const float* segsum;             // [32][seg_pitch]
  int seg_pitch;
  // ... and of the halo_rows frames before the batch (deep pipelining: the frames the launch transformed once more into halo_psd left
  // their maxima too, [32][kHaloSegPitch], frame -halo_rows + i at [c][i]); null: tiles that reach back before the batch are not tested
  const float* halo_segsum;
  const float* thr_tilemin;        // [32] min of thr over bins [256 c - 32, 256 c + 288)
  // [s], s < 8: nonzero once list s has been written through; [16 + kLiveCap s ...] the tiles of plan workgroup s that must be
  // evaluated; [kLiveCounts + kLiveCopyStride k + s], k < kLiveCopies: kLiveReady | their number, published as soon as it is
  // known (most consumers learn from it that there is nothing for them), in copies a page apart.
  // Written by the plan role (plan_tiles), consumed by the workgroups of the same launch (list_pair), the header set back to
  // zero by the call's emit stage.
  int* live;
  // Tile culling, long transforms (N = 256 x N2; k_plan_long below): [0] = how many tiles must be evaluated, [1 ...] = their
  // numbers, written by a launch of its own between the call's FFT and detect stages; null = every tile. hist_by_fft: no tile
  // writes ring rows — the FFT stage has written those of this batch itself (long transforms: fft256_kernels.h, RowsExtra), or
  // nobody needs them before the pipeline is drained, which then writes them from the call's PSD plane (8192 points, deep
  // pipelining: k_ring_fill, specscan.hip).
  const int* tile_list;
  int hist_by_fft;
  // What the library did (ss_get_stats): [0] tiles that went through a culling test, [1] tiles the test proved empty,
  // [2] workgroups that stopped waiting for a launch's plan and made it themselves (kStat*). Null: not counted.
  unsigned long long* stats;
#ifdef SS_DIAG
  long long* stamp_mid;  // measurement builds, four per tile: wall clock at the tile's start, after the first pass of phase 1 (thread 0's 36 rows have landed), after phase 1 and after phase 2
  unsigned* cull_stats;  // measurement builds: {tiles, tiles on the culling path, tiles culled}
#endif
};

// plane[row][byte offset coff]: block-uniform row base (scalar registers) + one 32-bit per-thread offset
__device__ __forceinline__ void store_row(float* plane, int row, int n, uint32_t coff, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(plane + (size_t)row * n) + coff) = v;
}

// time means of one column's tile, written into the LDS tile: direct sum for the tile's first frame, then the
// Averager's own update. frames_seen = Averager::m_frames after the tile's first frame (warm-up: -100 until G).
template <int G, int TF, int P, bool WARMUP>
__device__ __forceinline__ void time_means_to_tile(const float (&x)[G - 1 + TF], float* __restrict__ tile_col, int frames_seen) {
  float sum = 0.0f;
#pragma unroll
  for (int g = 0; g < G; ++g) sum += x[g];  // oldest first
#pragma unroll
  for (int j = 0; j < TF; ++j) {
    if (j > 0) {
      sum -= x[j - 1];      // Averager::subtract(front)
      sum += x[j + G - 1];  // Averager::add(new row)
    }
    const float m = div_const<G>(sum);  // m_sum[i] / m_groupSize
    tile_col[j * P] = (!WARMUP || frames_seen + j >= G) ? m : kNoData;  // Averager::updateAverage: m_groupSize <= m_frames
  }
}

// Spectrogram side branch inside the detect kernel (spectrogram.cpp:45-60). Every tile reduces its own frames:
//   spectrogram_tile_means   all 256 threads: wave w takes the tile's frames 4w..4w+3, lane l the bins 4l..4l+3 of the
//                            tile (one 16-byte load per frame; the rows were just read by phase 1 and come from L2).
//                            Per frame the mean of m adjacent raw-PSD bins, summed in ascending bin order as the
//                            reference does (for m > 4 the lanes to the right hand their four values over one by one),
//                            then / m. The means go to LDS (the avgY tile is free by then).
//   spectrogram_tile_sum     thread = output bin: the tile's frames added in frame order -> spec_partial[tile][bin]
// The partial sums of a batch are added to the container (in frame-tile order, whatever ran when) by the NEXT launch —
// spectrogram_fold, by the first-dispatched tile of each bin column — or by k_spec_combine when the host needs the
// container before that (ss_spectrogram_read, retune): kernel boundaries order the writes, no fences or counters.
template <int TF, int TB>
__device__ __forceinline__ void spectrogram_tile_means(const DetectArgs& a, float* __restrict__ means, int tid, int f0, int b0) {
  static_assert(TF == 16 && TB == 256, "four waves x four frames, 64 lanes x four bins");
  const int m = a.spec_m, n = a.n;
  const int per_tile = TB / m > 0 ? TB / m : 1;
  const int lane = tid & 63, w = tid >> 6;
  const int bin = b0 + 4 * lane;
  const bool inside = bin < n;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int fr = min(max(f0 + 4 * w + i, 0), a.nframes - 1);  // always a legal row; frames outside the batch are skipped by the sum
    v[i] = *reinterpret_cast<const float4*>(a.psd + (size_t)fr * n + (inside ? bin : 0));
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float* row = means + (4 * w + i) * per_tile;
    if (m == 1) {
      if (inside) *reinterpret_cast<float4*>(row + 4 * lane) = v[i];
    } else if (m == 2) {
      if (inside) *reinterpret_cast<float2*>(row + 2 * lane) = make_float2((v[i].x + v[i].y) / 2.0f, (v[i].z + v[i].w) / 2.0f);
    } else {
      float sum = ((v[i].x + v[i].y) + v[i].z) + v[i].w;
      const int lanes_per_bin = m >> 2;
      for (int d = 1; d < lanes_per_bin; ++d) {  // wave-uniform trip count
        sum += __shfl_down(v[i].x, d);
        sum += __shfl_down(v[i].y, d);
        sum += __shfl_down(v[i].z, d);
        sum += __shfl_down(v[i].w, d);
      }
      if (inside && (lane & (lanes_per_bin - 1)) == 0) row[lane / lanes_per_bin] = sum / (float)m;
    }
  }
}

template <int TF, int TB>
__device__ __forceinline__ void spectrogram_tile_sum(const DetectArgs& a, const float* __restrict__ means, int tid, int ft, int f0, int b0) {
  const int m = a.spec_m;
  const int per_tile = TB / m > 0 ? TB / m : 1;
  const int ob = b0 / m + tid;
  if (tid >= per_tile || ob >= a.spec_n) return;
  const int j_lo = max(0, -f0), j_hi = min(TF, a.nframes - f0);  // the tile's frames that belong to this batch
  float acc = 0.0f;
#pragma unroll
  for (int j = 0; j < TF; ++j)
    if (j >= j_lo && j < j_hi) acc += means[j * per_tile + tid];
  a.spec_partial[(size_t)ft * a.spec_n + ob] = acc;
}

template <int TB>
__device__ __forceinline__ void spectrogram_fold(const DetectArgs& a, int tid, int b0) {
  const int m = a.spec_m;
  const int per_tile = TB / m > 0 ? TB / m : 1;
  const int ob = b0 / m + tid;
  if (tid >= per_tile || ob >= a.spec_n) return;
  float acc = a.spec_prev_sum[ob];
  for (int t0 = 0; t0 < a.spec_prev_tiles; t0 += 16) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = a.spec_prev_partial[(size_t)min(t0 + i, a.spec_prev_tiles - 1) * a.spec_n + ob];
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (t0 + i < a.spec_prev_tiles) acc += r[i];
  }
  a.spec_prev_sum[ob] = acc;
}

constexpr int kLiveLists = 8;        // the most plan workgroups (lists) a stage may have
constexpr int kLiveCopies = 16;      // copies of the count words, kLiveCopyStride ints apart, behind the lists
constexpr int kLiveCopyStride = 1024;
constexpr int kLiveCounts = 8192;    // where the first copy starts
constexpr int kLiveInts = kLiveCounts + kLiveCopies * kLiveCopyStride;  // ints of a DetectArgs::live buffer
constexpr int kLiveHeader = 16;      // header words in front of DetectArgs::live's lists: the list-written flags
constexpr int kLiveCap = 640;        // capacity of a plan workgroup's list: its tile columns x the batch's frame tiles
constexpr int kLiveReady = 1 << 30;  // header word: the list is complete
constexpr int kPlanLdsFloats = 9600; // staging area of a plan workgroup (+ 64 ints of bookkeeping behind it)

// Frame tiles of a batch, and how many tile columns one plan workgroup can take (0: the stage cannot be planned).
__host__ __device__ inline int plan_frame_tiles(int nframes, int shift) { return (nframes + shift + 15) / 16; }
// a column's maxima in a plan wave's LDS: frame f of the batch (f >= -kPlanBefore: the frames before it whose maxima the halo frames
// left) at word F + F / 16, F = f + kPlanBefore (one pad word per frame tile)
constexpr int kHaloSegPitch = 64;
constexpr int kPlanBefore = 48;  // >= kHistRows = 35, a multiple of 16
__host__ __device__ inline int plan_col_floats(int nframes) { return nframes + kPlanBefore + ((nframes + kPlanBefore) >> 4) + 1; }
__host__ __device__ inline int plan_cols_per_wg(int nframes, int shift) {
  const int per_col = plan_col_floats(nframes), nft = plan_frame_tiles(nframes, shift);
  for (int cols = 8; cols >= 4; cols >>= 1)
    if (cols * per_col <= kPlanLdsFloats && cols * nft <= kLiveCap && nft <= 192) return cols;
  return 0;
}

// The plan role (8192-point frames: 32 tile columns). Workgroup `seg` takes `cols` tile columns, wave w the column
// seg * cols + w, lane = frame tile (three rounds at most). Tiles on detect_tile's straight-line path whose sole products are
// mask bits and counts are tested — no learning / warm-up / ring rows, no rel or avg plane wanted, not one of the tiles
// that refresh the ring; all others are evaluated. The wave first copies its column of Fft8192Args::segsum (one float per
// frame) into LDS with coalesced loads — a lane reading its own 36 frames straight from memory touches 36 cache lines
// nobody shares with it: measured, a plan workgroup took 24 us that way and every consumer waited for it — then every lane
// takes the maximum over its 36 frames from there (one pad word per 16 frames: lanes are 16 frames apart). Evaluated
// tiles go to the workgroup's list in (column, frame tile) order, then the header word. A culled tile writes NOTHING: its
// mask words are zero already — the emit stage of the buffer's previous user cleared every word it found set
// (EmitArgs::clear_masks) — where clearing them here, a megabyte of 32-byte writes per call, cost 1.4 us per step.
// `lds` = kPlanLdsFloats floats + 64 ints.
// `first` = false: the call is a consumer's fallback (the plan role's own workgroup has not been heard of): everything is
// written as the plan role writes it, nothing is counted twice.
template <int G, int GX, int TF, int TB_ = 256>
__device__ __forceinline__ void plan_tiles(const DetectArgs& a, int seg, int cols, int tid, float* __restrict__ lds, bool first) {
  using T = DetectTile<G, GX, TF, TB_>;
  constexpr int TB = T::TB, H = T::H;
  static_assert(TF == 16, "one pad word per frame tile");
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = a.n, nframes = a.nframes;
  const int tiles_per_row = n / TB;
  const int nft = plan_frame_tiles(nframes, a.shift);
  const int rounds = (nft + 63) >> 6;
  int* book = reinterpret_cast<int*>(lds + kPlanLdsFloats);  // [w * 4 + round]: tiles wave w lists in that round
  const int col = seg * cols + w;
  const bool has_col = w < cols && col < tiles_per_row;
  const bool plannable = a.segsum && !a.rel_out && !a.avg_out;
  float* mine = lds + w * plan_col_floats(nframes);
  const auto at = [](int f) { return f + kPlanBefore + ((f + kPlanBefore) >> 4); };  // frame f's word in `mine`
  // frames before the batch whose maxima are known (the halo frames' — until session 36 of round 5 they left none, and the 64 tiles of
  // a batch's first two frame tiles were evaluated whatever they held: the last workgroups of every launch, profiles/r05/s32_*)
  const int before = a.halo_segsum && a.halo_psd ? min(a.halo_rows, kPlanBefore) : 0;
  float tm = 0.0f;
  if (has_col && plannable) {
    const float* src = a.segsum + (size_t)col * a.seg_pitch;
    tm = a.thr_tilemin[col];
    // (eight loads in flight per trip: one after the other — load, wait, store, sixteen times for a 1024-frame batch — the copy alone
    // took the plan workgroups 8-13 us, and the launch's frame workgroups are through after 12: they wait for this list)
    // (round 6 tried sixteen at a time — ONE round trip for a 1024-frame batch: 23.2 against 22.8-23.1 us per step in alternating runs,
    // profiles/r06/s4_summary.txt: no gain, the frame workgroups do not wait for the plan as a rule; SS_PLAN_COPY_LOADS=16 keeps the form)
    float hm = 0.0f;
    if (lane < before) hm = a.halo_segsum[col * kHaloSegPitch + (a.halo_rows - before) + lane];
#ifndef SS_PLAN_COPY_LOADS  // (A/B builds, scripts/build_ab.py: 16 = one trip)
#define SS_PLAN_COPY_LOADS 8
#endif
    for (int fb = 0; fb < nframes; fb += 64 * SS_PLAN_COPY_LOADS) {
      float v[SS_PLAN_COPY_LOADS];
#pragma unroll
      for (int u = 0; u < SS_PLAN_COPY_LOADS; ++u) v[u] = src[min(fb + 64 * u + lane, nframes - 1)];
#pragma unroll
      for (int u = 0; u < SS_PLAN_COPY_LOADS; ++u) {
        const int f = fb + 64 * u + lane;
        if (f < nframes) mine[at(f)] = v[u];
      }
    }
    if (lane < before) mine[at(lane - before)] = hm;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's column is in LDS (nobody else reads it)
  unsigned long long live_mask[3] = {0ull, 0ull, 0ull}, dead_mask[3] = {0ull, 0ull, 0ull};
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r < rounds) {
      const int ft_seq = r * 64 + lane;
      const bool exists = has_col && ft_seq < nft;
      const int ft = (ft_seq + nft - 1) % nft;
      const int f0 = ft * TF - a.shift;
      // (its 36 frames' maxima are all there — the batch's, and before it the halo frames' —, none of them a learning frame, no ragged end)
      const bool steady = (a.n_learn == 0 ? f0 - (G - 1) >= -before : f0 - (G - 1) >= a.n_learn) && (f0 + TF <= nframes);
      const bool writes_hist = !a.hist_by_fft && f0 + TF > nframes - H;
      bool culled = false;
      if (exists && plannable && steady && !writes_hist) {
        // the 21 x 21 mean that ends at frame f is at most mean(M_{f-20} .. M_f) - tm, M_g = the column's maximum in frame g
        // (round 3; until then max(M) - tm over all 36 frames, which noise alone brings within 1.5 dB of an 8 dB threshold).
        // The sum slides over the tile's 16 frames; a sum that is not a number (a frame of NaNs, -inf leaving the window)
        // decides nothing: evaluated.
        const int b = f0 - (G - 1);
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < G; ++k) sum += mine[at(b + k)];
        float best = sum;
        bool unsure = sum != sum;
#pragma unroll
        for (int j = 1; j < TF; ++j) {
          sum -= mine[at(b + j - 1)];
          sum += mine[at(b + j + G - 1)];
          unsure = unsure || (sum != sum);
          best = fmaxf(best, sum);
        }
        culled = !unsure && (best * (1.0f / (float)G) - tm) < a.start_level - kCullMargin;  // (false for NaN)
      }
      live_mask[r] = __ballot(exists && !culled);
      dead_mask[r] = __ballot(culled);
      if (a.stats && first) {  // (wave-uniform)
        const int tested = __popcll(__ballot(exists && plannable && steady && !writes_hist));
        if (lane == 0 && tested) {
          atomicAdd(stat_word(a.stats, kStatTested), (unsigned long long)tested);
          atomicAdd(stat_word(a.stats, kStatCulled), (unsigned long long)__popcll(dead_mask[r]));
        }
      }
      if (lane == 0) book[w * 4 + r] = __popcll(live_mask[r]);
    } else if (lane == 0) {
      book[w * 4 + r] = 0;
    }
  }
  if (lane == 0) book[w * 4 + 3] = 0;
  __syncthreads();
  int base = 0, total = 0;
  for (int k = 0; k < 32; ++k) {  // (wave, round) in order
    const int c = book[k];
    base += k < w * 4 ? c : 0;
    total += c;
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    if (r < rounds) {
      const int ft_seq = r * 64 + lane;
      if ((live_mask[r] >> lane) & 1ull)
        __hip_atomic_store(&a.live[kLiveHeader + seg * kLiveCap + base + __popcll(live_mask[r] & ((1ull << lane) - 1ull))], ft_seq * tiles_per_row + col,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // write-through
      base += __popcll(live_mask[r]);
    }
  }
  // the count at once, in kLiveCopies copies a page apart: a thousand consumers asking one address are a thousand requests to
  // one memory channel (measured: 2 us per step) ...
  if (tid < kLiveCopies) __hip_atomic_store(&a.live[kLiveCounts + tid * kLiveCopyStride + seg], kLiveReady | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ... this wave's entries are in memory ...
  __syncthreads();
  if (tid == 0) __hip_atomic_store(&a.live[seg], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ... every wave's: the list may be read
#ifdef SS_DIAG
  if (a.cull_stats > reinterpret_cast<unsigned*>(1) && lane == 0 && has_col) {
    int lv = 0, dd = 0;
    for (int r = 0; r < 3; ++r) {
      lv += __popcll(live_mask[r]);
      dd += __popcll(dead_mask[r]);
    }
    atomicAdd(&a.cull_stats[0], (unsigned)(lv + dd));
    atomicAdd(&a.cull_stats[2], (unsigned)dd);
  }
#endif
}

// Consumer side (scan_step.h), called by every wave of a workgroup with the same arguments. Consumer `p` of `nseg` lists serves
// list p mod nseg, entries 2 (p div nseg) and the next one. All it has to read first is ONE header word, kLiveReady | count:
//   live_wait_count  `word` = the header word as read earlier (the FFT role asks before its pass 3, so the answer usually costs it
//                    nothing); if the list's count was not known by then, poll for it — at most `limit` times. Returns the word
//                    as last seen (kLiveReady missing: gave up);
//   live_wait_list   polls the list-written flag the same way (only consumers that have entries to fetch ask);
//   live_pair        the two tiles (-1: none), once count and entries are known to be there.
// A consumer that gives up makes the plan itself (see the top of this file): nothing waits without bound.
__host__ __device__ __forceinline__ int live_count_word(int p, int nseg) { return kLiveCounts + ((p / nseg) % kLiveCopies) * kLiveCopyStride + p % nseg; }
__device__ __forceinline__ int live_wait_count(const DetectArgs& a, int p, int nseg, int word, int limit) {
  word = __builtin_amdgcn_readfirstlane(word);
  for (int i = 0; i < limit && !(word & kLiveReady); ++i) {
    __builtin_amdgcn_s_sleep(8);
    word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.live[live_count_word(p, nseg)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  return word;
}
__device__ __forceinline__ bool live_wait_list(const DetectArgs& a, int seg, int limit) {
  int flag = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.live[seg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  for (int i = 0; i < limit && !flag; ++i) {
    __builtin_amdgcn_s_sleep(8);
    flag = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.live[seg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  return flag != 0;
}
__device__ __forceinline__ int2 live_pair(const DetectArgs& a, int p, int nseg, int count) {
  const int seg = p % nseg, q = p / nseg;
  int2 t = make_int2(-1, -1);
  if (2 * q < count) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(a.live + kLiveHeader + seg * kLiveCap + 2 * q), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t.x = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    if (2 * q + 1 < count) t.y = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    const int n_tiles = (a.n / 256) * plan_frame_tiles(a.nframes, a.shift);
    if ((unsigned)t.x >= (unsigned)n_tiles) t.x = t.y = -1;  // (never: a tile number that is none must not become an address)
    if ((unsigned)t.y >= (unsigned)n_tiles) t.y = -1;
  }
  return t;
}

// min of the noise ceiling over bins [256 c - 32, 256 c + 288), c = blockIdx.x (one wave each): DetectArgs::thr_tilemin,
// refreshed whenever the ceiling changes (learning batches).
__global__ __launch_bounds__(64) void k_thr_tilemin(const float* __restrict__ thr, int n, float* __restrict__ tilemin) {
  const int c = blockIdx.x, lane = threadIdx.x;
  float m = __builtin_inff();
  for (int i = lane; i < 320; i += 64) {
    const int bin = 256 * c - 32 + i;
    if (bin >= 0 && bin < n) m = fminf(m, thr[bin]);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) m = fminf(m, __shfl_xor(m, d));
  if (lane == 0) tilemin[c] = m;
}

// Tile culling for long transforms (N = 256 x N2: 65536 points and up, 256 and more tile columns). Same idea as plan_tiles — a
// tile that provably holds no candidate is not evaluated — from what the rows kernel of the FFT stage left per frame and 32-bin
// run (RowsExtra::smax, a ring over the frames since the last reset, so that the tile's rows from BEFORE the batch are covered
// too: a 16-frame call of 2^20-point rows has no others). The bound: with M_g the largest dB value of frame g over the tile's 256
// bins and one run on either side, and m the minimum of the noise ceiling over the same bins (thr_tilemin), every rel value of
// frame g in reach of the tile is <= M_g - m, so the 21 x 21 mean that ends at frame f is <= mean(M_{f-20} .. M_f) - m: the tile
// is dropped when that stays below start_level - kCullMargin for each of its frames (the margin covers the fp32 sums on either
// side: < 0.01 dB). A frame of zeros (M = -inf) makes the bound -inf for every window that holds it — those produce no candidate
// either — and a NaN anywhere makes the tile "cannot tell": evaluated. A tile is only tested when all its rows are frames since
// `clean_rel` (batch-relative, <= 0): no learning frame among them, one noise ceiling for all of them, the averager past its
// warm-up. The ring rows such a tile would have written are written by the rows kernel itself (RowsExtra::hist_out), its mask
// words are zero already (EmitArgs::clear_masks). The launch is a stage of its own between the call's rows kernel and the launch
// that carries its detect stage: kernel boundaries order everything, the lists need no hand-over protocol.
// A workgroup takes C tile columns with all their frame tiles: every thread first reduces (column, frame) pairs to M
// (ten ring values each) into LDS, then thread (column, frame tile) slides the 21-frame sum over its 16 frames; the tiles to be
// evaluated — numbers in detect_tile's numbering — are appended to ONE list (list[0] = count, zeroed before the launch; one
// atomic per workgroup), so that the detect role can spread whatever there is evenly over a fixed number of workgroups: a
// workgroup per tile pair, leaving at once when the pair is culled, cost the launch 11 us in DISPATCH alone (1024 workgroups
// of eight waves take the hardware that long to start). The order of the list depends on the scheduling, the
// results do not: tiles touch disjoint mask words, and the per-frame counts are integer sums.
struct PlanLongArgs {
  const float* smax;
  int smax_mask;
  int abs0;       // frames since the last reset before this batch
  int clean_rel;  // first batch-relative frame (<= 0 in a call without learning frames) from which every row qualifies
  int cols;       // C: tile columns per workgroup
  int logn;
  int* list;
  // 0: the ring holds dB values where k_fft_rows256_psd puts them (rows_smax_index); 1: 2^20 points in two passes
  // (fft1024_kernels.h): max_key values at rows1024_smax_index(run) — [k1 group][k2], gathered by atomic maxima;
  // 2: 65536 points by the radix-8 fold (fft65536_dif8.h): value g of tile column c is the largest dB value among the column's bins
  // of residue g, at [g][c] — a column's neighbours then count with all eight of their values, not with their nearest run;
  // 3: 262144 points as 256 columns x 1024-point rows (round 6, fft1024_kernels.h: fft_rows1024_tile<8>): max_key values of the eight
  // 32-bin runs of tile column c at [run][c] (rows1024x256_smax_index), gathered by atomic maxima like layout 1's — planned by
  // plan_x256_run below (a block per 32 columns x one frame tile), not by plan_long_run
  int layout;
};
constexpr int kPlanLongFloats = 8192;  // LDS of a plan workgroup: C x (16 nft + 20) values of M
// max_c: 8, or 32 for the two-pass layout of 2^20 points — there a workgroup's columns are consecutive floats of a frame's row of
// the ring, and 32 of them are a whole 128-byte line (with 8 the four workgroups that share a line each fetched all of it:
// 5.9 against 3.0 MB per launch, 12 against 7 us, profiles/r04/s4_summary.txt)
__host__ __device__ inline int plan_long_cols(int nframes, int shift, int tile_cols, int max_c = 8, int max_floats = 8192) {  // 0: the stage cannot be planned
  const int nft = (nframes + shift + 15) / 16, rows = 16 * nft + 20;
  int c = tile_cols / 256 > 1 ? tile_cols / 256 : 1;  // at least 256 workgroups where there are that many columns
  if (max_c > 8) c = max_c;
  if (c > max_c) c = max_c;
  if (c > max_floats / rows) c = max_floats / rows;
  if (nft > 0 && c > 256 / nft) c = 256 / nft;
  return c;
}

// which tile columns a plan workgroup takes: layout 0 — (wc, d0): the columns whose unshifted number is wc + nsub d, d = d0 .. d0 + C;
// layout 1 — the columns 4 k2 + wc for k2 = d0 .. d0 + C, the 32 / C workgroups whose columns share the 128-byte lines of a frame's row
// of the ring in consecutive slots of ONE XCD (block b runs on XCD b mod 8)
__host__ __device__ inline void plan_long_block(int layout, int block, int C, int lognsub, int* wc, int* d0) {
  if (layout == 2) {  // C consecutive columns (consecutive floats of each of the eight residue rows)
    *wc = 0;
    *d0 = block * C;
  } else if (layout) {
    const int M = (C <= 32 && (32 % C) == 0) ? 32 / C : 1;
    const int xcd = block & 7, slot = block >> 3;
    *wc = xcd & 3;
    *d0 = ((((slot / M) << 1) | (xcd >> 2)) * M + slot % M) * C;
  } else {
    *wc = block & ((1 << lognsub) - 1);
    *d0 = (block >> lognsub) * C;
  }
}
__host__ __device__ inline int plan_long_blocks(int layout, int C, int n) {  // workgroups of the plan launch
  if (!layout || layout == 2) return (n >> 16) * ((256 + C - 1) / C);
  const int line_groups = (C <= 32 && 32 % C == 0) ? 32 / C : 1;
  return 4 * ((((1024 + C - 1) / C + 2 * line_groups - 1) / (2 * line_groups)) * (2 * line_groups));
}

// What the plan needs of the call's DetectArgs (the whole struct is 200 bytes of kernel arguments the column launch of a
// 2^20-point frame — whose first workgroups may run the plan of the call before, k_fft_cols1024_plan — has no use for).
struct PlanLongDet {
  int n, nframes, shift, n_learn;
  int planes_out;  // a rel or avg plane is wanted: nothing is tested, every tile is listed
  float start_level;
  const float* thr_tilemin;
  unsigned long long* stats;
};
__host__ __device__ inline PlanLongDet plan_long_det(const DetectArgs& d) {
  PlanLongDet a{};
  a.n = d.n;
  a.nframes = d.nframes;
  a.shift = d.shift;
  a.n_learn = d.n_learn;
  a.planes_out = (d.rel_out || d.avg_out) ? 1 : 0;
  a.start_level = d.start_level;
  a.thr_tilemin = d.thr_tilemin;
  a.stats = d.stats;
  return a;
}
constexpr int kPlanLongInts = 16;  // bookkeeping words of a plan block behind its values of M

// One plan block: 256 threads (`tid`), `block` = its number (what blockIdx.x is for k_plan_long), `mrow` = room for C x (16 nft + 20)
// floats, `book` = kPlanLongInts ints. Every __syncthreads() below is reached by all threads of the workgroup whatever the block
// finds: several blocks may share a workgroup (k_fft_cols1024_plan: four of them in 1024 threads).
template <int G, int GX, int TF, int TB_ = 256>
__device__ __forceinline__ void plan_long_run(const PlanLongDet& a, const PlanLongArgs& p, int block_no, int tid, float* __restrict__ mrow, int* __restrict__ book) {
  using T = DetectTile<G, GX, TF, TB_>;
  constexpr int TB = T::TB;
  static_assert(TB == 256 && T::A <= 32 && TF == 16 && G == 21, "eight 32-bin runs per tile column, one more on either side");
  int* wave_cnt = book;             // [4]
  int* stat_cnt = book + 4;         // [4][2]
  int* list_base_p = book + 12;
  const int n = a.n, nframes = a.nframes;
  const int tiles_per_row = n / TB, groups = n >> 5;
  const int nft = (nframes + a.shift + TF - 1) / TF;
  const int rows = TF * nft + (G - 1);  // frames [-shift - 20, 16 nft - shift) of the batch's frame numbering
  // this workgroup's C tile columns: the ones whose run maxima lie side by side in a frame's row of smax (rows_smax_index: the
  // same c, consecutive d) — any C columns would do, these are read with 32-byte requests instead of 4-byte ones
  const int C = p.cols, lognsub = p.logn - 16;
  // layout 0: workgroup (wc, d0) takes the columns whose unshifted number is wc + nsub d; layout 1: the columns 4 k2 + wc of C
  // consecutive k2 — either way the ones whose maxima lie side by side in a frame's row of the ring
  // (layout 1: the 32 / C workgroups whose columns share the 128-byte lines of the ring are consecutive slots of ONE XCD — block b
  // runs on XCD b mod 8 — as they are in layout 0 by construction; spread over two XCDs they fetched every line twice: 12 against
  // 7 us per call, profiles/r04/s4_summary.txt)
  int wc, d0;
  plan_long_block(p.layout, block_no, C, lognsub, &wc, &d0);
  const auto column = [&](int i) {
    if (p.layout == 2) return d0 + i < 256 ? d0 + i : tiles_per_row;
    if (p.layout) return d0 + i < 1024 ? 4 * (d0 + i) + wc : tiles_per_row;
    return d0 + i < 256 ? (wc + ((d0 + i) << lognsub)) ^ (tiles_per_row >> 1) : tiles_per_row;
  };
  for (int e = tid; e < C * rows; e += 256) {
    const int i = e % C, r = e / C, col = column(i);
    float m = -__builtin_inff();
    if (col < tiles_per_row) {
      const int frame = r - a.shift - (G - 1);
      const float* row = p.smax + ((size_t)((p.abs0 + frame) & p.smax_mask) * groups);
      // the column's eight runs, the last run of the column below and the first of the one above (the band's edges: its own once more)
      float v[10];
      if (p.layout == 2) {
        float lo = -__builtin_inff(), hi = -__builtin_inff();  // (this layout's values never hold a NaN: fft8192_v2.h takes the maxima with fmaxf)
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          v[g] = row[256 * g + col];
          lo = fmaxf(lo, row[256 * g + (col > 0 ? col - 1 : col)]);
          hi = fmaxf(hi, row[256 * g + (col + 1 < tiles_per_row ? col + 1 : col)]);
        }
        v[8] = lo;
        v[9] = hi;
      } else if (p.layout) {
        const unsigned* krow = reinterpret_cast<const unsigned*>(row);
#pragma unroll
        for (int g = 0; g < 8; ++g) v[g] = max_key_value(krow[rows1024_smax_index(8 * col + g)]);
        v[8] = max_key_value(krow[rows1024_smax_index(col > 0 ? 8 * col - 1 : 8 * col)]);
        v[9] = max_key_value(krow[rows1024_smax_index(col + 1 < tiles_per_row ? 8 * col + 8 : 8 * col + 7)]);
      } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) v[g] = row[rows_smax_index(col, g, p.logn)];
      v[8] = row[col > 0 ? rows_smax_index(col - 1, 7, p.logn) : rows_smax_index(col, 0, p.logn)];
      v[9] = row[col + 1 < tiles_per_row ? rows_smax_index(col + 1, 0, p.logn) : rows_smax_index(col, 7, p.logn)];
      }
      bool bad = false;
#pragma unroll
      for (int g = 0; g < 10; ++g) {
        bad = bad || (v[g] != v[g]);
        m = fmaxf(m, v[g]);
      }
      if (bad) m = __builtin_nanf("");
    }
    mrow[i * rows + r] = m;
  }
  __syncthreads();
  bool live = false, tested = false;
  int block = 0;
  if (tid < C * nft && column(tid / nft) < tiles_per_row) {
    const int cl = tid / nft, ft_seq = tid % nft;
    const int ft = (ft_seq + nft - 1) % nft;
    const int f0 = ft * TF - a.shift;
    block = ft_seq * tiles_per_row + column(cl);
    live = true;
    if (a.n_learn == 0 && f0 - (G - 1) >= p.clean_rel && !a.planes_out) {
      tested = true;
      const float* mr = mrow + cl * rows + ft * TF;  // mr[k] = M of frame f0 - 20 + k
      float best = -__builtin_inff();
      bool unsure = false;
#pragma unroll
      for (int j = 0; j < TF; ++j) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < G; ++k) sum += mr[j + k];
        if (f0 + j >= 0 && f0 + j < nframes) {  // the frames of this batch the tile answers for
          unsure = unsure || (sum != sum);
          best = fmaxf(best, sum);
        }
      }
      const float bound = best * (1.0f / (float)G) - a.thr_tilemin[column(cl)];
      live = unsure || !(bound < a.start_level - kCullMargin);  // (a NaN difference: live)
    }
  }
  // this workgroup's list, in thread order
  const int lane = tid & 63, w = tid >> 6;
  const unsigned long long mask = __ballot(live);
  if (lane == 0) wave_cnt[w] = __popcll(mask);
  if (a.stats) {  // ss_get_stats: one pair of additions per workgroup (below), to the copy of its block index
    const int n_tested = __popcll(__ballot(tested)), n_culled = __popcll(__ballot(tested && !live));
    if (lane == 0) {
      stat_cnt[2 * w] = n_tested;
      stat_cnt[2 * w + 1] = n_culled;
    }
  }
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    base += k < w ? wave_cnt[k] : 0;
    total += wave_cnt[k];
  }
  if (tid == 0) *list_base_p = total ? atomicAdd(&p.list[0], total) : 0;
  if (tid == 1 && a.stats) {
    const int nt = stat_cnt[0] + stat_cnt[2] + stat_cnt[4] + stat_cnt[6], nc = stat_cnt[1] + stat_cnt[3] + stat_cnt[5] + stat_cnt[7];
    if (nt) {
      atomicAdd(stat_word(a.stats, kStatTested), (unsigned long long)nt);
      atomicAdd(stat_word(a.stats, kStatCulled), (unsigned long long)nc);
    }
  }
  __syncthreads();
  if (live) p.list[1 + *list_base_p + base + __popcll(mask & ((1ull << lane) - 1ull))] = block;
}

// The plan for the radix-8 fold's rows (layout 2): one block = 32 consecutive tile columns x ONE frame tile, 256 threads. What
// plan_long_run does for this layout — a block per tile column with all its frame tiles — fetched every 128-byte line of the run
// maxima thirty-two times over (a column's eight values per frame sit in eight different lines, each shared with 31 other columns:
// 143 MB of L1 fills per 512-frame call) and left all but a few dozen of its threads idle through 336 serial LDS reads; as the first
// 128 workgroups of the fold's launch it cost a 128-frame call 8 us and a 512-frame call 12-16 (profiles/r05/s8_summary.txt). Here:
//   1  R[colx][row] = max over the eight residues of the run maxima of column colx (34 of them: the block's 32 and one either side)
//      and frame row (36: the tile's 16 frames and the 20 before) — lanes along the columns, so a wave's load is one line;
//   2  M = max(R[col - 1], R[col], R[col + 1]) formed on the fly, S[col][j] = the 21-frame window sum that ends at frame j of the tile
//      (thread per (col, j): 21 x 3 LDS reads);
//   3  thread per tile: the largest of its window sums -> the same test, the same list, the same statistics as plan_long_run's.
// Block b: column group b mod 8 (32 columns: the eight groups of a frame tile are consecutive blocks, one per XCD), frame tile b / 8 in
// dispatch order (plan_long_run's ft_seq). `lds`: kPlanDif8Floats floats + kPlanLongInts ints behind them.
// Radix Q = 1 << logq (logq = p.logn - 13: 3 or 4): 32 Q tile columns, Q column groups per frame tile; a run of a residue's row (32 of
// its bins) spans Q / 8 tile columns.
constexpr int kPlanDif8Floats = 34 * 37 + 32 * 16;
__host__ __device__ inline int plan_dif8_blocks(int nframes, int shift, int logq = 3) { return ((nframes + shift + 15) / 16) << logq; }
template <int G, int GX, int TF, int TB_ = 256>
__device__ __forceinline__ void plan_dif8_run(const PlanLongDet& a, const PlanLongArgs& p, int block_no, int tid, float* __restrict__ lds, int* __restrict__ book) {
  static_assert(TB_ == 256 && TF == 16 && G == 21, "the fold's rows: tile columns of 256 bins, tiles of 16 frames");
  constexpr int ROWS = TF + G - 1, RP = ROWS + 1;  // 36 frame rows per tile, LDS pitch 37
  float* R = lds;              // [34][RP]
  float* S = lds + 34 * RP;    // [32][TF]
  int* wave_cnt = book;        // [4]
  int* stat_cnt = book + 4;    // [4][2]
  int* list_base_p = book + 12;
  const int logq = p.logn - 13, nres = 1 << logq;
  const int nframes = a.nframes, tiles_per_row = 32 << logq;
  const int nft = (nframes + a.shift + TF - 1) / TF;
  const int cg = block_no & (nres - 1), ft_seq = block_no >> logq;
  const bool in_range = ft_seq < nft;  // (a workgroup's second block may lie past the end: it keeps the barriers company)
  const int ft = in_range ? (ft_seq + nft - 1) % nft : 0;
  const int f0 = ft * TF - a.shift;  // batch-relative frame of the tile's first row of outputs
  const int c0 = 32 * cg;
  if (in_range) {
    // (a thread's five entries, eight residues each: forty loads in flight at once. One entry after the other, a residue after the
    // other, this was forty round trips to L2 beside workgroups that keep the memory system busy — a plan workgroup held its slot for
    // 14 us of a 128-frame launch and the 32 fold workgroups that had to wait for those slots ended it 7 us late, profiles/r05/s24_*)
    constexpr int NE = (34 * ROWS + 255) / 256;
    const float* src[NE];
    float m[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = min(tid + 256 * i, 34 * ROWS - 1);
      const int colx = e % 34, r = e / 34;
      const int col = min(max(c0 - 1 + colx, 0), tiles_per_row - 1);  // (the band's edges: the edge column once more)
      src[i] = p.smax + ((size_t)((p.abs0 + f0 - (G - 1) + r) & p.smax_mask) << (8 + logq)) + (col >> (logq - 3));  // (radix 16: a run of a residue's row covers two tile columns)
      m[i] = -__builtin_inff();
    }
    for (int g0 = 0; g0 < nres; g0 += 8) {
      float v[NE][8];
#pragma unroll
      for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int g = 0; g < 8; ++g) v[i][g] = src[i][256 * (g0 + g)];
#pragma unroll
      for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int g = 0; g < 8; ++g) m[i] = fmaxf(m[i], v[i][g]);  // (the fold's maxima hold no NaN: fft8192_v2.h takes them with fmaxf)
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      if (e < 34 * ROWS) R[(e % 34) * RP + e / 34] = m[i];
    }
  }
  __syncthreads();
  if (in_range) {
    for (int e = tid; e < 32 * TF; e += 256) {
      const int cl = e >> 4, j = e & 15;
      const float *r0 = R + cl * RP + j, *r1 = r0 + RP, *r2 = r1 + RP;
      float sum = 0.0f;
#pragma unroll
      for (int k = 0; k < G; ++k) sum += fmaxf(fmaxf(r0[k], r1[k]), r2[k]);  // M of frame f0 - 20 + j + k
      S[e] = sum;
    }
  }
  __syncthreads();
  bool live = false, tested = false;
  int block = 0;
  if (in_range && tid < 32) {
    const int col = c0 + tid;
    block = ft_seq * tiles_per_row + col;
    live = true;
    if (a.n_learn == 0 && f0 - (G - 1) >= p.clean_rel && !a.planes_out) {
      tested = true;
      float best = -__builtin_inff();
      bool unsure = false;
#pragma unroll
      for (int j = 0; j < TF; ++j) {
        const float sum = S[tid * TF + j];
        if (f0 + j >= 0 && f0 + j < nframes) {  // the frames of this batch the tile answers for
          unsure = unsure || (sum != sum);
          best = fmaxf(best, sum);
        }
      }
      const float bound = best * (1.0f / (float)G) - a.thr_tilemin[col];
      live = unsure || !(bound < a.start_level - kCullMargin);  // (a NaN difference: live)
    }
  }
  // this block's share of the list, in thread order (as plan_long_run)
  const int lane = tid & 63, w = tid >> 6;
  const unsigned long long mask = __ballot(live);
  if (lane == 0) wave_cnt[w] = __popcll(mask);
  if (a.stats) {
    const int n_tested = __popcll(__ballot(tested)), n_culled = __popcll(__ballot(tested && !live));
    if (lane == 0) {
      stat_cnt[2 * w] = n_tested;
      stat_cnt[2 * w + 1] = n_culled;
    }
  }
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    base += k < w ? wave_cnt[k] : 0;
    total += wave_cnt[k];
  }
  if (tid == 0) *list_base_p = total ? atomicAdd(&p.list[0], total) : 0;
  if (tid == 1 && a.stats) {
    const int nt = stat_cnt[0] + stat_cnt[2] + stat_cnt[4] + stat_cnt[6], nc = stat_cnt[1] + stat_cnt[3] + stat_cnt[5] + stat_cnt[7];
    if (nt) {
      atomicAdd(stat_word(a.stats, kStatTested), (unsigned long long)nt);
      atomicAdd(stat_word(a.stats, kStatCulled), (unsigned long long)nc);
    }
  }
  __syncthreads();
  if (live) p.list[1 + *list_base_p + base + __popcll(mask & ((1ull << lane) - 1ull))] = block;
}

// The plan for 262144-point frames (layout 3): the fold's decomposition — one block = 32 consecutive tile columns x ONE frame tile, 256
// threads, a thread's forty loads in one flight — over max_key values at [run g < 8][column < 1024] per frame. Block b: column group
// b mod 32, frame tile b / 32 in dispatch order. A key of "NaN" (a NaN bin in the run: cannot be bounded) counts as +inf: the tile is
// evaluated. `lds`: kPlanDif8Floats floats + kPlanLongInts ints behind them.
__host__ __device__ inline int plan_x256_blocks(int nframes, int shift) { return ((nframes + shift + 15) / 16) << 5; }
template <int G, int GX, int TF, int TB_ = 256>
__device__ __forceinline__ void plan_x256_run(const PlanLongDet& a, const PlanLongArgs& p, int block_no, int tid, float* __restrict__ lds, int* __restrict__ book) {
  static_assert(TB_ == 256 && TF == 16 && G == 21, "tile columns of 256 bins, tiles of 16 frames");
  constexpr int ROWS = TF + G - 1, RP = ROWS + 1;  // 36 frame rows per tile, LDS pitch 37
  float* R = lds;              // [34][RP]
  float* S = lds + 34 * RP;    // [32][TF]
  int* wave_cnt = book;        // [4]
  int* stat_cnt = book + 4;    // [4][2]
  int* list_base_p = book + 12;
  constexpr int tiles_per_row = 1024;
  const int nframes = a.nframes;
  const int nft = (nframes + a.shift + TF - 1) / TF;
  const int cg = block_no & 31, ft_seq = block_no >> 5;
  const bool in_range = ft_seq < nft;  // (a workgroup's second block may lie past the end: it keeps the barriers company)
  const int ft = in_range ? (ft_seq + nft - 1) % nft : 0;
  const int f0 = ft * TF - a.shift;  // batch-relative frame of the tile's first row of outputs
  const int c0 = 32 * cg;
  if (in_range) {
    constexpr int NE = (34 * ROWS + 255) / 256;
    const unsigned* src[NE];
    unsigned m[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = min(tid + 256 * i, 34 * ROWS - 1);
      const int colx = e % 34, r = e / 34;
      const int col = min(max(c0 - 1 + colx, 0), tiles_per_row - 1);  // (the band's edges: the edge column once more)
      src[i] = reinterpret_cast<const unsigned*>(p.smax) + ((size_t)((p.abs0 + f0 - (G - 1) + r) & p.smax_mask) << 13) + col;
    }
    unsigned v[NE][8];
#pragma unroll
    for (int i = 0; i < NE; ++i)
#pragma unroll
      for (int g = 0; g < 8; ++g) v[i][g] = src[i][1024 * g];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      m[i] = 0u;
#pragma unroll
      for (int g = 0; g < 8; ++g) m[i] = max(m[i], v[i][g]);  // (order-preserving keys: the NaN key is the largest)
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = tid + 256 * i;
      if (e < 34 * ROWS) R[(e % 34) * RP + e / 34] = m[i] == 0xffffffffu ? __builtin_inff() : max_key_value(m[i]);
    }
  }
  __syncthreads();
  if (in_range) {
    for (int e = tid; e < 32 * TF; e += 256) {
      const int cl = e >> 4, j = e & 15;
      const float *r0 = R + cl * RP + j, *r1 = r0 + RP, *r2 = r1 + RP;
      float sum = 0.0f;
#pragma unroll
      for (int k = 0; k < G; ++k) sum += fmaxf(fmaxf(r0[k], r1[k]), r2[k]);  // M of frame f0 - 20 + j + k (no NaN among them: see above)
      S[e] = sum;
    }
  }
  __syncthreads();
  bool live = false, tested = false;
  int block = 0;
  if (in_range && tid < 32) {
    const int col = c0 + tid;
    block = ft_seq * tiles_per_row + col;
    live = true;
    if (a.n_learn == 0 && f0 - (G - 1) >= p.clean_rel && !a.planes_out) {
      tested = true;
      float best = -__builtin_inff();
      bool unsure = false;
#pragma unroll
      for (int j = 0; j < TF; ++j) {
        const float sum = S[tid * TF + j];
        if (f0 + j >= 0 && f0 + j < nframes) {  // the frames of this batch the tile answers for
          unsure = unsure || (sum != sum);  // (+inf and -inf in one window)
          best = fmaxf(best, sum);
        }
      }
      const float bound = best * (1.0f / (float)G) - a.thr_tilemin[col];
      live = unsure || !(bound < a.start_level - kCullMargin);  // (a NaN difference: live)
    }
  }
  // this block's share of the list, in thread order (as plan_long_run)
  const int lane = tid & 63, w = tid >> 6;
  const unsigned long long mask = __ballot(live);
  if (lane == 0) wave_cnt[w] = __popcll(mask);
  if (a.stats) {
    const int n_tested = __popcll(__ballot(tested)), n_culled = __popcll(__ballot(tested && !live));
    if (lane == 0) {
      stat_cnt[2 * w] = n_tested;
      stat_cnt[2 * w + 1] = n_culled;
    }
  }
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    base += k < w ? wave_cnt[k] : 0;
    total += wave_cnt[k];
  }
  if (tid == 0) *list_base_p = total ? atomicAdd(&p.list[0], total) : 0;
  if (tid == 1 && a.stats) {
    const int nt = stat_cnt[0] + stat_cnt[2] + stat_cnt[4] + stat_cnt[6], nc = stat_cnt[1] + stat_cnt[3] + stat_cnt[5] + stat_cnt[7];
    if (nt) {
      atomicAdd(stat_word(a.stats, kStatTested), (unsigned long long)nt);
      atomicAdd(stat_word(a.stats, kStatCulled), (unsigned long long)nc);
    }
  }
  __syncthreads();
  if (live) p.list[1 + *list_base_p + base + __popcll(mask & ((1ull << lane) - 1ull))] = block;
}
// workgroups (blocks) of a long transform's plan, whatever its layout
__host__ __device__ inline int plan_blocks_of(const PlanLongDet& a, const PlanLongArgs& p, int n) {
  if (p.layout == 3) return plan_x256_blocks(a.nframes, a.shift);
  if (p.layout == 2) return plan_dif8_blocks(a.nframes, a.shift, p.logn - 13);
  return plan_long_blocks(p.layout, p.cols, n);
}

template <int G, int GX, int TF, int TB_ = 256>
__global__ __launch_bounds__(256) void k_plan_long(PlanLongDet a, PlanLongArgs p) {
  __shared__ float mrow[kPlanLongFloats];
  __shared__ int book[kPlanLongInts];
  if (p.layout == 2) return plan_dif8_run<G, GX, TF, TB_>(a, p, (int)blockIdx.x, (int)threadIdx.x, mrow, book);
  if (p.layout == 3) return plan_x256_run<G, GX, TF, TB_>(a, p, (int)blockIdx.x, (int)threadIdx.x, mrow, book);
  plan_long_run<G, GX, TF, TB_>(a, p, (int)blockIdx.x, (int)threadIdx.x, mrow, book);
}

// 2^20 points in two passes: the plan of call k as the FIRST workgroups of the column launch of call k + 1 (fft1024_kernels.h,
// 16-column tiles by 1024 threads). As a launch of its own between the row launch of call k and the column launch of call k + 1
// the plan cost a 16-frame call 11.6 us of a chip that waits for 3 MB of run maxima, plus a launch boundary; here its blocks —
// four to a workgroup of 1024 threads, so that the blocks whose columns share the 128-byte lines of the ring share a workgroup — hold
// `plan_wgs` of the launch's first slots for a few microseconds while the column tiles stream in beside them. What the plan reads
// (the run maxima of frames up to call k's last) was complete before this launch began; the rows of the ring the column tiles clear
// in the same launch are those of call k + 1's frames (the ring holds two batches and the averager's reach: ss_create), which the
// plan touches only where a ragged frame tile reaches past its batch and then ignores. Its consumer — the detect stage of call k —
// rides on the launch AFTER this one (the row half of call k + 1). `plan_wgs` is a multiple of 8: the column tiles keep their XCDs.
constexpr int kPlanFusedFloats = 4096;  // values of M per plan block when four blocks share the column tile's LDS (plan_long_cols: cap)
template <int FMT, bool WCALC>
__global__ __launch_bounds__(1024, 8) void k_fft_cols1024_plan(ColsArgs g, PlanLongDet a, PlanLongArgs p, int plan_wgs) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  static_assert(4 * (kPlanFusedFloats + kPlanLongInts) * 4 <= fft1024_cols_lds_bytes(4), "four plan blocks in a column tile's LDS");
  const int tid = (int)threadIdx.x, b = (int)blockIdx.x;
  if (b < plan_wgs) {  // (workgroup-uniform)
    const int sub = tid >> 8;
    float* mrow = reinterpret_cast<float*>(smem_raw) + sub * (kPlanFusedFloats + kPlanLongInts);
    // plan block number: XCD b mod 8 as k_plan_long's block numbering has it (plan_long_block), the workgroup's four blocks in
    // consecutive slots of that XCD
    plan_long_run<21, 21, 16, 256>(a, p, ((((b >> 3) << 2) + sub) << 3) | (b & 7), tid & 255, mrow, reinterpret_cast<int*>(mrow + kPlanFusedFloats));
    return;
  }
  fft_cols1024_tile<FMT, 4, WCALC>(g, b - plan_wgs, smem_raw, tid);
}
__host__ __device__ inline int plan_fused_wgs(int plan_blocks) { return (((plan_blocks + 3) / 4 + 7) / 8) * 8; }

// min of the noise ceiling over bins [256 c - 32, 256 c + 288) is k_thr_tilemin above, one wave per tile column.

// The general path's per-row work, written for the instruction count: a tile that takes this path is evaluated by a workgroup whose
// end the launch waits for, its waves share their SIMDs with six or seven others, and an instruction of theirs is issued every ~5 ns —
// the two passes cost such a tile 8.4 us each against 1.8 us for a steady tile's (profiles/r05/s32_*: 20 scalar instructions per row for
// its address and ~30, branches included, for its value). Both are block-uniform decisions on the row's frame number:
//   general_row_address   row fr < 0 at `before_end` + fr rows (clamped to the rows there are), fr >= 0 at psd + min(fr, nframes - 1) rows
//   general_row_value     fr < 0: x - (ceiling or 0.0f);  learning frame: kNoData;  frame of the batch: x - ceiling;  past its end: 0.0f
//                         — as masks on the operands (x - 0.0f is x bit for bit), no branch
#ifndef SS_GENERAL_ROW_OLD  // (A/B builds, scripts/build_ab.py: 1 = the expressions as they were until session 33 of round 5)
#define SS_GENERAL_ROW_OLD 0
#endif
__device__ __forceinline__ const char* general_row_address(const char* psd, const char* before_end, int before_rows, int nframes, ptrdiff_t row_bytes, int fr) {
#if SS_GENERAL_ROW_OLD
  const char* before_base = before_end - (ptrdiff_t)before_rows * row_bytes;
  return fr < 0 ? before_base + (size_t)max(before_rows + fr, 0) * (size_t)row_bytes : psd + (size_t)min(fr, nframes - 1) * (size_t)row_bytes;
#else
  const int idx = min(max(fr, -before_rows), nframes - 1);
  return (fr < 0 ? before_end : psd) + (ptrdiff_t)idx * row_bytes;  // (tried: a 32 x 32 -> 64-bit product by hand — the compiler expands __mulhi on scalars into the same six instructions)
#endif
}
__device__ __forceinline__ float general_row_value(float x, float t, int fr, int nframes, int n_learn, int ring_db_from, bool before_are_psd) {
#if SS_GENERAL_ROW_OLD
  const float t_before = before_are_psd ? t : 0.0f;
  const float w = fr < n_learn ? kNoData : x - t;
  return fr < 0 ? x - (fr >= ring_db_from ? t : t_before) : (fr < nframes ? w : 0.0f);
#endif
  const bool kept = fr < 0 || (fr < nframes && fr >= n_learn);                       // the value is x - (t or 0.0f)
  const bool with_t = fr >= 0 || fr >= ring_db_from || before_are_psd;                 // ... t: rows of the batch, dB rows of the ring, plane rows
  const uint32_t other = fr < nframes ? __float_as_uint(kNoData) : 0u;                 // not kept: a learning frame (noise_learner.cpp:49), or past the batch's end
  const uint32_t tm = with_t ? 0xffffffffu : 0u, km = kept ? 0xffffffffu : 0u;
  const float v = x - __uint_as_float(__float_as_uint(t) & tm);                       // noise_learner.cpp:55
  return __uint_as_float((__float_as_uint(v) & km) | (other & ~km));
}

// ... and both for the 36 rows at once: lane l of a wave works out row l's address and masks with a dozen VECTOR instructions, the
// wave then picks them up row by row with v_readlane (uniform again: the loads keep their scalar base) — three instructions per row
// for the address and load, six for the value, where the scalar unit spent 21 + 12 in turn (each behind the one before).
#ifndef SS_GENERAL_ROW_LANES  // (A/B builds: 0 = row by row on the scalar unit, as sessions 33's first form)
#define SS_GENERAL_ROW_LANES 1
#endif
struct GeneralRowsOfLane {
  unsigned lo, hi, tm, km, other;
};
__device__ __forceinline__ GeneralRowsOfLane general_rows_of_lane(const char* psd, const char* before_end, int before_rows, int nframes, ptrdiff_t row_bytes, int fr0, int lane,
                                                                   int n_learn, int ring_db_from, bool before_are_psd) {
  const int fr = fr0 + lane;  // (lanes beyond the 36th: a clamped, legal address nobody asks for)
  const unsigned long long at = (unsigned long long)(uintptr_t)general_row_address(psd, before_end, before_rows, nframes, row_bytes, fr);
  GeneralRowsOfLane g;
  g.lo = (unsigned)at;
  g.hi = (unsigned)(at >> 32);
  const bool kept = fr < 0 || (fr < nframes && fr >= n_learn);
  const bool with_t = fr >= 0 || fr >= ring_db_from || before_are_psd;
  g.other = fr < nframes ? __float_as_uint(kNoData) : 0u;
  g.tm = with_t ? 0xffffffffu : 0u;
  g.km = kept ? 0xffffffffu : 0u;
  return g;
}
typedef const __attribute__((address_space(1))) char* global_bytes;  // (global memory, said so: the loads keep a scalar base and a 32-bit lane offset)
template <int R>
__device__ __forceinline__ global_bytes general_row_address_at(const GeneralRowsOfLane& g) {
  // (v_readlane's result is an int: through unsigned, or the low word's sign bit floods the high one)
  const unsigned long long at = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(g.hi, R) << 32) | (unsigned long long)(unsigned)__builtin_amdgcn_readlane(g.lo, R);
  return (global_bytes)at;
}
template <int R>
__device__ __forceinline__ float general_row_value_at(const GeneralRowsOfLane& g, float x, float t) {
  const unsigned tm = __builtin_amdgcn_readlane(g.tm, R), km = __builtin_amdgcn_readlane(g.km, R), other = __builtin_amdgcn_readlane(g.other, R);
  const float v = x - __uint_as_float(__float_as_uint(t) & tm);
  return __uint_as_float((__float_as_uint(v) & km) | (other & ~km));
}
// x[r] = the row's value at byte offset `coff`, r = 0 .. ROWS - 1 (compile-time recursion: v_readlane takes its lane as an immediate)
template <int R, int ROWS>
__device__ __forceinline__ void general_rows_load(const GeneralRowsOfLane& g, uint32_t coff, float (&x)[ROWS]) {
  if constexpr (R < ROWS) {
    x[R] = *(const __attribute__((address_space(1))) float*)(general_row_address_at<R>(g) + coff);
    general_rows_load<R + 1, ROWS>(g, coff, x);
  }
}
template <int R, int ROWS>
__device__ __forceinline__ void general_rows_load2(const GeneralRowsOfLane& g, uint32_t coff0, uint32_t coff1, float (&x)[ROWS], float (&y)[ROWS]) {
  if constexpr (R < ROWS) {
    const global_bytes src = general_row_address_at<R>(g);
    x[R] = *(const __attribute__((address_space(1))) float*)(src + coff0);
    y[R] = *(const __attribute__((address_space(1))) float*)(src + coff1);
    general_rows_load2<R + 1, ROWS>(g, coff0, coff1, x, y);
  }
}
template <int R, int ROWS>
__device__ __forceinline__ void general_rows_settle(const GeneralRowsOfLane& g, float t, float (&x)[ROWS]) {
  if constexpr (R < ROWS) {
    x[R] = general_row_value_at<R>(g, x[R], t);
    general_rows_settle<R + 1, ROWS>(g, t, x);
  }
}

// Which of the tile's 276 columns (256 + 10 either side, in bin order) a thread takes in phase 1. Rows in bin order: thread = column
// (256, then 20 more). The fold's rows (PERM8 = log2 Q, fft65536_dif8.h): a tile's own 256 bins are one block of the row (Q = 8) or
// half of one (Q = 16) with the residues' runs side by side, consecutive BINS 128 bytes apart — so the first 256 threads take the
// tile's own columns in the block's order, thread g * (256 / Q) + j the bin Q j + g: a wave's load is 256 contiguous bytes (four
// runs of 64) instead of sixteen lanes on eight lines — and the 20 columns either side go to the second pass. The tile in LDS stays
// in bin order.
template <int PERM8, int A, int TB>
__device__ __forceinline__ int tile_column_of(int pass_c, int tid) {
  if constexpr (PERM8 == 0) {
    return pass_c * TB + tid;
  } else {
    constexpr int RUN = TB >> PERM8;  // bins of a residue in the tile
    if (pass_c == 0) return A + ((tid & (RUN - 1)) << PERM8) + tid / RUN;
    return tid < A ? tid : TB + tid;  // (tid < 2 A)
  }
}

// One tile of the fused back end. `block` = tile number (what blockIdx.x is for the stand-alone kernel), `tid` = 0..TB-1,
// `tile` / `cnt` = this tile's LDS (TF * P floats, TF ints). `valid` = false: the caller has no tile for these threads (odd
// tile count in a two-tile workgroup) — they only keep the workgroup's barriers company. Every __syncthreads() below is
// reached by all threads of the workgroup whatever `valid`, `steady` or `interior` are.
// PERM8 (0, or log2 of the fold's radix: 3, 4): the rows the tile reads (a.psd, the ring) and writes (the ring) are the radix-Q fold's rows
// of 8192 Q bins in BLOCKS of 32 Q bins — bin Q k' + g at (k' / 32) * 32 Q + 32 g + k' % 32 (fft65536_dif8.h: dif_bin_offset; dB values
// from ring_db_from on) —; calls that hand out no plane only (no rel_out). A tile's thread g * 32 + j takes the bin 8 j + g: a wave's 36
// row loads are 256 contiguous bytes (Q = 8) or four runs of 64 (Q = 16); everything else — the noise ceiling, the pass mask, mask bits,
// sparse averages — stays in bin order.
template <int G, int GX, int TF, int TB_ = 256, bool SPEC = false, int PERM8 = 0>
__device__ __forceinline__ void detect_tile(const DetectArgs& a, int block, int tid, float* __restrict__ tile, int* __restrict__ cnt, bool valid) {
  using T = DetectTile<G, GX, TF, TB_>;
  constexpr int A = T::A, TB = T::TB, P = T::P, ROWS = T::ROWS, H = T::H, SEGW = T::SEGW, NSEG = T::NSEG, YW = T::YW;
  const int n = a.n, nframes = a.nframes;
  const int tiles_per_row = (n + TB - 1) / TB;
  // Frame tiles are dispatched rotated by one: the ragged last tile and the tiles that read ring rows take the
  // slower general path, so they go first and the straight-line tiles fill in behind them instead of leaving a tail.
  const int nft = (nframes + a.shift + TF - 1) / TF;
  // (Tried for long rows and dropped: an XCD-aware column-major order — each XCD walking the frame tiles of one column
  // group after the other so that a tile finds its predecessor's halo rows in that XCD's L2. 65536 points x 128 frames:
  // the kernel went from 19.6 to 26.7 us.)
  const int ft_seq = block / tiles_per_row, bt = block % tiles_per_row;
  const int ft = (ft_seq + nft - 1) % nft;
  const int f0 = ft * TF - a.shift;  // batch-relative frame of the tile's first row of outputs; may be negative
  const int b0 = bt * TB;
  if (tid < TF) cnt[tid] = 0;
#ifdef SS_DIAG
  if (a.stamp_mid && valid && tid == 0) a.stamp_mid[4 * (size_t)block] = wall_clock64();
#endif
  if constexpr (SPEC) {
    if (valid && a.spec_prev_partial && ft_seq == 0) spectrogram_fold<TB>(a, tid, b0);  // the first-dispatched tile of each bin column
  }
  // block-uniform classification
  const bool interior = (b0 - A >= 0) && (b0 + TB + A <= n);
  const bool steady = (f0 - (G - 1) >= a.n_learn) && (f0 - (G - 1) >= 0) && (f0 + TF <= nframes);
  const int first_hist = a.hist_by_fft ? (1 << 30) : nframes - H;  // batch frames >= first_hist become the ring rows [frame - first_hist]
  const float* before_base = a.halo_psd ? a.halo_psd : a.hist_in;  // rows before the batch: the previous call's last frames, or the ring
  const int before_rows = a.halo_psd ? a.halo_rows : H;
  const bool writes_hist = f0 + TF > first_hist;

  // ---------------- phase 1: time means, thread = column ----------------
  if constexpr (PERM8 != 0) {
    // The fold's rows (KIND 8 / 9 of k_scan_step: 128 registers per thread). A pair of tiles holds a workgroup slot that a residue's
    // transform is waiting for, or ends the launch: what counts is how long ONE wave takes, and a wave issues an instruction every
    // four or five cycles — 1000 instructions are 2 us. Measured per tile, alone on the chip (profiles/r05/s29_*): first pass
    // 1.7 us, the second pass — twenty columns, the same 36 loads and the same latency over again — 1.5, and for the tiles that read
    // rows from before the batch 5-8 + 4-7 us: twenty scalar instructions per row and pass to pick the row's address. So here
    //   * a thread of the first twenty takes its second column's 36 rows in the same flight as its first;
    //   * rows that lie one after the other in memory — a steady tile's, and with the batch's rows right behind the ring's window
    //     (ring_place.h) every tile's but a ragged last one — are addressed as the steady path addresses them.
    // The arithmetic per value is the other paths' own.
    if (valid) {
      const bool two = tid < 2 * A;
      const int c0 = tile_column_of<PERM8, A, TB>(0, tid), c1 = tile_column_of<PERM8, A, TB>(1, tid);
      const int col0 = b0 - A + c0;  // one of the tile's own 256 columns: inside the band
      const int col1 = b0 - A + c1, col1c = min(max(col1, 0), n - 1);
      const bool in1 = col1 == col1c;
      const uint32_t off0 = (uint32_t)dif_bin_offset(col0, PERM8) * 4u, off1 = (uint32_t)dif_bin_offset(two ? col1c : col0, PERM8) * 4u;
      const float t0 = a.thr[col0], t1 = a.thr[two ? col1c : col0];
      const bool in_line = steady || (!a.halo_psd && a.psd == a.hist_in + (size_t)H * n && f0 + TF <= nframes);  // (f0 - 20 >= -H: shift < TF)
      float x[ROWS], y[ROWS];
      // (a non-steady tile's 36 rows on 36 lanes, general_rows_of_lane: every lane of the wave is active here, and the lanes stay together
      // wherever the result is read lane by lane — the second column is the whole first wave's business up to its stores, see the general
      // path below)
      const bool wave0 = tid < 64;
      [[maybe_unused]] GeneralRowsOfLane rows_l{};
#if SS_GENERAL_ROW_LANES
      if (!steady) {
        rows_l = general_rows_of_lane(reinterpret_cast<const char*>(a.psd), reinterpret_cast<const char*>(before_base) + (ptrdiff_t)before_rows * (ptrdiff_t)n * 4, before_rows, nframes,
                                      (ptrdiff_t)n * 4, f0 - (G - 1), tid & 63, a.n_learn, a.ring_db_from, a.halo_psd != nullptr);
        asm volatile("" : "+v"(rows_l.lo), "+v"(rows_l.hi), "+v"(rows_l.tm), "+v"(rows_l.km), "+v"(rows_l.other));
      }
#endif
      if (in_line) {
        const char* p = reinterpret_cast<const char*>(a.psd) + (ptrdiff_t)(f0 - (G - 1)) * (ptrdiff_t)n * 4;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) x[r] = load_row_value(p + (size_t)r * n * 4 + off0);
        if (wave0) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) y[r] = load_row_value(p + (size_t)r * n * 4 + off1);
        }
      } else {
#if SS_GENERAL_ROW_LANES
        general_rows_load2<0, ROWS>(rows_l, off0, off1, x, y);  // (the second column by every thread: a branch per row would cost more than the loads)
#else
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const char* src = general_row_address(reinterpret_cast<const char*>(a.psd), reinterpret_cast<const char*>(before_base) + (ptrdiff_t)before_rows * (ptrdiff_t)n * 4,
                                                before_rows, nframes, (ptrdiff_t)n * 4, f0 - (G - 1) + r);
          x[r] = *reinterpret_cast<const float*>(src + off0);
          y[r] = *reinterpret_cast<const float*>(src + off1);
        }
#endif
      }
      // rel = value - ceiling; learning frames, rows of the ring, frames past a ragged end as the general path below has them
      const auto settle = [&](float (&v)[ROWS], float t) {
        if (steady) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) v[r] -= t;
        } else {
#if SS_GENERAL_ROW_LANES
          general_rows_settle<0, ROWS>(rows_l, t, v);
#else
#pragma unroll
          for (int r = 0; r < ROWS; ++r) v[r] = general_row_value(v[r], t, f0 - (G - 1) + r, nframes, a.n_learn, a.ring_db_from, a.halo_psd != nullptr);
#endif
        }
      };
      settle(x, t0);
      if (wave0) settle(y, t1);
#pragma unroll
      for (int j = 0; j < TF; ++j) {
        const int fr = f0 + j;
        if (fr >= 0 && fr < nframes) {
          if (a.rel_out) store_row(a.rel_out, fr, n, off0, x[G - 1 + j]);
          if (fr >= first_hist) store_row(a.hist_out, fr - first_hist, n, off0, x[G - 1 + j]);
        }
      }
      if (steady) time_means_to_tile<G, TF, P, false>(x, &tile[c0], 0);
      else time_means_to_tile<G, TF, P, true>(x, &tile[c0], a.pushed_before + f0 + 1);
      if (two) {
        if (!in1) {
#pragma unroll
          for (int j = 0; j < TF; ++j) tile[j * P + c1] = 0.0f;  // outside the band: contributes exactly nothing to the clipped window sums
        } else if (steady) {
          time_means_to_tile<G, TF, P, false>(y, &tile[c1], 0);
        } else {
          time_means_to_tile<G, TF, P, true>(y, &tile[c1], a.pushed_before + f0 + 1);
        }
      }
    }
  } else if (!valid) {
    // nothing
  } else if (steady) {
    // straight line: ROWS independent, unconditional loads per column; 276 columns over 256 threads.
    // Columns outside the band (first / last tile of a row) read a clamped address and contribute 0.0f.
#pragma unroll
    for (int pass_c = 0; pass_c < 2; ++pass_c) {
#ifdef SS_DIAG
      if (pass_c == 1 && a.stamp_mid && tid == 0) a.stamp_mid[4 * (size_t)block + 1] = wall_clock64();
#endif
      const int c = tile_column_of<PERM8, A, TB>(pass_c, tid);
      if (pass_c == 0 || tid < 2 * A) {
        const int col = b0 - A + c;
        const int colc = min(max(col, 0), n - 1);
        const bool in_band = col == colc;
        const float t = a.thr[colc];
        // block-uniform row base (scalar registers) + one 32-bit per-thread byte offset
        const char* p = reinterpret_cast<const char*>(a.psd + (size_t)(f0 - (G - 1)) * n);
        const uint32_t coff = (uint32_t)(PERM8 ? dif_bin_offset(colc, PERM8) : colc) * 4u;
        float x[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) x[r] = load_row_value(p + (size_t)r * n * 4 + coff) - t;
        if (!interior) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) x[r] = in_band ? x[r] : 0.0f;
        }
        time_means_to_tile<G, TF, P, false>(x, &tile[c], 0);
        if (in_band && c >= A && c < A + TB) {
          if (a.rel_out) {
#pragma unroll
            for (int j = 0; j < TF; ++j) store_row(a.rel_out, f0 + j, n, coff, x[G - 1 + j]);
          }
          if (writes_hist) {
#pragma unroll
            for (int j = 0; j < TF; ++j)
              if (f0 + j >= first_hist) store_row(a.hist_out, f0 + j - first_hist, n, coff, x[G - 1 + j]);
          }
        }
      }
    }
  } else if (PERM8 == 0 && SS_STEADY_HALO && a.halo_psd && a.n_learn == 0 && f0 - (G - 1) >= -a.halo_rows && f0 + TF <= nframes && a.pushed_before >= G && !a.rel_out &&
             !writes_hist) {
    // Round 6 (SS_STEADY_HALO=1 builds; measured, not kept) — a steady tile whose first rows lie before the batch, in the halo frames' dB plane (8192 points, deep pipelining: the first
    // two frame tiles of every batch): every one of its 36 rows is a dB row of a full averager, as a steady tile's, only in two pieces of
    // memory — a scalar select of the row's base, the straight-line path's arithmetic on the same values. (On the general path — an
    // address and three masks per row for cases that cannot occur here — such a tile took 12 us against a steady tile's 6, and the
    // workgroups that evaluate them were the last of their launch and the whole of a drain's detect launch: profiles/r05/s37_summary.txt,
    // profiles/r06/s1_timeline_k20.txt.) The tile's frames before the batch (f0 + j < 0) get time means nobody reads: phase 2 skips them.
#pragma unroll
    for (int pass_c = 0; pass_c < 2; ++pass_c) {
#ifdef SS_DIAG
      if (pass_c == 1 && a.stamp_mid && tid == 0) a.stamp_mid[4 * (size_t)block + 1] = wall_clock64();
#endif
      const int c = tile_column_of<PERM8, A, TB>(pass_c, tid);
      if (pass_c == 0 || tid < 2 * A) {
        const int col = b0 - A + c;
        const int colc = min(max(col, 0), n - 1);
        const bool in_band = col == colc;
        const float t = a.thr[colc];
        const char* pb = reinterpret_cast<const char*>(a.psd);
        const char* hb = reinterpret_cast<const char*>(a.halo_psd + (size_t)a.halo_rows * n);  // (its end: frame -1 is the row before it)
        const uint32_t coff = (uint32_t)colc * 4u;
        float x[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
          const int fr = f0 - (G - 1) + r;  // (block-uniform)
          x[r] = load_row_value((fr < 0 ? hb : pb) + (ptrdiff_t)fr * (ptrdiff_t)n * 4 + coff) - t;
        }
        if (!interior) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) x[r] = in_band ? x[r] : 0.0f;
        }
        time_means_to_tile<G, TF, P, false>(x, &tile[c], 0);
      }
    }
  } else {
    // general path: ring rows from before the batch, learning frames, averager warm-up, ragged batch end.
    // Same arithmetic as above on the same values, so a frame's result does not depend on the path.
    // (two straight passes over the 276 columns, not a loop: a loop makes the compiler hoist all 36 row
    // pointers out of it and spill scalar registers)
#pragma unroll
    for (int pass_c = 0; pass_c < 2; ++pass_c) {
#ifdef SS_DIAG
      if (pass_c == 1 && a.stamp_mid && tid == 0) a.stamp_mid[4 * (size_t)block + 1] = wall_clock64();
#endif
#if SS_GENERAL_ROW_LANES
      // The 36 rows' addresses and masks on 36 lanes (general_rows_of_lane), read back lane by lane. A lane's registers are only safe from
      // the register allocator while the lane runs the same code as its readers (a lane that has branched off "needs them no more" and may
      // find them reused there): so a wave takes part in a pass as a whole or not at all — the second pass is the tile's first wave, its
      // lanes without a column and the lanes outside the band go through the loads on a legal column of their own and part only at the
      // stores.
      if (pass_c == 0 || tid < 64) {
        GeneralRowsOfLane rows_l = general_rows_of_lane(reinterpret_cast<const char*>(a.psd), reinterpret_cast<const char*>(before_base) + (ptrdiff_t)before_rows * (ptrdiff_t)n * 4,
                                                        before_rows, nframes, (ptrdiff_t)n * 4, f0 - (G - 1), tid & 63, a.n_learn, a.ring_db_from, a.halo_psd != nullptr);
        asm volatile("" : "+v"(rows_l.lo), "+v"(rows_l.hi), "+v"(rows_l.tm), "+v"(rows_l.km), "+v"(rows_l.other));
        const bool has_column = pass_c == 0 || tid < 2 * A;  // 276 columns over 256 threads
        const int c = tile_column_of<PERM8, A, TB>(pass_c, has_column ? tid : 0);
        const int col = b0 - A + c, colc = min(max(col, 0), n - 1);
        const bool in_band = col == colc;
        const float t = a.thr[colc];
        const uint32_t coff = (uint32_t)(PERM8 ? dif_bin_offset(colc, PERM8) : colc) * 4u;
        float x[ROWS];
        // all loads first, unconditional, on always-legal addresses: independent and in flight together. Ring rows hold rel values — or dB
        // values, from ring_db_from on —, plane rows (the halo frames) are PSD
        general_rows_load<0, ROWS>(rows_l, coff, x);
        general_rows_settle<0, ROWS>(rows_l, t, x);
        if (has_column) {
          if (!in_band) {
#pragma unroll
            for (int j = 0; j < TF; ++j) tile[j * P + c] = 0.0f;  // outside the band: contributes exactly nothing to the clipped window sums
          } else {
            if (c >= A && c < A + TB) {
#pragma unroll
              for (int j = 0; j < TF; ++j) {
                const int fr = f0 + j;
                if (fr >= 0 && fr < nframes) {
                  if (a.rel_out) store_row(a.rel_out, fr, n, coff, x[G - 1 + j]);
                  if (fr >= first_hist) store_row(a.hist_out, fr - first_hist, n, coff, x[G - 1 + j]);
                }
              }
            }
            time_means_to_tile<G, TF, P, true>(x, &tile[c], a.pushed_before + f0 + 1);
          }
        }
      }
#else
      const int c = tile_column_of<PERM8, A, TB>(pass_c, tid);
      const int col = b0 - A + c;
      if (pass_c == 1 && tid >= 2 * A) {
        // nothing: 276 columns over 256 threads
      } else if (col < 0 || col >= n) {
#pragma unroll
        for (int j = 0; j < TF; ++j) tile[j * P + c] = 0.0f;  // outside the band: contributes exactly nothing to the clipped window sums
      } else {
        const bool main_col = c >= A && c < A + TB;
        const float t = a.thr[col];
        // all loads first, unconditional, on always-legal addresses: independent and in flight together; a
        // branch around each load would serialise them on s_waitcnt. The row pointer is block-uniform:
        // frames before the batch come from the ring (row H + frame), frames past its end are clamped.
        const uint32_t coff = (uint32_t)(PERM8 ? dif_bin_offset(col, PERM8) : col) * 4u;
        float x[ROWS];
        const char* before_end = reinterpret_cast<const char*>(before_base) + (ptrdiff_t)before_rows * (ptrdiff_t)n * 4;
        // ring rows hold rel values — or dB values, from ring_db_from on —, plane rows (the halo frames) are PSD
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          x[r] = *reinterpret_cast<const float*>(general_row_address(reinterpret_cast<const char*>(a.psd), before_end, before_rows, nframes, (ptrdiff_t)n * 4, f0 - (G - 1) + r) + coff);
#pragma unroll
        for (int r = 0; r < ROWS; ++r) x[r] = general_row_value(x[r], t, f0 - (G - 1) + r, nframes, a.n_learn, a.ring_db_from, a.halo_psd != nullptr);
        if (main_col) {
#pragma unroll
          for (int j = 0; j < TF; ++j) {
            const int fr = f0 + j;
            if (fr >= 0 && fr < nframes) {
              if (a.rel_out) store_row(a.rel_out, fr, n, coff, x[G - 1 + j]);
              if (fr >= first_hist) store_row(a.hist_out, fr - first_hist, n, coff, x[G - 1 + j]);
            }
          }
        }
        time_means_to_tile<G, TF, P, true>(x, &tile[c], a.pushed_before + f0 + 1);
      }
#endif
    }
  }
  __syncthreads();
#ifdef SS_DIAG
  if (a.stamp_mid && valid && tid == 0) a.stamp_mid[4 * (size_t)block + 2] = wall_clock64();
#endif

  // ---------------- phase 2: frequency means + threshold, thread = (frame, 16-bin segment) ----------------
#pragma unroll
  for (int q0 = 0; q0 < TF * NSEG; q0 += TB) {
    const int q = q0 + tid;
    const int fj = q % TF;
    const int seg = q / TF;
    const int f = f0 + fj;
    const int i0 = b0 + seg * SEGW;  // first bin of the segment
    const bool live = valid && f >= 0 && f < nframes && i0 < n;
    uint32_t bits = 0;
    if (live) {
      float outv[SEGW];
      const float* row = &tile[fj * P + seg * SEGW];  // row[k] = avgY(f, i0 - A + k)
      float y[YW];
#pragma unroll
      for (int k4 = 0; k4 < YW / 4; ++k4) {
        const float4 v4 = *reinterpret_cast<const float4*>(row + 4 * k4);
        y[4 * k4] = v4.x;
        y[4 * k4 + 1] = v4.y;
        y[4 * k4 + 2] = v4.z;
        y[4 * k4 + 3] = v4.w;
      }
      // Bins outside the band hold 0.0f in the tile, so the same straight-line sums give the reference's
      // edge-clipped window (adding or subtracting 0.0f is exact); only the divisor changes there.
      float sum = 0.0f;
#pragma unroll
      for (int k = 0; k < GX; ++k) sum += y[k];  // lowest bin first
      float sums[SEGW];
      sums[0] = sum;
#pragma unroll
      for (int o = 1; o < SEGW; ++o) {
        sum -= y[o - 1];       // utils.cpp:41  sum -= input[first]
        sum += y[o + GX - 1];  // utils.cpp:45  sum += input[last]
        sums[o] = sum;
      }
      if (interior) {
#pragma unroll
        for (int o = 0; o < SEGW; ++o) outv[o] = div_const<GX>(sums[o]);
      } else {
#pragma unroll
        for (int o = 0; o < SEGW; ++o) {
          const int i = i0 + o;
          const int count = min(n - 1, i + A) - max(0, i - A) + 1;
          outv[o] = sums[o] / (float)count;  // sum / count (float / int), utils.cpp:50
        }
      }
      const uint4 pm = *reinterpret_cast<const uint4*>(a.pass + i0);  // 16 pass bytes (n is a multiple of 64)
      const uint32_t pw[4] = {pm.x, pm.y, pm.z, pm.w};
#pragma unroll
      for (int o = 0; o < SEGW; ++o) {
        const bool ok = ((pw[o >> 2] >> (8 * (o & 3))) & 0xffu) != 0;
        bits |= (a.start_level <= outv[o] && ok) ? (1u << o) : 0u;
      }
      if (a.avg_out) {
        float4* dst = reinterpret_cast<float4*>(a.avg_out + (size_t)f * n + i0);
#pragma unroll
        for (int k4 = 0; k4 < SEGW / 4; ++k4) dst[k4] = make_float4(outv[4 * k4], outv[4 * k4 + 1], outv[4 * k4 + 2], outv[4 * k4 + 3]);
      } else if (bits) {
        float* dst = a.avg_sparse + (size_t)f * n + i0;
#pragma unroll
        for (int o = 0; o < SEGW; ++o)
          if (bits & (1u << o)) dst[o] = outv[o];
      }
    }
    // lanes q and q + TF hold the same frame, segments seg and seg + 1: together one 32-bit mask word
    const uint32_t hi_bits = __shfl_down(bits, TF);
    if ((seg & 1) == 0 && live) {
      const uint32_t word = bits | (hi_bits << 16);
      a.maskbits[((size_t)f * n + i0) >> 5] = word;
      if (word) atomicAdd(&cnt[fj], __popc(word));
    }
  }
  __syncthreads();
#ifdef SS_DIAG
  if (a.stamp_mid && valid && tid == 0) a.stamp_mid[4 * (size_t)block + 3] = wall_clock64();
#endif
  if (valid && tid < TF && cnt[tid] != 0) atomicAdd(&a.counts[f0 + tid], cnt[tid]);
  if constexpr (SPEC) {
    // (the __syncthreads above freed the avgY tile)
    if (valid) spectrogram_tile_means<TF, TB>(a, tile, tid, f0, b0);
    __syncthreads();
    if (valid) spectrogram_tile_sum<TF, TB>(a, tile, tid, ft, f0, b0);
  }
}

// Stand-alone launch: one tile per workgroup of TB threads (FFT sizes other than 8192; the 8192-point chain runs the same
// tile code as a role of k_scan_step, scan_step.h).
template <int G, int GX, int TF, int TB_ = 256, bool SPEC = false>
__global__ __launch_bounds__(TB_) __attribute__((amdgpu_waves_per_eu(8))) void k_detect_fused(DetectArgs a) {
  using T = DetectTile<G, GX, TF, TB_>;
  __shared__ __attribute__((aligned(16))) float tile[TF * T::P];
  __shared__ int cnt[TF];
  detect_tile<G, GX, TF, TB_, SPEC>(a, (int)blockIdx.x, (int)threadIdx.x, tile, cnt, true);
}

// Row copy used when the sliding ring window reaches the end of its buffer and moves back to the front
// (source and destination never overlap: the buffer holds at least three windows).
__global__ void k_hist_shift(const float* __restrict__ hist_in, float* __restrict__ hist_out, int n, int keep_rows, int nframes) {
  const size_t total = (size_t)keep_rows * n;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    hist_out[e] = hist_in[e + (size_t)nframes * n];
  }
}

// Deep pipelining (8192 points): the ring rows a call's detect tiles would have written — rel = dB - thr of its newest `rows`
// frames, the same subtraction on the same values — written when the pipeline is drained instead of by every call: between
// drains nobody reads them (a call's rows from before its batch come from the previous call's frames, transformed once more),
// and the tiles that wrote them, three frame tiles of every call, could not be culled: a third of all evaluated tiles.
struct RingFillArgs {  // up to two calls' rows in one launch (blockIdx.y)
  const float* psd_tail[2];
  const float* thr[2];
  float* hist_out[2];
};
__global__ void k_ring_fill(RingFillArgs a, int n, int rows) {
  const float* __restrict__ src = a.psd_tail[blockIdx.y];
  const float* __restrict__ thr = a.thr[blockIdx.y];
  float* __restrict__ dst = a.hist_out[blockIdx.y];
  const size_t total = (size_t)rows * n;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) dst[e] = src[e] - thr[e % n];
}

// The drain of the deep pipeline (drain_deep, specscan.hip): a queue says "my last stage has run" (one wave behind it), and one wave
// on the public stream sleeps until `want` such words have been said since ss_create, or `limit` ticks of the 100 MHz clock have passed.
__global__ void k_drain_signal(unsigned* done) {
  if (threadIdx.x == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_drain_wait(const unsigned* done, unsigned want, long long limit) {
  const long long t0 = wall_clock64();
  // (signed distance: the counter wraps after 2^32 signals)
  while ((int)(__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0 && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(16);
}

// Candidate lists (CSR) from the mask bits: one wave per frame. The frame's offset is the sum of the
// counts of the frames before it (8 counts per lane per trip of 512 frames). The wave pulls 256 mask words
// per trip (one 16-byte load per lane), ranks them with one wave scan, expands the set bits into an LDS
// list at their ranks, then streams the list out with coalesced stores — ascending bins, frames in
// order, deterministic. counts_clear (the counter buffer a later batch will accumulate into, last used by a batch of
// clear_n frames) is zeroed on the way. Everything a trip needs from global memory is requested before anything is
// waited for. The wave synchronises with nobody: its LDS list is private (LDS operations of one wave complete in
// order), so eight frames can share a workgroup (k_scan_step) without a barrier in sight.
struct EmitArgs {
  const uint32_t* maskbits;
  int words_per_row, n, nframes;
  const int* counts;
  int* counts_clear;
  int clear_n;
  const float* avg;  // avg[f * n + bin] for every hit bin (the sparse plane, or the caller's full avg plane)
  int cap;
  int* live_clear;   // DetectArgs::live of the call (or null): its header words go back to zero here
  int clear_masks;   // set every mask word found set back to zero once it has been listed: the next user of the buffer may then
                     // leave the words of tiles it does not evaluate alone (tile culling)
  int* off_int;      // [nframes + 1] the library's own copy of the offsets
  int* off_out;      // caller's cand_off or null
  int* cand_idx;     // null: offsets only
  float* cand_avg;
};

constexpr int kEmitList = 1024;  // ints of LDS per wave

__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// offset of frame f = sum of counts[0..f); lane 0 records it (and the grand total behind the last frame) and zeroes this
// frame's share of the counter buffer a later batch will use. Returns the offset to every lane.
__device__ __forceinline__ int emit_frame_offset(const EmitArgs& a, int f, int lane, int mine) {
  const int nframes = a.nframes;
  int part = 0;
  for (int base = 0; base < f; base += 512) {  // 8 ints per lane per trip, all loads independent
    int v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a.counts[min(base + k * 64 + lane, nframes - 1)];
#pragma unroll
    for (int k = 0; k < 8; ++k) part += (base + k * 64 + lane) < f ? v[k] : 0;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  const int begin = part;
  if (f == 0 && a.live_clear) {  // the call's lists are spent: list-written flags and every copy of the count words back to zero
    if (lane < kLiveLists) a.live_clear[lane] = 0;
#pragma unroll
    for (int k = 0; k < kLiveCopies * kLiveLists / 64; ++k) {
      const int e = k * 64 + lane;
      a.live_clear[kLiveCounts + (e / kLiveLists) * kLiveCopyStride + e % kLiveLists] = 0;
    }
  }
  if (lane == 0) {
    a.off_int[f] = begin;
    if (a.off_out) a.off_out[f] = begin;
    if (f == nframes - 1) {
      a.off_int[nframes] = begin + mine;
      if (a.off_out) a.off_out[nframes] = begin + mine;
    }
    for (int g = f; g < a.clear_n; g += nframes) a.counts_clear[g] = 0;
  }
  return begin;
}

// The candidates of mask words [word_lo, word_hi) of frame f, written from list position `carry` on. `w4` holds the first
// trip's words when `preloaded` (requested before the offsets were summed, so the two latencies overlap).
__device__ __forceinline__ void emit_span(const EmitArgs& a, int f, int lane, int* __restrict__ list, int word_lo, int word_hi, int carry, uint4 w4,
                                          bool preloaded) {
  constexpr int LIST = kEmitList;
  const int words_per_row = a.words_per_row, n = a.n;
  const uint32_t* row = a.maskbits + (size_t)f * words_per_row;
  const bool wide = (words_per_row & 3) == 0;
  const float* arow = a.avg + (size_t)f * n;
  const int words_per_trip = wide ? 256 : 64;
  for (int base = word_lo; base < word_hi; base += words_per_trip) {
    if (base > word_lo || !preloaded) {
      w4 = make_uint4(0u, 0u, 0u, 0u);
      if (wide) {
        if (base + 4 * lane < word_hi) w4 = *reinterpret_cast<const uint4*>(row + base + 4 * lane);
      } else if (base + lane < word_hi) {
        w4.x = row[base + lane];
      }
    }
    const uint32_t wv[4] = {w4.x, w4.y, w4.z, w4.w};
    const int c = __popc(wv[0]) + __popc(wv[1]) + __popc(wv[2]) + __popc(wv[3]);
    if (a.clear_masks && c != 0) {  // listed below; the buffer's next user finds them zero (tile culling)
      uint32_t* wrow = const_cast<uint32_t*>(row);
      if (wide) *reinterpret_cast<uint4*>(wrow + base + 4 * lane) = make_uint4(0u, 0u, 0u, 0u);
      else wrow[base + lane] = 0u;
    }
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      if (lane >= d) incl += v;
    }
    const int total = __shfl(incl, 63);
    if (total == 0) continue;  // wave-uniform
    const int first_word = wide ? base + 4 * lane : base + lane;
    // rank of the first candidate of each of this lane's four words inside the trip
    int wstart[4];
    wstart[0] = incl - c;
    wstart[1] = wstart[0] + __popc(wv[0]);
    wstart[2] = wstart[1] + __popc(wv[1]);
    wstart[3] = wstart[2] + __popc(wv[2]);
    // expand in pieces of LIST entries (a whole row of hits does not fit the list at once)
    for (int lo = 0; lo < total; lo += LIST) {
      // Hits cluster in a few words (a transmission is a run of adjacent bins), so the words are expanded
      // one per half-wave, one lane per BIT: the half-wave's 32 lanes test the 32 bits of a broadcast word
      // and write the set ones at their ranks. No per-lane serial loop over bits.
      const int half = lane >> 5, bit = lane & 31;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        unsigned long long nz = __ballot(wv[k] != 0u);
        while (nz) {  // wave-uniform
          const int l0 = __ffsll((long long)nz) - 1;
          nz &= nz - 1;
          int l1 = l0;
          if (nz) {
            l1 = __ffsll((long long)nz) - 1;
            nz &= nz - 1;
          }
          const int src = half ? l1 : l0;
          const uint32_t bits = __shfl(wv[k], src);
          const int start = __shfl(wstart[k], src);
          const int word = __shfl(first_word, src) + k;
          const bool on = ((bits >> bit) & 1u) && !(half && l1 == l0);
          const int pos = start + __popc(bits & ((1u << bit) - 1u)) - lo;
          if (on && pos >= 0 && pos < LIST) list[pos] = word * 32 + bit;
        }
      }
      wave_lds_fence();  // this wave's list writes have landed
      const int cnt = min(LIST, total - lo);
      for (int p0 = 0; p0 < cnt; p0 += 256) {
        int idx[4];
        float av[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = p0 + k * 64 + lane;
          idx[k] = p < cnt ? list[p] : 0;
        }
        if (a.cand_avg) {
#pragma unroll
          for (int k = 0; k < 4; ++k) av[k] = arow[idx[k]];  // four independent loads in flight
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int p = p0 + k * 64 + lane;
          const int dst = carry + lo + p;
          if (p < cnt && dst < a.cap && a.cand_idx) {
            a.cand_idx[dst] = idx[k];
            if (a.cand_avg) a.cand_avg[dst] = av[k];
          }
        }
      }
      wave_lds_fence();  // the list has been read before the next piece overwrites it
    }
    carry += total;
  }
}

__device__ __forceinline__ void cand_emit_frame(const EmitArgs& a, int f, int lane, int* __restrict__ list) {
  const int words_per_row = a.words_per_row;
  const uint32_t* row = a.maskbits + (size_t)f * words_per_row;
  // first trip's mask words (words_per_row is a multiple of 2; rows of >= 256 words are 16-byte aligned)
  uint4 w4 = make_uint4(0u, 0u, 0u, 0u);
  if ((words_per_row & 3) == 0) {
    if (4 * lane < words_per_row) w4 = *reinterpret_cast<const uint4*>(row + 4 * lane);
  } else if (lane < words_per_row) {
    w4.x = row[lane];  // N = 64: two words per row, one per lane
  }
  const int mine = a.counts[f];
  const int begin = emit_frame_offset(a, f, lane, mine);
  if (mine == 0 || (!a.cand_idx && !a.clear_masks)) return;
  emit_span(a, f, lane, list, 0, words_per_row, begin, w4, true);
}

// Long rows (n >= 16384: 512 .. 32768 mask words per frame): W waves per frame, each taking a contiguous slice of the row —
// a count pass over its slice (the words come back from L2 for the expansion), a prefix over the W slice counts in LDS,
// then the same expansion from the slice's own start. One wave per frame walks 2^20-point rows in 128 serial trips (38 us
// per 16-frame batch); eight waves take 16 each.
// `lds` = (W * kEmitList + W + 1) ints; one frame per call, 64 * W threads.
template <int W>
__device__ __forceinline__ void cand_emit_frame_wide(const EmitArgs& a, int f, int tid, int* __restrict__ lds) {
  int* slice_cnt = lds + W * kEmitList;
  int* frame_begin = slice_cnt + W;
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int words_per_row = a.words_per_row;
  const int per = words_per_row / W;  // a multiple of 256 (the host picks W accordingly)
  const int lo = w * per, hi = lo + per;
  const uint32_t* row = a.maskbits + (size_t)f * words_per_row;
  const int mine = a.counts[f];
  int begin = 0;
  if (w == 0) begin = emit_frame_offset(a, f, lane, mine);
  int cnt = 0;
  const bool walk = mine != 0 && (a.cand_idx || a.clear_masks);  // (offsets only: the words still have to go back to zero)
  if (walk) {
    for (int base = lo; base < hi; base += 256) {
      const uint4 q = *reinterpret_cast<const uint4*>(row + base + 4 * lane);
      cnt += __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
  }
  if (lane == 0) slice_cnt[w] = cnt;
  if (w == 0 && lane == 0) *frame_begin = begin;
  __syncthreads();
  if (!walk) return;
  int carry = *frame_begin;
  for (int k = 0; k < w; ++k) carry += slice_cnt[k];
  if (cnt == 0) return;  // wave-uniform
  emit_span(a, f, lane, lds + w * kEmitList, lo, hi, carry, make_uint4(0u, 0u, 0u, 0u), false);
}

template <int W>
__global__ __launch_bounds__(64 * W) void k_cand_emit_wide(EmitArgs a) {
  __shared__ int lds[W * kEmitList + W + 1];
  cand_emit_frame_wide<W>(a, (int)blockIdx.x, (int)threadIdx.x, lds);
}

// Stand-alone launch: one wave per workgroup (FFT sizes other than 8192).
__global__ __launch_bounds__(64) void k_cand_emit(EmitArgs a) {
  __shared__ int list[kEmitList];
  cand_emit_frame(a, (int)blockIdx.x, (int)threadIdx.x, list);
}

}  // namespace ss
