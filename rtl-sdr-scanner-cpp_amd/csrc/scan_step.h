// scan_step.h — the chain's stages of three consecutive calls as ONE launch: k_scan_step (8192 points: the whole FFT + dB
// stage; 16384 points and more: the column half of the four-step FFT, the row half follows as its own launch).
//
// The reference runs its stages as a pipeline of blocks, each on its own thread, every block working on a different frame
// at any moment (sources/radio/sdr_device.cpp:161-171: blocker -> decimator -> fft -> psd -> noiseLearner -> transmission).
// The first generation of this engine ran them as three dependent launches per batch — FFT+dB, detect, emit — and each
// launch has its own fill and drain: the FFT kernel waits ~4 us for its first frames and leaves the vector pipe idle
// meanwhile, the detect kernel's tiles all load and then all compute, the emit kernel is one latency chain per frame, and
// every dependent launch boundary costs 1.5-2 us. What depends on what, across CALLS, is:
//
//     FFT+dB(k)  ->  detect(k)  ->  emit(k)            and            detect(k-1) -> detect(k)   (the averager ring)
//
// so FFT+dB(k), detect(k-1) and emit(k-2) are independent of each other. k_scan_step carries all three as ROLES of one
// launch: every workgroup takes one work item — one frame through the FFT, two 16-frame x 256-bin detect tiles of an earlier
// call, or the candidate lists of eight frames of a still earlier one — and whatever one role leaves idle (the FFT role's
// wait for HBM, the detect role's dependence on L2 latency) the others use. Between STAGES every dependency is a launch
// boundary. Inside a launch there is one hand-over since round 3 — the tile-culling plan of the detect stage that rides on the
// launch, from a few plan workgroups to the workgroups that evaluate the listed tiles — and it is bounded: a consumer that has
// polled StepArgs::wait_limit times makes the plan of its list itself (detect_fused.h), so no workgroup waits for another
// without bound and nothing can deadlock whatever the dispatch order. The host side (specscan.hip)
// keeps the deferred stages' arguments and drains them — launches with the finished roles empty — whenever a result is
// asked for (ss_sync, ss_flush, the host-buffer entry points, retunes and resets): results are exactly those of the
// three-launch chain, bit for bit, because every role runs the same code on the same data.
//
// Which calls share a launch is the host's choice. In order on one stream: FFT(k), detect(k-1), emit(k-2) — launch k + 1
// depends on launch k, so each launch runs its ramp and its tail alone. Deep pipelining (8192 points, ss_ctx::deep):
// FFT(L), detect(L-2), emit(L-4), launches alternating over two hardware queues; the rows detect(L) needs from before its
// batch — which detect(L-1) would write to the ring one launch earlier — launch L's FFT role produces itself by
// transforming the previous call's last frames once more (StepArgs::halo_*). Adjacent launches then have nothing to do
// with each other and fill each other's ramps and tails.
//
// Transforms of 16384 points and more are four-step (fft256_kernels.h): their column half — tiles of 32 columns x 256 rows,
// one per 512-thread workgroup — takes the FFT role's place (KIND 1, 2), the row half follows as a launch of its own, and
// the deferred stages ride on the column launch in the same way. 65536 points (what ships since round 4): a detect-mode call of up
// to 128 frames is ONE launch (KIND 7) — the column tiles of call k (the FFT role) and, dispatched behind them, the row tiles of
// call k - 1 (ROLE_ROWS, from the other of two work buffers), with the PLAN of call k - 2 (which of its averaging tiles can hold a
// candidate: ROLE_PLAN, two blocks of k_plan_long's numbering per workgroup), detect(k - 3) on the tiles its plan listed and
// emit(k - 4) ahead of both: two rounds of workgroups whose phases overlap. Longer calls and calls that keep a plane take two
// launches: the column launch of call k with the plan of call k - 1, detect(k - 2) and emit(k - 3), and the row half as the FFT role
// of a launch of its own (KIND 6) that carries nothing else.
// 2^20 points: 1024 x 1024 in two passes (fft1024_kernels.h) — the column half is a kernel of its own (1024 threads; the plan of the
// call before runs in its first workgroups), the row half the FFT role here (KIND 4) with detect(k - 1) and emit(k - 2) riding on it.
//
// Workgroup = 512 threads, <= 64 VGPRs, 39 KiB of LDS (the FFT role's; two detect tiles need 35 KiB, eight emit lists
// 32 KiB): four workgroups per CU whatever their roles.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "detect_fused.h"
#include "fft1024_kernels.h"
#include "fft256_kernels.h"
#include "fft8192_v2.h"

// KIND 8 (65536 points, the radix-8 fold): workgroups per frame — 4: two residues each (one fold for both, 128 registers, four waves per
// SIMD); 8: one residue each (64 registers, eight waves per SIMD). Build-time so that the two can be timed against each other
// (scripts/build_ab.py).
#ifndef SS_DIF8_W
#define SS_DIF8_W 4
#endif
// ... and the fold of the two-residue form: 1 = a radix-8 butterfly per point (round 6, fft65536_dif8.h: dif8_front2_bfly — 912 vector
// instructions per thread and fold), 0 = round 5's accumulating loop over q (1280). Build-time for the same reason.
#ifndef SS_DIF8_BFLY
#define SS_DIF8_BFLY 1
#endif

namespace ss {

struct StepArgs {
  Fft8192Args fft;  // KIND 0: 8192-point frames
  // KIND 0, deep pipelining (specscan.hip): FFT role frames [0, n_halo) are the last n_halo frames of the PREVIOUS call,
  // transformed again into a buffer of this launch's own, so that no stage has to read what the launch before wrote
  const void* halo_iq;  // first of those frames
  float* halo_psd;      // [n_halo][8192]
  float* halo_segsum;   // [32][kHaloSegPitch]: the halo frames' per-column maxima (null: none wanted)
  int n_halo;
  Rows256Args rows256;  // KIND 6: 256-point ROW tiles of a long transform (fft256_kernels.h: fft_rows256_tile) — 65536 points with tile culling: the column half is a launch of its own right before, the plan of the call before at its front
  Rows1024Args rows;  // KIND 4: 1024-point ROW tiles of a 2^20-point frame (fft1024_kernels.h) — the column half, 1024 threads per tile, is a launch of its own right before
  ColsArgs cols;    // KIND 1, 2: 256-point column tiles of a long transform (fft256_kernels.h); KIND 3: 1024-point column tiles of a 2^20-point frame (fft1024_kernels.h)
  DetectArgs det;
  EmitArgs emit;
  int n_fft;   // frames (KIND 0, n_halo included) or column tiles (KIND 1) of the FFT role (0: role absent)
  int n_det;   // detect TILES (two per workgroup)
  // KIND 0, tile culling (detect_fused.h): the n_plan tiles of `det` are PLANNED by S = 32 / plan_cols plan workgroups,
  // dispatched first, each listing those tiles of its plan_cols tile columns that must be evaluated; consumer p serves list
  // p mod S, entries 2 (p div S) and the next. Consumers are the launch's FFT workgroups once their frame is done
  // (plan_by_fft: the host makes sure there are enough of them) or, in a launch without an FFT role, detect workgroups of
  // which those beyond their list leave at once.
  int n_plan;
  int plan_cols;
  int plan_by_fft;
  // plan_by_fft, and the first plan_first consumers are detect workgroups of their own, dispatched ahead of the FFT role (FFT
  // workgroup i is consumer plan_first + i): they wait for the plan at the START of the launch and evaluate their pair while the
  // frames stream — an FFT workgroup that finds a pair behind its frame lives twice as long as its neighbours.
  int plan_first;
  // KIND 1, 2, tile culling (k_plan_long): the launch's column workgroups take the pairs of the detect stage's list once their
  // tile is done, pair p to column workgroup p; detect workgroups of their own only for the pairs beyond (n_det counts those).
  // (Evaluated BEFORE the column tile the same pairs cost the launch 5 us more: the workgroups that find one finish last. And
  // without k_plan_long — every workgroup testing its own pair from the run maxima after its column tile: 360 values, five
  // barriers — the launch grew by the 7 us the plan launch takes: 65536 x 128: 66.0 against 65.3 us per call, 2^20 x 16: 178
  // against 174, profiles/r03/s41.)
  int list_by_fft;
  // ... but the first list_first pairs go to detect workgroups of their own, dispatched ahead of the FFT role: an FFT workgroup that
  // finds a pair behind its tile lives twice as long as the others and is the launch's tail — a row launch of 65536-point frames
  // took 30 us with 22 listed pairs riding on it and takes 17 alone (profiles/r04/s18_summary.txt). Detect workgroup i < list_first
  // serves pair i (and leaves at once when the list is shorter), FFT workgroup p pair list_first + p, detect workgroup
  // i >= list_first pair n_fft + i.
  int list_first;
  // KIND 5 — launches without an FFT role of a 2^20-point context (the drain of its deferred stages): the
  // detect workgroups share k_plan_long's list out in a loop, pair item, item + W, item + 2 W, ... — a workgroup per POSSIBLE pair
  // (4096 of them, of which a few dozen find one) cost the launch 17 us in dispatch alone
  int list_loop;
  // KIND 1, 2 — a long transform's PLAN as a role (k_plan_long's blocks, two to a workgroup: block 2 item + tid / 256): 65536 points
  // with tile culling, where the plan of call k - 1 rides at the front of the column launch of call k together with emit(k - 2), and
  // the detect stage it plans on the row launch right behind (KIND 6). 0: no such role.
  int n_plan_long;  // workgroups
  PlanLongDet plan_det;
  PlanLongArgs plan_long;
  // KIND 7 — 65536 points, ONE launch per call (what ships for detect-mode calls of up to 128 frames): the column tiles of call k (the FFT role)
  // beside the ROW tiles of call k - 1 (ROLE_ROWS: rows256, n_rows of them, reading the other of two work buffers), the plan of call
  // k - 2, detect(k - 3) and emit(k - 4). 0: no such role.
  int n_rows;
  // KIND 8 — 65536 points, int8 IQ, NO work buffer: the FFT role is TWO residues of a frame, r and r + 4 — the radix-8 fold in the load
  // stage of the 8192-point transform, once for both, then the transform twice (fft65536_dif8.h: dif8_front2; 128 registers, so this
  // instantiation runs four waves per SIMD) —, four workgroups per frame, n_fft = 4 x frames, item j -> dif8_item<4> — whose rows (dB
  // values in the fold's blocked order: `fft.psd` is the ring's buffer) and run maxima the detect
  // stage of two calls later reads; the plan of call k - 1, detect(k - 2) (PERM8 tiles) and emit(k - 3) ride on the launch as they ride
  // on the column launch of the four-step form (KIND 2). `fft` carries the transform's tables and the rows' place, `dif` the fold's.
  // KIND 9 — the same for 131072-point frames (what getFft picks at 20 MS/s): radix 16, residues r and r + 8 per workgroup, n_fft = 8 x frames.
  // KIND 11 — KIND 8 with ONE residue per workgroup (fft65536_dif8.h: dif8_front, 64 registers, eight waves per SIMD), n_fft = 8 x frames: tried
  // for short calls, whose 4 x frames two-residue workgroups leave most of the chip's 256 CUs idle for 18 us each (round 6) — no gain: a
  // workgroup folds the WHOLE frame for one residue as for two and lives as long (SS_DIF8_SINGLE_MAX of the diagnostics build).
  // KIND 12 — 262144 points, ONE launch per call (round 6): KIND 10's roles and the 1024-point ROW tiles of call k - 1 (ROLE_ROWS: `rows`, n_rows of
  // them, from the other of two work buffers) behind the column tiles of call k — KIND 7's shape with this size's row tile and plan.
  // KIND 10 — 262144 points (round 6): KIND 2 — 256-point column tiles as the FFT role, the plan of call k - 1, detect(k - 2), emit(k - 3) —
  // whose plan role is plan_x256_run (PlanLongArgs::layout 3); an instantiation of its own so that the others keep their registers.
  Dif8Front dif;
  int n_emit;  // frames of the emit role
  int emit_per_wg;  // 8: one wave per frame; 1 (KIND 2): rows of 2048 mask words and more, the eight waves share one frame
  // One workgroup per work item, dispatched in blockIdx order, four resident per CU. WHICH item a workgroup takes decides
  // what shares a CU when: `order` (device memory, one word per workgroup: role << 24 | item) is built by the host once
  // per launch shape. A launch with one role only passes null (items in blockIdx order). (Tried and dropped: the order as
  // a run-length list in the kernel arguments, decoded with scalar instructions — ~800 of them per wave: 40 against 35 us
  // per step.)
  const uint32_t* order;
  // how many times a consumer of a planned stage's lists polls for its list before it makes the plan itself (detect_fused.h):
  // a poll is a sleep of ~0.25 us and a load, the plan role publishes within a few microseconds of the launch's start
  int wait_limit;
  int prio_fft, prio_other;  // s_setprio of the roles' waves (0..3)
#ifdef SS_DIAG
  int hint_mode;      // timing ablations of the list hand-over (garbage results): 1 = FFT workgroups ignore the lists, 2 = they do not wait for their header word; 3 (a test, correct results) = the plan workgroups never publish; 4 = the radix-8 fold's workgroups skip their transforms
  long long* stamps;  // measurement builds only: {start, end (100 MHz wall clock), role << 32 | item, XCC_ID << 32 | HW_ID} per workgroup
#endif
};
// (Tried in round 2 and dropped in round 3: FFT workgroups that take several frames each, so that the FFT role holds only
// two of a CU's four slots. The step got no faster, and the frame loop cost the kernel registers.)

constexpr int kStepThreads = 512;
constexpr int kStepLdsBytes = kFft8192V2LdsBytes;
static_assert(2 * (16 * DetectTile<21, 21, 16, 256>::P * 4 + 64) <= kFft8192V2LdsBytes, "two detect tiles per workgroup");
static_assert((8 * kEmitList + 9) * 4 <= kFft8192V2LdsBytes, "eight emit lists per workgroup");
static_assert(kFft256ColsLdsBytes <= kFft8192V2LdsBytes && kFft1024ColsLdsBytes <= kFft8192V2LdsBytes && kFft1024RowsLdsBytes <= kFft8192V2LdsBytes && kFft256RowsPsdLdsBytes <= kFft8192V2LdsBytes, "a column / row tile");
static_assert((kPlanLdsFloats + 64) * 4 <= kFft8192V2LdsBytes, "a plan workgroup's staging area");
static_assert(2 * (kPlanFusedFloats + kPlanLongInts) * 4 <= kFft8192V2LdsBytes, "two blocks of a long transform's plan");
static_assert(kPlanDif8Floats <= kPlanFusedFloats && kPlanDif8Floats <= kPlanLongFloats, "a block of the fold's plan in a plan block's LDS");
__host__ __device__ inline int step_fft_wgs(const StepArgs& a) { return a.n_fft; }
__host__ __device__ inline int step_emit_wgs(const StepArgs& a) { return a.emit_per_wg == 1 ? a.n_emit : (a.n_emit + 7) / 8; }
__host__ __device__ inline int step_plan_wgs(const StepArgs& a) { return a.n_plan ? 32 / a.plan_cols : a.n_plan_long; }
// consumers a planned stage needs: per list, a pair for every two tiles it may hold
__host__ __device__ inline int step_plan_consumers(const StepArgs& a) { return step_plan_wgs(a) * ((a.plan_cols * (a.n_plan / 32) + 1) / 2); }
// (Tried for the launches without an FFT role — the drain at the end of a run of calls: 32 detect workgroups per list, each taking
// every 32nd pair of it, instead of one per possible pair of which most leave at once. The tiles that must be evaluated sit in
// the lists of a few tile columns, a hundred pairs and more each: the launch went from 24 to 40 us, profiles/r03/s38_timeline_k20.txt.)
__host__ __device__ inline int step_det_wgs(const StepArgs& a) { return (a.n_det + 1) / 2 + (a.plan_by_fft ? a.plan_first : step_plan_consumers(a)); }
inline int step_items(const StepArgs& a) { return step_fft_wgs(a) + step_det_wgs(a) + step_emit_wgs(a) + step_plan_wgs(a) + a.n_rows; }

enum { ROLE_NONE = 0, ROLE_FFT = 1, ROLE_DET = 2, ROLE_EMIT = 3, ROLE_PLAN = 4, ROLE_ROWS = 5 };

template <int FMT, bool SPEC, int TW, bool SWZ, int KIND>
__device__ __forceinline__ void step_run_item(const StepArgs& a, int role, int item, unsigned char* smem_raw, int tid) {
  // Every big piece of code below has ONE call site (the compiler would otherwise carry several copies of a 3000-instruction
  // tile evaluation and of the transform, and spill registers for them): the roles first decide WHICH tiles, if any, this
  // workgroup evaluates — the detect role from its item number or from the planned stage's list, the FFT role from the list
  // once its frame is done — and the evaluation follows at the bottom.
  int tile_a = -1, tile_b = -1;  // the tiles of threads 0..255 / 256..511 (workgroup-uniform; -1: none)
  int list_pair_no = -1;  // long transforms: the pair of k_plan_long's list this workgroup evaluates (-1: it does not read the list)
  // 8192 points, planned stage: which consumer of the plan's lists this workgroup is (-1: none) and its list's header word as far
  // as it is known; the plan this workgroup makes (-1: none) — its own as a plan workgroup, or its list's when it has waited for
  // the plan workgroup in vain (detect_fused.h: nothing in a launch waits without bound)
  int consumer = -1, word = 0, plan_seg = -1;
  if constexpr (KIND == 7) {  // (only this instantiation knows the role: the others keep their registers)
    if (role == ROLE_ROWS) {
      fft_rows256_tile(a.rows256, item, smem_raw, tid);  // the row half of the call before: its column half ran in the launch before
      return;
    }
  }
  if constexpr (KIND == 12) {  // 262144 points, one launch per call: the same with the 1024-point row tile (StepArgs::rows)
    if (role == ROLE_ROWS) {
      fft_rows1024_tile<8>(a.rows, item, smem_raw, tid);
      return;
    }
  }
  if (role == ROLE_EMIT) {
    if constexpr (KIND >= 2) {
      // ---- emit role, long rows: the eight waves share one frame ----
      cand_emit_frame_wide<8>(a.emit, item, tid, reinterpret_cast<int*>(smem_raw));
    } else {
      // ---- emit role: one wave per frame ----
      const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
      const int f = item * 8 + w;
      if (f < a.n_emit) cand_emit_frame(a.emit, f, tid & 63, reinterpret_cast<int*>(smem_raw) + w * kEmitList);
    }
  } else if (role == ROLE_DET) {
    // ---- detect role: two tiles ----
    if constexpr (KIND == 5) {  // (an instantiation of its own: a loop around the tile evaluation costs every role of a kernel registers)
      if (a.det.tile_list && a.list_loop) {
        using T = DetectTile<21, 21, 16, 256>;
        const int n_tiles = (a.det.n / 256) * plan_frame_tiles(a.det.nframes, a.det.shift);
        const int cnt = min(a.det.tile_list[0], n_tiles), stride = (a.n_det + 1) / 2;
        const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
        float* tile = reinterpret_cast<float*>(smem_raw) + half * (16 * T::P + 16);
        int* tcnt = reinterpret_cast<int*>(tile + 16 * T::P);
        for (int p = item; 2 * p < cnt; p += stride) {
          int ta = a.det.tile_list[1 + 2 * p], tb = 2 * p + 1 < cnt ? a.det.tile_list[2 + 2 * p] : -1;
          if ((unsigned)ta >= (unsigned)n_tiles) ta = tb = -1;  // (a tile number that is none must not become an address)
          if ((unsigned)tb >= (unsigned)n_tiles) tb = -1;
          if (ta >= 0) {
            const int mine = half ? tb : ta;
            detect_tile<21, 21, 16, 256, SPEC>(a.det, mine < 0 ? ta : mine, tid & 255, tile, tcnt, mine >= 0);
            __syncthreads();  // the tile's LDS is free for the next pair
          }
        }
        return;
      }
    }
    if (KIND == 0 && a.n_plan) {  // the planned stage's lists (a launch without an FFT role)
      consumer = item;
    } else if (KIND >= 1 && a.det.tile_list) {
      // long transforms: the tiles k_plan_long listed. With an FFT role in the launch its workgroups take pairs 0 .. n_fft - 1
      // (below) and the detect workgroups the pairs beyond (calls that are no multiple of 16 frames have a few); without one,
      // a workgroup per possible pair. Workgroups beyond the list's end leave at once.
      list_pair_no = (a.list_by_fft && item >= a.list_first) ? item + a.n_fft : item;
    } else {
      tile_a = 2 * item;
      tile_b = 2 * item + 1 < a.n_det ? 2 * item + 1 : -1;
    }
  } else if (role == ROLE_PLAN) {
    if constexpr (KIND == 1 || KIND == 2 || KIND == 7 || KIND == 8 || KIND == 9 || KIND == 10 || KIND == 11 || KIND == 12) {  // a long transform's plan: two blocks of k_plan_long's numbering
      const int sub = tid >> 8;
      float* mrow = reinterpret_cast<float*>(smem_raw) + sub * (kPlanFusedFloats + kPlanLongInts);
      if constexpr (KIND == 8 || KIND == 9 || KIND == 11) plan_dif8_run<21, 21, 16, 256>(a.plan_det, a.plan_long, 2 * item + sub, tid & 255, mrow, reinterpret_cast<int*>(mrow + kPlanFusedFloats));  // (the fold's rows: layout 2)
      else if constexpr (KIND == 10 || KIND == 12) plan_x256_run<21, 21, 16, 256>(a.plan_det, a.plan_long, 2 * item + sub, tid & 255, mrow, reinterpret_cast<int*>(mrow + kPlanFusedFloats));  // (262144 points: layout 3)
      else plan_long_run<21, 21, 16, 256>(a.plan_det, a.plan_long, 2 * item + sub, tid & 255, mrow, reinterpret_cast<int*>(mrow + kPlanFusedFloats));
      return;
    }
    if constexpr (KIND == 0) plan_seg = item;
#ifdef SS_DIAG
    if (a.hint_mode == 3) plan_seg = -1;  // test switch: the plan workgroups never publish anything — every consumer has to help itself
#endif
  } else if constexpr (KIND >= 1) {
    // ---- FFT role, long transforms: one tile of 32 columns x 256 rows (KIND 3: 8 columns x 1024 rows of a 2^20-point frame) ----
    if constexpr (KIND == 5) return;  // (the drain of a 2^20-point context: launches without an FFT role only)
    else if constexpr (KIND == 8) {  // a residue of a 65536-point frame: fold + 8192-point transform + dB -> the ring's rows
      if constexpr (FMT != FMT_CF32) {
        int f, r, hdr;
        dif8_item<SS_DIF8_W>(item, a.dif.nframes, &f, &r);  // (W = 4: residues r and r + 4 by this workgroup)
#ifdef SS_DIAG
        if (a.hint_mode != 4)  // (timing ablation, garbage results: the passengers of the launch by themselves)
#endif
        fft8192_v2_frame<FMT, 2, true, false, SS_DIF8_W == 4 ? (SS_DIF8_BFLY ? 5 : 3) : 2>(a.fft, (size_t)(8 * (f - a.dif.first_hist) + r), smem_raw, tid, &hdr, &a.dif, (size_t)f, r);
      }
    }
    else if constexpr (KIND == 11) {  // a 65536-point frame's residue r alone: eight workgroups per frame (short calls, round 6)
      if constexpr (FMT != FMT_CF32) {
        int f, r, hdr;
        dif8_item<8>(item, a.dif.nframes, &f, &r);
        fft8192_v2_frame<FMT, 2, true, false, 2>(a.fft, (size_t)(8 * (f - a.dif.first_hist) + r), smem_raw, tid, &hdr, &a.dif, (size_t)f, r);
      }
    }
    else if constexpr (KIND == 9) {  // 131072 points, radix 16: residues r (< 8) and r + 8 of a frame, eight workgroups per frame
      if constexpr (FMT != FMT_CF32) {
        int f, r, hdr;
        dif8_item<8>(item, a.dif.nframes, &f, &r);
        fft8192_v2_frame<FMT, 2, true, false, SS_DIF8_BFLY ? 6 : 4>(a.fft, (size_t)(16 * (f - a.dif.first_hist) + r), smem_raw, tid, &hdr, &a.dif, (size_t)f, r);
      }
    }
    else if constexpr (KIND == 6) fft_rows256_tile(a.rows256, item, smem_raw, tid);  // (the ROW half of call k: its column half ran as its own launch right before)
    else if constexpr (KIND == 4) fft_rows1024_tile(a.rows, item, smem_raw, tid);  // (the ROW half of call k: its column half ran as its own launch right before)
    else if constexpr (KIND == 3) fft_cols1024_tile<FMT>(a.cols, item, smem_raw, tid);
    else fft_cols256_tile<FMT>(a.cols, item, smem_raw, tid);
    // ... then this workgroup's share of the tiles k_plan_long listed for the detect stage that rides on the launch. (The other
    // way round — the pairs first, while the memory system is still idle, then the column tile — was 5 us slower per launch at
    // 65536 points x 128 frames: the two dozen workgroups that find a pair then finish their column tile last.)
    if (a.list_by_fft) list_pair_no = a.list_first + item;
  } else {
    // ---- FFT role: one frame ----
    int hdr;
    const bool halo = item < a.n_halo;  // (workgroup-uniform)
    Fft8192Args g = a.fft;
    g.iq = halo ? a.halo_iq : a.fft.iq;
    g.psd = halo ? a.halo_psd : a.fft.psd;
    g.segsum = halo ? a.halo_segsum : a.fft.segsum;  // (the halo frames leave their maxima too: the tiles that read their rows are tested like the others)
    g.seg_pitch = halo ? kHaloSegPitch : a.fft.seg_pitch;
    g.live_hint = a.plan_by_fft ? a.det.live + live_count_word(a.plan_first + item, step_plan_wgs(a)) : nullptr;  // the list this workgroup serves for the detect stage that rides on the launch
#ifdef SS_DIAG
    if (a.hint_mode == 1) g.live_hint = nullptr;
    g.hint_nowait = a.hint_mode == 2;
#endif
    fft8192_v2_frame<FMT, TW, SWZ>(g, (size_t)(halo ? item : item - a.n_halo), smem_raw, tid, &hdr);
#ifdef SS_DIAG
    if (a.hint_mode == 1 || a.hint_mode == 2) hdr = kLiveReady;  // "complete, empty"
#endif
    if (a.plan_by_fft) {
      consumer = a.plan_first + item;
      word = hdr;
    }
  }
  if constexpr (KIND == 0) {
    // ---- the planned stage's lists: consumer `consumer` serves list consumer mod nseg, entries 2 (consumer div nseg) and the next ----
    // Everything that decides a branch here is workgroup-uniform: the header word came out of LDS (or is still unknown for the
    // whole workgroup), and where somebody has to wait the first wave waits — a bounded number of polls — and tells the others
    // through LDS (the role's own use of it is over). The common case — the count was there when the FFT role asked, and this
    // consumer's share of the list is empty — touches no barrier at all.
    int* note = reinterpret_cast<int*>(smem_raw);
    const int nseg = step_plan_wgs(a);
    bool helped_itself = false;
    if (consumer >= 0) {
      bool ok = true;
      if (!(word & kLiveReady)) {
        __syncthreads();
        if (tid < 64) {
          const int w = live_wait_count(a.det, consumer, nseg, word, a.wait_limit);
          if (tid == 0) note[0] = w;
        }
        __syncthreads();
        word = __builtin_amdgcn_readfirstlane(note[0]);
        ok = (word & kLiveReady) != 0;
      }
      if (ok && 2 * (consumer / nseg) < (word & (kLiveReady - 1))) {  // there are entries to fetch: is the list written?
        __syncthreads();
        if (tid < 64) {
          const bool f = live_wait_list(a.det, consumer % nseg, a.wait_limit);
          if (tid == 0) note[1] = f ? 1 : 0;
        }
        __syncthreads();
        ok = __builtin_amdgcn_readfirstlane(note[1]) != 0;
      }
      if (!ok) {  // the plan workgroup of this list has not been heard of: make its plan here
        plan_seg = consumer % nseg;
        helped_itself = true;
        __syncthreads();
        if (tid == 0 && a.det.stats) atomicAdd(stat_word(a.det.stats, kStatWaitFallbacks), 1ull);
      }
    }
    if (plan_seg >= 0) plan_tiles<21, 21, 16, 256>(a.det, plan_seg, a.plan_cols, tid, reinterpret_cast<float*>(smem_raw), !helped_itself);
    if (consumer >= 0) {
      if (helped_itself) {
        // (plan_tiles has drained its stores and passed a barrier: count and entries are where a write-through load finds them)
        __syncthreads();
        word = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&a.det.live[live_count_word(consumer, nseg)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      }
      const int2 pr = live_pair(a.det, consumer, nseg, word & (kLiveReady - 1));
      tile_a = pr.x;
      tile_b = pr.y;
      if (tile_a >= 0) __syncthreads();  // the frame's (or the plan's) last LDS reads are done
    }
  }
  if constexpr (KIND >= 1) {
    if (list_pair_no >= 0) {
      // (a tile number that is none must not become an address; the slot behind an odd count is not an entry)
      // (count and entries asked for together, the entries at an always-legal place — the list holds n_tiles + 2 words: one round trip
      // to memory instead of two in a chain of five that a pair's workgroup spends most of its time waiting in)
      const int n_tiles = (a.det.n / 256) * plan_frame_tiles(a.det.nframes, a.det.shift);
      const int at = min(1 + 2 * list_pair_no, n_tiles);
      const int listed = a.det.tile_list[0], entry_a = a.det.tile_list[at], entry_b = a.det.tile_list[at + 1];
      const int cnt = min(listed, n_tiles);
      if (2 * list_pair_no < cnt) {
        tile_a = entry_a;
        tile_b = 2 * list_pair_no + 1 < cnt ? entry_b : -1;
        if ((unsigned)tile_a >= (unsigned)n_tiles) tile_a = tile_b = -1;
        if ((unsigned)tile_b >= (unsigned)n_tiles) tile_b = -1;
        if (tile_a >= 0 && role == ROLE_FFT) __syncthreads();  // the column tile's last LDS reads are done
      }
    }
  }
  if (tile_a >= 0) {
    // ---- tile evaluation: threads 0..255 and 256..511 one tile each ----
    using T = DetectTile<21, 21, 16, 256>;
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);
    float* tile = reinterpret_cast<float*>(smem_raw) + half * (16 * T::P + 16);
    int* cnt = reinterpret_cast<int*>(tile + 16 * T::P);
    const int mine = half ? tile_b : tile_a;
    detect_tile<21, 21, 16, 256, SPEC, (KIND == 8 || KIND == 11) ? 3 : KIND == 9 ? 4 : 0>(a.det, mine < 0 ? tile_a : mine, tid & 255, tile, cnt, mine >= 0);
  }
}

template <int FMT, bool SPEC, int TW = 2, bool SWZ = true, bool PRIO = false, int KIND = 0>
__global__ __launch_bounds__(kStepThreads, ((KIND == 8 && SS_DIF8_W == 4) || KIND == 9) ? 4 : 8) void k_scan_step(StepArgs a_by_value) {
  // The arguments are read where they are used, straight from the kernel-argument segment (scalar loads from constant
  // memory): taken from the by-value parameter they are all loaded at the top of the kernel and kept alive through every
  // role — hundreds of scalar registers spilled into vector registers the 64-VGPR budget does not have.
  const StepArgs& a = *(const StepArgs*)__builtin_amdgcn_kernarg_segment_ptr();
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x;
  int role, item;
  if (a.order) {
    const uint32_t w = a.order[blockIdx.x];
    role = (int)(w >> 24);
    item = (int)(w & 0xffffffu);
  } else {
    // no table: the roles one after the other — plan, emit, detect, FFT. (The table's dependent scalar load stands between a
    // workgroup's start and its first frame load; the steady-state launch, whose detect work rides on its FFT workgroups,
    // has nothing to interleave and does without.)
    int b = (int)blockIdx.x;
    const int np = step_plan_wgs(a), ne = step_emit_wgs(a), nd = step_det_wgs(a);
#ifndef SS_ORDER_FFT_FIRST  // (A/B builds, scripts/build_ab.py: 1 = plan, FFT, emit, detect, rows — the frames ahead of the candidate lists)
#define SS_ORDER_FFT_FIRST 0
#endif
    if (SS_ORDER_FFT_FIRST && KIND == 0) {
      if (b < np) {
        role = ROLE_PLAN;
      } else if ((b -= np) < a.n_fft) {
        role = ROLE_FFT;
      } else if ((b -= a.n_fft) < ne) {
        role = ROLE_EMIT;
      } else if ((b -= ne) < nd) {
        role = ROLE_DET;
      } else {
        role = ROLE_ROWS;
        b -= nd;
      }
    } else if (b < np) {
      role = ROLE_PLAN;
    } else if ((b -= np) < ne) {
      role = ROLE_EMIT;
    } else if ((b -= ne) < nd) {
      role = ROLE_DET;
    } else if ((b -= nd) < a.n_rows) {
      role = ROLE_ROWS;
    } else {
      role = ROLE_FFT;
      b -= a.n_rows;
    }
    item = b;
  }
  if constexpr (PRIO) {  // (s_setprio takes an immediate)
    const int p = role == ROLE_FFT ? a.prio_fft : a.prio_other;
    if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else if (p == 3) __builtin_amdgcn_s_setprio(3);
  }
#ifdef SS_DIAG
  long long t_start = 0;
  if (a.stamps && tid == 0) t_start = wall_clock64();
#endif
  step_run_item<FMT, SPEC, TW, SWZ, KIND>(a, role, item, smem_raw, tid);
#ifdef SS_DIAG
  if (a.stamps && tid == 0) {
    unsigned hw_id = 0, xcc_id = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    long long* d = a.stamps + 4 * (size_t)blockIdx.x;
    d[0] = t_start;
    d[1] = wall_clock64();
    d[2] = ((long long)role << 32) | (unsigned)item;
    d[3] = ((long long)xcc_id << 32) | hw_id;
  }
#endif
}

}  // namespace ss
