// specscan.hip — host side of libspecscan.so: the C ABI of include/specscan.h over the gfx950 kernels
// in fft_kernels.h / detect_kernels.h. HIP runtime only; no torch, no CPU compute fallback: every entry
// point that needs the GPU fails with SS_ERR_NO_DEVICE / SS_ERR_HIP when there is none.
//
// One ss_ctx = one scan chain of the reference (the blocks SdrDevice::setupChains wires after the
// Blocker, sources/radio/sdr_device.cpp:161-168) pinned to one HIP device and one stream.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/specscan.h"
#include "detect_fused.h"
#include "detect_kernels.h"
#include "fft1024_kernels.h"
#include "fft256_kernels.h"
#include "reference_nan.h"
#include "fft_kernels.h"
#include "ring_place.h"
#include "scan_step.h"

namespace {

thread_local char g_create_err[512] = "";

// fused back end (grouping 21 x 21): frames per tile, and the ring rows kept between batches
constexpr int kFusedTF = 16;
// Deep pipelining (ss_ctx::deep): a call's last stage runs four launches after its first; the two queues synchronise once in
// kDeepSyncPeriod launches on the launch three back (a wait costs ~10 us of queue time: rarely, and not within the first 32
// launches of a run): everything launched more than 4 + kDeepSyncPeriod + 3 launches ago has finished.
constexpr int kDeepSpecSlots = 64;  // calls whose spectrogram partial sums may wait to be added to their container
constexpr int kDeepFoldBatch = 8;   // ... added in batches of this many calls
constexpr int kDeepSyncPeriod = 64;
constexpr int kDeepSyncPhase = 32;
constexpr int kMaxQueues = 4;  // launch queues of the deep pipelining (ss_ctx::nq)
constexpr int kDeepHorizon = 2 * kMaxQueues + kDeepSyncPeriod + 2 * kMaxQueues + 1;
constexpr int kHistRows = ss::DetectTile<21, 21, kFusedTF, 256>::H;  // 35

struct SpecState {  // Spectrogram::Container, sources/radio/blocks/spectrogram.h:10-16, one per centre frequency
  int32_t center = 0;
  float* d_sum = nullptr;
  int count = 0;
};

struct NoiseState {  // NoiseLearner::Noise, sources/radio/blocks/noise_learner.h:11-20, one per centre frequency
  int32_t center = 0;
  float* d_thr = nullptr;
  int samples = 0;
  bool ready = false;
  int64_t start_ms = 0;
  bool have_start = false;
};

// SS_FLAG_REFERENCE_NAN (reference_nan.h): what the poison stage between a call's detect and emit stages works on
struct NanStage {
  const float* psd = nullptr;
  int nframes = 0, n_learn = 0, pushed_before = 0;
  uint32_t* maskbits = nullptr;
  int* counts = nullptr;
  float* avg_full = nullptr;
};

}  // namespace

struct ss_ctx {
  ss_config cfg{};
  int n = 0, logn = 0;
  float db_off = 0.0f;  // 10*log10(fs), the constant term of PSD::work (psd.cpp:19)
  int32_t range_lo = 0, range_hi = 0;
  std::vector<int32_t> ignored;
  hipStream_t stream = nullptr;  // every kernel and copy of this chain is ordered on this stream
  unsigned long long batch_no = 0;
  // constants
  float* d_win = nullptr;
  float2* d_tw = nullptr;
  float2* d_tw8k = nullptr;   // 8192 points: tw2[256] ++ tw3a[1024] ++ tw3b[2048] (first-generation tables, SS_DIAG A/B runs)
  float2* d_tw8v2 = nullptr;  // 8192 points: tw2[256] ++ lane[384] ++ wave[96] (fft8192_v2.h)
  bool use_fft8192 = false;
  // Implementation choices. The shipped library fixes them here; a build with -DSS_DIAG (libspecscan_diag.so, used by
  // the A/B tests and the measurement scripts only) lets the environment override them once, at ss_create.
  struct Diag {
    bool backend_unfused = false;  // per-stage back-end kernels also for the 21 x 21 grouping
    bool fft_generic = false;      // radix-4 LDS kernels instead of the register-pass ones
    int fft_rows_r = -1;           // 0 never / 1 always use k_fft_rows256xR_psd for N2 = 512..4096 (-1: up to 2048)
    int fft_sub = -1;              // 0 never / 1 always split N2 > 256 rows into radix-A step + 256-point rows (-1: from 2048)
    bool fft_twopass = true;       // 2^20 points as 1024 x 1024 in two passes (SS_FFT_TWOPASS=0: 256 x 4096 in three, as until round 3)
    bool ring_only = true;         // 2^20 points, detect mode, calls shorter than the ring: no dB plane is written (SS_RING_ONLY=0: written as ever)
    bool cols1024_wide = true;     // 2^20 points: column tiles of 16 columns by 1024 threads as a launch of their own, the deferred stages in a launch without an FFT role (SS_C1024_WIDE=0: 8 columns by 512 threads as k_scan_step's FFT role, KIND 3)
    bool fft_xcd_map = true;       // XCD-aware tile order in k_fft_rows256xR_psd
    bool spec_standalone = false;  // spectrogram by its own two kernels instead of inside the detect tiles
    bool pipeline = true;          // 8192 points: defer detect / emit of a call into the next calls' launches (scan_step.h)
    int fft_tw = 2;                // 8192 points: where the twiddles come from (fft8192_v2.h: 0 global, 1 pass-2 table in LDS, 2 LDS + SGPRs)
    bool fft_swz = true;           // 8192 points: 16-byte swizzled first exchange
    int prio_fft = 0, prio_other = 0;  // s_setprio of k_scan_step's roles
    std::string stamp_path;        // SS_DIAG only: dump per-workgroup start / end stamps of the 40th full k_scan_step launch here
    long long* d_stamps = nullptr;
    int stamp_launches = 0;
    unsigned* d_cull_stats = nullptr;  // SS_DIAG only (SS_CULL_STATS=1): tiles seen / on the culling path / culled, printed by ss_destroy
    // SS_DIAG only, deep pipelining: the input canary. The launch of call L + 1 reads the last frames of call L's input once
    // more (the halo, run_call_deep); include/specscan.h tells callers to leave every input alone until ss_sync. A checksum of
    // those frames is taken right behind launch L (the public stream waits for it: whatever the caller enqueues there after the
    // call comes later) and again right before launch L + 1; ss_sync compares the two and fails loudly when a caller has
    // refilled the buffer in between — a case the address check of run_call_deep cannot see (the same bytes under another pointer).
    // Opt-in (SS_CANARY=1): the wait puts the public stream behind every launch, which is not what one wants to time.
    bool canary = false;
    unsigned long long* d_canary = nullptr;  // [64][2]
    hipEvent_t ev_canary = nullptr;
    std::vector<int> canary_pairs;           // slots whose two sums have both been enqueued
    unsigned canary_seq = 0;
    int canary_prev_slot = -1;               // the slot that holds the first sum of the previous call's frames (-1: none taken)
    bool emit_wide = true;         // long rows (n >= 16384): several waves per frame in the emit stage
    bool no_order_table = false;   // SS_DIAG: never use a dispatch-order table
    int hint_mode = 0;             // SS_DIAG timing ablations of the list hand-over (scan_step.h); 3: plan workgroups never publish (tests/test_gpu_wait_bound.py)
    int wait_limit = 0;            // SS_DIAG (SS_WAIT_LIMIT): StepArgs::wait_limit, 0 = the product's
    int queues = 2;                // 8192 points, deep pipelining: launch queues (2 .. 4)
    bool cull_65536 = true;        // tile culling at 65536 points (SS_CULL_65536=0: every averaging tile evaluated, as the product did until session 17 of round 4; see ss_create)
    bool rows256_step = true;      // 65536 points with tile culling: the column half as a launch of its own (the plan of the call before at its front), the ROW tiles as k_scan_step's FFT role (KIND 6) with the deferred stages riding on them (SS_ROWS256_STEP=0: columns as the FFT role, rows and plan as launches of their own)
    bool win_calc = true;          // 2^20 points in two passes and 65536 points, default window: the column tiles form their Hamming taps instead of loading them (SS_WIN_CALC=0: the table as ever)
    bool emit_on_rows = false;     // SS_DIAG (SS_EMIT_ON_ROWS=1): 65536 points, the emit stage on the row launch instead of the column launch (A/B)
    int merge_max_frames = 128;    // ... calls of up to this many frames (SS_MERGE_MAX)
    bool merge_65536 = true;       // 65536 points, detect-mode calls: ONE launch per call — the column half of call k beside the row half of call k - 1 (scan_step.h KIND 7; SS_MERGE_65536=0: two launches per call, session 20's form)
    bool det_lag2 = true;          // 65536 points with tile culling: detect(k - 2) on the column launch of call k (SS_DET_LAG2=0: detect(k - 1) on the row launch, session 19's form)
    bool dif8 = true;              // 65536 points, int8 IQ, default window, calls that keep no plane: NO work buffer — the radix-8 fold in the load stage of the 8192-point transform, eight workgroups per frame, one launch per call whatever its length (scan_step.h KIND 8, fft65536_dif8.h; SS_DIF8=0: the four-step forms of round 4)
    int plan_first = 0;            // 8192 points: the first plan_first pairs of every list of the launch's tile plan on detect workgroups of their own ahead of the FFT role (SS_PLAN_FIRST=n; 0: every pair behind an FFT workgroup's frame)
    bool rows1024x256 = true;      // 262144 points (what getFft picks at 61.44 MS/s): rows through the 1024-point row tile with run maxima and ring rows — culled, no dB plane in detect mode, the 65536-point two-launch pipeline (SS_ROWS1024X256=0: round 2's path, k_fft_rows256xR_psd, every tile evaluated)
    long long abs_start = 0;       // SS_ABS_START=n: the frame counter starts at n instead of 0 at creation and after ss_reset — sessions that cross 2^30 frames (the counter's 30-bit form indexes the long transforms' run maxima) without the twenty seconds of scanning it takes to get there (tests/test_gpu_cull.py)
    int drain_waiter_us = 0;       // 8192 points, deep pipelining (SS_DRAIN_WAITER_US=n): the drain's join behind a waiter on the public stream that sleeps at most n us (drain_deep) — measured in round 6 and not kept: 25.3-26.5 against 25.7-26.5 us per step with it, profiles/r06/s19_summary.txt
    bool drain_tail_event = true;  // ... an event recorded behind the drain's last command on the public stream (SS_DRAIN_TAIL_EVENT=0: none, as until session 18 of round 6): the 20-step form 25.2 against 25.85 us per step, medians of eight alternating runs (s20)
    int dif8_single_max = 0;       // 65536 points, the fold: calls of up to this many frames take ONE residue per workgroup (scan_step.h KIND 11; SS_DIF8_SINGLE_MAX=n) — measured in round 6 and not kept: a workgroup folds the whole frame whether it wants one residue of it or two, so it lives as long either way (16-frame calls 20.6 against 19.5 us, 32-frame calls 26.0 against 20.3: profiles/r06/s13_summary.txt)
    int chunk_65536 = 256;         // 65536 points with tile culling: calls of more frames go through in chunks of this many (SS_CHUNK_65536=0: in one piece)
    int chunk_long = 16;           // 2^20 points in two passes: calls of more frames go through in chunks of this many (SS_CHUNK_LONG=0: in one piece)
    bool halo_maxima = true;       // 8192 points, deep pipelining: the re-transformed halo frames leave per-column maxima, so that the tiles of a batch's first two frame tiles are tested like the others (SS_HALO_MAXIMA=0: evaluated whatever they hold, as until session 36 of round 5)
    int list_first_fold = -1;      // ... in the fold's launches: -1 = by the launch's size (launch_step), k > 0 = k - 1 pairs
    int list_first = 64;           // long transforms with tile culling: the first pairs of the plan's list go to detect workgroups of their own, dispatched ahead of the launch's FFT role (SS_LIST_FIRST=0: every pair behind an FFT workgroup's tile, as until session 19 of round 4)
    bool plan_fused = true;        // 2^20 points in two passes: the plan of call k at the front of call k + 1's column launch (SS_PLAN_FUSED=0: a launch of its own behind call k's rows, as until session 14 of round 4)
    int ablate_roles = 0;          // SS_DIAG timing ablation (garbage results): 1 = launches carry no detect role, 2 = no emit role
    bool cull = true;              // 8192 points: detect tiles that cannot hold a candidate are not evaluated (detect_fused.h)
    bool deep = true;              // 8192 points: consecutive step launches independent of each other, alternating over two queues (see ss_ctx::deep)
    bool step_long = true;         // n >= 16384: the column half of the FFT as the FFT role of k_scan_step (false: one launch per stage)
    // Dispatch order of k_scan_step's work items when all three roles ride one launch: "prefix|cycle", comma-separated
    // segments of a role letter (E emit, D detect, F FFT) and a workgroup count ('*' = all that are left); the cycle repeats
    // until every item is placed, a role that has run out is skipped.
    std::string step_order = "E*|D128,F1024";
    // n >= 16384 (the FFT role is the column half, short workgroups): detect first. 65536 x 128 frames: 63.3 us per step
    // against 64.7 with the order above and 70.7 with one launch per stage; 2^20 x 16: 216.8 / 235 / 216.9 (profiles/r02/s18).
    std::string step_order_long = "E*|D*,F*";
    // one launch per call (KIND 7): the column tiles of this call first, the row tiles of the call before behind them — two rounds of
    // workgroups whose phases overlap. In turn (R1,F1) they took 45.7 us per 128-frame call against 41.0-41.7 in two runs, 26.2 against
    // 24.0 per 64-frame call (rows first: 41.0 / 26.7): profiles/r04/s35_summary.txt
    std::string step_order_merged = "E*|D*,F*,R*";
    // ... of the radix-8 fold's launch (KIND 8; the plan's few dozen workgroups always first): the fold workgroups, then the candidate
    // lists, then the listed tiles' detect workgroups. A detect workgroup lives 10-20 us, and every one dispatched ahead of the fold
    // workgroups holds up one of the launch's last round by as much: detect first (the long transforms' order above) 52.8 us per
    // 128-frame call, this order 40.3; passengers spread between the fold workgroups (F8,E2,D1 and the like) 47-52
    // (profiles/r05/s5_summary.txt, s6_summary.txt); emit ahead of the fold workgroups: 71.7 / 129.5 us per 256- / 512-frame call against
    // 69.8 / 126.5 (s9_summary.txt)
    std::string step_order_fold = "F*,P*,E*,D*";
#ifdef SS_DIAG
    void read() {
      const auto is = [](const char* name, const char* value) {
        const char* v = getenv(name);
        return v && strcmp(v, value) == 0;
      };
      const auto tri = [](const char* name) {
        const char* v = getenv(name);
        return !v ? -1 : (v[0] == '1' ? 1 : 0);
      };
      const auto num = [](const char* name, int dflt) {
        const char* v = getenv(name);
        return v ? atoi(v) : dflt;
      };
      backend_unfused = is("SS_BACKEND", "unfused");
      fft_generic = is("SS_FFT_IMPL", "generic");
      fft_rows_r = tri("SS_FFT_ROWSR");
      fft_sub = tri("SS_FFT_SUB");
      fft_twopass = tri("SS_FFT_TWOPASS") != 0;
      ring_only = tri("SS_RING_ONLY") != 0;
      cols1024_wide = tri("SS_C1024_WIDE") != 0;
      fft_xcd_map = tri("SS_FFT_XCDMAP") != 0;
      spec_standalone = is("SS_SPEC_IMPL", "standalone");
      pipeline = tri("SS_PIPELINE") != 0;
      fft_tw = num("SS_FFT_TW", 2);
      fft_swz = tri("SS_FFT_SWZ") != 0;
      prio_fft = num("SS_STEP_PRIO_FFT", 0);
      prio_other = num("SS_STEP_PRIO_OTHER", 0);
      if (const char* v = getenv("SS_STEP_STAMPS")) stamp_path = v;
      emit_wide = tri("SS_EMIT_WIDE") != 0;
      step_long = tri("SS_STEP_LONG") != 0;
      deep = tri("SS_DEEP") != 0;
      cull = tri("SS_CULL") != 0;
      ablate_roles = num("SS_ABLATE_ROLES", 0);
      cull_65536 = tri("SS_CULL_65536") != 0;
      rows256_step = tri("SS_ROWS256_STEP") != 0;
      plan_fused = tri("SS_PLAN_FUSED") != 0;
      list_first = num("SS_LIST_FIRST", list_first);
      halo_maxima = tri("SS_HALO_MAXIMA") != 0;
      list_first_fold = getenv("SS_LIST_FIRST") ? list_first + 1 : num("SS_LIST_FIRST_FOLD", list_first_fold);
      chunk_long = num("SS_CHUNK_LONG", chunk_long);
      chunk_65536 = num("SS_CHUNK_65536", chunk_65536);
      rows1024x256 = tri("SS_ROWS1024X256") != 0;
      dif8_single_max = num("SS_DIF8_SINGLE_MAX", dif8_single_max);
      drain_waiter_us = num("SS_DRAIN_WAITER_US", drain_waiter_us);
      if (const char* v = getenv("SS_ABS_START")) abs_start = atoll(v);
      drain_tail_event = tri("SS_DRAIN_TAIL_EVENT") != 0;
      plan_first = num("SS_PLAN_FIRST", plan_first);
      det_lag2 = tri("SS_DET_LAG2") != 0;
      dif8 = tri("SS_DIF8") != 0;
      emit_on_rows = tri("SS_EMIT_ON_ROWS") == 1;
      merge_65536 = tri("SS_MERGE_65536") != 0;
      merge_max_frames = num("SS_MERGE_MAX", merge_max_frames);
      win_calc = tri("SS_WIN_CALC") != 0;
      canary = tri("SS_CANARY") == 1;
      queues = num("SS_QUEUES", queues);
      hint_mode = num("SS_HINT_MODE", 0);
      wait_limit = num("SS_WAIT_LIMIT", 0);
      no_order_table = tri("SS_ORDER_TABLE") == 0;
      if (tri("SS_PLAN_NOZERO") == 1) d_cull_stats = reinterpret_cast<unsigned*>(1);
      else if (tri("SS_CULL_STATS") == 1 && hipMalloc(&d_cull_stats, 3 * sizeof(unsigned)) == hipSuccess) (void)hipMemset(d_cull_stats, 0, 3 * sizeof(unsigned));
      if (const char* v = getenv("SS_STEP_ORDER")) step_order = step_order_long = step_order_fold = v;
      if (const char* v = getenv("SS_STEP_ORDER_MERGED")) step_order_merged = v;
    }
#else
    void read() {}
#endif
  } diag;
  uint8_t* d_pass = nullptr;
  bool pass_dirty = true;
  // state
  std::vector<NoiseState> noise;
  // spectrogram side branch (SS_FLAG_SPECTROGRAM)
  int spec_n = 0, spec_m = 0;  // output bins, input bins per output bin (spectrogram.cpp:14-15)
  bool spec_in_detect = false;  // accumulated by k_detect_fused instead of the two stand-alone kernels
  // in-detect form: the partial sums of a launch are folded into their container by the next launch (or by
  // spectrogram_flush); two buffers alternate
  float* d_spec_part2[2] = {nullptr, nullptr};
  int spec_cur = 0;
  // deep pipelining: detect stages of overlapping launches cannot add to a container one after the other, so every call's
  // partial sums keep a slot of their own until the next drain adds them, call by call and tile by tile — the same
  // additions in the same order as the in-detect form — on the public stream (drain_deep); a full ring drains.
  float* d_spec_ring = nullptr;
  size_t spec_slot_floats = 0;
  int spec_ring_next = 0;
  struct PendFold {
    const float* partial;
    int tiles;
    float* sum;
    int slot;
    long det_launch;  // the launch that carries (or will carry) this call's detect stage
  };
  std::deque<PendFold> deep_folds;
  // The additions run on a stream of their own (the public stream stays idle, so the queues need no fork): every
  // kDeepFoldBatch calls, behind one event from each queue, call by call in order. A slot is written again only after the
  // host has seen the event behind its additions complete.
  hipStream_t s_fold = nullptr;
  hipEvent_t ev_fold_src[kMaxQueues] = {}, ev_fold_done[16] = {};
  unsigned fold_batches = 0;
  int slot_batch[kDeepSpecSlots];  // the batch (index into ev_fold_done) that added the slot's sums, -1: nothing outstanding
  bool fold_dirty = false;         // s_fold holds work the public stream has not been made to wait for
  int fold_last = 0;               // the newest batch
  float* spec_pending_sum = nullptr;  // container (SpecState::d_sum) the partial sums in d_spec_part2[spec_cur ^ 1] belong to
  int spec_pending_tiles = 0;
  std::vector<SpecState> spec;
  float* d_spec_partial = nullptr;
  int frames_pushed = 0;  // Averager::m_frames, saturates at grouping_y
  int rot_frames = 0;     // rows of the previous batch still to be folded into the history rows (lazy ring rotation)
  // planes (frame-major rows of n floats)
  // back end, fused path (grouping 21 x 21, max_batch <= 65536): ring and counters are double-buffered
  bool fused = false;
  // Averager ring: the newest kHistRows rel rows (newest last) are a WINDOW [hist_start, hist_start + kHistRows) of a
  // longer buffer. A batch shorter than the ring appends its rows behind the window and the window slides (nothing is
  // copied until the buffer's end is reached); a longer batch writes a fresh window clear of the one it reads.
  float* d_hist = nullptr;
  int hist_rows = 0;   // capacity in rows
  int hist_start = 0;
  long long abs_frames = 0;               // frames since the last reset: frame tiles are aligned to this index
  // per-frame candidate counts, three buffers in rotation: batch k accumulates into [k % 3]; its emit stage — which may
  // run two launches later, next to the detect stage of batch k + 1 — reads them and zeroes [(k + 2) % 3] for batch k + 2
  // (deep pipelining, below: six buffers, the emit stage of batch k runs four launches after its FFT stage and zeroes
  // [(k + 4) % 6], whose last reader ran two launches earlier on the same queue)
  int* d_cnt3[3 * kMaxQueues] = {};
  int cnt_cur = 0;
  int cnt_frames[3 * kMaxQueues] = {};  // how many entries of each buffer may be non-zero
  int ncnt = 3, nbuf = 1, npsd = 1, lag = 1;  // buffers in rotation: counters, mask / avg planes / offsets, internal PSD planes; launches between a call's stages
  float* d_relplane = nullptr;            // full rel plane, only when a caller asks for it (lazy)
  int last_n_learn = 0;
  // Stage pipelining (8192 points, scan_step.h): the detect stage of the last call and the emit stage of the call before
  // it are kept as arguments until the next launch carries them (or flush_stages drains them). Buffers a deferred stage
  // reads while a later call already writes its own are doubled: mask bits, sparse avg plane, internal PSD plane.
  bool step_path = false;
  bool have_det = false, have_emit = false;
  // 2^20 points in two passes: the plan of the last call (which of its tiles the detect stage must evaluate, k_plan_long) is not a
  // launch of its own but rides at the front of the NEXT call's column launch (k_fft_cols1024_plan, detect_fused.h); until then
  // — or until flush_stages launches it on its own — its arguments wait here. pend_det.tile_list points at the list it fills.
  bool x256_tile = false;     // 262144 points: the row half through the 1024-point row tile (k_fft_rows1024_psd<8>) in every context, culled or not — the same bits either way
  bool rows1024x256 = false;  // 262144 points with tile culling (round 6): 256-point column tiles as k_scan_step's FFT role with the plan of the call before, detect(k - 2) and emit(k - 3); the rows as a launch of k_fft_rows1024_psd<8>
  bool rows256_step = false;  // 65536 points with tile culling: columns as their own launch, rows as k_scan_step's FFT role (see Diag::rows256_step)
  bool have_plan = false;
  ss::PlanLongDet pend_plan_det{};
  ss::PlanLongArgs pend_plan{};
  ss::DetectArgs pend_det{};
  int pend_det_tiles = 0;
  bool pend_det_spec = false;
  ss::EmitArgs pend_det_emit{};  // the emit stage that follows pend_det
  ss::EmitArgs pend_emit{};
  // 65536 points with tile culling (det_lag2): a call's detect stage rides on the COLUMN launch two calls later — the longer of a
  // call's two launches, where the quarter of a hundred workgroups that evaluate listed tiles (20 us apiece under the launch's
  // streaming loads) end with the column tiles instead of being the row launch's tail — because the plan it needs rides on the
  // column launch of the call in between. So two detect stages wait: pend_det (planned: rides on the next column launch) and
  // pend_det2 (its plan is have_plan / pend_plan).
  bool det_lag2 = false;
  // 65536 points, int8 IQ (fft65536_dif8.h): calls that keep no plane go through the radix-8 fold — one launch of 8 x frames
  // workgroups (4 x frames since two residues share one) that leaves dB rows in the fold's BLOCKED order (32 Q bins per block,
  // fft65536_dif8.h: dif_bin_offset; ring_db_rows) in the averager ring's buffer and run maxima in the plan's layout 2; every other call (learning frames, planes handed out, stream-ordered contexts) takes the four-step form, whose rows
  // are in bin order. ring_perm8 says which order the ring's window and the stages that wait are in; a call of the other kind
  // drains what waits and has the window's 35 rows rewritten first (set_ring_form).
  bool dif8 = false;
  int dif_logq = 3;               // log2 of the fold's radix: 3 (65536 points) or 4 (131072 points: sixteen residues, KIND 9)
  bool cull_fold_only = false;    // 131072 points: the culling machinery (run maxima, plan, ring rows by the FFT stage) serves the fold's calls only — the four-step form of that size has no such epilogue
  bool ring_perm8 = false;
  bool last_rows_perm8 = false;   // ... and the rows ss_read_window serves (last_hist, last_rel_rows)
  // Long transforms, detect-mode calls (ring-only): the FFT stage leaves its rows in the ring's buffer as dB values and the tiles that
  // are evaluated subtract the ceiling (DetectArgs::ring_db_from) — the ceiling loads cost such a launch 7-9 %. ring_db_rows: how
  // many of the window's newest rows are such rows, formed under the ceiling ring_db_thr; settle_ring_db subtracts it from them in
  // place before a call of another kind writes to the window, before the ceiling changes and before another centre frequency's is used.
  int ring_db_rows = 0;
  const float* ring_db_thr = nullptr;
  int last_db_from = 0;           // ss_read_window: rows of the last call from this batch-relative frame on (<= 0; frames >= 0 all) are dB values ... 
  bool last_rows_db = false;      // ... when the last call left such rows at all
  const float* last_settled_lo = nullptr;  // ... except the rows in [last_settled_lo, last_settled_hi): settle_ring_db has turned them into noise-relative
  const float* last_settled_hi = nullptr;  //     rows since (a retune or ss_reset_noise after the call) — ss_read_window must not subtract the ceiling twice
  float2* d_dif8_tab = nullptr;   // the fold's tables (dif8_host_tables)
  float* d_perm_tmp = nullptr;    // 35 rows: the window on its way from one order to the other
  bool have_det2 = false;
  ss::DetectArgs pend_det2{};
  int pend_det2_tiles = 0;
  bool pend_det2_spec = false;
  ss::EmitArgs pend_det2_emit{};
  ss::RingPrev hist_prev{0, -1, 0};  // what the detect stage of the last call reads of the ring's buffer (ring_place.h)
  ss::RingPrev hist_prev2{0, -1, 0}; // ... of the call before it (merged contexts protect two)
  // 65536 points, ONE launch per call (scan_step.h KIND 7; calls that keep no dB plane, up to 128 frames): the launch of call k carries
  // the column half of call k, the ROW half of call k - 1 (two work buffers), the plan of call k - 2, detect(k - 3) and emit(k - 4).
  // Waiting on the host besides pend_det / pend_det2 / pend_plan / pend_emit: the row stage of the last call, its plan (not ready
  // before that row stage has run) and its detect stage.
  bool merge = false;
  int merge_max = 128;  // the largest call (frames) that is one launch
  float2* d_work2 = nullptr;
  int work_cur = 0;
  bool have_rows = false;
  ss::Rows256Args pend_rows{};
  ss::Rows1024Args pend_rows1024{};  // (262144 points: the row stage that waits is one of 1024-point row tiles)
  int pend_rows_tiles = 0;
  bool have_plan2 = false;
  ss::PlanLongDet pend_plan2_det{};
  ss::PlanLongArgs pend_plan2{};
  bool have_det3 = false;
  ss::DetectArgs pend_det3{};
  int pend_det3_tiles = 0;
  bool pend_det3_spec = false;
  ss::EmitArgs pend_det3_emit{};
  int buf_cur = 0;               // which of the rotating buffers the NEXT batch writes
  int psd_cur = 0;
  // Deep pipelining (8192 points; diag.deep). With the stages of three consecutive calls in one
  // launch, launch k + 1 still depends on launch k (detect(k) needs FFT(k)), so every launch runs its ramp and its tail
  // alone — a quarter of a 1024-frame launch (scripts/ubench/launch_overlap_lab.hip). Independent launches on two
  // hardware queues fill each other's tails. Launch L therefore carries FFT(L), detect(L - 2) and emit(L - 4), and
  // launches alternate over two side streams, so that launch L follows launch L - 2 in stream order and never waits for
  // launch L - 1. What detect(L) needs from BEFORE its batch — the ring rows its predecessor writes, one launch earlier —
  // it gets without that predecessor: launch L's FFT role transforms the last 20 frames of call L - 1 (35 when the call
  // does not start on a tile boundary) once more — 2-3 % more FFT work; same kernel, same input: the same bits — into a
  // buffer of its own (d_halo), which detect(L) reads as DetectArgs::halo_psd. No launch reads anything an odd number of
  // launches back, so no events are needed between the two queues (an event record + wait per launch cost 3.7 us per
  // step, more than the overlap gains). The context's public stream only forks (ev_in, when it holds work) and joins
  // (flush_stages: the two queues drain their last stages side by side, then the public stream waits for both). Calls
  // that cannot overlap (learning frames, fewer frames than the ring holds, a caller reusing a buffer too soon) drain
  // first and run their three stages in order on the public stream.
  bool deep = false;
  // (Round 3: nq queues instead of two — launch L on queue L mod nq carries FFT(L), detect(L - nq), emit(L - 2 nq). With two,
  // a queue's next launch cannot start before its previous one has wholly finished, and while that one's last workgroups
  // drain, the OTHER queue's launch has long been dispatched in full: a third of the CU slots stand empty around every
  // hand-over. Everything said about "even" distances below reads "multiples of nq".)
  int nq = 2;
  hipStream_t s_ab[kMaxQueues] = {};
  hipEvent_t ev_launch[32] = {}, ev_in[4] = {}, ev_join[kMaxQueues] = {}, ev_tail = nullptr;
  unsigned* d_drain_done = nullptr;  // drain_deep: counts the queues' "my last stage has run" signals since ss_create (k_drain_signal), the waiter's word
  unsigned drain_signals = 0;        // ... how many have been enqueued
  float* d_halo[2 * kMaxQueues] = {};  // [kHistRows][n] each: written by launch L, read by detect(L) in launch L + nq
  const void* deep_prev_iq = nullptr;  // the previous call's frames (caller's buffer: untouched until ss_sync by contract)
  long long deep_prev_stride = 0;
  long deep_L = 0;             // launches since the last drain
  long deep_barrier = -10;     // a launch whose detect role read the ring: the next launch waits for it (only when the ring is short, see deep_ring_safe)
  // Detect stages write their batch's newest rows to the ring window after the current one, windows rotating through the
  // whole ring. A ring-reading detect stage — the first of an overlapped run, launch 2 — is finished for both queues by
  // the run's first synchronisation (launch kDeepSyncPhase): with more windows than that, the window it reads is not
  // written again while it runs, and nobody has to wait for it.
  bool deep_ring_safe = false;
  unsigned deep_forks = 0;
  struct PendDet {
    ss::DetectArgs a;
    int tiles;
    ss::EmitArgs emit;
    long ready;  // first launch that may carry it
    bool spec;   // with the spectrogram branch (k_scan_step<..., SPEC = true>)
  };
  struct PendEmit {
    ss::EmitArgs a;
    long ready;
  };
  std::deque<PendDet> pd;
  std::deque<PendEmit> pe;
  // Ring rows owed: overlapped calls do not write the averager ring (DetectArgs::hist_by_fft) — the drain does, for the last two
  // of them (the newest call's rows are what the next call starts from, the one before's are "the rows before the last batch"
  // ss_read_window hands out), from their PSD planes, which stay valid until then (the caller's by contract, the library's own
  // rotate over four).
  struct RingOwed {
    const float* psd = nullptr;
    int nframes = 0;
    float* hist_out = nullptr;
    const float* thr = nullptr;
  };
  RingOwed ring_owed[2];
  bool deep_prev_ok = false;   // the previous call overlapped: its PSD plane serves as this call's halo
  int deep_prev_frames = 0;
  // The caller's buffers of the last calls, which stages still in flight may read or write. A buffer handed to a later call
  // again is safe without further ado when the launches that touch it for the two calls share a queue (an even number of
  // launches apart, the earlier one at least two back); an odd distance needs an event from the other queue — recorded
  // after every launch from the first time a caller is seen rotating an odd number of buffer sets (about 1 us per step, plus
  // 2 us for the wait) — and anything closer drains the pipeline first.
  struct Buffers {
    const void* p[6];  // psd, rel | avg, cand_off, cand_idx, cand_avg
    size_t bytes[6];
    long launch;       // the call's FFT launch; its detect stage runs in launch + nq (last to touch psd / rel), its emit stage in launch + 2 nq
    uintptr_t lo, hi;  // address range that holds all of them (a cheap first test)
    uintptr_t iq_lo, iq_hi;  // the call's input frames: read by its FFT launch and, the last of them, once more by the next call's
  };
  std::deque<Buffers> deep_buffers;
  // A caller that waits after every call gains nothing from queues and deferred stages and pays for the fork and the join:
  // once ss_sync / ss_flush has come after a single call, calls run their three stages in order on the public stream
  // until two calls arrive without one in between.
  int deep_calls_since_sync = 0;
  bool deep_eager = false;
  bool deep_iq_recycled = false;  // the caller was seen refilling an input buffer a call in flight still reads: no overlap for this context any more
  bool deep_events = false;   // record ev_launch after every launch
  long deep_events_from = 0;  // first launch of this run of launches that has its event
  // k_scan_step's dispatch-order table for the current launch shape (rebuilt when the shape changes; two buffers so that a
  // launch still in flight keeps the table it was given)
  struct OrderTable {
    int key[6];  // FFT / detect / emit / plan / row workgroups of the launch shape, and which order pattern it was built from
    uint32_t* d;
    unsigned long long used;       // 0: free
    std::vector<uint32_t> host;    // what was uploaded (kept alive: the copy is asynchronous)
    hipStream_t stream;            // the stream it was uploaded on
  };
  std::vector<OrderTable> order_tables;  // one per launch shape met so far (a handful: steady state, pipeline fill, drain)
  size_t order_capacity = 0;
  unsigned long long order_clock = 0;
  int n_cus = 256;
  uint32_t* d_mask2[2 * kMaxQueues] = {};
  float* d_avg2[2 * kMaxQueues] = {};
  int* d_off4[2 * kMaxQueues] = {};  // the library's copy of the candidate offsets, per rotating set (d_off = the latest)
  float* d_psd2[2 * kMaxQueues + 1] = {};
  // Tile culling (8192 points, detect_fused.h): the FFT role's per-column maxima of a call's PSD rows, [32][max_batch],
  // rotating with the mask / avg / offset buffers (written by the call's FFT launch, read by its detect stage)
  float* d_segsum[2 * kMaxQueues] = {};
  int* d_live[2 * kMaxQueues] = {};  // DetectArgs::live: the tiles of a call that must be evaluated (plan role -> the launch's other workgroups)
  bool cull = false;
  // Tile culling, long transforms (65536 points and 2^19 / 2^20: the sizes whose rows go through k_fft_rows256_psd): the rows
  // kernel leaves the maximum of every 32-bin run per frame in d_smax, a ring of smax_rows frames (>= max_batch + 35) indexed by
  // the frame count since the last reset, and writes the averager ring itself in calls without learning frames; k_plan_long
  // lists the tiles that must be evaluated in d_tlist (rotating with the mask buffers). clean_abs: the first frame (counted
  // since the last reset) from which every frame is a non-learning frame under the current noise ceiling — a tile is only
  // tested when all 36 of its rows are at or behind it.
  bool cull_long = false;
  float* d_smax = nullptr;
  int smax_rows = 0;
  int* d_tlist[2 * kMaxQueues] = {};
  long long clean_abs = 0;
  const float* last_avg = nullptr;  // the avg plane (sparse or kept) of the last batch
  const float* last_hist = nullptr;       // ring rows as they were before the last batch
  float* last_thr = nullptr;
  // back end, unfused path (any other grouping): one buffer holds the ring rows + the batch rows
  float* d_rel = nullptr;   // (G-1) history rows + max_batch rows
  float* d_hist_tmp = nullptr;
  float* d_avgy = nullptr;
  float2* d_work = nullptr;  // four-step intermediate (N > 8192)
  float2* d_tw256 = nullptr; // W_256^(m r), r*16 + m: second pass of the 256-point register FFTs (N >= 65536)
  float2* d_tw_rowsR = nullptr;  // N = 2^17 .. 2^19: [q][k'] W_N2^(q k') for k_fft_rows256xR_psd (N2 = 512 .. 2048)
  float2* d_tw_small = nullptr;  // N = 1024, 2048, 4096: [q][k'] W_N^(q k') for the final radix-R pass of k_fft256xR_psd
  float2* d_tw_sub = nullptr;   // N = 2^19, 2^20: [c][b] W_N2^(b c) for k_fft_sub_dft (N2 = N / 256 = 256 A)
  float2* d_tw_cols = nullptr;  // N >= 65536: step-A twiddle factored for k_fft_cols256, [j][n2] W_N^(n2 j) then [k][n2] W_N^(16 n2 k)
  // N = 2^20 as 1024 x 1024 (fft1024_kernels.h): two passes over the work buffer instead of three
  bool two_pass = false;  // (its tables: one block in d_tw_cols)
  // Detect mode at 2^20 points (BASELINE config 5): the rows kernel writes the batch's rows as noise-relative rows (rel = dB - thr)
  // into the averager ring's buffer — behind the window for calls shorter than the ring, as a region of their own whose last 35 rows
  // are the next window otherwise (place_ring) — and no dB plane is written at all: the detect tiles take the batch's rows from
  // there, with a ceiling of zeros to subtract (x - 0.0f is x: the same bits) — unless somebody wants a plane.
  float* d_win1024 = nullptr;        // the window taps in the 1024-point column tiles' order
  float2* d_wtab1024 = nullptr;      // the default window only: (cos, sin)(2 pi m / (N - 1)), m < N / 16 — the column tiles of 2^20-point (fft1024_kernels.h, WCALC) and of 65536-point frames (fft256_kernels.h) form their taps from it
  float* d_zero_row = nullptr;       // n zeros
  const float* last_rel_rows = nullptr;  // the last batch's rows as rel values (ring-only calls: last_psd is null then)
  bool use_fft256 = false;
  int* d_counts = nullptr;
  int* d_off = nullptr;
  // host-entry staging
  void* d_in = nullptr;
  int* d_cand_idx = nullptr;
  float* d_cand_avg = nullptr;
  int cand_cap_alloc = 0;
  const float* last_psd = nullptr;  // where the last batch's PSD plane lives (device)
  int last_n = 0;
  // optional per-launch timing of the dominant (FFT+PSD) kernel: start/stop events attached to the
  // dispatch itself (hipExtLaunchKernelGGL), read back by ss_kernel_timing_read
  bool prof_on = false;
  int prof_every = 1;       // attach events to every prof_every-th launch only (the event packets cost ~4 us of GPU timeline each)
  unsigned prof_seen = 0;
  std::vector<hipEvent_t> prof_events;  // pairs
  std::vector<int> prof_slots;          // which kernel of the chain each pair timed (SS_KSLOT_*)
  std::vector<int> prof_frames;         // ... and how many frames that launch covered (a call taken through in chunks: the chunk's)
  int prof_launch_frames = 0;           // frames of the launches being enqueued now (run_batch, its chunk loops)
  size_t prof_used = 0;
  bool prof_call = false;   // the current call is a sampled one: its other kernels (rows, radix-A step, plan) carry events too
  // ss_get_stats: host-side counters, and the device-side ones (detect_fused.h kStat*: tiles tested / culled, wait fallbacks)
  ss_stats stats{};
  unsigned long long* d_stats = nullptr;
  // StepArgs::wait_limit. A poll is a ~0.25 us sleep and a load, ~0.75 us in all; the plan role publishes within 5-10 us of its launch's
  // start, so a consumer normally polls not at all (FFT role: it asks two thirds into its frame) or a dozen times (drain launches);
  // 128 polls are 0.1 ms. The bound is not decoration: with TWO PROCESSES on one GPU (bench.py --gpus 2 on a one-GPU box) the circular
  // wait the unbounded spin invited did happen — a launch's plan workgroups waiting for a slot on an XCD full of the other process's
  // spinning consumers, whose own plan workgroups waited for a slot held by this process's spinning consumers: with a limit of 2^14 every
  // consumer of such a launch polled for 12 ms, 762 fell back, and the run lost three quarters of its rate to that one launch
  // (profiles/r04/s9_summary.txt); round 3's spin would have hung there. A consumer that helps itself early costs 5 us.
  int wait_limit = 128;
  // ss_input_wait: the launches (or, in order on the public stream, the calls) whose events a producer may wait on
  hipEvent_t ev_call[32] = {};   // in-order contexts: recorded on the public stream behind every call once ss_input_wait has been used
  bool input_events = false;
  long input_events_from = 0;    // deep pipelining: first launch of the current run of launches that has its event (ev_launch)
  unsigned long long call_seq = 0;  // device calls so far (ss_process_device)
  // SS_FLAG_REFERENCE_NAN (reference_nan.h): per-frame first non-finite bins of the batch, the state carried between batches, per-frame poison limits
  bool ref_nan = false;
  int* d_nf = nullptr;        // [max_batch][4]: first NaN / -inf / +inf bin of every frame of the batch (n: none)
  int* d_nan_state = nullptr; // ss::NanState
  int* d_bad_from = nullptr;  // [max_batch]: bins >= this are NaN in the reference's avg row (n: none)
  NanStage nan_pending{};     // of the call whose detect stage is still deferred
  std::mutex mtx;
  char err[512] = "";
};

namespace {

int fail(ss_ctx* c, int status, const char* fmt, ...) {
  char* dst = c ? c->err : g_create_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return status;
}

#define SS_HIP(ctx, call)                                                                              \
  do {                                                                                                 \
    hipError_t e_ = (call);                                                                            \
    if (e_ != hipSuccess) return fail(ctx, SS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
  } while (0)

bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

// Wait for a stream. A blocking hipStreamSynchronize wakes the caller tens of microseconds after the work has finished — as
// long as a whole 1024-frame step, several times a work() call of a few frames — so the wait polls first (the scanner's
// calls finish within a few hundred microseconds) and only then blocks.
hipError_t stream_wait(hipStream_t stream) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t e = hipStreamQuery(stream);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(2000)) break;
  }
  return hipStreamSynchronize(stream);
}

size_t in_bytes_per_sample(int fmt) { return fmt == SS_FMT_CF32 ? 8 : 2; }

// getFft — sources/utils/radio_utils.cpp:98-104
int get_fft(int32_t sample_rate, int32_t max_step) {
  uint32_t v = 1;
  while ((double)max_step < (double)sample_rate / v) v <<= 1;
  return (int)v;
}

// isIndexInRange (sources/radio/sdr_device.cpp:153-158) && !isIndexIgnored (transmission.cpp:156-164),
// evaluated once per retune on the host and kept as one byte per bin on the device.
void build_pass_mask(const ss_ctx* c, std::vector<uint8_t>& out) {
  const int n = c->n;
  const int32_t fs = c->cfg.sample_rate;
  const double step = (double)fs / n;
  const int32_t center = (c->range_lo + c->range_hi) / 2;
  out.resize((size_t)n);
  for (int i = 0; i < n; ++i) {
    const int32_t f = center + (int32_t)(step * (i + 0.5)) - fs / 2;
    bool ok = c->range_lo <= f && f <= c->range_hi;
    for (size_t k = 0; ok && k + 1 < c->ignored.size(); k += 2) {
      if (c->ignored[k] <= f && f <= c->ignored[k + 1]) ok = false;
    }
    out[(size_t)i] = ok ? 1 : 0;
  }
}

NoiseState* noise_for(ss_ctx* c, int32_t center) {
  for (auto& z : c->noise) {
    if (z.center == center) return &z;
  }
  return nullptr;
}

// next start/stop event pair for a timed launch, or false when timing is off / the pool is exhausted
// A start/stop event pair for the launch of kernel `slot` (SS_KSLOT_*). The FFT launch of a call (slot 0) decides whether the call
// is a sampled one (every prof_every-th); the call's other kernels follow it (prof_call).
bool prof_take(ss_ctx* c, int slot, hipEvent_t* a, hipEvent_t* b) {
  if (c->prof_used + 2 > c->prof_events.size()) {
    if (c->prof_events.size() >= 2 * 8192) return false;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    c->prof_events.push_back(e0);
    c->prof_events.push_back(e1);
    c->prof_slots.push_back(0);
    c->prof_frames.push_back(0);
  }
  *a = c->prof_events[c->prof_used];
  *b = c->prof_events[c->prof_used + 1];
  c->prof_slots[c->prof_used / 2] = slot;
  c->prof_frames[c->prof_used / 2] = c->prof_launch_frames;
  c->prof_used += 2;
  return true;
}
bool prof_pair(ss_ctx* c, hipEvent_t* a, hipEvent_t* b) {
  c->prof_call = false;
  if (!c->prof_on) return false;
  if ((c->prof_seen++ % (unsigned)c->prof_every) != 0) return false;
  c->prof_call = prof_take(c, SS_KSLOT_STEP, a, b);
  return c->prof_call;
}
bool prof_slot(ss_ctx* c, int slot, hipEvent_t* a, hipEvent_t* b) { return c->prof_on && c->prof_call && prof_take(c, slot, a, b); }
// a launch on the context's stream that carries events when the call is a sampled one
#define SS_LAUNCH_SLOT(c, slot, kernel, grid, block, lds, ...)                                                        \
  do {                                                                                                                \
    hipEvent_t pe0_, pe1_;                                                                                            \
    if (prof_slot((c), (slot), &pe0_, &pe1_)) hipExtLaunchKernelGGL(kernel, grid, block, lds, (c)->stream, pe0_, pe1_, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(kernel, grid, block, lds, (c)->stream, __VA_ARGS__);                                      \
  } while (0)

// ---- FFT + PSD dispatch ---------------------------------------------------------------------------
template <int LOGN, int FMT>
void launch_lds(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, float* d_psd) {
  constexpr int LOGTOT = LOGN < 11 ? 11 : LOGN;
  constexpr int FPB = (1 << LOGTOT) >> LOGN;
  const int blocks = (nframes + FPB - 1) / FPB;
  const size_t lds = sizeof(float2) << LOGTOT;
  hipEvent_t e0, e1;
  if (prof_pair(c, &e0, &e1)) {
    hipExtLaunchKernelGGL((ss::k_fft_psd_lds<LOGN, LOGTOT, FMT>), dim3(blocks), dim3(ss::kFftThreads), lds, c->stream, e0, e1, 0, d_iq,
                          item_stride, nframes, c->d_win, c->d_tw, c->db_off, c->cfg.int_scale, d_psd);
  } else {
    hipLaunchKernelGGL((ss::k_fft_psd_lds<LOGN, LOGTOT, FMT>), dim3(blocks), dim3(ss::kFftThreads), lds, c->stream, d_iq, item_stride,
                       nframes, c->d_win, c->d_tw, c->db_off, c->cfg.int_scale, d_psd);
  }
}

template <int LOGN1, int LOGN2, int FMT>
void launch_four_step(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, float* d_psd) {
  constexpr int N1 = 1 << LOGN1, N2 = 1 << LOGN2;
  const size_t lds = sizeof(float2) * ((1 << 13) + 128);  // 8192 points + one pad element per sub-FFT (at most 128 of them)
  const int col_tiles = N2 >> (13 - LOGN1);
  const int row_tiles = N1 >> (13 - LOGN2);
  hipLaunchKernelGGL((ss::k_fft_cols<LOGN1, LOGN2, FMT>), dim3(nframes * col_tiles), dim3(ss::kFftThreads), lds, c->stream, d_iq,
                     item_stride, c->d_win, c->d_tw, c->cfg.int_scale, c->d_work);
  hipLaunchKernelGGL((ss::k_fft_rows_psd<LOGN1, LOGN2>), dim3(nframes * row_tiles), dim3(ss::kFftThreads), lds, c->stream, c->d_work,
                     c->d_tw, c->db_off, d_psd);
}

// N >= 16384: N = 256 x N2. Columns: 256-point register FFTs; rows: the same for N2 = 256, the generic LDS kernel otherwise.
ss::ColsArgs cols256_args(ss_ctx* c, const void* d_iq, long long item_stride) {
  ss::ColsArgs g{};
  g.iq = d_iq;
  g.item_stride = item_stride;
  g.win = c->two_pass ? c->d_win1024 : c->d_win;
  g.tw256 = c->d_tw256;
  g.twc = c->d_tw_cols;
  g.scale = c->cfg.int_scale;
  g.work = c->d_work;
  g.logn2 = c->two_pass ? 10 : c->logn - 8;
  if ((c->two_pass || c->rows1024x256) && c->cull_long) {  // (the column tiles clear the frames' rows of the run-maxima ring: the rows kernel gathers them by atomic maxima)
    g.smax = reinterpret_cast<unsigned*>(c->d_smax);
    g.smax_mask = c->smax_rows - 1;
    g.abs0 = (int)(c->abs_frames & 0x3fffffff);
  }
  g.wtab = c->d_wtab1024;
  return g;
}

// N = 2^20 in two passes (fft1024_kernels.h): the column tiles are k_scan_step's FFT role (KIND 3) or, for contexts without the
// step kernel, a launch of their own; the rows follow as their own launch either way.
// frame0: the first of these frames within its call (a call of many frames goes through in chunks, run_batch)
template <int FMT>
void launch_cols1024_fmt(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, int frame0 = 0) {
  ss::ColsArgs g = cols256_args(c, d_iq, item_stride);
  g.abs0 += frame0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (!prof_pair(c, &e0, &e1)) e0 = e1 = nullptr;
  auto go = [&](auto kernel, int tiles, int threads, int lds) {
    if (e0) hipExtLaunchKernelGGL(kernel, dim3(nframes * tiles), dim3(threads), lds, c->stream, e0, e1, 0, g);
    else hipLaunchKernelGGL(kernel, dim3(nframes * tiles), dim3(threads), lds, c->stream, g);
  };
  const bool wcalc = g.wtab != nullptr;
  if (c->have_plan && c->diag.cols1024_wide) {  // the plan of the call before at the front of this launch (detect_fused.h)
    const int plan_wgs = ss::plan_fused_wgs(ss::plan_long_blocks(c->pend_plan.layout, c->pend_plan.cols, c->n));
    const dim3 grid((unsigned)(plan_wgs + nframes * 64)), block(1024);
    auto gop = [&](auto kernel) {
      if (e0) hipExtLaunchKernelGGL(kernel, grid, block, ss::fft1024_cols_lds_bytes(4), c->stream, e0, e1, 0, g, c->pend_plan_det, c->pend_plan, plan_wgs);
      else hipLaunchKernelGGL(kernel, grid, block, ss::fft1024_cols_lds_bytes(4), c->stream, g, c->pend_plan_det, c->pend_plan, plan_wgs);
    };
    if (wcalc) gop(ss::k_fft_cols1024_plan<FMT, true>);
    else gop(ss::k_fft_cols1024_plan<FMT, false>);
    c->have_plan = false;
    return;
  }
  if (c->diag.cols1024_wide && wcalc) go(ss::k_fft_cols1024<FMT, 4, true>, 64, 1024, ss::fft1024_cols_lds_bytes(4));
  else if (c->diag.cols1024_wide) go(ss::k_fft_cols1024<FMT, 4>, 64, 1024, ss::fft1024_cols_lds_bytes(4));
  else go(ss::k_fft_cols1024<FMT, 3>, 128, 512, ss::kFft1024ColsLdsBytes);
}
void launch_cols1024(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, int frame0 = 0) {
  const size_t sample = c->cfg.in_format == SS_FMT_CF32 ? 8 : 2;
  const void* iq = static_cast<const char*>(d_iq) + (size_t)frame0 * (size_t)item_stride * sample;
  switch (c->cfg.in_format) {
    case SS_FMT_CF32: return launch_cols1024_fmt<ss::FMT_CF32>(c, iq, item_stride, nframes, frame0);
    case SS_FMT_CS8: return launch_cols1024_fmt<ss::FMT_CS8>(c, iq, item_stride, nframes, frame0);
    default: return launch_cols1024_fmt<ss::FMT_CU8>(c, iq, item_stride, nframes, frame0);
  }
}
ss::Rows1024Args rows1024_args(ss_ctx* c, float* d_psd, const ss::RowsExtra& rx) {
  ss::Rows1024Args g{};
  g.work = c->d_work;
  g.tw256 = c->d_tw256;
  g.tw1024 = c->d_tw_cols + ss::kFft1024Tw1024;
  g.db_off = c->db_off;
  g.psd = d_psd;
  g.x = rx;
  return g;
}
// 262144 points: the same tile over 256 rows of 1024 points behind the 256-point column tiles (W_1024^k' is row 1 of the x R rows kernel's table)
ss::Rows1024Args rows1024x256_args(ss_ctx* c, float* d_psd, const ss::RowsExtra& rx) {
  ss::Rows1024Args g{};
  g.work = c->d_work;
  g.tw256 = c->d_tw256;
  g.tw1024 = c->d_tw_rowsR + 256;
  g.db_off = c->db_off;
  g.psd = d_psd;
  g.x = rx;
  return g;
}
ss::Rows256Args rows256_args(ss_ctx* c, float* d_psd, const ss::RowsExtra& rx) {
  ss::Rows256Args g{};
  g.work = c->d_work;
  g.tw256 = c->d_tw256;
  g.db_off = c->db_off;
  g.psd = d_psd;
  g.logn = c->logn;
  g.lognsub = c->logn - 16;
  g.x = rx;
  return g;
}
void launch_rows1024(ss_ctx* c, int nframes, float* d_psd, const ss::RowsExtra& rx) {
  const ss::Rows1024Args g = rows1024_args(c, d_psd, rx);
  SS_LAUNCH_SLOT(c, SS_KSLOT_ROWS, ss::k_fft_rows1024_psd<10>, dim3(nframes * 128), dim3(512), ss::kFft1024RowsLdsBytes, g);
}
void launch_rows1024x256(ss_ctx* c, int nframes, float* d_psd, const ss::RowsExtra& rx) {
  const ss::Rows1024Args g = rows1024x256_args(c, d_psd, rx);
  SS_LAUNCH_SLOT(c, SS_KSLOT_ROWS, ss::k_fft_rows1024_psd<8>, dim3(nframes * 32), dim3(512), ss::kFft1024RowsLdsBytes, g);
}

// with_cols = false: the column half ran as a role of k_scan_step (run_batch), only the row half is launched here
template <int LOGN2, int FMT>
void launch_four_step256(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, float* d_psd, bool with_cols = true, const ss::RowsExtra& rx = ss::RowsExtra{}) {
  constexpr int N2 = 1 << LOGN2;
  if (with_cols)
    hipLaunchKernelGGL((ss::k_fft_cols256<FMT>), dim3(nframes * (N2 / 32)), dim3(512), ss::kFft256ColsLdsBytes, c->stream, cols256_args(c, d_iq, item_stride));
  if constexpr (LOGN2 == 8) {
    SS_LAUNCH_SLOT(c, SS_KSLOT_ROWS, ss::k_fft_rows256_psd, dim3(nframes * 8), dim3(512), ss::kFft256RowsPsdLdsBytes, (const float2*)c->d_work,
                   (const float2*)c->d_tw256, c->db_off, d_psd, 16, 0, rx);
  } else {
    bool done = false;
    if constexpr (LOGN2 == 10) {
      if (c->x256_tile) {  // 262144 points: the 1024-point row tile (with tile culling: run maxima, ring rows, the dB plane only where one is wanted)
        launch_rows1024x256(c, nframes, d_psd, rx);
        done = true;
      }
    }
    if constexpr (LOGN2 >= 9 && LOGN2 <= 12) {
      if (!done && c->d_tw_rowsR) {  // rows of 256 R points: R sub-sequences through the register passes, an R-point DFT across them
        constexpr int LOGR = LOGN2 - 8;
        SS_LAUNCH_SLOT(c, SS_KSLOT_ROWS, (ss::k_fft_rows256xR_psd<LOGR>), dim3(nframes * (256 / (32 >> LOGR))), dim3(512), ss::fft_rowsR_lds_bytes(LOGR),
                       (const float2*)c->d_work, (const float2*)c->d_tw256, (const float2*)c->d_tw_rowsR, c->db_off, d_psd,
                       c->diag.fft_xcd_map ? 1 : 0);
        done = true;
      }
    }
    if constexpr (LOGN2 >= 9) {
      if (!done && c->d_tw_sub) {
        // rows of 256 A points (A = 8, 16) as an in-place radix-A step over the stride-256 index, then 256-point rows
        constexpr int A = 1 << (LOGN2 - 8);
        SS_LAUNCH_SLOT(c, SS_KSLOT_SUB, (ss::k_fft_sub_dft<A>), dim3(nframes * 256), dim3(256), 0, c->d_work, (const float2*)c->d_tw_sub);
        SS_LAUNCH_SLOT(c, SS_KSLOT_ROWS, ss::k_fft_rows256_psd, dim3(nframes * A * 8), dim3(512), ss::kFft256RowsPsdLdsBytes, (const float2*)c->d_work,
                       (const float2*)c->d_tw256, c->db_off, d_psd, 8 + LOGN2, LOGN2 - 8, rx);
        done = true;
      }
    }
    if (!done) {
      const size_t lds = sizeof(float2) * ((1 << 13) + 128);
      const int row_tiles = 256 >> (13 - LOGN2);
      SS_LAUNCH_SLOT(c, SS_KSLOT_ROWS, (ss::k_fft_rows_psd<8, LOGN2>), dim3(nframes * row_tiles), dim3(ss::kFftThreads), lds, c->d_work, c->d_tw,
                     c->db_off, d_psd);
    }
  }
}

// ---- k_scan_step (8192 points): any subset of the three roles in one launch ------------------------------------------
// Dispatch order of a launch that carries more than one role, from the pattern in diag.step_order (see there): one word per
// workgroup (role << 24 | item) in device memory, rebuilt only when the launch shape changes.
void step_order(ss_ctx* c, ss::StepArgs& a, hipStream_t stream) {
  const int n_fft = ss::step_fft_wgs(a), wg_det = ss::step_det_wgs(a), wg_emit = ss::step_emit_wgs(a), wg_plan = ss::step_plan_wgs(a), wg_rows = a.n_rows;  // (FFT WORKGROUPS)
  a.order = nullptr;
  a.prio_fft = c->diag.prio_fft;
  a.prio_other = c->diag.prio_other;
  if (n_fft == 0 || (wg_det == 0 && wg_rows == 0) || c->diag.no_order_table) return;  // nothing to interleave: the kernel takes the roles one after the other (plan, emit, detect, FFT)
  // which pattern the table is built from is part of its key: a context's fold launches (4 workgroups per frame) and its four-step column
  // launches (8 per frame) can have the same counts — any permutation is correct, but the measured order of each form must not be lost to the other's
  const int pattern = wg_rows ? 1 : c->use_fft8192 ? 2 : a.dif.iq ? 3 : 4;
  for (auto& t : c->order_tables)
    if (t.used && t.stream == stream && t.key[0] == n_fft && t.key[1] == wg_det && t.key[2] == wg_emit && t.key[3] == wg_plan && t.key[4] == wg_rows && t.key[5] == pattern) {  // (uploaded on this stream: complete for this launch by stream order)
      t.used = ++c->order_clock;
      a.order = t.d;
      return;
    }
  struct Seg {
    int role, count;
  };
  std::vector<Seg> prefix, cycle;
  const std::string& sp = pattern == 1 ? c->diag.step_order_merged : pattern == 2 ? c->diag.step_order : pattern == 3 ? c->diag.step_order_fold : c->diag.step_order_long;
  {
    std::vector<Seg>* into = &prefix;
    for (size_t i = 0; i < sp.size();) {
      const char ch = sp[i];
      if (ch == '|') {
        into = &cycle;
        ++i;
        continue;
      }
      const int role = ch == 'F' ? ss::ROLE_FFT : ch == 'D' ? ss::ROLE_DET : ch == 'E' ? ss::ROLE_EMIT : ch == 'R' ? ss::ROLE_ROWS : ch == 'P' ? ss::ROLE_PLAN : ss::ROLE_NONE;
      ++i;
      int count = 0;
      if (i < sp.size() && sp[i] == '*') {
        count = 1 << 27;
        ++i;
      } else {
        while (i < sp.size() && sp[i] >= '0' && sp[i] <= '9') count = count * 10 + (sp[i++] - '0');
      }
      if (role != ss::ROLE_NONE && count > 0) into->push_back(Seg{role, count});
      while (i < sp.size() && sp[i] != '|' && !(sp[i] >= 'A' && sp[i] <= 'Z')) ++i;  // separators
    }
  }
  const int total[6] = {0, n_fft, wg_det, wg_emit, wg_plan, wg_rows};
  int next[6] = {0, 0, 0, 0, 0, 0};
  std::vector<uint32_t> out;
  out.clear();
  // the few plan workgroups first: the launch's other workgroups wait for their list, never the other way round — unless the pattern
  // places them itself ('P': a long transform's plan, which nothing in its own launch waits for)
  const bool plan_placed = sp.find('P') != std::string::npos && !c->use_fft8192;
  if (!plan_placed)
    for (int k = 0; k < wg_plan; ++k) out.push_back((uint32_t)ss::ROLE_PLAN << 24 | (uint32_t)k);
  const auto place = [&](const Seg& sg) {
    for (int k = 0; k < sg.count && next[sg.role] < total[sg.role]; ++k) out.push_back((uint32_t)sg.role << 24 | (uint32_t)next[sg.role]++);
  };
  for (const Seg& sg : prefix) place(sg);
  const size_t want = (size_t)n_fft + (size_t)wg_det + (size_t)wg_emit + (size_t)wg_plan + (size_t)wg_rows;
  while (out.size() < want) {
    const size_t before = out.size();
    for (const Seg& sg : cycle) place(sg);
    if (out.size() == before) {  // the cycle does not reach what is left: whatever remains, FFT first
      place(Seg{ss::ROLE_FFT, 1 << 27});
      place(Seg{ss::ROLE_ROWS, 1 << 27});
      place(Seg{ss::ROLE_DET, 1 << 27});
      place(Seg{ss::ROLE_EMIT, 1 << 27});
      place(Seg{ss::ROLE_PLAN, 1 << 27});
    }
  }
  // A new shape: one of the tables allocated at ss_create, filled by a copy on the launch's own stream (the table's host
  // image stays alive with it), so that no launch ever waits for the host: launches of other shapes keep their tables, and
  // this launch sees its own complete by stream order. Only when all sixteen tables are taken — a caller whose call sizes
  // keep changing AND whose launches carry detect workgroups of their own — is the least recently used one recycled, after
  // everything in flight has finished. (No table at all is always correct: the roles in segments.)
  if (out.size() > c->order_capacity || c->order_tables.empty()) return;
  ss_ctx::OrderTable* slot = nullptr;
  for (auto& t : c->order_tables)
    if (t.used == 0) {
      slot = &t;
      break;
    }
  if (!slot) {
    for (hipStream_t q : c->s_ab)
      if (q) (void)hipStreamSynchronize(q);
    (void)hipStreamSynchronize(c->stream);
    slot = &c->order_tables[0];
    for (auto& t : c->order_tables)
      if (t.used < slot->used) slot = &t;
  }
  slot->host = out;
  slot->stream = stream;
  if (hipMemcpyAsync(slot->d, slot->host.data(), sizeof(uint32_t) * slot->host.size(), hipMemcpyHostToDevice, stream) != hipSuccess) {
    slot->key[0] = -1;
    slot->used = 0;
    return;
  }
  slot->key[0] = n_fft;
  slot->key[1] = wg_det;
  slot->key[2] = wg_emit;
  slot->key[3] = wg_plan;
  slot->key[4] = wg_rows;
  slot->key[5] = pattern;
  slot->used = ++c->order_clock;
  a.order = slot->d;
}

template <int FMT, bool SPEC>
void launch_step_variant(ss_ctx* c, const ss::StepArgs& a, hipEvent_t e0, hipEvent_t e1, hipStream_t stream) {
  const dim3 grid((unsigned)ss::step_items(a)), block(ss::kStepThreads);
  auto go = [&](auto kernel) {
    if (e0) hipExtLaunchKernelGGL(kernel, grid, block, ss::kStepLdsBytes, stream, e0, e1, 0, a);
    else hipLaunchKernelGGL(kernel, grid, block, ss::kStepLdsBytes, stream, a);
  };
  // (2^20 points: rows of 32768 mask words, the emit role gives a frame to all eight waves; the FFT role is the row tiles — KIND 4,
  // the column half being a launch of its own — or, behind a switch of the diagnostics build, 8-column column tiles, KIND 3)
  // (KIND 5: a launch without an FFT role — the drain — whose detect workgroups share the plan's list out in a loop)
  if (c->two_pass && a.list_loop) return go(ss::k_scan_step<FMT, SPEC, 2, true, false, 5>);
  if (c->two_pass) return c->diag.cols1024_wide ? go(ss::k_scan_step<FMT, SPEC, 2, true, false, 4>) : go(ss::k_scan_step<FMT, SPEC, 2, true, false, 3>);
  if constexpr (FMT != ss::FMT_CF32 && !SPEC) {  // (the fold's launches, and the drains of the stages that wait behind them: their tiles read the fold's rows)
#ifdef SS_DIAG
    if (!c->use_fft8192 && c->ring_perm8 && c->dif_logq == 3 && (c->diag.prio_fft || c->diag.prio_other)) return go(ss::k_scan_step<FMT, SPEC, 2, true, true, 8>);
#endif
    if (!c->use_fft8192 && c->ring_perm8 && c->dif_logq == 3 && a.dif.iq && a.n_fft == 8 * a.dif.nframes && SS_DIF8_W == 4) return go(ss::k_scan_step<FMT, SPEC, 2, true, false, 11>);  // (a short call: one residue per workgroup)
    if (!c->use_fft8192 && c->ring_perm8) return c->dif_logq == 4 ? go(ss::k_scan_step<FMT, SPEC, 2, true, false, 9>) : go(ss::k_scan_step<FMT, SPEC, 2, true, false, 8>);
  }
  if (!c->use_fft8192 && a.n_fft && a.rows256.work && !a.n_rows) return go(ss::k_scan_step<FMT, SPEC, 2, true, false, 6>);  // (65536 points: the row tiles as the FFT role; rows of 2048 mask words: the wide emit role)
  if constexpr (!SPEC) {
    if (!c->use_fft8192 && c->rows1024x256) return c->merge ? go(ss::k_scan_step<FMT, SPEC, 2, true, false, 12>) : go(ss::k_scan_step<FMT, SPEC, 2, true, false, 10>);  // (262144 points: KIND 2 with the plan of layout 3; with the row tiles as one more role)
  }
  if (!c->use_fft8192 && c->merge) return go(ss::k_scan_step<FMT, SPEC, 2, true, false, 7>);  // (one launch per call: KIND 2's roles and the row tiles as one more; its drains too)
  if (!c->use_fft8192) return a.emit_per_wg == 1 ? go(ss::k_scan_step<FMT, SPEC, 2, true, false, 2>) : go(ss::k_scan_step<FMT, SPEC, 2, true, false, 1>);
#ifdef SS_DIAG
  if (c->diag.fft_tw == 0) return go(ss::k_scan_step<FMT, SPEC, 0, false>);
  if (c->diag.fft_tw == 1) return go(ss::k_scan_step<FMT, SPEC, 1, false>);
  if (!c->diag.fft_swz) return go(ss::k_scan_step<FMT, SPEC, 2, false>);
  if (c->diag.prio_fft || c->diag.prio_other) return go(ss::k_scan_step<FMT, SPEC, 2, true, true>);
#endif
  go(ss::k_scan_step<FMT, SPEC, 2, true>);
}

// The FFT role of a launch: 8192-point frames, or the column tiles of a long transform.
// the per-column maxima of the halo frames a launch transforms once more: behind their rows in the launch's d_halo buffer
inline float* halo_segsum_of(const ss_ctx* c, float* halo_psd) { return halo_psd + (size_t)c->n * (size_t)kHistRows; }

struct FftRole {
  const ss::Fft8192Args* frames = nullptr;
  const ss::ColsArgs* cols = nullptr;
  const ss::Rows1024Args* rows1024 = nullptr;  // 2^20 points in two passes: the ROW tiles (the column half is a launch of its own)
  const ss::Rows256Args* rows256 = nullptr;    // 65536 points with tile culling: the ROW tiles (the column half is a launch of its own)
  const ss::Dif8Front* dif = nullptr;          // 65536 points, the radix-8 fold: n = 4 x frames workgroups (two residues each); `frames` carries the transform's tables and the rows' place
  int n = 0;  // frames / column tiles
  const void* halo_iq = nullptr;  // deep pipelining: n_halo frames of the previous call go through the FFT again, into halo_psd
  float* halo_psd = nullptr;
  int n_halo = 0;
};

// fft / det / emit: null = role absent. Start/stop events ride on launches that carry an FFT role (the dominant work).
void launch_step(ss_ctx* c, const FftRole* fft, const ss::DetectArgs* det, int n_det_tiles, bool spec, const ss::EmitArgs* emit, hipStream_t stream = nullptr,
                 bool with_long_plan = false, const ss::Rows256Args* rows_role = nullptr, int n_rows = 0, const ss::Rows1024Args* rows1024_role = nullptr) {
  if (!stream) stream = c->stream;
#ifdef SS_DIAG
  if (c->diag.ablate_roles & 1) det = nullptr;
  if (c->diag.ablate_roles & 2) emit = nullptr;
#endif
  ss::StepArgs a{};
  a.emit_per_wg = (!c->use_fft8192 && (c->diag.emit_wide || c->two_pass) && c->n / 32 >= 2048) ? 1 : 8;
  if (fft && fft->dif) {
    a.fft = *fft->frames;
    a.dif = *fft->dif;
    a.n_fft = fft->n;
  } else if (fft && fft->frames) {
    const int n_fft = fft->n + fft->n_halo;
    a.fft = *fft->frames;
    a.n_fft = n_fft;
    a.halo_iq = fft->halo_iq;
    a.halo_psd = fft->halo_psd;
    a.n_halo = fft->n_halo;
    a.halo_segsum = (fft->halo_psd && c->cull && c->diag.halo_maxima) ? halo_segsum_of(c, fft->halo_psd) : nullptr;
  } else if (fft && fft->cols) {
    a.cols = *fft->cols;
    a.n_fft = fft->n;
  } else if (fft && fft->rows1024) {
    a.rows = *fft->rows1024;
    a.n_fft = fft->n;
  } else if (fft && fft->rows256) {
    a.rows256 = *fft->rows256;
    a.n_fft = fft->n;
  }
  // Tile culling (detect_fused.h): a detect stage whose only products are mask bits and counts is PLANNED — a few plan
  // workgroups list the tiles that may hold a candidate — and the list is evaluated by the launch's FFT workgroups after their
  // frames, two entries each (there are always enough: nframes + 20 workgroups for nframes + 16 pairs at most), or by detect
  // workgroups of its own in a launch without an FFT role.
  const int plan_cols = (det && c->n == 8192) ? ss::plan_cols_per_wg(det->nframes, det->shift) : 0;
  const bool planned = det && c->use_fft8192 && det->segsum && !det->rel_out && !det->avg_out && !spec && plan_cols > 0;
  if (det) {
    a.det = *det;
    if (planned) {
      a.n_plan = n_det_tiles;
      a.plan_cols = plan_cols;
      a.plan_first = c->diag.plan_first * ss::step_plan_wgs(a);  // (the first pairs of every list on detect workgroups of their own: scan_step.h)
      a.plan_by_fft = a.n_fft > 0 && a.n_fft + a.plan_first >= ss::step_plan_consumers(a) ? 1 : 0;  // (consumer p serves list p mod S)
      if (!a.plan_by_fft) a.plan_first = 0;
    } else if (det->tile_list && fft && (fft->cols || fft->rows1024 || fft->rows256 || fft->dif)) {
      a.list_by_fft = 1;  // long transforms, planned stage: FFT workgroup p takes pair list_first + p of the list after its own tile (scan_step.h)
      a.list_first = c->diag.list_first;  // (the first pairs on detect workgroups of their own, ahead of the FFT role: no tail)
      // The fold's launches dispatch their own workgroups first (step_order_fold). With more of them than the chip has CUs, the ones
      // dispatched first — one per CU, the older waves there — are through after 20 us, the second on each CU after 30: a pair of tiles
      // behind one of the first costs the launch nothing, on a detect workgroup it waits for a slot first (128-frame calls 34.8 -> 31.9 us,
      // profiles/r05/s30_*). With at most one per CU there are free slots from the start, and a pair behind a transform only makes
      // that workgroup late (64-frame calls 20.4 against 30.1 us).
      if (fft->dif && c->diag.list_first_fold != 0) a.list_first = c->diag.list_first_fold > 0 ? c->diag.list_first_fold - 1 : (fft->n > c->n_cus ? 0 : c->diag.list_first);
      a.n_det = 2 * (a.list_first + std::max(0, (n_det_tiles + 1) / 2 - a.list_first - fft->n));  // detect workgroups for the first pairs and for the pairs beyond
    } else if (det->tile_list && c->two_pass && !fft) {
      a.list_loop = 1;  // 2^20 points in two passes: 128 detect workgroups share the list out in a loop (scan_step.h)
      a.n_det = 2 * std::min(128, (n_det_tiles + 1) / 2);
    } else {
      a.n_det = n_det_tiles;
    }
  }
  if (emit) {
    a.emit = *emit;
    a.n_emit = emit->nframes;
  }
  if (rows_role && n_rows > 0) {  // one launch per call (KIND 7): the row half of the call before beside this call's column half
    a.rows256 = *rows_role;
    a.n_rows = n_rows;
  } else if (rows1024_role && n_rows > 0) {  // ... of 262144-point frames (KIND 12)
    a.rows = *rows1024_role;
    a.n_rows = n_rows;
  }
  if (with_long_plan && c->have_plan) {  // 65536 points: the plan of the call before as a role of this (column) launch
    a.plan_det = c->pend_plan_det;
    a.plan_long = c->pend_plan;
    a.n_plan_long = (ss::plan_blocks_of(c->pend_plan_det, c->pend_plan, c->n) + 1) / 2;
    c->have_plan = false;
  }
  if (ss::step_items(a) == 0) return;
  a.wait_limit = c->wait_limit;
  step_order(c, a, stream);
  // (no order table — none wanted, or none to be had: out of device memory — means the roles in segments, which is always correct)
#ifdef SS_DIAG
  a.hint_mode = c->diag.hint_mode;
  bool dump_stamps = false;
  if (!c->diag.stamp_path.empty() && fft && det && emit && ss::step_items(a) <= 8192) {
    if (!c->diag.d_stamps) (void)hipMalloc(&c->diag.d_stamps, sizeof(long long) * 4 * (8192 + 16384));  // (per workgroup, then per tile)
    if (c->diag.d_stamps && ++c->diag.stamp_launches == 40) {
      a.stamps = c->diag.d_stamps;
      a.det.stamp_mid = c->diag.d_stamps + 4 * 8192;  // (4 per tile, behind the per-workgroup stamps)
      (void)hipMemsetAsync(a.det.stamp_mid, 0, sizeof(long long) * 4 * 16384, stream);
      dump_stamps = true;
    }
  }
#endif
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (fft && (fft->rows1024 || fft->rows256)) {  // (the row half + passengers; the call's column launch has decided whether it is a sampled one)
    if (!prof_slot(c, SS_KSLOT_ROWS, &e0, &e1)) e0 = e1 = nullptr;
  } else if (fft && !prof_pair(c, &e0, &e1)) {
    e0 = e1 = nullptr;
  }
  const bool sp = spec && det;
  switch (c->cfg.in_format) {
    case SS_FMT_CF32: sp ? launch_step_variant<ss::FMT_CF32, true>(c, a, e0, e1, stream) : launch_step_variant<ss::FMT_CF32, false>(c, a, e0, e1, stream); break;
    case SS_FMT_CS8: sp ? launch_step_variant<ss::FMT_CS8, true>(c, a, e0, e1, stream) : launch_step_variant<ss::FMT_CS8, false>(c, a, e0, e1, stream); break;
    default: sp ? launch_step_variant<ss::FMT_CU8, true>(c, a, e0, e1, stream) : launch_step_variant<ss::FMT_CU8, false>(c, a, e0, e1, stream); break;
  }
#ifdef SS_DIAG
  if (dump_stamps) {
    const int wgs = ss::step_items(a);
    std::vector<long long> h((size_t)4 * wgs);  // (wgs <= 8192 here)
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy(h.data(), c->diag.d_stamps, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    if (FILE* fp = fopen(c->diag.stamp_path.c_str(), "w")) {
      for (int b = 0; b < wgs; ++b)
        fprintf(fp, "%d %lld %lld %lld %lld %lld %lld\n", b, h[4 * b], h[4 * b + 1], h[4 * b + 2] >> 32, h[4 * b + 2] & 0xffffffff, h[4 * b + 3] >> 32, h[4 * b + 3] & 0xffffffff);
      fclose(fp);
      const int n_tiles = std::min(16384, (a.det.n / 256) * ss::plan_frame_tiles(a.det.nframes, a.det.shift));
      std::vector<long long> m((size_t)4 * n_tiles);
      (void)hipMemcpy(m.data(), c->diag.d_stamps + 4 * 8192, sizeof(long long) * m.size(), hipMemcpyDeviceToHost);
      if (FILE* fm = fopen((c->diag.stamp_path + ".det").c_str(), "w")) {
        for (int t = 0; t < n_tiles; ++t) fprintf(fm, "%d %lld %lld %lld %lld\n", t, m[4 * t], m[4 * t + 1], m[4 * t + 2], m[4 * t + 3]);
        fclose(fm);
      }
    }
  }
#endif
}

ss::Fft8192Args fft8192_args(ss_ctx* c, const void* d_iq, long long item_stride, float* d_psd) {
  ss::Fft8192Args g{};
  g.iq = d_iq;
  g.item_stride = item_stride;
  g.win = c->d_win;
  g.tabs = ss::Fft8192V2Tables{c->d_tw8v2, c->d_tw8v2 + 256, c->d_tw8v2 + 256 + 384, c->d_tw8k + 256, c->d_tw8k + 256 + 1024};
  g.db_off = c->db_off;
  g.scale = c->cfg.int_scale;
  g.psd = d_psd;
  g.segsum = c->cull ? c->d_segsum[c->buf_cur] : nullptr;  // (run_backend_fused hands the same buffer to the call's detect stage)
  g.seg_pitch = c->cfg.max_batch;
  return g;
}

// Deep pipelining, spectrogram branch: add the partial sums of the oldest `count` waiting calls to their containers on
// s_fold, call by call and tile by tile (the additions of the in-detect form, in the same order). behind_public = false:
// behind everything the two queues hold now (the calls' detect stages are among it); true (a drain): behind the public
// stream, which has been made to wait for both queues and may have run detect stages itself.
void fold_spectrogram_slots(ss_ctx* c, size_t count, bool behind_public) {
  if (count == 0) return;
  if (behind_public) {
    (void)hipEventRecord(c->ev_fold_src[0], c->stream);
    (void)hipStreamWaitEvent(c->s_fold, c->ev_fold_src[0], 0);
  } else {
    for (int q = 0; q < c->nq; ++q) {
      (void)hipEventRecord(c->ev_fold_src[q], c->s_ab[q]);
      (void)hipStreamWaitEvent(c->s_fold, c->ev_fold_src[q], 0);
    }
  }
  const int batch = (int)(c->fold_batches++ & 15);
  for (size_t k = 0; k < count; ++k) {
    const ss_ctx::PendFold f = c->deep_folds.front();
    c->deep_folds.pop_front();
    hipLaunchKernelGGL(ss::k_spec_combine, dim3((c->spec_n + 255) / 256), dim3(256), 0, c->s_fold, f.partial, f.tiles, c->spec_n, f.sum);
    c->slot_batch[f.slot] = batch;
  }
  (void)hipEventRecord(c->ev_fold_done[batch], c->s_fold);
  c->fold_last = batch;
  c->fold_dirty = true;
}

// Deep pipelining: the stages still owed, then the public stream waits for both side streams. Each stage goes to the queue
// its launch would have used (detect(j) behind launch j, emit(j) behind detect(j): the parity of `ready`), so the two queues
// drain side by side; a detect stage that reads the ring (the first of an overlapped run) has its successor wait, so then —
// and after a call that did not overlap at all — everything runs in order on the public stream instead.
void drain_deep(ss_ctx* c) {
  bool in_order = c->deep_L == 0;
  for (const auto& d : c->pd) in_order = in_order || (d.a.halo_psd == nullptr && !c->deep_ring_safe);
  const auto join = [&]() {
    for (int q = 0; q < c->nq; ++q) {
      (void)hipEventRecord(c->ev_join[q], c->s_ab[q]);
      (void)hipStreamWaitEvent(c->stream, c->ev_join[q], 0);
    }
  };
  if (in_order && c->deep_L > 0) join();
  const bool launched_on_queues = !in_order && c->deep_L > 0;
  for (int parity = 0; parity < (in_order ? 1 : c->nq); ++parity) {
    hipStream_t q = in_order ? c->stream : c->s_ab[parity];
    const auto mine = [&](long ready) { return in_order || (int)(ready % c->nq) == parity; };
    for (;;) {
      auto d = c->pd.begin();
      while (d != c->pd.end() && !mine(d->ready)) ++d;
      auto e = c->pe.begin();
      while (e != c->pe.end() && !mine(e->ready)) ++e;  // (an emit stage in the queue belongs to a detect stage launched earlier)
      const bool has_det = d != c->pd.end(), has_emit = e != c->pe.end();
      if (!has_det && !has_emit) break;
      ss_ctx::PendDet dd{};
      ss_ctx::PendEmit ee{};
      if (has_det) {
        dd = *d;
        c->pd.erase(d);
      }
      if (has_emit) {
        ee = *e;
        c->pe.erase(e);
      }
      launch_step(c, nullptr, has_det ? &dd.a : nullptr, dd.tiles, dd.spec, has_emit ? &ee.a : nullptr, q);
      if (has_det) c->pe.push_back(ss_ctx::PendEmit{dd.emit, dd.ready});
    }
  }
  if (!in_order) {
    // The join is a pair of barrier packets at the head of the public stream's queue, which is idle: the kernel behind them starts
    // 11-13 us after the stage they wait for has ended (scripts/ubench/sync_tail_lab.hip, profiles/r06/s18_timeline_k20_tail_all.txt).
    // Measured and not kept (drain_waiter_us, off): a WAITER in front of them — one wave on the public stream that sleeps until each
    // queue's last stage has said so in device memory (k_drain_signal, one wave behind it). In the lab, whose kernels are one wave
    // each, the same 20 launches + drain end 16 us earlier on the device with it; in the product nothing moves — the ring fill still
    // starts 14-19 us behind the last stage, and the two one-wave launches sit on the tail (profiles/r06/s19_summary.txt).
    // Correctness never rested on it: the barriers behind it are the join.
    const bool waiter = c->diag.drain_waiter_us > 0 && c->d_drain_done && launched_on_queues;
    if (waiter) {
      for (int q = 0; q < c->nq; ++q) hipLaunchKernelGGL(ss::k_drain_signal, dim3(1), dim3(64), 0, c->s_ab[q], c->d_drain_done);
      c->drain_signals += (unsigned)c->nq;
      hipLaunchKernelGGL(ss::k_drain_wait, dim3(1), dim3(64), 0, c->stream, (const unsigned*)c->d_drain_done, c->drain_signals, (long long)c->diag.drain_waiter_us * 100);
    }
    join();
  }
  // the spectrogram partial sums still waiting join their containers, and the public stream waits for all of them
  fold_spectrogram_slots(c, c->deep_folds.size(), true);
  if (c->fold_dirty) {
    (void)hipStreamWaitEvent(c->stream, c->ev_fold_done[c->fold_last], 0);
    c->fold_dirty = false;
  }
  {  // the ring rows the overlapped calls left to the drain (on the public stream, which by now waits for everything the two queues held)
    ss::RingFillArgs rf{};
    unsigned owed = 0;
    for (auto& o : c->ring_owed) {
      if (o.psd) {
        rf.psd_tail[owed] = o.psd + (size_t)(o.nframes - kHistRows) * c->n;
        rf.thr[owed] = o.thr;
        rf.hist_out[owed] = o.hist_out;
        ++owed;
      }
      o = ss_ctx::RingOwed{};
    }
    if (owed) hipLaunchKernelGGL(ss::k_ring_fill, dim3((unsigned)((size_t)kHistRows * c->n / 1024), owed), dim3(256), 0, c->stream, rf, c->n, kHistRows);
  }
  // (a device-wide synchronisation that finds an event behind every stream's last command waits for those; otherwise it puts a marker
  // of its own on each stream first: 27 against 17 us from the device's last word to the host's return in the lab)
  if (launched_on_queues && c->diag.drain_tail_event && c->ev_tail) (void)hipEventRecord(c->ev_tail, c->stream);
  c->deep_L = 0;
  c->deep_barrier = -10;
  c->deep_prev_ok = false;  // after a drain the caller may reuse its planes: the next call takes its rows from the ring
  c->deep_buffers.clear();
  c->deep_events_from = 0;
  c->input_events_from = 0;
}

// SS_FLAG_REFERENCE_NAN (reference_nan.h): the poison between a call's detect and emit stages, on the public stream.
void launch_nan_stage(ss_ctx* c, const NanStage& g) {
  if (!g.psd || g.nframes <= 0) return;
  hipLaunchKernelGGL(ss::k_nonfinite_scan, dim3(g.nframes), dim3(256), 0, c->stream, g.psd, c->n, g.n_learn, c->d_nf);
  hipLaunchKernelGGL(ss::k_nan_plan, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<ss::NanState*>(c->d_nan_state), (const int*)c->d_nf, g.nframes, c->n, g.pushed_before, c->d_bad_from);
  hipLaunchKernelGGL(ss::k_nan_apply, dim3(g.nframes), dim3(256), 0, c->stream, (const int*)c->d_bad_from, c->n, g.maskbits, g.counts, g.avg_full);
}

// The plan of the last call as a launch of its own (it would have ridden on the next call's column launch, ss_ctx::have_plan).
void launch_pending_plan(ss_ctx* c) {
  if (!c->have_plan) return;
  const int plan_wgs = ss::plan_blocks_of(c->pend_plan_det, c->pend_plan, c->n);  // (groups past the band's end find no column)
  hipLaunchKernelGGL((ss::k_plan_long<21, 21, kFusedTF, 256>), dim3(plan_wgs), dim3(256), 0, c->stream, c->pend_plan_det, c->pend_plan);
  c->have_plan = false;
}

// A launch has carried the planned detect stage and the emit stage that waited: the detect stage's emit waits now, and the detect
// stage whose plan has run (det_lag2 contexts; none elsewhere) is the planned one.
void shift_pending(ss_ctx* c) {
  c->have_emit = c->have_det;
  c->pend_emit = c->pend_det_emit;
  c->have_det = c->have_det2;
  if (c->have_det2) {
    c->pend_det = c->pend_det2;
    c->pend_det_tiles = c->pend_det2_tiles;
    c->pend_det_spec = c->pend_det2_spec;
    c->pend_det_emit = c->pend_det2_emit;
    c->have_det2 = false;
  }
  // one launch per call: the row stage that waited has run — the detect stage behind it waits for its plan now, and that plan is ready
  if (c->have_det3) {
    c->have_det2 = true;
    c->pend_det2 = c->pend_det3;
    c->pend_det2_tiles = c->pend_det3_tiles;
    c->pend_det2_spec = c->pend_det3_spec;
    c->pend_det2_emit = c->pend_det3_emit;
    c->have_det3 = false;
  }
  if (c->have_plan2) {  // (launch_step took the ready plan, if there was one)
    c->have_plan = true;
    c->pend_plan = c->pend_plan2;
    c->pend_plan_det = c->pend_plan2_det;
    c->have_plan2 = false;
  }
  c->have_rows = false;
}

// Drain the deferred stages: detect (+ the emit stage before it), then the last emit. Nothing is synchronised.
void flush_stages(ss_ctx* c) {
  if (c->deep) {
    if (c->deep_L > 0 || !c->pd.empty() || !c->pe.empty()) ++c->stats.drains;
    return drain_deep(c);
  }
  if (c->have_det || c->have_det2 || c->have_det3 || c->have_emit || c->have_rows) ++c->stats.drains;
  if (c->have_rows || c->have_plan2 || c->have_det3) {
    // one launch per call (KIND 7): row stage, ready plan, planned detect stage and emit stage of four different calls per launch until
    // nothing waits (a plan is a ROLE of these launches: a launch boundary orders it ahead of the detect stage it plans)
    while (c->have_rows || c->have_plan || c->have_plan2 || c->have_det || c->have_det2 || c->have_det3 || c->have_emit) {
      launch_step(c, nullptr, c->have_det ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, c->have_emit ? &c->pend_emit : nullptr, nullptr, true,
                  (c->have_rows && !c->rows1024x256) ? &c->pend_rows : nullptr, c->have_rows ? c->pend_rows_tiles : 0, (c->have_rows && c->rows1024x256) ? &c->pend_rows1024 : nullptr);
      shift_pending(c);
    }
    return;
  }
  launch_pending_plan(c);  // (the detect stage that waits reads its list)
  while (c->have_det || c->have_det2 || c->have_emit) {
    launch_step(c, nullptr, c->have_det ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, c->have_emit ? &c->pend_emit : nullptr);
    if (c->have_det && c->ref_nan) {  // (such contexts never defer a stage across calls: no emit stage rode on that launch)
      launch_nan_stage(c, c->nan_pending);
      c->nan_pending = NanStage{};
    }
    shift_pending(c);
  }
}

template <int FMT>
int launch_fft_fmt(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, float* d_psd) {
  if (c->use_fft8192) {  // (only reached without the fused back end; with it run_batch builds the step itself)
    const ss::Fft8192Args g = fft8192_args(c, d_iq, item_stride, d_psd);
    FftRole role;
    role.frames = &g;
    role.n = nframes;
    launch_step(c, &role, nullptr, 0, false, nullptr);
    return SS_OK;
  }
  switch (c->logn) {
    case 6: launch_lds<6, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 7: launch_lds<7, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 8: launch_lds<8, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 9: launch_lds<9, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 10:
    case 11:
    case 12:
      if (c->d_tw_small) {
        const int frames_per_wg = 32 >> (c->logn - 8);
        const dim3 grid((unsigned)((nframes + frames_per_wg - 1) / frames_per_wg));
        if (c->logn == 12)
          hipLaunchKernelGGL((ss::k_fft256xR_psd<FMT, 4>), grid, dim3(512), ss::kFft256xRLdsBytes, c->stream, d_iq, item_stride, nframes,
                             (const float*)c->d_win, (const float2*)c->d_tw256, (const float2*)c->d_tw_small, c->db_off, c->cfg.int_scale, d_psd);
        else if (c->logn == 11)
          hipLaunchKernelGGL((ss::k_fft256xR_psd<FMT, 3>), grid, dim3(512), ss::kFft256xRLdsBytes, c->stream, d_iq, item_stride, nframes,
                             (const float*)c->d_win, (const float2*)c->d_tw256, (const float2*)c->d_tw_small, c->db_off, c->cfg.int_scale, d_psd);
        else
          hipLaunchKernelGGL((ss::k_fft256xR_psd<FMT, 2>), grid, dim3(512), ss::kFft256xRLdsBytes, c->stream, d_iq, item_stride, nframes,
                             (const float*)c->d_win, (const float2*)c->d_tw256, (const float2*)c->d_tw_small, c->db_off, c->cfg.int_scale, d_psd);
      } else if (c->logn == 10) {
        launch_lds<10, FMT>(c, d_iq, item_stride, nframes, d_psd);
      } else if (c->logn == 11) {
        launch_lds<11, FMT>(c, d_iq, item_stride, nframes, d_psd);
      } else {
        launch_lds<12, FMT>(c, d_iq, item_stride, nframes, d_psd);
      }
      break;
    case 13: launch_lds<13, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 14: c->use_fft256 ? launch_four_step256<6, FMT>(c, d_iq, item_stride, nframes, d_psd) : launch_four_step<7, 7, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 15: c->use_fft256 ? launch_four_step256<7, FMT>(c, d_iq, item_stride, nframes, d_psd) : launch_four_step<7, 8, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 16: c->use_fft256 ? launch_four_step256<8, FMT>(c, d_iq, item_stride, nframes, d_psd) : launch_four_step<8, 8, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 17: c->use_fft256 ? launch_four_step256<9, FMT>(c, d_iq, item_stride, nframes, d_psd) : launch_four_step<8, 9, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 18: c->use_fft256 ? launch_four_step256<10, FMT>(c, d_iq, item_stride, nframes, d_psd) : launch_four_step<9, 9, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 19: c->use_fft256 ? launch_four_step256<11, FMT>(c, d_iq, item_stride, nframes, d_psd) : launch_four_step<9, 10, FMT>(c, d_iq, item_stride, nframes, d_psd); break;
    case 20:
      if (c->two_pass) {
        launch_cols1024_fmt<FMT>(c, d_iq, item_stride, nframes);
        launch_rows1024(c, nframes, d_psd, ss::RowsExtra{});
      } else if (c->use_fft256) {
        launch_four_step256<12, FMT>(c, d_iq, item_stride, nframes, d_psd);
      } else {
        launch_four_step<10, 10, FMT>(c, d_iq, item_stride, nframes, d_psd);
      }
      break;
    default: return fail(c, SS_ERR_INVALID, "fft_size 2^%d unsupported", c->logn);
  }
  return SS_OK;
}

int launch_fft(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, float* d_psd) {
  switch (c->cfg.in_format) {
    case SS_FMT_CF32: return launch_fft_fmt<ss::FMT_CF32>(c, d_iq, item_stride, nframes, d_psd);
    case SS_FMT_CS8: return launch_fft_fmt<ss::FMT_CS8>(c, d_iq, item_stride, nframes, d_psd);
    default: return launch_fft_fmt<ss::FMT_CU8>(c, d_iq, item_stride, nframes, d_psd);
  }
}

// Row half of the four-step transform whose column half ran as a role of k_scan_step.
int launch_fft_rows(ss_ctx* c, int nframes, float* d_psd, const ss::RowsExtra& rx) {
  constexpr int F = ss::FMT_CF32;  // (the rows read the work buffer, whatever the input format was)
  switch (c->logn) {
    case 14: launch_four_step256<6, F>(c, nullptr, 0, nframes, d_psd, false, rx); break;
    case 15: launch_four_step256<7, F>(c, nullptr, 0, nframes, d_psd, false, rx); break;
    case 16: launch_four_step256<8, F>(c, nullptr, 0, nframes, d_psd, false, rx); break;
    case 17: launch_four_step256<9, F>(c, nullptr, 0, nframes, d_psd, false, rx); break;
    case 18: launch_four_step256<10, F>(c, nullptr, 0, nframes, d_psd, false, rx); break;
    case 19: launch_four_step256<11, F>(c, nullptr, 0, nframes, d_psd, false, rx); break;
    case 20:
      if (c->two_pass) launch_rows1024(c, nframes, d_psd, rx);
      else launch_four_step256<12, F>(c, nullptr, 0, nframes, d_psd, false, rx);
      break;
    default: return fail(c, SS_ERR_INVALID, "fft_size 2^%d has no column / row split", c->logn);
  }
  return SS_OK;
}

int grid_for(size_t work_items, int block) {
  size_t g = (work_items + block - 1) / block;
  const size_t cap = 256 * 16;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

// Back end for groupings other than 21 x 21: one simple kernel per reference stage.
int run_backend_unfused(ss_ctx* c, const float* d_psd, int nframes, int n_learn, NoiseState* z, float* d_rel_out, float* d_avg_out,
                        int32_t* d_cand_off, int32_t* d_cand_idx, float* d_cand_avg, int cand_cap) {
  const int n = c->n;
  const int G = c->cfg.grouping_y;
  // Fold the previous batch into the ring: history = newest G-1 rows of [history ++ previous batch].
  // Done lazily, at the start of the next batch, so that ss_read_window can still address the ring rows
  // as they were before the batch it describes (Transmission::getBestIndex, transmission.cpp:132-154).
  if (c->rot_frames > 0 && G > 1) {
    const size_t cnt4 = (size_t)(G - 1) * n / 4;
    const float* src = c->d_rel + (size_t)c->rot_frames * n;
    if (c->rot_frames >= G - 1) {
      hipLaunchKernelGGL(ss::k_copy_rows, dim3(grid_for(cnt4, 256)), dim3(256), 0, c->stream, src, c->d_rel, cnt4);
    } else {  // source and destination overlap: bounce through a scratch copy
      hipLaunchKernelGGL(ss::k_copy_rows, dim3(grid_for(cnt4, 256)), dim3(256), 0, c->stream, src, c->d_hist_tmp, cnt4);
      hipLaunchKernelGGL(ss::k_copy_rows, dim3(grid_for(cnt4, 256)), dim3(256), 0, c->stream, (const float*)c->d_hist_tmp, c->d_rel, cnt4);
    }
  }
  c->rot_frames = 0;
  float* rel_batch = c->d_rel + (size_t)(G - 1) * n;
  if (n_learn > 0) {
    hipLaunchKernelGGL(ss::k_noise_learn, dim3((n + 255) / 256), dim3(256), 0, c->stream, d_psd, n, n_learn, z->d_thr);
  }
  hipLaunchKernelGGL(ss::k_noise_apply, dim3(grid_for((size_t)nframes * n / 4, 256)), dim3(256), 0, c->stream, d_psd, z->d_thr, n, nframes,
                     n_learn, rel_batch, d_rel_out);
  hipLaunchKernelGGL(ss::k_time_mean, dim3(grid_for((size_t)nframes * n / 4, 256)), dim3(256), 0, c->stream, c->d_rel, n, nframes, G,
                     c->frames_pushed, c->d_avgy);
  const int a = c->cfg.grouping_x / 2;
  const int blocks_per_row = (n + 255) / 256;
  hipLaunchKernelGGL(ss::k_freq_mean_detect, dim3(nframes * blocks_per_row), dim3(256), sizeof(float) * (256 + 2 * a), c->stream, c->d_avgy,
                     n, nframes, c->cfg.grouping_x, c->cfg.start_level, c->d_pass, c->d_avg2[0], d_avg_out, c->d_mask2[0]);
  const int wpr = n / 32;
  hipLaunchKernelGGL(ss::k_cand_count, dim3(nframes), dim3(256), 0, c->stream, c->d_mask2[0], wpr, c->d_counts);
  hipLaunchKernelGGL(ss::k_cand_scan, dim3(1), dim3(256), 0, c->stream, c->d_counts, nframes, c->d_off, d_cand_off);
  if (d_cand_idx && cand_cap > 0) {
    hipLaunchKernelGGL(ss::k_cand_write, dim3(nframes), dim3(256), 0, c->stream, c->d_mask2[0], wpr, n, c->d_off, c->d_avg2[0], cand_cap, d_cand_idx,
                       d_cand_avg);
  }
  c->rot_frames = nframes;
  c->last_avg = c->d_avg2[0];
  return SS_OK;
}

// Back end for the reference's grouping (21 x 21): a detect stage that reads the PSD plane once and an emit stage.
// Builds the two stages' arguments and advances the host-side state (ring window, counter rotation, buffer rotation);
// launches them at once (deferred = false: FFT sizes other than 8192) or leaves them in *det_out / *emit_out for the
// caller to schedule (k_scan_step).
// Where a batch of nframes reads the averager ring and where its newest rows go. Called by run_backend_fused, and before it by
// run_batch when the rows kernel of a long transform writes the ring rows itself: the second call finds the window where the
// first left it and gives the same answer.
struct RingPlace {
  const float* in;
  float* out;
  int next_start;
  float* batch;  // 2^20 points in two passes, batches of at least H frames: where ALL the batch's rows go when the rows kernel writes them as rel rows (null otherwise)
  int batch_row; // ring_place.h: RingDecision::batch
};
RingPlace place_ring(ss_ctx* c, int nframes) {
  const int n = c->n;
  constexpr int H = kHistRows;
  // (the decision is ring_place.h's — a function of five integers that tests/host/ring_check.cpp runs on the CPU; here its consequences)
  const ss::RingPrev prevs[2] = {c->hist_prev, c->hist_prev2};
  // how many earlier calls' spans to keep clear of: from the chain's launch schedule (ring_place.h; a context that runs the fold also runs
  // two-launch calls — learning frames, planes handed out — and protects what the deeper of its chains needs)
  const ss::RingChain chain = c->dif8 ? ss::RING_FOLD : c->merge ? ss::RING_MERGED : c->det_lag2 ? ss::RING_DET_LAG2 : ss::RING_ROWS_THEN_DETECT;
  const ss::RingDecision d = ss::ring_place_decide(c->hist_start, prevs, ss::ring_spans_to_protect(ss::ring_schedule(chain)), c->hist_rows, nframes, H, c->cull_long);
  if (d.shift_first) {  // (stream order: a deferred detect stage that still reads or writes this window goes first)
    flush_stages(c);
    if (c->hist_start != 0)
      hipLaunchKernelGGL(ss::k_hist_shift, dim3(grid_for((size_t)H * n, 256)), dim3(256), 0, c->stream,
                         (const float*)(c->d_hist + (size_t)c->hist_start * n), c->d_hist, n, H, 0);
    c->hist_start = 0;
    c->hist_prev = c->hist_prev2 = ss::RingPrev{0, -1, 0};
  }
  RingPlace r{};
  r.in = c->d_hist + (size_t)d.in * n;
  r.next_start = d.next_start;
  r.out = c->d_hist + (size_t)d.next_start * n;
  r.batch = (d.batch >= 0 && nframes >= H) ? c->d_hist + (size_t)d.batch * n : nullptr;
  r.batch_row = d.batch;
  return r;
}

int run_backend_fused(ss_ctx* c, const float* d_psd, int nframes, int n_learn, NoiseState* z, SpecState* spec, float* d_rel_out,
                      float* d_avg_out, int32_t* d_cand_off, int32_t* d_cand_idx, float* d_cand_avg, int cand_cap, bool deferred,
                      ss::DetectArgs* det_out, int* det_tiles_out, ss::EmitArgs* emit_out) {
  const int n = c->n;
  constexpr int G = 21, GX = 21, TF = kFusedTF, TB = 256;  // measured against 32-frame and 128-bin tiles: 16 x 256 is fastest
  constexpr int H = kHistRows;
  static_assert(H == ss::DetectTile<G, GX, TF, TB>::H, "ring depth");
  const RingPlace ring = place_ring(c, nframes);
  const float* hist_in = ring.in;
  const int next_start = ring.next_start;
  float* hist_out = ring.out;
  const int cur = c->cnt_cur, clr = (c->cnt_cur + c->ncnt - c->lag) % c->ncnt;
  int* counts = c->d_cnt3[cur];
  const int b = c->buf_cur;
  const bool keep_planes = (c->cfg.flags & SS_FLAG_KEEP_PLANES) != 0;
  float* avg_full = d_avg_out ? d_avg_out : (keep_planes ? c->d_avg2[b] : nullptr);
  const int shift = (int)(c->abs_frames % TF);
  const int tiles = ((nframes + shift + TF - 1) / TF) * ((n + TB - 1) / TB);
  ss::DetectArgs da{};
  da.psd = d_psd;
  da.thr = z->d_thr;
  da.hist_in = hist_in;
  da.hist_out = hist_out;
  da.n = n;
  da.nframes = nframes;
  da.n_learn = n_learn;
  da.pushed_before = c->frames_pushed;
  da.shift = shift;
  da.start_level = c->cfg.start_level;
  da.pass = c->d_pass;
  da.maskbits = c->d_mask2[b];
  da.counts = counts;
  da.rel_out = d_rel_out;
  da.avg_out = avg_full;
  da.avg_sparse = c->d_avg2[b];
  da.segsum = (c->cull && !spec) ? c->d_segsum[b] : nullptr;
  da.seg_pitch = c->cfg.max_batch;
  da.thr_tilemin = z->d_thr + n;  // (kept behind the ceiling itself, get_noise)
  da.live = c->d_live[b];
  da.stats = c->d_stats;
  c->stats.tiles_total += (unsigned long long)tiles;
#ifdef SS_DIAG
  da.cull_stats = c->diag.d_cull_stats;
#endif
  if (spec) {
    da.spec_partial = c->d_spec_part2[c->spec_cur];
    da.spec_m = c->spec_m;
    da.spec_n = c->spec_n;
    if (c->spec_pending_sum) {
      da.spec_prev_partial = c->d_spec_part2[c->spec_cur ^ 1];
      da.spec_prev_sum = c->spec_pending_sum;
      da.spec_prev_tiles = c->spec_pending_tiles;
    }
    c->spec_pending_sum = spec->d_sum;  // (the containers live until the context is destroyed)
    c->spec_pending_tiles = (nframes + shift + TF - 1) / TF;
    c->spec_cur ^= 1;
  }
  ss::EmitArgs ea{};
  ea.maskbits = c->d_mask2[b];
  ea.words_per_row = n / 32;
  ea.n = n;
  ea.nframes = nframes;
  ea.counts = counts;
  ea.counts_clear = c->d_cnt3[clr];
  ea.clear_n = c->cnt_frames[clr];
  ea.avg = avg_full ? avg_full : c->d_avg2[b];
  ea.cap = cand_cap;
  ea.off_int = c->d_off4[b];
  ea.live_clear = da.live;
  ea.clear_masks = (c->cull || c->cull_long) ? 1 : 0;  // (every emit stage of a culling context: the mask buffers are all zero between uses)
  c->d_off = c->d_off4[b];
  ea.off_out = d_cand_off;
  ea.cand_idx = (d_cand_idx && cand_cap > 0) ? d_cand_idx : nullptr;
  ea.cand_avg = d_cand_avg;
  c->cnt_frames[cur] = nframes;  // this buffer now holds nframes counts (read by the emit stage, zeroed two batches on)
  c->cnt_frames[clr] = 0;
  c->cnt_cur = (cur + 1) % c->ncnt;
  c->buf_cur = (b + 1) % c->nbuf;
  c->last_avg = ea.avg;
  c->last_hist = hist_in;
  c->hist_prev2 = c->hist_prev;
  c->hist_prev = ss::RingPrev{c->hist_start, c->cull_long ? ring.batch_row : -1, nframes};  // (what this call's detect stage reads: ring_place.h)
  c->hist_start = next_start;
  c->last_n_learn = n_learn;
  c->last_thr = z->d_thr;
  NanStage ns;
  if (c->ref_nan) {
    ns.psd = d_psd;
    ns.nframes = nframes;
    ns.n_learn = n_learn;
    ns.pushed_before = da.pushed_before;
    ns.maskbits = da.maskbits;
    ns.counts = counts;
    ns.avg_full = avg_full;
  }
  if (deferred) {
    *det_out = da;
    *det_tiles_out = tiles;
    *emit_out = ea;
    c->nan_pending = ns;  // (flush_stages puts it between the two)
    return SS_OK;
  }
  if (spec) hipLaunchKernelGGL((ss::k_detect_fused<G, GX, TF, TB, true>), dim3(tiles), dim3(TB), 0, c->stream, da);
  else hipLaunchKernelGGL((ss::k_detect_fused<G, GX, TF, TB, false>), dim3(tiles), dim3(TB), 0, c->stream, da);
  if (c->ref_nan) launch_nan_stage(c, ns);
  // long rows: several waves per frame (slices of at least 256 mask words)
  const int words = n / 32;
  if (c->diag.emit_wide && words >= 2048) hipLaunchKernelGGL(ss::k_cand_emit_wide<8>, dim3(nframes), dim3(512), 0, c->stream, ea);
  else if (c->diag.emit_wide && words >= 1024) hipLaunchKernelGGL(ss::k_cand_emit_wide<4>, dim3(nframes), dim3(256), 0, c->stream, ea);
  else if (c->diag.emit_wide && words >= 512) hipLaunchKernelGGL(ss::k_cand_emit_wide<2>, dim3(nframes), dim3(128), 0, c->stream, ea);
  else hipLaunchKernelGGL(ss::k_cand_emit, dim3(nframes), dim3(64), 0, c->stream, ea);
  return SS_OK;
}

// Spectrogram::work/process for a batch (spectrogram.cpp:29-60): the container of the current centre frequency
// accumulates the bin-decimated raw PSD of every frame.
// m_containers[frequency] (spectrogram.cpp:33-37): the accumulator of the current centre frequency, created on first use
SpecState* spectrogram_container(ss_ctx* c) {
  const int32_t center = (c->range_lo + c->range_hi) / 2;
  for (auto& s : c->spec)
    if (s.center == center) return &s;
  SpecState ns;
  ns.center = center;
  if (hipMalloc(&ns.d_sum, sizeof(float) * (size_t)c->spec_n) != hipSuccess) return nullptr;
  if (hipMemsetAsync(ns.d_sum, 0, sizeof(float) * (size_t)c->spec_n, c->stream) != hipSuccess) {
    (void)hipFree(ns.d_sum);
    return nullptr;
  }
  c->spec.push_back(ns);
  return &c->spec.back();
}

// In-detect form: add the partial sums of the last launch to their container now (the next launch would have done it).
void spectrogram_flush(ss_ctx* c) {
  if (!c->spec_pending_sum) return;
  hipLaunchKernelGGL(ss::k_spec_combine, dim3((c->spec_n + 255) / 256), dim3(256), 0, c->stream, (const float*)c->d_spec_part2[c->spec_cur ^ 1],
                     c->spec_pending_tiles, c->spec_n, c->spec_pending_sum);
  c->spec_pending_sum = nullptr;
}

// Stand-alone form (groupings other than 21 x 21, decimation factors above 256): two more kernels over the PSD plane.
int spectrogram_accumulate(ss_ctx* c, SpecState* g, const float* d_psd, int nframes) {
  constexpr int kChunk = 32;
  const int nchunks = (nframes + kChunk - 1) / kChunk;
  const dim3 grid((c->spec_n + 255) / 256, nchunks);
  hipLaunchKernelGGL(ss::k_spec_partial, grid, dim3(256), 0, c->stream, d_psd, c->n, nframes, c->spec_m, c->spec_n, kChunk, c->d_spec_partial);
  hipLaunchKernelGGL(ss::k_spec_combine, dim3((c->spec_n + 255) / 256), dim3(256), 0, c->stream, (const float*)c->d_spec_partial, nchunks,
                     c->spec_n, g->d_sum);
  return SS_OK;
}

#ifdef SS_DIAG
// (input canary, Diag::d_canary) position-weighted sum of `nwords` 32-bit words: one workgroup, result in *out
__global__ __launch_bounds__(1024) void k_iq_checksum(const uint32_t* __restrict__ words, size_t nwords, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long part[16];
  unsigned long long acc = 0;
  for (size_t i = threadIdx.x; i < nwords; i += 1024) acc += (unsigned long long)words[i] * (unsigned long long)(2 * (i & 0xffff) + 1);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int k = 0; k < 16; ++k) t += part[k];
    *out = t;
  }
}
// the last kHistRows frames of a call's input, as the words the checksum runs over
static void canary_sum(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, hipStream_t q, unsigned long long* out) {
  const size_t bps = in_bytes_per_sample(c->cfg.in_format);
  const size_t off = (size_t)(nframes - kHistRows) * (size_t)item_stride * bps;
  const size_t bytes = ((size_t)(kHistRows - 1) * (size_t)item_stride + (size_t)c->n) * bps;  // (up to the end of the last frame's N samples)
  hipLaunchKernelGGL(k_iq_checksum, dim3(1), dim3(1024), 0, q, reinterpret_cast<const uint32_t*>(static_cast<const char*>(d_iq) + off), bytes / 4, out);
}
#endif

// One call on a context with deep pipelining (ss_ctx::deep): launch L = FFT(L) + detect(L - 2) + emit(L - 4) on queue L & 1, or
// — learning frames, short calls, callers that wait after every call — the three stages in order on the public stream.
int run_call_deep(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, int n_learn, NoiseState* z, SpecState* spec, float* d_psd,
                  float* d_psd_out, float* d_rel_out, float* d_avg_out, int32_t* d_cand_off, int32_t* d_cand_idx, float* d_cand_avg, int cand_cap,
                  bool allow_overlap) {
  int st = SS_OK;
  const ss::Fft8192Args g = fft8192_args(c, d_iq, item_stride, d_psd);
  FftRole role;
  role.frames = &g;
  role.n = nframes;
  if (allow_overlap && ++c->deep_calls_since_sync >= 2) c->deep_eager = false;
  const bool overlap = allow_overlap && !c->deep_eager && !c->deep_iq_recycled && c->diag.pipeline && n_learn == 0 && nframes >= kHistRows;
  const size_t plane_bytes = sizeof(float) * (size_t)nframes * (size_t)c->n;
  const auto clash = [](const void* a, size_t abytes, const void* b, size_t bbytes) {
    const char *pa = static_cast<const char*>(a), *pb = static_cast<const char*>(b);
    return a && b && abytes && bbytes && pa < pb + bbytes && pb < pa + abytes;
  };
  ss_ctx::Buffers mine{{d_psd_out, d_rel_out, d_avg_out, d_cand_off, d_cand_idx, d_cand_avg},
                       {plane_bytes, plane_bytes, plane_bytes, sizeof(int32_t) * ((size_t)nframes + 1), sizeof(int32_t) * (size_t)cand_cap, sizeof(float) * (size_t)cand_cap},
                       0, ~(uintptr_t)0, 0, 0, 0};
  mine.iq_lo = reinterpret_cast<uintptr_t>(d_iq);
  mine.iq_hi = mine.iq_lo + (size_t)nframes * (size_t)item_stride * in_bytes_per_sample(c->cfg.in_format);
  for (int x = 0; x < 6; ++x)
    if (mine.p[x] && mine.bytes[x]) {
      const uintptr_t a0 = reinterpret_cast<uintptr_t>(mine.p[x]);
      mine.lo = std::min(mine.lo, a0);
      mine.hi = std::max(mine.hi, a0 + mine.bytes[x]);
    }
  // how this call's launches relate to the stages of earlier calls that touch the same buffers (ss_ctx::Buffers)
  bool must_drain = !overlap, wait_other_queue = false;
  if (overlap) {
    const long L = c->deep_L;
    // A caller that hands in frames where the frames of a call still in flight lie — one input buffer filled again and again
    // — has broken the contract (include/specscan.h: every buffer untouched until ss_sync), and the library cannot undo
    // that: the producer that refilled the buffer ran on ss_stream, which overlapped launches do not hold up. What it can
    // do is stop trusting this caller's input buffers: drain now, and from here on take the stages in order on the public
    // stream, where a producer enqueued there is ordered against them (deep_eager, like a caller that waits after every call).
    // (A caller that uses ss_input_wait has taken the matter into its own hands: its refills are ordered behind the launches
    // that read the buffer, and rotating a few buffers is exactly what it is there for.)
    for (const auto& b : c->deep_buffers)
      if (!c->input_events && b.launch + 1 >= L - 2 * c->nq && b.iq_lo < mine.iq_hi && mine.iq_lo < b.iq_hi) {
        must_drain = true;
        if (!c->deep_iq_recycled) ++c->stats.demotions;
        c->deep_iq_recycled = true;
      }
    for (const auto& b : c->deep_buffers)
      if (b.lo < mine.hi && mine.lo < b.hi)
      for (int x = 0; x < 6 && !must_drain; ++x)
        for (int y = 0; y < 6 && !must_drain; ++y)
          if (clash(mine.p[x], mine.bytes[x], b.p[y], b.bytes[y])) {
            const long last = b.launch + (y < 2 ? c->nq : 2 * c->nq);  // the last launch that touches b.p[y]
            if (last > L - c->nq) must_drain = true;                   // too close: not even stream order helps
            else if ((L - last) % c->nq == 0) continue;                // same queue, earlier: stream order
            else if (c->deep_events && L - 2 * c->nq + 1 >= c->deep_events_from) wait_other_queue = true;  // (last <= L - nq - 1: each other queue's latest launch but one is at or behind it)
            else must_drain = true;
            if ((L - last) % c->nq != 0) c->deep_events = true;        // a caller rotating a number of sets that is no multiple of nq: keep events from now on
          }
  }
  if (spec) {
    // The slot this call's spectrogram partial sums will take is free once the additions of its last user, kDeepSpecSlots
    // calls ago, have run. A host that enqueues faster than the device works gets that far ahead: it waits here (the
    // device still has dozens of calls queued; a wait packet in a queue would cost every call ~10 us instead).
    const int b = c->slot_batch[c->spec_ring_next];
    if (b >= 0) SS_HIP(c, hipEventSynchronize(c->ev_fold_done[b]));
    if ((int)c->deep_folds.size() >= kDeepSpecSlots) must_drain = true;
  }
  if (must_drain) flush_stages(c);
  hipStream_t q = c->stream;
  ss_ctx::PendDet d{};
  ss_ctx::PendEmit e{};
  bool has_det = false, has_emit = false;
  long L = -1;
  ++(overlap ? c->stats.calls_overlapped : c->stats.calls_in_order);
  if (overlap) {
    L = c->deep_L++;
    q = c->s_ab[L % c->nq];
    // whatever the public stream holds (the caller's producers, a drain, a learning call) comes first
    if (hipStreamQuery(c->stream) != hipSuccess) {
      hipEvent_t ev = c->ev_in[c->deep_forks++ & 3];
      SS_HIP(c, hipEventRecord(ev, c->stream));
      SS_HIP(c, hipStreamWaitEvent(q, ev, 0));
    }
    // The two queues never wait for each other otherwise, so nothing bounds how far one may run ahead: once in
    // kDeepSyncPeriod launches each waits for the other's launch three back. Whatever a launch older than kDeepHorizon
    // touched is then finished for both queues, and the buffers of older calls need no tracking.
    const int phase = (int)(L % kDeepSyncPeriod);
    if ((wait_other_queue && !must_drain) || (phase >= kDeepSyncPhase && phase < kDeepSyncPhase + c->nq))
      for (long other = L - c->nq - 1; other > L - 2 * c->nq; --other)  // one launch of every other queue, each at least nq + 1 back
        if (other >= 0) SS_HIP(c, hipStreamWaitEvent(q, c->ev_launch[other & 31], 0));
    for (long prev = L - 1; prev > L - c->nq; --prev)
      if (c->deep_barrier == prev) SS_HIP(c, hipStreamWaitEvent(q, c->ev_launch[prev & 31], 0));  // a detect role that read the ring goes before the next ones write it (once per drain)
    if (c->deep_prev_ok) {  // this call's detect stage will want the rows before the batch: the previous call's last frames, once more
      // (20 rows for the 21-frame mean, up to 15 more when the batch does not start on a tile boundary: two launch shapes)
      role.n_halo = c->abs_frames % kFusedTF == 0 ? c->cfg.grouping_y - 1 : kHistRows;
      role.halo_iq = static_cast<const char*>(c->deep_prev_iq) +
                     (size_t)(c->deep_prev_frames - role.n_halo) * (size_t)c->deep_prev_stride * in_bytes_per_sample(c->cfg.in_format);
      role.halo_psd = c->d_halo[L % (2 * c->nq)];
#ifdef SS_DIAG
      if (c->diag.d_canary && c->diag.canary_prev_slot >= 0) {  // the frames this launch reads once more: are they what the launch before saw?
        canary_sum(c, c->deep_prev_iq, c->deep_prev_stride, c->deep_prev_frames, q, c->diag.d_canary + 2 * c->diag.canary_prev_slot + 1);
        c->diag.canary_pairs.push_back(c->diag.canary_prev_slot);
      }
#endif
    }
    if (!c->pd.empty() && c->pd.front().ready <= L) {
      d = c->pd.front();
      c->pd.pop_front();
      has_det = true;
    }
    if (!c->pe.empty() && c->pe.front().ready <= L) {
      e = c->pe.front();
      c->pe.pop_front();
      has_emit = true;
    }
  }
  launch_step(c, &role, has_det ? &d.a : nullptr, d.tiles, d.spec, has_emit ? &e.a : nullptr, q);
#ifdef SS_DIAG
  c->diag.canary_prev_slot = -1;
  if (overlap && c->diag.d_canary && nframes >= kHistRows && c->diag.canary_pairs.size() < 60) {
    c->diag.canary_prev_slot = (int)(c->diag.canary_seq++ & 63);
    canary_sum(c, d_iq, item_stride, nframes, q, c->diag.d_canary + 2 * c->diag.canary_prev_slot);
    SS_HIP(c, hipEventRecord(c->diag.ev_canary, q));
    SS_HIP(c, hipStreamWaitEvent(c->stream, c->diag.ev_canary, 0));
  }
#endif
  if (overlap) {
    const bool ring_reader = has_det && !d.a.halo_psd && !c->deep_ring_safe;
    const int rec_phase = (int)(L % kDeepSyncPeriod);
    if (c->deep_events || c->input_events || ring_reader || (rec_phase >= kDeepSyncPhase - 2 * c->nq && rec_phase < kDeepSyncPhase)) SS_HIP(c, hipEventRecord(c->ev_launch[L & 31], q));
    if (has_det) c->pe.push_back(ss_ctx::PendEmit{d.emit, L + c->nq});
    if (ring_reader) c->deep_barrier = L;
  }
  if (n_learn > 0) {
    hipLaunchKernelGGL(ss::k_noise_learn, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, (const float*)d_psd, c->n, n_learn, z->d_thr);
    if (c->cull || c->cull_long) hipLaunchKernelGGL(ss::k_thr_tilemin, dim3(c->n / 256), dim3(64), 0, c->stream, (const float*)z->d_thr, c->n, z->d_thr + c->n);
  }
  ss_ctx::PendDet mine_det{};
  st = run_backend_fused(c, d_psd, nframes, n_learn, z, nullptr, d_rel_out, d_avg_out, d_cand_off, d_cand_idx, d_cand_avg, cand_cap, true,
                         &mine_det.a, &mine_det.tiles, &mine_det.emit);
  if (st != SS_OK) return st;
  if (role.n_halo) {
    mine_det.a.halo_psd = role.halo_psd;
    mine_det.a.halo_rows = role.n_halo;
    mine_det.a.halo_segsum = (c->cull && c->diag.halo_maxima && mine_det.a.segsum) ? halo_segsum_of(c, role.halo_psd) : nullptr;  // (this launch's FFT role leaves them)
  }
#ifndef SS_RING_AT_DRAIN  // (A/B builds, scripts/build_ab.py: 0 = the ring rows by the tiles of every call, as until round 3)
#define SS_RING_AT_DRAIN 1
#endif
  if (overlap && SS_RING_AT_DRAIN) {  // this call's ring rows: at the next drain (nframes >= kHistRows here: all of them come from this call's plane)
    mine_det.a.hist_by_fft = 1;
    c->ring_owed[0] = c->ring_owed[1];
    c->ring_owed[1].psd = d_psd;
    c->ring_owed[1].nframes = nframes;
    c->ring_owed[1].hist_out = mine_det.a.hist_out;
    c->ring_owed[1].thr = z->d_thr;
  }
  if (spec) {  // Spectrogram::work (spectrogram.cpp:45-60): this call's frames, bin-decimated, summed per frame tile into a slot of their own
    const int slot_no = c->spec_ring_next;
    float* slot = c->d_spec_ring + (size_t)slot_no * c->spec_slot_floats;
    c->slot_batch[slot_no] = -1;
    c->spec_ring_next = (c->spec_ring_next + 1) % kDeepSpecSlots;
    mine_det.a.spec_partial = slot;
    mine_det.a.spec_m = c->spec_m;
    mine_det.a.spec_n = c->spec_n;
    mine_det.spec = true;
    c->deep_folds.push_back(ss_ctx::PendFold{slot, (nframes + mine_det.a.shift + kFusedTF - 1) / kFusedTF, spec->d_sum, slot_no, L + c->nq});
    spec->count += nframes;
    if (overlap) {  // calls whose detect stage is in a launch already enqueued: added in batches, beside the pipeline
      size_t ready = 0;
      while (ready < c->deep_folds.size() && c->deep_folds[ready].det_launch < c->deep_L) ++ready;
      if (ready >= (size_t)kDeepFoldBatch) fold_spectrogram_slots(c, ready, false);
    }
  }
  mine_det.ready = L + c->nq;
  c->pd.push_back(mine_det);
  if (overlap) {
    c->deep_prev_ok = true;
    c->deep_prev_iq = d_iq;
    c->deep_prev_stride = item_stride;
    c->deep_prev_frames = nframes;
    mine.launch = L;
    c->deep_buffers.push_back(mine);
    if (c->deep_buffers.size() > (size_t)kDeepHorizon) c->deep_buffers.pop_front();
  } else {
    flush_stages(c);
  }
  return SS_OK;
}

// The window's newest ring_db_rows rows hold dB values (the FFT stage of the detect-mode calls before left them so): subtract the ceiling
// they were formed under, in place — rel = dB - ceiling, the fp32 subtraction the rows kernel or a detect tile would have made on the
// same values (noise_learner.cpp:55) — after which every row of the window is noise-relative again. What waits is drained first (its
// tiles read the rows as they are).
__global__ void k_rows_sub_thr(float* __restrict__ rows, const float* __restrict__ thr, int n, int nrows, int logq) {
  const size_t total = (size_t)nrows * (size_t)n;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int pos = (int)(i % (size_t)n);
    const int bin = logq ? ss::dif_offset_bin(pos, logq) : pos;  // (the fold's rows: fft65536_dif8.h)
    rows[i] -= thr[bin];
  }
}
void settle_ring_db(ss_ctx* c) {
  if (c->ring_db_rows <= 0 || !c->ring_db_thr) {
    c->ring_db_rows = 0;
    return;
  }
  flush_stages(c);
  float* first = c->d_hist + (size_t)(c->hist_start + kHistRows - c->ring_db_rows) * (size_t)c->n;
  hipLaunchKernelGGL(k_rows_sub_thr, dim3(1024), dim3(256), 0, c->stream, first, c->ring_db_thr, c->n, c->ring_db_rows, c->ring_perm8 ? c->dif_logq : 0);
  c->last_settled_lo = first;  // (run_batch forgets this again: a new call's rows are what last_rows_db / last_db_from say)
  c->last_settled_hi = first + (size_t)c->ring_db_rows * (size_t)c->n;
  c->ring_db_rows = 0;
  c->ring_db_thr = nullptr;
}

// ... and the averager ring's rows are in the fold's blocked order while calls go through the fold, in bin order while they take the
// four-step form: a call of the other kind has what waits drained (those stages read the rows as they are), the window's 35 rows
// rewritten at the front of the buffer, and no tile tested whose rows reach back across the change (the run maxima of the frames
// before are in the other form's layout).
void set_ring_form(ss_ctx* c, bool perm8) {
  if (!c->dif8 || c->ring_perm8 == perm8) return;
  flush_stages(c);
  const size_t cnt = (size_t)kHistRows * (size_t)c->n;
  hipLaunchKernelGGL(ss::k_rows_perm8, dim3(1024), dim3(256), 0, c->stream, (const float*)(c->d_hist + (size_t)c->hist_start * c->n), c->d_perm_tmp, kHistRows, perm8 ? 1 : 0, c->dif_logq);
  hipLaunchKernelGGL(ss::k_copy_rows, dim3(grid_for(cnt / 4, 256)), dim3(256), 0, c->stream, (const float*)c->d_perm_tmp, c->d_hist, cnt / 4);
  c->hist_start = 0;
  c->hist_prev = c->hist_prev2 = ss::RingPrev{0, -1, 0};
  c->clean_abs = c->abs_frames;
  c->ring_perm8 = perm8;
}

// The chain for one batch, everything on c->stream, nothing synchronised.
// n_learn = leading frames that belong to the noise-learning phase (decided by the caller).
// 8192 points with the fused back end (c->step_path): the call's FFT stage is launched together with the deferred detect
// stage of the previous call and the emit stage of the one before (scan_step.h); this call's own detect and emit stages
// stay deferred until the next call or flush_stages.
int run_batch(ss_ctx* c, const void* d_iq, long long item_stride, int nframes, int n_learn, NoiseState* z, float* d_psd_out,
              float* d_rel_out, float* d_avg_out, int32_t* d_cand_off, int32_t* d_cand_idx, float* d_cand_avg, int cand_cap,
              bool allow_overlap = false) {
  const int G = c->cfg.grouping_y;
  if (c->pass_dirty) {
    flush_stages(c);  // (ss_set_frequency_range drained them already; the mask a deferred stage reads must not change under it)
    std::vector<uint8_t> pass;
    build_pass_mask(c, pass);
    SS_HIP(c, hipMemcpyAsync(c->d_pass, pass.data(), pass.size(), hipMemcpyHostToDevice, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));  // `pass` is pageable and dies at scope end
    c->pass_dirty = false;
  }
  float* d_psd = d_psd_out ? d_psd_out : c->d_psd2[c->psd_cur];
  if (!d_psd_out) c->psd_cur = (c->psd_cur + 1) % c->npsd;
  c->prof_call = false;
  c->prof_launch_frames = nframes;
  int st = SS_OK;
  const float* ring_only_rows = nullptr;  // set by a call that writes no dB plane (2^20 points, detect mode, shorter than the ring)
  SpecState* spec = nullptr;
  if (c->spec_n > 0) {
    spec = spectrogram_container(c);
    if (!spec) return fail(c, SS_ERR_NOMEM, "spectrogram container");
  }
  if (c->deep) {
    st = run_call_deep(c, d_iq, item_stride, nframes, n_learn, z, spec, d_psd, d_psd_out, d_rel_out, d_avg_out, d_cand_off, d_cand_idx, d_cand_avg, cand_cap,
                       allow_overlap);
    if (st != SS_OK) return st;
  } else if (c->step_path) {
    ss::Fft8192Args g{};
    ss::ColsArgs gc{};
    FftRole role;
    if (c->use_fft8192) {
      g = fft8192_args(c, d_iq, item_stride, d_psd);
      role.frames = &g;
      role.n = nframes;
    } else {  // N = 256 x N2 (or 1024 x 1024 at 2^20 points): the column half here, the row half right behind it
      gc = cols256_args(c, d_iq, item_stride);
      role.cols = &gc;
      role.n = nframes * (c->n >> 13);  // tiles of 8192 points either way: 32 columns x 256 rows, or 8 x 1024
    }
    // Tile culling for long transforms: the rows kernel leaves the run maxima of every frame and, in a call without learning
    // frames, writes the ring rows of the batch itself (placed now: moving the ring window drains the deferred stages).
    // 65536 points, int8 IQ: a call that keeps no plane goes through the radix-8 fold (KIND 8) — decided before the ring is placed,
    // because a change of form rewrites the window
    // (one predicate for the three decisions below — fold, dB rows in the ring, no dB plane at all: a device call without learning frames
    // that hands out no plane and keeps none; and the room for a whole batch's rows in the ring's buffer is ring_place.h's `whole` branch,
    // a function of the sizes alone, so a fold call knows before the window is rewritten that its rows have a place — a call that has
    // none takes the four-step form)
    const bool keeps_no_plane = allow_overlap && n_learn == 0 && !d_psd_out && !d_rel_out && !d_avg_out && !spec && !c->ref_nan &&
                                !(c->cfg.flags & SS_FLAG_KEEP_PLANES) && c->diag.ring_only;
    const bool ring_room = nframes < kHistRows || kHistRows + nframes <= c->hist_rows;
    const bool dif_call = c->dif8 && c->cull_long && c->diag.pipeline && keeps_no_plane && ring_room && !(c->cfg.flags & (SS_FLAG_STREAM_ORDERED | SS_FLAG_REFERENCE_NAN));
    // ... and the rows the FFT stage of a detect-mode call leaves in the ring are dB values (ring_db_rows): a call of any other kind, and
    // a call under another ceiling, has them turned into noise-relative rows first
    const bool db_call = c->cull_long && (!c->cull_fold_only || dif_call) && keeps_no_plane;  // (what ring_only below comes to, but for the room in the ring's buffer)
    if (c->ring_db_rows > 0 && (!db_call || c->ring_db_thr != z->d_thr)) settle_ring_db(c);
    set_ring_form(c, dif_call);
    ss::RowsExtra rx{};
    bool ring_by_rows = false, ring_only = false;
    const float* ring_rows = nullptr;
    const bool cull_call = c->cull_long && (!c->cull_fold_only || dif_call);  // (131072 points: only the fold's calls leave run maxima and write the ring themselves)
    if (cull_call) {
      rx.smax = c->d_smax;
      rx.smax_mask = c->smax_rows - 1;
      rx.abs0 = (int)(c->abs_frames & 0x3fffffff);
      if (n_learn == 0) {
        const RingPlace rp = place_ring(c, nframes);
        rx.thr = z->d_thr;
        rx.hist_out = rp.out;
        rx.first_hist = nframes - kHistRows;
        rx.zero_word = c->d_tlist[c->buf_cur];  // (the list this call's plan appends to; its last reader was the detect stage of the call before last)
        ring_by_rows = true;
        // no dB plane at all: a device call that hands out no plane, shorter than the ring, whose rows the new rows kernel writes
        ring_only = keeps_no_plane && (nframes < kHistRows || rp.batch);
        if (ring_only && nframes < kHistRows) {
          ring_rows = rp.in + (size_t)kHistRows * c->n;  // batch frame f = row H + f of the window being read: right behind it (place_ring)
        } else if (ring_only) {  // every frame of the batch as a row of the region place_ring reserved; its last H rows are the next call's ring
          rx.hist_out = rp.batch;
          rx.first_hist = 0;
          ring_rows = rp.batch;
        }
        if (ring_only) rx.thr = nullptr;  // ... as dB values: the tiles that are evaluated subtract the ceiling (ring_db_rows)
      }
    }
    if (!ring_only && c->ring_db_rows > 0) settle_ring_db(c);  // (a call that was to keep no plane and found no room for its rows in the ring's buffer)
    // (SS_FLAG_STREAM_ORDERED / SS_FLAG_REFERENCE_NAN: every stage of the call before the call returns, in order on the public stream)
    const bool overlap = c->diag.pipeline && n_learn == 0 && !(c->cfg.flags & (SS_FLAG_STREAM_ORDERED | SS_FLAG_REFERENCE_NAN));
    ++c->stats.calls_in_order;
    // A caller that hands the same PSD or avg plane to consecutive calls would have this call's stages write what a
    // deferred stage of the previous call still has to read in the same launch: drain first (no overlap for such callers).
    const size_t plane_bytes = sizeof(float) * (size_t)nframes * (size_t)c->n;
    const auto clash = [](const void* a, size_t abytes, const void* b, size_t bbytes) {
      const char *pa = static_cast<const char*>(a), *pb = static_cast<const char*>(b);
      return a && b && pa < pb + bbytes && pb < pa + abytes;
    };
    const auto clashes = [&](const ss::DetectArgs& pd, const ss::EmitArgs& pe) {
      const size_t pend_bytes = sizeof(float) * (size_t)pd.nframes * (size_t)c->n;
      return clash(d_psd, plane_bytes, pd.psd, pend_bytes) || clash(d_avg_out, plane_bytes, pe.avg, pend_bytes) || clash(d_psd, plane_bytes, pd.rel_out, pend_bytes) ||
             clash(d_rel_out, plane_bytes, pd.psd, pend_bytes);
    };
    const bool reused = (c->have_det && clashes(c->pend_det, c->pend_det_emit)) || (c->have_det2 && clashes(c->pend_det2, c->pend_det2_emit));
    if (!overlap || reused) flush_stages(c);
    const bool rows_by_step = (c->two_pass && c->diag.cols1024_wide) || c->rows256_step || c->rows1024x256;
    // one launch per call (SS_MERGE_65536): calls that keep no dB plane, in one piece, with stages allowed to overlap; any other call
    // drains what waits in that form and goes the two-launch way
    // (up to 128 frames: two 64 MiB work buffers in flight are what the Infinity Cache holds beside the rest — 256-frame calls lose a tenth
    // this way and have two rounds of workgroups per launch anyway, profiles/r04/s34_summary.txt)
    const bool merged_call = c->merge && (c->rows256_step || c->rows1024x256) && overlap && ring_only && !spec && nframes <= c->merge_max;
    if (c->merge && !merged_call && (c->have_rows || c->have_plan2 || c->have_det3)) flush_stages(c);
    if (dif_call) {
      if (!ring_only || !rx.hist_out) return fail(c, SS_ERR_INVALID, "internal: a fold call without a place for its rows");
      ss::Fft8192Args gf{};
      gf.tabs = ss::Fft8192V2Tables{c->d_tw8v2, c->d_tw8v2 + 256, c->d_tw8v2 + 256 + 384, nullptr, nullptr};
      gf.db_off = c->db_off;
      gf.scale = c->cfg.int_scale;
      gf.psd = rx.hist_out;  // row f - first_hist of this region is frame f's (RowsExtra, fft256_kernels.h)
      ss::Dif8Front df = ss::dif8_front_of(d_iq, item_stride, c->d_dif8_tab, 1 << c->dif_logq);
      df.smax = rx.smax;
      df.smax_mask = rx.smax_mask;
      df.abs0 = rx.abs0;
      df.first_hist = rx.first_hist;
      df.nframes = nframes;
      df.zero_word = rx.zero_word;
      FftRole drole;
      drole.frames = &gf;
      drole.dif = &df;
      // (two residues per workgroup: four per frame at radix 8, eight at radix 16; a radix-8 call of up to dif8_single_max frames — 0 in the
      // product — takes one residue per workgroup, eight per frame: KIND 11, a switch of the diagnostics build)
      const bool single = c->dif_logq == 3 && nframes <= c->diag.dif8_single_max;
      drole.n = (c->dif_logq == 4 || single ? 8 : SS_DIF8_W) * nframes;
      // the plan of the call before, the planned detect stage (of the call before that) and the emit stage behind it ride on the launch
      launch_step(c, &drole, c->have_det ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, c->have_emit ? &c->pend_emit : nullptr, nullptr, true);
    } else if (merged_call) {
      ss::ColsArgs gcm = gc;
      gcm.work = c->work_cur ? c->d_work2 : c->d_work;
      FftRole crole;
      crole.cols = &gcm;
      crole.n = nframes * (c->n >> 13);
      launch_step(c, &crole, c->have_det ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, c->have_emit ? &c->pend_emit : nullptr, nullptr, true,
                  (c->have_rows && !c->rows1024x256) ? &c->pend_rows : nullptr, c->have_rows ? c->pend_rows_tiles : 0, (c->have_rows && c->rows1024x256) ? &c->pend_rows1024 : nullptr);
    } else if (c->rows256_step) {
      // 65536 points with tile culling: both halves of the FFT as FFT roles of k_scan_step, the deferred stages shared out between
      // them — the column launch of call k carries the plan of call k - 1 (which of its tiles the detect stage must evaluate) and
      // emit(k - 2), the row launch right behind it detect(k - 1) on the tiles that plan listed. (All of them on the row launch:
      // 29.5 us for a launch that takes 17 alone; on the column launch the plan would have to be a launch of its own between the
      // two, 6.5 us: profiles/r04/s17_summary.txt.)
      // det_lag2 (what ships): the planned detect stage — detect(k - 2), whose plan rode on the column launch of call k - 1 — rides on
      // the COLUMN launch as well, and the row launch carries nothing: a workgroup that evaluates a pair of tiles lives ~20 us under
      // the streaming loads of its neighbours, longer than the row launch (17 us) and shorter than the column launch (24).
      const bool det_on_cols = c->det_lag2;
#ifdef SS_DIAG
      const bool emit_on_rows = c->diag.emit_on_rows;
#else
      const bool emit_on_rows = false;
#endif
      // A call of many frames goes through in chunks of 256 (a chunk's work buffer: 128 MiB, still in the Infinity Cache when its row
      // half reads it: the row half of a 512-frame call took 0.206 us per frame against 0.132, profiles/r04/s28_summary.txt); the
      // deferred stages ride on the first chunk's launches.
      const size_t sample = c->cfg.in_format == SS_FMT_CF32 ? 8 : 2;
      const int chunk = (c->diag.chunk_65536 > 0 && nframes > c->diag.chunk_65536) ? c->diag.chunk_65536 : nframes;
      for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = std::min(chunk, nframes - f0);
        const bool first = f0 == 0;
        c->prof_launch_frames = nf;
        ss::ColsArgs gcc = gc;
        gcc.iq = static_cast<const char*>(d_iq) + (size_t)f0 * (size_t)item_stride * sample;
        FftRole crole;
        crole.cols = &gcc;
        crole.n = nf * (c->n >> 13);
        launch_step(c, &crole, (first && det_on_cols && c->have_det) ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec,
                    (first && !emit_on_rows && c->have_emit) ? &c->pend_emit : nullptr, nullptr, first);
        ss::RowsExtra rxc = rx;
        rxc.abs0 += f0;
        rxc.first_hist -= f0;                 // (hist_out takes the call's frames >= first_hist; the tile sees chunk-local frame numbers)
        if (!first) rxc.zero_word = nullptr;  // (the first chunk's launch zeroes the count of the list this call's plan appends to)
        const ss::Rows256Args gr = rows256_args(c, (ring_only || !d_psd) ? nullptr : d_psd + (size_t)f0 * (size_t)c->n, rxc);
        FftRole rrole;
        rrole.rows256 = &gr;
        rrole.n = nf * 8;
        launch_step(c, &rrole, (first && !det_on_cols && c->have_det) ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec,
                    (first && emit_on_rows && c->have_emit) ? &c->pend_emit : nullptr);
      }
    } else if (c->rows1024x256) {
      // 262144 points (round 6): the 65536-point two-launch pipeline — the column launch of call k (256-point column tiles as k_scan_step's
      // FFT role) carries the plan of call k - 1, detect(k - 2) on the tiles that plan listed and emit(k - 3); the row half is a launch of
      // k_fft_rows1024_psd<8> that carries nothing. Calls of more than 64 frames go through in chunks (a chunk's work buffer: 128 MiB).
      const size_t sample = c->cfg.in_format == SS_FMT_CF32 ? 8 : 2;
      const int chunk = std::min(nframes, 64);
      for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = std::min(chunk, nframes - f0);
        const bool first = f0 == 0;
        c->prof_launch_frames = nf;
        ss::ColsArgs gcc = gc;
        gcc.iq = static_cast<const char*>(d_iq) + (size_t)f0 * (size_t)item_stride * sample;
        gcc.abs0 += f0;  // (the column tiles clear the ring words of THEIR frames' run maxima)
        FftRole crole;
        crole.cols = &gcc;
        crole.n = nf * (c->n >> 13);
        launch_step(c, &crole, (first && c->have_det) ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, (first && c->have_emit) ? &c->pend_emit : nullptr, nullptr, first);
        ss::RowsExtra rxc = rx;
        rxc.abs0 += f0;
        rxc.first_hist -= f0;
        if (!first) rxc.zero_word = nullptr;
        launch_rows1024x256(c, nf, (ring_only || !d_psd) ? nullptr : d_psd + (size_t)f0 * (size_t)c->n, rxc);
      }
    } else if (rows_by_step) {
      // 2^20 points: the column half of call k as a launch of its own (16 columns x 1024 rows per 1024-thread workgroup: too many
      // threads for a role), then ONE launch of k_scan_step whose FFT role is the ROW half of call k — 8 rows x 1024 points per
      // 512-thread workgroup — with detect(k - 1) and emit(k - 2) riding on it as they ride on the column launches of the other sizes
      // A call of many frames goes through in chunks of 16: the work buffer of a chunk (128 MiB) is still in the 256 MiB Infinity
      // Cache when the chunk's row half reads it — 2.8 us per frame for the row half against 4.1-4.3 when a 32- or 64-frame call's
      // work buffer has to come back from HBM (profiles/r04/s26_summary.txt). The deferred stages ride on the first chunk's row launch,
      // the plan of the call before on its column launch.
      const int chunk = (c->diag.chunk_long > 0 && nframes > c->diag.chunk_long) ? c->diag.chunk_long : nframes;
      for (int f0 = 0; f0 < nframes; f0 += chunk) {
        const int nf = std::min(chunk, nframes - f0);
        c->prof_launch_frames = nf;
        launch_cols1024(c, d_iq, item_stride, nf, f0);
        ss::RowsExtra rxc = rx;
        rxc.abs0 += f0;
        rxc.first_hist -= f0;                  // (hist_out takes the call's frames >= first_hist; the tile sees chunk-local frame numbers)
        if (f0 > 0) rxc.zero_word = nullptr;   // (the first chunk's launch zeroes the count of the list this call's plan appends to)
        float* psd_c = (ring_only || !d_psd) ? nullptr : d_psd + (size_t)f0 * (size_t)c->n;
        const ss::Rows1024Args gr = rows1024_args(c, psd_c, rxc);
        FftRole rrole;
        rrole.rows1024 = &gr;
        rrole.n = nf * 128;
        const bool first = f0 == 0;
        launch_step(c, &rrole, (first && c->have_det) ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, (first && c->have_emit) ? &c->pend_emit : nullptr);
      }
    } else
    launch_step(c, &role, c->have_det ? &c->pend_det : nullptr, c->pend_det_tiles, c->pend_det_spec, c->have_emit ? &c->pend_emit : nullptr);
    shift_pending(c);
    if (merged_call && c->rows1024x256) {  // (262144 points: 32 row tiles of 8 rows x 1024 points per frame)
      c->pend_rows1024 = rows1024x256_args(c, nullptr, rx);
      c->pend_rows1024.work = c->work_cur ? c->d_work2 : c->d_work;
      c->pend_rows_tiles = nframes * 32;
      c->have_rows = true;
      c->work_cur ^= 1;
    } else if (merged_call) {  // this call's row half waits for the next launch; it reads the work buffer this call's column half has just been told to fill
      c->pend_rows = rows256_args(c, nullptr, rx);
      c->pend_rows.work = c->work_cur ? c->d_work2 : c->d_work;
      c->pend_rows_tiles = nframes * 8;
      c->have_rows = true;
      c->work_cur ^= 1;
    }
    if (!c->use_fft8192 && !rows_by_step && !dif_call) {
      st = launch_fft_rows(c, nframes, ring_only ? nullptr : d_psd, rx);
      if (st != SS_OK) return st;
    }
    if (spec && !c->spec_in_detect) {
      st = spectrogram_accumulate(c, spec, d_psd, nframes);
      if (st != SS_OK) return st;
    }
    if (spec) spec->count += nframes;
    if (n_learn > 0) {
      hipLaunchKernelGGL(ss::k_noise_learn, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, (const float*)d_psd, c->n, n_learn, z->d_thr);
      if (c->cull || c->cull_long) hipLaunchKernelGGL(ss::k_thr_tilemin, dim3(c->n / 256), dim3(64), 0, c->stream, (const float*)z->d_thr, c->n, z->d_thr + c->n);
    }
    // (det_lag2: this call's detect stage waits for its plan, behind the planned one — which, if there is one, rode on this call's column launch)
    ss::DetectArgs& nd = merged_call ? c->pend_det3 : c->det_lag2 ? c->pend_det2 : c->pend_det;
    st = run_backend_fused(c, d_psd, nframes, n_learn, z, c->spec_in_detect ? spec : nullptr, d_rel_out, d_avg_out, d_cand_off, d_cand_idx, d_cand_avg,
                           cand_cap, true, &nd, merged_call ? &c->pend_det3_tiles : c->det_lag2 ? &c->pend_det2_tiles : &c->pend_det_tiles,
                           merged_call ? &c->pend_det3_emit : c->det_lag2 ? &c->pend_det2_emit : &c->pend_det_emit);
    if (st != SS_OK) return st;
    if (merged_call) {
      c->have_det3 = true;
      c->pend_det3_spec = false;
    } else if (c->det_lag2) {
      c->have_det2 = true;
      c->pend_det2_spec = c->spec_in_detect && spec != nullptr;
    } else {
      c->have_det = true;
      c->pend_det_spec = c->spec_in_detect && spec != nullptr;
    }
    if (cull_call) {
      nd.hist_by_fft = ring_by_rows ? 1 : 0;
      nd.tile_list = nullptr;
      if (ring_only) {  // the batch's rows are in the ring's buffer, as dB values — and so are the newest ring_db_rows rows of the window before them
        nd.psd = ring_rows;
        nd.thr = z->d_thr;
        nd.ring_db_from = -c->ring_db_rows;
        ring_only_rows = ring_rows;
        c->last_db_from = nd.ring_db_from;
        c->ring_db_rows = std::min(kHistRows, c->ring_db_rows + nframes);
        c->ring_db_thr = z->d_thr;
      }
      // the plan: which tiles of this call can hold a candidate at all (k_plan_long) — behind the rows kernel, ahead of the
      // launch that carries the detect stage. Only a stage whose sole products are mask bits and counts is planned.
      // (the plan of call k rides at the front of call k + 1's column launch: four plan blocks share a column tile's LDS)
      const bool plan_fused = (rows_by_step || dif_call) && c->diag.plan_fused && overlap;
      const int plan_cols = ss::plan_long_cols(nframes, nd.shift, c->n / 256, 8, plan_fused ? ss::kPlanFusedFloats : ss::kPlanLongFloats);  // (32 columns per workgroup — whole lines of the two-pass layout — halved the fetches and doubled the time: 128 workgroups are too few, profiles/r04/s6_summary.txt)
      if (ring_by_rows && !spec && !nd.rel_out && !nd.avg_out && plan_cols > 0) {
        int* list = c->d_tlist[(c->buf_cur + c->nbuf - 1) % c->nbuf];  // (run_backend_fused has moved buf_cur on: the set this call's mask bits go to)
        ss::PlanLongArgs pl{};
        pl.smax = c->d_smax;
        pl.smax_mask = c->smax_rows - 1;
        pl.abs0 = rx.abs0;
        pl.clean_rel = (int)std::max<long long>(c->clean_abs - c->abs_frames, -(1ll << 29));
        pl.cols = plan_cols;
        pl.logn = c->logn;
        pl.list = list;
        pl.layout = c->two_pass ? 1 : dif_call ? 2 : c->rows1024x256 ? 3 : 0;
        if (merged_call) {  // (not ready before this call's row half has run: one launch from now)
          c->have_plan2 = true;
          c->pend_plan2 = pl;
          c->pend_plan2_det = ss::plan_long_det(nd);
        } else if (plan_fused) {
          c->have_plan = true;
          c->pend_plan = pl;
          c->pend_plan_det = ss::plan_long_det(nd);
        } else {
          const int plan_wgs = ss::plan_blocks_of(ss::plan_long_det(nd), pl, c->n);  // (groups past the band's end find no column)
          SS_LAUNCH_SLOT(c, SS_KSLOT_PLAN, (ss::k_plan_long<21, 21, kFusedTF, 256>), dim3(plan_wgs), dim3(256), 0, ss::plan_long_det(nd), pl);
        }
        nd.tile_list = list;
      }
    }
    if (!overlap) flush_stages(c);
  } else {
    ++c->stats.calls_in_order;
    st = launch_fft(c, d_iq, item_stride, nframes, d_psd);
    if (st != SS_OK) return st;
    if (spec) {
      if (!c->spec_in_detect) {
        st = spectrogram_accumulate(c, spec, d_psd, nframes);
        if (st != SS_OK) return st;
      }
      spec->count += nframes;
    }
    if (c->fused) {
      if (n_learn > 0) hipLaunchKernelGGL(ss::k_noise_learn, dim3((c->n + 255) / 256), dim3(256), 0, c->stream, (const float*)d_psd, c->n, n_learn, z->d_thr);
      st = run_backend_fused(c, d_psd, nframes, n_learn, z, c->spec_in_detect ? spec : nullptr, d_rel_out, d_avg_out, d_cand_off, d_cand_idx, d_cand_avg,
                             cand_cap, false, nullptr, nullptr, nullptr);
    } else {
      st = run_backend_unfused(c, d_psd, nframes, n_learn, z, d_rel_out, d_avg_out, d_cand_off, d_cand_idx, d_cand_avg, cand_cap);
    }
    if (st != SS_OK) return st;
  }
  ++c->batch_no;
  ++c->stats.calls;
  SS_HIP(c, hipGetLastError());
  c->frames_pushed = c->frames_pushed + nframes < G ? c->frames_pushed + nframes : G;
  if (n_learn > 0) c->clean_abs = c->abs_frames + n_learn;  // (tile culling, long transforms: rows before this one hold learning frames)
  c->abs_frames += nframes;
  c->last_psd = d_psd;
  c->last_rel_rows = nullptr;
  if (ring_only_rows) {  // a ring-only call (above): no dB plane exists
    c->last_psd = nullptr;
    c->last_rel_rows = ring_only_rows;
  }
  c->last_rows_perm8 = c->ring_perm8;  // (the order last_hist and last_rel_rows are in: ss_read_window)
  c->last_rows_db = ring_only_rows != nullptr;  // (... and whether they hold dB values, from frame last_db_from on)
  c->last_settled_lo = c->last_settled_hi = nullptr;
  c->last_n = nframes;
  return SS_OK;
}

// Noise::add's bookkeeping for a batch (noise_learner.cpp:11-28): how many leading frames still learn.
int plan_learning(ss_ctx* c, NoiseState* z, int nframes, const int64_t* t_ms) {
  int n_learn = 0;
  for (int f = 0; f < nframes && !z->ready; ++f) {
    if (!z->have_start) {  // Noise::Noise() reads getTime() when m_noise[frequency] is first touched (:9, :42)
      z->start_ms = t_ms ? t_ms[f] : 0;
      z->have_start = true;
    }
    ++n_learn;
    ++z->samples;
    const bool done = t_ms ? (z->start_ms + c->cfg.learn_ms <= t_ms[f]) : (z->samples >= c->cfg.learn_frames);
    if (done) z->ready = true;
  }
  return n_learn;
}

int get_noise(ss_ctx* c, NoiseState** out) {
  const int32_t center = (c->range_lo + c->range_hi) / 2;
  NoiseState* z = noise_for(c, center);
  if (!z) {
    NoiseState nz;
    nz.center = center;
    const size_t extra = (size_t)std::max(32, c->n / 256);  // + the per-tile-column minima (tile culling)
    SS_HIP(c, hipMalloc(&nz.d_thr, sizeof(float) * ((size_t)c->n + extra)));
    hipLaunchKernelGGL(ss::k_fill, dim3(grid_for((size_t)c->n + extra, 256)), dim3(256), 0, c->stream, nz.d_thr, (size_t)c->n + extra, -FLT_MAX);
    c->noise.push_back(nz);
    z = &c->noise.back();
  }
  *out = z;
  return SS_OK;
}

void free_ctx(ss_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device_id);
  for (hipStream_t q : c->s_ab)
    if (q) (void)hipStreamSynchronize(q);
  if (c->s_fold) (void)hipStreamSynchronize(c->s_fold);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
#ifdef SS_DIAG
  if (c->diag.d_cull_stats > reinterpret_cast<unsigned*>(1)) {
    unsigned h[3] = {0, 0, 0};
    if (hipMemcpy(h, c->diag.d_cull_stats, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[specscan diag] detect tiles %u, on the culling path %u, culled %u\n", h[0], h[1], h[2]);
    (void)hipFree(c->diag.d_cull_stats);
  }
#endif
  for (hipStream_t q : c->s_ab)
    if (q) (void)hipStreamDestroy(q);
  if (c->s_fold) (void)hipStreamDestroy(c->s_fold);
  for (hipEvent_t e : c->ev_fold_src)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_fold_done)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_launch)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_in)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : c->ev_join)
    if (e) (void)hipEventDestroy(e);
  if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
  (void)hipFree(c->d_drain_done);
  for (auto& z : c->noise) (void)hipFree(z.d_thr);
  for (auto& g : c->spec) (void)hipFree(g.d_sum);
  (void)hipFree(c->d_spec_partial);
  for (auto e : c->prof_events) (void)hipEventDestroy(e);
  (void)hipFree(c->d_win);
  (void)hipFree(c->d_tw);
  (void)hipFree(c->d_tw8k);
  (void)hipFree(c->d_pass);
  (void)hipFree(c->d_rel);
  (void)hipFree(c->d_tw8v2);
  (void)hipFree(c->d_dif8_tab);
  (void)hipFree(c->d_perm_tmp);
  (void)hipFree(c->diag.d_stamps);
  for (auto& t : c->order_tables) (void)hipFree(t.d);
  (void)hipFree(c->d_hist);
  for (auto p : c->d_cnt3) (void)hipFree(p);
  for (auto p : c->d_mask2) (void)hipFree(p);
  for (auto p : c->d_avg2) (void)hipFree(p);
  for (auto p : c->d_off4) (void)hipFree(p);
  for (auto p : c->d_psd2) (void)hipFree(p);
  for (auto p : c->d_segsum) (void)hipFree(p);
  for (auto p : c->d_live) (void)hipFree(p);
  for (auto p : c->d_tlist) (void)hipFree(p);
  (void)hipFree(c->diag.d_canary);
  if (c->diag.ev_canary) (void)hipEventDestroy(c->diag.ev_canary);
  (void)hipFree(c->d_smax);
  for (auto p : c->d_halo) (void)hipFree(p);
  (void)hipFree(c->d_relplane);
  (void)hipFree(c->d_hist_tmp);
  (void)hipFree(c->d_avgy);
  (void)hipFree(c->d_work);
  (void)hipFree(c->d_tw256);
  (void)hipFree(c->d_tw_cols);
  (void)hipFree(c->d_zero_row);
  (void)hipFree(c->d_win1024);
  (void)hipFree(c->d_wtab1024);
  (void)hipFree(c->d_work2);
  (void)hipFree(c->d_tw_sub);
  (void)hipFree(c->d_tw_small);
  (void)hipFree(c->d_tw_rowsR);
  (void)hipFree(c->d_spec_ring);
  (void)hipFree(c->d_spec_part2[0]);
  (void)hipFree(c->d_spec_part2[1]);
  (void)hipFree(c->d_counts);
  (void)hipFree(c->d_stats);
  (void)hipFree(c->d_nf);
  (void)hipFree(c->d_bad_from);
  (void)hipFree(c->d_nan_state);
  for (hipEvent_t e : c->ev_call)
    if (e) (void)hipEventDestroy(e);
  (void)hipFree(c->d_in);
  (void)hipFree(c->d_cand_idx);
  (void)hipFree(c->d_cand_avg);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // namespace

// exhaustive check of div_const<21> (detect_fused.h) against the hardware IEEE division
__global__ void k_selftest_div21(unsigned long long* mismatches) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long bad = 0;
  for (unsigned long long u = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; u < (1ull << 32); u += stride) {
    const float s = __uint_as_float((unsigned)u);
    if (s != s || (fabsf(s) > 1e30f && fabsf(s) < __builtin_inff())) continue;  // NaN, and the overflow corner that dB sums cannot reach; the infinities count (a row of zeros)
    const float q = ss::div_const<21>(s);
    const float r = s / 21.0f;
    if (__float_as_uint(q) != __float_as_uint(r) && !(q == 0.0f && r == 0.0f)) ++bad;
  }
  if (bad) atomicAdd(mismatches, bad);
}

extern "C" {

// Device self-tests of arithmetic shortcuts. which = 0: div_const<21>(s) == s / 21.0f for every float
// |s| <= 1e30. Returns the number of mismatching inputs, or a negative ss_status.
long long ss_selftest(int device_id, int which) {
  if (which != 0) return SS_ERR_INVALID;
  if (hipSetDevice(device_id) != hipSuccess) return SS_ERR_NO_DEVICE;
  unsigned long long* d = nullptr;
  unsigned long long h = 0;
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return SS_ERR_HIP;
  (void)hipMemset(d, 0, sizeof(h));
  hipLaunchKernelGGL(k_selftest_div21, dim3(256 * 8), dim3(256), 0, 0, d);
  const hipError_t e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  return e == hipSuccess ? (long long)h : (long long)SS_ERR_HIP;
}

void ss_default_config(ss_config* cfg, int32_t sample_rate, int32_t center_hz) {
  if (!cfg) return;
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = SS_ABI_VERSION;
  cfg->fft_size = get_fft(sample_rate, 250);  // SIGNAL_DETECTION_MAX_STEP, config.h:33; sdr_device.cpp:149
  cfg->sample_rate = sample_rate;
  const double step = (double)sample_rate / cfg->fft_size;  // sdr_device.cpp:150
  const int d = (int)(step / 50);                           // SIGNAL_DETECTION_FPS, config.h:32; sdr_device.cpp:152
  cfg->decim = d > 1 ? d : 1;
  cfg->in_format = SS_FMT_CF32;
  cfg->grouping_x = 21;     // config.h:28
  cfg->grouping_y = 21;     // config.h:29
  cfg->start_level = 8.0f;  // config.h:30
  cfg->range_lo = center_hz - sample_rate / 2;
  cfg->range_hi = center_hz + sample_rate / 2;
  cfg->learn_frames = 100;  // NOISE_LEARNING_TIME (config.h:24) at 50 frames/s
  cfg->learn_ms = 2000;
  cfg->max_batch = 1024;
  cfg->device_id = 0;
}

int ss_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int ss_create(const ss_config* cfg, ss_ctx** out) {
  if (!cfg || !out) return fail(nullptr, SS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->abi_version != SS_ABI_VERSION) return fail(nullptr, SS_ERR_INVALID, "abi_version %d != %d", cfg->abi_version, SS_ABI_VERSION);
  if (!is_pow2(cfg->fft_size) || cfg->fft_size < 64 || cfg->fft_size > (1 << 20))
    return fail(nullptr, SS_ERR_INVALID, "fft_size %d must be a power of two in [64, 2^20]", cfg->fft_size);
  if (cfg->sample_rate <= 0 || cfg->decim < 1 || cfg->max_batch < 1 || cfg->n_ignored < 0 || cfg->learn_frames < 1 ||
      cfg->grouping_y < 1 || cfg->grouping_x < 1 || (cfg->grouping_x & 1) == 0 || cfg->grouping_x > 1025 ||
      cfg->in_format < SS_FMT_CF32 || cfg->in_format > SS_FMT_CU8 || (cfg->n_ignored > 0 && !cfg->ignored))
    return fail(nullptr, SS_ERR_INVALID, "invalid ss_config field");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(nullptr, SS_ERR_NO_DEVICE, "no HIP device available");
  if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(nullptr, SS_ERR_NO_DEVICE, "device_id %d out of range (%d devices)", cfg->device_id, ndev);

  ss_ctx* c = new (std::nothrow) ss_ctx();
  if (!c) return fail(nullptr, SS_ERR_NOMEM, "out of host memory");
  c->cfg = *cfg;
  c->cfg.window = nullptr;
  c->cfg.ignored = nullptr;
  c->n = cfg->fft_size;
  c->db_off = (float)(10.0 * log10((double)cfg->sample_rate));
  while ((1 << c->logn) < c->n) ++c->logn;
  c->range_lo = cfg->range_lo;
  c->range_hi = cfg->range_hi;
  c->ignored.assign(cfg->ignored, cfg->ignored + 2 * (size_t)cfg->n_ignored);
  if (c->cfg.int_scale == 0.0f) c->cfg.int_scale = cfg->in_format == SS_FMT_CU8 ? 1.0f / 127.5f : 1.0f / 128.0f;

  const int n = c->n;
  const int G = cfg->grouping_y;
  const size_t plane = sizeof(float) * (size_t)n * (size_t)cfg->max_batch;
#define CREATE_HIP(call)                                                                      \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      fail(nullptr, SS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));               \
      free_ctx(c);                                                                            \
      return SS_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)
  CREATE_HIP(hipSetDevice(cfg->device_id));
  CREATE_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  CREATE_HIP(hipMalloc(&c->d_win, sizeof(float) * (size_t)n));
  CREATE_HIP(hipMalloc(&c->d_tw, sizeof(float2) * (size_t)n));
  CREATE_HIP(hipMalloc(&c->d_pass, (size_t)n));
  c->diag.read();
  c->abs_frames = c->clean_abs = c->diag.abs_start;
  c->fused = G == 21 && cfg->grouping_x == 21 && cfg->max_batch <= 65536 && !c->diag.backend_unfused;
  // 8192 points (and the four-step sizes) with the fused back end: stages of consecutive calls overlap (scan_step.h), so
  // what a deferred stage reads rotates over several buffers; every other configuration uses set 0 only
  c->step_path = c->fused && !c->diag.fft_generic && (n == 8192 || (n >= 16384 && c->diag.step_long));
  if (cfg->flags & SS_FLAG_SPECTROGRAM) {
    // output size rule of the Spectrogram block: min(SPECTROGRAM_MAX_FFT, getFft(fs, SPECTROGRAM_PREFERRED_MAX_STEP)),
    // sources/radio/blocks/spectrogram.cpp:14, config.h:36-37
    int out_n = get_fft(cfg->sample_rate, 1000);
    if (out_n > 16384) out_n = 16384;
    if (out_n > n) out_n = n;
    c->spec_n = out_n;
    c->spec_m = n / out_n;
    // inside the detect kernel when that kernel runs and a tile's 256 bins hold whole groups of m: one partial row per
    // frame tile (a batch that starts inside a tile touches one more)
    c->spec_in_detect = c->fused && c->spec_m <= 256 && !c->diag.spec_standalone;
  }
  c->ref_nan = (cfg->flags & SS_FLAG_REFERENCE_NAN) != 0;
  if (c->ref_nan && !c->fused) {
    fail(nullptr, SS_ERR_INVALID, "SS_FLAG_REFERENCE_NAN needs the 21 x 21 grouping (and max_batch <= 65536)");
    free_ctx(c);
    return SS_ERR_INVALID;
  }
  c->deep = c->step_path && n == 8192 && c->diag.deep && !(cfg->flags & (SS_FLAG_STREAM_ORDERED | SS_FLAG_REFERENCE_NAN)) && cfg->max_batch >= kHistRows && (!(cfg->flags & SS_FLAG_SPECTROGRAM) || c->spec_in_detect);
  if (c->diag.wait_limit > 0) c->wait_limit = c->diag.wait_limit;
  CREATE_HIP(hipMalloc(&c->d_stats, sizeof(unsigned long long) * ss::kStatShards * ss::kStatShardStride));
  CREATE_HIP(hipMemsetAsync(c->d_stats, 0, sizeof(unsigned long long) * ss::kStatShards * ss::kStatShardStride, c->stream));
  if (c->ref_nan) {
    CREATE_HIP(hipMalloc(&c->d_nf, sizeof(int) * 4 * (size_t)cfg->max_batch));
    CREATE_HIP(hipMalloc(&c->d_bad_from, sizeof(int) * (size_t)cfg->max_batch));
    CREATE_HIP(hipMalloc(&c->d_nan_state, sizeof(ss::NanState)));
    hipLaunchKernelGGL(ss::k_nan_state_reset, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<ss::NanState*>(c->d_nan_state), n);
  }
  c->nq = c->deep ? std::min(std::max(c->diag.queues, 2), kMaxQueues) : 1;
  // 65536 points with tile culling: a call's detect stage rides two calls later (ss_ctx::det_lag2) — decided here, where the rotating
  // buffers are sized: two more sets than launches in order need
  // (131072 points — what getFft picks at 20 MS/s — have the pipeline of 65536 points where the fold can run at all: int8 IQ, default window)
  const bool fold_ok = c->diag.dif8 && c->diag.emit_wide && c->diag.ring_only && (cfg->in_format == SS_FMT_CS8 || cfg->in_format == SS_FMT_CU8) && !cfg->window &&
                       !(cfg->flags & (SS_FLAG_STREAM_ORDERED | SS_FLAG_REFERENCE_NAN | SS_FLAG_KEEP_PLANES | SS_FLAG_SPECTROGRAM));
  const bool x256_ok = n == 262144 && c->diag.rows1024x256 && c->diag.emit_wide && c->diag.fft_rows_r < 0 && c->diag.fft_sub < 0 && !(cfg->flags & SS_FLAG_SPECTROGRAM);
  c->det_lag2 = !c->deep && c->step_path && (n == 65536 || (n == 131072 && fold_ok) || x256_ok) && !c->diag.fft_generic && c->diag.cull_65536 && c->diag.cull && !(cfg->flags & SS_FLAG_NO_CULL) &&
                c->diag.rows256_step && c->diag.step_long && c->diag.det_lag2 && c->fused;
  c->merge = c->det_lag2 && (n == 65536 || x256_ok) && c->diag.merge_65536 && c->diag.emit_wide;  // (one launch per call: scan_step.h KIND 7 / KIND 12, whose emit role is the wide one)
  c->merge_max = std::max(1, (int)((long long)c->diag.merge_max_frames * 65536 / n));  // (128 frames of 65536 points, 32 of 262144: 64 MiB of work buffer either way)
  // ... and with int8 IQ and the default window no work buffer at all: the radix-8 fold (scan_step.h KIND 8). It takes every call the
  // one-launch form above would take — and longer ones — so that form is off then.
  c->dif8 = c->det_lag2 && fold_ok && (n == 65536 || n == 131072);
  c->dif_logq = n == 131072 ? 4 : 3;
  c->cull_fold_only = c->dif8 && n == 131072;
  if (c->dif8) c->merge = false;
  c->lag = c->deep ? c->nq : (c->det_lag2 ? 2 : 1);
  c->ncnt = c->deep ? 3 * c->nq : (c->merge ? 8 : c->det_lag2 ? 6 : 3);
  c->nbuf = c->deep ? 2 * c->nq : (c->merge ? 6 : c->det_lag2 ? 4 : (c->step_path ? 2 : 1));
  c->npsd = c->deep ? 2 * c->nq : (c->step_path ? 2 : 1);  // (deep: written by launch L, read by launch L + nq, written again by launch L + 2 nq on the same queue)
  if (c->fused) {
    {
      // ring capacity: at least three windows (a long batch needs a free one next to the one it reads), more when rows
      // are small, so that short batches slide for a long time before the window is moved back to the front
      // (the window slides by a batch's frames; when it reaches the end, kHistRows rows are copied back to the front:
      // 147 MB for 2^20-point rows, so long rows get at least ten windows — one copy per ~20 sixteen-frame batches)
      long long rows = (64ll << 20) / ((long long)n * 4);
      const long long min_rows = (long long)n >= 65536 ? 10 * kHistRows : 3 * kHistRows;
      if (rows < min_rows) rows = min_rows;
      if (rows > 64 * kHistRows) rows = 64 * kHistRows;
      // (round 6: long transforms on the step path get 1024 rows — 1 GiB at 262144 points, 4 GiB at 2^20: a call shorter than the window
      // slides it along the buffer, and every return to the front is a drain of the stages that still read the old place; with 350 rows
      // 32-frame calls of 262144 points drained every eighth call, 74 us each: 9.6 of their 63 us per call, profiles/r06/s7_summary.txt;
      // 16-frame calls of 2^20 points every thirteenth. HBM is what this chip has most of.)
      if (c->step_path && n >= 65536 && rows < 1024) rows = 1024;
      // (long transforms, whose rows kernel may write ALL of a batch's rows into this buffer: three batches and the averager's reach,
      // so that a batch can go back to the front of the buffer while the one before it is still to be read — ring_place.h)
      if (n >= 65536 && c->step_path && rows < 3ll * cfg->max_batch + 3 * kHistRows) rows = 3ll * cfg->max_batch + 3 * kHistRows;
      if ((c->merge || c->dif8) && rows < 4ll * (cfg->max_batch + kHistRows)) rows = 4ll * (cfg->max_batch + kHistRows);  // (two spans to protect: ring_place.h)
      c->hist_rows = (int)rows;
      CREATE_HIP(hipMalloc(&c->d_hist, sizeof(float) * (size_t)n * (size_t)rows));
      CREATE_HIP(hipMemsetAsync(c->d_hist, 0, sizeof(float) * (size_t)n * (size_t)kHistRows, c->stream));  // Averager ctor, averager.cpp:7-12
    }
    for (int k = 0; k < c->ncnt; ++k) {
      CREATE_HIP(hipMalloc(&c->d_cnt3[k], sizeof(int) * (size_t)cfg->max_batch));
      CREATE_HIP(hipMemsetAsync(c->d_cnt3[k], 0, sizeof(int) * (size_t)cfg->max_batch, c->stream));
    }
  } else {
    CREATE_HIP(hipMalloc(&c->d_rel, sizeof(float) * (size_t)n * (size_t)(G - 1 + cfg->max_batch)));
    CREATE_HIP(hipMalloc(&c->d_hist_tmp, sizeof(float) * (size_t)n * (size_t)(G > 1 ? G - 1 : 1)));
    CREATE_HIP(hipMalloc(&c->d_avgy, plane));
    CREATE_HIP(hipMemsetAsync(c->d_rel, 0, sizeof(float) * (size_t)n * (size_t)(G - 1 + cfg->max_batch), c->stream));
  }
  for (int k = 0; k < c->npsd; ++k) CREATE_HIP(hipMalloc(&c->d_psd2[k], plane));
  for (int k = 0; k < c->nbuf; ++k) {
    CREATE_HIP(hipMalloc(&c->d_avg2[k], plane));
    CREATE_HIP(hipMalloc(&c->d_mask2[k], sizeof(uint32_t) * (size_t)(n / 32) * (size_t)cfg->max_batch));
    CREATE_HIP(hipMemsetAsync(c->d_mask2[k], 0, sizeof(uint32_t) * (size_t)(n / 32) * (size_t)cfg->max_batch, c->stream));  // (tile culling relies on it: EmitArgs::clear_masks)
    CREATE_HIP(hipMalloc(&c->d_off4[k], sizeof(int) * ((size_t)cfg->max_batch + 1)));
  }
  c->d_off = c->d_off4[0];
  c->cull = c->fused && n == 8192 && !c->diag.fft_generic && !(cfg->flags & SS_FLAG_NO_CULL) && c->diag.cull;
  if (c->cull)
    for (int k = 0; k < c->nbuf; ++k) {
      CREATE_HIP(hipMalloc(&c->d_segsum[k], sizeof(float) * 32 * (size_t)cfg->max_batch));
      static_assert(ss::kLiveHeader + ss::kLiveLists * ss::kLiveCap <= ss::kLiveCounts, "the lists end before the count copies");
      const size_t live_ints = ss::kLiveInts;
      CREATE_HIP(hipMalloc(&c->d_live[k], sizeof(int) * live_ints));
      CREATE_HIP(hipMemsetAsync(c->d_live[k], 0, sizeof(int) * live_ints, c->stream));
    }
  if (c->deep) {
#ifdef SS_DIAG
    if (c->diag.canary) {
      CREATE_HIP(hipMalloc(&c->diag.d_canary, sizeof(unsigned long long) * 128));
      CREATE_HIP(hipMemset(c->diag.d_canary, 0, sizeof(unsigned long long) * 128));
      CREATE_HIP(hipEventCreateWithFlags(&c->diag.ev_canary, hipEventDisableTiming));
    }
#endif
    c->deep_ring_safe = c->hist_rows / kHistRows >= kDeepSyncPhase + 8;
    for (int k = 0; k < c->nq; ++k) CREATE_HIP(hipStreamCreateWithFlags(&c->s_ab[k], hipStreamNonBlocking));
    for (int k = 0; k < 2 * c->nq; ++k)  // (+ the halo frames' per-column maxima behind their rows: halo_segsum_of)
      CREATE_HIP(hipMalloc(&c->d_halo[k], sizeof(float) * ((size_t)n * (size_t)kHistRows + 32 * (size_t)ss::kHaloSegPitch)));
    for (auto& e : c->ev_launch) CREATE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_in) CREATE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev_join) CREATE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CREATE_HIP(hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming));
    CREATE_HIP(hipMalloc(&c->d_drain_done, 64));
    CREATE_HIP(hipMemset(c->d_drain_done, 0, 64));
  }
  if (c->step_path) {
    hipDeviceProp_t prop;
    CREATE_HIP(hipGetDeviceProperties(&prop, cfg->device_id));
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t max_items = (size_t)(cfg->max_batch + kHistRows) * (size_t)(n / 8192) + ((size_t)cfg->max_batch / kFusedTF + 2) * (size_t)(n / 256) / 2 + (size_t)cfg->max_batch + 4 +
                             (n >= 65536 ? (size_t)cfg->max_batch * (size_t)(n / 8192) + 512 : 0) +  // (one launch per call: the row tiles of the call before, the plan workgroups)
                             (size_t)ss::kLiveLists * (ss::kLiveCap / 2 + 1);  // (a planned detect stage without an FFT role: 256 consumers per list, + the plan workgroups)
    c->order_capacity = max_items;
    c->order_tables.resize(16);
    for (auto& t : c->order_tables) {
      t.used = 0;
      t.d = nullptr;
      CREATE_HIP(hipMalloc(&t.d, sizeof(uint32_t) * c->order_capacity));
    }
  }
  CREATE_HIP(hipMalloc(&c->d_counts, sizeof(int) * (size_t)cfg->max_batch));
  if (cfg->flags & SS_FLAG_SPECTROGRAM) {
    const int out_n = c->spec_n;
    if (c->spec_in_detect) {
      const size_t tiles = (size_t)(cfg->max_batch + kFusedTF - 1) / kFusedTF + 1;
      for (int k = 0; k < 2; ++k) CREATE_HIP(hipMalloc(&c->d_spec_part2[k], sizeof(float) * (size_t)out_n * tiles));
      if (c->deep) {  // one slot of partial sums per call in flight or not yet added to its container (ss_ctx::deep_folds)
        c->spec_slot_floats = (size_t)out_n * tiles;
        CREATE_HIP(hipMalloc(&c->d_spec_ring, sizeof(float) * c->spec_slot_floats * (size_t)kDeepSpecSlots));
        CREATE_HIP(hipStreamCreateWithFlags(&c->s_fold, hipStreamNonBlocking));
        for (auto& e : c->ev_fold_src) CREATE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& e : c->ev_fold_done) CREATE_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto& b : c->slot_batch) b = -1;
      }
    } else {
      CREATE_HIP(hipMalloc(&c->d_spec_partial, sizeof(float) * (size_t)out_n * ((size_t)(cfg->max_batch + 31) / 32)));
    }
  }
  if (c->logn > 13) CREATE_HIP(hipMalloc(&c->d_work, sizeof(float2) * (size_t)n * (size_t)cfg->max_batch));
  // (the one-launch form only takes calls of up to merge_max_frames: the second work buffer need not hold more)
  if (c->merge) CREATE_HIP(hipMalloc(&c->d_work2, sizeof(float2) * (size_t)n * (size_t)std::min(cfg->max_batch, c->merge_max)));

  // window: caller's taps or gr::fft::window::hamming(N) (sdr_device.cpp:164; GNU Radio's definition:
  // 0.54 - 0.46*cos(2*pi*n/(N-1)) in double, stored as float). Twiddles W_N^k from double.
  {
    std::vector<float> win((size_t)n);
    if (cfg->window) {
      memcpy(win.data(), cfg->window, sizeof(float) * (size_t)n);
    } else {
      const float M = (float)(n - 1);
      for (int i = 0; i < n; ++i) win[(size_t)i] = (float)(0.54 - 0.46 * cos((2.0 * M_PI * i) / M));
    }
    std::vector<float2> tw((size_t)n);
    for (int k = 0; k < n; ++k) {
      const double ang = -2.0 * M_PI * (double)k / (double)n;
      tw[(size_t)k] = make_float2((float)cos(ang), (float)sin(ang));
    }
    if ((n == 1024 || n == 2048 || n == 4096) && !c->diag.fft_generic) {
      const int R = n / 256;
      std::vector<float2> t256(256), tn((size_t)R * 256);
      for (int r = 0; r < 16; ++r)
        for (int m = 0; m < 16; ++m) {
          const double ang = -2.0 * M_PI * (double)(m * r) / 256.0;
          t256[(size_t)(r * 16 + m)] = make_float2((float)cos(ang), (float)sin(ang));
        }
      for (int q = 0; q < R; ++q)
        for (int k = 0; k < 256; ++k) {
          const double ang = -2.0 * M_PI * (double)(q * k) / (double)n;
          tn[(size_t)(q * 256 + k)] = make_float2((float)cos(ang), (float)sin(ang));
        }
      CREATE_HIP(hipMalloc(&c->d_tw256, sizeof(float2) * t256.size()));
      CREATE_HIP(hipMemcpy(c->d_tw256, t256.data(), sizeof(float2) * t256.size(), hipMemcpyHostToDevice));
      CREATE_HIP(hipMalloc(&c->d_tw_small, sizeof(float2) * tn.size()));
      CREATE_HIP(hipMemcpy(c->d_tw_small, tn.data(), sizeof(float2) * tn.size(), hipMemcpyHostToDevice));
    }
    if (n >= 16384) {
      c->use_fft256 = !c->diag.fft_generic;
      std::vector<float2> t256(256);
      for (int r = 0; r < 16; ++r)
        for (int m = 0; m < 16; ++m) {
          const double ang = -2.0 * M_PI * (double)(m * r) / 256.0;
          t256[(size_t)(r * 16 + m)] = make_float2((float)cos(ang), (float)sin(ang));
        }
      CREATE_HIP(hipMalloc(&c->d_tw256, sizeof(float2) * t256.size()));
      CREATE_HIP(hipMemcpy(c->d_tw256, t256.data(), sizeof(float2) * t256.size(), hipMemcpyHostToDevice));
      c->two_pass = c->use_fft256 && c->logn == 20 && c->diag.fft_twopass && c->diag.fft_rows_r < 0 && c->diag.fft_sub < 0;
      if (c->two_pass) {  // one block of tables for both halves (fft1024_kernels.h), in the place of the 256-point column tiles' tables
        std::vector<float2> tab((size_t)ss::kFft1024TableEntries);
        ss::fft1024_host_tables(tab.data());
        CREATE_HIP(hipMalloc(&c->d_tw_cols, sizeof(float2) * tab.size()));
        CREATE_HIP(hipMemcpy(c->d_tw_cols, tab.data(), sizeof(float2) * tab.size(), hipMemcpyHostToDevice));
      } else {
        const int n2size = n / 256;
        std::vector<float2> tc((size_t)32 * (size_t)n2size);
        for (int i = 0; i < 16; ++i)
          for (int n2 = 0; n2 < n2size; ++n2) {
            const double a1 = -2.0 * M_PI * ((double)n2 * i) / (double)n, a2 = -2.0 * M_PI * (16.0 * (double)n2 * i) / (double)n;
            tc[(size_t)i * n2size + n2] = make_float2((float)cos(a1), (float)sin(a1));
            tc[(size_t)(16 + i) * n2size + n2] = make_float2((float)cos(a2), (float)sin(a2));
          }
        CREATE_HIP(hipMalloc(&c->d_tw_cols, sizeof(float2) * tc.size()));
        CREATE_HIP(hipMemcpy(c->d_tw_cols, tc.data(), sizeof(float2) * tc.size(), hipMemcpyHostToDevice));
        if (n2size >= 512 && n2size <= 4096 && (c->diag.fft_rows_r >= 0 ? c->diag.fft_rows_r == 1 : n2size <= 2048)) {  // 4096-point rows: a tie with the two-kernel form
          const int R = n2size / 256;
          std::vector<float2> tr((size_t)R * 256);
          for (int q = 0; q < R; ++q)
            for (int k = 0; k < 256; ++k) {
              const double ang = -2.0 * M_PI * ((double)q * k) / (double)n2size;
              tr[(size_t)q * 256 + k] = make_float2((float)cos(ang), (float)sin(ang));
            }
          CREATE_HIP(hipMalloc(&c->d_tw_rowsR, sizeof(float2) * tr.size()));
          CREATE_HIP(hipMemcpy(c->d_tw_rowsR, tr.data(), sizeof(float2) * tr.size(), hipMemcpyHostToDevice));
        }
        if (n2size > 256 && !c->d_tw_rowsR && (c->diag.fft_sub >= 0 ? c->diag.fft_sub == 1 : n2size >= 2048)) {
          const int A = n2size / 256;
          std::vector<float2> ts((size_t)A * 256);
          for (int cc = 0; cc < A; ++cc)
            for (int b = 0; b < 256; ++b) {
              const double ang = -2.0 * M_PI * ((double)b * cc) / (double)n2size;
              ts[(size_t)cc * 256 + b] = make_float2((float)cos(ang), (float)sin(ang));
            }
          CREATE_HIP(hipMalloc(&c->d_tw_sub, sizeof(float2) * ts.size()));
          CREATE_HIP(hipMemcpy(c->d_tw_sub, ts.data(), sizeof(float2) * ts.size(), hipMemcpyHostToDevice));
        }
      }
    }
    if (n == 8192) {
      c->use_fft8192 = !c->diag.fft_generic;
      std::vector<float2> t8((size_t)(256 + 1024 + 2048));
      auto W = [](double num, double den) {
        const double ang = -2.0 * M_PI * num / den;
        return make_float2((float)cos(ang), (float)sin(ang));
      };
      for (int r = 0; r < 16; ++r)
        for (int m = 0; m < 16; ++m) t8[(size_t)(r * 16 + m)] = W((double)m * r, 256.0);
      for (int r1 = 0; r1 < 4; ++r1)
        for (int t = 0; t < 256; ++t) t8[(size_t)(256 + r1 * 256 + t)] = W((double)t * r1, 8192.0);
      for (int r2 = 0; r2 < 8; ++r2)
        for (int t = 0; t < 256; ++t) t8[(size_t)(256 + 1024 + r2 * 256 + t)] = W((double)t * r2, 2048.0);
      CREATE_HIP(hipMalloc(&c->d_tw8k, sizeof(float2) * t8.size()));
      CREATE_HIP(hipMemcpy(c->d_tw8k, t8.data(), sizeof(float2) * t8.size(), hipMemcpyHostToDevice));
      std::vector<float2> v2((size_t)(256 + 384 + 96));
      ss::fft8192_v2_host_tables(v2.data(), v2.data() + 256, v2.data() + 256 + 384);
      CREATE_HIP(hipMalloc(&c->d_tw8v2, sizeof(float2) * v2.size()));
      CREATE_HIP(hipMemcpy(c->d_tw8v2, v2.data(), sizeof(float2) * v2.size(), hipMemcpyHostToDevice));
    }
    CREATE_HIP(hipMemcpy(c->d_win, win.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
    if (c->two_pass) {  // the same taps in the 1024-point column tiles' own order (fft1024_kernels.h)
      std::vector<float> wk((size_t)n);
      ss::fft1024_window_order(win.data(), wk.data(), c->diag.cols1024_wide ? 4 : 3);
      CREATE_HIP(hipMalloc(&c->d_win1024, sizeof(float) * (size_t)n));
      CREATE_HIP(hipMemcpy(c->d_win1024, wk.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice));
      if (!cfg->window && c->diag.cols1024_wide && c->diag.win_calc) {  // the default window: Hamming taps formed in the kernel
        std::vector<float2> wt(65536);
        ss::fft1024_window_rotation_table(wt.data());
        CREATE_HIP(hipMalloc(&c->d_wtab1024, sizeof(float2) * wt.size()));
        CREATE_HIP(hipMemcpy(c->d_wtab1024, wt.data(), sizeof(float2) * wt.size(), hipMemcpyHostToDevice));
      }
    }
    CREATE_HIP(hipMemcpy(c->d_tw, tw.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    if (n == 65536 && c->use_fft256 && !cfg->window && c->diag.win_calc) {  // the default window: the 256-point column tiles form their Hamming taps (fft256_kernels.h)
      std::vector<float2> wt(4096);
      ss::fft65536_window_rotation_table(wt.data());
      CREATE_HIP(hipMalloc(&c->d_wtab1024, sizeof(float2) * wt.size()));
      CREATE_HIP(hipMemcpy(c->d_wtab1024, wt.data(), sizeof(float2) * wt.size(), hipMemcpyHostToDevice));
    }
    if (c->dif8) {  // the radix-8 fold: its own tables and the 8192-point transform's
      const double scale = (double)c->cfg.int_scale;  // (what load_iq multiplies with in the other front ends)
      std::vector<float2> dt((size_t)ss::dif_table_float2(1 << c->dif_logq));
      ss::dif8_host_tables(dt.data(), scale, 1 << c->dif_logq);
      CREATE_HIP(hipMalloc(&c->d_dif8_tab, sizeof(float2) * dt.size()));
      CREATE_HIP(hipMemcpy(c->d_dif8_tab, dt.data(), sizeof(float2) * dt.size(), hipMemcpyHostToDevice));
      std::vector<float2> v2((size_t)(256 + 384 + 96));
      ss::fft8192_v2_host_tables(v2.data(), v2.data() + 256, v2.data() + 256 + 384);
      CREATE_HIP(hipMalloc(&c->d_tw8v2, sizeof(float2) * v2.size()));
      CREATE_HIP(hipMemcpy(c->d_tw8v2, v2.data(), sizeof(float2) * v2.size(), hipMemcpyHostToDevice));
      CREATE_HIP(hipMalloc(&c->d_perm_tmp, sizeof(float) * (size_t)n * (size_t)kHistRows));
    }
  }
  // Tile culling for long transforms: the sizes whose rows go through k_fft_rows256_psd (N2 = 256, or the radix-A step in front)
  // and the two-pass form of 2^20 points. On by default where it pays: 2^20 points. At 65536 points it takes 7 B/sample off the
  // fabric and no time off the call: the column launch gets 10 us shorter, the plan launch costs 6.5 and the ring rows 1.4
  // (57.4 against 56.9 us per 128-frame call now that detect-mode calls write no dB plane, run_batch: ring_only; 57.1 against
  // 53.4 before; 38 against 29 us per 16-frame call either way: profiles/r04/s11_summary.txt, profiles/r03/s53_summary.txt).
  // There only the diagnostics build switches it on (SS_CULL_65536=1; tests/test_gpu_cull.py keeps it honest).
  c->x256_tile = n == 262144 && c->use_fft256 && c->d_tw_rowsR != nullptr && c->diag.rows1024x256 && c->diag.fft_rows_r < 0 && c->diag.fft_sub < 0;
  c->rows1024x256 = c->det_lag2 && c->x256_tile;
  c->cull_long = c->step_path && c->use_fft256 && ((n == 65536 && c->diag.cull_65536) || (n > 65536 && c->d_tw_sub && !c->d_tw_rowsR) || c->two_pass || c->cull_fold_only || c->rows1024x256) &&
                 !(cfg->flags & SS_FLAG_NO_CULL) && c->diag.cull;
  if (!c->cull_long) c->rows1024x256 = false;
  if (n == 262144 && !c->rows1024x256) c->merge = false;  // (never: decided from the same switches; the second work buffer and the rotating sets sized for it do no harm)
  // 65536 points: with the plan of call k at the front of call k + 1's column launch — which therefore is a launch of its own, the
  // row tiles taking the FFT role of k_scan_step in its place (KIND 6) — the culling pays there too (session 17 of round 4).
  c->rows256_step = c->cull_long && !c->two_pass && c->logn == 16 && c->diag.rows256_step && c->diag.step_long;  // (no emit stage ever rides on the row launch — KIND 6, whose emit role is the wide one — but under SS_EMIT_ON_ROWS)
  if (!c->rows256_step && !c->cull_fold_only && !c->rows1024x256) c->det_lag2 = false;  // (never: the two are decided from the same switches; the rotating buffers sized for it do no harm)
  if (c->dif8 && !c->cull_long) c->dif8 = c->cull_fold_only = false;
  if (c->cull_long) {
    CREATE_HIP(hipMalloc(&c->d_zero_row, sizeof(float) * (size_t)n));  // (what a ring-only call's detect stage subtracts from rows that are noise-relative already)
    CREATE_HIP(hipMemset(c->d_zero_row, 0, sizeof(float) * (size_t)n));
    int rows = 64;
    // (two batches and the averager's reach: the plan of call k — which reads the maxima of its frames and of the 35 before —
    // runs beside the column tiles of call k + 1, which clear the rows of THEIR frames: k_fft_cols1024_plan)
    // (262144 points in one launch per call: the column tiles of call k clear their frames' words while the plan of call k - 2 rides on the
    // same launch: three batches)
    while (rows < ((c->merge && c->rows1024x256) ? 3 : 2) * cfg->max_batch + kHistRows + 1) rows <<= 1;
    c->smax_rows = rows;
    CREATE_HIP(hipMalloc(&c->d_smax, sizeof(float) * (size_t)rows * (size_t)(n / 32)));
    if (c->rows1024x256) CREATE_HIP(hipMemsetAsync(c->d_smax, 0, sizeof(float) * (size_t)rows * (size_t)(n / 32), c->stream));  // (keys: 0 = nothing seen)
    const size_t max_tiles = ((size_t)cfg->max_batch / kFusedTF + 2) * (size_t)(n / 256);
    for (int k = 0; k < std::max(2, c->nbuf); ++k) CREATE_HIP(hipMalloc(&c->d_tlist[k], sizeof(int) * (max_tiles + 2)));  // (count, entries, one slot behind an odd count)
  }
  // 64 KiB of dynamic LDS needs no opt-in on gfx950 (160 KiB/CU), but say so explicitly for clarity
  CREATE_HIP(hipStreamSynchronize(c->stream));
#undef CREATE_HIP
  *out = c;
  return SS_OK;
}

void ss_destroy(ss_ctx* ctx) { free_ctx(ctx); }

const char* ss_last_error(const ss_ctx* ctx) { return ctx ? ctx->err : g_create_err; }

void* ss_stream(ss_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

// (ss_ctx::deep_eager) called by ss_flush / ss_sync: did the caller wait after a single call?
static void note_caller_sync(ss_ctx* c) {
  if (c->deep_calls_since_sync == 1) c->deep_eager = true;
  else if (c->deep_calls_since_sync > 1) c->deep_eager = false;
  c->deep_calls_since_sync = 0;
}

int ss_flush(ss_ctx* ctx) {
  if (!ctx) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(ctx->mtx);
  SS_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
  note_caller_sync(ctx);
  flush_stages(ctx);
  SS_HIP(ctx, hipGetLastError());
  return SS_OK;
}

int ss_sync(ss_ctx* ctx) {
  if (!ctx) return SS_ERR_INVALID;
  hipStream_t stream = nullptr;
  {
    std::lock_guard<std::mutex> lock(ctx->mtx);
    SS_HIP(ctx, hipSetDevice(ctx->cfg.device_id));
    note_caller_sync(ctx);
    flush_stages(ctx);
    SS_HIP(ctx, hipGetLastError());
    stream = ctx->stream;
  }
  // (the wait itself without the context's lock: a producer thread of ss_feed_*, or a control call, is not held up by it)
  const hipError_t e = stream_wait(stream);
  if (e != hipSuccess) {
    std::lock_guard<std::mutex> lock(ctx->mtx);
    return fail(ctx, SS_ERR_HIP, "stream_wait failed: %s", hipGetErrorString(e));
  }
#ifdef SS_DIAG
  {
    std::lock_guard<std::mutex> lock(ctx->mtx);
    if (ctx->diag.d_canary && !ctx->diag.canary_pairs.empty()) {
      unsigned long long h[128];
      SS_HIP(ctx, hipMemcpy(h, ctx->diag.d_canary, sizeof(h), hipMemcpyDeviceToHost));
      int bad = -1;
      for (int slot : ctx->diag.canary_pairs)
        if (h[2 * slot] != h[2 * slot + 1]) bad = slot;
      ctx->diag.canary_pairs.clear();
      if (bad >= 0)
        return fail(ctx, SS_ERR_INVALID, "input canary: the last frames of a call's d_iq changed before the next call's launch had read them again "
                                         "(every buffer of ss_process_device must stay untouched until ss_sync; SS_FLAG_STREAM_ORDERED lifts that)");
    }
  }
#endif
  return SS_OK;
}

// ss_get_stats: the host-side counters as they stand, the device-side ones as far as the device has got (a blocking 8 KiB
// copy that synchronises with none of the context's streams: they are all non-blocking streams).
int ss_get_stats(ss_ctx* c, ss_stats* out) {
  if (!c || !out || out->size < sizeof(uint32_t) * 2) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  static_assert(ss::kStatWords <= ss::kStatShardStride, "a shard holds every counter");
  unsigned long long dev[ss::kStatShards * ss::kStatShardStride] = {};
  SS_HIP(c, hipMemcpy(dev, c->d_stats, sizeof(dev), hipMemcpyDeviceToHost));
  ss_stats st = c->stats;
  st.tiles_tested = st.tiles_culled = st.wait_fallbacks = 0;
  for (int k = 0; k < ss::kStatShards; ++k) {  // (the device adds to the copy of the workgroup's block index, detect_fused.h)
    st.tiles_tested += dev[k * ss::kStatShardStride + ss::kStatTested];
    st.tiles_culled += dev[k * ss::kStatShardStride + ss::kStatCulled];
    st.wait_fallbacks += dev[k * ss::kStatShardStride + ss::kStatWaitFallbacks];
  }
  st.state = (c->cull || c->cull_long ? SS_STATE_CULLING : 0u) | (c->deep ? SS_STATE_OVERLAP : 0u) | (c->deep && c->deep_iq_recycled ? SS_STATE_DEMOTED : 0u) |
             (c->deep && c->deep_eager ? SS_STATE_EAGER : 0u);
  const uint32_t want = out->size < sizeof(ss_stats) ? out->size : (uint32_t)sizeof(ss_stats);
  st.size = want;
  memcpy(out, &st, want);
  return SS_OK;
}

// ss_input_wait (include/specscan.h): `stream` waits until the input frames of every call up to the one `calls_back` before the
// latest have been read for the last time — by their own launch and, their last frames, by the launch of the call after it.
int ss_input_wait(ss_ctx* c, void* stream_v, int32_t calls_back) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (calls_back < 1) return fail(c, SS_ERR_INVALID, "ss_input_wait: calls_back %d < 1 (the latest call's last frames are read again by the call after it)", calls_back);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  hipStream_t stream = stream_v ? static_cast<hipStream_t>(stream_v) : c->stream;
  const auto behind_public = [&]() -> int {  // whatever the public stream holds now
    if (stream == c->stream) return SS_OK;
    hipEvent_t& ev = c->ev_call[c->call_seq & 31];
    if (!ev) SS_HIP(c, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    SS_HIP(c, hipEventRecord(ev, c->stream));
    SS_HIP(c, hipStreamWaitEvent(stream, ev, 0));
    ++c->call_seq;
    return SS_OK;
  };
  if (!c->deep || c->deep_L == 0) return behind_public();  // every stage so far is on (or joined into) the public stream
  const long target = c->deep_L - 1 - calls_back;  // the launch of the newest call whose input is to be dead
  const bool first_use = !c->input_events;
  c->input_events = true;
  if (target < 0) {
    // (calls from before this run of overlapped launches: in order on the public stream, and the run's first launch takes its
    // rows from before the batch from the ring, not from their frames)
    return behind_public();
  }
  const long hi = target + 1, lo = std::max(0l, hi - (c->nq - 1));  // one launch per queue, the newest at or below target + 1
  const bool recorded = !first_use && lo >= c->input_events_from && (32 % c->nq) == 0;
  if (recorded) {
    for (long L = hi; L >= lo; --L) SS_HIP(c, hipStreamWaitEvent(stream, c->ev_launch[L & 31], 0));  // (a slot recorded again since then belongs to a later launch of the same queue: waits longer, never shorter)
  } else {
    // no per-launch events yet (first use, or launches from before it): behind everything the queues hold now
    for (int q = 0; q < c->nq; ++q) {
      SS_HIP(c, hipEventRecord(c->ev_join[q], c->s_ab[q]));
      SS_HIP(c, hipStreamWaitEvent(stream, c->ev_join[q], 0));
    }
    if (first_use) c->input_events_from = c->deep_L;
  }
  return SS_OK;
}

int ss_kernel_timing(ss_ctx* c, int enable) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  c->prof_on = enable != 0;
  c->prof_every = enable > 1 ? enable : 1;
  c->prof_seen = (unsigned)c->prof_every / 2;  // (every k-th launch from the k/2-th on: not the first launches of a run, which fill the pipeline)
  c->prof_used = 0;
  return SS_OK;
}

int ss_kernel_timing_read_slots(ss_ctx* c, double* ms_by_slot, int32_t* launches_by_slot) { return ss_kernel_timing_read_frames(c, ms_by_slot, launches_by_slot, nullptr); }

int ss_kernel_timing_read_frames(ss_ctx* c, double* ms_by_slot, int32_t* launches_by_slot, int64_t* frames_by_slot) {
  if (!c || !ms_by_slot || !launches_by_slot) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  flush_stages(c);  // (timed launches may sit on the side streams of the deep pipelining)
  SS_HIP(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < SS_KSLOT_COUNT; ++k) {
    ms_by_slot[k] = 0.0;
    launches_by_slot[k] = 0;
    if (frames_by_slot) frames_by_slot[k] = 0;
  }
  for (size_t i = 0; i + 1 < c->prof_used; i += 2) {
    float ms = 0.f;
    SS_HIP(c, hipEventElapsedTime(&ms, c->prof_events[i], c->prof_events[i + 1]));
    const int slot = c->prof_slots[i / 2];
    ms_by_slot[slot] += ms;
    ++launches_by_slot[slot];
    if (frames_by_slot) frames_by_slot[slot] += c->prof_frames[i / 2];
  }
  c->prof_used = 0;
  c->prof_call = false;
  return SS_OK;
}

int ss_kernel_timing_read(ss_ctx* c, double* total_ms, int32_t* launches) {
  if (!total_ms || !launches) return SS_ERR_INVALID;
  double ms[SS_KSLOT_COUNT];
  int32_t cnt[SS_KSLOT_COUNT];
  const int st = ss_kernel_timing_read_slots(c, ms, cnt);
  if (st != SS_OK) return st;
  *total_ms = ms[SS_KSLOT_STEP];
  *launches = cnt[SS_KSLOT_STEP];
  return SS_OK;
}

int ss_process_device(ss_ctx* c, const void* d_iq, int32_t nframes, float* d_psd_db, float* d_rel_db, float* d_avg_db, int32_t* d_cand_off,
                      int32_t* d_cand_idx, float* d_cand_avg, int32_t cand_cap) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (nframes < 0 || (nframes > 0 && !d_iq) || cand_cap < 0) return fail(c, SS_ERR_INVALID, "bad d_iq/nframes/cand_cap");
  if (nframes > c->cfg.max_batch) return fail(c, SS_ERR_BATCH, "nframes %d > max_batch %d", nframes, c->cfg.max_batch);
  if (nframes == 0) {
    if (d_cand_off) SS_HIP(c, hipMemsetAsync(d_cand_off, 0, sizeof(int32_t), c->stream));
    c->last_n = 0;
    return SS_OK;
  }
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  NoiseState* z = nullptr;
  int st = get_noise(c, &z);
  if (st != SS_OK) return st;
  const int n_learn = plan_learning(c, z, nframes, nullptr);
  return run_batch(c, d_iq, (long long)c->n * c->cfg.decim, nframes, n_learn, z, d_psd_db, d_rel_db, d_avg_db, d_cand_off, d_cand_idx,
                   d_cand_avg, cand_cap, true);
}

int ss_process(ss_ctx* c, const void* iq, int32_t nframes, const int64_t* t_ms, float* psd_db, float* rel_db, float* avg_db,
               int32_t* cand_off, int32_t* cand_idx, float* cand_avg, int32_t cand_cap) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (nframes < 0 || (nframes > 0 && !iq) || cand_cap < 0) return fail(c, SS_ERR_INVALID, "bad iq/nframes/cand_cap");
  if (nframes > c->cfg.max_batch) return fail(c, SS_ERR_BATCH, "nframes %d > max_batch %d", nframes, c->cfg.max_batch);
  if (cand_off) cand_off[0] = 0;
  if (nframes == 0) {
    c->last_n = 0;
    return SS_OK;
  }
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
#ifdef SS_DIAG
  static const bool trace = getenv("SS_TRACE_PROCESS") != nullptr;  // (where a call's time goes, phase by phase: stderr)
  auto t_last = std::chrono::steady_clock::now();
  const auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[ss_process] %-28s %9.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
#else
  const auto lap = [](const char*) {};
#endif
  const int n = c->n;
  const size_t bps = in_bytes_per_sample(c->cfg.in_format);
  const size_t row_bytes = (size_t)n * bps;
  if (!c->d_in) SS_HIP(c, hipMalloc(&c->d_in, row_bytes * (size_t)c->cfg.max_batch));
  lap("set device, input buffer");
  // Decimator: only the first N samples of each N*D item ever reach the GPU (decimator.h:15-22)
  // The caller's buffer is pageable (the scheduler's) and the call hands its results back before it returns: a synchronous copy — D = 1:
  // the items are one run of bytes. (128 MiB reach the device in 2.4 ms, 56 GB/s, by hipMemcpy, hipMemcpyAsync on any stream or the 2-D form
  // alike — profiles/r06/s7_pageable_copy_lab.txt; what made this copy take 19-29 ms in round 5's and this round's bench lines was the
  // CALLER unmapping 128 MiB of host memory between calls — the Python wrapper's default candidate arrays —, profiles/r06/s10_summary.txt.)
  // Everything this context enqueued earlier is complete (the previous call drained it), so nothing else has to be ordered with the copy.
  if (c->cfg.decim == 1) SS_HIP(c, hipMemcpy(c->d_in, iq, row_bytes * (size_t)nframes, hipMemcpyHostToDevice));
  else SS_HIP(c, hipMemcpy2D(c->d_in, row_bytes, iq, row_bytes * (size_t)c->cfg.decim, row_bytes, (size_t)nframes, hipMemcpyHostToDevice));
  lap("copy in");
  if (cand_cap > c->cand_cap_alloc) {
    (void)hipFree(c->d_cand_idx);
    (void)hipFree(c->d_cand_avg);
    c->d_cand_idx = nullptr;
    c->d_cand_avg = nullptr;
    c->cand_cap_alloc = 0;
    SS_HIP(c, hipMalloc(&c->d_cand_idx, sizeof(int) * (size_t)cand_cap));
    SS_HIP(c, hipMalloc(&c->d_cand_avg, sizeof(float) * (size_t)cand_cap));
    c->cand_cap_alloc = cand_cap;
  }
  NoiseState* z = nullptr;
  int st = get_noise(c, &z);
  if (st != SS_OK) return st;
  const int n_learn = plan_learning(c, z, nframes, t_ms);
  const bool want_cands = cand_idx && cand_cap > 0;
  float* d_rel_plane = nullptr;
  if (c->fused && rel_db) {
    if (!c->d_relplane) SS_HIP(c, hipMalloc(&c->d_relplane, sizeof(float) * (size_t)n * (size_t)c->cfg.max_batch));
    d_rel_plane = c->d_relplane;
  }
  lap("noise state, buffers");
  st = run_batch(c, c->d_in, (long long)n, nframes, n_learn, z, nullptr, d_rel_plane, (c->fused && avg_db) ? c->d_avg2[c->buf_cur] : nullptr, nullptr,
                 want_cands ? c->d_cand_idx : nullptr, want_cands && cand_avg ? c->d_cand_avg : nullptr, cand_cap);
  if (st != SS_OK) return st;
  lap("run_batch (enqueue)");
  flush_stages(c);  // a work() call hands its results back before it returns
  lap("flush_stages (enqueue)");
  const size_t plane = sizeof(float) * (size_t)n * (size_t)nframes;
  if (psd_db) SS_HIP(c, hipMemcpyAsync(psd_db, c->last_psd, plane, hipMemcpyDeviceToHost, c->stream));
  if (rel_db) {
    const float* src = c->fused ? c->d_relplane : c->d_rel + (size_t)(c->cfg.grouping_y - 1) * n;
    SS_HIP(c, hipMemcpyAsync(rel_db, src, plane, hipMemcpyDeviceToHost, c->stream));
  }
  if (avg_db) SS_HIP(c, hipMemcpyAsync(avg_db, c->last_avg, plane, hipMemcpyDeviceToHost, c->stream));
  std::vector<int> off((size_t)nframes + 1);
  SS_HIP(c, hipMemcpyAsync(off.data(), c->d_off, sizeof(int) * ((size_t)nframes + 1), hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, stream_wait(c->stream));
  lap("wait for the stream");
  if (cand_off) memcpy(cand_off, off.data(), sizeof(int) * ((size_t)nframes + 1));
  const int total = off[(size_t)nframes];
  if (want_cands) {
    const int ncopy = total < cand_cap ? total : cand_cap;
    if (ncopy > 0) {
      SS_HIP(c, hipMemcpy(cand_idx, c->d_cand_idx, sizeof(int) * (size_t)ncopy, hipMemcpyDeviceToHost));
      if (cand_avg) SS_HIP(c, hipMemcpy(cand_avg, c->d_cand_avg, sizeof(float) * (size_t)ncopy, hipMemcpyDeviceToHost));
    }
  }
  lap("candidates out");
  if (cand_cap > 0 && total > cand_cap) return fail(c, SS_ERR_CAND_OVERFLOW, "%d candidates > cand_cap %d", total, cand_cap);
  return SS_OK;
}

int ss_set_frequency_range(ss_ctx* c, int32_t lo_hz, int32_t hi_hz) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  flush_stages(c);  // deferred stages belong to the old range (pass mask, noise ceiling)
  settle_ring_db(c);  // (ring rows left as dB values belong to the old centre frequency's ceiling)
  c->range_lo = lo_hz;
  c->range_hi = hi_hz;
  c->pass_dirty = true;
  c->clean_abs = c->abs_frames;  // another centre frequency, another noise ceiling: the ring rows so far were formed with the old one
  return SS_OK;
}

int ss_reset(ss_ctx* c) {  // Transmission::resetBuffers -> Averager::reset: rows and sums to zero, m_frames = 0
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  flush_stages(c);  // deferred stages still read and write the ring
  c->ring_db_rows = 0;
  c->ring_db_thr = nullptr;
  const int G = c->cfg.grouping_y;
  if (c->fused) {
    c->hist_start = 0;
    c->hist_prev = c->hist_prev2 = ss::RingPrev{0, -1, 0};
    SS_HIP(c, hipMemsetAsync(c->d_hist, 0, sizeof(float) * (size_t)c->n * (size_t)kHistRows, c->stream));
  } else if (G > 1) {
    SS_HIP(c, hipMemsetAsync(c->d_rel, 0, sizeof(float) * (size_t)c->n * (size_t)(G - 1), c->stream));
  }
  c->frames_pushed = 0;
  c->abs_frames = c->diag.abs_start;
  c->clean_abs = c->diag.abs_start;
  c->rot_frames = 0;
  c->last_n = 0;
  c->deep_iq_recycled = false;  // (a retune is a fresh start for the caller's buffers too; ss_get_stats keeps the count)
  if (c->ref_nan) hipLaunchKernelGGL(ss::k_nan_state_reset, dim3(1), dim3(64), 0, c->stream, reinterpret_cast<ss::NanState*>(c->d_nan_state), c->n);  // Averager::reset zeroes the sums: the poison goes with them
  return SS_OK;
}

int ss_reset_noise(ss_ctx* c) {
  if (!c) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  flush_stages(c);
  settle_ring_db(c);  // (ring rows left as dB values: the ceiling they belong to goes away)
  SS_HIP(c, hipStreamSynchronize(c->stream));
  for (auto& z : c->noise) (void)hipFree(z.d_thr);
  c->noise.clear();
  c->clean_abs = c->abs_frames;
  c->last_thr = nullptr;  // ss_read_window(SS_PLANE_REL) of the last batch has nothing to subtract any more
  c->last_n = 0;
  return SS_OK;
}

// bins [lo, lo + cnt) of a row in the fold's order (fft65536_dif8.h): the row comes over whole, the host picks
static int read_perm8_row(ss_ctx* c, const float* row, int lo, size_t cnt, float* out) {
  std::vector<float> h((size_t)c->n);
  SS_HIP(c, hipMemcpyAsync(h.data(), row, sizeof(float) * h.size(), hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < cnt; ++k) out[k] = h[(size_t)ss::dif_bin_offset(lo + (int)k, c->dif_logq)];
  return SS_OK;
}

// out[k] -= ceiling[lo + k]: a window of a row that holds dB values, as the noise-relative values the reference's chain has (noise_learner.cpp:55)
static int sub_thr_window(ss_ctx* c, int lo, size_t cnt, float* out) {
  if (!c->last_thr) return fail(c, SS_ERR_INVALID, "no noise ceiling to subtract");
  std::vector<float> thr(cnt);
  SS_HIP(c, hipMemcpyAsync(thr.data(), c->last_thr + lo, sizeof(float) * cnt, hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < cnt; ++k) out[k] = out[k] - thr[k];
  return SS_OK;
}

int ss_read_window(ss_ctx* c, int32_t plane, int32_t frame, int32_t lo, int32_t hi, float* out) {
  if (!c || !out) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  const int n = c->n;
  const int G = c->cfg.grouping_y;
  if (lo < 0 || hi > n || lo > hi || frame >= c->last_n || c->last_n <= 0 || (!c->last_psd && !c->last_rel_rows)) return fail(c, SS_ERR_INVALID, "window out of range (or no batch processed yet)");
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  flush_stages(c);
  const size_t cnt = (size_t)(hi - lo);
  if (!c->last_psd) {
    // the last batch kept no dB plane (a 2^20-point device call that handed out no plane: its rows went straight to the averager
    // ring as noise-relative values): the rel rows are there, bit for bit; a dB window needs a call with d_psd_db
    if (plane == SS_PLANE_REL && frame >= 0) {
      if (c->last_rows_perm8) {
        const int st = read_perm8_row(c, c->last_rel_rows + (size_t)frame * n, lo, cnt, out);
        if (st != SS_OK) return st;
      } else {
        SS_HIP(c, hipMemcpyAsync(out, c->last_rel_rows + (size_t)frame * n + lo, sizeof(float) * cnt, hipMemcpyDeviceToHost, c->stream));
        SS_HIP(c, hipStreamSynchronize(c->stream));
      }
      const float* row = c->last_rel_rows + (size_t)frame * n;
      const bool settled = c->last_settled_lo && row >= c->last_settled_lo && row < c->last_settled_hi;  // (a retune since: the window's rows are noise-relative already)
      return c->last_rows_db && !settled ? sub_thr_window(c, lo, cnt, out) : SS_OK;  // (the rows are dB values: rel = dB - ceiling, noise_learner.cpp:55)
    }
    if (frame >= 0) return fail(c, SS_ERR_INVALID, "the last ss_process_device call kept no dB / avg plane (detect mode): pass d_psd_db, or SS_FLAG_KEEP_PLANES at ss_create");
  }
  if (c->fused && plane == SS_PLANE_REL && frame >= 0) {
    // the fused back end never stores rel: rebuild the window as the kernel computes it,
    // psd - thr in fp32 (noise_learner.cpp:55), or -100 for a learning frame (:49)
    if (frame < c->last_n_learn) {
      for (size_t k = 0; k < cnt; ++k) out[k] = SS_NO_DATA;
      return SS_OK;
    }
    std::vector<float> thr(cnt);
    SS_HIP(c, hipMemcpyAsync(out, c->last_psd + (size_t)frame * n + lo, sizeof(float) * cnt, hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipMemcpyAsync(thr.data(), c->last_thr + lo, sizeof(float) * cnt, hipMemcpyDeviceToHost, c->stream));
    SS_HIP(c, hipStreamSynchronize(c->stream));
    for (size_t k = 0; k < cnt; ++k) out[k] = out[k] - thr[k];
    return SS_OK;
  }
  const float* src = nullptr;
  if (frame >= 0) {
    if (plane == SS_PLANE_PSD) src = c->last_psd + (size_t)frame * n;
    if (plane == SS_PLANE_AVG) {
      if (c->fused && !(c->cfg.flags & SS_FLAG_KEEP_PLANES)) return fail(c, SS_ERR_INVALID, "SS_PLANE_AVG needs SS_FLAG_KEEP_PLANES at ss_create");
      src = c->last_avg + (size_t)frame * n;
    }
    if (plane == SS_PLANE_REL && !c->fused) src = c->d_rel + (size_t)(G - 1 + frame) * n;
  } else if (plane == SS_PLANE_REL && frame >= -(G - 1)) {
    // ring rows as they were before the batch: the ring's newest row is frame -1
    src = c->fused ? c->last_hist + (size_t)(kHistRows + frame) * n : c->d_rel + (size_t)(G - 1 + frame) * n;
  }
  if (!src) return fail(c, SS_ERR_INVALID, "bad plane/frame");
  const bool ring_row_db = frame < 0 && c->fused && plane == SS_PLANE_REL && c->last_rows_db && frame >= c->last_db_from &&
                          !(c->last_settled_lo && src >= c->last_settled_lo && src < c->last_settled_hi);  // (a ring row an earlier detect-mode call left as dB values, not settled since)
  if (frame < 0 && c->fused && c->last_rows_perm8) {  // (ring rows the fold left)
    const int st = read_perm8_row(c, src, lo, cnt, out);
    if (st != SS_OK) return st;
    return ring_row_db ? sub_thr_window(c, lo, cnt, out) : SS_OK;
  }
  SS_HIP(c, hipMemcpyAsync(out, src + lo, sizeof(float) * cnt, hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipStreamSynchronize(c->stream));
  return ring_row_db ? sub_thr_window(c, lo, cnt, out) : SS_OK;
}

int ss_spectrogram_size(const ss_ctx* c) { return c ? c->spec_n : 0; }

// Spectrogram::send without its 1000 ms gate (spectrogram.cpp:62-75): out[j] = int8(sum[j] / count) — the C++
// float -> int8 conversion, i.e. truncation toward zero — for the current centre frequency; the container is
// cleared. mean_out (nullable) receives the float before conversion. Returns the number of frames that were
// accumulated (0: nothing to send, out untouched), or a negative status.
int ss_spectrogram_read(ss_ctx* c, int8_t* out, float* mean_out) {
  if (!c || !out) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (c->spec_n == 0) return fail(c, SS_ERR_INVALID, "spectrogram not enabled (SS_FLAG_SPECTROGRAM)");
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  const int32_t center = (c->range_lo + c->range_hi) / 2;
  SpecState* g = nullptr;
  for (auto& s : c->spec)
    if (s.center == center) g = &s;
  if (!g || g->count == 0) return 0;
  flush_stages(c);
  spectrogram_flush(c);
  std::vector<float> sum((size_t)c->spec_n);
  SS_HIP(c, hipMemcpyAsync(sum.data(), g->d_sum, sizeof(float) * sum.size(), hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipMemsetAsync(g->d_sum, 0, sizeof(float) * sum.size(), c->stream));
  SS_HIP(c, hipStreamSynchronize(c->stream));
  const int count = g->count;
  for (int j = 0; j < c->spec_n; ++j) {
    const float v = sum[(size_t)j] / count;  // container.m_sum[j] / container.m_counter (float / int)
    if (mean_out) mean_out[j] = v;
    out[j] = (int8_t)v;
  }
  g->count = 0;
  return count;
}

int ss_spectrogram_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* row, int32_t size, uint8_t* out,
                           int32_t cap) {
  if (size < 1 || (out && !row)) return SS_ERR_INVALID;  // step = sampleRate / size: the reference never sends an empty row
  const long long total = (long long)(sizeof(uint64_t) + 3 * sizeof(int32_t) + sizeof(uint32_t)) + size;
  if (!out) return (int)total;
  if (total > cap) return SS_ERR_INVALID;
  const int32_t head[3] = {frequency - sample_rate / 2, frequency + sample_rate / 2, sample_rate / size};  // data_controller.cpp:45-47
  const uint32_t count = (uint32_t)size;
  uint8_t* p = out;
  memcpy(p, &time_ms, sizeof(time_ms));
  p += sizeof(time_ms);
  memcpy(p, head, sizeof(head));
  p += sizeof(head);
  memcpy(p, &count, sizeof(count));
  p += sizeof(count);
  memcpy(p, row, (size_t)size);
  return (int)total;
}

int ss_read_noise(ss_ctx* c, float* thr) {
  if (!c || !thr) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  flush_stages(c);  // (like every call that reads state; nothing deferred writes the ceiling today)
  NoiseState* z = noise_for(c, (c->range_lo + c->range_hi) / 2);
  if (!z) {
    for (int i = 0; i < c->n; ++i) thr[i] = -FLT_MAX;
    return 0;
  }
  SS_HIP(c, hipMemcpyAsync(thr, z->d_thr, sizeof(float) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipStreamSynchronize(c->stream));
  return z->ready ? 1 : 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Pipelined host feeding (include/specscan.h): pinned staging slots, H2D on a copy stream, the chain on
// the context's stream, exact-size candidate read-back on a third stream at collect time.
// ------------------------------------------------------------------------------------------------
struct ss_feed_slot {
  void* h_in = nullptr;  // pinned
  void* d_in = nullptr;
  int32_t* h_off = nullptr;  // pinned
  int32_t* h_idx = nullptr;
  float* h_avg = nullptr;
  float* h_psd = nullptr;
  int32_t* d_off = nullptr;
  int32_t* d_idx = nullptr;
  float* d_avg = nullptr;
  hipEvent_t ev_h2d = nullptr, ev_done = nullptr;
  int nframes = 0;
  int64_t tag = 0;
  int state = 0;  // 0 free, 1 acquired (being filled), 2 submitted
};

struct ss_feed {
  ss_ctx* c = nullptr;
  int depth = 0, cand_cap = 0;
  bool want_psd = false;
  std::vector<ss_feed_slot> slots;
  int next_acquire = 0, next_collect = 0, pending = 0, acquired = -1;
  hipStream_t copy_stream = nullptr, d2h_stream = nullptr;
};

namespace {
void feed_free(ss_feed* f) {
  if (!f) return;
  for (auto& s : f->slots) {
    if (s.h_in) (void)hipHostFree(s.h_in);
    if (s.h_off) (void)hipHostFree(s.h_off);
    if (s.h_idx) (void)hipHostFree(s.h_idx);
    if (s.h_avg) (void)hipHostFree(s.h_avg);
    if (s.h_psd) (void)hipHostFree(s.h_psd);
    (void)hipFree(s.d_in);
    (void)hipFree(s.d_off);
    (void)hipFree(s.d_idx);
    (void)hipFree(s.d_avg);
    if (s.ev_h2d) (void)hipEventDestroy(s.ev_h2d);
    if (s.ev_done) (void)hipEventDestroy(s.ev_done);
  }
  if (f->copy_stream) (void)hipStreamDestroy(f->copy_stream);
  if (f->d2h_stream) (void)hipStreamDestroy(f->d2h_stream);
  delete f;
}
}  // namespace

extern "C" {

int ss_feed_create(ss_ctx* c, int32_t depth, int32_t cand_cap, int32_t want_psd, ss_feed** out) {
  if (!c || !out) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (depth < 2 || depth > 8 || cand_cap < 0) return fail(c, SS_ERR_INVALID, "feed depth %d (2..8) / cand_cap %d", depth, cand_cap);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  ss_feed* f = new ss_feed;
  f->c = c;
  f->depth = depth;
  f->cand_cap = cand_cap;
  f->want_psd = want_psd != 0;
  f->slots.resize((size_t)depth);
  const size_t in_bytes = in_bytes_per_sample(c->cfg.in_format) * (size_t)c->n * (size_t)c->cfg.max_batch;
  const size_t plane = sizeof(float) * (size_t)c->n * (size_t)c->cfg.max_batch;
  const size_t ncand = (size_t)(cand_cap > 0 ? cand_cap : 1);
#define FEED_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      feed_free(f);                                                                 \
      return fail(c, SS_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
    }                                                                               \
  } while (0)
  FEED_HIP(hipStreamCreateWithFlags(&f->copy_stream, hipStreamNonBlocking));
  FEED_HIP(hipStreamCreateWithFlags(&f->d2h_stream, hipStreamNonBlocking));
  for (auto& s : f->slots) {
    FEED_HIP(hipHostMalloc(&s.h_in, in_bytes, hipHostMallocDefault));
    FEED_HIP(hipMalloc(&s.d_in, in_bytes));
    FEED_HIP(hipHostMalloc((void**)&s.h_off, sizeof(int32_t) * ((size_t)c->cfg.max_batch + 1), hipHostMallocDefault));
    FEED_HIP(hipHostMalloc((void**)&s.h_idx, sizeof(int32_t) * ncand, hipHostMallocDefault));
    FEED_HIP(hipHostMalloc((void**)&s.h_avg, sizeof(float) * ncand, hipHostMallocDefault));
    if (f->want_psd) FEED_HIP(hipHostMalloc((void**)&s.h_psd, plane, hipHostMallocDefault));
    FEED_HIP(hipMalloc(&s.d_off, sizeof(int32_t) * ((size_t)c->cfg.max_batch + 1)));
    FEED_HIP(hipMalloc(&s.d_idx, sizeof(int32_t) * ncand));
    FEED_HIP(hipMalloc(&s.d_avg, sizeof(float) * ncand));
    FEED_HIP(hipEventCreateWithFlags(&s.ev_h2d, hipEventDisableTiming));
    FEED_HIP(hipEventCreateWithFlags(&s.ev_done, hipEventDisableTiming));
  }
#undef FEED_HIP
  *out = f;
  return SS_OK;
}

void ss_feed_destroy(ss_feed* f) {
  if (!f) return;
  {
    std::lock_guard<std::mutex> lock(f->c->mtx);
    (void)hipSetDevice(f->c->cfg.device_id);
    (void)hipStreamSynchronize(f->copy_stream);
    (void)hipStreamSynchronize(f->c->stream);
    (void)hipStreamSynchronize(f->d2h_stream);
  }
  feed_free(f);
}

int ss_feed_pending(const ss_feed* f) {
  if (!f) return 0;
  std::lock_guard<std::mutex> lock(f->c->mtx);
  return f->pending;
}

int ss_feed_acquire(ss_feed* f, void** frames) {
  if (!f || !frames) return SS_ERR_INVALID;
  ss_ctx* c = f->c;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (f->acquired >= 0) {  // acquiring twice without a submit hands out the same buffer
    *frames = f->slots[(size_t)f->acquired].h_in;
    return SS_OK;
  }
  ss_feed_slot& s = f->slots[(size_t)f->next_acquire];
  if (s.state != 0) return fail(c, SS_ERR_INVALID, "no free feed slot: collect a batch first");
  s.state = 1;
  f->acquired = f->next_acquire;
  f->next_acquire = (f->next_acquire + 1) % f->depth;
  *frames = s.h_in;
  return SS_OK;
}

int ss_feed_submit(ss_feed* f, int32_t nframes, const int64_t* t_ms, int64_t user_tag) {
  if (!f) return SS_ERR_INVALID;
  ss_ctx* c = f->c;
  std::lock_guard<std::mutex> lock(c->mtx);
  if (f->acquired < 0) return fail(c, SS_ERR_INVALID, "ss_feed_submit without ss_feed_acquire");
  if (nframes <= 0) return fail(c, SS_ERR_INVALID, "nframes %d", nframes);
  if (nframes > c->cfg.max_batch) return fail(c, SS_ERR_BATCH, "nframes %d > max_batch %d", nframes, c->cfg.max_batch);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  ss_feed_slot& s = f->slots[(size_t)f->acquired];
  const int n = c->n;
  const size_t row_bytes = in_bytes_per_sample(c->cfg.in_format) * (size_t)n;
  SS_HIP(c, hipMemcpyAsync(s.d_in, s.h_in, row_bytes * (size_t)nframes, hipMemcpyHostToDevice, f->copy_stream));
  SS_HIP(c, hipEventRecord(s.ev_h2d, f->copy_stream));
  SS_HIP(c, hipStreamWaitEvent(c->stream, s.ev_h2d, 0));
  NoiseState* z = nullptr;
  int st = get_noise(c, &z);
  if (st != SS_OK) return st;
  const int n_learn = plan_learning(c, z, nframes, t_ms);
  const bool want_cands = f->cand_cap > 0;
  st = run_batch(c, s.d_in, (long long)n, nframes, n_learn, z, nullptr, nullptr, nullptr, s.d_off, want_cands ? s.d_idx : nullptr,
                 want_cands ? s.d_avg : nullptr, f->cand_cap);
  if (st != SS_OK) return st;
  flush_stages(c);  // the slot's results are copied out right behind the batch
  SS_HIP(c, hipMemcpyAsync(s.h_off, s.d_off, sizeof(int32_t) * ((size_t)nframes + 1), hipMemcpyDeviceToHost, c->stream));
  if (f->want_psd) SS_HIP(c, hipMemcpyAsync(s.h_psd, c->last_psd, sizeof(float) * (size_t)n * (size_t)nframes, hipMemcpyDeviceToHost, c->stream));
  SS_HIP(c, hipEventRecord(s.ev_done, c->stream));
  s.nframes = nframes;
  s.tag = user_tag;
  s.state = 2;
  f->acquired = -1;
  ++f->pending;
  return SS_OK;
}

int ss_feed_collect(ss_feed* f, ss_feed_result* out) {
  if (!f || !out) return SS_ERR_INVALID;
  ss_ctx* c = f->c;
  ss_feed_slot* s = nullptr;
  {
    std::lock_guard<std::mutex> lock(c->mtx);
    if (f->pending == 0) return fail(c, SS_ERR_INVALID, "ss_feed_collect: nothing pending");
    s = &f->slots[(size_t)f->next_collect];
  }
  // wait outside the lock: a producer thread may acquire / submit the next slots meanwhile
  hipError_t e = hipEventSynchronize(s->ev_done);
  std::lock_guard<std::mutex> lock(c->mtx);
  if (e != hipSuccess) return fail(c, SS_ERR_HIP, "hipEventSynchronize failed: %s", hipGetErrorString(e));
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  const int total = s->h_off[s->nframes];
  const int ncopy = total < f->cand_cap ? total : f->cand_cap;
  if (ncopy > 0) {
    SS_HIP(c, hipMemcpyAsync(s->h_idx, s->d_idx, sizeof(int32_t) * (size_t)ncopy, hipMemcpyDeviceToHost, f->d2h_stream));
    SS_HIP(c, hipMemcpyAsync(s->h_avg, s->d_avg, sizeof(float) * (size_t)ncopy, hipMemcpyDeviceToHost, f->d2h_stream));
    SS_HIP(c, hipStreamSynchronize(f->d2h_stream));
  }
  out->nframes = s->nframes;
  out->status = (f->cand_cap > 0 && total > f->cand_cap) ? SS_ERR_CAND_OVERFLOW : SS_OK;
  out->user_tag = s->tag;
  out->cand_off = s->h_off;
  out->cand_idx = s->h_idx;
  out->cand_avg = s->h_avg;
  out->psd_db = f->want_psd ? s->h_psd : nullptr;
  s->state = 0;
  f->next_collect = (f->next_collect + 1) % f->depth;
  --f->pending;
  return SS_OK;
}

}  // extern "C"
