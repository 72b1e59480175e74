// specscan_pipe_impl.h — ss_pipe_* (include/specscan.h): textually included at the end of specscan.hip, inside its extern "C"
// block, because it drives the contexts through that file's internals (run_batch, plan_learning, the ss_ctx fields).
// ---------------------------------------------------------------------------------------------------------------
// ss_pipe: several lanes (contexts) taking the calls of one band in turn — include/specscan.h
//
// What a lane's turn costs the host decides whether lanes pay: every HIP call is 3-4 us here (also from several threads:
// the runtime serialises them), a batch takes ~35 us on the device with two or three lanes. A turn is therefore kept to six
// calls and no events: the tail of call k is copied onto the stream of the lane that takes call k + 1 (its own slot,
// written and read in stream order), the averager restart is bookkeeping only (whatever the ring holds reaches halo frames
// only, see kHaloMax), and the halo runs the chain for the ring alone (FFT + detect, no candidates, no emit).
// ---------------------------------------------------------------------------------------------------------------
struct ss_pipe {
  // A frame's time mean slides from the first frame of its 16-frame tile (k_detect_fused), whose own 21-term sum reaches
  // 20 frames further back: a lane that starts at frame s must therefore re-scan from the tile boundary at or below
  // floor(s / 16) * 16 - 20, i.e. from floor(s / 16) * 16 - 32: 32..47 frames. Frames of the halo itself may see what the
  // ring held before (the first tiles reach back into it); they are dropped.
  static constexpr int kHaloMax = 47;
  static constexpr int kSmallCall = 64;  // calls shorter than this go to every lane (a halo must fit inside the previous call)
  struct Lane {
    ss_ctx* c = nullptr;
    long long next_abs = 0;   // absolute index of the frame the lane expects next
    void* d_tail = nullptr;   // the last kHaloMax input frames of some call, items of N samples (the decimated part of the items)
    long long tail_call = -1;  // which call's
  };
  ss_config cfg{};
  std::vector<Lane> lanes;
  int turn = 0;
  long long abs = 0;         // frames since the last reset of the pipe
  long long prev_start = 0;  // absolute start and length of the previous call
  int prev_n = 0;
  long long calls = 0;
  std::vector<std::pair<int32_t, long long>> seen;  // noise learning per centre frequency, by frame count (plan_learning)
  int32_t range_lo = 0, range_hi = 0;
  std::mutex mtx;
  char err[512] = {0};
};

namespace {
int pipe_fail(ss_pipe* p, int status, const char* fmt, ...) {
  char* dst = p ? p->err : g_create_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return status;
}

long long* pipe_seen(ss_pipe* p) {
  const int32_t center = (p->range_lo + p->range_hi) / 2;
  for (auto& kv : p->seen)
    if (kv.first == center) return &kv.second;
  p->seen.emplace_back(center, 0);
  return &p->seen.back().second;
}

int lane_status(ss_pipe* p, ss_ctx* lane, int st) {
  if (st != SS_OK) snprintf(p->err, sizeof(p->err), "%s", lane->err);
  return st;
}

// Averager::reset + the halo: afterwards the lane's ring holds the frames before `start` and its tile origin is the halo's first frame
int lane_rescan(ss_pipe* p, ss_pipe::Lane& L, int halo) {
  ss_ctx* c = L.c;
  const size_t row = (size_t)p->cfg.fft_size * in_bytes_per_sample(p->cfg.in_format);
  const char* tail = static_cast<const char*>(L.d_tail) + (size_t)(ss_pipe::kHaloMax - halo) * row;
  if (!c->fused) {  // per-stage back end: the plain restart and a plain batch whose outputs nobody takes
    int st = ss_reset(c);
    if (st != SS_OK) return lane_status(p, c, st);
    std::lock_guard<std::mutex> lane_lock(c->mtx);
    NoiseState* z = nullptr;
    st = get_noise(c, &z);
    if (st == SS_OK) st = run_batch(c, tail, (long long)p->cfg.fft_size, halo, plan_learning(c, z, halo, nullptr), z, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    return lane_status(p, c, st);
  }
  std::lock_guard<std::mutex> lane_lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  c->frames_pushed = 0;  // ss_reset's bookkeeping without clearing the ring
  c->abs_frames = 0;
  c->rot_frames = 0;
  c->last_n = 0;
  NoiseState* z = nullptr;
  int st = get_noise(c, &z);
  if (st == SS_OK) {
    c->history_only = true;
    st = run_batch(c, tail, (long long)p->cfg.fft_size, halo, plan_learning(c, z, halo, nullptr), z, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    c->history_only = false;
  }
  return lane_status(p, c, st);
}

// 8192 points, 21 x 21: the restart, the halo and the call as ONE batch of halo + nframes frames — the FFT kernel reads the
// frames below `halo` from the lane's tail slot and the rest from the caller, the detect kernel reads PSD rows from the two
// planes accordingly and lets only the caller's frames report candidates, the emit kernel numbers frames from `halo` on.
// Three launches instead of five, and no small kernels in front of the big ones.
int lane_rescan_and_process(ss_pipe* p, ss_pipe::Lane& L, int halo, const void* d_iq, int nframes, float* d_psd_db, int32_t* d_cand_off,
                            int32_t* d_cand_idx, float* d_cand_avg, int cand_cap) {
  ss_ctx* c = L.c;
  const size_t row = (size_t)p->cfg.fft_size * in_bytes_per_sample(p->cfg.in_format);
  const char* tail = static_cast<const char*>(L.d_tail) + (size_t)(ss_pipe::kHaloMax - halo) * row;
  std::lock_guard<std::mutex> lane_lock(c->mtx);
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  c->frames_pushed = 0;  // ss_reset's bookkeeping without clearing the ring
  c->abs_frames = 0;
  c->rot_frames = 0;
  c->last_n = 0;
  NoiseState* z = nullptr;
  int st = get_noise(c, &z);
  if (st != SS_OK) return lane_status(p, c, st);
  const int n_learn = plan_learning(c, z, halo + nframes, nullptr);  // 0: lanes take turns only once the ceiling is learned
  c->two.iq_b = d_iq;
  c->two.stride_b = (long long)c->n * c->cfg.decim;
  c->two.psd_b = d_psd_db ? d_psd_db : c->d_psd + (size_t)halo * (size_t)c->n;
  c->two.split = halo;
  st = run_batch(c, tail, (long long)c->n, halo + nframes, n_learn, z, nullptr, nullptr, nullptr, d_cand_off, d_cand_idx, d_cand_avg, cand_cap);
  c->two = ss_ctx::Two{};
  return lane_status(p, c, st);
}

// the last kHaloMax frames of a call's input -> the lane's slot, on the lane's stream
int lane_keep_tail(ss_pipe* p, ss_pipe::Lane& L, const void* d_iq, int nframes, long long call) {
  ss_ctx* c = L.c;
  const size_t sample = in_bytes_per_sample(p->cfg.in_format);
  const size_t n = (size_t)p->cfg.fft_size;
  const size_t item = n * (size_t)p->cfg.decim * sample;
  const char* src = static_cast<const char*>(d_iq) + (size_t)(nframes - ss_pipe::kHaloMax) * item;
  SS_HIP(c, hipSetDevice(c->cfg.device_id));
  if ((n * sample) % 16 == 0 && item % 16 == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
    // (a kernel launch costs the host a third of what hipMemcpy2DAsync does)
    hipLaunchKernelGGL(ss::k_copy_pitched, dim3(256), dim3(256), 0, c->stream, reinterpret_cast<const uint4*>(src), item / 16,
                       static_cast<uint4*>(L.d_tail), (int)(n * sample / 16), ss_pipe::kHaloMax);
  } else {
    SS_HIP(c, hipMemcpy2DAsync(L.d_tail, n * sample, src, item, n * sample, (size_t)ss_pipe::kHaloMax, hipMemcpyDeviceToDevice, c->stream));
  }
  L.tail_call = call;
  return SS_OK;
}
}  // namespace

const char* ss_pipe_last_error(const ss_pipe* pipe) { return pipe ? pipe->err : g_create_err; }

void ss_pipe_destroy(ss_pipe* p) {
  if (!p) return;
  for (auto& L : p->lanes) {
    if (L.c) (void)hipStreamSynchronize(L.c->stream);
  }
  for (auto& L : p->lanes) {
    (void)hipFree(L.d_tail);
    ss_destroy(L.c);
  }
  delete p;
}

int ss_pipe_create(const ss_config* cfg, int32_t lanes, ss_pipe** out) {
  if (!cfg || !out) return pipe_fail(nullptr, SS_ERR_INVALID, "null argument");
  *out = nullptr;
  if (lanes < 1 || lanes > 4) return pipe_fail(nullptr, SS_ERR_INVALID, "lanes %d not in 1..4", lanes);
  if (cfg->max_batch < ss_pipe::kSmallCall) return pipe_fail(nullptr, SS_ERR_INVALID, "max_batch %d < %d", cfg->max_batch, ss_pipe::kSmallCall);
  if (cfg->flags & (SS_FLAG_SPECTROGRAM | SS_FLAG_KEEP_PLANES)) return pipe_fail(nullptr, SS_ERR_INVALID, "per-context flags are not available on a pipe");
  if (cfg->grouping_y - 1 > 20) return pipe_fail(nullptr, SS_ERR_INVALID, "grouping_y %d: the halo covers 20 frames of history", cfg->grouping_y);
  ss_pipe* p = new (std::nothrow) ss_pipe();
  if (!p) return pipe_fail(nullptr, SS_ERR_NOMEM, "out of host memory");
  p->cfg = *cfg;
  p->range_lo = cfg->range_lo;
  p->range_hi = cfg->range_hi;
  const size_t tail_bytes = (size_t)ss_pipe::kHaloMax * (size_t)cfg->fft_size * in_bytes_per_sample(cfg->in_format);
  ss_config lane_cfg = *cfg;
  // a lane may take its halo and the call as one batch; the fused back end stops at 65536 frames per batch, so a pipe whose
  // calls may come that close keeps the lanes at the limit and runs such calls' halos as batches of their own
  lane_cfg.max_batch = (cfg->max_batch <= 65536 && cfg->max_batch + ss_pipe::kHaloMax > 65536) ? 65536 : cfg->max_batch + ss_pipe::kHaloMax;
  for (int l = 0; l < lanes; ++l) {
    ss_pipe::Lane L;
    const int st = ss_create(&lane_cfg, &L.c);
    if (st != SS_OK) {
      ss_pipe_destroy(p);
      return st;  // ss_last_error(NULL) holds the message
    }
    p->lanes.push_back(L);
    if (hipMalloc(&p->lanes.back().d_tail, tail_bytes) != hipSuccess) {
      ss_pipe_destroy(p);
      return pipe_fail(nullptr, SS_ERR_HIP, "ss_pipe_create: tail slot allocation failed");
    }
  }
  *out = p;
  return SS_OK;
}

int ss_pipe_sync(ss_pipe* p) {
  if (!p) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(p->mtx);
  for (auto& L : p->lanes) {
    const int st = ss_sync(L.c);
    if (st != SS_OK) return lane_status(p, L.c, st);
  }
  return SS_OK;
}

int ss_pipe_set_frequency_range(ss_pipe* p, int32_t lo_hz, int32_t hi_hz) {
  if (!p) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(p->mtx);
  p->range_lo = lo_hz;
  p->range_hi = hi_hz;
  for (auto& L : p->lanes) {
    const int st = ss_set_frequency_range(L.c, lo_hz, hi_hz);
    if (st != SS_OK) return lane_status(p, L.c, st);
  }
  return SS_OK;
}

int ss_pipe_reset(ss_pipe* p) {
  if (!p) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(p->mtx);
  for (auto& L : p->lanes) {
    const int st = ss_reset(L.c);
    if (st != SS_OK) return lane_status(p, L.c, st);
    L.next_abs = 0;
    L.tail_call = -1;
  }
  p->abs = 0;
  p->prev_start = 0;
  p->prev_n = 0;
  return SS_OK;
}

int ss_pipe_process_device(ss_pipe* p, const void* d_iq, int32_t nframes, float* d_psd_db, int32_t* d_cand_off, int32_t* d_cand_idx,
                           float* d_cand_avg, int32_t cand_cap) {
  if (!p) return SS_ERR_INVALID;
  std::lock_guard<std::mutex> lock(p->mtx);
  if (nframes < 0 || (nframes > 0 && !d_iq) || cand_cap < 0) return pipe_fail(p, SS_ERR_INVALID, "bad d_iq/nframes/cand_cap");
  if (nframes > p->cfg.max_batch) return pipe_fail(p, SS_ERR_BATCH, "nframes %d > max_batch %d", nframes, p->cfg.max_batch);
  const int nlanes = (int)p->lanes.size();
  if (nframes == 0) return lane_status(p, p->lanes[0].c, ss_process_device(p->lanes[0].c, d_iq, 0, d_psd_db, nullptr, nullptr, d_cand_off, d_cand_idx, d_cand_avg, cand_cap));
  const long long start = p->abs;
  long long* seen = pipe_seen(p);
  const bool learning = *seen < p->cfg.learn_frames;  // this call still holds learning frames: every lane must take it
  // halo of a lane that missed the previous call: back to the tile boundary 32 frames below this call's first tile
  const int halo = 32 + (int)(start % 16);
  const bool halo_ok = p->prev_n >= ss_pipe::kHaloMax && p->prev_start + p->prev_n == start && start - halo >= 0;
  const bool everyone = nlanes == 1 || learning || nframes < ss_pipe::kSmallCall;
  const int owner = everyone ? 0 : p->turn;

  for (int l = 0; l < nlanes; ++l) {
    if (!everyone && l != owner) continue;
    ss_pipe::Lane& L = p->lanes[(size_t)l];
    if (L.next_abs != start) {
      if (!halo_ok) return pipe_fail(p, SS_ERR_INVALID, "lane %d is not contiguous and no halo is available (internal)", l);
      if (L.tail_call != p->calls - 1) {
        // only the lane next in turn was given the previous call's tail; a short or learning call after calls taken in turn
        // needs it on the others too (three lanes or more): let the device finish and copy it across. Rare.
        const ss_pipe::Lane* from = nullptr;
        for (const auto& o : p->lanes)
          if (o.tail_call == p->calls - 1) from = &o;
        if (!from) return pipe_fail(p, SS_ERR_INVALID, "the previous call's tail was not kept (internal)");
        for (auto& o : p->lanes) {
          const int st = ss_sync(o.c);
          if (st != SS_OK) return lane_status(p, o.c, st);
        }
        const size_t tail_bytes = (size_t)ss_pipe::kHaloMax * (size_t)p->cfg.fft_size * in_bytes_per_sample(p->cfg.in_format);
        SS_HIP(L.c, hipMemcpyAsync(L.d_tail, from->d_tail, tail_bytes, hipMemcpyDeviceToDevice, L.c->stream));
        L.tail_call = p->calls - 1;
      }
      if (L.c->fused && L.c->use_fft8192 && L.c->fft8192_variant != 2 && !L.c->diag.fft_ablate && L.c->diag.fft_stamp_path.empty() &&
          halo + nframes <= L.c->cfg.max_batch) {
        const bool out = l == owner;
        const int st = lane_rescan_and_process(p, L, halo, d_iq, nframes, out ? d_psd_db : nullptr, out ? d_cand_off : nullptr,
                                               out ? d_cand_idx : nullptr, out ? d_cand_avg : nullptr, out ? cand_cap : 0);
        if (st != SS_OK && st != SS_ERR_CAND_OVERFLOW) return st;
        L.next_abs = start + nframes;
        continue;
      }
      const int st = lane_rescan(p, L, halo);
      if (st != SS_OK) return st;
    }
    const int st = l == owner ? ss_process_device(L.c, d_iq, nframes, d_psd_db, nullptr, nullptr, d_cand_off, d_cand_idx, d_cand_avg, cand_cap)
                              : ss_process_device(L.c, d_iq, nframes, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
    if (st != SS_OK && st != SS_ERR_CAND_OVERFLOW) return lane_status(p, L.c, st);
    L.next_abs = start + nframes;
  }
  if (!everyone) {
    // the lane next in turn will have missed this call: give it the tail now, on its own stream (d_iq stays valid until
    // ss_pipe_sync, by contract)
    p->turn = (p->turn + 1) % nlanes;
    if (nframes >= ss_pipe::kHaloMax) {
      const int st = lane_keep_tail(p, p->lanes[(size_t)p->turn], d_iq, nframes, p->calls);
      if (st != SS_OK) return st;
    }
  }
  *seen += nframes;
  p->prev_start = start;
  p->prev_n = nframes;
  p->abs = start + nframes;
  ++p->calls;
  return SS_OK;
}
