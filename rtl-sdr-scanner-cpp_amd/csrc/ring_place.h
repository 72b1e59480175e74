// ring_place.h — where a batch reads the averager ring and where its newest rows go: the decision of specscan.hip's place_ring as a
// function of a few integers (no HIP in here: tests/host/ring_check.cpp runs it on the CPU against the invariant it exists for).
//
// The ring (Averager::m_rows of the reference, sources/radio/averager.cpp:14-25: the newest G - 1 rows, here H = G - 1 + TF - 1 = 35
// so that frame tiles align to the absolute frame index) is a window [start, start + H) sliding over a longer buffer of `rows` rows:
//   * a batch shorter than H appends its rows behind the window (the window moves on by the batch's frames);
//   * a batch of H frames and more writes a whole new window clear of the one being read (three windows rotate) — or, in a context
//     whose rows kernel writes ALL the batch's rows into the buffer (long transforms with tile culling: `whole`), a region for the
//     whole batch behind the window, whose last H rows are the next window; at the end of the buffer the region goes back to the front;
//   * when the rows would land on something that is still to be read, the pending stages are drained and the window is copied to the
//     front first (`shift_first`).
// The invariant (ring_check.cpp): the rows a call's FFT stage writes never overlap what the detect stage of the call BEFORE still
// has to read — its window and, in `whole` contexts, its batch's rows: that stage runs beside or after this call's FFT stage,
// wherever it rides (scan_step.h) — nor this call's own window. With the buffer ss_create allocates (three of the largest batches
// and three windows for long transforms) a stream of calls goes round the buffer without ever draining.
#pragma once

namespace ss {

struct RingPrev {  // what the detect stage of the call before reads (none: n = 0)
  int start;       // its window [start, start + H)
  int batch;       // its batch's rows [batch, batch + n) in the buffer (-1: they are not in the buffer)
  int n;
};
struct RingDecision {
  bool shift_first;  // drain the pending stages and copy the window [start, start + H) to [0, H) before anything else
  int in;            // first row of the window this call reads (after the shift, if any)
  int next_start;    // first row of the window the NEXT call reads; this call's newest min(nframes, H) rows are its last ones
  int batch;         // first row of the batch's own rows in the buffer: the region of a `whole` batch of >= H frames, the rows a
                     // batch of < H frames appends behind its window; -1: a batch of >= H frames that writes only its newest H rows
  int write_lo, write_hi;  // the rows this call's stages write
};

inline RingDecision ring_place_at(int start, int rows, int nframes, int H, bool whole) {
  RingDecision d{};
  d.in = start;
  if (nframes < H) {  // old rows [nframes, H) stay where they are, new ones land behind them
    d.next_start = start + nframes;
    d.batch = start + H;
    d.write_lo = start + H;
    d.write_hi = start + H + nframes;
  } else if (whole && H + nframes <= rows) {  // the whole batch behind the window, or at the front of the buffer
    int b = start + H;
    if (b + nframes > rows) b = 0;
    d.next_start = b + nframes - H;
    d.batch = b;
    d.write_lo = b;
    d.write_hi = b + nframes;
  } else {  // a whole new window, clear of the one being read (the ring holds at least three)
    d.next_start = start + 2 * H <= rows ? start + H : 0;
    d.batch = -1;
    d.write_lo = d.next_start;
    d.write_hi = d.next_start + H;
  }
  return d;
}

// prev[0 .. nprev): what the detect stages of the calls before still read when this call's rows are written — one in the chains
// whose detect stage rides beside or right behind the next call's FFT stage, two where it rides a launch later still (65536 points,
// one launch per call: SS_MERGE_65536)
inline RingDecision ring_place_decide(int start, const RingPrev* prev, int nprev, int rows, int nframes, int H, bool whole) {
  const auto hits = [](int lo, int hi, int a, int b) { return lo < b && a < hi && a < b; };
  RingDecision d = ring_place_at(start, rows, nframes, H, whole);
  bool bad = d.write_lo < 0 || d.write_hi > rows || d.next_start + H > rows || hits(d.write_lo, d.write_hi, start, start + H);
  for (int k = 0; k < nprev && !bad; ++k)
    bad = prev[k].n > 0 && (hits(d.write_lo, d.write_hi, prev[k].start, prev[k].start + H) ||
                            (prev[k].batch >= 0 && hits(d.write_lo, d.write_hi, prev[k].batch, prev[k].batch + prev[k].n)));
  if (bad) {  // drain, window to the front, place again (nothing before it to protect then)
    d = ring_place_at(0, rows, nframes, H, whole);
    d.shift_first = true;
  }
  return d;
}
inline RingDecision ring_place_decide(int start, const RingPrev& prev, int rows, int nframes, int H, bool whole) {
  return ring_place_decide(start, &prev, 1, rows, nframes, H, whole);
}

// HOW MANY earlier calls' spans a call must keep clear of follows from where the chain's launches put the stages (scan_step.h): with
// L launches per call, the rows of call k written by launch L k + rows_at and the detect stage of call k riding on launch
// L (k + detect_call_lag) + detect_at, the detect stage of call k - j has not finished before rows(k) are written exactly when
// L (k - j + detect_call_lag) + detect_at >= L k + rows_at (the same launch counts: its roles run side by side). The chains:
//   ROWS_THEN_DETECT  two launches per call — column half, row half; rows(k) by the row launch, detect(k - 1) riding on it (2^20 points;
//                     65536 points with SS_DET_LAG2=0)                                                                       -> 1
//   DET_LAG2          the same two launches, detect(k - 2) on the COLUMN launch of call k (65536 points, calls the fold does not take)  -> 1
//   MERGED            one launch per call, the row half of call k - 1 beside the column half of call k, detect(k - 3) riding (KIND 7)    -> 2
//   FOLD              one launch per call that writes the call's own rows, detect(k - 2) riding (KIND 8)                       -> 2
// tests/host/ring_check.cpp plays random streams through each chain's launch timeline with this many spans protected (and, to show
// that the check can fail, with one fewer).
enum RingChain { RING_ROWS_THEN_DETECT = 0, RING_DET_LAG2 = 1, RING_MERGED = 2, RING_FOLD = 3 };
struct RingSchedule {
  int launches_per_call, rows_at, detect_call_lag, detect_at;
};
constexpr RingSchedule ring_schedule(RingChain c) {
  return c == RING_ROWS_THEN_DETECT ? RingSchedule{2, 1, 1, 1} : c == RING_DET_LAG2 ? RingSchedule{2, 1, 2, 0} : c == RING_MERGED ? RingSchedule{1, 1, 3, 0} : RingSchedule{1, 0, 2, 0};
}
constexpr int ring_spans_to_protect(RingSchedule s) {
  int n = 0;
  for (int j = 1; j <= 8; ++j)
    if (s.launches_per_call * (s.detect_call_lag - j) + s.detect_at >= s.rows_at) ++n;
  return n;
}

}  // namespace ss
