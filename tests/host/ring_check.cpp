// Host-side check of the averager ring's placement rule (csrc/ring_place.h), compiled with g++ and run on the CPU by
// tests/test_host_index_algebra.py. Random streams of calls — sizes from one frame to the largest batch, buffers from the smallest
// the library allocates to roomy ones (and smaller than it allocates: the rule must hold whatever the size), contexts whose rows
// kernel writes whole batches into the buffer and contexts whose detect tiles write the newest H rows — are played through
// ring_place_decide, and for every call the rows it WRITES are checked against what is still to be READ:
//   * the detect stage of the call before — it rides beside this call's FFT stage or after it (scan_step.h: on the same launch, on
//     the row launch, or on the column launch of the call after) — reads its window [in, in + H) and its batch's rows where they
//     are in the buffer;
//   * this call's own detect stage reads this call's window.
// A call that drains the pending stages first (shift_first) has nothing before it to protect. With the buffer ss_create allocates
// for long transforms a stream of largest batches goes round the buffer without draining.
#include <cstdio>
#include <random>
#include <vector>

#include "../../rtl-sdr-scanner-cpp_amd/csrc/ring_place.h"

static bool overlap(int alo, int ahi, int blo, int bhi) { return alo < bhi && blo < ahi && alo < ahi && blo < bhi; }

int main() {
  const int H = 35;
  long long calls = 0, shifts = 0, wraps = 0;
  int bad = 0;
  std::mt19937 rng(12345);
  for (int trial = 0; trial < 6000; ++trial) {
    const bool whole = trial & 1;
    const int nprev = 1 + ((trial >> 1) & 1);  // (two spans to protect: the chain whose detect stage rides one launch later still)
    const int max_batch = 1 + (int)(rng() % 300);
    int rows = 3 * H + (int)(rng() % (12 * H));
    if (whole && (rng() & 1) && rows < (2 + nprev) * (max_batch + H)) rows = (2 + nprev) * (max_batch + H);  // (ss_create's size for long transforms; the other half: smaller)
    int start = 0;
    ss::RingPrev prev[2] = {{0, -1, 0}, {0, -1, 0}};
    const int fixed = (rng() % 3) ? 0 : 1 + (int)(rng() % max_batch);  // (a third of the streams: every call the same size)
    for (int call = 0; call < 200; ++call) {
      const int nframes = fixed ? fixed : 1 + (int)(rng() % max_batch);
      const ss::RingDecision d = ss::ring_place_decide(start, prev, nprev, rows, nframes, H, whole);
      ++calls;
      if (d.shift_first) {
        ++shifts;
        prev[0] = prev[1] = ss::RingPrev{0, -1, 0};
        if (d.in != 0) ++bad;
      } else if (d.in != start) {
        ++bad;
      }
      if (d.write_lo < 0 || d.write_hi > rows || d.write_lo >= d.write_hi) ++bad;
      if (d.in < 0 || d.in + H > rows || d.next_start < 0 || d.next_start + H > rows) ++bad;
      if (overlap(d.write_lo, d.write_hi, d.in, d.in + H)) ++bad;  // its own window
      for (int k = 0; k < nprev; ++k)
        if (prev[k].n > 0) {
          if (overlap(d.write_lo, d.write_hi, prev[k].start, prev[k].start + H)) ++bad;
          if (prev[k].batch >= 0 && overlap(d.write_lo, d.write_hi, prev[k].batch, prev[k].batch + prev[k].n)) ++bad;
        }
      // the next window is made of the newest rows: the batch's last H, or the old window's tail and the appended rows
      if (nframes >= H && d.batch >= 0 && (d.next_start != d.batch + nframes - H || d.write_lo != d.batch || d.write_hi != d.batch + nframes)) ++bad;
      if (nframes < H && (d.next_start != d.in + nframes || d.batch != d.in + H || d.write_lo != d.in + H || d.write_hi != d.in + H + nframes)) ++bad;
      if (nframes >= H && d.batch < 0 && (d.write_lo != d.next_start || d.write_hi != d.next_start + H)) ++bad;
      if (nframes >= H && d.batch == 0 && !d.shift_first && start != 0) ++wraps;
      prev[1] = prev[0];
      prev[0] = ss::RingPrev{d.in, whole ? d.batch : -1, nframes};
      start = d.next_start;
    }
  }
  // with the library's sizing a stream of largest batches never drains: the region goes round the buffer
  for (int nprev : {1, 2})
    for (int max_batch : {35, 48, 64, 100, 128, 200, 512}) {
      const int rows = (2 + nprev) * (max_batch + H);
      int start = 0, drains = 0;
      ss::RingPrev prev[2] = {{0, -1, 0}, {0, -1, 0}};
      for (int call = 0; call < 200; ++call) {
        const ss::RingDecision d = ss::ring_place_decide(start, prev, nprev, rows, max_batch, H, true);
        drains += d.shift_first ? 1 : 0;
        prev[1] = d.shift_first ? ss::RingPrev{0, -1, 0} : prev[0];
        prev[0] = ss::RingPrev{d.in, d.batch, max_batch};
        start = d.next_start;
      }
      if (drains) {
        printf("%d spans, max_batch %d: %d drains\n", nprev, max_batch, drains);
        ++bad;
      }
    }
  // The launch schedule itself (ring_place.h: ring_schedule): every chain's calls on its launch timeline — rows(k) written by launch
  // W(k), detect(k) riding on launch D(k) — with as many earlier spans protected as ring_spans_to_protect derives from the schedule.
  // Whatever a call writes must not touch what ANY earlier call's detect stage still has to read (D(k') >= W(k), no drain in between).
  // One span fewer must fail somewhere: the check is not vacuous.
  long long sched_calls = 0;
  for (int chain_no = 0; chain_no < 4; ++chain_no) {
    const ss::RingSchedule sc = ss::ring_schedule((ss::RingChain)chain_no);
    const int need = ss::ring_spans_to_protect(sc);
    const int expect[4] = {1, 1, 2, 2};
    if (need != expect[chain_no]) {
      printf("chain %d: %d spans derived, %d expected\n", chain_no, need, expect[chain_no]);
      ++bad;
    }
    for (int short_by = 0; short_by < 2; ++short_by) {
      const int nprev = need - short_by;
      if (nprev < 1) continue;
      int collisions = 0;
      std::mt19937 r2(777 + chain_no);
      for (int trial = 0; trial < 1500; ++trial) {
        const int max_batch = 1 + (int)(r2() % 300);
        // (buffers from the smallest that holds a window and a batch to the size ss_create allocates: the rule must hold whatever the size)
        const int rows_full = (2 + need) * (max_batch + H);
        const int rows = (r2() & 1) ? rows_full + (int)(r2() % (4 * H)) : 3 * H + (int)(r2() % (unsigned)(rows_full - 3 * H + 1));
        const int fixed = (r2() % 3) ? 0 : 1 + (int)(r2() % max_batch);
        struct Past {
          int start, batch, n;
          long detect_launch;
        };
        std::vector<Past> past;  // calls whose detect stage may still have to run
        int start = 0;
        ss::RingPrev prev[2] = {{0, -1, 0}, {0, -1, 0}};
        long k0 = 0;  // call index of the first call after the last drain (the drain completed everything before it)
        for (long k = 0; k < 120; ++k) {
          const int nframes = fixed ? fixed : 1 + (int)(r2() % max_batch);
          const ss::RingDecision d = ss::ring_place_decide(start, prev, nprev, rows, nframes, H, true);
          ++sched_calls;
          if (d.shift_first) {
            past.clear();
            prev[0] = prev[1] = ss::RingPrev{0, -1, 0};
            k0 = k;
          }
          const long w = sc.launches_per_call * k + sc.rows_at;
          for (const Past& q : past)
            if (q.detect_launch >= w && q.n > 0 &&
                (overlap(d.write_lo, d.write_hi, q.start, q.start + H) || (q.batch >= 0 && overlap(d.write_lo, d.write_hi, q.batch, q.batch + q.n))))
              ++collisions;
          past.push_back(Past{d.in, d.batch, nframes, sc.launches_per_call * (k + sc.detect_call_lag) + sc.detect_at});
          if (past.size() > 8) past.erase(past.begin());
          prev[1] = prev[0];
          prev[0] = ss::RingPrev{d.in, d.batch, nframes};
          start = d.next_start;
          (void)k0;
        }
      }
      if (short_by == 0 && collisions) {
        printf("chain %d with %d spans protected: %d collisions\n", chain_no, nprev, collisions);
        ++bad;
      }
      if (short_by == 1 && !collisions) {
        printf("chain %d with only %d spans protected: no collision found — the schedule check cannot fail\n", chain_no, nprev);
        ++bad;
      }
    }
  }
  printf("%lld calls, %lld drains, %lld returns to the front; %lld calls on the chains' launch timelines; bad %d\n", calls, shifts, wraps, sched_calls, bad);
  return bad ? 1 : 0;
}
