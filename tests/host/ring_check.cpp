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
  printf("%lld calls, %lld drains, %lld returns to the front; bad %d\n", calls, shifts, wraps, bad);
  return bad ? 1 : 0;
}
