// Host-side check of the index algebra behind the tile culling of long transforms (csrc/fft256_kernels.h, csrc/detect_fused.h):
// compiled with hipcc for the HOST only and run on the CPU by tests/test_host_index_algebra.py — no GPU, no HIP call.
//   * rows_smax_index is a bijection from (tile column, run) onto a frame's row of run maxima, and it is the place the rows
//     kernel's workgroup (c, r0) writes for d = t: the 32-bin run that starts at bin (r0 + 256 c + (t << log_row)) ^ (N / 2);
//   * plan_long_cols keeps a plan workgroup inside its LDS and its 256 threads;
//   * plan_frame_tiles / plan_cols_per_wg (8192 points) keep the lists inside kLiveCap.
#include <cstdio>
#include <vector>

#include "../../rtl-sdr-scanner-cpp_amd/csrc/detect_fused.h"

int main() {
  int bad = 0;
  for (int logn : {16, 20}) {
    const int n = 1 << logn, cols = n >> 8, lognsub = logn - 16, nsub = 1 << lognsub, log_row = 8 + lognsub, half = n >> 1;
    std::vector<int> seen((size_t)(n >> 5), 0);
    for (int col = 0; col < cols; ++col)
      for (int run = 0; run < 8; ++run) {
        const int idx = ss::rows_smax_index(col, run, logn);
        if (idx < 0 || idx >= (n >> 5) || seen[(size_t)idx]++) ++bad;
      }
    // the writer: workgroup (c, r0), thread t < 256 -> word ((r0 >> 5) << (logn - 8)) + (c << 8) + t
    for (int c = 0; c < nsub; ++c)
      for (int r0 = 0; r0 < 256; r0 += 32)
        for (int t = 0; t < 256; ++t) {
          const int first = (r0 + (c << 8) + (t << log_row)) ^ half;  // first bin of the run, in output order
          const int col = first >> 8, run = (first >> 5) & 7;
          const int written = ((r0 >> 5) << (logn - 8)) + (c << 8) + t;
          if ((first & 31) != 0 || ss::rows_smax_index(col, run, logn) != written) ++bad;
        }
    printf("logn %d: %d tile columns x 8 runs, bad %d\n", logn, cols, bad);
  }
  for (int tile_cols : {256, 4096})
    for (int nframes = 1; nframes <= 600; ++nframes)
      for (int shift = 0; shift < 16; ++shift) {
        const int c = ss::plan_long_cols(nframes, shift, tile_cols), nft = (nframes + shift + 15) / 16, rows = 16 * nft + 20;
        if (c < 0 || c > 8 || (c > 0 && (c * rows > ss::kPlanLongFloats || c * nft > 256))) ++bad;
        if (nframes <= 128 && c < 1) ++bad;  // the benchmark's shapes are planned
      }
  for (int nframes = 35; nframes <= 4096; nframes += 7)
    for (int shift = 0; shift < 16; ++shift) {
      const int cols = ss::plan_cols_per_wg(nframes, shift), nft = ss::plan_frame_tiles(nframes, shift);
      if (cols != 0 && (cols * nft > ss::kLiveCap || cols * (nframes + (nframes >> 4) + 1) > ss::kPlanLdsFloats || 32 % cols != 0)) ++bad;
    }
  printf("bad %d\n", bad);
  return bad ? 1 : 0;
}
