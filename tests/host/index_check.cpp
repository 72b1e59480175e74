// Host-side check of the index algebra behind the tile culling of long transforms (csrc/fft256_kernels.h, csrc/detect_fused.h):
// compiled with hipcc for the HOST only and run on the CPU by tests/test_host_index_algebra.py — no GPU, no HIP call.
//   * rows_smax_index is a bijection from (tile column, run) onto a frame's row of run maxima, and it is the place the rows
//     kernel's workgroup (c, r0) writes for d = t: the 32-bin run that starts at bin (r0 + 256 c + (t << log_row)) ^ (N / 2);
//   * plan_long_cols keeps a plan workgroup inside its LDS and its 256 threads;
//   * plan_frame_tiles / plan_cols_per_wg (8192 points) keep the lists inside kLiveCap;
//   * 2^20 points in two passes (csrc/fft1024_kernels.h): the block decodes of the column tiles, the row tiles and the plan are
//     bijections that put the workgroups sharing a 128-byte line on one XCD; rows1024_smax_index is where the row tiles write and
//     the plan reads; the window in the column tiles' order is a permutation of the taps, element by element what the tile loads;
//     the atomic maxima's keys keep the order of the floats;
//   * the fold's rows (csrc/fft65536_dif8.h): the layout's bijection, the epilogue's store address, a tile's loads (bottom of main).
#include <cmath>
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
  // (the limits of the 8192-point plan since the halo frames' maxima have a place in every column, round 5: a plan workgroup takes 8 tile
  // columns up to 1081-frame batches — the benchmark's 1024 among them —, 4 up to 2210; 2211 frames and more are not planned: DESIGN.md 4.1)
  if (ss::plan_cols_per_wg(1024, 0) != 8 || ss::plan_cols_per_wg(1081, 0) != 8 || ss::plan_cols_per_wg(1082, 0) != 4 || ss::plan_cols_per_wg(2210, 0) != 4 ||
      ss::plan_cols_per_wg(2211, 0) != 0)
    ++bad;
  // ---- 2^20 points in two passes (csrc/fft1024_kernels.h) ----
  {
    const int n = 1 << 20, half = n >> 1, runs = n >> 5;
    // rows1024_smax_index is a bijection of the runs, and the place the four row tiles that share a run write
    std::vector<int> seen((size_t)runs, 0), written((size_t)runs, 0);
    for (int run = 0; run < runs; ++run) {
      const int idx = ss::rows1024_smax_index(run);
      if (idx < 0 || idx >= runs || seen[(size_t)idx]++) ++bad;
    }
    // the row tiles: every (frame, 8-row group) exactly once; the four tiles of a 32-row group — which fill the dB plane's lines and
    // share every run — in consecutive slots of one XCD; every run of a frame written (by four tiles each)
    for (int frames : {1, 3}) {
      std::vector<int> rows_seen((size_t)frames * 128, 0);
      for (int b = 0; b < frames * 128; ++b) {
        int f, r0;
        ss::rows1024_block(b, &f, &r0);
        if (f < 0 || f >= frames || r0 < 0 || r0 >= 1024 || (r0 & 7) || rows_seen[(size_t)f * 128 + (r0 >> 3)]++) ++bad;
        int f2, r2;
        ss::rows1024_block(b ^ 8, &f2, &r2);  // the neighbouring slot of the same XCD
        if (((r0 >> 3) & 3) < 2 && (f2 != f || (r2 >> 5) != (r0 >> 5))) ++bad;
        if (f == 0)
          for (int k2 = 0; k2 < 1024; ++k2) ++written[(size_t)ss::rows1024_smax_index(((r0 + (k2 << 10)) ^ half) >> 5)];
      }
    }
    for (int run = 0; run < runs; ++run)
      if (written[(size_t)run] != 8) ++bad;  // (4 tiles x the two `frames` passes above)
    // ---- 262144 points behind 256-point column tiles (round 6): the same row tile over 256 rows — every (frame, 8-row group) once, the four
    // tiles of a 32-row group in consecutive slots of one XCD, every word of a frame's 8192 run maxima written by four tiles, a tile
    // column's eight runs at [run][column] with the column in output order (k2 ^ 512) ----
    {
      const int n2 = 1 << 18, half2 = n2 >> 1, runs2 = n2 >> 5;
      std::vector<int> written2((size_t)runs2, 0);
      for (int frames : {1, 3}) {
        std::vector<int> rows_seen((size_t)frames * 32, 0);
        for (int b = 0; b < frames * 32; ++b) {
          int f, r0;
          ss::rows1024_block<8>(b, &f, &r0);
          if (f < 0 || f >= frames || r0 < 0 || r0 >= 256 || (r0 & 7) || rows_seen[(size_t)f * 32 + (r0 >> 3)]++) ++bad;
          int f2, r2;
          ss::rows1024_block<8>(b ^ 8, &f2, &r2);  // the neighbouring slot of the same XCD
          if (((r0 >> 3) & 3) < 2 && (f2 != f || (r2 >> 5) != (r0 >> 5))) ++bad;
          if (f == 0)
            for (int k2 = 0; k2 < 1024; ++k2) {
              const int idx = ss::rows1024x256_smax_index(r0 >> 5, k2 ^ 512);
              const int bin = (r0 + (k2 << 8)) ^ half2;  // the tile's first bin of this k2, in output order
              if (idx < 0 || idx >= runs2 || (bin >> 8) != (k2 ^ 512) || ((bin & 255) >> 5) != (r0 >> 5)) ++bad;
              else ++written2[(size_t)idx];
            }
        }
      }
      for (int run = 0; run < runs2; ++run)
        if (written2[(size_t)run] != 8) ++bad;  // (4 tiles x the two `frames` passes above)
      // the column tiles clear 256 words each: 32 tiles cover the frame's 8192 words once
      if (ss::plan_x256_blocks(32, 0) != 64 || ss::plan_x256_blocks(32, 5) != 96 || ss::plan_x256_blocks(16, 0) != 32) ++bad;
    }
    // the column tiles, 8 and 16 columns wide: every (frame, tile) once; the window in the kernel's order is a permutation of the taps
    for (int logc : {3, 4}) {
      const int tiles = 1024 >> logc;
      std::vector<int> tseen((size_t)2 * tiles, 0);
      for (int b = 0; b < 2 * tiles; ++b) {
        int f, tile;
        ss::cols1024_block(b, logc, &f, &tile);
        if (f < 0 || f > 1 || tile < 0 || tile >= tiles || tseen[(size_t)f * tiles + tile]++) ++bad;
        int f2, t2;
        ss::cols1024_block(b ^ 8, logc, &f2, &t2);
        if (logc == 3 && (b & 8) == 0 && (f2 != f || (t2 >> 1) != (tile >> 1))) ++bad;  // the two tiles of a 128-byte line: neighbouring slots of one XCD
      }
      std::vector<float> win((size_t)n), out((size_t)n, -1.0f);
      for (int i = 0; i < n; ++i) win[(size_t)i] = (float)i;
      ss::fft1024_window_order(win.data(), out.data(), logc);
      std::vector<char> hit((size_t)n, 0);
      for (int i = 0; i < n; ++i) {
        const int v = (int)out[(size_t)i];
        if (v < 0 || v >= n || hit[(size_t)v]++) ++bad;
      }
      // thread t of tile `tile`, r-th load: sample 4 (j + 16 r) + q of column tile * cols + c
      const int cols = 1 << logc, nsub = 4 * cols, threads = 16 * nsub;
      for (int tile : {0, 5, tiles - 1})
        for (int t : {0, 1, nsub - 1, nsub, threads - 1})
          for (int r : {0, 7, 15}) {
            const int sub = t & (nsub - 1), j = t / nsub, c = sub & (cols - 1), q = sub >> logc;
            if ((int)out[((size_t)tile * threads + t) * 16 + r] != ((4 * (j + 16 * r) + q) << 10) + tile * cols + c) ++bad;
          }
    }
    // the plan's block decode, layout 1: every (quarter, group of C consecutive k2) exactly once, the groups that share a line on one XCD
    for (int C : {1, 2, 4, 8, 16, 32, 5}) {
      const int blocks = ss::plan_long_blocks(1, C, n), groups = blocks / 4;
      std::vector<int> gseen((size_t)blocks, 0);
      for (int b = 0; b < blocks; ++b) {
        int wc, d0;
        ss::plan_long_block(1, b, C, 4, &wc, &d0);
        if (wc < 0 || wc > 3 || d0 % C || d0 / C >= groups || gseen[(size_t)wc * groups + d0 / C]++) ++bad;
        if (32 % C == 0 && C < 32) {
          int wc2, d2;
          ss::plan_long_block(1, b ^ 8, C, 4, &wc2, &d2);
          if (wc2 != wc || (d2 * 4 + wc2) / 128 != (d0 * 4 + wc) / 128) { /* neighbouring slots of an XCD: the same 32 k2, i.e. the same lines */
            if ((d2 / 32) != (d0 / 32)) ++bad;
          }
        }
      }
      if (groups * C < 1024) ++bad;
    }
    // the plan at the front of the next call's column launch (k_fft_cols1024_plan): workgroup b < plan_fused_wgs(blocks) runs, in its
    // four groups of 256 threads, the plan blocks ((b >> 3) * 4 + sub) * 8 + (b & 7): every block of k_plan_long's numbering exactly
    // once, on the XCD that numbering gives it (b mod 8), the workgroup's four in consecutive slots; the blocks past the numbering's
    // end find no column; the workgroup count is a multiple of 8 (the column tiles behind it keep their XCDs); what a block needs of
    // LDS fits a quarter of a column tile's
    for (int C : {1, 2, 4, 8}) {
      const int blocks = ss::plan_long_blocks(1, C, n), wgs = ss::plan_fused_wgs(blocks);
      std::vector<int> bseen((size_t)4 * wgs, 0);
      if (wgs % 8 || 4 * wgs < blocks) ++bad;
      for (int b = 0; b < wgs; ++b)
        for (int sub = 0; sub < 4; ++sub) {
          const int vb = ((((b >> 3) << 2) + sub) << 3) | (b & 7);
          if (vb < 0 || vb >= 4 * wgs || bseen[(size_t)vb]++ || (vb & 7) != (b & 7) || (vb >> 3) != (b >> 3) * 4 + sub) ++bad;
          if (vb >= blocks) {
            int wc, d0;
            ss::plan_long_block(1, vb, C, 4, &wc, &d0);
            if (d0 < 1024) ++bad;
          }
        }
      for (int vb = 0; vb < blocks; ++vb)
        if (!bseen[(size_t)vb]) ++bad;
    }
    for (int nframes = 1; nframes <= 600; ++nframes)
      for (int shift = 0; shift < 16; ++shift) {
        const int c = ss::plan_long_cols(nframes, shift, 4096, 8, ss::kPlanFusedFloats), nft = (nframes + shift + 15) / 16, rows = 16 * nft + 20;
        if (c < 0 || c > 8 || (c > 0 && (c * rows > ss::kPlanFusedFloats || c * nft > 256))) ++bad;
        if (nframes <= 128 && c < 1) ++bad;
      }
    if (4 * (ss::kPlanFusedFloats + ss::kPlanLongInts) * 4 > ss::fft1024_cols_lds_bytes(4)) ++bad;
    // the Hamming taps the column tiles form (WCALC) against the taps as ss_create computes them: at most 1.2e-7 apart
    {
      std::vector<float2> wt(65536);
      ss::fft1024_window_rotation_table(wt.data());
      double worst = 0.0, sq = 0.0;
      for (int r = 0; r < 16; ++r) {
        const double phi = 2.0 * 3.14159265358979323846 * 65536.0 * r / 1048575.0;
        if (ss::kWin1024C[r] != (float)(-0.46 * cos(phi)) || ss::kWin1024S[r] != (float)(0.46 * sin(phi))) ++bad;
        for (int m = 0; m < 65536; ++m) {
          const float w = __builtin_fmaf(wt[(size_t)m].x, ss::kWin1024C[r], __builtin_fmaf(wt[(size_t)m].y, ss::kWin1024S[r], 0.54f));
          const float M = (float)(n - 1);
          const float ref = (float)(0.54 - 0.46 * cos((2.0 * 3.14159265358979323846 * (double)(m + 65536 * r)) / M));
          const double d = (double)w - (double)ref;
          worst = d < 0 ? (-d > worst ? -d : worst) : (d > worst ? d : worst);
          sq += d * d;
        }
      }
      printf("formed Hamming taps: worst %.3g, rms %.3g\n", worst, sqrt(sq / n));
      if (worst > 1.5e-7 || sqrt(sq / n) > 5e-8) ++bad;
    }
    // the same for the 256-point column tiles of a 65536-point frame (fft256_kernels.h): n = m + 4096 r, m < 4096
    {
      std::vector<float2> wt(4096);
      ss::fft65536_window_rotation_table(wt.data());
      double worst = 0.0, sq = 0.0;
      for (int r = 0; r < 16; ++r) {
        const double phi = 2.0 * 3.14159265358979323846 * 4096.0 * r / 65535.0;
        if (ss::kWin65536C[r] != (float)(-0.46 * cos(phi)) || ss::kWin65536S[r] != (float)(0.46 * sin(phi))) ++bad;
        for (int m = 0; m < 4096; ++m) {
          const float w = __builtin_fmaf(wt[(size_t)m].x, ss::kWin65536C[r], __builtin_fmaf(wt[(size_t)m].y, ss::kWin65536S[r], 0.54f));
          const float M = 65535.0f;
          const float ref = (float)(0.54 - 0.46 * cos((2.0 * 3.14159265358979323846 * (double)(m + 4096 * r)) / M));
          const double d = (double)w - (double)ref;
          worst = d < 0 ? (-d > worst ? -d : worst) : (d > worst ? d : worst);
          sq += d * d;
        }
      }
      printf("formed Hamming taps, 65536 points: worst %.3g, rms %.3g\n", worst, sqrt(sq / 65536.0));
      if (worst > 1.5e-7 || sqrt(sq / 65536.0) > 5e-8) ++bad;
    }
    // layout 0 as before
    for (int C : {1, 2, 4, 8}) {
      const int blocks = ss::plan_long_blocks(0, C, n);
      if (blocks != 16 * ((256 + C - 1) / C)) ++bad;
    }
    // the keys of the atomic maxima keep the order of the floats; nothing-seen is below everything, NaN above
    const float vals[] = {-__builtin_inff(), -3.0e38f, -100.0f, -1.0e-30f, -0.0f, 0.0f, 1.0e-30f, 7.5f, 3.0e38f, __builtin_inff()};
    unsigned prev = 0u;
    for (float v : vals) {
      const unsigned k = ss::max_key(v);
      if (k <= prev && !(v == 0.0f && prev == ss::max_key(-0.0f))) ++bad;
      if (!(ss::max_key_value(k) == v)) ++bad;
      prev = k;
    }
    if (ss::max_key(__builtin_nanf("")) != 0xffffffffu || ss::max_key(-__builtin_nanf("")) != 0xffffffffu || !(ss::max_key_value(0xffffffffu) != ss::max_key_value(0xffffffffu))) ++bad;
    if (!(ss::max_key_value(0u) == -__builtin_inff())) ++bad;
    printf("2^20 in two passes: bad %d\n", bad);
  }
  // ---- the fold's rows (csrc/fft65536_dif8.h: blocks of 32 Q bins, the 32 consecutive k' of every residue side by side) ----
  //   * dif_bin_offset is a bijection of a row onto itself and dif_offset_bin its inverse;
  //   * the place the transform's epilogue stores output k' of residue r (csrc/fft8192_v2.h, FRONT != 0: the per-thread offset
  //     gvoff = (voff & 0x7c) | ((voff & 0x2380) << LOGQ) out of voff = (j + 2048 h) * 4, j = 32 w + lane % 32, the constants
  //     kOutStep * kk and 16 * kOutStep for the upper half, the residue's 128 bytes in the row's base) is 4 * dif_bin_offset(Q k' + r);
  //   * a detect tile's own 256 bins are ONE block (Q = 8: 1 KB) or half of one (Q = 16), and a wave of the tile's first pass — thread
  //     g * (256 / Q) + j takes bin Q j + g (detect_fused.h: tile_column_of) — reads 64 consecutive floats (Q = 8) / four runs of 16.
  for (int logq : {3, 4}) {
    const int Q = 1 << logq, n = 8192 << logq;
    std::vector<int> seen((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
      const int o = ss::dif_bin_offset(i, logq);
      if (o < 0 || o >= n || seen[(size_t)o]++ || ss::dif_offset_bin(o, logq) != i) ++bad;
    }
    const int kOutStep = (256 << logq) * 4;
    for (int r = 0; r < Q; ++r)
      for (int t = 0; t < 512; ++t) {
        const int w = t >> 6, lam = t & 31, h = (t >> 5) & 1;
        const int j = 32 * w + lam, voff = (j + 2048 * h) * 4;
        const int gvoff = (voff & 0x7c) | ((voff & 0x2380) << logq);
        for (int kk = 0; kk < 8; ++kk)
          for (int s2 = 0; s2 < 2; ++s2) {
            const int kprime = j + 2048 * h + 256 * kk + 4096 * s2;
            const long long bytes = (long long)r * 128 + gvoff + (long long)kOutStep * kk + (s2 ? 16ll * kOutStep : 0ll);
            if (bytes != 4ll * ss::dif_bin_offset(Q * kprime + r, logq)) ++bad;
          }
      }
    const int run = 256 >> logq;
    for (int col = 0; col < n / 256; col += 37)
      for (int wave = 0; wave < 4; ++wave) {
        int lo = 1 << 30, hi = -1;
        for (int lane = 0; lane < 64; ++lane) {
          const int tid = 64 * wave + lane, bin = 256 * col + ((tid & (run - 1)) << logq) + tid / run;
          const int o = ss::dif_bin_offset(bin, logq);
          lo = o < lo ? o : lo;
          hi = o > hi ? o : hi;
          if (o / (32 * Q) != (256 * col) / (32 * Q)) ++bad;  // inside the tile's block
        }
        if (logq == 3 && hi - lo != 63) ++bad;               // 256 contiguous bytes
        if (logq == 4 && hi - lo != 3 * 32 + 15) ++bad;      // four runs of 16 floats, 32 apart
      }
    printf("the fold's rows, Q = %d: bad %d\n", Q, bad);
  }
  printf("bad %d\n", bad);
  return bad ? 1 : 0;
}
