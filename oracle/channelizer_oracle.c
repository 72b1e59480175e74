/*
 * channelizer_oracle.c — see channelizer_oracle.h. TEST INFRASTRUCTURE ONLY.
 * Built by oracle/Makefile with -O2 -ffp-contract=off so every fp32 operation rounds once, as written.
 */
#include "channelizer_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * getResamplersFactors and helpers — the reference's own code, sources/utils/radio_utils.cpp
 * ------------------------------------------------------------------------------------------------ */

/* getPrimeFactors(n).size() — radio_utils.cpp:105-126 */
static int prime_factor_count(int n) {
  if (n == 1) return 1;
  int count = 0;
  while (n % 2 == 0) {
    ++count;
    n /= 2;
  }
  for (int i = 3; i <= sqrt((double)n); i += 2) {
    while (n % i == 0) {
      ++count;
      n /= i;
    }
  }
  if (n > 2) ++count;
  return count;
}

/* split(value, factors, threshold) — radio_utils.cpp:9-34: halve around sqrt until every factor is <= threshold or prime */
static void split_factors(int value, int* factors, int* nf, int cap, int threshold) {
  if (threshold < value && prime_factor_count(value) != 1) {
    int f1 = 1, f2 = value;
    for (int i = (int)sqrt((double)value); 1 <= i; --i) {
      if (value % i == 0) {
        f1 = i;
        f2 = value / i;
        break;
      }
    }
    if (threshold < f1) split_factors(f1, factors, nf, cap, threshold);
    else if (*nf < cap) factors[(*nf)++] = f1;
    if (threshold < f2) split_factors(f2, factors, nf, cap, threshold);
    else if (*nf < cap) factors[(*nf)++] = f2;
  } else if (*nf < cap) {
    factors[(*nf)++] = value;
  }
}

static int gcd_i(int a, int b) {
  while (b) {
    const int t = a % b;
    a = b;
    b = t;
  }
  return a < 0 ? -a : a;
}

static int cmp_int(const void* a, const void* b) { return *(const int*)a - *(const int*)b; }

/* radio_utils.cpp:128-152 */
int cho_resampler_factors(int32_t sample_rate, int32_t bandwidth, int threshold, int* interp, int* decim, int cap) {
  const int g = gcd_i(sample_rate, bandwidth);
  const int left = bandwidth / g, right = sample_rate / g;
  int lf[64], rf[64], nl = 0, nr = 0;
  split_factors(left, lf, &nl, 64, threshold);
  split_factors(right, rf, &nr, 64, threshold);
  while (nl < nr) lf[nl++] = 1;
  while (nr < nl) rf[nr++] = 1;
  qsort(lf, (size_t)nl, sizeof(int), cmp_int);
  qsort(rf, (size_t)nr, sizeof(int), cmp_int);
  for (int i = 0; i < nl && i < cap; ++i) {
    interp[i] = lf[i];
    decim[i] = rf[i];
  }
  return nl;
}

/* ------------------------------------------------------------------------------------------------
 * GNU Radio 3.10 restated: gr::filter::firdes::low_pass + fft::window::kaiser + design_resampler_filter
 * (gr-filter/lib/firdes.cc, gr-fft/lib/window.cc, gr-filter/lib/rational_resampler_impl.cc) — [EXT], unpinned
 * ------------------------------------------------------------------------------------------------ */

static double izero(double x) { /* window.cc: Izero, IzeroEPSILON 1e-21 */
  double sum = 1, u = 1, n = 1;
  const double halfx = x / 2.0;
  double temp;
  do {
    temp = halfx / n;
    n += 1;
    temp *= temp;
    u *= temp;
    sum += u;
  } while (u >= 1e-21 * sum);
  return sum;
}

static void kaiser_window(int ntaps, double beta, float* w) { /* window::kaiser */
  const double ibeta = 1.0 / izero(beta);
  const double inm1 = 1.0 / ((double)(ntaps - 1));
  for (int i = 0; i < ntaps; ++i) {
    const double temp = 2 * i * inm1 - 1;
    w[i] = (float)(izero(beta * sqrt(1.0 - temp * temp)) * ibeta);
  }
}

static int low_pass_kaiser(double gain, double fs, double cutoff, double tw, double beta, float* taps, int cap) {
  const double a = beta / 0.1102 + 8.7; /* window::max_attenuation(WIN_KAISER, beta) */
  int ntaps = (int)(a * fs / (22.0 * tw)); /* firdes::compute_ntaps */
  if ((ntaps & 1) == 0) ntaps++;
  if (!taps) return ntaps;
  if (ntaps > cap) return -ntaps;
  float* w = (float*)malloc(sizeof(float) * (size_t)ntaps);
  kaiser_window(ntaps, beta, w);
  const int M = (ntaps - 1) / 2;
  const double fwT0 = 2 * M_PI * cutoff / fs;
  for (int n = -M; n <= M; ++n) {
    if (n == 0) taps[n + M] = (float)(fwT0 / M_PI * w[n + M]);
    else taps[n + M] = (float)(sin(n * fwT0) / (n * M_PI) * w[n + M]);
  }
  double fmax = taps[0 + M];
  for (int n = 1; n <= M; ++n) fmax += 2 * taps[n + M];
  gain /= fmax;
  for (int i = 0; i < ntaps; ++i) taps[i] = (float)(taps[i] * gain);
  free(w);
  return ntaps;
}

int cho_design_taps(int interp, int decim, float* taps, int cap) {
  /* rational_resampler::make: no taps given -> reduce by the gcd, fractional_bw 0 -> 0.4 */
  const int d = gcd_i(interp, decim);
  interp /= d;
  decim /= d;
  const float fractional_bw = 0.4f;
  /* design_resampler_filter: the locals are floats in GNU Radio */
  const float beta = 7.0f, halfband = 0.5f;
  const float rate = (float)interp / (float)decim;
  float trans_width, mid_transition_band;
  if (rate >= 1.0f) {
    trans_width = halfband - fractional_bw;
    mid_transition_band = (float)(halfband - trans_width / 2.0);
  } else {
    trans_width = rate * (halfband - fractional_bw);
    mid_transition_band = (float)(rate * halfband - trans_width / 2.0);
  }
  return low_pass_kaiser((double)interp, (double)interp, (double)mid_transition_band, (double)trans_width, (double)beta, taps, cap);
}

/* ------------------------------------------------------------------------------------------------
 * the chain of one recording slot
 * ------------------------------------------------------------------------------------------------ */

typedef struct {
  int interp, decim, ntaps, nt; /* nt = taps per polyphase arm = history */
  float* arm;                   /* [interp][nt]: arm[i][j] = taps[i + j*interp] (rational_resampler_impl::install_taps) */
  float* hist;                  /* the newest nt-1 input samples (re,im), oldest first */
  int ctr;                      /* d_ctr */
  int skip;                     /* input samples still to pass before the next output's window ends (carried across calls) */
} cho_stage;

struct cho_chain {
  int32_t sample_rate, bandwidth;
  int nstages;
  cho_stage st[CHO_MAX_STAGES];
  float phase_re, phase_im; /* rotator d_phase */
  float inc_re, inc_im;     /* rotator d_phase_incr */
};

cho_chain* cho_create(int32_t sample_rate, int32_t bandwidth, int threshold) {
  cho_chain* c = (cho_chain*)calloc(1, sizeof(cho_chain));
  c->sample_rate = sample_rate;
  c->bandwidth = bandwidth;
  int in[CHO_MAX_STAGES], de[CHO_MAX_STAGES];
  c->nstages = cho_resampler_factors(sample_rate, bandwidth, threshold, in, de, CHO_MAX_STAGES); /* recorder.cpp:29 */
  if (c->nstages > CHO_MAX_STAGES) {
    free(c);
    return NULL;
  }
  for (int s = 0; s < c->nstages; ++s) {
    cho_stage* st = &c->st[s];
    const int d = gcd_i(in[s], de[s]);
    st->interp = in[s] / d;
    st->decim = de[s] / d;
    const int n = cho_design_taps(in[s], de[s], NULL, 0);
    float* taps = (float*)calloc((size_t)n + (size_t)st->interp, sizeof(float));
    cho_design_taps(in[s], de[s], taps, n);
    st->ntaps = n;
    st->nt = (n + st->interp - 1) / st->interp; /* set_taps pads with zeros to a multiple of interpolation */
    st->arm = (float*)calloc((size_t)st->interp * (size_t)st->nt, sizeof(float));
    for (int k = 0; k < st->nt * st->interp; ++k) st->arm[(size_t)(k % st->interp) * st->nt + k / st->interp] = k < n ? taps[k] : 0.0f;
    st->hist = (float*)calloc((size_t)(st->nt > 1 ? st->nt - 1 : 1) * 2, sizeof(float)); /* the scheduler's zero history */
    st->ctr = 0;
    free(taps);
  }
  c->phase_re = 1.0f; /* rotator(): d_phase(1), d_phase_incr(exp(0)) */
  c->phase_im = 0.0f;
  c->inc_re = 1.0f;
  c->inc_im = 0.0f;
  return c;
}

void cho_destroy(cho_chain* c) {
  if (!c) return;
  for (int s = 0; s < c->nstages; ++s) {
    free(c->st[s].arm);
    free(c->st[s].hist);
  }
  free(c);
}

int cho_stage_count(const cho_chain* c) { return c->nstages; }

void cho_stage_info(const cho_chain* c, int stage, int* interp, int* decim, int* ntaps) {
  if (interp) *interp = c->st[stage].interp;
  if (decim) *decim = c->st[stage].decim;
  if (ntaps) *ntaps = c->st[stage].ntaps;
}

void cho_set_shift(cho_chain* c, int32_t shift_hz) {
  /* recorder.cpp:64: set_phase_inc(2.0l * M_PIl * (static_cast<double>(-shift) / static_cast<float>(m_sampleRate))) -> double arg;
   * rotator_cc_impl::set_phase_inc: d_r.set_phase_incr(exp(gr_complex(0, phase_inc))) -> float angle, cosf/sinf;
   * rotator::set_phase_incr: incr / abs(incr) */
  const double ratio = (double)(-shift_hz) / (float)c->sample_rate;
  const double phase_inc = (double)(2.0L * 3.141592653589793238462643383279502884L * (long double)ratio);
  const float ang = (float)phase_inc;
  const float re = cosf(ang), im = sinf(ang);
  const float mag = hypotf(re, im);
  c->inc_re = re / mag;
  c->inc_im = im / mag;
}

/* volk_32fc_s32fc_x2_rotator_32fc_generic: ROTATOR_RELOAD 512, and "normalize phase on every call" */
static void rotate_n(cho_chain* c, const float* in, float* out, int n) {
  float pr = c->phase_re, pi = c->phase_im;
  const float ir = c->inc_re, ii = c->inc_im;
  int done = 0;
  for (int blk = 0; blk < n / 512; ++blk) {
    for (int j = 0; j < 512; ++j, ++done) {
      const float xr = in[2 * done], xi = in[2 * done + 1];
      out[2 * done] = xr * pr - xi * pi;
      out[2 * done + 1] = xr * pi + xi * pr;
      const float nr = pr * ir - pi * ii, ni = pr * ii + pi * ir;
      pr = nr;
      pi = ni;
    }
    const float mag = hypotf(pr, pi);
    pr /= mag;
    pi /= mag;
  }
  const int rest = n % 512;
  for (int j = 0; j < rest; ++j, ++done) {
    const float xr = in[2 * done], xi = in[2 * done + 1];
    out[2 * done] = xr * pr - xi * pi;
    out[2 * done + 1] = xr * pi + xi * pr;
    const float nr = pr * ir - pi * ii, ni = pr * ii + pi * ir;
    pr = nr;
    pi = ni;
  }
  if (rest) {
    const float mag = hypotf(pr, pi);
    pr /= mag;
    pi /= mag;
  }
  c->phase_re = pr;
  c->phase_im = pi;
}

/* rational_resampler_impl::general_work as a stream: out = firs[ctr].filter(window ending at the current sample);
 * ctr += decimation; while (ctr >= interpolation) { ctr -= interpolation; advance one input sample }.
 * in: n new samples; returns the number of outputs (written to out, capacity cap samples). */
static int stage_run(cho_stage* st, const float* in, int n, float* out, int cap) {
  const int h = st->nt - 1;
  float* buf = (float*)malloc(sizeof(float) * 2 * (size_t)(h + n + 1));
  memcpy(buf, st->hist, sizeof(float) * 2 * (size_t)h);
  memcpy(buf + 2 * (size_t)h, in, sizeof(float) * 2 * (size_t)n);
  int p = st->skip, produced = 0, ctr = st->ctr; /* p: index, within the new samples, of the newest sample of the window */
  while (p < n && produced < cap) {
    const float* arm = st->arm + (size_t)ctr * st->nt;
    const float* newest = buf + 2 * (size_t)(h + p);
    float sr = 0.0f, si = 0.0f;
    /* fir_filter::filter: sum over k of taps_reversed[k] * in[k]  ==  sum over j of arm[j] * x[p - j] (taps are real: imag 0) */
    for (int j = st->nt - 1; j >= 0; --j) {
      sr += newest[-2 * j] * arm[j];
      si += newest[-2 * j + 1] * arm[j];
    }
    out[2 * produced] = sr;
    out[2 * produced + 1] = si;
    ++produced;
    ctr += st->decim;
    while (ctr >= st->interp) {
      ctr -= st->interp;
      ++p;
    }
  }
  st->ctr = ctr;
  st->skip = p - n;
  /* keep the newest h samples */
  if (n >= h) memcpy(st->hist, buf + 2 * (size_t)n, sizeof(float) * 2 * (size_t)h);
  else {
    memmove(st->hist, st->hist + 2 * (size_t)n, sizeof(float) * 2 * (size_t)(h - n));
    memcpy(st->hist + 2 * (size_t)(h - n), in, sizeof(float) * 2 * (size_t)n);
  }
  free(buf);
  return produced;
}

int cho_process(cho_chain* c, const float* iq, int n, float* out_cf32, int8_t* out_i8, int cap) {
  float* cur = (float*)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
  rotate_n(c, iq, cur, n);
  int count = n;
  for (int s = 0; s < c->nstages; ++s) {
    cho_stage* st = &c->st[s];
    const int outcap = (int)(((long long)count * st->interp) / st->decim) + 2;
    float* nxt = (float*)malloc(sizeof(float) * 2 * (size_t)outcap);
    const int produced = stage_run(st, cur, count, nxt, outcap);
    free(cur);
    cur = nxt;
    count = produced;
  }
  const int nout = count < cap ? count : cap;
  if (out_cf32) memcpy(out_cf32, cur, sizeof(float) * 2 * (size_t)nout);
  if (out_i8) {
    /* complex_to_interleaved_char(vector, 127.0) -> volk_32f_s32f_convert_8i: r = in * scalar; saturate; rintf */
    for (int i = 0; i < 2 * nout; ++i) {
      const float r = cur[i] * 127.0f;
      out_i8[i] = r > 127.0f ? (int8_t)127 : (r < -128.0f ? (int8_t)-128 : (int8_t)rintf(r));
    }
  }
  free(cur);
  return nout;
}

/* data_controller.cpp:27-42 */
int cho_transmission_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* iq_i8, int nsamples, uint8_t* out, int cap) {
  const int header = (int)(sizeof(uint64_t) + 2 * sizeof(int32_t) + sizeof(uint32_t));
  const int total = header + 2 * nsamples;
  if (!out) return total;
  if (total > cap) return -total;
  const int32_t start = frequency - sample_rate / 2, stop = frequency + sample_rate / 2;
  const uint32_t rate = (uint32_t)sample_rate;
  int off = 0;
  memcpy(out + off, &time_ms, sizeof(time_ms));
  off += (int)sizeof(time_ms);
  memcpy(out + off, &start, sizeof(start));
  off += (int)sizeof(start);
  memcpy(out + off, &stop, sizeof(stop));
  off += (int)sizeof(stop);
  memcpy(out + off, &rate, sizeof(rate));
  off += (int)sizeof(rate);
  for (int i = 0; i < 2 * nsamples; ++i) out[off + i] = (uint8_t)iq_i8[i] ^ 0x80u;
  return total;
}
