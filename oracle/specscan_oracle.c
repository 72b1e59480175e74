/*
 * specscan_oracle.c — CPU oracle for the spectral-scan hot path. TEST INFRASTRUCTURE ONLY
 * (see specscan_oracle.h: who may load it, and what pins it).
 *
 * Plain C99, single thread, no SIMD intrinsics; build with -O2 -ffp-contract=off and WITHOUT
 * -ffast-math so every float operation below is exactly one IEEE rounding, in source order, like the
 * reference built by its CMakeLists.txt (no -march, no -ffast-math: CMakeLists.txt:8).
 *
 * Citations are relative to /root/reference (shajen/rtl-sdr-scanner-cpp @ 2025-10-31).
 */
#define _GNU_SOURCE
#include "specscan_oracle.h"

#include <dlfcn.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------------
 * small helpers restated from sources/utils
 * ---------------------------------------------------------------------------------------------- */

/* getFft — sources/utils/radio_utils.cpp:98-104: smallest power of two with fs/N <= maxStep.
 * Known answers: tests/test_radio_utils.cpp:4-16. */
int orc_get_fft(int32_t sample_rate, int32_t max_step) {
  uint32_t new_fft = 1;
  while ((double)max_step < (double)sample_rate / new_fft) {
    new_fft = new_fft << 1;
  }
  return (int)new_fft;
}

/* getTunedFrequency — sources/utils/radio_utils.cpp:86-96 (round to nearest multiple of step, ties up).
 * Known answers: tests/test_radio_utils.cpp:71-103. */
int32_t orc_get_tuned_frequency(int32_t frequency, int32_t step) {
  const int32_t rest = frequency < 0 ? frequency % step + step : frequency % step;
  const int32_t down = frequency - rest;
  const int32_t up = down + step;
  if (rest < step - rest) {
    return down;
  } else {
    return up;
  }
}

/* indexToShift lambda — sources/radio/sdr_device.cpp:150,154: step is a double, the product is
 * truncated to int32, fs/2 is an integer division. */
int32_t orc_index_to_shift(int32_t sample_rate, int fft_size, int index) {
  const double step = (double)sample_rate / fft_size;
  return (int32_t)(step * (index + 0.5)) - sample_rate / 2;
}

/* average — sources/utils/utils.cpp:31-53. One running float `sum` walked from i = -a; at every i
 * FIRST the element that leaves the window (i-a-1) is subtracted, THEN the one that enters (i+a) is
 * added, then sum/count is stored. Known answer: tests/test_utils.cpp:4-13. */
void orc_average(const float* input, float* output, int size, int group_size) {
  const int a = group_size / 2;
  float sum = 0.0f;
  int count = 0;
  for (int i = -a; i < size + a - 1; ++i) {
    const int first = i - a - 1;
    const int last = i + a;
    if (0 <= first && first < size) {
      sum -= input[first];
      count--;
    }
    if (0 <= last && last < size) {
      sum += input[last];
      count++;
    }
    if (0 <= i && i < size) {
      output[i] = sum / count; /* float / int -> float division */
    }
  }
}

/* getMaxIndex — sources/utils/collection_utils.h:8-14: first maximum of [index-g/2, index+g/2+1)
 * clipped to [0,size). Known answers: tests/test_collection_utils.cpp:81-106. */
int orc_get_max_index(const float* data, int size, int index, int group_size) {
  int lo = index - group_size / 2;
  int hi = index + group_size / 2 + 1;
  if (lo < 0) lo = 0;
  if (hi > size) hi = size;
  int best = lo;
  for (int i = lo + 1; i < hi; ++i) {
    if (data[best] < data[i]) best = i; /* std::max_element keeps the first of equal maxima */
  }
  return best;
}

/* containsWithMargin — sources/utils/collection_utils.h:16-27 on the sorted key set of the std::map.
 * Known answers: tests/test_collection_utils.cpp:4-46. */
int orc_contains_with_margin(const int* keys, int nkeys, int index, int margin, int* found) {
  const int submargin = margin % 2 == 0 ? margin / 2 : margin / 2 + 1;
  const int left = index - submargin;
  const int right = index + submargin;
  for (int k = 0; k < nkeys; ++k) { /* lower_bound(left) */
    if (keys[k] >= left) {
      if (keys[k] <= right) {
        if (found) *found = keys[k];
        return 1;
      }
      return 0;
    }
  }
  return 0;
}

/* mostFrequentValue — sources/utils/collection_utils.h:29-50: among the values with the highest count,
 * sorted ascending, the one at position size/2. Known answers: tests/test_collection_utils.cpp:48-63.
 * The reference is undefined for an empty vector (transmission.cpp:151); the oracle returns -1. */
static int cmp_int(const void* a, const void* b) {
  const int x = *(const int*)a, y = *(const int*)b;
  return (x > y) - (x < y);
}
int orc_most_frequent_value(const int* data, int n) {
  if (n <= 0) return -1;
  int* s = (int*)malloc(sizeof(int) * (size_t)n);
  memcpy(s, data, sizeof(int) * (size_t)n);
  qsort(s, (size_t)n, sizeof(int), cmp_int);
  int best_count = 0;
  for (int i = 0; i < n;) {
    int j = i;
    while (j < n && s[j] == s[i]) ++j;
    if (j - i > best_count) best_count = j - i;
    i = j;
  }
  int* tied = (int*)malloc(sizeof(int) * (size_t)n);
  int nt = 0;
  for (int i = 0; i < n;) {
    int j = i;
    while (j < n && s[j] == s[i]) ++j;
    if (j - i == best_count) tied[nt++] = s[i];
    i = j;
  }
  const int r = tied[nt / 2];
  free(tied);
  free(s);
  return r;
}

/* ------------------------------------------------------------------------------------------------
 * front end: the three GNU Radio pieces at sources/radio/sdr_device.cpp:164
 *   gr::fft::fft_v<gr_complex, true>::make(fftSize, gr::fft::window::hamming(fftSize), true)
 * GNU Radio / VOLK / FFTW are un-vendored apt packages (Dockerfile:4), version unpinned
 * (GNU Radio 3.10.x on ubuntu:24.04). PARITY UNPINNED by any reference test; restated from the
 * published definitions:
 *   window::hamming(ntaps):  taps[n] = 0.54 - 0.46*cos(2*pi*n/(ntaps-1)), evaluated in double, stored
 *                            as float (gr-fft/lib/window.cc)
 *   fft_v work():            dst = in * window (volk_32fc_32f_multiply_32fc: one rounding per part),
 *                            unnormalised forward c2c fftwf, then with shift=true
 *                            out[0..N-len) = X[len..N), out[N-len..N) = X[0..len), len = ceil(N/2)
 *                            (gr-fft/lib/fft_v_fftw.cc)
 * ---------------------------------------------------------------------------------------------- */
void orc_hamming(int n, float* taps) {
  const float M = (float)(n - 1);
  for (int i = 0; i < n; ++i) {
    taps[i] = (float)(0.54 - 0.46 * cos((2.0 * M_PI * i) / M));
  }
}

/* --- FFT back ends --- */
static int g_fft_backend = 0;

typedef struct {
  int n;
  float* tw32;  /* n/2 twiddles re,im as float (rounded from double) */
  double* tw64; /* n/2 twiddles re,im as double */
  float* tmp32;
  double* tmp64;
} fft_tables;
static fft_tables g_tab = {0, NULL, NULL, NULL, NULL};

static void fft_prepare(int n) {
  if (g_tab.n == n) return;
  free(g_tab.tw32);
  free(g_tab.tw64);
  free(g_tab.tmp32);
  free(g_tab.tmp64);
  g_tab.n = n;
  g_tab.tw32 = (float*)malloc(sizeof(float) * (size_t)n);
  g_tab.tw64 = (double*)malloc(sizeof(double) * (size_t)n);
  g_tab.tmp32 = (float*)malloc(sizeof(float) * 2 * (size_t)n);
  g_tab.tmp64 = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  for (int k = 0; k < n / 2; ++k) {
    const double ang = -2.0 * M_PI * (double)k / (double)n;
    g_tab.tw64[2 * k] = cos(ang);
    g_tab.tw64[2 * k + 1] = sin(ang);
    g_tab.tw32[2 * k] = (float)g_tab.tw64[2 * k];
    g_tab.tw32[2 * k + 1] = (float)g_tab.tw64[2 * k + 1];
  }
}

static unsigned bitrev(unsigned x, int bits) {
  unsigned r = 0;
  for (int i = 0; i < bits; ++i) {
    r = (r << 1) | (x & 1u);
    x >>= 1;
  }
  return r;
}

/* textbook in-order radix-2 decimation-in-time, fp32 arithmetic, twiddles rounded from fp64 */
static void fft_builtin_f32(int n, const float* in, float* out) {
  fft_prepare(n);
  int bits = 0;
  while ((1 << bits) < n) ++bits;
  float* x = g_tab.tmp32;
  for (int i = 0; i < n; ++i) {
    const unsigned j = bitrev((unsigned)i, bits);
    x[2 * j] = in[2 * i];
    x[2 * j + 1] = in[2 * i + 1];
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1;
    const int tstep = n / len;
    for (int base = 0; base < n; base += len) {
      for (int k = 0; k < half; ++k) {
        const float wr = g_tab.tw32[2 * (k * tstep)];
        const float wi = g_tab.tw32[2 * (k * tstep) + 1];
        float* a = &x[2 * (base + k)];
        float* b = &x[2 * (base + k + half)];
        const float tr = b[0] * wr - b[1] * wi;
        const float ti = b[0] * wi + b[1] * wr;
        b[0] = a[0] - tr;
        b[1] = a[1] - ti;
        a[0] = a[0] + tr;
        a[1] = a[1] + ti;
      }
    }
  }
  memcpy(out, x, sizeof(float) * 2 * (size_t)n);
}

static void fft_builtin_f64(int n, const float* in, float* out) {
  fft_prepare(n);
  int bits = 0;
  while ((1 << bits) < n) ++bits;
  double* x = g_tab.tmp64;
  for (int i = 0; i < n; ++i) {
    const unsigned j = bitrev((unsigned)i, bits);
    x[2 * j] = in[2 * i];
    x[2 * j + 1] = in[2 * i + 1];
  }
  for (int len = 2; len <= n; len <<= 1) {
    const int half = len >> 1;
    const int tstep = n / len;
    for (int base = 0; base < n; base += len) {
      for (int k = 0; k < half; ++k) {
        const double wr = g_tab.tw64[2 * (k * tstep)];
        const double wi = g_tab.tw64[2 * (k * tstep) + 1];
        double* a = &x[2 * (base + k)];
        double* b = &x[2 * (base + k + half)];
        const double tr = b[0] * wr - b[1] * wi;
        const double ti = b[0] * wi + b[1] * wr;
        b[0] = a[0] - tr;
        b[1] = a[1] - ti;
        a[0] = a[0] + tr;
        a[1] = a[1] + ti;
      }
    }
  }
  for (int i = 0; i < 2 * n; ++i) out[i] = (float)x[i];
}

/* MKL's FFTW3 single-precision interface — the same API (fftwf_plan_dft_1d / fftwf_execute) that
 * gr::fft::fft_complex_fwd drives in the reference's dependency. Prototypes declared by hand: there
 * is no fftw3.h in this image. */
typedef void* (*fn_malloc)(size_t);
typedef void (*fn_free)(void*);
typedef void* (*fn_plan)(int, void*, void*, int, unsigned);
typedef void (*fn_exec)(void*);
typedef void (*fn_destroy)(void*);
static struct {
  void* lib;
  fn_malloc f_malloc;
  fn_free f_free;
  fn_plan f_plan;
  fn_exec f_exec;
  fn_destroy f_destroy;
  int n;
  void* plan;
  float *inbuf, *outbuf;
} g_mkl = {0};

static int mkl_load(void) {
  if (g_mkl.lib) return 0;
  const char* names[] = {"libmkl_rt.so", "libmkl_rt.so.1", "libmkl_rt.so.2", "/opt/conda/lib/libmkl_rt.so",
                         "/opt/conda/lib/libmkl_rt.so.1", "libfftw3f.so.3", NULL};
  const char* env = getenv("ORC_FFTW_LIB");
  void* lib = NULL;
  if (env) lib = dlopen(env, RTLD_NOW | RTLD_LOCAL);
  for (int i = 0; !lib && names[i]; ++i) lib = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
  if (!lib) return -1;
  g_mkl.f_malloc = (fn_malloc)dlsym(lib, "fftwf_malloc");
  g_mkl.f_free = (fn_free)dlsym(lib, "fftwf_free");
  g_mkl.f_plan = (fn_plan)dlsym(lib, "fftwf_plan_dft_1d");
  g_mkl.f_exec = (fn_exec)dlsym(lib, "fftwf_execute");
  g_mkl.f_destroy = (fn_destroy)dlsym(lib, "fftwf_destroy_plan");
  if (!g_mkl.f_malloc || !g_mkl.f_free || !g_mkl.f_plan || !g_mkl.f_exec || !g_mkl.f_destroy) {
    dlclose(lib);
    return -1;
  }
  g_mkl.lib = lib;
  return 0;
}

static void fft_fftw(int n, const float* in, float* out) {
  if (g_mkl.n != n) {
    if (g_mkl.plan) {
      g_mkl.f_destroy(g_mkl.plan);
      g_mkl.f_free(g_mkl.inbuf);
      g_mkl.f_free(g_mkl.outbuf);
    }
    g_mkl.inbuf = (float*)g_mkl.f_malloc(sizeof(float) * 2 * (size_t)n);
    g_mkl.outbuf = (float*)g_mkl.f_malloc(sizeof(float) * 2 * (size_t)n);
    /* FFTW_FORWARD = -1, FFTW_MEASURE = 0 (what gr::fft uses); MKL ignores the planner flag */
    g_mkl.plan = g_mkl.f_plan(n, g_mkl.inbuf, g_mkl.outbuf, -1, 0u);
    g_mkl.n = n;
  }
  memcpy(g_mkl.inbuf, in, sizeof(float) * 2 * (size_t)n);
  g_mkl.f_exec(g_mkl.plan);
  memcpy(out, g_mkl.outbuf, sizeof(float) * 2 * (size_t)n);
}

int orc_set_fft_backend(int which) {
  if (which == 2) {
    if (mkl_load() != 0) return -1;
  } else if (which != 0 && which != 1) {
    return -1;
  }
  g_fft_backend = which;
  return 0;
}

void orc_fft_forward(int n, const float* in, float* out) {
  if (g_fft_backend == 2) {
    fft_fftw(n, in, out);
  } else if (g_fft_backend == 1) {
    fft_builtin_f64(n, in, out);
  } else {
    fft_builtin_f32(n, in, out);
  }
}

void orc_fft_v(int n, const float* window, const float* in, float* out) {
  float* buf = (float*)malloc(sizeof(float) * 4 * (size_t)n);
  float* spec = buf + 2 * (size_t)n;
  for (int i = 0; i < n; ++i) { /* volk_32fc_32f_multiply_32fc */
    buf[2 * i] = in[2 * i] * window[i];
    buf[2 * i + 1] = in[2 * i + 1] * window[i];
  }
  orc_fft_forward(n, buf, spec);
  const int len = (n + 1) / 2; /* ceil(n / 2.0) */
  memcpy(&out[0], &spec[2 * len], sizeof(float) * 2 * (size_t)(n - len));
  memcpy(&out[2 * (n - len)], &spec[0], sizeof(float) * 2 * (size_t)len);
  free(buf);
}

/* PSD::work — sources/radio/blocks/psd.cpp:18-20:
 *   out[i] = 10.0f * std::log10(std::pow(std::abs(in[i]), 2.0f) / m_sampleRate)
 * std::abs(complex<float>) is cabsf (= hypotf), pow(x, 2.0f) is lowered by g++ -O2 to x*x, the divisor
 * is the int32 sample rate converted to float, log10 is libm's log10f. */
void orc_psd(const float* x, float* out_db, int n, int32_t sample_rate) {
  const float fs = (float)sample_rate;
  for (int i = 0; i < n; ++i) {
    const float mag = hypotf(x[2 * i], x[2 * i + 1]);
    out_db[i] = 10.0f * log10f((mag * mag) / fs);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Averager — sources/radio/averager.cpp. Known answers: tests/test_averager.cpp:46-140.
 * ---------------------------------------------------------------------------------------------- */
struct orc_averager {
  int size, group;
  float* sum;     /* m_sum */
  float* average; /* m_average */
  float* rows;    /* m_buffers as a ring: group rows of size floats */
  int head;       /* index of the OLDEST row (deque front) */
  int frames;     /* m_frames */
};

static void averager_update(orc_averager* a) { /* updateAverage, averager.cpp:52-61 */
  if (a->group <= a->frames) {
    for (int i = 0; i < a->size; ++i) a->average[i] = a->sum[i] / a->group; /* float / int */
  } else {
    for (int i = 0; i < a->size; ++i) a->average[i] = SS_NO_DATA; /* setNoData, radio_utils.cpp:72-76 */
  }
}

orc_averager* orc_averager_create(int size, int group_size) { /* ctor, averager.cpp:7-12 */
  orc_averager* a = (orc_averager*)calloc(1, sizeof(*a));
  a->size = size;
  a->group = group_size;
  a->sum = (float*)calloc((size_t)size, sizeof(float));
  a->average = (float*)calloc((size_t)size, sizeof(float));
  a->rows = (float*)calloc((size_t)size * (size_t)group_size, sizeof(float));
  a->head = 0;
  a->frames = 0;
  averager_update(a);
  return a;
}

void orc_averager_destroy(orc_averager* a) {
  if (!a) return;
  free(a->sum);
  free(a->average);
  free(a->rows);
  free(a);
}

void orc_averager_push(orc_averager* a, const float* data) { /* push, averager.cpp:14-25 */
  a->frames = a->frames + 1 < a->group ? a->frames + 1 : a->group;
  float* buffer = &a->rows[(size_t)a->head * (size_t)a->size]; /* m_buffers.front() */
  for (int i = 0; i < a->size; ++i) a->sum[i] -= buffer[i];    /* subtract, :46-50 */
  memcpy(buffer, data, sizeof(float) * (size_t)a->size);
  for (int i = 0; i < a->size; ++i) a->sum[i] += buffer[i];    /* add, :40-44 */
  a->head = (a->head + 1) % a->group;                          /* pop_front + push_back */
  averager_update(a);
}

void orc_averager_reset(orc_averager* a) { /* reset, averager.cpp:27-34 */
  memset(a->sum, 0, sizeof(float) * (size_t)a->size);
  memset(a->rows, 0, sizeof(float) * (size_t)a->size * (size_t)a->group);
  a->frames = 0;
  averager_update(a);
}

const float* orc_averager_average(const orc_averager* a) { return a->average; }

const float* orc_averager_row(const orc_averager* a, int row) {
  return &a->rows[(size_t)((a->head + row) % a->group) * (size_t)a->size];
}

/* ------------------------------------------------------------------------------------------------
 * the chain
 * ---------------------------------------------------------------------------------------------- */
typedef struct orc_noise { /* NoiseLearner::Noise, noise_learner.h:11-20 */
  int32_t center;
  float* thr; /* m_threshold */
  int samples;
  int ready;
  int64_t start_ms; /* m_startLearningTime */
  int have_start;
  struct orc_noise* next;
} orc_noise;

struct orc_ctx {
  ss_config cfg;
  float* window;
  int32_t* ignored;
  int32_t range_lo, range_hi;
  orc_noise* noise;
  orc_averager* avgr;
  float *frame_in, *spec, *psd_row, *rel_row, *avg_row;
  /* planes of the last batch + the ring as it was before it, for orc_read_window */
  float *last_psd, *last_rel, *last_avg, *hist;
  int last_n;
  double stage[6];
  char err[256];
};

static char g_create_err[256] = "";

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void orc_default_config(ss_config* cfg, int32_t sample_rate, int32_t center_hz) {
  memset(cfg, 0, sizeof(*cfg));
  cfg->abi_version = SS_ABI_VERSION;
  cfg->fft_size = orc_get_fft(sample_rate, 250);                      /* config.h:33, sdr_device.cpp:149 */
  cfg->sample_rate = sample_rate;
  const double step = (double)sample_rate / cfg->fft_size;            /* sdr_device.cpp:150 */
  const int d = (int)(step / 50);                                     /* sdr_device.cpp:152, config.h:32 */
  cfg->decim = d > 1 ? d : 1;
  cfg->in_format = SS_FMT_CF32;
  cfg->int_scale = 0.0f;
  cfg->window = NULL;
  cfg->grouping_x = 21;                                               /* config.h:28 */
  cfg->grouping_y = 21;                                               /* config.h:29 */
  cfg->start_level = 8.0f;                                            /* config.h:30 */
  cfg->range_lo = center_hz - sample_rate / 2;
  cfg->range_hi = center_hz + sample_rate / 2;
  cfg->learn_frames = 100;                                            /* 2000 ms * 50 fps */
  cfg->learn_ms = 2000;                                               /* config.h:24 */
  cfg->max_batch = 1024;
  cfg->device_id = 0;
}

static int is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

int orc_create(const ss_config* cfg, orc_ctx** out) {
  if (!cfg || !out) {
    snprintf(g_create_err, sizeof g_create_err, "null argument");
    return SS_ERR_INVALID;
  }
  if (cfg->abi_version != SS_ABI_VERSION || !is_pow2(cfg->fft_size) || cfg->fft_size < 2 || cfg->sample_rate <= 0 ||
      cfg->decim < 1 || cfg->grouping_x < 1 || cfg->grouping_y < 1 || cfg->max_batch < 1 || cfg->n_ignored < 0 ||
      cfg->in_format < SS_FMT_CF32 || cfg->in_format > SS_FMT_CU8 || cfg->learn_frames < 1) {
    snprintf(g_create_err, sizeof g_create_err, "invalid ss_config");
    return SS_ERR_INVALID;
  }
  orc_ctx* c = (orc_ctx*)calloc(1, sizeof(*c));
  const size_t n = (size_t)cfg->fft_size;
  c->cfg = *cfg;
  c->window = (float*)malloc(sizeof(float) * n);
  if (cfg->window) {
    memcpy(c->window, cfg->window, sizeof(float) * n);
  } else {
    orc_hamming(cfg->fft_size, c->window);
  }
  c->cfg.window = NULL;
  c->ignored = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(cfg->n_ignored + 1));
  if (cfg->n_ignored) memcpy(c->ignored, cfg->ignored, sizeof(int32_t) * 2 * (size_t)cfg->n_ignored);
  c->cfg.ignored = NULL;
  if (c->cfg.int_scale == 0.0f) c->cfg.int_scale = cfg->in_format == SS_FMT_CU8 ? 1.0f / 127.5f : 1.0f / 128.0f;
  c->range_lo = cfg->range_lo;
  c->range_hi = cfg->range_hi;
  c->avgr = orc_averager_create(cfg->fft_size, cfg->grouping_y); /* transmission.cpp:24 */
  c->frame_in = (float*)malloc(sizeof(float) * 2 * n);
  c->spec = (float*)malloc(sizeof(float) * 2 * n);
  c->psd_row = (float*)malloc(sizeof(float) * n);
  c->rel_row = (float*)malloc(sizeof(float) * n);
  c->avg_row = (float*)malloc(sizeof(float) * n);
  c->last_psd = (float*)malloc(sizeof(float) * n * (size_t)cfg->max_batch);
  c->last_rel = (float*)malloc(sizeof(float) * n * (size_t)cfg->max_batch);
  c->last_avg = (float*)malloc(sizeof(float) * n * (size_t)cfg->max_batch);
  c->hist = (float*)calloc(n * (size_t)cfg->grouping_y, sizeof(float));
  c->last_n = 0;
  *out = c;
  return SS_OK;
}

static void free_noise(orc_ctx* c) {
  orc_noise* p = c->noise;
  while (p) {
    orc_noise* nx = p->next;
    free(p->thr);
    free(p);
    p = nx;
  }
  c->noise = NULL;
}

void orc_destroy(orc_ctx* c) {
  if (!c) return;
  free_noise(c);
  orc_averager_destroy(c->avgr);
  free(c->window);
  free(c->ignored);
  free(c->frame_in);
  free(c->spec);
  free(c->psd_row);
  free(c->rel_row);
  free(c->avg_row);
  free(c->last_psd);
  free(c->last_rel);
  free(c->last_avg);
  free(c->hist);
  free(c);
}

const char* orc_last_error(const orc_ctx* c) { return c ? c->err : g_create_err; }

static int32_t center_of(const orc_ctx* c) { return (c->range_lo + c->range_hi) / 2; } /* sdr_device.cpp:146 */

/* indexToFrequency — sources/radio/sdr_device.cpp:153 */
static int32_t index_to_frequency(const orc_ctx* c, int index) {
  return center_of(c) + orc_index_to_shift(c->cfg.sample_rate, c->cfg.fft_size, index);
}

/* isIndexInRange — sdr_device.cpp:155-158; isIndexIgnored — transmission.cpp:156-164 */
static int index_passes(const orc_ctx* c, int index) {
  const int32_t f = index_to_frequency(c, index);
  if (!(c->range_lo <= f && f <= c->range_hi)) return 0;
  for (int k = 0; k < c->cfg.n_ignored; ++k) {
    if (c->ignored[2 * k] <= f && f <= c->ignored[2 * k + 1]) return 0;
  }
  return 1;
}

static orc_noise* noise_for(orc_ctx* c, int32_t center, int create) { /* m_noise[frequency], noise_learner.cpp:41-42 */
  for (orc_noise* p = c->noise; p; p = p->next) {
    if (p->center == center) return p;
  }
  if (!create) return NULL;
  orc_noise* p = (orc_noise*)calloc(1, sizeof(*p));
  p->center = center;
  p->next = c->noise;
  c->noise = p;
  return p;
}

/* Noise::add — noise_learner.cpp:11-28. `now` is the injected getTime(); with no timestamps the
 * learning ends after learn_frames frames instead of NOISE_LEARNING_TIME of wall clock. */
static int noise_add(orc_ctx* c, orc_noise* z, const float* data, int size, int has_time, int64_t now) {
  if (z->ready) return 1;
  if (!z->thr) {
    z->thr = (float*)malloc(sizeof(float) * (size_t)size);
    for (int i = 0; i < size; ++i) z->thr[i] = -FLT_MAX; /* resize(size, -numeric_limits<float>::max()) */
  }
  for (int i = 0; i < size; ++i) {
    z->thr[i] = z->thr[i] < data[i] ? data[i] : z->thr[i]; /* std::max(thr, data): returns thr unless thr < data */
  }
  z->samples++;
  const int done = has_time ? (z->start_ms + c->cfg.learn_ms <= now) : (z->samples >= c->cfg.learn_frames);
  if (done) {
    z->ready = 1;
    return 1;
  }
  return 0;
}

static void load_frame(const orc_ctx* c, const void* iq, int frame, float* dst) {
  const size_t n = (size_t)c->cfg.fft_size;
  const size_t item = n * (size_t)c->cfg.decim; /* stream_to_vector item, sdr_device.cpp:161 */
  /* Decimator::decimate keeps the first N samples of the item — decimator.h:15-22 */
  if (c->cfg.in_format == SS_FMT_CF32) {
    memcpy(dst, (const float*)iq + 2 * item * (size_t)frame, sizeof(float) * 2 * n);
  } else if (c->cfg.in_format == SS_FMT_CS8) {
    const int8_t* p = (const int8_t*)iq + 2 * item * (size_t)frame;
    for (size_t i = 0; i < 2 * n; ++i) dst[i] = (float)p[i] * c->cfg.int_scale;
  } else {
    const uint8_t* p = (const uint8_t*)iq + 2 * item * (size_t)frame;
    for (size_t i = 0; i < 2 * n; ++i) dst[i] = ((float)p[i] - 127.5f) * c->cfg.int_scale;
  }
}

int orc_process(orc_ctx* c, const void* iq, int32_t nframes, const int64_t* t_ms, float* psd_db, float* rel_db,
                float* avg_db, int32_t* cand_off, int32_t* cand_idx, float* cand_avg, int32_t cand_cap) {
  if (!c) return SS_ERR_INVALID;
  if (nframes < 0 || (nframes > 0 && !iq)) {
    snprintf(c->err, sizeof c->err, "bad iq/nframes");
    return SS_ERR_INVALID;
  }
  if (nframes > c->cfg.max_batch) {
    snprintf(c->err, sizeof c->err, "nframes %d > max_batch %d", nframes, c->cfg.max_batch);
    return SS_ERR_BATCH;
  }
  const int n = c->cfg.fft_size;
  const int gy = c->cfg.grouping_y;
  int overflow = 0;
  int32_t ncand = 0;
  if (cand_off) cand_off[0] = 0;

  /* ring rows as they are before this batch (oldest..newest), for orc_read_window */
  for (int r = 0; r < gy; ++r) memcpy(&c->hist[(size_t)r * n], orc_averager_row(c->avgr, r), sizeof(float) * (size_t)n);

  const int32_t center = center_of(c);
  orc_noise* z = NULL;
  for (int f = 0; f < nframes; ++f) {
    double t0 = now_s();
    load_frame(c, iq, f, c->frame_in);
    orc_fft_v(n, c->window, c->frame_in, c->spec);
    double t1 = now_s();
    c->stage[0] += t1 - t0;
    orc_psd(c->spec, c->psd_row, n, c->cfg.sample_rate);
    double t2 = now_s();
    c->stage[1] += t2 - t1;

    /* NoiseLearner::work — noise_learner.cpp:36-67 */
    if (!z) {
      z = noise_for(c, center, 1);
      if (!z->have_start) { /* Noise::Noise() reads getTime() when first touched, noise_learner.cpp:9 */
        z->start_ms = t_ms ? t_ms[f] : 0;
        z->have_start = 1;
      }
    }
    if (!z->ready) {
      noise_add(c, z, c->psd_row, n, t_ms != NULL, t_ms ? t_ms[f] : 0);
      for (int j = 0; j < n; ++j) c->rel_row[j] = SS_NO_DATA; /* setNoData, :49 — also on the frame that completes learning */
    } else {
      for (int j = 0; j < n; ++j) c->rel_row[j] = c->psd_row[j] - z->thr[j]; /* :55 */
    }
    double t3 = now_s();
    c->stage[2] += t3 - t2;

    /* Transmission::process — transmission.cpp:57-61 */
    orc_averager_push(c->avgr, c->rel_row);
    double t4 = now_s();
    c->stage[3] += t4 - t3;
    memset(c->avg_row, 0, sizeof(float) * (size_t)n); /* std::vector<float> avgPower(size, 0.0), transmission.cpp:60: with GROUPING_X = 1 average() never writes the last bin */
    orc_average(orc_averager_average(c->avgr), c->avg_row, n, c->cfg.grouping_x);
    double t5 = now_s();
    c->stage[4] += t5 - t4;

    /* Transmission::addSignals, candidate detection proper — transmission.cpp:90-94 */
    for (int i = 0; i < n; ++i) {
      if (c->cfg.start_level <= c->avg_row[i] && index_passes(c, i)) {
        if (ncand < cand_cap && cand_idx) {
          cand_idx[ncand] = i;
          if (cand_avg) cand_avg[ncand] = c->avg_row[i];
        } else {
          overflow = 1;
        }
        ncand++;
      }
    }
    if (cand_off) cand_off[f + 1] = ncand;
    c->stage[5] += now_s() - t5;

    if (psd_db) memcpy(&psd_db[(size_t)f * n], c->psd_row, sizeof(float) * (size_t)n);
    if (rel_db) memcpy(&rel_db[(size_t)f * n], c->rel_row, sizeof(float) * (size_t)n);
    if (avg_db) memcpy(&avg_db[(size_t)f * n], c->avg_row, sizeof(float) * (size_t)n);
    memcpy(&c->last_psd[(size_t)f * n], c->psd_row, sizeof(float) * (size_t)n);
    memcpy(&c->last_rel[(size_t)f * n], c->rel_row, sizeof(float) * (size_t)n);
    memcpy(&c->last_avg[(size_t)f * n], c->avg_row, sizeof(float) * (size_t)n);
  }
  c->last_n = nframes;
  if (overflow && cand_cap > 0) {
    snprintf(c->err, sizeof c->err, "%d candidates > cand_cap %d", ncand, cand_cap);
    return SS_ERR_CAND_OVERFLOW;
  }
  return SS_OK;
}

int orc_set_frequency_range(orc_ctx* c, int32_t lo, int32_t hi) {
  if (!c) return SS_ERR_INVALID;
  c->range_lo = lo;
  c->range_hi = hi;
  return SS_OK;
}

int orc_reset(orc_ctx* c) { /* Transmission::resetBuffers, transmission.cpp:42-55 */
  if (!c) return SS_ERR_INVALID;
  orc_averager_reset(c->avgr);
  return SS_OK;
}

int orc_reset_noise(orc_ctx* c) { /* NoiseLearner::resetBuffers, noise_learner.cpp:69-72 */
  if (!c) return SS_ERR_INVALID;
  free_noise(c);
  return SS_OK;
}

int orc_read_window(orc_ctx* c, int32_t plane, int32_t frame, int32_t lo, int32_t hi, float* out) {
  if (!c || !out) return SS_ERR_INVALID;
  const int n = c->cfg.fft_size;
  const int gy = c->cfg.grouping_y;
  if (lo < 0 || hi > n || lo > hi || frame >= c->last_n) {
    snprintf(c->err, sizeof c->err, "window out of range");
    return SS_ERR_INVALID;
  }
  const float* src = NULL;
  if (frame >= 0) {
    src = plane == SS_PLANE_PSD ? c->last_psd : plane == SS_PLANE_REL ? c->last_rel : plane == SS_PLANE_AVG ? c->last_avg : NULL;
    if (src) src += (size_t)frame * n;
  } else if (plane == SS_PLANE_REL && frame >= -(gy - 1)) {
    src = &c->hist[(size_t)(gy + frame) * n]; /* hist row gy-1 is the newest pre-batch ring row */
  }
  if (!src) {
    snprintf(c->err, sizeof c->err, "bad plane/frame");
    return SS_ERR_INVALID;
  }
  memcpy(out, src + lo, sizeof(float) * (size_t)(hi - lo));
  return SS_OK;
}

int orc_read_noise(orc_ctx* c, float* thr) {
  if (!c || !thr) return SS_ERR_INVALID;
  orc_noise* z = noise_for(c, center_of(c), 0);
  const int n = c->cfg.fft_size;
  if (!z || !z->thr) {
    for (int i = 0; i < n; ++i) thr[i] = -FLT_MAX;
    return 0;
  }
  memcpy(thr, z->thr, sizeof(float) * (size_t)n);
  return z->ready ? 1 : 0;
}

/* ------------------------------------------------------------------------------------------------
 * Spectrogram side branch — sources/radio/blocks/spectrogram.cpp. Fed with the raw PSD rows (PSD::work
 * output, before the noise subtraction: the branch is wired psd -> spectrogram at sdr_device.cpp:170-171).
 * ---------------------------------------------------------------------------------------------- */
struct orc_spectrogram {
  int in_size, out_size, factor; /* m_inputSize, m_outputSize, m_decimatorFactor (spectrogram.cpp:13-15) */
  float* sum;                    /* Container::m_sum */
  int counter;                   /* Container::m_counter */
};

orc_spectrogram* orc_spectrogram_create(int in_size, int32_t sample_rate) {
  orc_spectrogram* g = (orc_spectrogram*)calloc(1, sizeof(*g));
  const int pref = orc_get_fft(sample_rate, 1000); /* SPECTROGRAM_PREFERRED_MAX_STEP, config.h:36 */
  g->in_size = in_size;
  g->out_size = pref < 16384 ? pref : 16384; /* SPECTROGRAM_MAX_FFT, config.h:37; spectrogram.cpp:14 */
  if (g->out_size > in_size) g->out_size = in_size;
  g->factor = in_size / g->out_size;
  g->sum = (float*)calloc((size_t)g->out_size, sizeof(float));
  return g;
}
void orc_spectrogram_destroy(orc_spectrogram* g) {
  if (!g) return;
  free(g->sum);
  free(g);
}
int orc_spectrogram_size(const orc_spectrogram* g) { return g->out_size; }

/* Spectrogram::process — spectrogram.cpp:45-60 */
void orc_spectrogram_process(orc_spectrogram* g, const float* data) {
  if (g->factor == 1) {
    for (int i = 0; i < g->out_size; ++i) g->sum[i] += data[i];
  } else {
    for (int i = 0; i < g->out_size; ++i) {
      float sum = 0.0f;
      for (int j = 0; j < g->factor; ++j) sum += data[i * g->factor + j];
      g->sum[i] += sum / g->factor; /* float / int */
    }
  }
  g->counter++;
}

/* Spectrogram::send without the 1000 ms gate — spectrogram.cpp:66-72: int8 = float -> int8 conversion of sum/count */
int orc_spectrogram_send(orc_spectrogram* g, int8_t* out, float* mean_out) {
  const int count = g->counter;
  for (int j = 0; j < g->out_size; ++j) {
    const float v = g->sum[j] / g->counter;
    if (mean_out) mean_out[j] = v;
    out[j] = (int8_t)v;
  }
  memset(g->sum, 0, sizeof(float) * (size_t)g->out_size);
  g->counter = 0;
  return count;
}

/* DataController::pushSpectrogram — data_controller.cpp:44-57. `add` appends the raw bytes of each field in turn
 * (data_controller.cpp:8-20): time (uint64), start, stop, step (Frequency = int32), size (uint32), the int8 row. */
static void put(uint8_t* buf, size_t* at, const void* field, size_t bytes) {
  memcpy(buf + *at, field, bytes);
  *at += bytes;
}
int orc_spectrogram_payload(uint64_t time_ms, int32_t frequency, int32_t sample_rate, const int8_t* row, int32_t size, uint8_t* out, int32_t cap) {
  const int32_t start = frequency - sample_rate / 2; /* :45 */
  const int32_t stop = frequency + sample_rate / 2;  /* :46 */
  const int32_t step = sample_rate / size;           /* :47 */
  const uint32_t n = (uint32_t)size;
  const size_t need = sizeof(uint64_t) + 3 * sizeof(int32_t) + sizeof(uint32_t) + (size_t)size; /* :48 */
  size_t at = 0;
  if ((size_t)cap < need) return -1;
  put(out, &at, &time_ms, sizeof(time_ms));
  put(out, &at, &start, sizeof(start));
  put(out, &at, &stop, sizeof(stop));
  put(out, &at, &step, sizeof(step));
  put(out, &at, &n, sizeof(n));
  put(out, &at, row, (size_t)size);
  return (int)at;
}

void orc_stage_seconds(orc_ctx* c, double out[6]) {
  for (int i = 0; i < 6; ++i) {
    out[i] = c->stage[i];
    c->stage[i] = 0.0;
  }
}
