// sync_tail_lab — what the end of a short pipelined run costs on the host's clock (round 6, session 18: the 20-step form's GPU span is
// 458 us, the host measures 496). The shape of ss_flush's drain, with kernels that only pass time: 20 launches alternating over two
// non-blocking streams (each lasts two launch periods, so two are in flight), then per queue a "detect" and an "emit" launch, then the
// public stream waits for both queues (event record + stream wait) and runs a short "ring fill"; then the host waits.
//   joins  J0  as the library does it: ring fill on the public stream behind the join
//          J1  ring fill on the public stream BEFORE the drain launches are enqueued (it waits for the last transform launches only),
//              the join behind the drain carries no kernel
//          J2  no public stream at the end: the ring fill rides behind queue 1's emit (queue 1 waits for queue 0's detect instead)
//          J3  J2, then the public stream joins both queues (no kernel behind the join): the library's contract kept
//          J4  J3 with a WAITER on the public stream in front of the join: one wave that sleeps until the last kernels of both queues
//              have said so in device memory (bounded), so that the join's barrier packets are looked at when they are already satisfied
//          J5  J0 with the same waiter in front of its join
//   waits  W0  hipDeviceSynchronize
//          W1  hipStreamSynchronize(public), then hipDeviceSynchronize
//          W2  hipEventSynchronize on an event recorded behind the last command of each stream, then hipDeviceSynchronize
//          W3  spin on a word in pinned host memory the last kernel writes, then hipDeviceSynchronize
//          W4  an event recorded behind the last command of each stream (nobody waits for it), then hipDeviceSynchronize
// Per combination: host time from the first enqueue to the return of hipDeviceSynchronize (median of 40 runs), the same to the moment a
// second host thread sees the last kernel's word (the GPU's end on the host's clock), and the GPU's own span (wall_clock64 stamps).
// build: hipcc --offload-arch=gfx950 -O3 -pthread -o sync_tail_lab sync_tail_lab.hip    run: gpurun -- scripts/ubench/sync_tail_lab
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

// passes `ticks` of the 100 MHz clock; stamps[0] = first tick seen, stamps[1] = last; then writes `value` to *flag (host memory) if any
__global__ void k_pass(long long ticks, long long* stamps, volatile unsigned* flag, unsigned value, unsigned* done = nullptr) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0 && stamps) stamps[0] = t0;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if (stamps) stamps[1] = wall_clock64();
    if (flag) {
      __threadfence_system();
      *flag = value;
    }
    if (done) {
      __threadfence();
      __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// the waiter: sleeps until *done == want (or `limit` ticks have passed), then `linger` ticks more
__global__ void k_wait(unsigned* done, unsigned want, long long limit, long long linger) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want && wall_clock64() - t0 < limit) __builtin_amdgcn_s_sleep(16);
  const long long t1 = wall_clock64();
  while (wall_clock64() - t1 < linger) __builtin_amdgcn_s_sleep(4);
}

using clk = std::chrono::steady_clock;
static double us_since(clk::time_point a) { return std::chrono::duration<double, std::micro>(clk::now() - a).count(); }

int main() {
  hipStream_t pub, q[2];
  // (PUB_LAST=1: the public stream is created behind the two queues — does a device-wide synchronisation visit streams in that order?)
  const bool pub_last = getenv("PUB_LAST") && atoi(getenv("PUB_LAST"));
  if (!pub_last) CK(hipStreamCreateWithFlags(&pub, hipStreamNonBlocking));
  for (auto& s : q) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  if (pub_last) CK(hipStreamCreateWithFlags(&pub, hipStreamNonBlocking));
  hipEvent_t ev[8];
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  long long* d_stamps;
  CK(hipMalloc(&d_stamps, sizeof(long long) * 64));
  unsigned* h_flag;
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocCoherent));
  *h_flag = 0;
  std::atomic<double> seen_us{0};
  std::atomic<unsigned> want{0};
  std::atomic<bool> quit{false};
  clk::time_point t0;
  std::thread watcher([&]() {
    while (!quit.load(std::memory_order_relaxed)) {
      const unsigned w = want.load(std::memory_order_acquire);
      if (w && *(volatile unsigned*)h_flag == w) {
        seen_us.store(us_since(t0));
        want.store(0, std::memory_order_release);
      }
    }
  });
  const int steps = 20;
  const long long T = 100;  // ticks per us
  unsigned seq = 0;
  unsigned* d_done;
  CK(hipMalloc(&d_done, 64));
  const long long linger = getenv("LINGER") ? atoll(getenv("LINGER")) : 2 * T;
  {  // hipDeviceSynchronize with nothing to wait for
    CK(hipDeviceSynchronize());
    std::vector<double> v;
    for (int i = 0; i < 50; ++i) {
      const auto a = clk::now();
      CK(hipDeviceSynchronize());
      v.push_back(us_since(a));
    }
    std::sort(v.begin(), v.end());
    printf("hipDeviceSynchronize on an idle device: %.1f us (median of 50)\n", v[25]);
  }
  for (int J = 0; J < 6; ++J)
    for (int W = 0; W < 5; ++W) {
      if (J >= 3 && W != 0 && W != 4) continue;
      std::vector<double> total, gpu_end, span, tail_gap;
      for (int rep = 0; rep < 44; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipMemset(d_stamps, 0, sizeof(long long) * 64));
        CK(hipMemset(d_done, 0, 64));
        CK(hipDeviceSynchronize());
        const unsigned value = ++seq;
        seen_us.store(0);
        want.store(value, std::memory_order_release);
        t0 = clk::now();
        for (int L = 0; L < steps; ++L) hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, q[L & 1], 40 * T, L == 0 ? d_stamps : nullptr, nullptr, 0u);
        volatile unsigned* last_flag = h_flag;
        if (J == 1) {  // the ring fill needs the transforms only
          for (int k = 0; k < 2; ++k) {
            CK(hipEventRecord(ev[4 + k], q[k]));
            CK(hipStreamWaitEvent(pub, ev[4 + k], 0));
          }
          hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, pub, 4 * T, d_stamps + 6, nullptr, 0u);
        }
        for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, q[k], 20 * T, nullptr, nullptr, 0u);  // "detect"
        const bool fill_on_q1 = J >= 2 && J <= 4, waiter = J >= 4;
        if (fill_on_q1) {
          CK(hipEventRecord(ev[0], q[0]));
          CK(hipStreamWaitEvent(q[1], ev[0], 0));
        }
        hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, q[0], 7 * T, d_stamps + 2, nullptr, 0u, waiter ? d_done : nullptr);  // "emit"
        hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, q[1], 7 * T, d_stamps + 4, J == 1 ? last_flag : nullptr, value, J == 5 ? d_done : nullptr);
        if (fill_on_q1) hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, q[1], 4 * T, d_stamps + 6, last_flag, value, waiter ? d_done : nullptr);
        if (waiter) hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, pub, d_done, 2u, 2000 * T, linger);
        if (J != 2) {
          for (int k = 0; k < 2; ++k) {
            CK(hipEventRecord(ev[k], q[k]));
            CK(hipStreamWaitEvent(pub, ev[k], 0));
          }
        }
        if (J == 0 || J == 5) hipLaunchKernelGGL(k_pass, dim3(1), dim3(64), 0, pub, 4 * T, d_stamps + 6, last_flag, value);
        if (W == 1) CK(hipStreamSynchronize(pub));
        if (W == 4) {
          CK(hipEventRecord(ev[2], pub));
          CK(hipEventRecord(ev[3], q[0]));
          CK(hipEventRecord(ev[6], q[1]));
        }
        if (W == 2) {
          CK(hipEventRecord(ev[2], pub));
          CK(hipEventRecord(ev[3], q[0]));
          CK(hipEventRecord(ev[6], q[1]));
          CK(hipEventSynchronize(ev[2]));
          CK(hipEventSynchronize(ev[3]));
          CK(hipEventSynchronize(ev[6]));
        }
        if (W == 3)
          while (*(volatile unsigned*)h_flag != value) {
          }
        const double t_w = us_since(t0);
        CK(hipDeviceSynchronize());
        const double t_all = us_since(t0);
        while (want.load(std::memory_order_acquire) != 0) {
        }
        long long st[8];
        CK(hipMemcpy(st, d_stamps, sizeof(st), hipMemcpyDeviceToHost));
        if (rep < 4) continue;
        total.push_back(t_all);
        gpu_end.push_back(seen_us.load());
        const long long last_end = std::max(std::max(st[3], st[5]), st[7]);
        span.push_back((last_end - st[0]) / (double)T);
        tail_gap.push_back((st[6] - std::max(st[3], st[5])) / (double)T);  // ring fill's start behind the later emit's end (negative: it ran beside the drain)
        (void)t_w;
      }
      const auto med = [](std::vector<double>& v) {
        std::sort(v.begin(), v.end());
        return v[v.size() / 2];
      };
      const double a = med(total), b = med(gpu_end), c = med(span), d = med(tail_gap);
      printf("J%d W%d: host total %7.1f us | GPU's last word seen at %7.1f (host returns %5.1f later) | GPU span %7.1f | ring fill starts %6.1f us behind the later emit's end\n", J, W, a, b, a - b, c, d);
    }
  quit.store(true);
  watcher.join();
  return 0;
}
