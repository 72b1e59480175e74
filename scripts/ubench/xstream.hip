// Micro-benchmark 6: cost of a cross-stream dependency (hipEventRecord on A + hipStreamWaitEvent on B) versus
// same-stream ordering, and the same DAG replayed as a hipGraph. Kernels are ~5 us each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void spin(float* p, int iters) {
  float x = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0001f, 0.5f);
  p[threadIdx.x] = x;
}
int main() {
  float *a, *b; hipMalloc(&a, 4096); hipMalloc(&b, 4096);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  hipEvent_t ea, eb; hipEventCreateWithFlags(&ea, hipEventDisableTiming); hipEventCreateWithFlags(&eb, hipEventDisableTiming);
  const int iters = 4000, hops = 200;
  auto now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  // same stream
  for (int w = 0; w < 2; ++w) {
    hipDeviceSynchronize(); double t0 = now();
    for (int i = 0; i < hops; ++i) { hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, a, iters); hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, b, iters); }
    hipDeviceSynchronize(); double t1 = now();
    if (w) printf("same stream           : %.2f us per kernel\n", (t1 - t0) / (2 * hops));
  }
  // ping-pong across two streams: A -> B -> A -> ...
  for (int w = 0; w < 2; ++w) {
    hipDeviceSynchronize(); double t0 = now();
    for (int i = 0; i < hops; ++i) {
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, a, iters); hipEventRecord(ea, sa); hipStreamWaitEvent(sb, ea, 0);
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sb, b, iters); hipEventRecord(eb, sb); hipStreamWaitEvent(sa, eb, 0);
    }
    hipDeviceSynchronize(); double t1 = now();
    if (w) printf("cross-stream ping-pong: %.2f us per kernel\n", (t1 - t0) / (2 * hops));
  }
  // the same chain captured in a graph
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(sa, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 20; ++i) {
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sa, a, iters); hipEventRecord(ea, sa); hipStreamWaitEvent(sb, ea, 0);
    hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, sb, b, iters); hipEventRecord(eb, sb); hipStreamWaitEvent(sa, eb, 0);
  }
  hipStreamEndCapture(sa, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 2; ++w) {
    hipDeviceSynchronize(); double t0 = now();
    for (int i = 0; i < 10; ++i) hipGraphLaunch(ge, sa);
    hipDeviceSynchronize(); double t1 = now();
    if (w) printf("graph of the ping-pong: %.2f us per kernel\n", (t1 - t0) / (2 * 20 * 10));
  }
  return 0;
}
