// Micro-benchmark 7: two dependent kernel chains (F: 25 us, D: 15 us + E: 5 us per step, half a chip each) with the
// scan chain's cross edges (F_i -> D_i, D_i -> F_{i+2}), as (a) one stream, sequential; (b) one hipGraph of S steps
// built with explicit nodes, launched repeatedly. Reports GPU time per step and host time per hipGraphLaunch.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void busy(float* p, long long ticks) {  // holds its workgroups for `ticks` of the 100 MHz clock
  const long long until = wall_clock64() + ticks;
  float x = p[threadIdx.x & 63];
  while (wall_clock64() < until) x = fmaf(x, 1.0001f, 0.5f);
  if (x == 12345.0f) p[0] = x;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
  const int S = argc > 1 ? atoi(argv[1]) : 8;
  float* buf; hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const long long tF = 2500, tD = 1500, tE = 500;
  const dim3 gridF(512), gridD(512), gridE(256), block(256);  // F and D each fill half of the 1024+ resident slots
  // (a) sequential on one stream
  for (int w = 0; w < 2; ++w) {
    hipDeviceSynchronize(); const double t0 = now();
    for (int i = 0; i < 10 * S; ++i) {
      hipLaunchKernelGGL(busy, gridF, block, 0, st, buf, tF);
      hipLaunchKernelGGL(busy, gridD, block, 0, st, buf, tD);
      hipLaunchKernelGGL(busy, gridE, block, 0, st, buf, tE);
    }
    hipDeviceSynchronize(); const double t1 = now();
    if (w) printf("one stream, sequential : %.1f us per step\n", (t1 - t0) / (10 * S));
  }
  // (b) graph with explicit nodes
  hipGraph_t g; hipGraphCreate(&g, 0);
  std::vector<hipGraphNode_t> F(S), D(S), E(S);
  auto add = [&](hipGraphNode_t* node, std::vector<hipGraphNode_t> deps, dim3 grid, long long ticks) {
    hipKernelNodeParams kp{}; void* args[2]; float* pb = buf; long long tk = ticks;
    args[0] = &pb; args[1] = &tk;
    kp.func = (void*)busy; kp.gridDim = grid; kp.blockDim = block; kp.sharedMemBytes = 0; kp.kernelParams = args; kp.extra = nullptr;
    if (hipGraphAddKernelNode(node, g, deps.data(), deps.size(), &kp) != hipSuccess) { printf("add node failed\n"); exit(1); }
  };
  for (int i = 0; i < S; ++i) {
    std::vector<hipGraphNode_t> dF; if (i > 0) dF.push_back(F[i - 1]); if (i > 1) dF.push_back(D[i - 2]);
    add(&F[i], dF, gridF, tF);
    std::vector<hipGraphNode_t> dD{F[i]}; if (i > 0) dD.push_back(E[i - 1]);
    add(&D[i], dD, gridD, tD);
    add(&E[i], {D[i]}, gridE, tE);
  }
  hipGraphExec_t ge;
  if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
  for (int w = 0; w < 2; ++w) {
    hipDeviceSynchronize(); const double t0 = now(); double host = 0;
    for (int i = 0; i < 10; ++i) { const double h0 = now(); hipGraphLaunch(ge, st); host += now() - h0; }
    hipDeviceSynchronize(); const double t1 = now();
    if (w) printf("graph of %d steps       : %.1f us per step, hipGraphLaunch %.1f us host each (%.1f us per step)\n", S, (t1 - t0) / (10 * S), host / 10, host / 10 / S);
  }
  return 0;
}
