// The headline kernel alone, compiled for the device only by tests/test_kernel_resources.py: its register budget is part of the
// design (512 threads x 64 VGPRs = four workgroups per CU whatever their roles) and easy to lose — a loop around
// the tile evaluation, even one never taken, cost eleven more spilled registers and 12 % of the step (round 3).
#include "../../rtl-sdr-scanner-cpp_amd/csrc/scan_step.h"

template __global__ void ss::k_scan_step<ss::FMT_CF32, false, 2, true, false, 0>(ss::StepArgs);  // 8192 points, CF32, no spectrogram branch: what bench.py times
template __global__ void ss::k_scan_step<ss::FMT_CS8, false, 2, true, false, 2>(ss::StepArgs);   // long transforms, int8: config 3's column launch
template __global__ void ss::k_scan_step<ss::FMT_CF32, false, 2, true, false, 4>(ss::StepArgs);  // 2^20 points in two passes: the row tiles as the FFT role (config 5)
template __global__ void ss::k_scan_step<ss::FMT_CS8, false, 2, true, false, 7>(ss::StepArgs);   // 65536 points, CF32 / plane-keeping int8 calls of up to 128 frames: one launch per call in the four-step form (and the form config 3 shipped in round 4)
template __global__ void ss::k_scan_step<ss::FMT_CS8, false, 2, true, false, 8>(ss::StepArgs);   // 65536 points, int8, detect mode: the radix-8 fold, two residues per workgroup (config 3 as it ships: 128 registers, four waves per SIMD)
template __global__ void ss::k_scan_step<ss::FMT_CS8, false, 2, true, false, 9>(ss::StepArgs);   // 131072 points, int8, detect mode: the same fold with radix 16 (what getFft picks at 20 MS/s)
