// l2_scratch_lab.hip — does a four-step transform's work buffer stay off the fabric when it is a small, reused, XCD-local
// scratch instead of a batch-sized buffer? (round 4; the review's item 2a; results: profiles/r04/)
//
// The memory system's side of a 65536-point frame (256 x 256), no arithmetic to speak of:
//   phase A  ("column tile")  one workgroup of 512 threads reads 32 columns x 256 rows of int8 IQ (16 two-byte loads per thread,
//            64-byte runs at a 512-byte stride: what fft_cols256_tile reads) and writes 8192 complex values to
//            work[k1 * 256 + n2] (16 eight-byte stores per thread, 256-byte runs)
//   phase B  ("row tile")     one workgroup reads 32 rows x 256 points of the work buffer (16 eight-byte loads per thread,
//            contiguous 64 KiB) and writes 8192 floats (dB values) in 128-byte runs
// A frame has 8 tiles of either kind. Three ways to run `frames` frames:
//   mode 0   two launches, a work buffer as large as the batch (what ships): every byte of it crosses the fabric twice
//   mode 1   ONE persistent launch: a group of 8 workgroups on one XCD (block b runs on XCD b mod 8: observed, not promised —
//            it only matters for speed) takes frame after frame; A, group barrier, B, group barrier; the group's 512 KiB
//            scratch slot is reused for every frame and is written and read through the same L2
//   mode 2   the same launch with a scratch slot per FRAME (batch-sized again): what the persistent form costs by itself
// G groups per XCD: 8 G workgroups per XCD = G / 4 per CU; scratch per XCD = G x 512 KiB (4 MiB of L2 per XCD).
// Output: us per batch and GB/s of IQ for each (mode, G); run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the bytes.
//
// The group barrier spins (all 8 workgroups of a group are resident by construction here: <= 1024 workgroups on an idle
// chip); a product kernel would have to bound that wait. This lab only asks what there is to gain.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t buffer_of(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

// phase A: tile `tile` (0..7) of a frame of int8 IQ -> work (float2[65536])
template <int ST_AUX>
__device__ __forceinline__ void phase_a(const char* __restrict__ iq_frame, float2* __restrict__ work, int tile, int t) {
  const int q = t & 31, j = t >> 5;
  const int n2 = tile * 32 + q;
  const __amdgpu_buffer_rsrc_t rin = buffer_of(iq_frame, 65536 * 2);
  float2 a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const unsigned short raw = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rin, ((j << 8) + n2) * 2, (16 * r) << 9, 2 /* nt */);
    a[r] = make_float2((float)(signed char)(raw & 0xff), (float)(signed char)(raw >> 8));
  }
  const __amdgpu_buffer_rsrc_t rw = buffer_of(work, 65536 * 8);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int k1 = j + 16 * k;
    const float2 y = make_float2(a[k].x + a[(k + 1) & 15].y, a[k].y - a[(k + 3) & 15].x);
    __builtin_amdgcn_raw_buffer_store_b64(*(const __attribute__((ext_vector_type(2))) unsigned*)&y, rw, ((k1 << 8) + n2) * 8, 0, ST_AUX);
  }
}

// phase B: rows [32 tile, 32 tile + 32) of work -> 8192 floats of out (float[65536]); LD_AUX = 16 (sc1) bypasses the CU's L1
template <int LD_AUX>
__device__ __forceinline__ void phase_b(const float2* __restrict__ work, float* __restrict__ out, int tile, int t) {
  const int rho = t >> 4, j = t & 15;
  const __amdgpu_buffer_rsrc_t rw = buffer_of(work, 65536 * 8);
  float acc[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rw, (((tile * 32 + rho) << 8) + j + 16 * r) * 8, 0, LD_AUX);
    acc[r] = __uint_as_float(v[0]) * 0.5f + __uint_as_float(v[1]);
  }
  const __amdgpu_buffer_rsrc_t ro = buffer_of(out, 65536 * 4);
  const int rr = t & 31, kb = t >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int d = kb + 16 * i;
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[i] + (float)d), ro, ((d << 8) + tile * 32 + rr) * 4, 0, 16 /* sc1: write-through */);
  }
}

__global__ __launch_bounds__(512, 8) void k_phase_a(const char* iq, float2* work) {
  const int f = blockIdx.x >> 3, tile = blockIdx.x & 7;
  phase_a<0>(iq + (size_t)f * 131072, work + (size_t)f * 65536, tile, threadIdx.x);
}
__global__ __launch_bounds__(512, 8) void k_phase_b(const float2* work, float* out) {
  const int f = blockIdx.x >> 3, tile = blockIdx.x & 7;
  phase_b<0>(work + (size_t)f * 65536, out + (size_t)f * 65536, tile, threadIdx.x);
}

// group barrier: 8 workgroups, one counter, monotonic target
__device__ __forceinline__ void group_barrier(unsigned* counter, unsigned target, int t) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's scratch stores have reached the L2
  __syncthreads();
  if (t == 0) {
    __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spin = 0; spin < (1 << 22) && __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) __builtin_amdgcn_s_sleep(2);  // (bounded: a lab must not hang the box)
  }
  __syncthreads();
}

// persistent: block b = x + 8 (w + 8 g): XCD x, workgroup w of group g of that XCD
__global__ __launch_bounds__(512, 8) void k_persistent(const char* iq, float2* scratch, float* out, unsigned* counters, int frames, int groups_per_xcd, int per_frame_scratch,
                                                       unsigned epoch0) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int x = b & 7, w = (b >> 3) & 7, g = b >> 6;
  const int group = x * groups_per_xcd + g, ngroups = 8 * groups_per_xcd;
  unsigned* counter = counters + 32 * group;  // (a cache line of its own)
  unsigned phase = epoch0;
  for (int f = group; f < frames; f += ngroups) {
    float2* work = scratch + (size_t)(per_frame_scratch ? f : group) * 65536;
    phase_a<0>(iq + (size_t)f * 131072, work, w, t);
    phase += 8;
    group_barrier(counter, phase, t);
    phase_b<16>(work, out + (size_t)f * 65536, w, t);
    phase += 8;
    group_barrier(counter, phase, t);  // every row tile has read the slot before the next frame's column tiles overwrite it
  }
}

int main(int argc, char** argv) {
  const int frames = argc > 1 ? atoi(argv[1]) : 128;
  const int reps = argc > 2 ? atoi(argv[2]) : 200;
  const int sets = 6;  // input / output sets in rotation (6 x (16 + 32) MiB at 128 frames: past the Infinity Cache)
  char* iq[sets];
  float* out[sets];
  float2 *work = nullptr, *scratch = nullptr;
  unsigned* counters = nullptr;
  for (int s = 0; s < sets; ++s) {
    CHECK(hipMalloc(&iq[s], (size_t)frames * 131072));
    CHECK(hipMemset(iq[s], 3 + s, (size_t)frames * 131072));
    CHECK(hipMalloc(&out[s], (size_t)frames * 65536 * 4));
  }
  CHECK(hipMalloc(&work, (size_t)frames * 65536 * 8));
  CHECK(hipMalloc(&scratch, (size_t)frames * 65536 * 8));
  CHECK(hipMalloc(&counters, 4 * 32 * 256));
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const double iq_bytes = (double)frames * 131072;
  printf("frames %d (65536 points, int8 IQ), reps %d; algorithmic: 2 B in + 4 B out per sample\n", frames, reps);
  printf("%-46s %10s %10s %12s\n", "variant", "us/batch", "GS/s", "6 B x S / t");
  auto report = [&](const char* name, float ms) {
    const double us = ms * 1e3 / reps;
    printf("%-46s %10.2f %10.1f %9.0f GB/s\n", name, us, frames * 65536.0 / us / 1e3, 3.0 * iq_bytes / us / 1e3);
  };
  // mode 0: two launches
  for (int pass = 0; pass < 2; ++pass) {
    CHECK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) {
      hipLaunchKernelGGL(k_phase_a, dim3(frames * 8), dim3(512), 0, st, (const char*)iq[r % sets], work);
      hipLaunchKernelGGL(k_phase_b, dim3(frames * 8), dim3(512), 0, st, (const float2*)work, out[r % sets]);
    }
    CHECK(hipEventRecord(e1, st));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (pass) report("two launches, batch-sized work buffer", ms);
  }
  unsigned epoch = 0;
  for (int per_frame = 0; per_frame < 2; ++per_frame) {
    for (int G : {2, 4, 8, 16}) {
      if (8 * G > frames) continue;
      CHECK(hipMemsetAsync(counters, 0, 4 * 32 * 256, st));
      epoch = 0;
      const int per_group = (frames + 8 * G - 1) / (8 * G);  // frames per group (the same for every group when 8 G divides frames)
      for (int pass = 0; pass < 2; ++pass) {
        CHECK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) {
          hipLaunchKernelGGL(k_persistent, dim3(64 * G), dim3(512), 0, st, (const char*)iq[r % sets], scratch, out[r % sets], counters, frames, G, per_frame, epoch);
          epoch += 16u * (unsigned)per_group;
        }
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (pass) {
          char name[96];
          snprintf(name, sizeof name, "persistent, G=%d (%.2f WG/CU), %s", G, G / 4.0, per_frame ? "scratch slot per frame" : "slot reused (G x 512 KiB / XCD)");
          report(name, ms);
        }
      }
    }
  }
  return 0;
}
