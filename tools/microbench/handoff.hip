// handoff.hip -- how fast can two CONCURRENTLY running kernels hand a round of work to each other on MI355X?
//
// The sampler's slot is {k_ctrl ; k_rows} with a kernel boundary on either side (~1.2 us each, plus the
// ~1.2-1.8 us a fresh kernel needs before it has its first data).  The alternative this program prices:
// a control kernel that STAYS RESIDENT (39 workgroups) next to a stream of row-pass kernels (1024
// workgroups each), the two handing over through flags in device memory:
//     ctrl, round r : wait until every workgroup of rows(r-1) has signalled; [work]; publish "jobs r"
//     rows(r)       : its workgroups spin until "jobs r" is published; [work]; signal done (64 counter lines)
// It measures the time per round with no work at all in either kernel, against the same number of
// rounds as plain back-to-back launches of two empty kernels on one stream.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/handoff.hip -o /tmp/handoff && /tmp/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <vector>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define DONE_LINES 64
#define LINE 16 /* unsigned long long per 128-byte line */
#define SPIN_LIMIT 20000000ll /* 100 MHz ticks = 0.2 s: a lost hand-off ends the run instead of hanging the GPU */

struct Sh {
  unsigned long long jobs_round;             // published by ctrl: rounds whose jobs are ready
  unsigned long long pad0[15];
  unsigned long long done[DONE_LINES * LINE];  // per line: workgroups of the row passes that have finished (cumulative)
  unsigned long long started, failed;
  unsigned long long payload[64 * 8];          // what ctrl writes per round and rows read back (one line per lane)
  unsigned long long check;
};

// polling load: relaxed, agent scope (goes to the memory side, no cache maintenance); ONE acquire fence after
// the wait has ended (an acquire load in the loop invalidates the caches on every poll: 55 us per round)
__device__ __forceinline__ unsigned long long ld_acq(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void acquire_fence() { __atomic_thread_fence(__ATOMIC_ACQUIRE); }

__global__ __launch_bounds__(256) void k_ctrl_resident(Sh* s, int rounds, int rows_wgs, long long* t_round) {
  const int lane = threadIdx.x & 63;
  if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(&s->started, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x < 64) {
      // wait for rows(r-1): the 64 counter lines add up to rows_wgs * r
      const unsigned long long want = (unsigned long long)rows_wgs * (unsigned long long)r;
      const long long t0 = wall_clock64();
      for (;;) {
        unsigned long long v = ld_acq(&s->done[lane * LINE]);
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (v >= want) { acquire_fence(); break; }
        if (ld_acq(&s->failed) != 0) return;
        if (wall_clock64() - t0 > SPIN_LIMIT) { __hip_atomic_store(&s->failed, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); return; }
      }
    }
    __syncthreads();
    // "work": one line per workgroup of payload, then publish (the last workgroup to arrive publishes)
    if (threadIdx.x == 0) {
      s->payload[blockIdx.x * 8] = (unsigned long long)r + 1;
      __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned long long t = atomicAdd(&s->payload[63 * 8 + 1], 1ull);  // ticket
      if (t == (unsigned long long)gridDim.x * (r + 1) - 1) {
        __hip_atomic_store(&s->jobs_round, (unsigned long long)r + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (t_round) t_round[r] = wall_clock64();
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_rows_waiting(Sh* s, int r) {
  __shared__ unsigned long long seen;
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (ld_acq(&s->jobs_round) < (unsigned long long)r + 1) {
      if (ld_acq(&s->failed) != 0) break;  // somebody already gave up: drain quickly
      if (wall_clock64() - t0 > SPIN_LIMIT) { __hip_atomic_store(&s->failed, 2ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); break; }
    }
    acquire_fence();
    seen = s->payload[(blockIdx.x % 39) * 8];  // read something ctrl wrote for this round
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (seen != (unsigned long long)r + 1) atomicAdd(&s->check, 1ull);
#ifndef NO_ROWS_FENCE
    __threadfence();  // release of the workgroup's plain stores (an L2 write-back on this chip)
#endif
    atomicAdd(&s->done[(blockIdx.x % DONE_LINES) * LINE], 1ull);
  }
}

__global__ __launch_bounds__(256) void k_empty_ctrl(Sh* s, int r) {
  if (threadIdx.x == 0) s->payload[blockIdx.x * 8] = r;
}
__global__ __launch_bounds__(256) void k_empty_rows(Sh* s, int r) {
  if (threadIdx.x == 0 && s->payload[(blockIdx.x % 39) * 8] == 0xdeadbeefull) s->check = 1;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  const int rows_wgs = argc > 2 ? atoi(argv[2]) : 1024;
  const int use_mask = argc > 3 ? atoi(argv[3]) : 0;
  Sh* s;
  HC(hipMalloc(&s, sizeof(Sh)));
  long long* t_round;
  HC(hipMalloc(&t_round, sizeof(long long) * rounds));
  hipStream_t sa, sb;
  if (use_mask) {
    // disjoint CU sets: ctrl on the first 48 CUs (6 per XCD if the mask bits interleave over the XCDs), rows on the rest
    std::vector<uint32_t> ma(8, 0), mb(8, 0);
    for (int i = 0; i < 256; ++i) (i < 48 ? ma : mb)[i / 32] |= 1u << (i % 32);
    HC(hipExtStreamCreateWithCUMask(&sa, 8, ma.data()));
    HC(hipExtStreamCreateWithCUMask(&sb, 8, mb.data()));
  } else {
    HC(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    HC(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  }
  hipEvent_t e0, e1;
  HC(hipEventCreate(&e0));
  HC(hipEventCreate(&e1));

  // ---- baseline: two empty kernels per round, one stream
  for (int rep = 0; rep < 2; ++rep) {
    HC(hipMemset(s, 0, sizeof(Sh)));
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0, sa));
    for (int r = 0; r < rounds; ++r) {
      hipLaunchKernelGGL(k_empty_ctrl, dim3(39), dim3(256), 0, sa, s, r);
      hipLaunchKernelGGL(k_empty_rows, dim3(rows_wgs), dim3(256), 0, sa, s, r);
    }
    HC(hipEventRecord(e1, sa));
    HC(hipEventSynchronize(e1));
    float ms;
    HC(hipEventElapsedTime(&ms, e0, e1));
    if (rep) printf("two empty kernels per round, one stream:         %.2f us per round\n", 1e3 * ms / rounds);
  }
  // ---- resident control kernel + one row-pass kernel per round, flags
  for (int rep = 0; rep < 2; ++rep) {
    HC(hipMemset(s, 0, sizeof(Sh)));
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0, sb));
    hipLaunchKernelGGL(k_ctrl_resident, dim3(39), dim3(256), 0, sa, s, rounds, rows_wgs, t_round);
    // the resident kernel must be running before anything that waits for it is queued
    unsigned long long started = 0;
    for (int i = 0; i < 2000000 && !started; ++i) HC(hipMemcpy(&started, &s->started, 8, hipMemcpyDeviceToHost));
    if (!started) { printf("control kernel did not start\n"); return 1; }
    for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_rows_waiting, dim3(rows_wgs), dim3(256), 0, sb, s, r);
    HC(hipEventRecord(e1, sb));
    HC(hipEventSynchronize(e1));
    HC(hipStreamSynchronize(sa));
    float ms;
    HC(hipEventElapsedTime(&ms, e0, e1));
    Sh h;
    HC(hipMemcpy(&h, s, sizeof(Sh), hipMemcpyDeviceToHost));
    std::vector<long long> tr(rounds);
    HC(hipMemcpy(tr.data(), t_round, sizeof(long long) * rounds, hipMemcpyDeviceToHost));
    if (rep) {
      double med = 0;
      std::vector<double> d;
      for (int r = rounds / 2; r + 1 < rounds; ++r) d.push_back((tr[r + 1] - tr[r]) * 0.01);
      std::sort(d.begin(), d.end());
      med = d[d.size() / 2];
      printf("resident control + row-pass kernel per round (%s): %.2f us per round by events, %.2f us median by the device clock; failed=%llu stale reads=%llu\n",
             use_mask ? "disjoint CU masks" : "no CU masks", 1e3 * ms / rounds, med, h.failed, h.check);
    }
  }
  return 0;
}
