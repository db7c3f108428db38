// overlap.hip -- can the control kernel of slot s+1 START while the row pass of slot s still runs, and take over
// through ONE completion counter, so that the kernel boundary rows(s) -> ctrl(s+1) disappears?
//
// The sampler's chain of dependencies is rows(s) -> ctrl(s+1) -> rows(s+1) -> ...; both arrows are kernel boundaries
// today (~1.2-1.5 us from the last workgroup's end to the next kernel's first instruction, plus the ~1.5 us a fresh
// kernel needs until its first data).  round 2 priced a RESIDENT control kernel with every row workgroup spinning
// (tools/microbench/handoff.hip: twice the price of the boundaries).  This prices the other way round: the row pass
// stays an ordinary launch and never waits; the control kernel is launched WITHOUT the barrier bit
// (hipExtAnyOrderLaunch), does what does not depend on the row pass, then its 39 workgroups poll one counter that
// every row workgroup increments once, last thing; the next row pass is an ordinary launch again (barrier: it waits
// for both).  "Work" is emulated by timed spins so that only the hand-over differs between the modes:
//   rows: ROWS_US of work per workgroup, one payload word written write-through (agent-scope relaxed store), counter++
//   ctrl: PRE_US of work that needs nothing from the row pass, [mode 1: poll], POST_US of dependent work, checks the
//         payload words (a stale one is counted)
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define SPIN_LIMIT 20000000ll /* 0.2 s of the 100 MHz clock: a lost hand-over ends the run instead of hanging the GPU */

struct Sh {
  unsigned long long done[64 * 16];  // row workgroups finished (cumulative), spread over 64 lines: atomics on ONE line
                                     // serialise at ~12 ns each (1024 of them: 12 us)
  unsigned long long failed, stale, pad1[14];
  unsigned long long payload[1024 * 16];  // one 128-byte line per row workgroup
  unsigned long long jobs[64 * 16];       // what ctrl writes for the next row pass
};

__device__ __forceinline__ void spin_us(double us) {
  const long long t0 = wall_clock64();
  const long long ticks = (long long)(us * 100.0);
  while (wall_clock64() - t0 < ticks) {}
}

__global__ __launch_bounds__(256) void k_rows(Sh* s, unsigned long long round, double rows_us) {
  // reads what ctrl wrote (after a kernel boundary: plainly visible)
  const unsigned long long j = s->jobs[(blockIdx.x & 63) * 16];
  if (j != round && threadIdx.x == 0) atomicAdd(&s->stale, 1ull << 32);
  spin_us(rows_us);
  if (threadIdx.x == 0)  // the result ctrl will read: write-through, no workgroup-level L2 write-back
    __hip_atomic_store(&s->payload[blockIdx.x * 16], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __builtin_amdgcn_s_waitcnt(0);  // the store is acknowledged by the memory side
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&s->done[(blockIdx.x & 63) * 16], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool POLL>
__global__ __launch_bounds__(256) void k_ctrl(Sh* s, unsigned long long round, unsigned long long want_done, int rows_wgs,
                                              double pre_us, double post_us) {
  spin_us(pre_us);  // control word, previous job records, draws: nothing of the row pass
  if (POLL) {
    if (threadIdx.x < 64) {  // wave 0: lane l polls line l, the 64 lines add up
      const long long t0 = wall_clock64();
      for (;;) {
        unsigned long long v = __hip_atomic_load(&s->done[threadIdx.x * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (v >= want_done) break;
        if (wall_clock64() - t0 > SPIN_LIMIT) { __hip_atomic_store(&s->failed, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);  // once, after the wait
  }
  // the row pass's results: every workgroup looks at a few lines
  for (int i = threadIdx.x + blockIdx.x * 256; i < rows_wgs; i += 256 * gridDim.x)
    if (s->payload[i * 16] != round + 1) atomicAdd(&s->stale, 1ull);
  spin_us(post_us);
  if (threadIdx.x == 0) s->jobs[blockIdx.x * 16] = round + 1;
  if (threadIdx.x == 0 && blockIdx.x < 25) s->jobs[(39 + blockIdx.x) * 16] = round + 1;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 3000;
  const int rows_wgs = argc > 2 ? atoi(argv[2]) : 1024;
  const double rows_us = argc > 3 ? atof(argv[3]) : 3.0, pre_us = argc > 4 ? atof(argv[4]) : 1.0, post_us = argc > 5 ? atof(argv[5]) : 3.0;
  Sh* s;
  HC(hipMalloc(&s, sizeof(Sh)));
  static Sh h;
  hipStream_t st;
  HC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: two ordinary launches per round; 1: ctrl polls but is launched in order (the price of the poll);
    // 2: ctrl launched without the barrier bit, polls
    HC(hipMemsetAsync(s, 0, sizeof(Sh), st));
    HC(hipStreamSynchronize(st));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < rounds; ++r) {
      hipLaunchKernelGGL(k_rows, dim3(rows_wgs), dim3(256), 0, st, s, (unsigned long long)r, rows_us);
      const unsigned long long want = (unsigned long long)rows_wgs * (unsigned long long)(r + 1);
      if (mode == 0) hipLaunchKernelGGL(k_ctrl<false>, dim3(39), dim3(256), 0, st, s, (unsigned long long)r, want, rows_wgs, pre_us, post_us);
      else if (mode == 1) hipLaunchKernelGGL(k_ctrl<true>, dim3(39), dim3(256), 0, st, s, (unsigned long long)r, want, rows_wgs, pre_us, post_us);
      else hipExtLaunchKernelGGL(k_ctrl<true>, dim3(39), dim3(256), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, s, (unsigned long long)r, want, rows_wgs, pre_us, post_us);
    }
    HC(hipStreamSynchronize(st));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    HC(hipMemcpy(&h, s, sizeof(h.done) + 256, hipMemcpyDeviceToHost));
    unsigned long long dsum = 0;
    for (int i = 0; i < 64; ++i) dsum += h.done[i * 16];
    printf("mode %d (%s): %.2f us per round (rows %d wgs %.1f us, ctrl pre %.1f + post %.1f us)  done=%llu failed=%llu stale payload=%llu stale jobs=%llu\n",
           mode, mode == 0 ? "two ordinary launches" : mode == 1 ? "ctrl polls, launched in order" : "ctrl launched without the barrier bit, polls",
           us, rows_wgs, rows_us, pre_us, post_us, dsum, h.failed, h.stale & 0xffffffffull, h.stale >> 32);
  }
  return 0;
}
