// probe_boundary.hip -- what does a PHASE BOUNDARY cost on an MI355X: (A) a kernel boundary inside a hipGraph (the step has 16 of them),
// (B) the same kernels on a stream, (C) a grid barrier inside ONE persistent kernel (agent-scope release / acquire around an arrival
// counter) -- each with 1 KB, 64 KB and 256 KB of fresh output per workgroup and phase, which the NEXT phase of ANOTHER workgroup
// (another XCD) reads back and checks.  Decides whether a persistent per-step kernel could ever beat the launches (DESIGN (f)).
// Round 5 (the review's request): (D) the same persistent kernel with an XCD-HIERARCHICAL barrier (MI355X_MICROARCH.md row
// `barrier-xcd`: per-XCC arrival counter, the last arriver of an XCC releases, arrives at a top counter, acquires, and publishes the
// XCC's generation; everybody else polls its own XCC's generation), and a third store kind: sc1 write-through stores (row
// `publish-large`), so that the release has nothing left to write back.
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/probe_boundary tools/probe_boundary.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int NTHR = 1024;

__device__ __forceinline__ float val(int ph, int wg, size_t i) { return (float)((ph * 131 + wg * 7 + (int)(i & 1023)) & 0xffff); }

__device__ __forceinline__ void spin(int ticks) {                 // wall_clock64: 100 MHz
  const unsigned long long t0 = wall_clock64();
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(2);
}

// one phase: wait `ticks`, check what workgroup (wg + 37) % n wrote in phase ph - 1 (first 1024 floats), write own region for ph
__device__ __forceinline__ void phase_body(float* buf, size_t w, int ticks, int ph, int* errs, int nt) {
  const int wg = blockIdx.x, n = gridDim.x, tid = threadIdx.x;
  spin(ticks);
  if (ph > 0) {
    const int nb = (wg + 37) % n;
    const float* r = buf + ((size_t)((ph - 1) & 1) * n + nb) * w;
    const size_t i = (size_t)tid % w;
    const float got = nt ? __builtin_nontemporal_load(r + i) : r[i];
    if (got != val(ph - 1, nb, i)) atomicAdd(errs, 1);
  }
  float* o = buf + ((size_t)(ph & 1) * n + wg) * w;
  for (size_t i = tid; i < w; i += NTHR) {
    const float v = val(ph, wg, i);
    if (nt == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(o + i), "v"(v) : "memory");     // write-through to memory
    else if (nt) __builtin_nontemporal_store(v, o + i);
    else o[i] = v;
  }
}

__global__ __launch_bounds__(NTHR) void k_phase(float* buf, size_t w, int ticks, int ph, int* errs, int nt) {
  phase_body(buf, w, ticks, ph, errs, nt);
}

__global__ __launch_bounds__(NTHR) void k_persist(float* buf, size_t w, int ticks, int nphase, int* errs, unsigned* cnt, int nt) {
  for (int ph = 0; ph < nphase; ++ph) {
    phase_body(buf, w, ticks, ph, errs, nt);
    // __syncthreads() alone is `s_barrier` here (workgroup scope needs no wait for global stores on gfx942+): wave 0's write-back
    // below would overtake the other waves' stores still on their way to L2.  Every wave first waits for its own acknowledgements.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);     // release: L2 write-back at agent scope
      const unsigned target = (unsigned)(ph + 1) * gridDim.x;
      const unsigned long long tb = wall_clock64();              // never hang the box: give up after 0.5 s and count it
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - tb > 50000000ull) { atomicAdd(errs, 1000000); break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");          // acquire: invalidate
    }
    __syncthreads();
  }
}

// (D) XCD-hierarchical barrier.  bar: [0] top counter, [32 + 32 x] arrivals of XCC x, [320 + 32 x] generation of XCC x,
// [640 + 32 x] workgroups on XCC x (census), [960] census counter; one 128-byte line per word that is polled.
__device__ __forceinline__ bool poll_ge(unsigned* p, unsigned target, int* errs) {
  const unsigned long long tb = wall_clock64();
  while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(1);
    if (wall_clock64() - tb > 50000000ull) { atomicAdd(errs, 1000000); return false; }     // never hang the box
  }
  return true;
}
__global__ __launch_bounds__(NTHR) void k_persist_xcd(float* buf, size_t w, int ticks, int nphase, int* errs, unsigned* bar, int nt) {
  __shared__ unsigned s_nx, s_nact, s_xcc;
  if (threadIdx.x == 0) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 20) & 7;       // HW_REG_XCC_ID[3:0]
    s_xcc = xcc;
    // census (once): how many workgroups run on my XCC, how many XCCs are populated -- the block -> XCD map is not defined
    __hip_atomic_fetch_add(bar + 640 + 32 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(bar + 960, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    poll_ge(bar + 960, gridDim.x, errs);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    unsigned nact = 0;
    for (int x = 0; x < 8; ++x) nact += __hip_atomic_load(bar + 640 + 32 * x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0;
    s_nx = __hip_atomic_load(bar + 640 + 32 * xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_nact = nact;
  }
  __syncthreads();
  const unsigned nx = s_nx, nact = s_nact, xcc = s_xcc;
  for (int ph = 0; ph < nphase; ++ph) {
    phase_body(buf, w, ticks, ph, errs, nt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every wave: its stores are acknowledged by L2 (or memory: sc1)
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned gen = (unsigned)ph + 1;
      const unsigned old = __hip_atomic_fetch_add(bar + 32 + 32 * xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == nx * gen) {                                  // last arriver of this XCC: the XCC's leader for this phase
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");        // ONE L2 write-back per XCC
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        poll_ge(bar, nact * gen, errs);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(bar + 320 + 32 * xcc, gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        poll_ge(bar + 320 + 32 * xcc, gen, errs);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int nphase = argc > 1 ? atoi(argv[1]) : 32, grid = argc > 2 ? atoi(argv[2]) : 256, reps = 20;
  const int spin_us[2] = {2, 20};
  const size_t ws[3] = {256, 16384, 65536};                       // floats per workgroup and phase: 1 KB, 64 KB, 256 KB
  float* buf; int* errs; unsigned* cnt; unsigned* bar;
  CK(hipMalloc(&bar, 1024 * 4));
  CK(hipMalloc(&buf, 2 * (size_t)grid * 65536 * sizeof(float)));
  CK(hipMalloc(&errs, 4)); CK(hipMalloc(&cnt, 4));
  CK(hipMemset(errs, 0, 4));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("grid %d x %d threads, %d phases per run, %d runs; us per phase = (total / phases) - spin\n", grid, NTHR, nphase, reps);
  for (int nt = 0; nt < 3; ++nt)
    for (int si = 0; si < 2; ++si)
      for (int wi = 0; wi < 3; ++wi) {
        const size_t w = ws[wi]; const int ticks = spin_us[si] * 100;
        float ms;
        // (A) hipGraph of nphase dependent kernel nodes
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int ph = 0; ph < nphase; ++ph) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, ph, errs, nt);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double a = ms * 1e3 / (reps * nphase) - spin_us[si];
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // (B) the same launches on the stream
        for (int ph = 0; ph < nphase; ++ph) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, ph, errs, nt);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r)
          for (int ph = 0; ph < nphase; ++ph) hipLaunchKernelGGL(k_phase, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, ph, errs, nt);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        const double b = ms * 1e3 / (reps * nphase) - spin_us[si];
        // (C) one persistent kernel, grid barrier per phase
        double c = -1;
        if (grid <= 256) {
          CK(hipMemsetAsync(cnt, 0, 4, st));
          hipLaunchKernelGGL(k_persist, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, nphase, errs, cnt, nt);
          CK(hipStreamSynchronize(st));
          CK(hipEventRecord(e0, st));
          for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(cnt, 0, 4, st));
            hipLaunchKernelGGL(k_persist, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, nphase, errs, cnt, nt);
          }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          c = ms * 1e3 / (reps * nphase) - spin_us[si];
        }
        // (D) one persistent kernel, XCD-hierarchical barrier per phase
        double d = -1;
        if (grid <= 256) {
          CK(hipMemsetAsync(bar, 0, 1024 * 4, st));
          hipLaunchKernelGGL(k_persist_xcd, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, nphase, errs, bar, nt);
          CK(hipStreamSynchronize(st));
          CK(hipEventRecord(e0, st));
          for (int r = 0; r < reps; ++r) {
            CK(hipMemsetAsync(bar, 0, 1024 * 4, st));
            hipLaunchKernelGGL(k_persist_xcd, dim3(grid), dim3(NTHR), 0, st, buf, w, ticks, nphase, errs, bar, nt);
          }
          CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
          d = ms * 1e3 / (reps * nphase) - spin_us[si];
        }
        int herr = 0; CK(hipMemcpy(&herr, errs, 4, hipMemcpyDeviceToHost));
        printf("%s stores, spin %2d us, %3zu KB / workgroup / phase (%5.1f MB per phase): graph %6.2f  stream %6.2f  persistent + counter barrier %6.2f  persistent + XCD-hierarchical barrier %6.2f   (check errors so far: %d)\n",
               nt == 2 ? "sc1 (wr-thru)" : nt ? "nontemporal  " : "plain        ", spin_us[si], w * 4 / 1024, grid * w * 4 / 1048576.0, a, b, c, d, herr);
      }
  return 0;
}
