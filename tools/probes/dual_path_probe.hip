// Probe (round 4): how many operand bytes per clock can a CU pull into its LDS, by path?
// The persistent GEMM's k-loop is bound by its operand staging: 32 KiB per 32-deep k-step of a 256 x 256 tile through LDS-DMA
// (global_load_lds_dwordx4) take ~1.74 k cycles alone, ~19 B/clk/CU, where the MFMAs of the step need 1.02 k.
// Question: is that the LDS-DMA path's own limit -- so that moving HALF of the bytes over the other path (global_load_dwordx4 into
// VGPRs + ds_write_b128) would add bandwidth -- or the shared front end (address units / L1 / L2 port) that both paths use?
// Same access pattern as the GEMM: per step 512 operand rows (256 of A, 256 of W) x 64 B, row stride 1536 B (K = 768 bf16),
// every workgroup on its own rows of A and on shared rows of W, ring of 4 stages, one workgroup of 512 threads per CU.
//   mode 0: all 32 pieces of a step by LDS-DMA                (what gemm256.hip does)
//   mode 1: all 32 pieces by global_load_dwordx4 + ds_write_b128
//   mode 2: A by LDS-DMA, W by global_load + ds_write         (16 + 16)
// and, per mode, with every workgroup on its OWN A rows (half of the bytes come from beyond the XCD's L2, as in a GEMM whose A
// tile is shared by few column tiles) or with only 12 distinct A tiles on the chip (everything an L2 hit after the first pass)
// build: hipcc -O3 --offload-arch=gfx950 -o dual_path_probe dual_path_probe.hip ; run: ./dual_path_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

constexpr int STAGE = 32 * 1024, NS = 4, KB = 768 * 2;  // bytes of a stage, ring depth, operand row bytes

template <int MODE>
__global__ __launch_bounds__(512) void probe(const uint8_t* A, const uint8_t* W, int steps, int ksteps, int a_tiles, unsigned long long* cycles, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // piece j (16 rows x 64 B): wave w owns pieces w, w + 8 of A and of W; lane l -> row 16 j + (l >> 2), 16-B chunk l & 3
  uint32_t a_off[2], w_off[2];
  for (int i = 0; i < 2; ++i) {
    const int row = 16 * (wave + 8 * i) + (lane >> 2);
    a_off[i] = (uint32_t)(((size_t)(blockIdx.x % a_tiles) * 256 + row) * KB + (lane & 3) * 16);  // a_tiles = grid: every workgroup its own A rows
    w_off[i] = (uint32_t)(((size_t)(blockIdx.x % 12) * 256 + row) * KB + (lane & 3) * 16);
  }
  u32x4_t ra[NS][2], rw[NS][2];
  auto issue = [&](int st, int ks) {
    const int kk = (ks % ksteps) * 64;
    uint8_t* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (MODE == 0 || MODE == 2) __builtin_amdgcn_global_load_lds((gbl_void_t*)(A + a_off[i] + kk), (lds_void_t*)(base + (wave + 8 * i) * 1024), 16, 0, 0);
      else ra[st][i] = *reinterpret_cast<const u32x4_t*>(A + a_off[i] + kk);
      if (MODE == 0) __builtin_amdgcn_global_load_lds((gbl_void_t*)(W + w_off[i] + kk), (lds_void_t*)(base + 16384 + (wave + 8 * i) * 1024), 16, 0, 0);
      else rw[st][i] = *reinterpret_cast<const u32x4_t*>(W + w_off[i] + kk);
    }
  };
  auto land = [&](int st) {  // register-staged halves: into LDS once they have arrived (the compiler inserts the vmcnt wait)
    uint8_t* base = smem + st * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (MODE == 1) *reinterpret_cast<u32x4_t*>(base + (wave + 8 * i) * 1024 + lane * 16) = ra[st][i];
      if (MODE != 0) *reinterpret_cast<u32x4_t*>(base + 16384 + (wave + 8 * i) * 1024 + lane * 16) = rw[st][i];
    }
  };
#pragma unroll
  for (int i = 0; i < NS - 1; ++i) issue(i, i);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  uint32_t acc = 0;
#pragma unroll 1
  for (int ks0 = 0; ks0 < steps; ks0 += NS) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int ks = ks0 + u;
      // stage u has landed when at most the NS - 2 younger stages are in flight
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      land(u);
      if (MODE == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      acc += *reinterpret_cast<const uint32_t*>(smem + u * STAGE + ((lane * 64 + ks) & (STAGE - 4)));  // one LDS read per lane: keeps the stage live
      issue((u + NS - 1) % NS, ks + NS - 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int n_wg = 256, steps = 960, ksteps = 24;
  uint8_t *A, *W;
  unsigned long long* cyc;
  uint32_t* sink;
  hipMalloc(&A, (size_t)n_wg * 256 * KB + 4096);
  hipMalloc(&W, (size_t)12 * 256 * KB + 4096);
  hipMalloc(&cyc, n_wg * 8);
  hipMalloc(&sink, 4);
  hipMemset(A, 1, (size_t)n_wg * 256 * KB);
  hipMemset(W, 2, (size_t)12 * 256 * KB);
  const int smem = NS * STAGE;
  hipFuncSetAttribute((const void*)&probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)&probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipFuncSetAttribute((const void*)&probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  const char* names[3] = {"all LDS-DMA", "all global_load + ds_write", "A LDS-DMA, W global_load + ds_write"};
  for (int wgs : {256, 128, 32})
   for (int a_tiles : {wgs, 12}) {
    for (int rep = 0; rep < 2; ++rep)
      for (int mode = 0; mode < 3; ++mode) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(512), smem, 0, A, W, steps, ksteps, a_tiles, cyc, sink);
        if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(512), smem, 0, A, W, steps, ksteps, a_tiles, cyc, sink);
        if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(512), smem, 0, A, W, steps, ksteps, a_tiles, cyc, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(wgs);
        hipMemcpy(h.data(), cyc, wgs * 8, hipMemcpyDeviceToHost);
        double avg = 0;
        for (auto v : h) avg += (double)v;
        avg /= wgs;
        if (rep == 1)
          printf("%3d workgroups  %3d A tiles  %-38s  %7.1f us  %6.0f shader-clock ticks per 32-KiB step (s_memtime domain)  %5.1f GB/s per CU  %6.2f TB/s chip\n", wgs, a_tiles, names[mode],
                 ms * 1e3, avg / steps, 32768.0 * steps / (ms * 1e-3) / 1e9, 32768.0 * steps * wgs / (ms * 1e-3) / 1e12);
      }
  }
  return 0;
}
