// PROTOTYPE (round 4, not part of the library): a producer / consumer form of the persistent bf16 GEMM.
//
// Why: the k-step of gemm256.hip takes ~2.2 k cycles where its MFMAs need 1.02 k and the chip can stage the operands in ~1.3 k
// (tools/probes/dual_path_probe.hip: 49-64 GB/s per CU at full chip).  Every one of its 8 waves issues LDS-DMA (60-185 cycles of
// issue per 1-KiB piece inside a loaded phase), reads fragments and computes, in two phases per step with a barrier each.  Here the
// roles are split: 4 LOADER waves (one per SIMD) do nothing but issue the LDS-DMA of the ring, 8 CONSUMER waves (two per SIMD) do
// nothing but read fragments and issue MFMAs; one barrier per k-step.  Three waves per SIMD leave 168 registers per wave, so the
// tile is 256 x 192 (96 accumulator registers per consumer) instead of 256 x 256 (110 instead of 128 FLOP per staged byte).
//   consumer k:  read fragments of stage k -> s_barrier -> 24 MFMAs (they drain while the wave reads stage k + 1)
//   loader   k:  issue the DMA of step k + NS - 1 -> wait until step k + 1 has landed -> s_barrier
// The barrier behind step k tells the loaders that slot k % NS has been read (it is refilled in their next iteration) and the
// consumers that stage k + 1 is in LDS.  The k-steps of successive tiles form one sequence through the ring, as in gemm256.hip.
// Epilogue: bias-seeded accumulators, plain 8-byte bf16 fragment stores (a prototype: the library kernel's LDS-transposed epilogue
// is worth ~20 us on the QKV shape).
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o gemm_pc_bench gemm_pc_bench.hip     run: ./gemm_pc_bench [iters]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

constexpr int BM = 256, BN = 192, BKS = 32, NS = 4;
constexpr int A_BYTES = BM * BKS * 2, W_BYTES = BN * BKS * 2, STAGE = A_BYTES + W_BYTES;  // 16 + 12 KiB
constexpr int PIECES = (BM + BN) / 16;   // 28 one-KiB pieces per stage
constexpr int PPL = PIECES / 4;          // 7 per loader wave
constexpr int MI = 8, NJ = 3;            // consumer wave: 128 x 48 = 8 x 3 MFMA tiles

struct Args {
  const bf16_t* A;
  const bf16_t* W;
  const float* bias;
  bf16_t* out;
  int64_t M;
  int N, K, group_m;
};

__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  const f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

__global__ __launch_bounds__(768) void gemm_pc_kernel(Args g) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_m = (int)(g.M / BM), tiles_n = g.N / BN;
  const int total = tiles_m * tiles_n;
  // XCD-aware static deal of the grouped tile order (gemm256.hip): XCD x owns a contiguous eighth of the ids
  const int xcd = blockIdx.x & 7, bpx = gridDim.x >> 3;
  const int cq = total >> 3, crm = total & 7;
  const int chunk_lo = xcd * cq + (xcd < crm ? xcd : crm);
  const int chunk_hi = chunk_lo + cq + (xcd < crm ? 1 : 0);
  const int first = chunk_lo + (blockIdx.x >> 3);
  const int my_tiles = first < chunk_hi ? (chunk_hi - first + bpx - 1) / bpx : 0;
  const int KS = g.K / BKS;
  const int steps = my_tiles * KS;
  auto tile_origin = [&](int id, int& m0, int& n0) {
    const int GM = g.group_m, per_group = GM * tiles_n;
    const int grp = id / per_group, first_m = grp * GM;
    const int gsize = tiles_m - first_m < GM ? tiles_m - first_m : GM;
    const int in_grp = id - grp * per_group;
    m0 = (first_m + in_grp % gsize) * BM;
    n0 = (in_grp / gsize) * BN;
  };
  if (steps == 0) return;

  if (wave >= 8) {
    // ---------------- loader: pieces l, l + 4, ... of every stage (piece j < 16: A rows 16 j .., else W rows 16 (j - 16) ..)
    const int l = wave - 8;
    const int src_chunk = (lane & 3) ^ (3 * ((lane >> 5) & 1));
    int f_tile = 0, f_ks = 0;
    uint32_t off[PPL];
    auto set_offsets = [&]() {
      int m0, n0;
      tile_origin(first + f_tile * bpx, m0, n0);
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        const int j = l + 4 * i;
        off[i] = j < 16 ? (uint32_t)((((int64_t)m0 + 16 * j + (lane >> 2)) * g.K + src_chunk * 8) * 2)
                        : (uint32_t)((((int64_t)n0 + 16 * (j - 16) + (lane >> 2)) * g.K + src_chunk * 8) * 2);
      }
    };
    set_offsets();
    auto issue = [&](int st) {
      uint8_t* base = smem + st * STAGE;
      const int kk = f_ks * (BKS * 2);
#pragma unroll
      for (int i = 0; i < PPL; ++i) {
        const int j = l + 4 * i;
        const uint8_t* src = reinterpret_cast<const uint8_t*>(j < 16 ? g.A : g.W) + off[i] + kk;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)(base + j * 1024), 16, 0, 0);
      }
      if (++f_ks == KS) {
        f_ks = 0;
        if (++f_tile < my_tiles) set_offsets();
      }
    };
    int issued = 0;
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
      if (issued < steps) issue(issued % NS), ++issued;
    // stage 0 landed: at most the NS - 2 younger stages outstanding
    if (steps >= NS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int k = 0; k < steps; ++k) {
      if (issued < steps) issue(issued % NS), ++issued;
      // stage k + 1 landed
      if (issued - (k + 2) >= NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PPL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---------------- consumer
  const int r = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int frag_off = r * 64 + ((q ^ (3 * ((r >> 3) & 1))) << 4);
  const int a_base = (wm * 128) * 64 + frag_off;
  const int w_base = A_BYTES + (wn * 48) * 64 + frag_off;
  __builtin_amdgcn_s_barrier();  // prologue: stage 0 is in LDS
  int k = 0;
  for (int t = 0; t < my_tiles; ++t) {
    int m0, n0;
    tile_origin(first + t * bpx, m0, n0);
    f32x4_t acc[NJ][MI];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(g.bias + n0 + wn * 48 + j * 16 + 4 * q);
#pragma unroll
      for (int i = 0; i < MI; ++i) acc[j][i] = bv;
    }
    for (int ks = 0; ks < KS; ++ks, ++k) {
      const uint8_t* base = smem + (k % NS) * STAGE;
      bf16x8_t wf[NJ], af[MI];
#pragma unroll
      for (int j = 0; j < NJ; ++j) wf[j] = *reinterpret_cast<const bf16x8_t*>(base + w_base + j * 1024);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(base + a_base + i * 1024);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[j][i], 0, 0, 0);
    }
    // epilogue: lane holds D[n = 4 q + e][m = r]
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int64_t m = (int64_t)m0 + wm * 128 + i * 16 + r;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4_t v = acc[j][i];
        typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
        const u32x2_t ph = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        __builtin_nontemporal_store(ph, reinterpret_cast<u32x2_t*>(g.out + m * g.N + n0 + wn * 48 + j * 16 + 4 * q));
      }
    }
  }
}

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

static uint16_t f2bf_host(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf2f_host(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  const int64_t M = 50432;
  struct Shape { const char* name; int N, K; };
  const Shape shapes[] = {{"qkv  N2304 K768 ", 2304, 768}, {"out  N768  K768 ", 768, 768}, {"fc   N3072 K768 ", 3072, 768}, {"proj N768  K3072", 768, 3072}};
  const int smem = NS * STAGE;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem));
  for (const Shape& sh : shapes) {
    std::vector<uint16_t> hA((size_t)M * sh.K), hW((size_t)sh.N * sh.K);
    std::vector<float> hb(sh.N);
    uint64_t s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (float)((s >> 40) & 0xFFFFFF) / 8388608.0f - 1.0f; };
    for (auto& v : hA) v = f2bf_host(rnd());
    for (auto& v : hW) v = f2bf_host(rnd() * 0.05f);
    for (auto& v : hb) v = rnd() * 0.1f;
    bf16_t *A, *W, *out;
    float* bias;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&W, hW.size() * 2)); CK(hipMalloc(&out, (size_t)M * sh.N * 2)); CK(hipMalloc(&bias, sh.N * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(W, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
    Args g{A, W, bias, out, M, sh.N, sh.K, 6};
    for (int gm : {6, 3, 8}) {
      g.group_m = gm;
      hipLaunchKernelGGL(gemm_pc_kernel, dim3(256), dim3(768), smem, 0, g);
      CK(hipDeviceSynchronize());
      if (gm == 6) {  // sampled check against a host double product
        std::vector<uint16_t> ho((size_t)M * sh.N);
        CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
        double worst = 0;
        for (int t = 0; t < 400; ++t) {
          const int64_t m = (int64_t)((t * 7919ull + (t % 3) * 50431ull) % M);
          const int n = (int)((t * 104729ull) % sh.N);
          double ref = hb[n];
          for (int kk = 0; kk < sh.K; ++kk) ref += (double)bf2f_host(hA[m * sh.K + kk]) * (double)bf2f_host(hW[(size_t)n * sh.K + kk]);
          const double got = bf2f_host(ho[m * sh.N + n]);
          worst = fmax(worst, fabs(got - ref) / (fabs(ref) + 1.0));
        }
        printf("%s  sampled max rel err %.2e %s\n", sh.name, worst, worst < 1e-2 ? "ok" : "WRONG");
      }
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      std::vector<float> t;
      for (int it = 0; it < iters; ++it) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(gemm_pc_kernel, dim3(256), dim3(768), smem, 0, g);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms * 1e3f);
      }
      std::sort(t.begin(), t.end());
      const double fl = 2.0 * M * sh.N * sh.K;
      printf("%s group_m %d: median %7.1f us  %7.1f TFLOP/s   min %7.1f us\n", sh.name, gm, t[t.size() / 2], fl / (t[t.size() / 2] * 1e-6) / 1e12, t[0]);
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(out)); CK(hipFree(bias));
  }
  return 0;
}
