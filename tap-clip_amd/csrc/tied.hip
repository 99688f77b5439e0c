// Tied padding rows of the prompt sequences: compaction, verification and expansion kernels.
//
// FullModel feeds the text transformer WITHOUT positional embedding and WITHOUT attention mask (reference
// models/model_wrapper.py:58,72), on prompts built as [P context rows | token_embedding(tokenizer(text))] (reference
// models/prompt_learner.py:31-34,62-65).  A tokenised prompt is SOT, a few word ids, EOT and then zeros up to 77: every
// padding position carries the SAME embedding row, and with neither position nor mask in play identical input rows stay
// identical through every block (LayerNorm, the GEMMs and the MLP are row-wise; attention is permutation-equivariant).
// At BASELINE configs[2] that is 68-70 of the 93 rows of every sequence.  So the tower may run on the DISTINCT rows
// only -- Tc = T - run + 1 rows per sequence, the last one standing for `run` identical rows -- provided the attention
// core counts that last key `run` times in every softmax (AttnArgs.last_key_bias = ln(run): exp(s + ln m) = m exp(s)).
// The result is the same function of the input, to fp32 round-off (tests/test_oracle.py holds the identity on the CPU
// oracle, tests/test_gpu_tied.py on the kernels): 3.6x fewer rows in every GEMM and LayerNorm of the text tower.
//
// The claim "the last `run` rows are identical" is the CALLER's; tied_compact_kernel verifies it bitwise on the way and
// raises a device flag, and every expand kernel writes NaN instead of a result while the flag is up -- a wrong claim can
// not produce a silently wrong tensor, and the hot path needs no host synchronisation.
#define TAPCLIP_TU_NO_PK_F32
#include "common.h"
#include "kernels.h"

namespace tapclip {
namespace {

__device__ __forceinline__ bool same_bits(const float4& a, const float4& b) {
  return __float_as_uint(a.x) == __float_as_uint(b.x) && __float_as_uint(a.y) == __float_as_uint(b.y) &&
         __float_as_uint(a.z) == __float_as_uint(b.z) && __float_as_uint(a.w) == __float_as_uint(b.w);
}
__device__ __forceinline__ float4 nan4() {
  const float q = __uint_as_float(0x7fc00000u);
  return make_float4(q, q, q, q);
}

// longest run of trailing rows identical to the sequence's last row, minimum over the sequences: one workgroup per
// sequence, one wave per row (rows walked T-2, T-3, ... in rounds of 4 until a round holds a differing row)
__global__ __launch_bounds__(256) void tail_run_kernel(const float* __restrict__ x, int T, int D, int* run_min) {
  __shared__ int differs[4];
  // (threadIdx / blockIdx are device-library calls that do not inline into a no-packed-fp32 function: the builtins)
  const int tid = (int)__builtin_amdgcn_workitem_id_x();
  const int wave = tid >> 6, lane = tid & 63;
  const float* seq = x + (int64_t)__builtin_amdgcn_workgroup_id_x() * T * D;
  const float* last = seq + (int64_t)(T - 1) * D;
  int run = 1;
  for (int t0 = T - 2; t0 >= 0; t0 -= 4) {
    const int t = t0 - wave;
    bool diff = false;
    if (t >= 0) {
      for (int c = lane; c < D; c += 64) diff |= __float_as_uint(seq[(int64_t)t * D + c]) != __float_as_uint(last[c]);
    }
    diff = __any(diff);
    if (lane == 0) differs[wave] = (t < 0 || diff) ? 1 : 0;
    __syncthreads();
    int add = 0;
    while (add < 4 && !differs[add]) ++add;
    run += add;
    __syncthreads();
    if (add < 4) break;  // (workgroup-uniform: every thread read the same four flags)
  }
  if (tid == 0) atomicMin(run_min, run);
}

// x [n, T, D] -> xc [n, Tc, D] (the first Tc rows of every sequence); rows Tc .. T-1 are compared with row Tc - 1
__global__ __launch_bounds__(256) void tied_compact_kernel(const float4* __restrict__ x, int T, int Tc, int D4, int64_t total,
                                                           float4* __restrict__ xc, int* flag) {
  const int64_t i = (int64_t)__builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();
  if (i >= total) return;
  const int64_t row = i / D4;
  const int c = (int)(i - row * D4);
  const int64_t s = row / T;
  const int t = (int)(row - s * T);
  const float4 v = x[i];
  if (t < Tc) {
    xc[(s * Tc + t) * D4 + c] = v;
  } else if (!same_bits(v, x[(s * T + Tc - 1) * D4 + c])) {
    *flag = 1;
  }
}

// g [n, T, D] -> gc [n, Tc, D]: rows < Tc - 1 copied, row Tc - 1 = sum of rows Tc - 1 .. T - 1 (ascending: deterministic)
__global__ __launch_bounds__(256) void tied_sum_tail_kernel(const float4* __restrict__ g, int T, int Tc, int D4, int64_t total,
                                                            float4* __restrict__ gc) {
  const int64_t i = (int64_t)__builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();
  if (i >= total) return;  // total = n * Tc * D4
  const int64_t row = i / D4;
  const int c = (int)(i - row * D4);
  const int64_t s = row / Tc;
  const int t = (int)(row - s * Tc);
  float4 v = g[(s * T + t) * D4 + c];
  if (t == Tc - 1) {
    for (int u = Tc; u < T; ++u) {
      const float4 w = g[(s * T + u) * D4 + c];
      v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
  }
  gc[i] = v;
}

// hc [n, Tc, D] -> h [n, T, D]: rows >= Tc - 1 replicate the last compact row (zero_tail: rows >= Tc are zero instead --
// the gradient of a tied group lives in its first row)
__global__ __launch_bounds__(256) void tied_expand_rows_kernel(const float4* __restrict__ hc, int T, int Tc, int D4, int64_t total,
                                                               int zero_tail, float4* __restrict__ h, const int* __restrict__ flag) {
  const int64_t i = (int64_t)__builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();
  if (i >= total) return;  // total = n * T * D4
  const int64_t row = i / D4;
  const int c = (int)(i - row * D4);
  const int64_t s = row / T;
  const int t = (int)(row - s * T);
  float4 v;
  if (*flag) v = nan4();
  else if (t >= Tc && zero_tail) v = make_float4(0.f, 0.f, 0.f, 0.f);
  else v = hc[(s * Tc + (t < Tc ? t : Tc - 1)) * D4 + c];
  h[i] = v;
}

// pc [nb, Hm, Tc, Tc] -> out [nb, T, T]: mean over Hm (summed h = 0, 1, ...: torch's .mean(dim=1) order for small H), the
// compact last row / column spread over the rows / columns it stands for (a column's probability mass divided by run)
__global__ __launch_bounds__(256) void tied_expand_map_kernel(const float* __restrict__ pc, int Hm, int T, int Tc, int64_t total,
                                                              float h_f, float run_f, float* __restrict__ out,
                                                              const int* __restrict__ flag) {
  const int64_t i = (int64_t)__builtin_amdgcn_workgroup_id_x() * 256 + __builtin_amdgcn_workitem_id_x();
  if (i >= total) return;  // total = nb * T * T
  const int64_t b = i / ((int64_t)T * T);
  const int rem = (int)(i - b * (int64_t)T * T);
  const int q = rem / T, k = rem - q * T;
  const int cq = q < Tc ? q : Tc - 1, ck = k < Tc ? k : Tc - 1;
  const float* p = pc + (b * Hm * Tc + cq) * (int64_t)Tc + ck;
  float s = 0.f;
  for (int h = 0; h < Hm; ++h) s += p[(int64_t)h * Tc * Tc];
  if (Hm > 1) s /= h_f;        // (divisions, as head_mean_kernel and the softmax's own 1 / sum: the untied path's roundings)
  if (k >= Tc - 1) s /= run_f;
  out[i] = *flag ? __uint_as_float(0x7fc00000u) : s;
}

unsigned blocks_of(int64_t total) { return (unsigned)((total + 255) / 256); }

}  // namespace

hipError_t launch_tail_run(const float* x, int32_t n, int32_t T, int32_t D, int32_t* run_min_dev, hipStream_t s) {
  if (n <= 0 || T <= 0 || D <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(tail_run_kernel, dim3((unsigned)n), dim3(256), 0, s, x, T, D, run_min_dev);
  return hipGetLastError();
}

hipError_t launch_tied_compact(const float* x, int32_t n, int32_t T, int32_t Tc, int32_t D, float* xc, int* flag, hipStream_t s) {
  if (D % 4 != 0 || Tc < 1 || Tc > T) return hipErrorInvalidValue;
  const int64_t total = (int64_t)n * T * (D / 4);
  hipLaunchKernelGGL(tied_compact_kernel, dim3(blocks_of(total)), dim3(256), 0, s, reinterpret_cast<const float4*>(x), T, Tc, D / 4, total,
                     reinterpret_cast<float4*>(xc), flag);
  return hipGetLastError();
}

hipError_t launch_tied_sum_tail(const float* g, int32_t n, int32_t T, int32_t Tc, int32_t D, float* gc, hipStream_t s) {
  if (D % 4 != 0 || Tc < 1 || Tc > T) return hipErrorInvalidValue;
  const int64_t total = (int64_t)n * Tc * (D / 4);
  hipLaunchKernelGGL(tied_sum_tail_kernel, dim3(blocks_of(total)), dim3(256), 0, s, reinterpret_cast<const float4*>(g), T, Tc, D / 4, total,
                     reinterpret_cast<float4*>(gc));
  return hipGetLastError();
}

hipError_t launch_tied_expand_rows(const float* hc, int32_t n, int32_t T, int32_t Tc, int32_t D, int32_t zero_tail, float* h, const int* flag,
                                   hipStream_t s) {
  if (D % 4 != 0 || Tc < 1 || Tc > T) return hipErrorInvalidValue;
  const int64_t total = (int64_t)n * T * (D / 4);
  hipLaunchKernelGGL(tied_expand_rows_kernel, dim3(blocks_of(total)), dim3(256), 0, s, reinterpret_cast<const float4*>(hc), T, Tc, D / 4, total,
                     zero_tail, reinterpret_cast<float4*>(h), flag);
  return hipGetLastError();
}

hipError_t launch_tied_expand_map(const float* pc, int32_t nb, int32_t Hm, int32_t T, int32_t Tc, float* out, const int* flag, hipStream_t s) {
  if (Hm < 1 || Tc < 1 || Tc > T) return hipErrorInvalidValue;
  const int64_t total = (int64_t)nb * T * T;
  hipLaunchKernelGGL(tied_expand_map_kernel, dim3(blocks_of(total)), dim3(256), 0, s, pc, Hm, T, Tc, total, (float)Hm,
                     (float)(T - Tc + 1), out, flag);
  return hipGetLastError();
}

}  // namespace tapclip
TAPCLIP_TU_NO_PK_F32_END
