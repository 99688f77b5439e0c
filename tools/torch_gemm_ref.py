"""Library bf16 GEMM (torch.matmul -> hipBLASLt/rocBLAS) on the image tower's shapes, for comparison with
tools/gemm_bench.  Measurement aid only; not part of the product path."""
import torch

M = 50432
shapes = [("qkv", 2304, 768), ("out_proj", 768, 768), ("fc", 3072, 768), ("proj", 768, 3072)]
for name, N, K in shapes:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for _ in range(10):
        torch.nn.functional.linear(a, w, bias)
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.nn.functional.linear(a, w, bias)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{name:9s} N{N} K{K}: median {med:8.1f} us  {2 * M * N * K / med / 1e6:7.1f} TFLOP/s  min {ts[0]:8.1f} us", flush=True)
