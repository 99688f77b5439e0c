"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the MXFP8 operand format of the fp8 path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(tap-clip_amd/) never does.

What it restates: the OCP Microscaling Formats (MX) v1.0 specification, section 6.3 "Conversion from
vector of scalar floats to MX block": block size 32 along k, shared scale X = 2^(floor(log2(max|v|)) -
emax_elem) stored as E8M0 (byte = exponent + 127), elements v / X converted to FP8 E4M3 (OCP "fn"
encoding, emax_elem = 8, largest normal 448) with round-to-nearest-even and saturation.  The reference
repository has no fp8 path (it runs fp32 PyTorch, models/clip_wrapper.py:47): this format comes from
BASELINE.json configs[4] ("fp8 MFMA"), so results of the fp8 mode are "parity unpinned" against the
reference and are judged (a) bit-exactly against this restatement at the kernel level and (b) with a
loose, stated tolerance against the fp32 oracle end to end.
"""
from __future__ import annotations

import torch

BLOCK = 32


def scale_bytes(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 byte of 2^(floor(log2(amax)) - 8); zero / denormal blocks get byte 0 (2^-127)."""
    bits = amax.to(torch.float32).contiguous().view(torch.int32)
    return ((bits >> 23) - 8).clamp_(min=0).to(torch.uint8)


def quantize(x: torch.Tensor):
    """x [rows, K] fp32 -> (q uint8 [rows, K] e4m3 bit patterns, s uint8 [rows, K/32] e8m0 bytes)."""
    rows, K = x.shape
    assert K % BLOCK == 0
    xb = x.to(torch.float32).reshape(rows, K // BLOCK, BLOCK)
    s = scale_bytes(xb.abs().amax(dim=-1))
    inv = torch.ldexp(torch.ones((), dtype=torch.float32), 127 - s.to(torch.int32))       # 2^(127 - byte)
    y = (xb * inv[..., None]).clamp_(-448.0, 448.0)
    q = y.to(torch.float8_e4m3fn).view(torch.uint8).reshape(rows, K)
    return q, s


def dequantize(q: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    rows, K = q.shape
    v = q.view(torch.float8_e4m3fn).to(torch.float32).reshape(rows, K // BLOCK, BLOCK)
    scale = torch.ldexp(torch.ones((), dtype=torch.float32), s.to(torch.int32) - 127)
    return (v * scale[..., None]).reshape(rows, K)


def fake_quant(x: torch.Tensor) -> torch.Tensor:
    """Round-trip x (last dim = k, a multiple of 32) through MXFP8: what a GEMM operand looks like on the fp8 path."""
    shape = x.shape
    q, s = quantize(x.reshape(-1, shape[-1]))
    return dequantize(q, s).reshape(shape)


def scales_to_kstep_major(s: torch.Tensor, rows_pad: int) -> torch.Tensor:
    """[rows, K/32] -> the kernels' layout [K/64, rows_pad, 2] (include/tapclip.h, tapclip_mx8_quantize)."""
    rows, nb = s.shape
    out = torch.zeros(nb // 2, rows_pad, 2, dtype=torch.uint8)
    out[:, :rows, :] = s.reshape(rows, nb // 2, 2).permute(1, 0, 2)
    return out


def scales_from_kstep_major(t: torch.Tensor, rows: int) -> torch.Tensor:
    ks, rows_pad, _ = t.shape
    return t[:, :rows, :].permute(1, 0, 2).reshape(rows, ks * 2)
