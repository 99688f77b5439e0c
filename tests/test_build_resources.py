"""Compile-time guard for the hot kernels: none of the fast-path instantiations may use scratch memory.

hipcc demotes an accumulator array to scratch without any warning when, for example, a nested lambda captures
it by reference; the kernel stays correct and runs ~25x slower (seen once on the bf16 GEMM: 784 B/lane of
scratch, 42 TFLOP/s).  This test recompiles the three hot translation units for gfx950 with
-Rpass-analysis=kernel-resource-usage (no GPU needed) and checks ScratchSize of the instantiations the towers use."""
import os
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tap-clip_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# (source, regex of demangled-ish kernel names that must not spill)
HOT = [
    # the bf16 (non-split) epilogues the towers launch (EPI 5, the GELU-backward epilogue, spills 84 B/lane at
    # 256-wide tiles and is therefore always launched 128-wide: gemm256.hip launch_e)
    ("gemm256.hip", r"gemm256_kernelILi[0-4]ELb0ELi(256|128)ELi4E|gemm256_kernelILi5ELb0ELi128ELi4E"),
    ("gemm_mx8.hip", r"gemm_mx8_kernelILi[046]E"),
    ("attention.hip", r"attn_kernelILi(6|14)ELb0E|attn_flash_kernelILi(4|8)ELi[23]ELb0E"),
    # every geometry of the LDS-DMA kernel: a spill would put scratch traffic into its hand-counted vmcnt queue
    ("attention_long.hip", r"attn_flash2_kernelILi"),
]
# the 13-key-tile attention kernel is compiled for 6 waves per SIMD (three workgroups per CU) and parks 3 dwords
SMALL_SPILL = {"attention.hip": (r"attn_kernelILi13ELb0E", 16)}


# per-file flags of csrc/Makefile (the attention kernels are built without NaNs to honour: their maxima are plain fmaxf)
EXTRA = {"attention.hip": ["-fno-honor-nans", "-DTAPCLIP_TU_NO_NANS"], "attention_long.hip": ["-fno-honor-nans", "-DTAPCLIP_TU_NO_NANS"]}


def _usage(src):
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-c",
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", *EXTRA.get(src, []), "-o", os.devnull, os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = {}
    name = None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            out[name] = int(m.group(1))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_hot_kernels_use_no_scratch():
    with ThreadPoolExecutor(max_workers=4) as ex:
        results = list(ex.map(_usage, [s for s, _ in HOT]))
    mk = open(os.path.join(CSRC, "Makefile")).read()
    for src, flags in EXTRA.items():  # (the table above is the Makefile's)
        assert all(f in mk for f in flags) and src.replace(".hip", ".o") in mk, f"csrc/Makefile no longer builds {src} with {flags}"
    for (src, pat), usage in zip(HOT, results):
        hot = {k: v for k, v in usage.items() if re.search(pat, k)}
        assert hot, f"no kernel of {src} matched {pat}: {sorted(usage)[:5]}"
        spilled = {k: v for k, v in hot.items() if v != 0}
        assert not spilled, f"{src}: scratch in hot kernels {spilled}"
        if src in SMALL_SPILL:
            pat2, limit = SMALL_SPILL[src]
            lean = {k: v for k, v in usage.items() if re.search(pat2, k)}
            assert lean and all(v <= limit for v in lean.values()), f"{src}: {lean}"


def test_no_asm_statement_computes_on_vector_registers():
    """hipcc pads its OWN instructions for the MFMA -> VALU read hazard (gfx950 has no interlock: an 8-pass MFMA's result may be read
    11 wait states after its issue) and does not look at the operands of an asm statement.  An asm `v_max3_f32` chain over score
    registers read stale values four instructions after their MFMA (profiles/r05_flash2_asm_hazard.txt).  So the kernel sources may
    hold VALU instructions in asm statements only where listed here, each with the reason its operands cannot be MFMA results."""
    allowed = {
        # max (m, |a|, |b|) over values the GELU's own VALU ops produced (gemm_mx8.hip: the identity epilogue takes amax3_visible)
        ("common.h", "v_max3_f32"),
    }
    found = set()
    for name in sorted(os.listdir(CSRC)):
        if not name.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(CSRC, name)).read()
        for m in re.finditer(r"\basm\s*(?:volatile)?\s*\(((?:[^()]|\((?:[^()]|\([^()]*\))*\))*)\)", text):
            for op in re.findall(r"\b(v_[a-z0-9_]+)", m.group(1)):
                if not op.startswith(("v_mfma",)):
                    found.add((name, op))
    assert found <= allowed, f"VALU instructions inside asm statements: {sorted(found - allowed)} (see this test's docstring)"
    assert ("common.h", "v_max3_f32") in found, "the scan no longer sees the one statement it should"
