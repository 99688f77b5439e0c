#!/usr/bin/env python3
"""Benchmark of the hot path on MI355X.  Contract: `python bench.py --gpus N --steps K --warmup W`
(N > 1: launched by torch.distributed.run, one rank per GPU) prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
  ViT-B/16 image encoder, synthetic 224x224x3 fp32 images, batch 256 per GPU.  Headline precision: "fp16" -- IEEE-half
  operands on the 16-bit MFMA at the bf16 rate (libtapclip_fp16.so), the library's default and the fastest mode whose FullModel
  logits are inside BASELINE.json's 1e-3 of the CPU fp32 oracle (checked live: `headline_meets_tolerance`); the bf16-operand mode
  configs[1] names is 1 % faster and 2e-2 off, and is reported beside it as `bf16_mode`, never as `value`.
One step = `encode_image` of the rank's 256 images (L2-normalised, reference
models/model_wrapper.py:40-41) -> all-gather of the embeddings over RCCL (identity at N = 1) ->
65-class cosine logits (model_wrapper.py:79,83) against text features computed ONCE before the timed
region by the HIP text tower (they do not depend on the images).  value = images embedded per
second by the whole job (weak scaling: 256 images per GPU).  The SECOND metric of BASELINE.json, logits/s, is
SURVEY.md section 8d's: B * n_cls / t(FullModel.forward), both text passes inside the timed region, every row of
every image block computed (top-level `logits_per_sec`, details under `full_forward`).

Extra objects on the same line:
  roofline      dominant kernel = the bf16 MFMA GEMM (QKV / out-proj / c_fc+GELU / c_proj launches):
                algorithmic FLOPs per launch / mean launch duration from HIP events recorded on the
                launch stream, live in this run: a SECOND pass of the same K steps (the ~100 event
                records per step cost ~5 % of the step, so `value` is timed on the clean first pass),
                against the 2.5 PFLOP/s dense bf16 peak (5 PFLOP/s for --precision fp8).
  cpu_baseline  the CPU fp32 oracle (oracle/clip_ref.py + full_model_ref.py, a port: open_clip is absent) on bounded
                samples of the same workloads, rank 0, N = 1 only: the headline row (image tower, batch 32) plus the
                other two rows of BASELINE.md section 3 under "rows"; host core counts under "host".
  sustained     >= 10 s of the same encode step with one HIP event per step: img/s of the whole leg, median step time of its first
                and of its last second (the clock of a loaded chip settles over seconds; the headline is a 1-s sprint).
  batch_sweep   the reference's own operating points (train.py:29-39,75-81: batch 32, 5 classes, P = 5): B in {8 .. 256} x
                {encode img/s, MFMA fraction, FullModel forward ms and prompt-tuning step ms at 5 classes / P = 5 and at
                65 / 16}, library defaults, with the per-kernel table at B = 32.
  precisions    the same step in every precision (bf16, fp16 = IEEE-half image tower + split-bf16 text tower, bf16x3 =
                split-bf16 everywhere, fp8 = MXFP8 block GEMMs), each with its embedding error against bf16x3 and the
                error of its FullModel LOGITS against the CPU oracle; `parity_mode` names the fastest one inside 1e-3.
  full_forward  FullModel.forward at configs[2] (image + text towers, 65 classes, 16 context tokens,
                attention-map write-back on), on every rank (embeddings gathered inside the forward at N > 1): the
                full computation (-> top-level logits_per_sec) and the library's default path beside it.
  ranks         (N > 1) what the first multi-GPU run needs to be read: ranks RCCL saw, each rank's device, per-rank
                min / median / max step time, the all-gather's own duration from HIP events.
  configs4      BASELINE configs[4] at its per-GPU shape (ViT-L/14@336, fp8 MFMA, batch 128): encode img/s, fraction of
                the 5 PFLOP/s MXFP8 peak, GEMM family, prompt-tuning step -- a bounded leg (--no-configs4 skips it).
  input_side    CLIP's eval transform (bicubic resize, crop, normalise) of uint8 photos on the GPU
                (tapclip_preprocess_u8), with the reference's Pillow CPU path on one core beside it.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0   # dense, block-scaled v_mfma_scale_f32_*_f8f6f4 with e4m3 operands (same guide)
HBM_PEAK_GBS = 8000.0


def gemm_flops_per_image(cfg):
    v = cfg.vision
    n, d, f = cfg.n_tokens, v.width, v.mlp
    per_layer = {"gemm_qkv": 2 * n * d * 3 * d, "gemm_out_proj": 2 * n * d * d, "gemm_fc_gelu": 2 * n * d * f,
                 "gemm_proj": 2 * n * f * d}
    return {k: val * v.layers for k, val in per_layer.items()}


def encoder_flops_per_image(cfg, pruned_last_block=False):
    """35.127 GFLOP per ViT-B/16 image (blocks + patch embed + projection).
    pruned_last_block: what the library's default `encode_image` EXECUTES -- the last block computes K and V for every
    token and Q, the attention core, out_proj and the MLP for the CLS row only (the other rows' results are discarded by
    the pooling, in the reference too): 32.68 GFLOP per ViT-B/16 image."""
    v = cfg.vision
    n, d, f, hd = cfg.n_tokens, v.width, v.mlp, 64
    block = 2 * n * d * 3 * d + 2 * n * d * d + 2 * 2 * n * d * f + v.heads * (2 * 2 * n * n * hd)
    patch = 2 * (n - 1) * (3 * cfg.patch * cfg.patch) * d
    total = v.layers * block + patch + 2 * d * cfg.embed_dim
    if pruned_last_block:
        last = 2 * n * d * 2 * d + 2 * d * d + v.heads * (2 * 2 * n * hd) + 2 * d * d + 2 * 2 * d * f
        total += last - block
    return total


def source_sha16():
    """sha256[:16] over the GEMM kernel sources the roofline object is about (what a PMC profile was taken on)"""
    import hashlib

    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "tap-clip_amd", "csrc")
    for f in ("gemm256.hip", "gemm_mx8.hip", "common.h"):
        h.update(open(os.path.join(csrc, f), "rb").read())
    # of kernels.h only what the GEMM kernels see: their argument structs (the rest declares other kernels' launchers)
    text = open(os.path.join(csrc, "kernels.h")).read()
    for name in ("struct GemmArgs {", "struct Mx8GemmArgs {"):
        i = text.index(name)
        h.update(text[i:text.index("};", i) + 2].encode())
    return h.hexdigest()[:16]


def pmc_traffic(args, suffix=""):
    """L2-fabric-side bytes per launch of the GEMM family from the newest committed PMC passes (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE runs of this script, tools/pmc_traffic.py; `suffix` "" = the bf16 headline, "_fp16" = the
    IEEE-half build ...) -> (bytes or None, note).  The passes are of the kernels as they were when they were taken: the
    file names the GEMM sources it measured, and a library built from other sources reports null, not a stale figure."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r??_pmc_traffic_bench{suffix}.json")), reverse=True)
    if not files or (args.batch, args.model) != (256, "ViT-B-16"):
        return None, "no committed PMC pass for this configuration"
    tname = os.path.basename(files[0])
    try:
        tj = json.load(open(files[0]))
        want, have = tj.get("gemm_source_sha16"), source_sha16()
        if want != have:
            return None, (f"profiles/{tname} was measured on GEMM sources {want}, this library is built from {have}: "
                          "traffic withheld until the PMC passes are re-run (tools/collect_profiles.sh)")
        return tj.get("gemm_family_hbm_bytes_per_launch"), (
            f"bytes/launch at the L2 fabric side (FETCH_SIZE x2 + WRITE_SIZE, Infinity-Cache hits included) from profiles/{tname}, "
            "a separate rocprofv3 --pmc run; algorithmic operand+output bytes per launch average 313 MB")
    except Exception as e:  # noqa: BLE001
        return None, f"profiles/{tname} unreadable: {e}"


def cgroup_cpu_quota():
    """CPUs the container may use per its cgroup quota (v2 cpu.max, v1 cpu.cfs_quota_us), or None without a quota."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return max(1, q // per)
    except Exception:
        pass
    return None


def spawn_ranks(n: int) -> int:
    """Run this script as n ranks of one node under torch.distributed.run (rendezvous on 127.0.0.1, a free port) and
    return the launcher's exit code.  Nothing here touches the GPU."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def sustained_leg(step_fn, batch, dev, seconds):
    """`seconds` of back-to-back steps with ONE HIP event per step boundary: the figure a loaded chip sustains once its clock
    has settled, beside the sprint the headline is (the GEMM clock was still falling between a 3-step
    and a 100-step run).  Steps are enqueued in slices so the host never runs more than ~0.25 s ahead of the device."""
    for _ in range(3):
        step_fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    step_fn()
    torch.cuda.synchronize(dev)
    one = max(time.perf_counter() - t0, 1e-4)
    n = max(8, int(math.ceil(seconds / one)))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    sl = max(1, int(0.25 / one))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(n):
        step_fn()
        ev[i + 1].record()
        if (i + 1) % sl == 0 and i + 1 >= 2 * sl:
            ev[i + 1 - sl].synchronize()  # stay at most two slices ahead
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
    total = sum(ms) * 1e-3

    def median_of_window(from_start):
        acc, win = 0.0, []
        for v in (ms if from_start else reversed(ms)):
            win.append(v)
            acc += v
            if acc >= 1000.0:
                break
        return sorted(win)[len(win) // 2]

    first, last = median_of_window(True), median_of_window(False)
    return {"seconds": round(total, 2), "wall_seconds": round(wall, 2), "steps": n, "img_per_s": round(batch * n / total, 1),
            "median_step_ms_first_second": round(first, 4), "median_step_ms_last_second": round(last, 4),
            "last_over_first": round(last / first, 4),
            "what": "back-to-back steps, one HIP event per step boundary; img_per_s over the whole leg (device time between the first and the last event)"}


def batch_sweep_leg(clip, cfg, dev, events, flops_full, peak_tflops):
    """The reference's own operating points: its scripts run batch 32, 5 classes, 5 context tokens (reference train.py:29-39,
    75-81; test_cross_domain.py:30; test_cross_domain2.py:56) -- the headline's batch 256 is BASELINE.json's, not theirs.
    Library defaults throughout (the precision the wrapper was built with, CLS-only last image block, tied padding rows)."""
    import contextlib

    from tap_clip_amd import synth
    from tap_clip_amd.models import FullModel

    t_leg = time.perf_counter()
    vision = clip._vision
    vision.set_prune_last_block(True)
    vision.set_ksplit(True)

    def timed(fn, budget):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        t_ = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        its = max(3, min(200, int(budget / max(time.perf_counter() - t_, 1e-5))))
        t_ = time.perf_counter()
        for _ in range(its):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t_) / its

    models = {}
    for tag, (n_cls, P) in (("c5_p5", (5, 5)), ("c65_p16", (65, 16))):
        with contextlib.redirect_stdout(sys.stderr):
            fm = FullModel([f"class_{i}" for i in range(n_cls)], clip, prompt_len=P, class_specific=True)
        with torch.no_grad():
            ctx = synth.make_prompts(n_cls, P, cfg, seed=1)[0]
            for i in range(n_cls):
                fm.prompt_learner.context_bank[f"class_{i}"].copy_(ctx[i])
        models[tag] = (fm, n_cls)
    all_images = synth.make_images(256, cfg, seed=100).to(dev)
    rows = {}
    per_image = {}
    for B in (8, 32, 64, 128, 256):
        images = all_images[:B].contiguous()
        row = {}
        with torch.no_grad():
            dt = timed(lambda: vision.encode_image(images, normalize=True), 0.25)
        per_image[B] = dt / B
        ex = encoder_flops_per_image(cfg, pruned_last_block=True)
        row["encode"] = {"img_per_s": round(B / dt, 1), "ms": round(1e3 * dt, 3),
                         "encoder_mfma_frac_executed": round(ex * B / dt / (peak_tflops * 1e12), 4),
                         "encoder_mfma_frac_full_flops": round(flops_full * B / dt / (peak_tflops * 1e12), 4)}
        for tag, (fm, n_cls) in models.items():
            fm.eval()
            with torch.no_grad():
                dt_f = timed(lambda: fm(images), 0.2)
            labels = (torch.arange(B, device=dev) % n_cls)
            opt = torch.optim.AdamW(fm.prompt_learner.parameters(), lr=0.0, weight_decay=0.0)
            fm.train()

            def train_step():
                o = fm(images, labels)
                opt.zero_grad(set_to_none=True)
                o["loss"].backward()
                opt.step()

            dt_t = timed(train_step, 0.2)
            fm.eval()
            del opt
            row[tag] = {"full_forward_ms": round(1e3 * dt_f, 3), "logits_per_sec": round(B * n_cls / dt_f, 1), "train_step_ms": round(1e3 * dt_t, 3)}
        if events and B == 32:
            vision.profile(True)
            vision.profile_read()
            torch.cuda.synchronize(dev)
            with torch.no_grad():
                for _ in range(20):
                    vision.encode_image(images, normalize=True)
            torch.cuda.synchronize(dev)
            prof = vision.profile_read()
            vision.profile(False)
            gf = gemm_flops_per_image(cfg)
            row["kernels"] = {k: {"ms_per_step": round(ms / 20, 4), "launches_per_step": n / 20, "avg_us": round(1e3 * ms / n, 2),
                                  **({"tflops": round(gf[k] * B * 20 / (ms * 1e-3) / 1e12, 1)} if k in gf else {})}
                              for k, (ms, n) in prof.items() if n}
        rows[str(B)] = row
    out = {"workload": f"{cfg.name if hasattr(cfg, 'name') else 'ViT-B-16'}, precision {vision.precision} (text tower {clip._text.precision}), library defaults; "
                       "c5_p5 = the reference scripts' 5 classes x 5 context tokens, c65_p16 = BASELINE configs[2]",
           "rows": rows,
           "per_image_rate_vs_b256": {str(B): round(per_image[256] / per_image[B], 3) for B in per_image},
           "leg_seconds": round(time.perf_counter() - t_leg, 1)}
    for fm, _ in models.values():
        del fm
    return out


def configs4_leg(dev, events):
    """BASELINE.json configs[4] at its per-GPU shape, inside the driver's own run: ViT-L/14@336, fp8 (MXFP8) MFMA image
    tower, batch 128, and the prompt-tuning step of the reference's training loop (reference train.py:95-105; image encoder
    frozen, prompt gradients only: 65 classes x 16 context tokens here, the text tower in bf16 beside the fp8 image tower).
    A bounded leg: weights drawn on the device (synth.make_state_dict_device: nothing here is compared with a golden --
    the fp8 parity tests are tests/test_gpu_mx8.py); every timed region is at least a second long, and a 3-s sustained run of
    the encode step follows the first one.  Every row of every block is computed for `img_per_s`;
    the train step runs the library defaults."""
    import contextlib

    import tap_clip_amd  # noqa: F401
    from tap_clip_amd import configs, engine, synth
    from tap_clip_amd.models import CLIPWrapper, FullModel

    t_leg = time.perf_counter()
    name, batch, n_cls, P = "ViT-L-14-336", 128, 65, 16
    cfg = configs.get_config(name)
    sd = synth.make_state_dict_device(cfg, seed=2, device=dev)
    cw = CLIPWrapper(name, None, str(dev), precision="fp8", attn_semantics="intended", state_dict=sd)
    del sd
    tw = cw._vision
    images = torch.randn(batch, 3, cfg.image_size, cfg.image_size, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    flops = encoder_flops_per_image(cfg)

    def timed(fn, its):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(dev)
        t_ = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        its = max(its, int(math.ceil(1.0 / max(time.perf_counter() - t_, 1e-4))))  # >= 1 s per region
        t_ = time.perf_counter()
        for _ in range(its):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t_) / its

    out = {"workload": f"BASELINE configs[4] per GPU: {name}, fp8 (MXFP8 e4m3 + e8m0/32) MFMA image tower, batch {batch}, synthetic "
                       f"{cfg.image_size}x{cfg.image_size}x3; weights seeded on the device", "dtype": "fp8 (MXFP8) image tower, bf16 text tower"}
    with torch.no_grad():
        tw.set_prune_last_block(False)
        dt = timed(lambda: tw.encode_image(images, normalize=True), 8)
        out.update({"img_per_s": round(batch / dt, 1), "ms_per_step": round(1e3 * dt, 3),
                    "encoder_gflop_per_image": round(flops / 1e9, 2),
                    "encoder_mfma_frac": round(flops * batch / dt / (PEAK_FP8_TFLOPS * 1e12), 4), "peak_TFLOPs": PEAK_FP8_TFLOPS})
        out["sustained"] = sustained_leg(lambda: tw.encode_image(images, normalize=True), batch, dev, 3.0)
        if events:
            tw.profile(True)
            tw.profile_read()
            torch.cuda.synchronize(dev)
            for _ in range(4):
                tw.encode_image(images, normalize=True)
            torch.cuda.synchronize(dev)
            prof = tw.profile_read()
            tw.profile(False)
            gf = gemm_flops_per_image(cfg)
            g_ms = sum(prof[k][0] for k in gf)
            g_n = sum(prof[k][1] for k in gf)
            g_fl = sum(gf[k] * batch * 4 for k in gf)
            if g_ms > 0:
                out["gemm_family"] = {"achieved": round(g_fl / (g_ms * 1e-3) / 1e12, 1), "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                                      "frac": round(g_fl / (g_ms * 1e-3) / 1e12 / PEAK_FP8_TFLOPS, 4), "avg_launch_us": round(1e3 * g_ms / g_n, 2),
                                      "per_family": {k: {"avg_us": round(1e3 * prof[k][0] / max(prof[k][1], 1), 1),
                                                         "tflops": round(gf[k] * batch * 4 / (prof[k][0] * 1e-3) / 1e12, 1)} for k in gf if prof[k][1]}}
            out["kernels_ms_per_step"] = {k: round(v[0] / 4, 3) for k, v in prof.items() if v[1]}
        tw.set_prune_last_block(True)
        dt_d = timed(lambda: tw.encode_image(images, normalize=True), 8)
        out["img_per_s_default_path"] = round(batch / dt_d, 1)
    names = [f"class_{i}" for i in range(n_cls)]
    with contextlib.redirect_stdout(sys.stderr):
        fm = FullModel(names, cw, prompt_len=P, class_specific=True)
    labels = torch.arange(batch, device=dev) % n_cls
    opt = torch.optim.AdamW(fm.prompt_learner.parameters(), lr=2e-3, weight_decay=0.01)
    fm.train()

    def train_step():
        o = fm(images, labels)
        opt.zero_grad(set_to_none=True)
        o["loss"].backward()
        opt.step()

    dt_t = timed(train_step, 5)
    fm.eval()
    with torch.no_grad():
        dt_f = timed(lambda: fm(images), 5)
    out["full_forward_ms"] = round(1e3 * dt_f, 3)
    out["train_step_ms"] = round(1e3 * dt_t, 3)
    out["train_step"] = f"reference train.py:95-105 loop body at {n_cls} classes x {P} context tokens: forward, CE, backward to the prompts, AdamW"
    out["leg_seconds"] = round(time.perf_counter() - t_leg, 1)
    del fm, opt, cw, tw, images
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 1.1 s of timed region at batch 256 (20 steps were 0.23 s of an
    ap.add_argument("--warmup", type=int, default=10)   # 18-s run: too short for an outside observer to sample)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--model", default="ViT-B-16")
    ap.add_argument("--classes", type=int, default=65)
    ap.add_argument("--prompt-len", type=int, default=16)
    ap.add_argument("--precision", default="fp16", choices=["bf16", "bf16x3", "fp8", "fp16"],
                    help="fp16 = IEEE-half image tower + split-bf16 text tower: the library default, the headline (inside 1e-3); "
                         "bf16 = bf16 operands (2e-2 on logits); bf16x3 = split-bf16 everywhere (3 MFMA products)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-forward", action="store_true")
    ap.add_argument("--no-input-side", action="store_true", help="skip the GPU preprocess measurement")
    ap.add_argument("--no-precisions", action="store_true", help="skip the bf16 / fp16 / fp8 comparison table")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--no-configs4", action="store_true", help="skip the ViT-L/14@336 fp8 batch-128 leg (BASELINE configs[4])")
    ap.add_argument("--no-sustained", action="store_true", help="skip the 10-s sustained encode leg")
    ap.add_argument("--sustained-seconds", type=float, default=10.0)
    ap.add_argument("--no-batch-sweep", action="store_true", help="skip the B = 8 .. 256 sweep at the reference's operating points")
    ap.add_argument("--dump-logits", default=None, help="rank 0 saves the last step's [global_batch, classes] logits here (.npy): tests")
    args = ap.parse_args()

    t_start = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as the driver calls it for N = 1: start the N ranks here, one process per GPU,
        # BEFORE this process makes any GPU call (a process that has initialised the GPU must not be replaced, and
        # need not hold a context beside its ranks); relay rank 0's JSON line (the children share stdout) and the
        # launcher's exit code.
        raise SystemExit(spawn_ranks(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    # (TAPCLIP_DIST_BACKEND=gloo rehearses N > 1 on a box with fewer GPUs than ranks: ranks share devices)
    backend = os.environ.get("TAPCLIP_DIST_BACKEND", "nccl")
    if world > 1 and backend == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit(f"--gpus {world} with RCCL needs {world} GPUs, this box shows {torch.cuda.device_count()} "
                         "(TAPCLIP_DIST_BACKEND=gloo rehearses the launch with ranks sharing devices; its number is not a scaling result)")
    dev = torch.device("cuda", local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank)
    torch.cuda.set_device(dev)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import tap_clip_amd
    from tap_clip_amd import configs, engine, synth
    from tap_clip_amd.dist import all_gather_rows
    from tap_clip_amd.models import CLIPWrapper, FullModel

    cfg = configs.get_config(args.model)
    sd = synth.make_state_dict(cfg, seed=2)
    clip = CLIPWrapper(args.model, None, str(dev), precision=args.precision, attn_semantics="intended", state_dict=sd)
    names = [f"class_{i}" for i in range(args.classes)]
    import contextlib

    with contextlib.redirect_stdout(sys.stderr):  # PromptLearner prints a config line like the reference (prompt_learner.py:21)
        model = FullModel(names, clip, prompt_len=args.prompt_len, class_specific=True, gather_images=world > 1).eval()
    with torch.no_grad():  # seeded context (the reference draws torch.randn; any N(0,1) sample is the same workload)
        ctx = synth.make_prompts(args.classes, args.prompt_len, cfg, seed=1)[0]
        for i, c in enumerate(names):
            model.prompt_learner.context_bank[c].copy_(ctx[i])
    images = synth.make_images(args.batch, cfg, seed=100 + rank).to(dev)  # resident in HBM before timing
    vision = clip._vision
    # The headline times the FULL computation -- every row of every block, the 35.127 GFLOP per image of SURVEY.md
    # section 8d -- so the library's default CLS-only last block (include/tapclip.h TAPCLIP_FLAG_PRUNE_LAST_BLOCK: same
    # embeddings, 7 % fewer FLOPs) is switched off here and reported separately below as `default_path`.
    pruning = True
    if pruning:
        vision.set_prune_last_block(False)
    scale = float(model.logit_scale.detach().exp())
    with torch.no_grad():
        text_feat = model.text_features()  # once, outside the timed region (image independent)

    def step():
        emb = vision.encode_image(images, normalize=True)
        emb = all_gather_rows(emb)
        return engine.logits(emb, text_feat, scale)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if world > 1:
        # The FIRST collective under a deadline: a rank whose peer never arrives (a dead process, a GPU that is not there, an
        # RCCL bootstrap that hangs) must exit non-zero naming itself, not sit in a stream wait until the driver kills the
        # whole run.  The step is enqueued, an event behind it polled; past the deadline this process reports its rank and
        # device on stderr and ends itself (os._exit: no clean-up that could block on the same collective; the launcher then
        # tears the other ranks down).  torch's own watchdog covers asynchronous RCCL errors of its process group
        # (TORCH_NCCL_ASYNC_ERROR_HANDLING); the C-ABI exchange step has tapclip_comm_check for the same purpose.
        deadline = float(os.environ.get("TAPCLIP_FIRST_COLLECTIVE_TIMEOUT", "180"))
        step()
        first = torch.cuda.Event()
        first.record()
        t_w = time.perf_counter()
        while not first.query():
            if time.perf_counter() - t_w > deadline:
                print(f"[bench] rank {rank} (device {dev}, pid {os.getpid()}): the first all-gather did not complete within {deadline:.0f} s "
                      f"-- {world} ranks expected, backend {backend}", file=sys.stderr, flush=True)
                os._exit(3)
            time.sleep(0.002)
        print(f"[bench] rank {rank}: first collective done after {time.perf_counter() - t_w:.2f}s", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        step()
    events = not args.no_kernel_events
    # one HIP event per step boundary on the launch stream (a ~1 us host call every ~11 ms): per-step durations of THIS
    # rank for the `ranks` table -- the first multi-GPU run must show which rank, if any, is the slow one
    step_ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    sync_all()
    t0 = time.perf_counter()
    step_ev[0].record()
    for i in range(args.steps):
        out = step()
        step_ev[i + 1].record()
    sync_all()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(args.steps))
    # Second pass of the same K steps with a HIP event pair around every kernel family (recorded on the launch
    # stream by the library, tapclip_profile_*): the per-kernel durations of the roofline object.  It is a pass
    # of its own because ~100 event records per step cost ~5 % of the step; `value` is the clean pass above.
    prof = None
    elapsed_events = None
    gather_us = None
    if events:
        vision.profile(True)
        vision.profile_read()
        ag = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)] if world > 1 else []
        sync_all()
        t1 = time.perf_counter()
        for i in range(args.steps):
            if world > 1:  # the exchange step's own duration on this rank's stream (the wait for the slowest rank included)
                emb = vision.encode_image(images, normalize=True)
                ag[i][0].record()
                emb = all_gather_rows(emb)
                ag[i][1].record()
                engine.logits(emb, text_feat, scale)
            else:
                step()
        sync_all()
        elapsed_events = time.perf_counter() - t1
        prof = vision.profile_read()
        vision.profile(False)
        if ag:
            g_us = sorted(1e3 * a.elapsed_time(b) for a, b in ag)
            gather_us = {"min": round(g_us[0], 1), "median": round(g_us[len(g_us) // 2], 1), "max": round(g_us[-1], 1)}
    # the library's default path (CLS-only last block) on the same step, same number of steps
    elapsed_default = None
    if pruning:
        vision.set_prune_last_block(True)
        for _ in range(max(2, args.warmup // 2)):
            step()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        elapsed_default = time.perf_counter() - t1
    sustained = None
    if world == 1 and not args.no_sustained:
        # the same step as the headline (every row of every block), for >= 10 s
        vision.set_prune_last_block(False)
        sustained = sustained_leg(step, args.batch, dev, args.sustained_seconds)
        vision.set_prune_last_block(True)
        print(f"[bench] sustained leg done at {time.perf_counter() - t_start:.1f}s", file=sys.stderr, flush=True)
    if world > 1:
        t = torch.tensor([elapsed, elapsed_default or 0.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0].item())
        elapsed_default = float(t[1].item()) if pruning else None
    ranks_info = None
    if world > 1:
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "device": str(dev), "name": props.name, "uuid": str(getattr(props, "uuid", "")),
                "pci": "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0), getattr(props, "pci_device_id", 0)),
                "step_ms": {"min": round(step_ms[0], 3), "median": round(step_ms[len(step_ms) // 2], 3), "max": round(step_ms[-1], 3)},
                "allgather_us": gather_us}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
    assert out.shape == (args.batch * world, args.classes) and bool(torch.isfinite(out).all())
    if args.dump_logits and rank == 0:
        import numpy as np

        np.save(args.dump_logits, out.cpu().numpy())

    total_images = args.batch * world * args.steps
    value = total_images / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    enc_flops = encoder_flops_per_image(cfg)
    result = {
        "metric": "image_embeddings_per_sec", "value": round(value, 1), "unit": "img/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: " if (args.model, args.batch) == ("ViT-B-16", 256) else "") +
                               f"{args.model} image encoder, synthetic {cfg.image_size}x{cfg.image_size}x3, "
                               f"batch {args.batch}/GPU, + all-gather of embeddings and {args.classes}-class logits",
                   "batch_per_gpu": args.batch, "global_batch": args.batch * world, "tokens_per_image": cfg.n_tokens,
                   "classes": args.classes, "parallelism": f"dp{world}", "weights": "seeded random (no checkpoint offline)",
                   "collective": ("none (one rank)" if world == 1 else
                                  f"RCCL all_gather_into_tensor of the [{args.batch}, {cfg.embed_dim}] fp32 embeddings over xGMI" if backend == "nccl" else
                                  f"{backend} all_gather through host memory -- REHEARSAL, {world} ranks on "
                                  f"{torch.cuda.device_count()} GPU(s): not a scaling measurement")},
        # B * n_cls / t(FullModel.forward) -- set by the full-forward leg below (null when it is skipped)
        "logits_per_sec": None,
        "step_ms": {"min": round(step_ms[0], 3), "median": round(step_ms[len(step_ms) // 2], 3), "max": round(step_ms[-1], 3),
                    "what": "this rank's per-step durations from one HIP event per step boundary (rank 0 here; every rank under `ranks`)"},
        # against the dense peak of the headline precision's MFMA (bf16 / IEEE half: 2.5 PFLOP/s; fp8: the block-scaled 5 PFLOP/s)
        "encoder_mfma_frac": round(enc_flops * args.batch * world * args.steps / elapsed /
                                   ((PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS) * 1e12 * world), 4),
        "encoder_gflop_per_image": round(enc_flops / 1e9, 3),
    }
    if sustained is not None:
        sustained["encoder_mfma_frac"] = round(enc_flops * sustained["img_per_s"] /
                                               ((PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS) * 1e12), 4)
        result["sustained"] = sustained
    if world > 1:
        result["ranks"] = {"rccl_ranks_seen": dist.get_world_size(), "backend": dist.get_backend(), "visible_devices": torch.cuda.device_count(),
                           "distinct_devices": len({(r["uuid"], r["pci"], r["device"]) for r in ranks_info}), "per_rank": ranks_info}
    if elapsed_default is not None:
        ex = encoder_flops_per_image(cfg, pruned_last_block=True)
        result["default_path"] = {
            "what": "the library's default encode_image (CLIPWrapper / VisionTower as shipped): in the LAST block K and V are computed for every "
                    "token, Q / attention / out_proj / MLP for the CLS row only -- the reference pools token 0 and discards the other rows of that "
                    "block (models/clip_wrapper.py:46-47), so the embeddings are the same (tests/test_gpu_configs.py::test_pruned_last_block_*); "
                    "the headline `value` above does NOT use it (it times every row of every block)",
            "img_per_s": round(total_images / elapsed_default, 1), "ms_per_step": round(1e3 * elapsed_default / args.steps, 4),
            "executed_gflop_per_image": round(ex / 1e9, 3),
            "encoder_mfma_frac_executed": round(ex * total_images / elapsed_default /
                                                ((PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS) * 1e12 * world), 4),
        }

    def kernel_table(prof_, steps_):
        """per-family table from the library's HIP-event sums + the GEMM family's FLOPs, time and launch count"""
        gf = gemm_flops_per_image(cfg)
        kern_ = {}
        g_ms_ = g_fl_ = 0.0
        g_n_ = 0
        for k, (ms, n) in prof_.items():
            if n == 0:
                continue
            e = {"ms_per_step": round(ms / steps_, 4), "launches_per_step": n / steps_, "avg_us": round(1e3 * ms / n, 2)}
            if k in gf:
                fl = gf[k] * args.batch * steps_
                e["tflops"] = round(fl / (ms * 1e-3) / 1e12, 1)
                g_ms_ += ms
                g_fl_ += fl
                g_n_ += n
            kern_[k] = e
        return kern_, g_ms_, g_fl_, g_n_

    if prof is not None:
        kern, g_ms, g_fl, g_n = kernel_table(prof, args.steps)
        achieved = g_fl / (g_ms * 1e-3) / 1e12
        traffic, traffic_note = pmc_traffic(args, "" if args.precision == "bf16" else "_" + args.precision)
        fp8 = args.precision == "fp8"
        peak = PEAK_FP8_TFLOPS if fp8 else PEAK_BF16_TFLOPS
        result["roofline"] = {
            "kernel": ("gemm_mx8_kernel<EPI> (persistent MXFP8 MFMA 32x32x64 GEMM, 256x256 tiles, 4-stage LDS-DMA ring with e8m0 scales): the QKV + out_proj + c_fc/GELU + c_proj launches"
                       if fp8 else
                       "gemm256_kernel<EPI,false,256,4> (persistent 16-bit MFMA 16x16x32 GEMM -- " + ("IEEE-half operands, libtapclip_fp16.so" if args.precision == "fp16" else "bf16 operands")
                       + " -- 256x256 tiles, 4-stage LDS-DMA ring): the QKV + out_proj + c_fc/GELU + c_proj launches"),
            "bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_note": traffic_note,
            "flops_per_launch": round(g_fl / g_n), "avg_launch_us": round(1e3 * g_ms / g_n, 2), "launches_per_step": g_n / args.steps,
            "measured_in": f"a second pass of the same {args.steps} steps with per-kernel HIP events ({1e3 * elapsed_events / args.steps:.3f} ms/step with the events in the stream)",
        }
        ln = prof.get("layernorm")
        if ln and ln[1]:
            rows = args.batch * cfg.n_tokens
            rd = rows * cfg.vision.width
            # fp32 residual (the default): ln_pre (fp32 -> fp32: 8 B/element), LN1 of block 0 (fp32 -> bf16: 6),
            # LN2 of blocks 0..L-2 (read x + one branch, write the 16-bit output only: 8), LN2 of the last block (also
            # writes x: 12), LN1 of blocks 1..L-1 (read x + two branches, write x + output: 14).
            # 24-bit residual planes (default; TAPCLIP_X24=0 for fp32; layernorm.hip XF = 2): ln_pre + LN1 of block 0 in one kernel
            # (read fp32, write planes + output: 9), LN2 7 / 10, LN1 12.  (fp8: 16-bit stream, 1.03-B outputs; not modelled.)
            L = cfg.vision.layers
            x24 = args.precision in ("bf16", "fp16") and cfg.vision.width % 256 == 0 and os.environ.get("TAPCLIP_X24", "1") != "0"
            per_elt = (9 + (L - 1) * 7 + 10 + (L - 1) * 12) if x24 else (8 + 6 + (L - 1) * 8 + 12 + (L - 1) * 14)
            ln_bytes = rd * per_elt
            result["layernorm_hbm"] = {"achieved_GBps": round(ln_bytes * args.steps / (ln[0] * 1e-3) / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
                                       "bytes_per_step": ln_bytes, "residual_stream": "24-bit planes" if x24 else "fp32"}
        result["kernels"] = kern

    def timed_all_ranks(fn, n_it):
        """seconds per call of fn: three timed segments of n_it / 3 calls each, bracketed like the headline (barrier + synchronize on
        both sides, MAX over ranks per segment), the MEDIAN segment reported -- one segment of a shared box hit by something else
        (seen once: 13.2 ms where its neighbours gave 11.8) does not become the figure"""
        for _ in range(2):
            fn()
        seg = max(3, (n_it + 2) // 3)
        times = []
        for _ in range(3):
            sync_all()
            t_ = time.perf_counter()
            for _ in range(seg):
                fn()
            sync_all()
            d_ = (time.perf_counter() - t_) / seg
            if world > 1:
                tt = torch.tensor([d_], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d_ = float(tt.item())
            times.append(d_)
        return sorted(times)[1]

    if not args.no_full_forward:
        # FullModel.forward on EVERY rank (at N > 1 the embeddings are gathered inside it, on the image tower's stream, beside
        # the replicated text tower).  First with every row of every image block computed -- the figure the top-level
        # logits_per_sec is made of -- then the library's default path (CLS-only last image block, same logits).
        n_it = max(10, args.steps // 4)  # (at the driver's --steps 20 five forwards were a noisy sample: 12.3 .. 12.7 ms)
        gb = args.batch * world

        def fwd():
            with torch.no_grad():
                return model(images)["logits"]

        vision.set_prune_last_block(False)
        dt_full = timed_all_ranks(fwd, n_it)
        vision.set_prune_last_block(True)
        dt = timed_all_ranks(fwd, n_it)
        # for the record: the same forward with EVERY row of every prompt through the text tower (the repeated padding rows not
        # merged) -- same logits, the figure round 3 reported
        tied_before = model.tie_padding
        model.tie_padding = False
        dt_untied = timed_all_ranks(fwd, max(3, n_it // 3))
        model.tie_padding = tied_before
        is_cfg2 = (args.model, args.batch, args.classes, args.prompt_len) == ("ViT-B-16", 256, 65, 16)
        result["logits_per_sec"] = round(gb * args.classes / dt_full, 1)
        result["full_forward"] = {
            "workload": ("BASELINE configs[2]: " if is_cfg2 else "") + f"{args.model} image+text towers, {args.classes} classes, P={args.prompt_len}, "
                        f"attention-map write-back on, batch {args.batch}/GPU; FullModel.forward with BOTH text passes inside the timed region"
                        + (f", embeddings all-gathered inside the forward over {world} ranks (max over ranks)" if world > 1 else ""),
            "ms_per_forward": round(1e3 * dt_full, 3), "logits_per_sec": round(gb * args.classes / dt_full, 1),
            "images_per_sec": round(gb / dt_full, 1), "cls_only_last_block": False,
            "text_rows_per_sequence": {"input": args.prompt_len + cfg.ctx, "computed": args.prompt_len + cfg.ctx - model._tail_run() + 1,
                                       "why": "the padding rows of a prompt are one embedding row repeated and the reference adds no position / mask "
                                              "(models/model_wrapper.py:58,72): the text tower runs on the distinct rows (include/tapclip.h, tied padding rows)",
                                       "ms_per_forward_every_row": round(1e3 * dt_untied, 3)},
            "default_path": {"ms_per_forward": round(1e3 * dt, 3), "logits_per_sec": round(gb * args.classes / dt, 1),
                             "images_per_sec": round(gb / dt, 1), "cls_only_last_block": True}}
        # prompt-tuning step (reference train.py:99-105): forward + loss + backward to context_bank + AdamW, library defaults
        labels = synth.make_labels(args.batch, args.classes, seed=3 + rank).to(dev)
        opt = torch.optim.AdamW(model.prompt_learner.parameters(), lr=2e-3, weight_decay=0.01)
        model.train()

        def train_step():
            out_t = model(images, labels)
            opt.zero_grad(set_to_none=True)
            out_t["loss"].backward()
            opt.step()

        dt_t = timed_all_ranks(train_step, n_it)
        model.eval()
        del opt
        with torch.no_grad():  # back to the SEEDED prompts: every check below is made at that point, whatever --steps was
            for i, c in enumerate(names):
                model.prompt_learner.context_bank[c].copy_(ctx[i])
        result["train_step"] = {"workload": f"prompt-tuning step ({args.model}): FullModel forward + CE + backward to {args.classes} x "
                                            f"[{args.prompt_len},{cfg.text.width}] context tokens + AdamW, "
                                            "batch %d/GPU (image tower forward only: frozen; library defaults: CLS-only last image block)" % args.batch,
                                "ms_per_step": round(1e3 * dt_t, 3), "images_per_sec": round(gb / dt_t, 1)}
    if rank == 0 and world == 1 and not args.no_batch_sweep and not args.no_full_forward:
        result["batch_sweep"] = batch_sweep_leg(clip, cfg, dev, events, enc_flops, PEAK_FP8_TFLOPS if args.precision == "fp8" else PEAK_BF16_TFLOPS)
        vision.set_prune_last_block(True)
        print(f"[bench] batch sweep done at {time.perf_counter() - t_start:.1f}s", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_input_side:
        # Input side: CLIP's eval transform of decoded uint8 photos on the GPU, bit-identical to the
        # Pillow + torchvision transform the reference runs per sample in its loader workers (dataset.py:29-35).
        g = torch.Generator(device="cpu").manual_seed(5)
        photo_hw = (375, 500)
        photos = [torch.randint(0, 256, (*photo_hw, 3), dtype=torch.uint8, generator=g).to(dev) for _ in range(args.batch)]
        size = cfg.image_size
        for _ in range(2):
            pre = engine.preprocess_u8(photos, size=size, device=dev)
        torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n_it = max(3, args.steps // 4)
        t1 = time.perf_counter()
        for _ in range(n_it):
            pre = engine.preprocess_u8(photos, size=size, device=dev)
        torch.cuda.synchronize(dev)
        dt_host = (time.perf_counter() - t1) / n_it
        # the two kernels alone: the C-ABI call on prepared descriptors (the python wrapper above spends ~1.5 us per
        # image building them)
        import ctypes as C
        from tap_clip_amd import _lib

        desc = torch.tensor([(p.data_ptr() - photos[0].data_ptr(), photo_hw[0], photo_hw[1], i * photo_hw[0] * size * 3)
                             for i, p in enumerate(photos)], dtype=torch.int64).to(dev)
        ws = torch.empty(args.batch * photo_hw[0] * size * 3, dtype=torch.uint8, device=dev)
        ms6 = (C.c_float * 6)(*engine.CLIP_MEAN, *engine.CLIP_STD)
        lib, st = _lib.load(), torch.cuda.current_stream(dev).cuda_stream
        call = lambda: _lib.check(lib.tapclip_preprocess_u8(photos[0].data_ptr(), desc.data_ptr(), args.batch, size, ms6,
                                                            ws.data_ptr(), pre.data_ptr(), st))
        call()
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(n_it):
            call()
        e1.record()
        torch.cuda.synchronize(dev)
        dt_dev = e0.elapsed_time(e1) * 1e-3 / n_it
        with torch.no_grad():
            vision.encode_image(pre, normalize=True)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(n_it):
                vision.encode_image(engine.preprocess_u8(photos, size=size, device=dev), normalize=True)
            torch.cuda.synchronize(dev)
            dt_both = (time.perf_counter() - t1) / n_it
        # algorithmic bytes: read the photo once, write + read the horizontally resampled rows, write fp32 CHW
        pp_bytes = args.batch * (photo_hw[0] * photo_hw[1] * 3 + 2 * photo_hw[0] * size * 3 + size * size * 3 * 4)
        entry = {"workload": f"{args.batch} uint8 RGB photos {photo_hw[0]}x{photo_hw[1]} resident in HBM -> [{args.batch},3,{size},{size}] fp32: "
                             "Resize(bicubic) + CenterCrop + ToTensor + Normalize (tapclip_preprocess_u8; images_per_sec through the python wrapper)",
                 "images_per_sec": round(args.batch / dt_host, 1), "ms_per_batch": round(1e3 * dt_host, 3),
                 "ms_per_batch_kernels": round(1e3 * dt_dev, 3),
                 "hbm": {"achieved_GBps": round(pp_bytes / dt_dev / 1e9, 1), "peak_GBps": HBM_PEAK_GBS, "bytes_per_batch": pp_bytes},
                 "preprocess_plus_encode_images_per_sec": round(args.batch / dt_both, 1)}
        try:  # the reference's own CPU path for the same photos, one core (Pillow is what its transform calls)
            from PIL import Image
            from tap_clip_amd.models.clip_wrapper import _make_preprocess

            cpu_pre = _make_preprocess(size)
            pil = [Image.fromarray(p.cpu().numpy()) for p in photos[:16]]
            torch.set_num_threads(1)
            t1 = time.perf_counter()
            outs = [cpu_pre(im) for im in pil]
            dt_cpu = (time.perf_counter() - t1) / len(pil)
            entry["cpu_pillow_images_per_sec_one_core"] = round(1.0 / dt_cpu, 1)
            entry["bit_identical_to_cpu_path"] = bool(torch.equal(torch.stack(outs), pre[:16].cpu()))
        except ImportError:
            pass
        result["input_side"] = entry
        del photos, pre
    # ---- CPU baseline: the fp32 oracle -- a port, open_clip is absent -- on bounded samples of
    # the same workloads, on this box's host cores, rank 0 at N = 1 only.  Its collapsed full forward doubles as the
    # reference for the logits error of every GPU precision below.
    oracle_logits = None
    n_ref = min(32, args.batch)
    print(f"[bench] GPU legs done at {time.perf_counter() - t_start:.1f}s", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import clip_ref, full_model_ref  # the CPU port, timed as the baseline / used as the checker only

        ncpu_os = os.cpu_count() or 1
        ncpu_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else ncpu_os
        ncpu_quota = cgroup_cpu_quota()
        # default: the cores this process may really run on -- its affinity mask, cut to the container's CPU quota when
        # one is visible, and to 16: the pool gives a 1-GPU job a 16-core share of the host, and more threads than the
        # share only thrash (TAPCLIP_CPU_THREADS overrides).  The real counts are printed beside the used one ("host").
        ncpu = min(ncpu_aff, ncpu_quota or ncpu_aff, 16)
        ncpu = int(os.environ.get("TAPCLIP_CPU_THREADS", str(ncpu)))
        torch.set_num_threads(ncpu)
        sample = images[:n_ref].cpu()
        sd_v = {k: v for k, v in sd.items() if k.startswith("visual.")}
        ocfg = clip_ref.CONFIGS[args.model]
        with torch.no_grad():
            clip_ref.encode_image(sample[:8], sd_v, ocfg)  # warm-up
            times = []
            for _ in range(3):
                t1 = time.perf_counter()
                clip_ref.encode_image(sample, sd_v, ocfg)
                times.append(time.perf_counter() - t1)
            med = sorted(times)[1]
            # row 3 of BASELINE.md section 3: the full forward (collapsed text side), batch 32
            prompts = model.prompt_learner().detach().cpu() if model is not None else None
            t_fulls = []
            for _ in range(3):
                t1 = time.perf_counter()
                o_full = full_model_ref.forward_collapsed(sample, prompts, args.prompt_len, sd, ocfg, attn_semantics="intended")
                t_fulls.append(time.perf_counter() - t1)
            t_full = sorted(t_fulls)[1]
            oracle_logits = o_full["logits"]
            # row 1: BASELINE configs[0] as the reference runs it -- ViT-B/32, batch 8, 10 classes, P = 5, the literal
            # per-sample attribution loop (reference models/model_wrapper.py:47-83): 10 x (8 + 1) text passes
            c1 = configs.get_config("ViT-B-32")
            sd1 = synth.make_state_dict(c1, seed=2)
            ctx1, tok1 = synth.make_prompts(10, 5, c1, seed=1)
            im1 = synth.make_images(8, c1, seed=0)
            t1 = time.perf_counter()
            full_model_ref.forward_literal(im1, torch.cat([ctx1, tok1], 1), 5, sd1, clip_ref.CONFIGS["ViT-B-32"], attn_semantics="intended")
            t_lit = time.perf_counter() - t1
            del sd1
        cpu_model = ""
        try:
            cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
        except Exception:
            pass
        result["cpu_baseline"] = {"value": round(n_ref / med, 2), "unit": "img/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": f"oracle/clip_ref.py encode_image fp32, {args.model}, batch {n_ref} (of the {args.batch}), median of 3 after 1 warm-up",
                                  "host": {"os_cpu_count": ncpu_os, "affinity_cpus": ncpu_aff, "cgroup_cpu_quota": ncpu_quota, "threads_used": torch.get_num_threads(), "cpu_model": cpu_model},
                                  "rows": {
                                      "cfg1_literal_loop": {"workload": "BASELINE configs[0]: ViT-B-32, batch 8, 10 classes, P=5 (T=82), literal per-sample attribution loop (oracle/full_model_ref.forward_literal), 1 run",
                                                            "seconds": round(t_lit, 3), "logits_per_sec": round(80 / t_lit, 2), "images_per_sec": round(8 / t_lit, 3)},
                                      "image_tower": {"workload": f"{args.model} image tower fp32, batch {n_ref}", "images_per_sec": round(n_ref / med, 2)},
                                      "full_forward_collapsed": {"workload": f"{args.model} full forward, {args.classes} classes, P={args.prompt_len}, collapsed text path (oracle/full_model_ref.forward_collapsed), batch {n_ref}, median of 3",
                                                                 "seconds": round(t_full, 3), "logits_per_sec": round(n_ref * args.classes / t_full, 1),
                                                                 "images_per_sec": round(n_ref / t_full, 2)}}}

    print(f"[bench] CPU baseline done at {time.perf_counter() - t_start:.1f}s", file=sys.stderr, flush=True)
    if rank == 0 and world == 1 and not args.no_precisions:
        # Every precision of the towers on the same step (fewer steps): img/s, the live embedding error against the
        # split-bf16 parity mode, and the LOGITS error of the whole FullModel forward (image + text towers + attribution)
        # against the CPU fp32 oracle on the first 32 images -- the quantity BASELINE.json bounds at 1e-3.
        ctx_bank = [model.prompt_learner.context_bank[c].detach().clone() for c in names]
        del model
        with torch.no_grad():
            ref_emb = engine.VisionTower(cfg, sd, dev, "bf16x3").encode_image(images, normalize=True)
            table = {}
            n_it = max(3, min(10, args.steps))
            for prec in ("bf16", "fp16", "bf16x3", "fp8"):
                if prec == "fp8" and (cfg.vision.width % 256 or cfg.vision.mlp % 256):
                    continue
                own = prec == args.precision
                cw = clip if own else CLIPWrapper(args.model, None, str(dev), precision=prec, attn_semantics="intended", state_dict=sd)
                tw = cw._vision
                tw.set_prune_last_block(False)  # img_per_s / kernels: the full computation, like the headline
                its = n_it if prec != "bf16x3" else 3
                for _ in range(2):
                    e = tw.encode_image(images, normalize=True)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(its):
                    e = tw.encode_image(images, normalize=True)
                    engine.logits(e, text_feat, scale)
                torch.cuda.synchronize(dev)
                dt_p = (time.perf_counter() - t1) / its
                row = {"img_per_s": round(args.batch / dt_p, 1), "ms_per_step": round(1e3 * dt_p, 3),
                       "text_tower": cw._text.precision,
                       # against the dense bf16 peak for every mode but fp8 (IEEE-half MFMAs run at the bf16 rate)
                       "encoder_mfma_frac": round(enc_flops * args.batch / dt_p / ((PEAK_FP8_TFLOPS if prec == "fp8" else PEAK_BF16_TFLOPS) * 1e12), 4),
                       "embedding_rel_l2_vs_bf16x3": float("%.3e" % float((e - ref_emb).norm() / ref_emb.norm()))}
                if events and prec == "fp16":
                    # the mode that satisfies BOTH halves of north_star (speed and 1e-3) gets its own per-kernel table:
                    # a pass of the same steps with the library's HIP events in the stream, as for the headline
                    tw.profile(True)
                    tw.profile_read()
                    torch.cuda.synchronize(dev)
                    for _ in range(its):
                        engine.logits(tw.encode_image(images, normalize=True), text_feat, scale)
                    torch.cuda.synchronize(dev)
                    kern_p, g_ms_p, g_fl_p, g_n_p = kernel_table(tw.profile_read(), its)
                    tw.profile(False)
                    row["kernels"] = kern_p
                    if g_ms_p > 0:
                        row["gemm_family"] = {"achieved": round(g_fl_p / (g_ms_p * 1e-3) / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                              "frac": round(g_fl_p / (g_ms_p * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                              "avg_launch_us": round(1e3 * g_ms_p / g_n_p, 2), "flops_per_launch": round(g_fl_p / g_n_p)}
                tw.set_prune_last_block(True)  # the shipped default: its speed, and the errors below are ITS errors
                for _ in range(2):
                    tw.encode_image(images, normalize=True)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(its):
                    engine.logits(tw.encode_image(images, normalize=True), text_feat, scale)
                torch.cuda.synchronize(dev)
                row["img_per_s_default_path"] = round(args.batch * its / (time.perf_counter() - t1), 1)
                with contextlib.redirect_stdout(sys.stderr):
                    fm = FullModel(names, cw, prompt_len=args.prompt_len, class_specific=True).eval()
                for c, t in zip(names, ctx_bank):
                    fm.prompt_learner.context_bank[c].copy_(t)
                if not args.no_full_forward:
                    # the whole FullModel forward and the prompt-tuning step in THIS precision (the `full_forward` /
                    # `train_step` objects above are the headline precision's; "fp16", the library default, runs its
                    # text tower -- forward and backward -- in split-bf16: three MFMA products per GEMM)
                    n_ff = 3 if prec == "bf16x3" else 5
                    for _ in range(2):
                        fm(images)
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    for _ in range(n_ff):
                        fm(images)
                    torch.cuda.synchronize(dev)
                    row["full_forward_ms"] = round(1e3 * (time.perf_counter() - t1) / n_ff, 3)
                    with torch.enable_grad():
                        labels_p = synth.make_labels(args.batch, args.classes).to(dev)
                        opt_p = torch.optim.AdamW(fm.prompt_learner.parameters(), lr=0.0, weight_decay=0.0)  # (lr 0: the
                        fm.train()                                                  # logits check below sees the same prompts)

                        def step_p():
                            out_p = fm(images, labels_p)
                            opt_p.zero_grad(set_to_none=True)
                            out_p["loss"].backward()
                            opt_p.step()

                        for _ in range(2):
                            step_p()
                        torch.cuda.synchronize(dev)
                        t1 = time.perf_counter()
                        for _ in range(n_ff):
                            step_p()
                        torch.cuda.synchronize(dev)
                        row["train_step_ms"] = round(1e3 * (time.perf_counter() - t1) / n_ff, 3)
                        fm.eval()
                        del opt_p
                if oracle_logits is not None:
                    lg = fm(images[:n_ref])["logits"].cpu()
                    err = (lg - oracle_logits).abs()
                    row["logits_rel_max_vs_cpu_oracle"] = float("%.3e" % float(err.max() / oracle_logits.abs().max()))
                    row["logits_rel_l2_vs_cpu_oracle"] = float("%.3e" % float(err.norm() / oracle_logits.norm()))
                    row["meets_1e-3"] = bool(err.max() / oracle_logits.abs().max() < 1e-3)
                    # how the error compares with what decides a prediction: per image, its largest logit error over the
                    # oracle's top-1 / top-2 margin (>= 0.5 can flip the arg-max; on this random-weight model the margins
                    # themselves are small, which is why arg-max agreement says little here)
                    top2 = oracle_logits.topk(2, dim=1).values
                    ratio = err.max(dim=1).values / (top2[:, 0] - top2[:, 1]).clamp_min(1e-12)
                    row["logits_err_over_top2_margin"] = {"median": float("%.3e" % float(ratio.median())), "max": float("%.3e" % float(ratio.max()))}
                del fm
                table[prec] = row
                print(f"[bench] precision {prec} done at {time.perf_counter() - t_start:.1f}s", file=sys.stderr, flush=True)
                if not own:
                    del cw, tw
                    torch.cuda.empty_cache()
        result["precisions"] = table
        if args.precision in table and "meets_1e-3" in table[args.precision]:
            # does the HEADLINE number's own mode hold BASELINE.json's 1e-3 on the FullModel logits?  (the default headline, fp16,
            # does; --precision bf16 does not: that is a throughput figure, reported as `bf16_mode` in the default run.)
            result["headline_meets_tolerance"] = table[args.precision]["meets_1e-3"]
            result["headline_logits_rel_max_vs_cpu_oracle"] = table[args.precision]["logits_rel_max_vs_cpu_oracle"]
        ok = [k for k, v in table.items() if v.get("meets_1e-3")]
        if ok:
            best = max(ok, key=lambda k: table[k]["img_per_s"])
            gfam = table[best].get("gemm_family")
            roof = None
            if gfam:
                tr, tr_note = pmc_traffic(args, "" if best == "bf16" else "_" + best)
                roof = {"kernel": "the same persistent 256x256 MFMA GEMM family as the headline, in this mode's operand type "
                                  "(IEEE half on v_mfma_f32_16x16x32_f16 for fp16): QKV + out_proj + c_fc/GELU + c_proj launches",
                        "bound": "mfma", "achieved": gfam["achieved"], "peak": gfam["peak"], "unit": "TFLOP/s", "frac": gfam["frac"],
                        "avg_launch_us": gfam["avg_launch_us"], "flops_per_launch": gfam.get("flops_per_launch"),
                        "traffic": tr, "traffic_note": tr_note}
            result["parity_mode"] = {"precision": best, "img_per_s": table[best]["img_per_s"],
                                     "encoder_mfma_frac": table[best]["encoder_mfma_frac"],
                                     "img_per_s_default_path": table[best].get("img_per_s_default_path"),
                                     "full_forward_ms": table[best].get("full_forward_ms"), "train_step_ms": table[best].get("train_step_ms"),
                                     "logits_per_sec": (round(args.batch * args.classes / (table[best]["full_forward_ms"] * 1e-3), 1)
                                                        if table[best].get("full_forward_ms") else None),
                                     "roofline": roof, "kernels": table[best].pop("kernels", None),
                                     "logits_rel_max_vs_cpu_oracle": table[best]["logits_rel_max_vs_cpu_oracle"],
                                     "logits_rel_l2_vs_cpu_oracle": table[best]["logits_rel_l2_vs_cpu_oracle"],
                                     "logits_err_over_top2_margin": table[best].get("logits_err_over_top2_margin"),
                                     "note": "fastest precision whose FullModel logits are within BASELINE.json's 1e-3 of the CPU fp32 oracle "
                                             f"(first {n_ref} images, {args.classes} classes, the seeded prompts)"
                                             + (": it IS the headline precision" if best == args.precision else f"; the headline `value` is --precision {args.precision}")
                                             + ".  full_forward_ms / train_step_ms: library defaults (CLS-only last image block)"}
        if args.precision != "bf16" and "bf16" in table:
            b16 = table["bf16"]
            result["bf16_mode"] = {"what": "the bf16-operand mode BASELINE configs[1] names, for the record: OUTSIDE BASELINE.json's 1e-3 on the "
                                           "FullModel logits (8 significand bits per operand), so never the headline",
                                   "img_per_s": b16["img_per_s"], "encoder_mfma_frac": b16["encoder_mfma_frac"],
                                   "img_per_s_default_path": b16.get("img_per_s_default_path"), "full_forward_ms": b16.get("full_forward_ms"),
                                   "train_step_ms": b16.get("train_step_ms"), "logits_rel_max_vs_cpu_oracle": b16.get("logits_rel_max_vs_cpu_oracle"),
                                   "meets_1e-3": b16.get("meets_1e-3")}
    if rank == 0 and world == 1 and not args.no_configs4 and (args.model, args.precision) == ("ViT-B-16", "fp16"):
        result["configs4"] = configs4_leg(dev, events)
        print(f"[bench] configs4 done at {time.perf_counter() - t_start:.1f}s", file=sys.stderr, flush=True)
    if world > 1:
        dist.barrier()

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
