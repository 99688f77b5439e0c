"""Guard-zone check: run a tower with its workspace placed in the middle of a poisoned buffer and look for writes outside."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-32"
cfg = configs.get_config(name)
sd = synth.make_state_dict(cfg, seed=2)
G = 64 << 20
def guarded(tower, n_seq, tokens, extra=0):
    need = int(tower.lib.tapclip_tower_workspace_bytes(tower.handle, n_seq, tokens)) + extra
    need = (need + 255) // 256 * 256
    big = torch.full((need + 2 * G,), 0xAB, dtype=torch.uint8, device="cuda")
    tower._ws = big[G:G + need]
    return big, need
def check(big, need, tag):
    torch.cuda.synchronize()
    lo, hi = big[:G], big[G + need:]
    bl, bh = int((lo != 0xAB).sum()), int((hi != 0xAB).sum())
    msg = "clean" if bl + bh == 0 else f"CORRUPT: {bl} bytes below, {bh} bytes above"
    if bl:
        idx = (lo != 0xAB).nonzero().flatten(); msg += f" [below: offsets {int(idx[0]) - G}..{int(idx[-1]) - G}]"
    if bh:
        idx = (hi != 0xAB).nonzero().flatten(); msg += f" [above: +{int(idx[0])}..+{int(idx[-1])}]"
    print(f"{tag}: workspace {need} B: {msg}")
ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
prompts = torch.cat([ctx, tok], 1).cuda()
for tprec in ("bf16x3", "bf16"):
    text = engine.TextTower(cfg, sd, "cuda:0", tprec)
    big, need = guarded(text, prompts.shape[0], prompts.shape[1])
    text.forward(prompts, want_hidden=False, want_mean=True)
    check(big, need, f"text {tprec} capture pass")
    text.forward(prompts, want_heads=True, want_attn_out=True)
    check(big, need, f"text {tprec} full pass")
    h, saved = text.forward_saved(prompts)
    check(big, need, f"text {tprec} forward_saved")
    text.backward_saved(saved, torch.randn_like(h))
    check(big, need, f"text {tprec} backward_saved")
images = synth.make_images(8, cfg, 0).cuda()
for prec in ("bf16x3", "bf16", "fp16", "fp8"):
    tw = engine.VisionTower(cfg, sd, "cuda:0", prec)
    big, need = guarded(tw, 8, cfg.n_tokens)
    tw.encode_image(images)
    check(big, need, f"vision {prec}")
