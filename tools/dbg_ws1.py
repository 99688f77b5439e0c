import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
base_cfg = configs.get_config("ViT-B-32")
cfg = configs.ClipDims("b32-short", 512, 224, 32, configs.TowerDims(768, L, 12, 3072), base_cfg.text)
sd = synth.make_state_dict(cfg, seed=2)
images = synth.make_images(8, cfg, 0).cuda()
ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
prompts = torch.cat([ctx, tok], 1).cuda()
text = engine.TextTower(cfg, sd, "cuda:0", "bf16x3")
tw = engine.VisionTower(cfg, sd, "cuda:0", "bf16")
base = tw.encode_image(images, normalize=True).clone()
torch.cuda.synchronize()
ws_s = tw._ws.clone()
side = torch.cuda.Stream()
M, D, F = 8 * cfg.n_tokens, cfg.vision.width, cfg.vision.mlp
al = lambda v: (v + 255) // 256 * 256
hid = max(M * F, 8 * (cfg.n_tokens - 1) * 3072)
regs = [("x", M * D * 4), ("xn", M * D * 2), ("qkv", M * 3 * D * 2), ("ao", M * D * 2), ("h", hid * 2), ("d", M * D * 2), ("a", M * D * 2), ("x24_hi", M * D * 2), ("x24_lo", M * D)]
for it in range(6):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        e = tw.encode_image(images, normalize=True)
    text.forward(prompts, want_hidden=False, want_mean=True)
    text.forward(prompts)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    d = (tw._ws != ws_s).nonzero().flatten()
    line = f"layers={L} iter {it}: equal {bool(torch.equal(e, base))}; "
    off = 0
    for nme, sz in regs:
        sel = d[(d >= off) & (d < off + sz)] - off
        if sel.numel():
            rb = sz // M
            rows = torch.unique(sel // rb)
            line += f"{nme}:{sel.numel()}B rows[{int(rows[0])}..{int(rows[-1])}]#{rows.numel()} "
        off += al(sz)
    print(line)
