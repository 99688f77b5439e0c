import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
cfg = configs.get_config("ViT-B-32")
sd = synth.make_state_dict(cfg, seed=2)
images = synth.make_images(8, cfg, 0).cuda()
ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
prompts = torch.cat([ctx, tok], 1).cuda()
text = engine.TextTower(cfg, sd, "cuda:0", "bf16x3")
tw = engine.VisionTower(cfg, sd, "cuda:0", "bf16")
base = tw.encode_image(images, normalize=True).clone()
torch.cuda.synchronize()
snap = tw._ws.clone()
print("vision ws", hex(tw._ws.data_ptr()), tw._ws.numel())
text.forward(prompts, want_hidden=False, want_mean=True)
text.forward(prompts)
torch.cuda.synchronize()
print("text ws", hex(text._ws.data_ptr()), text._ws.numel())
d = (tw._ws != snap).nonzero().flatten()
print("vision workspace bytes changed by the text tower running ALONE:", d.numel(), (int(d[0]), int(d[-1])) if d.numel() else "")
# now sequential (no concurrency): vision after text
e = tw.encode_image(images, normalize=True)
torch.cuda.synchronize()
print("vision after text, sequential: equal to solo run:", bool(torch.equal(e, base)))
# concurrent, then diff the workspace against a sequential re-run
side = torch.cuda.Stream()
for it in range(3):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        e = tw.encode_image(images, normalize=True)
    text.forward(prompts, want_hidden=False, want_mean=True)
    text.forward(prompts)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    ws_c = tw._ws.clone()
    e2 = tw.encode_image(images, normalize=True)
    torch.cuda.synchronize()
    ws_s = tw._ws.clone()
    d = (ws_c != ws_s).nonzero().flatten()
    print(f"iter {it}: concurrent result equal: {bool(torch.equal(e, base))}; sequential re-run equal: {bool(torch.equal(e2, base))}; workspace bytes differing: {d.numel()}",
          (int(d[0]), int(d[-1])) if d.numel() else "")
    if d.numel():
        # which carved regions?  (layout of tower.hip carve(): x, xn, qkv, ao, h, d, a, x24_hi, x24_lo; 256-B aligned)
        M, D, F = 8 * cfg.n_tokens, cfg.vision.width, cfg.vision.mlp
        al = lambda v: (v + 255) // 256 * 256
        hid = max(M * F, 8 * (cfg.n_tokens - 1) * 3072)
        regs = [("x", M * D * 4), ("xn", M * D * 2), ("qkv", M * 3 * D * 2), ("ao", M * D * 2), ("h", hid * 2), ("d", M * D * 2), ("a", M * D * 2), ("x24_hi", M * D * 2), ("x24_lo", M * D)]
        off = 0
        for nme, sz in regs:
            n = int(((d >= off) & (d < off + sz)).sum())
            if n: print(f"    region {nme}: {n} bytes differ (rows {int((d[(d >= off)][0] - off)) // (sz // M)}..)")
            off += al(sz)
