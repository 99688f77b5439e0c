"""The reproducer that cracked the "x24 race" (it was not a race: csrc/common.h TAPCLIP_TU_NO_PK_F32).  A 2-block image
tower (TAPCLIP_X24=1) on a side stream beside the bf16x3 text tower, stopped after block 0's c_fc (TAPCLIP_DEBUG_STOP=4;
a TAPCLIP_DEBUG_STOP=5 run first saves the first LayerNorm's output): which bytes of the LayerNorm output xn differ
from the solo run?  With the packed-fp32 build: always bytes [1408, 1536) of a row = lanes 48..63 of the third
vector, one component in four, holding beta exactly (BIG_GAMMA=100: exactly 0) -- gamma * xhat had vanished in the
low half of a v_pk_fma_f32 with op_sel:[0,1,0]."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
L = 2
stop = int(os.environ.get("TAPCLIP_DEBUG_STOP", "0"))
base_cfg = configs.get_config("ViT-B-32")
cfg = configs.ClipDims("b32-short", 512, 224, 32, configs.TowerDims(768, L, 12, 3072), base_cfg.text)
sd = synth.make_state_dict(cfg, seed=2)
if os.environ.get("BIG_GAMMA"):
    sd["visual.transformer.resblocks.0.ln_2.weight"] = torch.full_like(sd["visual.transformer.resblocks.0.ln_2.weight"], 2.0 ** float(os.environ["BIG_GAMMA"]))
    sd["visual.transformer.resblocks.0.ln_2.bias"] = torch.zeros_like(sd["visual.transformer.resblocks.0.ln_2.bias"])
images = synth.make_images(8, cfg, 0).cuda()
ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
prompts = torch.cat([ctx, tok], 1).cuda()
text = engine.TextTower(cfg, sd, "cuda:0", "bf16x3")
tw = engine.VisionTower(cfg, sd, "cuda:0", "bf16")
tw.encode_image(images, normalize=True)
torch.cuda.synchronize()
ws_s = tw._ws.clone()
M, D, F = 8 * cfg.n_tokens, cfg.vision.width, cfg.vision.mlp
al = lambda v: (v + 255) // 256 * 256
xn_off, xn_sz = al(M * D * 4), M * D * 2
if stop == 5:
    torch.save(ws_s[xn_off:xn_off + xn_sz].cpu(), "/tmp/xn0.pt")
    print("saved xn0")
    sys.exit(0)
xn0 = torch.load("/tmp/xn0.pt").cuda()
side = torch.cuda.Stream()
for it in range(int(os.environ.get("ITERS", "60"))):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tw.encode_image(images, normalize=True)
    text.forward(prompts, want_hidden=False, want_mean=True)
    text.forward(prompts)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    got = tw._ws[xn_off:xn_off + xn_sz]
    ref = ws_s[xn_off:xn_off + xn_sz]
    d = (got != ref).nonzero().flatten()
    if d.numel() == 0:
        continue
    rows = torch.unique(d // (D * 2))
    for r in rows.tolist():
        sel = d[(d // (D * 2)) == r] - r * D * 2
        lo, hi = int(sel[0]) // 32 * 32, (int(sel[-1]) // 32 + 1) * 32
        seg = slice(r * D * 2 + lo, r * D * 2 + hi)
        stale = bool(torch.equal(got[seg], xn0[seg]))
        if not globals().get("_shown"):
            _shown = True
            g16 = got[seg].view(torch.bfloat16).float(); r16 = ref[seg].view(torch.bfloat16).float()
            gi = got[seg].view(torch.int16).int(); ri = ref[seg].view(torch.int16).int()
            print("  ref :", [round(v, 4) for v in r16.tolist()])
            print("  got :", [round(v, 4) for v in g16.tolist()])
            print("  ulps:", (gi - ri).tolist())
            c0 = lo // 2
            bt = sd["visual.transformer.resblocks.0.ln_2.bias"].float().cuda()[c0:c0 + 64]
            gm = sd["visual.transformer.resblocks.0.ln_2.weight"].float().cuda()[c0:c0 + 64]
            print("  beta[::4] :", [round(v, 4) for v in bt[::4].tolist()])
            print("  got[::4]  :", [v for v in g16[::4].tolist()])
            print("  ref[::4]  :", [v for v in r16[::4].tolist()])
            print("  gamma[::4]:", [round(v, 4) for v in gm[::4].tolist()])
            print("  (ref-beta)/gamma [::4]:", [round(v, 4) for v in ((r16 - bt) / gm)[::4].tolist()])
            print("  (got-beta)/gamma [::4]:", [round(v, 4) for v in ((g16 - bt) / gm)[::4].tolist()])
        print(f"iter {it}: row {r} bytes [{lo},{hi}) #{sel.numel()} differ; equals first-LN output there: {stale}; "
              f"addr%128={(tw._ws.data_ptr() + xn_off + r * D * 2 + lo) % 128}")
