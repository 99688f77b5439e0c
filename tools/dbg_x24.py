import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = configs.get_config(name)
sd = synth.make_state_dict(cfg, seed=2)
images = synth.make_images(B, cfg, 0).cuda()
ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
prompts = torch.cat([ctx, tok], 1).cuda()
for tprec in ("bf16x3", "bf16"):
    text = engine.TextTower(cfg, sd, "cuda:0", tprec)
    for prec in ("bf16",):
        tw = engine.VisionTower(cfg, sd, "cuda:0", prec)
        base = tw.encode_image(images, normalize=True).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        bad = 0
        worst = 0.0
        for it in range(20):
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                e = tw.encode_image(images, normalize=True)
            text.forward(prompts, want_hidden=False, want_mean=True)      # concurrently, on the main stream
            text.forward(prompts)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            if not torch.equal(e, base):
                bad += 1
                worst = max(worst, float((e - base).abs().max() / base.abs().max()))
        print(f"{name} B={B} image {prec} beside text {tprec}: {bad}/20 runs differ from the solo run, worst rel_max {worst:.3e}",
              "(x24)" if not os.environ.get("TAPCLIP_NO_X24") else "(fp32 x)")
