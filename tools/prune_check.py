"""CLS-only last block against the full computation: max difference of the embeddings, and speed."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, engine, synth
name = sys.argv[1] if len(sys.argv) > 1 else "ViT-B-16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cfg = configs.get_config(name)
sd = synth.make_state_dict(cfg, seed=2, text=False)
im = synth.make_images(B, cfg, 0).to("cuda:0")
for prec in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["bf16", "fp16", "bf16x3"]):
    full = engine.VisionTower(cfg, sd, "cuda:0", prec, prune_last_block=False)
    pr = engine.VisionTower(cfg, sd, "cuda:0", prec, prune_last_block=True)
    a = full.encode_image(im, normalize=True); b = pr.encode_image(im, normalize=True)
    b2 = pr.encode_image(im, normalize=True)
    rel = float((a - b).norm() / a.norm()); mx = float((a - b).abs().max() / a.abs().max())
    def t(tw, n=20):
        for _ in range(3): tw.encode_image(im, normalize=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): tw.encode_image(im, normalize=True)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    print(f"{name} B={B} {prec}: pruned vs full rel_l2 {rel:.3e} rel_max {mx:.3e} deterministic {torch.equal(b, b2)} finite {bool(torch.isfinite(b).all())} | full {t(full):.3f} ms  pruned {t(pr):.3f} ms")
    pr.profile(True); pr.profile_read(); pr.encode_image(im); torch.cuda.synchronize(); print("   ", {k: round(v[0]*1e3/max(v[1],1),1) for k, v in pr.profile_read().items()}, {k: v[1] for k, v in pr.profile_read().items()})
    del full, pr
