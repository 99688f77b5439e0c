import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
cfg = configs.get_config("ViT-B-32")
sd = synth.make_state_dict(cfg, seed=2)
images = synth.make_images(8, cfg, 0).cuda()
ctx, tok = synth.make_prompts(10, 5, cfg, seed=1)
prompts = torch.cat([ctx, tok], 1).cuda()
text3 = engine.TextTower(cfg, sd, "cuda:0", "bf16x3")
A = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
Bm = torch.randn(1 << 24, device="cuda")
x3 = torch.randn(820, 512, device="cuda"); g = torch.ones(512, device="cuda"); b = torch.zeros(512, device="cuda")
w3 = torch.randn(1536, 512, device="cuda")
def nb_matmul():
    for _ in range(20): (A @ A)
def nb_elem():
    for _ in range(20): Bm.mul_(1.0001)
def nb_text(): 
    text3.forward(prompts, want_hidden=False, want_mean=True); text3.forward(prompts)
def nb_gemm3():
    for _ in range(12): engine.gemm(x3, w3, None, "bf16x3")
def nb_gemm1():
    for _ in range(12): engine.gemm(x3, w3, None, "bf16")
def nb_ln():
    for _ in range(40): engine.layernorm(x3, g, b)
tw = engine.VisionTower(cfg, sd, "cuda:0", "bf16")
base = tw.encode_image(images, normalize=True).clone()
torch.cuda.synchronize()
side = torch.cuda.Stream()
for name, nb in (("torch matmul", nb_matmul),):
    bad = 0
    for it in range(20):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            e = tw.encode_image(images, normalize=True)
        nb()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        bad += int(not torch.equal(e, base))
    print(f"image bf16 (x24) beside {name}: {bad}/20 differ")
