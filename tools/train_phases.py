"""Where a prompt-tuning step's time goes (ViT-B/16, 65 classes, P = 16, batch 256): each phase alone, then the step."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth
from tap_clip_amd.models import CLIPWrapper, FullModel
dev = "cuda:0"
cfg = configs.get_config("ViT-B-16")
sd = synth.make_state_dict(cfg, seed=2)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
clip = CLIPWrapper("ViT-B-16", None, dev, precision=prec, state_dict=sd)
names = [f"class{i}" for i in range(65)]
model = FullModel(names, clip, prompt_len=16, class_specific=True).to(dev)
images = synth.make_images(256, cfg, 0).to(dev)
labels = (torch.arange(256) % 65).to(dev)
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
model.train()
def fwd_bwd():
    opt.zero_grad(set_to_none=True)
    model(images, labels)["loss"].backward()
def step():
    opt.zero_grad(set_to_none=True)
    out = model(images, labels)
    out["loss"].backward()
    opt.step()
def fwd_only():
    with torch.no_grad(): model(images, labels)
def fwd_grad():
    model(images, labels)
def img_only():
    with torch.no_grad(): clip._vision.encode_image(images, normalize=True)
def text_only():
    with torch.no_grad(): model.text_features()
def text_grad():
    f = model.text_features(); return f
def text_grad_bwd():
    f = model.text_features(); f.sum().backward()
print(f"precision {prec}, tied padding run {model._tail_run()}")
print(f"image tower alone                      {timeit(img_only):7.3f} ms")
print(f"text features, no grad                 {timeit(text_only):7.3f} ms")
print(f"forward, no grad (towers overlapped)   {timeit(fwd_only):7.3f} ms")
print(f"forward, grad mode                     {timeit(fwd_grad):7.3f} ms")
print(f"forward + backward                     {timeit(fwd_bwd):7.3f} ms")
print(f"full step (forward, backward, AdamW)   {timeit(step):7.3f} ms")
