"""Where a prompt-tuning step's time goes at the reference scripts' operating point (ViT-B/16, batch 32, 5 classes, P = 5; train.py:29-39)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth
from tap_clip_amd.models import CLIPWrapper, FullModel
dev = "cuda:0"
cfg = configs.get_config("ViT-B-16")
sd = synth.make_state_dict(cfg, seed=2)
n_cls = int(sys.argv[1]) if len(sys.argv) > 1 else 5
P = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
clip = CLIPWrapper("ViT-B-16", None, dev, precision="fp16", state_dict=sd)
model = FullModel([f"class{i}" for i in range(n_cls)], clip, prompt_len=P, class_specific=True).to(dev)
images = synth.make_images(B, cfg, 0).to(dev)
labels = (torch.arange(B) % n_cls).to(dev)
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
def timeit(fn, n=40):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
model.train()
def step():
    opt.zero_grad(set_to_none=True)
    out = model(images, labels); out["loss"].backward(); opt.step()
def fwd_bwd():
    opt.zero_grad(set_to_none=True)
    model(images, labels)["loss"].backward()
def fwd_grad(): model(images, labels)
def fwd_only():
    with torch.no_grad(): model(images, labels)
def img_only():
    with torch.no_grad(): clip._vision.encode_image(images, normalize=True)
def text_only():
    with torch.no_grad(): model.text_features()
def text_grad(): model.text_features()
def text_fwd_bwd():
    opt.zero_grad(set_to_none=True)
    f = model.text_features()
    (f[0] if isinstance(f, (tuple, list)) else f).sum().backward()
def cpu_only_step():  # host time of one step's enqueue (no sync inside the loop: the queue fills, so this is min(host, device))
    step()
rows = [("image tower", img_only), ("text features, no grad", text_only), ("text features, grad (saved)", text_grad),
        ("forward, no grad", fwd_only), ("forward, grad", fwd_grad), ("forward + backward", fwd_bwd), ("step (+ AdamW)", step)]
try:
    text_fwd_bwd(); rows.insert(3, ("text features + their backward", text_fwd_bwd))
except Exception as e:
    print("text_fwd_bwd unavailable:", type(e).__name__, str(e)[:80])
def host_time(fn, n=20):  # the call's own duration on the host with an EMPTY queue in front of it (it returns before the device is done)
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    return sorted(ts)[len(ts) // 2] * 1e3
print(f"ViT-B/16 fp16, batch {B}, {n_cls} classes, P = {P}")
for name, fn in rows:
    print(f"{name:34s} {timeit(fn):7.3f} ms   host side of one call {host_time(fn):7.3f} ms")
# host-side cost of a step: time to ENQUEUE 40 steps without waiting for the device
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(40): step()
host = (time.perf_counter() - t) / 40 * 1e3
torch.cuda.synchronize()
print(f"{'host enqueue per step (no sync)':34s} {host:7.3f} ms")
