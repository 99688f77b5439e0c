"""Experiment: the batch of 256 as two half-batches on two streams (two tower handles), against one stream."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth, engine
cfg = configs.get_config("ViT-B-16")
sd = synth.make_state_dict(cfg, seed=2)
B = 256
parts = int(sys.argv[1]) if len(sys.argv) > 1 else 2
images = synth.make_images(B, cfg, 0).cuda()
tw = engine.VisionTower(cfg, sd, "cuda:0", "bf16")
def run_single(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): tw.encode_image(images, normalize=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
run_single(3)
print(f"one stream, batch {B}: {run_single(20):.3f} ms")
tws = [tw] + [engine.VisionTower(cfg, sd, "cuda:0", "bf16") for _ in range(parts - 1)]
streams = [torch.cuda.Stream() for _ in range(parts)]
chunks = list(images.chunk(parts))
def run_multi(n, stagger=False):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        for tw_i, s, x in zip(tws, streams, chunks):
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                tw_i.encode_image(x, normalize=True)
        for s in streams: torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
run_multi(3)
print(f"{parts} streams x batch {B // parts}: {run_multi(20):.3f} ms")
# sequential halves on one stream, for reference
def run_seq(n):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        for tw_i, x in zip(tws, chunks): tw_i.encode_image(x, normalize=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3
run_seq(3)
print(f"{parts} x batch {B // parts} back to back on one stream: {run_seq(20):.3f} ms")
