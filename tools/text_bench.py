"""The text side of FullModel.forward alone (BASELINE configs[2]: 65 classes x (16 context + 77) tokens): ms per
text_features() call on an idle GPU -- under rocprofv3 --kernel-trace --stats its kernels' own durations (in the
full forward they share the chip with the image tower's persistent GEMMs, so the bench trace inflates them)."""
import contextlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd  # noqa
from tap_clip_amd import configs, synth
from tap_clip_amd.models import CLIPWrapper, FullModel

cfg = configs.get_config("ViT-B-16")
sd = synth.make_state_dict(cfg, seed=2)
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"  # "fp16": the library default (split-bf16 text tower)
clip = CLIPWrapper("ViT-B-16", None, "cuda", precision=prec, attn_semantics="intended", state_dict=sd)
names = [f"class_{i}" for i in range(65)]
with contextlib.redirect_stdout(sys.stderr):
    model = FullModel(names, clip, prompt_len=16, class_specific=True).eval()
with torch.no_grad():
    for _ in range(3):
        model.text_features()
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 20
    for _ in range(n):
        model.text_features()
    torch.cuda.synchronize()
    print(f"text_features [{prec}]: {(time.perf_counter() - t) / n * 1e3:.3f} ms per call")
