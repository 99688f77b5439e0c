"""The reference scripts' operating point (ViT-B/16, batch 32, 5 classes, P = 5: train.py:29-39) as a training loop, for a kernel trace:
  rocprofv3 --kernel-trace --output-format csv -d DIR -o tr -- python3 tools/train_small_trace.py [steps]
  python3 tools/train_small_trace.py --analyse DIR      (per step: kernels, span, busy, idle; the kernels by total time)"""
import csv, glob, os, sys, time

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    f = [p for p in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)][0]
    rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))), key=lambda r: r[0])
    starts = [i for i, r in enumerate(rows) if "im2col" in r[2]]  # a step starts at its patch gather
    print("kernels", len(rows), "steps", len(starts))
    for a, b in list(zip(starts[:-1], starts[1:]))[-4:]:
        seg = rows[a:b]
        busy = sum(e - s for s, e, _ in seg)
        span = seg[-1][1] - seg[0][0]
        gaps = sorted(seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1))
        print(f"kernels {len(seg)}  span {span / 1e3:.1f} us  busy(sum of durations) {busy / 1e3:.1f} us  median gap {gaps[len(gaps) // 2] / 1e3:.2f} us  "
              f"gaps > 3 us: {sum(g > 3000 for g in gaps)} totalling {sum(g for g in gaps if g > 3000) / 1e3:.1f} us")
    seg = rows[starts[-2]:starts[-1]]
    # time during which at least one kernel runs (the towers overlap on two streams)
    union, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"last step: some kernel running for {union / 1e3:.1f} us of {(seg[-1][1] - seg[0][0]) / 1e3:.1f} us")
    import re
    agg = {}
    for s, e, n in seg:
        m = re.search(r"([A-Za-z_0-9]+)\s*(<[^(]*)?\(", n.replace("(anonymous namespace)::", ""))
        k = (m.group(1) + (m.group(2) or ""))[:70] if m else n[:70]
        a = agg.setdefault(k, [0, 0])
        a[0] += 1; a[1] += e - s
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{t / 1e3:9.1f} us  x{c:<4d} {k}")
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_clip_amd
from tap_clip_amd import configs, synth
from tap_clip_amd.models import CLIPWrapper, FullModel
dev = "cuda:0"
cfg = configs.get_config("ViT-B-16")
sd = synth.make_state_dict(cfg, seed=2)
clip = CLIPWrapper("ViT-B-16", None, dev, precision="fp16", state_dict=sd)
model = FullModel([f"class{i}" for i in range(5)], clip, prompt_len=5, class_specific=True).to(dev)
images = synth.make_images(32, cfg, 0).to(dev)
labels = (torch.arange(32) % 5).to(dev)
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
model.train()
def step():
    opt.zero_grad(set_to_none=True)
    out = model(images, labels)
    out["loss"].backward()
    opt.step()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(5): step()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(n): step()
torch.cuda.synchronize()
print(f"train step {1e3 * (time.perf_counter() - t) / n:.3f} ms (wall, {n} steps)")
