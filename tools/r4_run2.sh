set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -x -q > gpurun_out/r4b/gpu_tests.log 2>&1; echo "gpu tests rc=$?" | tee -a gpurun_out/r4b/gpu_tests.log
tail -15 gpurun_out/r4b/gpu_tests.log
bash tools/ab_pk.sh > gpurun_out/r4b/ab_pk.log 2>&1
cat gpurun_out/r4b/ab_pk.log
python bench.py > gpurun_out/r4b/bench.json 2> gpurun_out/r4b/bench.err; echo "bench rc=$?"
tail -5 gpurun_out/r4b/bench.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r4b/bench.json'))
for k in ("value","ms_per_step","logits_per_sec","encoder_mfma_frac","headline_meets_tolerance","step_ms"): print(k, r.get(k))
print("roofline", {k:r["roofline"][k] for k in ("achieved","frac","avg_launch_us","traffic")})
print("full_forward", r["full_forward"]["ms_per_forward"], r["full_forward"]["default_path"], r["train_step"]["ms_per_step"])
pm=r.get("parity_mode",{}); print("parity", {k:pm.get(k) for k in ("precision","img_per_s","encoder_mfma_frac","full_forward_ms","train_step_ms","logits_rel_max_vs_cpu_oracle","logits_err_over_top2_margin")}, pm.get("roofline"))
for p,row in r.get("precisions",{}).items(): print(p, {k:row.get(k) for k in ("img_per_s","full_forward_ms","train_step_ms","logits_rel_max_vs_cpu_oracle","logits_err_over_top2_margin")})
print("configs4", r.get("configs4"))
print("cpu", r.get("cpu_baseline",{}).get("value"), r.get("cpu_baseline",{}).get("rows",{}).get("full_forward_collapsed"))
PY
