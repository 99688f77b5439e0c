cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
python -m pytest tests -m gpu -q -rP > gpurun_out/r4g/gpu_tests.log 2>&1; echo "gpu tests rc=$?"
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r4g/gpu_tests.log | tail -10
grep -h "gradient floor\|context-gradient floor\|\[tied\]\|\[hf\]" gpurun_out/r4g/gpu_tests.log | sort | uniq | head -40
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4g/bench_driver_cmd.json 2> gpurun_out/r4g/bench_driver_cmd.err; echo "bench rc=$?"
python -c "
import json; r=json.load(open('gpurun_out/r4g/bench_driver_cmd.json'))
print(r['value'], r['ms_per_step'], r['kernels']['attention'], r['logits_per_sec'], r['full_forward']['ms_per_forward'], r['train_step']['ms_per_step'])
pm=r['parity_mode']; print(pm['img_per_s'], pm['full_forward_ms'], pm['train_step_ms'], pm['logits_rel_max_vs_cpu_oracle'], pm['roofline']['frac'])
print({p:(v['logits_rel_max_vs_cpu_oracle'], v['logits_err_over_top2_margin']) for p,v in r['precisions'].items()})
"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
