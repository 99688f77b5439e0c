#!/bin/bash
# on the GPU box: the wide-score regression tests against a library built from the commit BEFORE the maxima became
# compiler-visible (a git worktree under _old/, not committed) and against the current one
out=gpurun_out/$1; mkdir -p $out
(cd _old && timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide_scores" 2>&1 | grep -v "^$" | tail -25) > $out/old_lib_wide.log 2>&1
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "wide_scores" -s 2>&1 | grep "parity\|passed\|failed\|rror" > $out/new_lib_wide.log 2>&1
cat $out/old_lib_wide.log; echo ======; cat $out/new_lib_wide.log
