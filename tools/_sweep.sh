timeout -k 10 600 python -m pytest tests/test_training.py -x -q -m gpu -k "weight_gradient_kernel or fused_training_step_against or fused_training_path_matches or ddp or overlap or flat_adam" 2>&1 | tail -2
for m in swinir hat; do for wv in 1 0; do
  echo "$m SR_WG_WIDE=$wv: $(SR_WG_WIDE=$wv timeout -k 10 200 python bench.py --mode train --model $m --skip-cpu 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"])')"
done; done
