timeout -k 10 300 python -m pytest tests/test_training.py -x -q -m gpu -k "window_attention_backward" 2>&1 | tail -3
for v in shipped w8noo2; do
  if [ "$v" = shipped ]; then unset SR_LIB_PATH; else export SR_LIB_PATH="$PWD/studiosr_amd/lib/variants/$v.so"; fi
  for sh in 0 4; do echo "$v: $(python tools/attn_bwd_w8_time.py 64 $sh 2>/dev/null | tail -1)"; done
done
