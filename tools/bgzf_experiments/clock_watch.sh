#!/bin/bash
# kernel time of a fresh process next to the shader / memory clocks and power rocm-smi reports while it runs
for i in 1 2 3 4 5 6; do
  python tools/bgzf_experiments/kernel_jitter.py 400 > /tmp/j_$i.log 2>&1 &
  pid=$!
  sleep 14
  s=""
  for k in 1 2 3; do
    s="$s | $(rocm-smi --showclocks --showpower -d 0 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | sed -E 's/.*(sclk|mclk|fclk|Average Graphics Package Power|Current Socket Graphics Package Power)[^:]*: *//' | tr '\n' ' ')"
    sleep 0.5
  done
  wait $pid
  echo "process $i: $(grep median /tmp/j_$i.log | cut -c1-60) $s"
done
