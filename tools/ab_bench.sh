#!/bin/bash
# bench.py under a list of environment settings inside one GPU call:  ENVS="A=1;B=2" bash tools/ab_bench.sh   (two rounds, interleaved)
cd /root/repo
IFS=';' read -ra LIST <<< "${ENVS:-;}"
for r in 1 2; do for e in "${LIST[@]}"; do
  echo "== $e : $(env $e python bench.py --no-side-configs --no-cpu-baseline --no-final-psnr --no-roofline ${BENCH_ARGS:-} | grep -o '"value": [0-9.]*')"
done; done
