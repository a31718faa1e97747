#!/bin/bash
# On the GPU box: rocprofv3 kernel-stats of the default bench under two settings of one environment variable;
# usage: run_kernel_ab.sh VAR A B [kernel-name-regex]
export TMPDIR=/tmp
V=$1; A=$2; B=$3; PAT=${4:-k_}
for val in $A $B; do
  D=gpurun_out/ab_${V}_$val
  rm -rf $D; mkdir -p $D
  env $V=$val true
  export $V=$val
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D/prof -o p --output-format csv -- python3 bench.py --no-cpu-baseline --psnr-rays 0 --steps 10 --warmup 3 > $D/stats.log 2>&1 || { tail -5 $D/stats.log; exit 1; }
  echo "== $V=$val"
  grep -rhE "$PAT" --include=p_kernel_stats.csv $D | cut -d, -f1-4 | cut -c1-120 | head -12
done
