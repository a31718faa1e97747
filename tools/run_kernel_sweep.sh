#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the default bench for several values of one environment variable;
# usage: run_kernel_sweep.sh VAR "v1 v2 ..." kernel-name-prefix...
export TMPDIR=/tmp
V=$1; VALS=$2; shift; shift
for val in $VALS; do
  D=gpurun_out/sweep_${V}_$val
  rm -rf $D; mkdir -p $D
  export $V=$val
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D/prof -o p --output-format csv -- python3 bench.py --no-cpu-baseline --psnr-rays 0 --steps 10 --warmup 3 > $D/stats.log 2>&1 || { tail -5 $D/stats.log; exit 1; }
  python3 - "$D" "$V=$val" "$@" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/prof/**/p_kernel_stats.csv', recursive=True)[0]
out = []
for r in csv.reader(open(f)):
    if any(r[0].startswith(p) for p in sys.argv[3:]):
        out.append('%s %.1f us' % (r[0].split('(')[0][:28], float(r[3]) / 1e3))
print(sys.argv[2], ' | '.join(out))
PY
done
