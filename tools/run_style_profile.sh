#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the stylisation iteration (configs[2]); MIOpen's find-mode kernels of the first
# iterations are in the totals, so read the per-call averages and call counts.
export TMPDIR=/tmp
D=gpurun_out/style_prof
rm -rf $D; mkdir -p $D
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $D/prof -o p --output-format csv -- python3 bench.py --no-cpu-baseline --psnr-rays 0 --stage style --steps 30 --warmup 5 > $D/stats.log 2>&1 || { tail -5 $D/stats.log; exit 1; }
python3 - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/prof/**/p_kernel_stats.csv', recursive=True)[0]
rows = list(csv.reader(open(f)))[1:]
tot = sum(float(r[2]) for r in rows)
print('total kernel ms', tot / 1e6, 'per iteration (35 its)', tot / 1e6 / 35)
for r in rows[:45]:
    print(r[0][:100].ljust(100), r[1].rjust(6), '%9.1f us' % (float(r[3]) / 1e3), '%7.2f ms/it' % (float(r[2]) / 1e6 / 35))
PY
