#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the reference-sized captured step (bf16 + hipGraph); usage: run_graph_profile.sh [rays]
export TMPDIR=/tmp
R=${1:-4096}
D=gpurun_out/graph_prof_$R
rm -rf $D; mkdir -p $D
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D/prof -o p --output-format csv -- python3 bench.py --no-cpu-baseline --psnr-rays 0 --compute-dtype bf16 --graph --rays-per-gpu $R --steps 100 --warmup 10 > $D/stats.log 2>&1 || { tail -5 $D/stats.log; exit 1; }
tail -1 $D/stats.log | cut -c1-160
python3 - "$D" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/prof/**/p_kernel_stats.csv', recursive=True)[0]
for r in list(csv.reader(open(f)))[1:40]:
    print(r[0][:90].ljust(90), r[1], '%.1f us' % (float(r[3]) / 1e3), r[4])
PY
