#!/bin/bash
# on the GPU box: per ablation library under tools/abl_run/, the average duration of the backward kernels in the default bench
export TMPDIR=/tmp
for lib in "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = "base" ]; then unset NSR_LIB_PATH; name=base; else export NSR_LIB_PATH=$PWD/$lib; fi
  rm -rf gpurun_out/abl_$name
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/abl_$name -o p --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --psnr-rays 0 > gpurun_out/abl_$name.log 2>&1
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/abl_$name/p_kernel_stats.csv')))
out=[]
for r in rows:
    if 'k_table_scatter' in r['Name'] or ('k_field_bwd' in r['Name']) or 'k_field_fwd' in r['Name'] and 'Lb0' in r['Name']:
        out.append('%s %.2f ms'%(r['Name'].split('(')[0][-28:], float(r['AverageNs'])/1e6))
print('$name:', '; '.join(out), flush=True)
PY
done
