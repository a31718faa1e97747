#!/bin/bash
# usage: tools/quick_bench.sh TAG [runs] [extra bench args...]: short benches of the default workload, step and per-kernel times
tag=$1; runs=${2:-2}; shift; shift
mkdir -p gpurun_out/qb
for i in $(seq 1 $runs); do
  timeout -k 10 200 python bench.py --no-cpu-baseline --psnr-rays 0 --steps 30 "$@" > gpurun_out/qb/${tag}_$i.json 2>gpurun_out/qb/${tag}_$i.err || { tail -5 gpurun_out/qb/${tag}_$i.err; exit 1; }
  python - "$tag" "$i" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/qb/%s_%s.json' % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], d['ms_per_step'], d['value'], d.get('kernel_ms_per_step'))
PY
done
