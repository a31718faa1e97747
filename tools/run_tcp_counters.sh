#!/bin/bash
# On the GPU box: texture-path counters of the field kernels (one rocprofv3 --pmc pass per group, --kernel-trace only).
export TMPDIR=/tmp
D=gpurun_out/tcp
rm -rf $D; mkdir -p $D
B="python3 bench.py --no-cpu-baseline --psnr-rays 0 --steps 2 --warmup 1"
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp -d $D/g$i -o p --output-format csv -- $B > $D/g$i.log 2>&1 || { tail -5 $D/g$i.log; echo "group $i failed"; }
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob('gpurun_out/tcp/g*/**/p_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')[:44]
        if 'k_field' not in k and 'k_table' not in k:
            continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        n[k].add(r['Dispatch_Id'])
for k, v in acc.items():
    print(k, 'launches/pass ~', len(n[k]))
    for c, x in sorted(v.items()):
        print('   %-44s %.4g' % (c, x))
PY
