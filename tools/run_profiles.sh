#!/bin/bash
# On the GPU box: the rocprofv3 passes behind profiles/rNN_* (kernel stats; FETCH_SIZE, WRITE_SIZE, TCC_EA0_ATOMIC_sum and the SQ
# counters each in their OWN pass with --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Outputs under gpurun_out/prof_set/;
# afterwards, in the repo:  python tools/make_profiles.py gpurun_out/prof_set rNN <samples_per_launch>
export TMPDIR=/tmp
D=gpurun_out/prof_set
rm -rf $D; mkdir -p $D
B="python3 bench.py --no-cpu-baseline --psnr-rays 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $D/prof -o p --output-format csv -- $B --steps 10 --warmup 3 > $D/stats.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_ATOMIC_sum; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c -d $D/pmc_$c -o p --output-format csv -- $B --steps 2 --warmup 1 > $D/pmc_$c.log 2>&1 || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $D/pmc_SQ -o p --output-format csv -- $B --steps 2 --warmup 1 > $D/pmc_SQ.log 2>&1 || exit 1
tail -1 $D/stats.log | cut -c1-400
echo profiles done
