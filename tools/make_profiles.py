"""Turn the rocprofv3 outputs of the default bench command into the committed profile summaries.

On the GPU box (Measurement):
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o p --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc FETCH_SIZE         -d gpurun_out/pmc_FETCH_SIZE -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
  (same for WRITE_SIZE and TCC_EA0_ATOMIC_sum: one counter group per pass, kernel trace only)
then here:  python tools/make_profiles.py gpurun_out r01
"""
import collections, csv, glob, json, os, shutil, subprocess, sys


def find(d, name):
    hits = glob.glob(os.path.join(d, '**', name), recursive=True)
    return hits[0] if hits else None


def counter_per_launch(d, counter):
    acc, n = collections.defaultdict(float), collections.Counter()
    f = find(d, 'p_counter_collection.csv')
    if not f:
        return {}
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0]
        acc[k] += float(r['Counter_Value'])
        n[k] += 1
    return {k: acc[k] / n[k] for k in acc}


def main(root, tag, samples_per_launch):
    os.makedirs('profiles', exist_ok=True)
    stats = find(os.path.join(root, 'prof'), 'p_kernel_stats.csv')
    if stats:
        shutil.copy(stats, 'profiles/%s_bench_kernel_stats.csv' % tag)
    fetch = counter_per_launch(os.path.join(root, 'pmc_FETCH_SIZE'), 'FETCH_SIZE')
    write = counter_per_launch(os.path.join(root, 'pmc_WRITE_SIZE'), 'WRITE_SIZE')
    atom = counter_per_launch(os.path.join(root, 'pmc_TCC_EA0_ATOMIC_sum'), 'TCC_EA0_ATOMIC_sum')
    out = {
        'source': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / --pmc TCC_EA0_ATOMIC_sum (three separate passes, '
                  '--kernel-trace only), python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline (default workload)',
        'samples_per_launch': samples_per_launch,
        'commit': subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None,
        'correction': 'MI355X_MICROARCH.md HBM section: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE half-counts '
                      'wide reads; WRITE_SIZE exact, float atomics are counted as writes); gather access widths are uncalibrated',
        'kernels': {},
    }
    for k in sorted(set(fetch) | set(write)):
        short = next((n for n in ('k_field_bwd', 'k_field_fwd', 'k_table_scatter', 'k_order_keys', 'k_sort_downsweep', 'k_sort_upsweep',
                                  'k_comp_fwd', 'k_comp_bwd', 'k_composite_train_fwd', 'k_composite_train_bwd', 'k_recon_loss',
                                  'k_grad_check', 'k_march_count', 'k_march_emit', 'k_adam') if n in k), None)
        if not short:
            continue
        if short == 'k_field_fwd' and ('Lb1EEv9FieldArgs' in k or 'true>' in k):
            short = 'k_field_fwd_sigma_only'          # the occupancy update's sigma-only instantiation (4.19 M points)
        fb, wb = fetch.get(k, 0.0), write.get(k, 0.0)
        e = {'FETCH_SIZE_KB': round(fb, 1), 'WRITE_SIZE_KB': round(wb, 1),
             'traffic_bytes_per_launch': int((2 * fb + wb) * 1024),
             'traffic_bytes_per_sample': round((2 * fb + wb) * 1024 / samples_per_launch, 1)}
        if k in atom and atom[k] > 0:
            e['TCC_EA0_ATOMIC_sum_per_launch'] = int(atom[k])
            e['atomic_requests_per_sample'] = round(atom[k] / samples_per_launch, 2)
        out['kernels'][short] = e
    with open('profiles/%s_pmc_traffic.json' % tag, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out['kernels'].get('k_field_bwd'), indent=1))


def sq_counters(root, tag):
    """profiles/<tag>_sq_counters.json from a pass with
    --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"""
    f = find(os.path.join(root, 'pmc_SQ'), 'p_counter_collection.csv')
    if not f:
        return
    acc, n = collections.defaultdict(lambda: collections.defaultdict(float)), collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if (k, r['Dispatch_Id']) not in seen:
            seen.add((k, r['Dispatch_Id']))
            n[k] += 1
    out = {'source': 'rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS '
                     'SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -- python3 bench.py --steps 2 --warmup 1 '
                     '--no-cpu-baseline --psnr-rays 0 (one pass; sums over the launches of a kernel)',
           'units': 'fractions of SQ_WAVE_CYCLES (summed over waves): wait_any = parked in s_waitcnt, wait_inst_any = issue stall, '
                    'active_any = executing; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES)',
           'commit': subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip() or None,
           'kernels': {}}
    for k, v in acc.items():
        w = v.get('SQ_WAVE_CYCLES', 0.0)
        if w < 1e8 or not k.startswith(('k_', '_Z11k_field', 'k_field')) and 'k_field' not in k and 'k_comp' not in k:
            continue
        out['kernels'][k[:60]] = {'launches': n[k], 'SQ_WAVE_CYCLES': w, 'wait_any': round(v['SQ_WAIT_ANY'] / w, 3),
                                  'wait_inst_any': round(v['SQ_WAIT_INST_ANY'] / w, 3), 'active_any': round(v['SQ_ACTIVE_INST_ANY'] / w, 3),
                                  'active_valu': round(v['SQ_ACTIVE_INST_VALU'] / w, 3), 'active_lds': round(v['SQ_ACTIVE_INST_LDS'] / w, 4),
                                  'wait_inst_lds': round(v['SQ_WAIT_INST_LDS'] / w, 4),
                                  'mfma_busy': round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * w), 3)}
    with open('profiles/%s_sq_counters.json' % tag, 'w') as fo:
        json.dump(out, fo, indent=1)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 48565319)
    sq_counters(sys.argv[1], sys.argv[2])
