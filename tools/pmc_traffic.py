#!/usr/bin/env python3
"""HBM bytes per launch of the hot kernels from two rocprofv3 counter passes.

Run on the GPU box (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md §HBM):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o p -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o p -- python3 $REPO/bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python3 $REPO/tools/pmc_traffic.py /tmp/pmc_f/p_counter_collection.csv /tmp/pmc_w/p_counter_collection.csv N > traffic.json

Counters are KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes).  Launches much smaller
than a kernel's largest (the (degree, class) table launches of the struct stage) are left out of the average."""
import collections
import csv
import json
import sys

LAUNCHER = {'k_struct_stage_fwd_x3': 'mgv_struct_stage_fwd_x3', 'k_struct_stage_bwd_x3': 'mgv_struct_stage_bwd_x3',
            'k_level_fwd_x3': 'mgv_func_sweep_fwd_x3 (one level)', 'k_level_bwd_x3': 'mgv_func_sweep_bwd_x3 (one level)',
            'k_sweep_wgrad_x3': 'mgv_func_sweep_bwd_x3 (weight gradient, one slot)', 'k_recon_bwd_pull2': 'mgv_recon_loss_bwd_csr',
            'k_recon<': 'mgv_recon_loss_fwd', 'k_class_pull_sum': 'mgv_class_pull_sum'}


def per_kernel(path):
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        vals[r['Kernel_Name']].append(float(r['Counter_Value']))
    return vals


def main():
    fetch, write, N = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), int(sys.argv[3])
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 1, config 2); KiB units; '
                   'FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide reads on gfx950; launches under half of a kernel\'s '
                   'largest are left out (table-sized launches of the first half round)', 'N': N, 'kernels': {}}
    for name, fv in fetch.items():
        key = next((v for k, v in LAUNCHER.items() if k in name), None)
        if key is None:
            continue
        wv = write.get(name, [])
        big_f = [v for v in fv if v >= 0.5 * max(fv)]
        big_w = [v for v in wv if v >= 0.5 * max(wv)] if wv and max(wv) > 0 else [0.0]
        f, w = sum(big_f) / len(big_f), sum(big_w) / len(big_w)
        out['kernels'][key] = {'FETCH_SIZE_KiB_avg': f, 'WRITE_SIZE_KiB_avg': w, 'launches': len(big_f),
                               'hbm_bytes_per_launch': (2 * f + w) * 1024}
    json.dump(out, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
