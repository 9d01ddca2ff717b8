#!/usr/bin/env python3
"""Per-launch hardware counters of the hot kernels from rocprofv3 counter passes (run on the GPU box, counters only:
no trace domains beside --pmc; the program goes directly after `--`).

  traffic mode (L2 -> fabric bytes; FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md §HBM):
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_f -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_w -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline
    python3 tools/pmc_traffic.py traffic <fetch csv> <write csv> N > profiles/rNN_pmc_traffic.json
  busy mode (matrix / vector pipe occupancy, and — second counter pass, second csv — the LDS side):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE ... -- python3 bench.py ...
    rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES -- python3 bench.py ...
    python3 tools/pmc_traffic.py busy <csv>[,<lds csv>] N > profiles/rNN_pmc_mfma.json

FETCH_SIZE / WRITE_SIZE are KiB of L2 <-> fabric traffic (TCC_EA0 requests): Infinity-Cache hits are INCLUDED, so this is an
upper bound on HBM bytes, not HBM bytes.  FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes).
Launches much smaller than a kernel's largest (the (degree, class) table launches of the struct stage) are left out."""
import collections
import csv
import json
import sys

LAUNCHER = {'k_struct_stage_fwd_x3': 'mgv_struct_stage_fwd_x3', 'k_struct_stage_bwd2_x3': 'mgv_struct_stage_bwd2_x3',
            'k_struct_stage_bwd_x3': 'mgv_struct_stage_bwd_x3',
            'k_level_fwd_x3': 'mgv_func_sweep_fwd_x3 (one level)', 'k_level_bwd_x3': 'mgv_func_sweep_bwd_x3 (one level)',
            'k_sweep_fwd_x3': 'mgv_func_sweep_fwd_x3', 'k_sweep_bwd_x3': 'mgv_func_sweep_bwd_x3',
            'k_sweep_wgrad_x3': 'mgv_func_sweep_bwd_x3 (weight gradient, one slot)', 'k_recon_bwd_pull2': 'mgv_recon_loss_bwd_csr',
            'k_recon<': 'mgv_recon_loss_fwd', 'k_seg_sum': 'mgv_seg_sum', 'k_sweep_fwd_persist': 'mgv_func_sweep_fwd_persist_x3', 'k_sweep_bwd_persist': 'mgv_func_sweep_bwd_persist_x3', 'k_class_pull_sum': 'mgv_class_pull_sum', 'k_linear_fwd_x3': 'mgv_linear_fwd_x3'}


def per_kernel(path):
    """{kernel name: {counter: [value per dispatch]}} (values of one dispatch summed over its rows)."""
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
    for r in csv.DictReader(open(path)):
        acc[r['Kernel_Name']][r['Counter_Name']][r.get('Dispatch_Id', r.get('Correlation_Id', '0'))] += float(r['Counter_Value'])
    return {k: {c: list(d.values()) for c, d in v.items()} for k, v in acc.items()}


def launcher_of(name):
    return next((v for k, v in LAUNCHER.items() if k in name), None)


def big(vals, ref=None):
    ref = ref if ref is not None else vals
    m = max(ref) if ref else 0.0
    return [v for v, rv in zip(vals, ref) if rv >= 0.5 * m] if m > 0 else vals


def traffic(fetch_csv, write_csv, N):
    fetch, write = per_kernel(fetch_csv), per_kernel(write_csv)
    out = {'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (bench.py --steps 1 --warmup 1, config 2); KiB units; '
                   'FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 bytes); L2<->fabric traffic, Infinity-Cache hits '
                   'included: an upper bound on HBM bytes; launches under half of a kernel\'s largest are left out', 'N': N, 'kernels': {}}
    for name, cs in fetch.items():
        key = launcher_of(name)
        if key is None or 'FETCH_SIZE' not in cs:
            continue
        fv = big(cs['FETCH_SIZE'])
        wv = big(write.get(name, {}).get('WRITE_SIZE', [0.0]))
        f, w = sum(fv) / len(fv), sum(wv) / max(len(wv), 1)
        out['kernels'][key] = {'FETCH_SIZE_KiB_avg': f, 'WRITE_SIZE_KiB_avg': w, 'launches': len(fv), 'fabric_bytes_per_launch': (2 * f + w) * 1024}
    return out


def busy(path, N):
    paths = path.split(',')
    data = per_kernel(paths[0])
    lds = per_kernel(paths[1]) if len(paths) > 1 else {}
    out = {'note': 'rocprofv3 --pmc pass of bench.py --steps 1 --warmup 1 (config 2); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): '
                   'share of SIMD cycles with the matrix pipe busy; valu_busy = 4 x SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES / 4 (quad-cycles '
                   'of vector issue per SIMD cycle); wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers); coexec_share = SQ_VALU_MFMA_COEXEC_CYCLES / (4 x SQ_BUSY_CU_CYCLES); lds: a second counter pass (SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS, SQ_LDS_BANK_CONFLICT, SQ_WAIT_INST_LDS, SQ_LDS_IDX_ACTIVE): lds_active_share = SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES (the CU has ONE LDS pipe, 128 B/clk), bank_conflict_of_lds_active = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, wait_inst_lds_share = SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES', 'N': N, 'kernels': {}}
    for name, cs in data.items():
        key = launcher_of(name)
        if key is None or 'SQ_BUSY_CU_CYCLES' not in cs:
            continue
        ref = cs['SQ_BUSY_CU_CYCLES']
        tot = lambda c: sum(big(cs[c], ref)) if c in cs else None
        cu = tot('SQ_BUSY_CU_CYCLES')
        rec = {'launches': len(big(ref)), 'SQ_BUSY_CU_CYCLES': cu}
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY',
                  'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_INSTS_VALU_MFMA_MOPS_BF16', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE'):
            if c in cs:
                rec[c] = tot(c)
        if cu and 'SQ_VALU_MFMA_BUSY_CYCLES' in rec:
            rec['mfma_busy'] = rec['SQ_VALU_MFMA_BUSY_CYCLES'] / (4.0 * cu)
        if cu and 'SQ_ACTIVE_INST_VALU' in rec:
            rec['valu_busy'] = rec['SQ_ACTIVE_INST_VALU'] * 4.0 / (4.0 * cu)
        if cu and 'SQ_VALU_MFMA_COEXEC_CYCLES' in rec:
            # cycles in which vector and matrix instructions execute together (MI355X_MICROARCH.md, two waves per SIMD, item 9), as a
            # share of SIMD cycles and of the matrix pipe's busy cycles
            rec['coexec_share'] = rec['SQ_VALU_MFMA_COEXEC_CYCLES'] / (4.0 * cu)
            if rec.get('SQ_VALU_MFMA_BUSY_CYCLES'):
                rec['coexec_of_mfma_busy'] = rec['SQ_VALU_MFMA_COEXEC_CYCLES'] / rec['SQ_VALU_MFMA_BUSY_CYCLES']
        if rec.get('SQ_WAVE_CYCLES') and 'SQ_WAIT_ANY' in rec:
            rec['wait_share'] = rec['SQ_WAIT_ANY'] / rec['SQ_WAVE_CYCLES']
        ls = lds.get(name)
        if ls and 'SQ_BUSY_CU_CYCLES' in ls:
            # the LDS pass is another run of the same program: its ratios are formed with ITS OWN cycle counters
            lref = ls['SQ_BUSY_CU_CYCLES']
            ltot = lambda c: sum(big(ls[c], lref)) if c in ls else None
            lrec = {c: ltot(c) for c in ('SQ_INSTS_LDS', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT', 'SQ_WAIT_INST_LDS', 'SQ_LDS_IDX_ACTIVE',
                                         'SQ_LDS_ADDR_CONFLICT', 'SQ_LDS_UNALIGNED_STALL', 'SQ_BUSY_CU_CYCLES', 'SQ_WAVE_CYCLES') if c in ls}
            lcu = lrec.get('SQ_BUSY_CU_CYCLES')
            if lcu:
                if lrec.get('SQ_LDS_IDX_ACTIVE') is not None:
                    lrec['lds_active_share'] = lrec['SQ_LDS_IDX_ACTIVE'] / lcu          # cycles the CU's LDS pipe works / CU busy cycles
                if lrec.get('SQ_LDS_BANK_CONFLICT') is not None:
                    lrec['lds_bank_conflict_share'] = lrec['SQ_LDS_BANK_CONFLICT'] / lcu
                    if lrec.get('SQ_LDS_IDX_ACTIVE'):
                        lrec['bank_conflict_of_lds_active'] = lrec['SQ_LDS_BANK_CONFLICT'] / lrec['SQ_LDS_IDX_ACTIVE']
                if lrec.get('SQ_ACTIVE_INST_LDS') is not None:
                    lrec['lds_inst_active_share'] = lrec['SQ_ACTIVE_INST_LDS'] / (4.0 * lcu)
            if lrec.get('SQ_WAVE_CYCLES') and lrec.get('SQ_WAIT_INST_LDS') is not None:
                lrec['wait_inst_lds_share'] = lrec['SQ_WAIT_INST_LDS'] / lrec['SQ_WAVE_CYCLES']
            rec['lds'] = lrec
        out['kernels'][key] = rec
    return out


def main():
    mode = sys.argv[1]
    if mode == 'traffic':
        res = traffic(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif mode == 'busy':
        res = busy(sys.argv[2], int(sys.argv[3]))
    else:
        raise SystemExit(__doc__)
    json.dump(res, sys.stdout, indent=1)


if __name__ == '__main__':
    main()
