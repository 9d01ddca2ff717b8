#!/usr/bin/env python3
"""Headline benchmark: DG_AE train step (run_batch + backward + gradient all-reduce + Adam) on
BASELINE.json config 2 — AIG, batch 64 x 65,536-node synthetic AIGs per GPU (N = 4,194,304 nodes,
E = 6,558,720 edges, 120 levels), H=64, 4+4 structural rounds, LayerNorm, loss weights [1,4,4].

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0): whole-job graphs/s with inputs resident in HBM, plus `roofline`
(dominant kernel: algorithmic FLOPs / HIP-event device time vs the fp32 MFMA peak) and `cpu_baseline`
(the oracle, a vectorised PyTorch-CPU port pinned to the reference's golden vectors, timed on the
host cores on a bounded sample of the same workload).  Batch construction (CSR, level tiles) is
outside the timed region and reported as `plan_ms`.
"""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'multi-gate-vae_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, fp32-input matrix peak (dense)
PEAK_HBM_GBPS = 8000.0           # HBM3E spec (6.3 TB/s is what a streaming copy reaches)


def cpu_baseline(cfg, H, rounds, seed_sd, graphs=4, hip_losses=None):
    """Oracle train step (oracle/ref_cpu.py, O(edges) sweep) on `graphs` graphs of the workload — a
    bounded sample — on the host cores this process may use (capped at the box's 16-core share).
    `hip_losses(arrays)`: the three losses of the HIP path on the same sample (seed weights, the sample's fixed
    negatives, dropout off), compared with the oracle's in `loss_parity`."""
    from deepgate import synthetic as syn
    from oracle import ref_cpu as R
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(cores)
    ctype = syn.CONFIGS[cfg]['ctype']

    def one_step(arrays):
        ob = R.batch_from_arrays(lambda k: arrays[k])
        p = {k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() and 'running_' not in k else v.detach().cpu().clone())
             for k, v in seed_sd.items()}
        bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
        opt = torch.optim.Adam(R.trainable(p), lr=1e-4)
        plan = R.LevelPlan(ctype, ob['edge_index'], ob['gate'], ob['forward_level'])
        t0 = time.time()
        opt.zero_grad()
        ls = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.2, s_rounds=rounds, t_rounds=rounds, plan=plan, fast=True)
        R.weighted_loss(ls, [1.0, 4.0, 4.0]).backward()
        opt.step()
        return time.time() - t0

    print('[cpu_baseline] warm-up on a 1k-node graph ...', file=sys.stderr, flush=True)
    one_step(syn.collate([syn.make_graph(ctype, 1024, 30, 1, n_inputs=64)]))
    print('[cpu_baseline] timing 1 step on %d graphs of the workload, %d threads ...' % (graphs, cores), file=sys.stderr, flush=True)
    arrays = syn.make_batch(cfg, batch=graphs)
    dt = one_step(arrays)
    print('[cpu_baseline] %.1f s' % dt, file=sys.stderr, flush=True)
    out = {'value': graphs / dt, 'unit': 'graphs/s', 'cores': cores, 'kind': 'port',
           'sample': '1 train step on %d graphs of the workload (%d nodes), oracle/ref_cpu.py with the O(edges) sweep, %.1f s'
                     % (graphs, arrays['num_nodes'], dt)}
    if hip_losses is not None:
        print('[cpu_baseline] loss parity on the same sample (dropout off) ...', file=sys.stderr, flush=True)
        ob = R.batch_from_arrays(lambda k: arrays[k])
        p = {k: v.detach().cpu().clone() for k, v in seed_sd.items()}
        bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
        with torch.no_grad():
            ls = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.0, s_rounds=rounds, t_rounds=rounds,
                             plan=R.LevelPlan(ctype, ob['edge_index'], ob['gate'], ob['forward_level']), fast=True)
        ref = [float(ls[k]) for k in ('recon_loss', 'prob_loss', 'func_loss')]
        got = hip_losses(arrays)
        out['loss_parity'] = {'oracle': ref, 'hip': got, 'abs_diff': [abs(a - b) for a, b in zip(got, ref)],
                              'note': 'recon, prob, func on the cpu_baseline sample; target 1e-4 (north_star)'}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--config', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None, help='graphs per GPU (default: the config\'s)')
    ap.add_argument('--neg', choices=['sampled', 'fixed'], default='sampled',
                    help='negative edges drawn on the device every step (as the reference does) or fixed')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit('--gpus %d needs a torchrun launch with that many ranks (WORLD_SIZE=%d)' % (a.gpus, world))
    ndev = max(torch.cuda.device_count(), 1)
    local_dev = local_rank % ndev           # one rank per GPU; wraps only in single-GPU rehearsals of the N>1 path
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    if world > 1 and not dist.is_initialized():
        backend = os.environ.get('MGV_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; gloo only for rehearsals
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', init_method='env://', device_id=dev)
        else:
            dist.init_process_group(backend=backend, init_method='env://')

    import deepgate
    from deepgate import _hip, ops, synthetic as syn
    from deepgate.data import plan_of
    cfg = syn.CONFIGS[a.config]
    ctype = cfg['ctype']
    B = a.batch if a.batch is not None else cfg['batch']
    H, rounds = 64, 4
    arrays = syn.make_batch(a.config, batch=B, first_graph=rank * B)      # every rank its own graphs
    batch = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    if a.neg == 'sampled':
        del batch.neg_edge_index
    N, E = arrays['num_nodes'], arrays['edge_index'].shape[1]

    torch.manual_seed(0)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
    mod = {'aig': deepgate.dg_ae_model_aig, 'mig': deepgate.dg_ae_model_mig, 'xag': deepgate.dg_ae_model_xag,
           'xmg': deepgate.dg_ae_model_xmg}[ctype]
    model = mod.Model(struct_encoder=enc, dim_hidden=H, enable_encode=True, enable_reverse=True)
    seed_sd = {k: v.clone() for k, v in model.state_dict().items()}
    targs = types.SimpleNamespace(model='DG_AE')
    tr = deepgate.Trainer(targs, model, training_id='bench', save_dir='/tmp/mgv_bench_%d' % rank, lr=1e-4,
                          rc_prob_func_weight=[1.0, 4.0, 4.0], device=str(dev), batch_size=B, distributed=(world > 1))
    model.train()

    t0 = time.time()
    plan_of(batch, [g for _, g in model.GATES])
    torch.cuda.synchronize()
    plan_ms = (time.time() - t0) * 1e3

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        tr.train_step(batch)
    sync_all()
    _hip.profile(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ls = tr.train_step(batch)
    sync_all()
    elapsed = time.perf_counter() - t0
    table = _hip.profile(False)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    losses = [float(ls[k].detach()) for k in ('recon_loss', 'prob_loss', 'func_loss')]

    if rank == 0:
        summ = _hip.profile_summary(table)
        order = sorted(summ.items(), key=lambda kv: -kv[1][1])
        for name, (calls, ms) in order:
            print('  %-24s %6d calls %10.3f ms total %9.3f ms/call' % (name, calls, ms, ms / max(calls, 1)), file=sys.stderr)
        dom = order[0][0]
        # full-size launches only: the first half round of each encoder runs the same kernel on a (degree, class)
        # table of a few rows, which must not dilute the per-launch average the roofline is priced on
        times = _hip.profile_times(table, dom)
        full = [t for t in times if t >= 0.5 * max(times)]
        per_launch_s = sum(full) / len(full) * 1e-3
        # algorithmic cost of one launch: rows gathered/streamed once, fp32 storage
        fl = {'mgv_struct_stage_bwd': 36.0 * H * H * N, 'mgv_struct_stage_fwd': 12.0 * H * H * N,
              'mgv_struct_stage_bwd_x3': 36.0 * H * H * N, 'mgv_struct_stage_fwd_x3': 12.0 * H * H * N}.get(dom)
        by = {'mgv_struct_stage_bwd': 4.0 * H * (2 * E + 4 * N) + 8.0 * (N + E), 'mgv_struct_stage_fwd': 4.0 * H * (E + 2 * N) + 4.0 * (2 * N + E),
              'mgv_struct_stage_bwd_x3': 4.0 * H * (2 * E + 4 * N) + 8.0 * (N + E),
              'mgv_struct_stage_fwd_x3': 4.0 * H * (E + 2 * N) + 4.0 * (2 * N + E)}.get(dom)
        roof = {'kernel': dom, 'launch_ms': per_launch_s * 1e3, 'launches_averaged': len(full), 'traffic': None}
        try:        # HBM bytes per launch from the committed PMC passes (same workload only)
            pmc = json.load(open(os.path.join(ROOT, 'profiles', 'r01_pmc_traffic.json')))
            if pmc.get('N') == N and dom in pmc['kernels']:
                roof['traffic'] = pmc['kernels'][dom]['hbm_bytes_per_launch']
                roof['traffic_source'] = 'profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH doubled per the gfx950 note)'
        except (OSError, ValueError, KeyError):
            pass
        if dom.endswith('_x3') and by is not None:
            ach = by / per_launch_s / 1e9          # split-precision MFMA makes the half round HBM-bound
            roof.update(bound='hbm', achieved=ach, peak=PEAK_HBM_GBPS, unit='GB/s', frac=ach / PEAK_HBM_GBPS,
                        algorithmic_TFLOPs=fl / per_launch_s / 1e12)
        elif fl is not None:
            ach = fl / per_launch_s / 1e12
            roof.update(bound='mfma', achieved=ach, peak=PEAK_F32_MFMA_TFLOPS, unit='TFLOP/s', frac=ach / PEAK_F32_MFMA_TFLOPS,
                        algorithmic_GBps=by / per_launch_s / 1e9)
        else:
            roof.update(bound='hbm', achieved=None, peak=PEAK_HBM_GBPS, unit='GB/s', frac=None)
        out = {
            'metric': 'circuit-graphs/sec (train step), AIG-64k batch=64 per GPU' if (a.config in (2, 4) and B == 64) else
                      'circuit-graphs/sec (train step), config %d, %s, batch=%d per GPU' % (a.config, ctype, B),
            'value': world * B * a.steps / elapsed,
            'unit': 'graphs/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32 (dense products as bf16x3 split-precision MFMA, fp32 accumulate)' if ops.PRECISION == 'x3' else 'f32', 'data': 'synthetic',
            'nodes_per_s': world * N * a.steps / elapsed,
            'config': {'workload': 'cfg%d: DG_AE --type %s, %d x %d-node synthetic levelised DAGs per GPU (N=%d, E=%d, %d levels), '
                                   'H=64, 4+4 rounds, layernorm, weights [1,4,4], negatives %s' % (
                                       a.config, ctype, B, cfg['n_nodes'], N, E, cfg['n_levels'], a.neg),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world},
            'plan_ms': plan_ms, 'losses': losses, 'roofline': roof,
        }
        if world == 1 and not a.no_cpu_baseline:
            def hip_losses(sample):
                enc2 = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
                m2 = mod.Model(struct_encoder=enc2, dim_hidden=H, enable_encode=True, enable_reverse=True)
                m2.load_state_dict(seed_sd)
                for mm in m2.modules():
                    if isinstance(mm, torch.nn.Dropout):
                        mm.p = 0.0
                t2 = deepgate.Trainer(targs, m2, training_id='parity', save_dir='/tmp/mgv_bench_parity', lr=1e-4,
                                      rc_prob_func_weight=[1.0, 4.0, 4.0], device=str(dev), batch_size=1, distributed=False)
                m2.train()
                with torch.no_grad():
                    l2 = t2.run_batch(deepgate.CircuitBatch.from_arrays(sample, device=dev))     # fixed negatives of the sample
                return [float(l2[k]) for k in ('recon_loss', 'prob_loss', 'func_loss')]
            out['cpu_baseline'] = cpu_baseline(a.config, H, rounds, seed_sd, hip_losses=hip_losses)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
