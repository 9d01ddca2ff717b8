#!/usr/bin/env python3
"""Headline benchmark: DG_AE train step (run_batch + backward + gradient all-reduce + Adam + the step's 7-scalar
device->host copy) on BASELINE.json config 2 — AIG, batch 64 x 65,536-node synthetic AIGs per GPU (N = 4,194,304 nodes,
E = 6,558,720 edges, 120 levels), H=64, 4+4 structural rounds, LayerNorm, loss weights [1,4,4].

  python bench.py --gpus N --steps K --warmup W        (N > 1: starts its N ranks itself, one per GPU, over RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...      (what the driver does)

Prints ONE JSON line on stdout (rank 0): whole-job graphs/s with inputs resident in HBM, plus
  `roofline`     dominant kernel (struct-stage backward): SURVEY.md §8(d) algorithmic bytes per launch / its HIP-event time,
                 both byte models (`frac_8d`, `frac_gather_model`), fabric bytes and MFMA-busy share from the committed PMC passes;
  `cpu_baseline` the oracle (vectorised PyTorch-CPU port, pinned to the reference's golden vectors) timed on the host cores on a
                 bounded sample of the same workload, plus the literal restatement (per-node subgraph scans, dense N x N mask)
                 at config 1, which ties back to the survey's timing of the real reference.
Batch construction (CSR, level tiles) is outside the timed region and reported as `plan_ms` (cold) / `plan_ms_steady`.
The per-launcher HIP-event breakdown is taken in a separate short pass AFTER the timed loop (profiling off while timing).
"""
import argparse
import json
import os
import subprocess
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'multi-gate-vae_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, fp32-input matrix peak (dense)
PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 peak (the x3 mode issues 3 bf16 MFMAs per fp32 product)
PEAK_HBM_GBPS = 8000.0           # HBM3E spec (6.3 TB/s is what a streaming copy reaches)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--config', type=int, default=2)
    ap.add_argument('--batch', type=int, default=None, help='graphs per GPU (default: the config\'s)')
    ap.add_argument('--neg', choices=['sampled', 'fixed'], default='sampled',
                    help='negative edges drawn on the device every step (as the reference does) or fixed')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-fresh', action='store_true', help='skip the second timed loop over fresh batches (collate + H2D + plan per step)')
    ap.add_argument('--loader-workers', type=int, default=8)
    ap.add_argument('--no-other-configs', action='store_true',
                    help='skip the short config 3 / config 5 runs (child processes, after the timed loops) a default 1-GPU run appends as "other_configs"')
    ap.add_argument('--master-port', type=int, default=29541)
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as CHILD processes (torch.distributed.run) before this
    process has touched a GPU, stream their output through, exit with their code.  torch.cuda.device_count() does not
    initialise the device on this image."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < a.gpus and os.environ.get('MGV_DIST_BACKEND', 'nccl') == 'nccl':
        raise SystemExit('--gpus %d but only %d GPU(s) visible: one rank per GPU over RCCL (set MGV_DIST_BACKEND=gloo to rehearse '
                         'the N-rank path with ranks stacked on the visible GPUs)' % (a.gpus, ndev))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(a.master_port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    raise SystemExit(subprocess.call(cmd, env=env))


def cpu_baseline(cfg, H, rounds, seed_sd, graphs=4, hip_losses=None):
    """Oracle train step (oracle/ref_cpu.py, O(edges) sweep) on `graphs` graphs of the workload — a bounded sample — on the host
    cores this process may use (capped at the box's 16-core share), plus the literal restatement at config 1.
    `hip_losses(arrays)`: the three losses of the HIP path on the same sample (seed weights, the sample's fixed negatives,
    dropout off), compared with the oracle's in `loss_parity`."""
    import torch
    from deepgate import synthetic as syn
    from oracle import ref_cpu as R
    from oracle import ref_cpu_literal as L
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    torch.set_num_threads(cores)
    ctype = syn.CONFIGS[cfg]['ctype']

    def fresh():
        p = {k: (v.detach().cpu().clone().requires_grad_(True) if v.is_floating_point() and 'running_' not in k else v.detach().cpu().clone())
             for k, v in seed_sd.items()}
        return p, {k: v.clone() for k, v in p.items() if 'running_' in k}

    def one_step(arrays, literal=False, ctype_=None):
        ct = ctype_ or ctype
        ob = R.batch_from_arrays(lambda k: arrays[k])
        p, bn = fresh()
        opt = torch.optim.Adam(R.trainable(p), lr=1e-4)
        t0 = time.time()
        opt.zero_grad()
        if literal:
            ls = L.run_batch(p, ct, ob, training=True, bn_state=bn, p_drop=0.2, s_rounds=rounds, t_rounds=rounds)
        else:
            plan = R.LevelPlan(ct, ob['edge_index'], ob['gate'], ob['forward_level'])      # the reference rebuilds its masks every step too
            ls = R.run_batch(p, ct, ob, training=True, bn_state=bn, p_drop=0.2, s_rounds=rounds, t_rounds=rounds, plan=plan, fast=True)
        R.weighted_loss(ls, [1.0, 4.0, 4.0]).backward()
        opt.step()
        return time.time() - t0

    log = lambda m: print('[cpu_baseline] ' + m, file=sys.stderr, flush=True)
    log('warm-up on a 1k-node graph ...')
    one_step(syn.collate([syn.make_graph(ctype, 1024, 30, 1, n_inputs=64)]))
    log('timing 1 step on %d graphs of the workload, %d threads ...' % (graphs, cores))
    arrays = syn.make_batch(cfg, batch=graphs)
    dt = one_step(arrays)
    log('%.1f s' % dt)
    full_b = syn.CONFIGS[cfg]['batch']
    out = {'value': graphs / dt, 'unit': 'graphs/s', 'cores': cores, 'kind': 'port', 's_per_graph': dt / graphs,
           'sample': '1 train step on %d graphs of the workload (%d nodes), oracle/ref_cpu.py with the O(edges) sweep, %.1f s; the full '
                     'batch of %d graphs would take ~%.0f s (cost is linear in graphs), beyond what a default bench run may spend'
                     % (graphs, arrays['num_nodes'], dt, full_b, dt / graphs * full_b)}
    if ctype == 'aig':
        # (ii) SURVEY.md §8d: the literal restatement at config 1 (4 x 1,024-node AIGs), where the survey measured the real
        # reference at 1.5-1.8 s/step on 8 cores
        a1 = syn.make_batch(1)
        one_step(a1, literal=True)
        ts = [one_step(a1, literal=True) for _ in range(3)]
        tv = [one_step(a1) for _ in range(3)]
        out['literal_cfg1'] = {'s_per_step': sorted(ts)[1], 'graphs_per_s': 4 / sorted(ts)[1], 'port_s_per_step': sorted(tv)[1], 'cores': cores,
                               'sample': 'median of 3 train steps, config 1 (4 x 1,024-node AIGs, N=4,096, E=6,480): oracle/ref_cpu_literal.py keeps the '
                                         'per-node subgraph scans, the full-state level loop and the dense N x N edge-split mask of the reference; '
                                         'the survey timed the reference itself at 1.5-1.8 s/step on 8 cores (SURVEY.md §6)'}
        log('literal cfg1: %.2f s/step (vectorised port: %.3f s/step)' % (sorted(ts)[1], sorted(tv)[1]))
    if hip_losses is not None:
        log('loss parity on the same sample (dropout off) ...')
        ob = R.batch_from_arrays(lambda k: arrays[k])
        p = {k: v.detach().cpu().clone() for k, v in seed_sd.items()}
        bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
        with torch.no_grad():
            ls = R.run_batch(p, ctype, ob, training=True, bn_state=bn, p_drop=0.0, s_rounds=rounds, t_rounds=rounds,
                             plan=R.LevelPlan(ctype, ob['edge_index'], ob['gate'], ob['forward_level']), fast=True)
        ref = [float(ls[k]) for k in ('recon_loss', 'prob_loss', 'func_loss')]
        got = hip_losses(arrays)
        out['loss_parity'] = {'oracle': ref, 'hip': got, 'abs_diff': [abs(x - y) for x, y in zip(got, ref)],
                              'note': 'recon, prob, func on the cpu_baseline sample; target 1e-4 (north_star)'}
    return out


def roofline_record(summ, table, N, E, H, step_s):
    """Roofline of the dominant kernel.  Bytes per launch, two models, both stated in DESIGN.md §4:
       8d      SURVEY.md §8(d): each operand once, gathered rows cache-resident: bwd 3*N*H*4 + 8*(N+E), fwd 2*N*H*4 + 4*(N+E)
       gather  every gathered row priced as memory traffic: bwd (2E+4N)*4H + 8*(N+E), fwd (E+2N)*4H + 4*(2N+E)"""
    from deepgate import _hip
    order = sorted(summ.items(), key=lambda kv: -kv[1][1])
    dom = order[0][0]
    times = _hip.profile_times(table, dom)
    full = [t for t in times if t >= 0.5 * max(times)]          # the two table-sized launches per step must not dilute the average
    per_launch_s = sum(full) / len(full) * 1e-3
    is_bwd = 'bwd' in dom
    by_8d = (3.0 if is_bwd else 2.0) * N * H * 4 + (8.0 if is_bwd else 4.0) * (N + E)
    by_g = 4.0 * H * ((2 * E + 4 * N) if is_bwd else (E + 2 * N)) + (8.0 * (N + E) if is_bwd else 4.0 * (2 * N + E))
    fl = (36.0 if is_bwd else 12.0) * H * H * N
    roof = {'kernel': dom, 'launch_ms': per_launch_s * 1e3, 'launches_averaged': len(full), 'bound': 'hbm', 'unit': 'GB/s', 'peak': PEAK_HBM_GBPS,
            'achieved': by_8d / per_launch_s / 1e9, 'frac': by_8d / per_launch_s / 1e9 / PEAK_HBM_GBPS,
            'frac_8d': by_8d / per_launch_s / 1e9 / PEAK_HBM_GBPS, 'frac_gather_model': by_g / per_launch_s / 1e9 / PEAK_HBM_GBPS,
            'bytes_8d': by_8d, 'bytes_gather_model': by_g,
            'algorithmic_TFLOPs': fl / per_launch_s / 1e12, 'mfma_frac_bf16x3': 3 * fl / per_launch_s / 1e12 / PEAK_BF16_MFMA_TFLOPS,
            'traffic': None, 'fabric_over_algorithmic': None, 'mfma_busy': None}
    if not dom.endswith('_x3'):
        roof.update(bound='mfma', unit='TFLOP/s', peak=PEAK_F32_MFMA_TFLOPS, achieved=fl / per_launch_s / 1e12,
                    frac=fl / per_launch_s / 1e12 / PEAK_F32_MFMA_TFLOPS)
    # whole step against SURVEY.md §8(d)'s step model (fp32 storage): B_step = 104*N*H*4 + 192*(N+E), F_step = 3 * F_fwd
    n_g = N - N // 16
    b_step = 104.0 * N * H * 4 + 192.0 * (N + E)
    f_step = 3.0 * (993472.0 * N + 33280.0 * E + 49152.0 * n_g)
    roof['step'] = {'bytes_8d': b_step, 'hbm_frac': b_step / step_s / 1e9 / PEAK_HBM_GBPS, 'flops_8d': f_step,
                    'mfma_frac_bf16x3': 3 * f_step / step_s / 1e12 / PEAK_BF16_MFMA_TFLOPS}
    import glob
    for suffix, key in (('_pmc_traffic.json', 'traffic'), ('_pmc_mfma.json', 'mfma')):
        names = sorted((os.path.basename(f) for f in glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]' + suffix))), reverse=True)
        pmc, fname = None, None
        for fname in names:              # the newest committed counter pass
            try:
                pmc = json.load(open(os.path.join(ROOT, 'profiles', fname)))
                break
            except (OSError, ValueError):
                continue
        if pmc is None:
            continue
        if pmc.get('N') != N:
            continue
        kern = pmc.get('kernels', {}).get(dom)
        if kern is None:
            continue
        if key == 'traffic':
            roof['traffic'] = kern['fabric_bytes_per_launch']
            roof['fabric_over_algorithmic'] = kern['fabric_bytes_per_launch'] / by_8d
            roof['traffic_source'] = 'profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH doubled per the gfx950 note; ' \
                                     'L2->fabric bytes, Infinity-Cache hits INCLUDED (MI355X_MICROARCH.md), so an upper bound on HBM bytes' % fname
        else:
            roof['mfma_busy'] = kern.get('mfma_busy')
            roof['valu_busy'] = kern.get('valu_busy')
            if kern.get('coexec_share') is not None:
                roof['valu_mfma_coexec_share'] = kern.get('coexec_share')
            if kern.get('mfma_busy') is not None and kern.get('valu_busy') is not None:
                # matrix and vector instructions of a SIMD's waves take turns on this part (tools/micro/coexec.hip,
                # profiles/r03_coexec_micro.txt): the two busy shares add up to the SIMDs' issue occupancy
                roof['simd_issue_share'] = kern['mfma_busy'] + kern['valu_busy']
            roof['mfma_source'] = 'profiles/%s: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES), rocprofv3 --pmc' % fname
    return roof, order


def other_configs(dev):
    """{'cfg3': {...}, 'cfg5': {...}}: ms_per_step / value / workload / dominant-kernel roofline of `bench.py --config K` (6 steps, resident
    batch, per-GPU share of the config), each from a child process; a failure is recorded as such, never hidden."""
    import torch
    torch.cuda.empty_cache()
    res = {}
    for k in (3, 5):
        cmd = [sys.executable, os.path.abspath(__file__), '--config', str(k), '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--no-fresh',
               '--no-other-configs']
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
            d = json.loads(r.stdout.decode().strip().splitlines()[-1])
            res['cfg%d' % k] = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'nodes_per_s': d['nodes_per_s'],
                                'workload': d['config']['workload'], 'losses': d['losses'],
                                'roofline': {kk: d['roofline'].get(kk) for kk in ('kernel', 'launch_ms', 'bound', 'frac', 'achieved', 'peak', 'unit')}}
        except Exception as exc:         # noqa: BLE001 (reported in the line)
            res['cfg%d' % k] = {'error': '%s: %s' % (type(exc).__name__, str(exc)[:200])}
    return res


def main():
    a = parse_args()
    world_env = int(os.environ.get('WORLD_SIZE', '0') or 0)
    if a.gpus > 1 and world_env == 0:
        self_launch(a)
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = max(world_env, 1)
    if a.gpus != world:
        raise SystemExit('--gpus %d but the launcher started %d rank(s)' % (a.gpus, world))
    backend = os.environ.get('MGV_DIST_BACKEND', 'nccl')      # 'nccl' is RCCL on ROCm; gloo only for rehearsals
    ndev = max(torch.cuda.device_count(), 1)
    if local_rank >= ndev and backend == 'nccl':
        raise SystemExit('LOCAL_RANK %d but %d GPU(s) visible' % (local_rank, ndev))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    dev = torch.device('cuda', local_dev)
    if world > 1 and not dist.is_initialized():
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', init_method='env://', device_id=dev)
        else:
            dist.init_process_group(backend=backend, init_method='env://')

    import contextlib
    import deepgate
    from deepgate import _hip, ops, synthetic as syn
    from deepgate.data import plan_of
    cfg = syn.CONFIGS[a.config]
    ctype = cfg['ctype']
    B = a.batch if a.batch is not None else cfg['batch']
    H, rounds = 64, 4
    graphs = syn.make_graphs(a.config, batch=B, first_graph=rank * B)      # every rank its own graphs
    arrays = syn.collate(graphs)
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
    with contextlib.redirect_stdout(sys.stderr):
        tr = deepgate.Trainer(targs, model, training_id='bench', save_dir='/tmp/mgv_bench_%d' % rank, lr=1e-4,
                              rc_prob_func_weight=[1.0, 4.0, 4.0], device=str(dev), batch_size=B, distributed=(world > 1))
    model.train()

    gate_ids = [g for _, g in model.GATES]
    def build_plan(b):
        # everything a FRESH batch needs before its first step: CSRs, level tiles, and the per-batch caches of the structural
        # encoder (colour refinement of the quotient stages, or the first-stage table): what the prefetcher's workers do per batch
        pl = plan_of(b, gate_ids)
        pl.warm(pl.xcls, quotient_stages=2 * rounds)
        return pl

    t0 = time.time()
    build_plan(batch)
    torch.cuda.synchronize()
    plan_ms = (time.time() - t0) * 1e3
    # steady state: the same construction on a second, equally shaped batch (allocator and kernels warm)
    b2 = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
    torch.cuda.synchronize()
    t0 = time.time()
    build_plan(b2)
    torch.cuda.synchronize()
    plan_ms_steady = (time.time() - t0) * 1e3
    del b2

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ls = tr.train_step(batch)
        # the reference's step ends with a host copy of its metrics (trainer.py:236-244); here: 3 losses + 4 counters, copied
        # every step into pinned memory and read one step behind (Trainer.enqueue_metrics, as Trainer.train does)
        return tr.enqueue_metrics(ls)

    for _ in range(a.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    vals = tr.flush_metrics()
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    losses = vals[:3]

    # ---- second timed loop: FRESH batches.  Every step's batch is collated again from the per-graph arrays (in a rotated order: a
    #      different batch tensor every step), copied over PCIe and planned afresh — nothing of it is cached on the device.  Collate,
    #      copy and plan build run on the prefetcher's worker threads / streams beside the previous step (deepgate/prefetch.py: what
    #      Trainer.train uses); the reference's loop does the same work with DataLoader workers + `batch.to(device)` (trainer.py:189-195,223).
    fresh = None
    if not a.no_fresh:
        from deepgate.prefetch import BatchPrefetcher, _Staging, collate_into
        skip = ('neg_edge_index',) if a.neg == 'sampled' else ()
        # a SUSTAINED rate: enough untimed steps to drain what the prefetcher queued while nothing consumed, enough timed ones to average
        # over its bursts (the first ~10 steps after a warm-up run 2 % faster than the loop's steady state)
        f_warm, f_steps = max(a.warmup, 12), max(a.steps, 40)
        n_fresh = f_warm + f_steps

        def chunks():
            for s_ in range(n_fresh):
                yield graphs[s_ % B:] + graphs[:s_ % B]
        t0 = time.perf_counter()
        host, _ = collate_into(graphs, _Staging(False), [k for k in ('x', 'edge_index', 'gate', 'forward_level', 'forward_index', 'prob',
                                                                     'tt_pair_index', 'tt_sim', 'neg_edge_index') if k not in skip])
        collate_ms = (time.perf_counter() - t0) * 1e3
        h2d_bytes = sum(int(t.numel()) * t.element_size() for t in host.values())
        del host
        pf = BatchPrefetcher(chunks(), dev, gate_ids=gate_ids, workers=a.loader_workers, skip=skip)
        it = iter(pf)
        for _ in range(f_warm):
            tr.enqueue_metrics(tr.train_step(next(it)))
        sync_all()
        t0 = time.perf_counter()
        for _ in range(f_steps):
            tr.enqueue_metrics(tr.train_step(next(it)))
        tr.flush_metrics()
        sync_all()
        el_f = time.perf_counter() - t0
        pf.close()
        if world > 1:
            tmax = torch.tensor([el_f], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            el_f = float(tmax.item())
        fresh = {'value': world * B * f_steps / el_f, 'ms_per_step': el_f / f_steps * 1e3, 'steps': f_steps, 'warmup': f_warm, 'loader_workers': a.loader_workers,
                 'h2d_bytes_per_batch': h2d_bytes, 'host_collate_ms_one_thread': collate_ms,
                 'note': 'graphs/s over batches that are collated (pinned staging), copied host-to-device and planned per step on %d worker '
                         'threads with their own HIP streams, overlapped with the previous step; PCIe-inclusive' % a.loader_workers}

    # ---- separate short pass: per-launcher HIP events (and the all-reduce alone), profiling was OFF in the timed loop
    _hip.profile(True)
    for _ in range(2):
        step()
    table = _hip.profile(False)
    allreduce_ms = None
    if world > 1:
        f = tr.optimizer.flat_buffers()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync_all()
        ts = []
        for _ in range(5):
            s.record()
            dist.all_reduce(f['grad'], op=dist.ReduceOp.SUM)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        allreduce_ms = sorted(ts)[2]

    if rank == 0:
        summ = _hip.profile_summary(table)
        roof, order = roofline_record(summ, table, N, E, H, elapsed / a.steps)
        for name, (calls, ms) in order:
            print('  %-26s %6d calls %10.3f ms total %9.3f ms/call' % (name, calls, ms, ms / max(calls, 1)), file=sys.stderr)
        mode = 'f32 storage; dense products as bf16x3 split-precision MFMA, fp32 accumulate' if ops.PRECISION == 'x3' else 'f32 (fp32-input MFMA)'
        out = {
            'metric': 'circuit-graphs/sec (train step), AIG-64k batch=64 per GPU' if (a.config in (2, 4) and B == 64) else
                      'circuit-graphs/sec (train step), config %d, %s, batch=%d per GPU' % (a.config, ctype, B),
            'value': world * B * a.steps / elapsed,
            'unit': 'graphs/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': mode, 'data': 'synthetic',
            'nodes_per_s': world * N * a.steps / elapsed,
            'config': {'workload': 'cfg%d: DG_AE --type %s, %d x %d-node synthetic levelised DAGs per GPU (N=%d, E=%d, %d levels), '
                                   'H=64, 4+4 rounds, layernorm, weights [1,4,4], negatives %s; step includes the 7-scalar metrics copy (pinned, read one step behind)' % (
                                       a.config, ctype, B, cfg['n_nodes'], N, E, cfg['n_levels'], a.neg),
                       'global_batch': world * B, 'parallelism': 'dp%d' % world},
            'plan_ms': plan_ms, 'plan_ms_steady': plan_ms_steady,
            # a loop that built every batch's plan (CSRs, level tiles, colour refinement) SERIALLY in front of its step; the prefetcher
            # overlaps that with the previous step: value_fresh_batches is the measured loop rate
            'value_with_plan_build': world * B / (elapsed / a.steps + plan_ms_steady * 1e-3),
            'losses': losses, 'roofline': roof,
        }
        quot = batch._mgv_plan.quotient(batch._mgv_plan.xcls, 2 * rounds) if ops.QUOTIENT else []
        out['struct_encoder'] = {'half_rounds': 2 * rounds, 'rows_per_half_round': [s['C'] for s in quot] + [N] * (2 * rounds - len(quot)),
                                 'note': 'every node starts from ones (digae_layer.py:260): the first half rounds have few distinct rows (colour '
                                         'refinement) and are computed on one row per colour, exactly (DESIGN.md 4.3; MGV_QUOTIENT=0 turns it off)'}
        if fresh is not None:
            out['value_fresh_batches'] = fresh['value']
            out['fresh_batches'] = fresh
        if allreduce_ms is not None:
            out['allreduce_ms'] = allreduce_ms
            out['allreduce_bytes'] = int(tr.optimizer.flat_buffers()['grad'].numel()) * 4
        if world == 1 and not a.no_cpu_baseline:
            def hip_losses(sample):
                enc2 = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=rounds, t_rounds=rounds, layernorm=True)
                m2 = mod.Model(struct_encoder=enc2, dim_hidden=H, enable_encode=True, enable_reverse=True)
                m2.load_state_dict(seed_sd)
                for mm in m2.modules():
                    if isinstance(mm, torch.nn.Dropout):
                        mm.p = 0.0
                with contextlib.redirect_stdout(sys.stderr):
                    t2 = deepgate.Trainer(targs, m2, training_id='parity', save_dir='/tmp/mgv_bench_parity', lr=1e-4,
                                          rc_prob_func_weight=[1.0, 4.0, 4.0], device=str(dev), batch_size=1, distributed=False)
                m2.train()
                with torch.no_grad():
                    l2 = t2.run_batch(deepgate.CircuitBatch.from_arrays(sample, device=dev))     # fixed negatives of the sample
                return [float(l2[k]) for k in ('recon_loss', 'prob_loss', 'func_loss')]
            out['cpu_baseline'] = cpu_baseline(a.config, H, rounds, seed_sd, hip_losses=hip_losses)
        if world == 1 and a.config == 2 and a.batch is None and not a.no_other_configs:
            # BASELINE.json's other single-GPU configurations (parity-test cases, not the metric): a short resident-batch run each, in a
            # child process after this one's timed loops are over, so that the driver's default run records them too
            out['other_configs'] = other_configs(dev)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
