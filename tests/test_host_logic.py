"""CPU tests of the host-side logic: graph plan vs brute force, synthetic generator, loader sharding,
C-ABI header/library agreement, flat optimiser bookkeeping, and the world_size-2 gradient exchange
over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

import deepgate
from deepgate import synthetic as syn
from deepgate.graph_plan import GraphPlan


def test_header_and_library_agree():
    """Every `int mgv_*` declared in include/mgvae_hip.h is exported by the built library (no compute)."""
    from deepgate import _hip
    sigs = _hip.parse_header()
    assert len(sigs) >= 26
    if not os.path.exists(_hip.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in sigs:
        assert hasattr(lib, name), name
    out = subprocess.run(['nm', '-D', '--defined-only', _hip.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r'\bT (mgv_\w+)', out))
    assert exported == set(sigs), exported ^ set(sigs)
    assert lib.mgv_abi_version() >= 1


def test_missing_gpu_fails_loudly():
    """No CPU fallback: a CPU tensor is refused before anything is launched."""
    from deepgate import _hip, ops
    with pytest.raises(_hip.HipLibraryError):
        ops.l1_loss(torch.zeros(4), torch.zeros(4))


def test_synthetic_generator_matches_survey_counts_and_is_deterministic():
    a = syn.make_batch(1)
    b = syn.make_batch(1)
    assert a['num_nodes'] == 4096 and a['edge_index'].shape == (2, 6480)
    for k in ('edge_index', 'prob', 'tt_sim', 'neg_edge_index'):
        assert np.array_equal(a[k], b[k])
    g = syn.make_graph('xmg', 96, 6, 5, n_inputs=12)
    ei, lv = g['edge_index'], g['forward_level']
    asap = np.zeros(96, dtype=np.int64)
    np.maximum.at(asap, ei[1], lv[ei[0]] + 1)          # valid because ids are level-sorted
    assert np.array_equal(asap, lv)
    key = ei[0] * 96 + ei[1]
    assert len(np.unique(key)) == len(key)            # distinct fan-ins
    nkey = g['neg_edge_index'][0] * 96 + g['neg_edge_index'][1]
    assert not np.isin(nkey, key).any() and (g['neg_edge_index'][0] != g['neg_edge_index'][1]).all()
    fan = np.bincount(ei[1], minlength=96)
    gate = g['gate'].reshape(-1).astype(int)
    want = {0: 0, 1: 3, 2: 1, 3: 2, 4: 2, 5: 2}
    assert all(fan[i] == want[gate[i]] for i in range(96))


def test_graph_plan_against_brute_force():
    g = syn.collate([syn.make_graph('xmg', 96, 6, 5, n_inputs=12), syn.make_graph('xmg', 80, 4, 6, n_inputs=8)])
    ei = torch.from_numpy(g['edge_index'])
    N = g['num_nodes']
    plan = GraphPlan(ei, N).set_levels(torch.from_numpy(g['gate']), torch.from_numpy(g['forward_level']), [3, 2, 5, 1, 4])
    src, dst = g['edge_index']
    for v in range(N):
        ins = sorted(src[dst == v].tolist())
        got = plan.in_src[plan.in_ptr[v]:plan.in_ptr[v + 1]].tolist()
        assert sorted(got) == ins
        outs = sorted(dst[src == v].tolist())
        assert sorted(plan.out_dst[plan.out_ptr[v]:plan.out_ptr[v + 1]].tolist()) == outs
    # out_slot points at the same edge in the in-CSR
    for v in range(N):
        for e in range(int(plan.out_ptr[v]), int(plan.out_ptr[v + 1])):
            sl = int(plan.out_slot[e])
            assert int(plan.in_src[sl]) == v and int(plan.in_dst[sl]) == int(plan.out_dst[e])
    # tiles: single slot, single level, cover every updated node exactly once, levels ascending
    gate = g['gate'].reshape(-1).astype(int)
    lv = g['forward_level']
    slot_of = {3: 0, 2: 1, 5: 2, 1: 3, 4: 4}
    seen = []
    for lvl in range(plan.num_levels):
        for t in range(plan.level_tile_ptr[lvl], plan.level_tile_ptr[lvl + 1]):
            s, c, sl = int(plan.tile_start[t]), int(plan.tile_count[t]), int(plan.tile_slot[t])
            assert 1 <= c <= 64
            nodes = plan.order[s:s + c].tolist()
            assert all(lv[n] == lvl and slot_of[gate[n]] == sl for n in nodes)
            seen += nodes
    want = [n for n in range(N) if lv[n] >= 1 and gate[n] in slot_of]
    assert sorted(seen) == want and plan.level_tile_ptr[1] == 0
    assert all(int(plan.gslot[n]) == (slot_of[gate[n]] if n in set(want) else 255) for n in range(N))


def test_graph_plan_rejects_non_topological_levels():
    ei = torch.tensor([[0, 1], [1, 2]])
    with pytest.raises(ValueError):
        GraphPlan(ei, 3).set_levels(torch.tensor([[0.], [1.], [1.]]), torch.tensor([0, 1, 1]), [1])


def test_loader_rank_striding_matches_distributed_sampler():
    graphs = [syn.make_graph('aig', 64, 3, i, n_inputs=4) for i in range(10)]
    seen = []
    for r in range(2):
        ld = deepgate.GraphLoader(graphs, batch_size=2, shuffle=False, rank=r, world_size=2)
        assert len(ld) == 2
        for b in ld:
            assert b.num_nodes == 128 and b.edge_index.max() < 128
            seen.append(int(b.prob.shape[0]))
    assert len(seen) == 4
    smp = torch.utils.data.distributed.DistributedSampler(graphs, num_replicas=2, rank=1, shuffle=False)
    ld = deepgate.GraphLoader(graphs, batch_size=1, shuffle=False, rank=1, world_size=2)
    assert list(smp) == ld._indices()


def test_flat_adam_views_and_checkpoint_format():
    m = torch.nn.Linear(5, 3)
    opt = deepgate.FlatAdam(m.parameters(), lr=1e-3)
    f = opt.flat_buffers()
    assert m.weight.data_ptr() == f['param'].data_ptr() and m.weight.grad.data_ptr() == f['grad'].data_ptr()
    m(torch.ones(2, 5)).sum().backward()
    assert float(f['grad'].abs().sum()) > 0           # autograd accumulated into the flat buffer
    opt.zero_grad()                                   # gradients are dropped: autograd assigns the next ones (no per-parameter add)
    assert m.weight.grad is None and m.bias.grad is None
    m(torch.ones(2, 5)).sum().backward()
    f['grad'].zero_()
    assert opt.reduce_gradients() == 1.0              # gathers them into the flat buffer with one multi-tensor copy (no process group: factor 1)
    assert torch.equal(f['grad'][:15].view(3, 5), m.weight.grad) and float(f['grad'].abs().sum()) > 0
    sd = opt.state_dict()
    ref = torch.optim.Adam(torch.nn.Linear(5, 3).parameters(), lr=1e-3).state_dict()
    assert set(ref['param_groups'][0]) <= set(sd['param_groups'][0]) | {'decoupled_weight_decay'}
    assert sd['param_groups'][0]['params'] == [0, 1]


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'multi-gate-vae_amd'))
import deepgate
from deepgate import synthetic as syn
from oracle import ref_cpu as R
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', init_method='env://')
z = np.load(os.path.join(sys.argv[1], 'tests', 'golden', 'g1_aig.npz'))

def shard_grads(r):
    p = R.params_from_npz(z)
    g = syn.collate([syn.make_graph('aig', 40, 5, 300 + r, n_inputs=5)])
    b = R.batch_from_arrays(lambda k: g[k])
    bn = {k: v.clone() for k, v in p.items() if 'running_' in k}
    ls = R.run_batch(p, 'aig', b, training=True, bn_state=bn, p_drop=0.0, s_rounds=2, t_rounds=2)
    R.weighted_loss(ls, [1.0, 4.0, 4.0]).backward()
    return p

p = shard_grads(rank)
names = [k for k, v in p.items() if v.requires_grad]
params = [torch.nn.Parameter(p[k].detach().clone()) for k in names]
for q, k in zip(params, names):
    q.grad = p[k].grad.clone() if p[k].grad is not None else torch.zeros_like(q)
opt = deepgate.FlatAdam(params, lr=1e-4)
scale = opt.reduce_gradients()
assert abs(scale - 1.0 / world) < 1e-12
if rank == 0:
    others = [shard_grads(r) for r in range(world)]
    for q, k in zip(params, names):
        mean = sum((o[k].grad if o[k].grad is not None else torch.zeros_like(o[k])) for o in others) / world
        assert torch.allclose(q.grad * scale, mean, rtol=1e-6, atol=1e-9), k
    print('EXCHANGE_OK')
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_gradient_exchange_is_the_mean_of_shard_gradients(tmp_path):
    """N>1 path on CPU: 2 processes over gloo, each with its own graphs; after the single flat
    all-reduce every rank holds the arithmetic mean of the per-shard oracle gradients."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(script), ROOT]
    env = dict(os.environ, OMP_NUM_THREADS='2')
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert 'EXCHANGE_OK' in out.stdout


def test_heavy_lists_and_segments_cover_the_long_lists_exactly(monkeypatch):
    """GraphPlan.heavy / heavy_segments (the lists taken out of the per-node kernels): with the thresholds lowered, every list
    longer than HEAVY_ROW is listed once, its segments tile [ptr[n], ptr[n+1]) in order, never-updated and updated nodes split as
    asked, and the updated ones come grouped by level."""
    import numpy as np
    import torch
    import deepgate
    from deepgate import synthetic as syn
    from deepgate.graph_plan import GraphPlan
    monkeypatch.setattr(GraphPlan, 'HEAVY_ROW', 2)
    monkeypatch.setattr(GraphPlan, 'HEAVY_SEG', 3)
    arrays = syn.collate([syn.make_graph('aig', 122, 7, 40 + i, n_inputs=10) for i in range(2)])
    N = arrays['num_nodes']
    plan = GraphPlan(torch.from_numpy(arrays['edge_index']), N)
    plan.set_levels(torch.from_numpy(arrays['gate']), torch.from_numpy(arrays['forward_level']), [1, 2])
    for rev in (False, True):
        ptr = plan.csr(rev)[0].numpy()
        deg = np.diff(ptr)
        K, nodes = plan.heavy(rev)
        assert K == int((deg > 2).sum()) and np.array_equal(nodes.numpy(), np.nonzero(deg > 2)[0])
        for kw in ({}, {'inactive_only': True}, {'active_by_level': True}):
            hv = plan.heavy_segments(rev, **kw)
            gs = plan.gslot.numpy()
            want = [n for n in np.nonzero(deg > 2)[0] if not kw or (gs[n] == 255) == ('inactive_only' in kw)]
            if not want:
                assert hv is None
                continue
            got = hv['nodes'].numpy().tolist()
            assert sorted(got) == sorted(want) and hv['K'] == len(want)
            nsp, e0, e1, sn = (hv[k].numpy() for k in ('node_seg_ptr', 'seg_e0', 'seg_e1', 'seg_node'))
            for k, n in enumerate(got):
                segs = range(nsp[k], nsp[k + 1])
                assert all(sn[s] == k for s in segs)
                assert e0[nsp[k]] == ptr[n] and e1[nsp[k + 1] - 1] == ptr[n + 1]
                assert all(e1[s] == e0[s + 1] for s in list(segs)[:-1]) and all(0 < e1[s] - e0[s] <= 3 for s in segs)
            if 'active_by_level' in kw:
                lv = plan.level.numpy()[got]
                assert np.all(np.diff(lv) >= 0)
                kp, sp = hv['lvl_k_ptr'], hv['lvl_seg_ptr']
                assert len(kp) == plan.num_levels + 1 and kp[-1] == hv['K'] and sp[-1] == hv['S']
                for level in range(plan.num_levels):
                    assert all(lv[k] == level for k in range(kp[level], kp[level + 1]))
                    assert sp[level] == nsp[kp[level]]


def test_level_caches_are_dropped_when_levels_are_set_again():
    """data.plan_of re-runs set_levels when a batch meets a model with another gate set: the per-plan caches derived from
    gslot / level (per-slot node lists, heavy segments, tagged neighbour arrays) must not survive that."""
    from deepgate.graph_plan import GraphPlan
    ei = torch.tensor([[0, 1, 2, 2, 3], [2, 2, 3, 4, 4]])
    gate = torch.tensor([0., 0., 1., 2., 1.])
    lv = torch.tensor([0, 0, 1, 2, 3])
    p = GraphPlan(ei, 5)
    p.set_levels(gate, lv, [1, 2])
    s1 = p.slot_nodes()
    assert [s.tolist() for s in s1] == [[2, 4], [3]]         # AND nodes, NOT nodes
    cid = torch.tensor([0, 0, 1, 2, 1], dtype=torch.int32)
    t1 = p.tagged_idx(False, cid)
    assert p.tagged_idx(False, cid) is t1                    # same class-id tensor object: cached
    cid2 = cid.clone()
    assert p.tagged_idx(False, cid2) is not t1               # another tensor (even at a reused address): rebuilt
    p.set_levels(gate, lv, [1])                              # gate 2 no longer has an aggregator
    assert '_slot_nodes' not in p.__dict__ and '_heavy_seg' not in p.__dict__ and '_tagged' not in p.__dict__
    assert [s.tolist() for s in p.slot_nodes()] == [[2, 4]]


def test_prefetcher_collates_like_collate_and_keeps_order():
    """deepgate/prefetch.py on a CPU device: same tensors as synthetic.collate, batches in submission order, plans built."""
    from deepgate import synthetic as syn
    from deepgate.prefetch import BatchPrefetcher
    graphs = [syn.make_graph('aig', 20 + 6 * (30 + 4 * i), 6, 50 + i, n_inputs=20) for i in range(6)]
    chunks = [graphs[0:3], graphs[3:6], graphs[1:4]]
    got = list(BatchPrefetcher(iter(chunks), 'cpu', gate_ids=[1, 2], workers=2))
    assert len(got) == 3
    for b, ch in zip(got, chunks):
        ref = syn.collate(ch)
        for k in ('x', 'edge_index', 'gate', 'forward_level', 'forward_index', 'prob', 'tt_pair_index', 'tt_sim', 'neg_edge_index'):
            assert torch.equal(getattr(b, k), torch.from_numpy(ref[k])), k
        assert b.graph_ptr.tolist() == ref['graph_ptr'].tolist() and b.num_graphs == 3
        assert b._mgv_plan.has_levels and b._mgv_plan.N == ref['num_nodes']
    skipped = next(iter(BatchPrefetcher(iter(chunks[:1]), 'cpu', skip=('neg_edge_index',))))
    assert not hasattr(skipped, 'neg_edge_index') and not hasattr(skipped, '_mgv_plan')


@pytest.mark.parametrize('ctype', ['aig', 'xmg'])
def test_quotient_colours_equal_brute_force_refinement(ctype):
    """GraphPlan.quotient (host logic of the structural encoder's quotient stages): the colours of half round t equal colour
    refinement done literally — (feature class, previous colour, sorted tuple of the neighbours' previous colours) over the in-CSR for odd t,
    the out-CSR for even t — and a representative's list names its neighbours' previous colours."""
    from deepgate import synthetic as syn
    from deepgate.graph_plan import GraphPlan
    a = syn.collate([syn.make_graph(ctype, 200 + 30 * 40, 40, 5 + i, n_inputs=200) for i in range(2)])
    N, ei = a['num_nodes'], a['edge_index']
    plan = GraphPlan(torch.from_numpy(ei), N)
    xcls = torch.from_numpy(a['x'][:, 1].astype('uint8'))
    old = GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES
    GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = 1.5, 1      # small graphs: let three half rounds qualify
    try:
        q = plan.quotient(xcls, 3)
    finally:
        GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = old
    assert len(q) >= 2
    col = [0] * N
    for t, s in enumerate(q, start=1):
        src, dst = (ei[1], ei[0]) if t % 2 == 0 else (ei[0], ei[1])
        nb = [[] for _ in range(N)]
        for u, v in zip(src.tolist(), dst.tolist()):
            nb[v].append(col[u])
        keys = [(int(xcls[i]), col[i], tuple(sorted(nb[i]))) for i in range(N)]
        ids = {}
        for kk in keys:
            ids.setdefault(kk, len(ids))
        cid = s['cid'].tolist()
        assert len(set(zip((ids[kk] for kk in keys), cid))) == len(ids) == s['C'], t          # the same partition
        # representatives' lists: entries = C + previous colour (in the numbering of stage t-1: q[t-2]['cid'], or 0)
        prev_cid = q[t - 2]['cid'].tolist() if t > 1 else [0] * N
        first = {}
        for i, c in enumerate(cid):
            first.setdefault(c, i)
        ptr, idx, own = s['ptr'].tolist(), s['idx'].tolist(), s['own'].tolist()
        for c in range(s['C']):
            r = first[c]
            assert own[c] == prev_cid[r]
            src_nodes = [u for u, v in zip(src.tolist(), dst.tolist()) if v == r]
            assert sorted(idx[ptr[c]:ptr[c + 1]]) == sorted(s['C'] + prev_cid[u] for u in src_nodes), (t, c)
        col = [ids[kk] for kk in keys]
    # segment tables of the per-colour sums: runs of <= 64 members of one colour, in colour order, down to one row per colour
    order, levels = q[-1]['sum_levels']
    cid = q[-1]['cid']
    assert sorted(order.tolist()) == list(range(N)) and bool((cid[order.long()][1:] >= cid[order.long()][:-1]).all())
    # run the tables the way mgv_seg_sum does (out[out_row[s]] = sum of the segment's items) on one number per node: every colour's
    # final row must be its member count, every segment <= 64 items, every row of the buffer written exactly once
    C = q[-1]['C']
    assert levels['C'] == C and levels['rows'] >= C
    buf = [None] * levels['rows']
    for li, (n_seg, sp, out_row, src_row) in enumerate(levels['levels']):
        sp = sp.tolist()
        rows = out_row.tolist() if out_row is not None else list(range(n_seg))
        assert len(sp) == n_seg + 1 and sp[0] == 0 and all(0 <= b - a_ <= 64 for a_, b in zip(sp[:-1], sp[1:]))
        src = [1] * N if li == 0 else buf[src_row:]
        assert sp[-1] == (N if li == 0 else len([v for v in src if v is not None]))
        for s_, r in enumerate(rows):
            assert buf[r] is None
            buf[r] = sum(src[m] for m in range(sp[s_], sp[s_ + 1]))
    assert all(v is not None for v in buf)
    assert buf[:C] == torch.bincount(cid.long(), minlength=C).tolist()


def test_quotient_stages_only_from_the_break_even_batch_size():
    """Below GraphPlan.QUOTIENT_MIN_NODES nodes no quotient stage is built (measured break-even: between 65,536 and 262,144 nodes)."""
    from deepgate.graph_plan import GraphPlan
    import deepgate.graph_plan as gp
    import importlib
    src = open(gp.__file__).read()
    assert 'QUOTIENT_MIN_NODES = 131072' in src              # (the session fixture lowers it for the tests: the shipped default)
    old = GraphPlan.QUOTIENT_MIN_NODES
    try:
        GraphPlan.QUOTIENT_MIN_NODES = 10 ** 9
        g = torch.Generator().manual_seed(1)
        n = 500
        src_n = torch.randint(0, n - 1, (900,), generator=g)
        dst_n = torch.minimum(src_n + 1 + torch.randint(0, 50, (900,), generator=g), torch.tensor(n - 1))
        plan = GraphPlan(torch.stack([src_n, dst_n]), n)
        assert plan.quotient(torch.zeros(n, dtype=torch.uint8), 4) == []
    finally:
        GraphPlan.QUOTIENT_MIN_NODES = old


def test_load_pretrained_without_shipped_weights_raises_file_not_found():
    """dg_ae_model_aig.py:157-160: `load_pretrained('')` resolves <package>/pretrained/model.pth and loads it; the weights are
    absent upstream too (.MISSING_LARGE_BLOBS), so the call ends in torch.load's FileNotFoundError — not a NameError."""
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=16, s_rounds=1, t_rounds=1, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=16)
    with pytest.raises(FileNotFoundError):
        model.load_pretrained('')
    with pytest.raises(FileNotFoundError):
        model.load_pretrained('/nonexistent/model.pth')


def test_quotient_cache_keeps_both_encoders_stage_counts():
    """--s_rounds != --t_rounds: the two encoders ask GraphPlan.quotient for different stage counts every step; neither request
    may evict the other (each rebuild costs sorts and host read-backs)."""
    g = syn.make_graph('aig', 4096, 16, 5, n_inputs=256)
    a = syn.collate([g])
    plan = GraphPlan(torch.from_numpy(a['edge_index']), a['num_nodes'])
    xcls = torch.from_numpy(a['x'][:, 1]).to(torch.uint8).contiguous()
    old = GraphPlan.QUOTIENT_MIN_NODES
    GraphPlan.QUOTIENT_MIN_NODES = 1024
    try:
        q2, q4 = plan.quotient(xcls, 2), plan.quotient(xcls, 4)
        assert plan.quotient(xcls, 2) is q2 and plan.quotient(xcls, 4) is q4
        assert len(q2) <= 2 and len(q4) >= len(q2)
        plan.warm(xcls, quotient_stages=[2, 4])
        assert plan.quotient(xcls, 2) is q2
    finally:
        GraphPlan.QUOTIENT_MIN_NODES = old


def test_persistent_sweep_roles_cover_every_slot_and_fit_the_grid():
    """GraphPlan.persist_roles: workgroup ranges per aggregator slot of the persistent sweep kernels — monotone, a slot with tiles
    owns a workgroup, never more workgroups than its widest level has tiles, never more than the grid; small plans get one tile per
    workgroup and level."""
    g = [syn.make_graph('xmg', 3000, 20, 31 + i, n_inputs=200) for i in range(3)]
    a = syn.collate(g)
    plan = GraphPlan(torch.from_numpy(a['edge_index']), a['num_nodes'])
    gates = [1, 2, 3, 4, 5]
    plan.set_levels(torch.from_numpy(a['gate']), torch.from_numpy(a['forward_level']), gates)
    T, L = len(gates), plan.num_levels
    ktp = np.asarray(plan._key_tile_ptr_host)
    assert ktp.shape[0] == L * T + 1 and ktp[-1] == plan.num_tiles and np.array_equal(ktp, plan.key_tile_ptr.numpy())
    assert np.array_equal(ktp[::T], np.asarray(plan.level_tile_ptr))            # level ranges are the slot ranges' union
    cnt = (ktp[1:] - ktp[:-1]).reshape(L, T)
    for grid in (256, 16, T):
        roles = plan.persist_roles(grid)
        w = np.diff(np.asarray(roles))
        assert roles[0] == 0 and roles[-1] <= grid and np.all(w >= (cnt.sum(0) > 0)) and np.all(w <= np.maximum(cnt.max(0), 0))
    assert np.array_equal(np.diff(np.asarray(plan.persist_roles(256))), cnt.max(0))     # everything fits: one tile per workgroup and level
    assert plan.persist_roles(T - 1) is None                                    # fewer workgroups than slots with tiles


def test_packed_sweep_rows_name_the_first_sources_and_consumers_of_every_updated_node():
    """GraphPlan.order_rows (the 128-byte rows the level kernels read): per updated node, in sweep order, the CSR spans, the first 4
    in-edge sources, the first 8 consumers with their slot in the consumer's in-list and the consumer's aggregator slot; -1 / 255
    beyond a node's lists."""
    g = [syn.make_graph('xmg', 12 + 12 * 40, 12, 7 + i, n_inputs=12) for i in range(2)]
    a = syn.collate(g)
    plan = GraphPlan(torch.from_numpy(a['edge_index']), a['num_nodes'])
    plan.set_levels(torch.from_numpy(a['gate']), torch.from_numpy(a['forward_level']), [1, 2, 3, 4, 5])
    rows = plan.order_rows.numpy()
    assert rows.shape == (plan.n_active, 32) and rows.dtype == np.int32
    ip, isrc, op, od, osl, gs = (getattr(plan, n).numpy() for n in ('in_ptr', 'in_src', 'out_ptr', 'out_dst', 'out_slot', 'gslot'))
    assert max(op[1:] - op[:-1]) > 8                         # some consumer list runs past the packed row
    for i, v in enumerate(plan.order.numpy()):
        r = rows[i]
        assert list(r[:4]) == [ip[v], ip[v + 1], op[v], op[v + 1]]
        want_in = list(isrc[ip[v]:ip[v + 1]][:4])
        assert list(r[4:8]) == want_in + [-1] * (4 - len(want_in))
        cons = list(range(op[v], op[v + 1]))[:8]
        pairs = [x for e in cons for x in (od[e], osl[e])]
        assert list(r[8:24]) == pairs + [-1] * (16 - len(pairs))
        gc = [gs[od[e]] for e in cons]
        assert list(r[24:26].view(np.uint8)) == gc + [255] * (8 - len(gc))
        assert np.all(r[26:] == -1)
        for e in cons:                                       # the pair really is that edge seen from the consumer
            assert isrc[osl[e]] == v


def _run_seg_tables(levels, items_value):
    """Run mgv_seg_sum's tables on one number per item: the buffer of `rows` sums (every row written exactly once)."""
    buf = [None] * levels['rows']
    for li, (n_seg, sp, out_row, src_row) in enumerate(levels['levels']):
        sp = sp.tolist()
        rows = out_row.tolist() if out_row is not None else list(range(n_seg))
        assert len(sp) == n_seg + 1 and sp[0] == 0 and all(0 <= b - a_ <= 64 for a_, b in zip(sp[:-1], sp[1:]))
        src = items_value if li == 0 else buf[src_row:]
        for s_, r in enumerate(rows):
            assert buf[r] is None
            buf[r] = sum(src[m] for m in range(sp[s_], sp[s_ + 1]))
    assert all(v is not None for v in buf)
    return buf


def test_batch_quotient_assembled_from_per_graph_stages():
    """GraphPlan.assemble_quotient: a batch's quotient stages put together from its graphs' own (cached) stages by index arithmetic.
    Per graph the colours equal brute-force colour refinement; colours are never shared between graphs; a representative's list
    names its neighbours' previous colours in the BATCH numbering; the segment tables (own / ent / final sums) add up."""
    graphs = [syn.make_graph('xmg', 120 + 25 * 30, 30, 50 + i, n_inputs=120) for i in range(3)]
    graphs.append(syn.make_graph('xmg', 120 + 25 * 30, 30, 50, n_inputs=120))          # a copy of graph 0: still its own colours
    a = syn.collate(graphs)
    N, ei = a['num_nodes'], a['edge_index']
    node_off = a['graph_ptr'].tolist()
    old = GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES
    GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = 1.3, 1
    try:
        parts = []
        for g in graphs:
            xc = torch.from_numpy(g['x'][:, 1].astype('uint8'))
            parts.append(GraphPlan(torch.from_numpy(g['edge_index']), g['num_nodes']).quotient(xc, 3, force=True))
        plan = GraphPlan(torch.from_numpy(ei), N)
        plan.xcls = torch.from_numpy(a['x'][:, 1].astype('uint8'))
        out = plan.assemble_quotient(parts, node_off, [2, 3])
        assert plan.quotient(plan.xcls, 3) is out[3] and plan.quotient(plan.xcls, 2) is out[2]      # installed as the plan's cache
    finally:
        GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = old
    q = out[3]
    assert len(q) >= 2 and len(out[2]) == 2 and 'sum_levels' in out[2][-1] and 'sum_levels' in q[-1]
    xcls = plan.xcls
    gid = np.repeat(np.arange(len(graphs)), np.diff(node_off))
    col = [0] * N
    for t, s in enumerate(q, start=1):
        src, dst = (ei[1], ei[0]) if t % 2 == 0 else (ei[0], ei[1])
        nb = [[] for _ in range(N)]
        for u, v in zip(src.tolist(), dst.tolist()):
            nb[v].append(col[u])
        keys = [(int(gid[i]), int(xcls[i]), col[i], tuple(sorted(nb[i]))) for i in range(N)]      # the graph id is part of the colour
        ids = {}
        for kk in keys:
            ids.setdefault(kk, len(ids))
        cid = s['cid'].tolist()
        assert len(set(zip((ids[kk] for kk in keys), cid))) == len(ids) == s['C'], t
        prev_cid = q[t - 2]['cid'].tolist() if t > 1 else [0] * N
        Cp = q[t - 2]['C'] if t > 1 else 1
        first = {}
        for i, c in enumerate(cid):
            first.setdefault(c, i)
        ptr, idx, ent, own, xc = s['ptr'].tolist(), s['idx'].tolist(), s['ent_idx'].tolist(), s['own'].tolist(), s['xcls'].tolist()
        for c in range(s['C']):
            members = [i for i in range(N) if cid[i] == c]
            assert len({(prev_cid[i], int(xcls[i])) for i in members}) == 1 and xc[c] == int(xcls[members[0]]) and own[c] == prev_cid[members[0]]
            r = members[0]
            want = sorted(prev_cid[u] for u, v in zip(src.tolist(), dst.tolist()) if v == r)
            assert sorted(ent[ptr[c]:ptr[c + 1]]) == want and sorted(idx[ptr[c]:ptr[c + 1]]) == [s['C'] + w for w in want], (t, c)
        # colour-level sums for stage t-1: per previous colour, the representatives that own it / the entries that name it
        buf = _run_seg_tables(s['own_levels'], [1] * s['C'])
        assert s['own_levels']['C'] == Cp and sorted(s['own_rows'].tolist()) == list(range(s['C']))
        assert buf[:Cp] == np.bincount(np.asarray(own), minlength=Cp).tolist()
        n_ent = ptr[-1]
        if n_ent:
            rows_of_entries = s['ent_rows'].tolist()
            owner = np.repeat(np.arange(s['C']), np.diff(ptr))
            # ent_rows[k] = the colour whose list holds the k-th entry in previous-colour order: a permutation of the entries' owners
            assert sorted(rows_of_entries) == sorted(owner.tolist())
            buf = _run_seg_tables(s['ent_levels'], [1] * n_ent)
            assert buf[:Cp] == np.bincount(np.asarray(ent[:n_ent]), minlength=Cp).tolist()
        col = [ids[kk] for kk in keys]
    for lst in (out[2], out[3]):
        order, levels = lst[-1]['sum_levels']
        cidl = lst[-1]['cid']
        assert sorted(order.tolist()) == list(range(N)) and bool((cidl[order.long()][1:] >= cidl[order.long()][:-1]).all())
        assert _run_seg_tables(levels, [1] * N)[:lst[-1]['C']] == torch.bincount(cidl.long(), minlength=lst[-1]['C']).tolist()


def test_batch_quotient_merged_through_the_dataset_wide_colour_dictionary():
    """GraphPlan.assemble_quotient_merged + ColourDictionary: the graphs' own colours merged by their dataset-wide global ids give
    exactly the batch-level colour refinement (the partition brute force finds on the whole batch, colours shared ACROSS graphs —
    a copy of a graph adds no colour), the representatives' lists name merged previous colours, the segment tables add up."""
    from deepgate.graph_plan import ColourDictionary
    graphs = [syn.make_graph('xmg', 120 + 25 * 30, 30, 70 + i, n_inputs=120) for i in range(3)]
    graphs.append(syn.make_graph('xmg', 120 + 25 * 30, 30, 70, n_inputs=120))          # a copy of graph 0
    a = syn.collate(graphs)
    N, ei = a['num_nodes'], a['edge_index']
    node_off = a['graph_ptr'].tolist()
    old = GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES
    GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = 1.3, 1
    try:
        cd = ColourDictionary()
        parts, gcols = [], []
        for g in graphs:
            xc = torch.from_numpy(g['x'][:, 1].astype('uint8'))
            st = GraphPlan(torch.from_numpy(g['edge_index']), g['num_nodes']).quotient(xc, 3, force=True)
            parts.append(st)
            host = [dict(ptr=s_['raw']['rptr'].numpy(), ent=s_['raw']['ent'].numpy(), own=s_['raw']['own'].numpy(), xcls=s_['xcls'].numpy()) for s_ in st]
            gcols.append([torch.from_numpy(v) for v in cd.globals_of(host)])
        assert all(torch.equal(x, y) for x, y in zip(gcols[0], gcols[3]))               # the copy: the same global colours
        plan = GraphPlan(torch.from_numpy(ei), N)
        plan.xcls = torch.from_numpy(a['x'][:, 1].astype('uint8'))
        out = plan.assemble_quotient_merged(parts, gcols, node_off, [2, 3])
        assert plan.quotient(plan.xcls, 3) is out[3]
        ref = GraphPlan(torch.from_numpy(ei), N).quotient(plan.xcls, 3)                # the batch-level refinement
    finally:
        GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = old
    q = out[3]
    assert len(q) >= 2 and len(q) == len(ref) and [s['C'] for s in q] == [s['C'] for s in ref]
    xcls = plan.xcls
    col = [0] * N
    for t, s in enumerate(q, start=1):
        src, dst = (ei[1], ei[0]) if t % 2 == 0 else (ei[0], ei[1])
        nb = [[] for _ in range(N)]
        for u, v in zip(src.tolist(), dst.tolist()):
            nb[v].append(col[u])
        keys = [(int(xcls[i]), col[i], tuple(sorted(nb[i]))) for i in range(N)]        # no graph id: colours are shared
        ids = {}
        for kk in keys:
            ids.setdefault(kk, len(ids))
        cid = s['cid'].tolist()
        assert len(set(zip((ids[kk] for kk in keys), cid))) == len(ids) == s['C'], t
        assert len(set(zip(cid, ref[t - 1]['cid'].tolist()))) == s['C']                 # the same partition as GraphPlan.quotient
        prev_cid = q[t - 2]['cid'].tolist() if t > 1 else [0] * N
        Cp = q[t - 2]['C'] if t > 1 else 1
        ptr, ent, own, xc = s['ptr'].tolist(), s['ent_idx'].tolist(), s['own'].tolist(), s['xcls'].tolist()
        first = {}
        for i, c in enumerate(cid):
            first.setdefault(c, i)
        for c in range(s['C']):
            r = first[c]
            assert xc[c] == int(xcls[r]) and own[c] == prev_cid[r]
            assert sorted(ent[ptr[c]:ptr[c + 1]]) == sorted(prev_cid[u] for u, v in zip(src.tolist(), dst.tolist()) if v == r), (t, c)
        assert _run_seg_tables(s['own_levels'], [1] * s['C'])[:Cp] == np.bincount(np.asarray(own), minlength=Cp).tolist()
        if ptr[-1]:
            assert _run_seg_tables(s['ent_levels'], [1] * ptr[-1])[:Cp] == np.bincount(np.asarray(ent[:ptr[-1]]), minlength=Cp).tolist()
        col = [ids[kk] for kk in keys]
    order, levels = q[-1]['sum_levels']
    assert _run_seg_tables(levels, [1] * N)[:q[-1]['C']] == torch.bincount(q[-1]['cid'].long(), minlength=q[-1]['C']).tolist()


def test_self_loop_count_and_key_bits():
    """GraphPlan.count_self_loops (what the device negative sampler subtracts from its draw count; cached on the plan so that a fresh
    batch's first step does not read it back in the middle of the step) and GraphPlan._key_bits (bits of the grouping key a colour
    refinement stage sorts on: at least 24, at most 63, growing with the colours to expect, never below what N nodes need)."""
    ei = torch.tensor([[0, 1, 2, 2, 3, 4, 4], [1, 1, 2, 3, 3, 0, 4]])          # self loops: (1,1), (2,2), (3,3), (4,4)
    plan = GraphPlan(ei, 6)
    assert plan.count_self_loops() == 4 and plan.num_self_loops == 4
    assert GraphPlan(torch.zeros(2, 0, dtype=torch.long), 3).count_self_loops() == 0
    big = GraphPlan(torch.tensor([[0], [1]]), 1 << 22)
    bits = [big._key_bits(c) for c in (1, 3, 152, 34377, 1 << 21)]
    assert bits == sorted(bits) and bits[0] >= 24 and bits[-1] == 63 and bits[0] < 48
    assert GraphPlan(torch.tensor([[0], [1]]), 100)._key_bits(1 << 20) == 24 + 10      # few nodes: never more colours than nodes (2 * 7 bits + 20)
