"""GPU tests of the on-device batch builder (csrc/plan_build.hip): every array of the HIP-built GraphPlan
equals the torch-built one (stable sorts by destination / source / (level, slot)), device ASAP levels equal the reference's
`return_order_info` output (tests/golden/g6_loader.npz, utils/dag_utils.py:10-37,80-88) and the host levelisation, and the
building blocks (scan, stable counting sort) hold at ragged sizes."""
import os
import time

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _plans(ei, N, monkeypatch):
    from deepgate.graph_plan import GraphPlan
    monkeypatch.setenv('MGV_PLAN', 'torch')
    ref = GraphPlan(ei, N)
    monkeypatch.setenv('MGV_PLAN', 'hip')
    got = GraphPlan(ei, N)
    assert got.hip and not ref.hip
    return ref, got


def _same(ref, got, names):
    for n in names:
        a, b = getattr(ref, n), getattr(got, n)
        if torch.is_tensor(a):
            assert a.shape == b.shape and a.dtype == b.dtype, (n, a.shape, b.shape, a.dtype, b.dtype)
            assert torch.equal(a, b), n
        else:
            assert a == b, (n, a, b)


CSR = ['in_ptr', 'in_src', 'in_dst', 'out_ptr', 'out_dst', 'out_slot']
LEVELS = ['gslot', 'level', 'num_levels', 'order', 'order_span', 'order_rows', 'tile_start', 'tile_count', 'tile_slot', 'slot_tiles', 'slot_tile_ptr',
          'level_tile_ptr', 'num_tiles', 'n_active', 'num_slots']


def _random_multigraph(rng, N, E, hub=None, hub_deg=0):
    src = rng.integers(0, N, size=E)
    dst = rng.integers(0, N, size=E)
    if hub is not None:                              # one node with a very long out-list and another with a long in-list
        src[:hub_deg] = hub
        dst[E - hub_deg // 2:] = (hub + 1) % N
    return np.stack([src, dst]).astype(np.int64)


@pytest.mark.parametrize('case', ['tiny', 'empty', 'dupes', 'hub_bitonic', 'hub_rank', 'ragged'])
def test_csr_equals_stable_sorts(case, monkeypatch):
    dev = _dev()
    rng = np.random.Generator(np.random.PCG64(11))
    if case == 'tiny':
        N, ei = 1, np.zeros((2, 0), dtype=np.int64)
    elif case == 'empty':
        N, ei = 777, np.zeros((2, 0), dtype=np.int64)
    elif case == 'dupes':
        N, ei = 50, _random_multigraph(rng, 50, 4000)              # many parallel edges: every list longer than the register path
    elif case == 'hub_bitonic':
        N, ei = 3000, _random_multigraph(rng, 3000, 9000, hub=7, hub_deg=3500)
    elif case == 'hub_rank':
        N, ei = 2049, _random_multigraph(rng, 2049, 20000, hub=2048, hub_deg=9000)      # > 4096: rank-sort path
    else:
        N, ei = 100003, _random_multigraph(rng, 100003, 263111)
    ref, got = _plans(torch.from_numpy(ei).to(dev), N, monkeypatch)
    got._check_status()
    _same(ref, got, CSR)


def test_out_of_range_node_ids_are_reported(monkeypatch):
    dev = _dev()
    from deepgate.graph_plan import GraphPlan
    monkeypatch.setenv('MGV_PLAN', 'hip')
    p = GraphPlan(torch.tensor([[0, 1, 9], [1, 2, 0]], device=dev), 3)
    # the CSR kernels skip the bad edge: two valid slots, the unfilled tail zeroed, every stored id inside [0, N)
    assert p.in_ptr.tolist() == [0, 0, 1, 2] and p.out_ptr.tolist() == [0, 1, 2, 2]
    for arr in (p.in_src, p.in_dst, p.out_dst, p.out_slot):
        assert int(arr.min()) >= 0 and int(arr.max()) < 3 and int(arr[2]) == 0
    with pytest.raises(ValueError):
        p._check_status()
    # the status is read BEFORE anything consumes the CSR
    p2 = GraphPlan(torch.tensor([[0, 1, -4], [1, 2, 0]], device=dev), 3)
    with pytest.raises(ValueError, match='outside'):
        p2.set_levels(torch.tensor([0., 1., 1.], device=dev), torch.tensor([0, 1, 2], device=dev), [1, 2])


def test_out_of_range_pairs_and_negatives_raise():
    dev = _dev()
    from deepgate import ops, sampling
    with pytest.raises(ValueError, match='tt_pair_index'):
        ops.pair_lists(torch.tensor([[0, 1, 7], [1, 2, 0]], device=dev), 3)
    with pytest.raises(ValueError, match='neg_edge_index'):
        sampling.bucket_negatives(torch.tensor([[0, 1, 2], [1, 5, 0]], device=dev), 3)


@pytest.mark.parametrize('ctype', ['aig', 'mig', 'xag', 'xmg'])
def test_level_buckets_tiles_and_first_stage_classes_equal_the_torch_plan(ctype, monkeypatch):
    dev = _dev()
    import deepgate
    from deepgate import synthetic as syn
    model_mod = getattr(deepgate, 'dg_ae_model_' + ctype)
    gate_ids = [g for _, g in model_mod.Model.GATES]
    # ragged batch: graphs of different depth and width, so levels hold mixed gate types and partial tiles
    graphs = [syn.make_graph(ctype, P + L * per, L, 300 + L, n_inputs=P) for P, L, per in ((200, 10, 480), (300, 17, 300), (400, 24, 210))]
    arrays = syn.collate(graphs)
    ei = torch.from_numpy(arrays['edge_index']).to(dev)
    gate = torch.from_numpy(arrays['gate']).to(dev)
    lv = torch.from_numpy(arrays['forward_level']).to(dev)
    ref, got = _plans(ei, arrays['num_nodes'], monkeypatch)
    monkeypatch.setenv('MGV_PLAN', 'torch')
    ref.set_levels(gate, lv, gate_ids)
    monkeypatch.setenv('MGV_PLAN', 'hip')
    got.set_levels(gate, lv, gate_ids)
    _same(ref, got, CSR + LEVELS)
    xcls = torch.from_numpy(arrays['x'][:, 1].astype('uint8')).to(dev)
    a, b = ref.first_stage_classes(xcls), got.first_stage_classes(xcls)
    assert a[1] == b[1]
    for u, v in zip((a[0], a[2], a[3], a[4]), (b[0], b[2], b[3], b[4])):
        assert torch.equal(u, v)


def test_bad_levels_are_rejected(monkeypatch):
    dev = _dev()
    from deepgate.graph_plan import GraphPlan
    monkeypatch.setenv('MGV_PLAN', 'hip')
    p = GraphPlan(torch.tensor([[0, 1], [1, 2]], device=dev), 3)
    with pytest.raises(ValueError):
        p.set_levels(torch.tensor([[0.], [1.], [1.]], device=dev), torch.tensor([0, 1, 1], device=dev), [1, 2])


def test_device_levels_equal_the_reference_levelisation(monkeypatch):
    dev = _dev()
    from deepgate.graph_plan import GraphPlan
    from deepgate.parser import forward_levels
    monkeypatch.setenv('MGV_PLAN', 'hip')
    z = np.load(os.path.join(GOLDEN, 'g6_loader.npz'))
    for tag in ('aig', 'xmg'):                       # shuffled node ids; levels written by the reference's return_order_info
        ei = z[tag + '_edge_index']
        n = z[tag + '_x'].shape[0]
        got = GraphPlan(torch.from_numpy(ei).to(dev), n).asap_levels().cpu().numpy()
        assert np.array_equal(got, z[tag + '_forward_level'].astype(np.int64)), tag
        back = GraphPlan(torch.from_numpy(np.ascontiguousarray(ei[::-1])).to(dev), n).asap_levels().cpu().numpy()
        assert np.array_equal(back, z[tag + '_backward_level'].astype(np.int64)), tag
    # a deep chain (more levels than the first batch of rounds) with side branches, and a large random DAG
    rng = np.random.Generator(np.random.PCG64(5))
    n = 3000
    chain = np.stack([np.arange(n - 1), np.arange(1, n)])
    extra = np.sort(rng.integers(0, n, size=(2, 4000)), axis=0)
    extra = extra[:, extra[0] != extra[1]]
    ei = np.concatenate([chain, extra], axis=1)
    got = GraphPlan(torch.from_numpy(ei).to(dev), n).asap_levels().cpu().numpy()
    assert np.array_equal(got, forward_levels(ei, n)) and got.max() == n - 1
    with pytest.raises(ValueError):
        GraphPlan(torch.tensor([[0, 1, 2], [1, 2, 0]], device=dev), 4).asap_levels()


def test_a_batch_without_host_levels_trains_like_one_with(monkeypatch):
    """NpzParser(levelise=False) graphs: the plan levelises on the device; same losses as with the host levels."""
    dev = _dev()
    import types
    import deepgate
    from deepgate import synthetic as syn
    graphs = [syn.make_graph('aig', 1024, 30, 40 + i, n_inputs=64) for i in range(2)]
    full = syn.collate(graphs)
    bare = syn.collate([{k: v for k, v in g.items() if k not in ('forward_level', 'forward_index')} for g in graphs])
    assert 'forward_level' not in bare
    torch.manual_seed(0)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=2, t_rounds=2, layernorm=True)
    model = deepgate.dg_ae_model_aig.Model(struct_encoder=enc, dim_hidden=64).to(dev).eval()
    tr = deepgate.Trainer(types.SimpleNamespace(model='DG_AE'), model, training_id='lv', save_dir='/tmp/mgv_plan_test', lr=1e-4,
                          rc_prob_func_weight=[1.0, 4.0, 4.0], device='cuda:0', batch_size=2, distributed=False)
    with torch.no_grad():
        a = tr.run_batch(deepgate.CircuitBatch.from_arrays(full, device=dev))
        b = tr.run_batch(deepgate.CircuitBatch.from_arrays(bare, device=dev))
    for k in ('recon_loss', 'prob_loss', 'func_loss'):
        assert float(a[k]) == float(b[k]), k


def test_scan_and_counting_sort_building_blocks():
    dev = _dev()
    from deepgate import _hip
    from deepgate._hip import ptr
    rng = np.random.Generator(np.random.PCG64(3))
    for n in (0, 1, 2047, 2048, 2049, 1 << 20, 3_000_001):
        x = torch.from_numpy(rng.integers(0, 5, size=n).astype(np.int32)).to(dev)
        out = torch.empty(n + 1, dtype=torch.int32, device=dev)
        scratch = torch.empty(n // 2048 + 2, dtype=torch.int32, device=dev)
        _hip.call('mgv_scan_exclusive_i32', n, ptr(x), ptr(out), ptr(scratch))
        ref = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        ref[1:] = torch.cumsum(x.long(), 0)
        assert torch.equal(out.long(), ref), n
    for n, K in ((0, 3), (5, 1), (70000, 7), (1_234_567, 1205), (4096, 8192)):
        key = torch.from_numpy(rng.integers(-1, K, size=n).astype(np.int32)).to(dev)
        order = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        ks = torch.empty(K + 1, dtype=torch.int32, device=dev)
        ns = _hip.call_value('mgv_count_sort_scratch_ints', n, K)
        scratch = torch.empty(ns, dtype=torch.int32, device=dev)
        _hip.call('mgv_count_sort_i32', n, ptr(key), K, ptr(order), ptr(ks), ptr(scratch), ns)
        keep = torch.nonzero(key >= 0).reshape(-1)
        ref = keep[torch.sort(key[keep].long(), stable=True).indices]
        m = int(ks[K].item())
        assert m == keep.numel() and torch.equal(order[:m].long(), ref), (n, K)
        cnt = torch.bincount(key[keep].long(), minlength=K)
        assert torch.equal((ks[1:] - ks[:-1]).long(), cnt)


def test_plan_build_time_at_config_2():
    """Not a parity test: records the cold and warm plan-build times at BASELINE config 2 (795 ms cold with torch ops)."""
    dev = _dev()
    import deepgate
    from deepgate import synthetic as syn
    from deepgate.data import plan_of
    arrays = syn.make_batch(2, batch=16)
    times = []
    for _ in range(3):
        b = deepgate.CircuitBatch.from_arrays(arrays, device=dev)
        torch.cuda.synchronize()
        t0 = time.time()
        plan_of(b, [1, 2])
        torch.cuda.synchronize()
        times.append((time.time() - t0) * 1e3)
    print('plan build, 16 x 65,536-node graphs: cold %.1f ms, warm %.2f ms' % (times[0], min(times[1:])))
    assert min(times[1:]) < 50.0


@pytest.mark.gpu
def test_device_colour_refinement_equals_the_host_one():
    """GraphPlan.quotient on the device (mgv_colour_keys / mgv_colour_check + torch sorts) against the torch-only host path: the same
    partition of the nodes per half round; twin hubs with 80 consumers each (lists beyond the kernel's pairwise check: the sort-based
    check decides) land in one colour."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate import synthetic as syn
    from deepgate.graph_plan import GraphPlan
    a = syn.collate([syn.make_graph('aig', 256 + 64 * 40, 40, 21 + i, n_inputs=256) for i in range(3)])
    ei, n = a['edge_index'], a['num_nodes']
    lv = a['forward_level']
    later = np.nonzero(lv >= 2)[0]
    rng = np.random.Generator(np.random.PCG64(1))
    extra = []
    for hub in (3, 7):                                   # two primary inputs that drive 80 gates each
        extra.append(np.stack([np.full(80, hub), rng.choice(later, size=80, replace=False)]))
    ei = np.unique(np.concatenate([ei] + extra, axis=1), axis=1)
    xcls = torch.from_numpy(a['x'][:, 1].astype('uint8'))
    old = GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES
    GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = 1.5, 1
    try:
        host = GraphPlan(torch.from_numpy(ei), n).quotient(xcls, 3)
        dev = GraphPlan(torch.from_numpy(ei).cuda(), n).quotient(xcls.cuda(), 3)
    finally:
        GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = old
    assert len(host) == len(dev) >= 2
    for h, d in zip(host, dev):
        assert h['C'] == d['C']
        pairs = set(zip(h['cid'].tolist(), d['cid'].cpu().tolist()))
        assert len(pairs) == h['C']                      # one-to-one: the same partition (numbering may differ)


def _same_tables(a, b, name):
    assert a['C'] == b['C'] and a['rows'] == b['rows'] and len(a['levels']) == len(b['levels']), name
    for (na, spa, ora, sra), (nb, spb, orb, srb) in zip(a['levels'], b['levels']):
        assert na == nb and sra == srb and torch.equal(spa, spb), name
        assert (ora is None) == (orb is None) and (ora is None or torch.equal(ora, orb)), name


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['aig_hubs', 'xmg', 'long_runs'])
def test_device_built_stage_tables_equal_the_torch_composition(case):
    """GraphPlan._quotient_dev (every table of a refinement stage by the plan builder's kernels: radix sort of the keys, runs ->
    colours, representatives' lists, stable sort by previous colour, segment tables level by level) against the torch composition
    of the same tables in GraphPlan.quotient, on the same device plan: the same colour NUMBERING and every table equal, field by
    field.  Cases: twin hubs whose lists go to the sort-based check; five gate types; few colours with hundreds of thousands of
    members (several table levels, partial rows, the radix path of the stable sort)."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate import synthetic as syn
    from deepgate.graph_plan import GraphPlan
    dev = torch.device('cuda:0')
    stages_n, frac = 4, 1.5
    if case == 'aig_hubs':
        a = syn.collate([syn.make_graph('aig', 256 + 64 * 40, 40, 21 + i, n_inputs=256) for i in range(3)])
        ei = a['edge_index']
        later = np.nonzero(a['forward_level'] >= 2)[0]
        rng = np.random.Generator(np.random.PCG64(1))
        extra = [np.stack([np.full(80, hub), rng.choice(later, size=80, replace=False)]) for hub in (3, 7)]
        ei = np.unique(np.concatenate([ei] + extra, axis=1), axis=1)
    elif case == 'xmg':
        a = syn.collate([syn.make_graph('xmg', 300 + 50 * 60, 50, 5 + i, n_inputs=300) for i in range(4)])
        ei = a['edge_index']
    else:
        a = syn.collate([syn.make_graph('aig', 4096 + 30 * 8192, 30, 3 + i, n_inputs=4096) for i in range(2)])
        ei, stages_n, frac = a['edge_index'], 5, 1.2
    n = a['num_nodes']
    xcls = torch.from_numpy(a['x'][:, 1].astype('uint8')).to(dev)
    plan = GraphPlan(torch.from_numpy(ei).to(dev), n)
    assert plan.hip
    old = GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES, GraphPlan.QUOTIENT_DEVICE
    GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES = frac, 1
    try:
        GraphPlan.QUOTIENT_DEVICE = True
        got = plan.quotient(xcls, stages_n)
        plan._quotient = None
        GraphPlan.QUOTIENT_DEVICE = False
        ref = plan.quotient(xcls, stages_n)
    finally:
        GraphPlan.QUOTIENT_FRACTION, GraphPlan.QUOTIENT_MIN_NODES, GraphPlan.QUOTIENT_DEVICE = old
    assert len(got) == len(ref) >= (2 if case == 'xmg' else 3)
    if case == 'long_runs':
        assert len(ref[0]['own_levels']['levels']) + len(ref[-1]['sum_levels'][1]['levels']) > 2       # multi-level tables are in play
    for k, (g, r) in enumerate(zip(got, ref)):
        assert g['C'] == r['C'] and g['rev'] == r['rev'] and g['heavy'][0] == r['heavy'][0], k
        for name in ('cid', 'ptr', 'idx', 'ent_idx', 'own', 'own32', 'xcls', 'own_rows', 'ent_rows'):
            assert g[name].dtype == r[name].dtype and torch.equal(g[name], r[name]), (k, name)
        assert torch.equal(g['heavy'][1], r['heavy'][1]), k
        _same_tables(g['own_levels'], r['own_levels'], (k, 'own_levels'))
        _same_tables(g['ent_levels'], r['ent_levels'], (k, 'ent_levels'))
        assert ('sum_levels' in g) == ('sum_levels' in r) == (k == len(ref) - 1)
    assert torch.equal(got[-1]['sum_levels'][0], ref[-1]['sum_levels'][0])
    _same_tables(got[-1]['sum_levels'][1], ref[-1]['sum_levels'][1], 'sum_levels')


@pytest.mark.gpu
def test_colour_check_rejects_a_wrong_grouping():
    """mgv_colour_check is what makes the quotient stages exact: a grouping that merges nodes with different neighbour-colour
    multisets (what a collision of the 64-bit grouping key would produce) must be flagged; the true grouping must pass."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate import _hip
    from deepgate._hip import ptr
    from deepgate.graph_plan import GraphPlan
    dev = torch.device('cuda:0')
    # nodes 0..3 inputs with colours 0,0,1,1; nodes 4..7: 4 <- {0,2}, 5 <- {1,3} (same multiset {0,1}), 6 <- {0,1} ({0,0}), 7 <- {2,3} ({1,1})
    ei = torch.tensor([[0, 2, 1, 3, 0, 1, 2, 3], [4, 4, 5, 5, 6, 6, 7, 7]], device=dev)
    plan = GraphPlan(ei, 8)
    prev = torch.tensor([0, 0, 1, 1, 2, 2, 2, 2], dtype=torch.int32, device=dev)
    xcls = torch.zeros(8, dtype=torch.uint8, device=dev)

    def check(cid, rep):
        flags = torch.zeros(2, dtype=torch.int32, device=dev)
        cid_t, rep_t = torch.tensor(cid, dtype=torch.int32, device=dev), torch.tensor(rep, dtype=torch.int32, device=dev)      # (kept alive across the launch)
        _hip.call('mgv_colour_check', 8, ptr(plan.in_ptr), ptr(plan.in_src), ptr(prev), ptr(xcls), ptr(cid_t), ptr(rep_t), ptr(flags))
        return flags.tolist()
    good = ([0, 0, 1, 1, 2, 2, 3, 4], [0, 2, 4, 6, 7])            # 4 and 5 share a colour, 6 and 7 have their own
    assert check(*good) == [0, 0]
    assert check([0, 0, 1, 1, 2, 2, 2, 3], [0, 2, 4, 7])[0] == 1   # 6 ({0,0}) merged with 4 ({0,1}): same degree, another multiset
    assert check([0, 0, 0, 1, 2, 2, 3, 4], [0, 3, 4, 6, 7])[0] == 1   # node 2 (previous colour 1) merged with nodes of previous colour 0


@pytest.mark.gpu
@pytest.mark.parametrize('n,K', [(1, 1), (5000, 3), (70000, 152), (300000, 8192), (300000, 40000), (0, 17)])
def test_device_stable_sort_by_key_equals_torch(n, K):
    """GraphPlan._sort_by_key_dev (counting sort of the tile builder, or rocPRIM radix sort over the keys' bits with the permutation as
    int32): the order of torch's stable sort and the members per key, for key ranges on both sides of the switch."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate.graph_plan import GraphPlan
    dev = torch.device('cuda:0')
    plan = GraphPlan(torch.tensor([[0], [1]], device=dev), 2)
    g = torch.Generator(device=dev)
    g.manual_seed(n + K)
    keys = torch.randint(0, K, (n,), generator=g, device=dev, dtype=torch.int32)
    if n > 10:
        keys[: n // 3] = keys[0]                         # one long run
    order, counts = plan._sort_by_key_dev(keys, n, K)
    ref = torch.sort(keys.long(), stable=True)
    assert order.dtype == torch.int32 and torch.equal(order.long(), ref.indices)
    assert torch.equal(counts.long(), torch.bincount(keys.long(), minlength=K))


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['small', 'wide', 'deep', 'zeros'])
def test_device_segment_tables_equal_the_torch_ones(case):
    """GraphPlan._class_sum_levels_dev against class_sum_levels (the torch composition) from the same members-per-colour counts: every
    level's segment pointers, output rows and source rows.  'wide': more than 65,536 colours (the multi-kernel scan path); 'deep': a
    colour with 64^2 < members (three levels); 'zeros': colours nobody carries keep their zero row."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from deepgate.graph_plan import GraphPlan
    dev = torch.device('cuda:0')
    plan = GraphPlan(torch.tensor([[0], [1]], device=dev), 2)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    if case == 'small':
        counts = torch.tensor([3, 70, 1, 64, 65, 4096, 4097], device=dev)
    elif case == 'wide':
        counts = torch.randint(0, 6, (150000,), generator=g, device=dev)
        counts[::5000] = 700
    elif case == 'deep':
        counts = torch.tensor([5000, 2, 300000, 64 * 64, 64 * 64 + 1], device=dev)
    else:
        counts = torch.tensor([0, 0, 5, 0, 129, 0], device=dev)
    C = int(counts.numel())
    cid = torch.repeat_interleave(torch.arange(C, device=dev), counts)
    order = torch.arange(cid.numel(), device=dev)
    _, ref = plan.class_sum_levels(cid, C, presorted=(order, counts.long()))
    got = plan._class_sum_levels_dev(counts.to(torch.int32).contiguous(), C)
    _same_tables(got, ref, case)
    if case == 'deep':
        assert len(ref['levels']) >= 3
