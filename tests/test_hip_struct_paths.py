"""GPU cross-checks of the struct-stage code paths on a graph with hub nodes (fan-out 3000, fan-in 1500): the neighbour
index list of a hub's tile does not fit the LDS staging buffer, so the kernels take their generic per-row path there
while the other tiles take the chunked one.  Compared: bf16x3 kernels vs exact-fp32 kernels (independent code), and the
(degree, class)-table first half round vs the per-node launch (which must agree to rounding: same kernel arithmetic)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _hub_graph(rng, n=6000):
    src, dst = [], []
    for v in range(64, n):                                  # a random DAG by node order
        for p in rng.choice(v, size=int(rng.integers(1, 4)), replace=False):
            src.append(int(p)); dst.append(v)
    for v in rng.choice(np.arange(64, n), size=3000, replace=False):     # fan-out hub: node 3
        src.append(3); dst.append(int(v))
    for p in rng.choice(np.arange(0, n - 1), size=1500, replace=False):  # fan-in hub: the last node
        src.append(int(p)); dst.append(n - 1)
    return np.unique(np.array([src, dst], dtype=np.int64), axis=1), n


def _run(dev, ei, n, precision, first_stage):
    import deepgate
    from deepgate import ops
    old = (ops.PRECISION, ops.FIRST_STAGE_TABLE)
    ops.PRECISION, ops.FIRST_STAGE_TABLE = precision, first_stage == 'table'
    try:
        torch.manual_seed(11)
        enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=2, t_rounds=2, layernorm=True).to(dev)
        rng = np.random.default_rng(2)
        x = torch.zeros(n, 6, device=dev)
        x[torch.arange(n), torch.from_numpy(rng.integers(0, 6, n)).to(dev)] = 1.0
        s, t = enc(x, x, torch.from_numpy(ei).to(dev))
        gs, gt = torch.randn(n, 64, device=dev, generator=torch.Generator(dev).manual_seed(5)), torch.randn(n, 64, device=dev, generator=torch.Generator(dev).manual_seed(6))
        ((s * gs).sum() + (t * gt).sum()).backward()
        return [s.detach(), t.detach()] + [p.grad.detach().clone() for p in enc.parameters()], [k for k, _ in enc.named_parameters()]
    finally:
        ops.PRECISION, ops.FIRST_STAGE_TABLE = old


def _truth(ei, n):
    """float64 torch restatement of the two encoders (digae_layer.py:257-277) on the CPU, same seeds as _run."""
    import deepgate
    import torch.nn.functional as F
    torch.manual_seed(11)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=64, s_rounds=2, t_rounds=2, layernorm=True).double()
    rng = np.random.default_rng(2)
    x = torch.zeros(n, 6, dtype=torch.float64)
    x[torch.arange(n), torch.from_numpy(rng.integers(0, 6, n))] = 1.0
    src, dst = torch.from_numpy(ei[0]), torch.from_numpy(ei[1])
    outs = []
    for conv in (enc.source_conv, enc.target_conv):
        h = torch.ones(n, 64, dtype=torch.float64)
        for _ in range(2):
            for ag, gru, a, b in ((conv.aggr, conv.update, src, dst), (conv.aggr_r, conv.update_r, dst, src)):
                m = torch.zeros(n, 64, dtype=torch.float64).index_add_(0, b, h[a] @ ag.msg.weight.t() + ag.msg.bias)
                gi = torch.cat([m, x], 1) @ gru.weight_ih_l0.t() + gru.bias_ih_l0
                gh = h @ gru.weight_hh_l0.t() + gru.bias_hh_l0
                (ir, iz, inn), (hr, hz, hn) = gi.chunk(3, 1), gh.chunk(3, 1)
                r, z = torch.sigmoid(ir + hr), torch.sigmoid(iz + hz)
                h = F.layer_norm((1 - z) * torch.tanh(inn + r * hn) + z * h, (64,), conv.ln.weight, conv.ln.bias)
        outs.append(h)
    dev = torch.device('cuda:0')
    gs = torch.randn(n, 64, device=dev, generator=torch.Generator(dev).manual_seed(5)).cpu().double()
    gt = torch.randn(n, 64, device=dev, generator=torch.Generator(dev).manual_seed(6)).cpu().double()
    ((outs[0] * gs).sum() + (outs[1] * gt).sum()).backward()
    return [outs[0].detach(), outs[1].detach()] + [p.grad.clone() for p in enc.parameters()]


def test_struct_paths_agree_on_a_hub_graph():
    dev = _dev()
    ei, n = _hub_graph(np.random.default_rng(0))
    truth = _truth(ei, n)
    runs = {}
    for precision in ('f32', 'x3'):
        for first in ('full', 'table'):
            runs[precision, first], names = _run(dev, ei, n, precision, first)
    names = ['s', 't'] + names

    def worst(a_list, b_list):
        return max((float((a.cpu().double() - b.cpu().double()).abs().max()) / float(b.abs().max()), nm)
                   for nm, a, b in zip(names, a_list, b_list) if float(b.abs().max()) > 1e-6)
    # against float64: the hubs sum 1500-3000 rows, which conditions the gradients badly; measured 5e-4 (fp32 kernels) and
    # 3e-3 (bf16x3), against 3e-5 for bf16x3 on the same graph without the hub edges
    assert worst(runs['f32', 'full'], truth)[0] <= 2e-3, worst(runs['f32', 'full'], truth)
    assert worst(runs['x3', 'full'], truth)[0] <= 1e-2, worst(runs['x3', 'full'], truth)
    # the (degree, class)-table first half round is the same arithmetic as the per-node launch
    for precision, tol in (('f32', 2e-4), ('x3', 2e-3)):
        assert worst(runs[precision, 'table'], runs[precision, 'full'])[0] <= tol, (precision, worst(runs[precision, 'table'], runs[precision, 'full']))


@pytest.mark.parametrize('ctype,H', [('aig', 64), ('xmg', 64), ('mig', 32), ('aig', 16)])
def test_quotient_stages_equal_the_per_node_stages(ctype, H):
    """Early half rounds run on one row per colour (GraphPlan.quotient: rows that are identical by construction).  Same encoder,
    same batch, with and without: embeddings to rounding (a colour's row is computed from the same inputs in another tile), every
    parameter gradient to the rounding of its sums (per-colour sums first, instead of per-tile sums)."""
    dev = _dev()
    import deepgate
    from deepgate import ops, synthetic as syn
    from deepgate.graph_plan import GraphPlan
    arrays = syn.collate([syn.make_graph(ctype, 512 + 120 * 60, 60, 900 + i, n_inputs=512) for i in range(3)])
    n = arrays['num_nodes']
    # two nets with hundreds of consumers (a primary input and a level-1 gate): singleton colours whose representative rows take
    # the heavy-row pre-pass inside the quotient stages
    rng = np.random.Generator(np.random.PCG64(8))
    lv = arrays['forward_level']
    hub2 = int(np.nonzero(lv == 1)[0][2])
    extra = [np.stack([np.full(400, 5), rng.choice(np.nonzero(lv > 0)[0], size=400, replace=False)]),
             np.stack([np.full(300, hub2), rng.choice(np.nonzero(lv >= 2)[0], size=300, replace=False)])]
    arrays['edge_index'] = np.unique(np.concatenate([arrays['edge_index']] + extra, axis=1), axis=1)
    ei = torch.from_numpy(arrays['edge_index']).to(dev)
    x = torch.from_numpy(arrays['x']).to(dev)
    plan = GraphPlan(ei, n)
    xcls = x[:, 1].to(torch.uint8).contiguous()
    quot = plan.quotient(xcls, 4)
    assert len(quot) >= 2 and quot[0]['C'] <= 16 and max(s['heavy'][0] for s in quot) >= 1, [(s['C'], s['heavy'][0]) for s in quot]
    res = {}
    for flag in (True, False):
        old = ops.QUOTIENT
        ops.QUOTIENT = flag
        try:
            torch.manual_seed(11)
            enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=2, t_rounds=2, layernorm=True).to(dev)
            s, t = enc(x, x, ei)
            gs = torch.randn(n, H, device=dev, generator=torch.Generator(dev).manual_seed(5))
            gt = torch.randn(n, H, device=dev, generator=torch.Generator(dev).manual_seed(6))
            ((s * gs).sum() + (t * gt).sum()).backward()
            res[flag] = [s.detach(), t.detach()] + [p.grad.detach().clone() for p in enc.parameters()]
            names = ['s', 't'] + [k for k, _ in enc.named_parameters()]
        finally:
            ops.QUOTIENT = old
    for nm, a, b in zip(names, res[True], res[False]):
        scale = float(b.abs().max())
        if scale > 1e-6:
            tol = 2e-5 if nm in ('s', 't') else 1e-4      # (a member's neighbours may come in another order than its representative's)
            assert float((a - b).abs().max()) <= tol * scale, (nm, float((a - b).abs().max()) / scale)
    # and twice the same: bit-identical (segment sums in list order, gathers over the colour lists: no atomics)
    torch.manual_seed(11)
    enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=2, t_rounds=2, layernorm=True).to(dev)
    s, t = enc(x, x, ei)
    ((s * gs).sum() + (t * gt).sum()).backward()
    again = [s.detach(), t.detach()] + [p.grad.detach().clone() for p in enc.parameters()]
    if H == 64:          # (the H = 32 / 16 backward kernels add their weight gradients with float atomics)
        assert all(torch.equal(a, b) for a, b in zip(again, res[True]))


def test_segment_sums_against_float64():
    """mgv_seg_sum through GraphPlan.class_sum_levels: per-colour sums of (direct[i] + sum of agg over i's list) with one colour that
    holds half the nodes, singletons and an empty colour, against a float64 index_add; and bit-identical twice."""
    dev = _dev()
    from deepgate import ops
    from deepgate.graph_plan import GraphPlan
    g = torch.Generator().manual_seed(3)
    N, E, H, C = 50000, 120000, 64, 700
    ei = torch.stack([torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)]).to(dev)
    plan = GraphPlan(ei, N)
    cid = torch.randint(1, C - 1, (N,), generator=g)
    cid[torch.rand(N, generator=g) < 0.5] = 0                       # a colour with half the nodes; colour C-1 stays empty
    cid = cid.to(dev)
    direct = torch.randn(N, H, generator=g).to(dev)
    agg = torch.randn(N, H, generator=g).to(dev)
    order, levels = plan.class_sum_levels(cid, C)
    p, i = plan.csr(False)
    got = ops._seg_sums(H, levels, order, direct, agg, p, i)
    dy = direct.double().clone()
    dy.index_add_(0, plan.in_dst.long(), agg.double()[plan.in_src.long()])
    ref = torch.zeros(C, H, dtype=torch.float64, device=dev).index_add_(0, cid.long(), dy)
    assert got.shape == (C, H) and float(got[C - 1].abs().max()) == 0.0
    assert float((got.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(got, ops._seg_sums(H, levels, order, direct, agg, p, i))
    plain = ops._seg_sums(H, levels, order, direct)
    ref2 = torch.zeros(C, H, dtype=torch.float64, device=dev).index_add_(0, cid.long(), direct.double())
    assert float((plain.double() - ref2).abs().max()) <= 2e-5 * float(ref2.abs().max())


@pytest.mark.parametrize('H', [64, 32])
def test_every_half_round_on_colours_for_a_regular_netlist(H):
    """A batch of 70 identical balanced AND trees (256 leaves each, an inverter behind every gate of alternate levels): the colours stay a few dozen through
    all four half rounds, so EVERY half round of both encoders is a quotient stage and the expansion to N rows happens once, at the
    end; against the per-node path, and bit-identical twice."""
    dev = _dev()
    import deepgate
    from deepgate import ops
    from deepgate.graph_plan import GraphPlan
    src, dst, gate = [], [], []
    leaves, per = 256, None
    # one tree: nodes 0..255 inputs, then level by level: AND of two children, every second level followed by an inverter per node
    def tree(base):
        ids = list(range(base, base + leaves))
        g = [0] * leaves
        nxt = base + leaves
        lvl = 0
        while len(ids) > 1:
            new = []
            for a, b in zip(ids[0::2], ids[1::2]):
                src.extend([a, b]); dst.extend([nxt, nxt]); g.append(1); v = nxt; nxt += 1
                if lvl % 2 == 1:
                    src.append(v); dst.append(nxt); g.append(2); v = nxt; nxt += 1
                new.append(v)
            ids, lvl = new, lvl + 1
        return g, nxt
    base = 0
    for _ in range(70):
        g, base = tree(base)
        gate.extend(g)
    n = base
    assert n >= GraphPlan.QUOTIENT_MIN_NODES
    ei = torch.tensor([src, dst], dtype=torch.int64, device=dev)
    x = torch.zeros(n, 6, device=dev)
    x[torch.arange(n, device=dev), torch.tensor(gate, device=dev)] = 1.0
    plan = GraphPlan(ei, n)
    quot = plan.quotient(x[:, 1].to(torch.uint8).contiguous(), 4)
    assert len(quot) == 4 and quot[-1]['C'] <= 64, [s['C'] for s in quot]
    res = {}
    for flag in (True, False, True):
        old = ops.QUOTIENT
        ops.QUOTIENT = flag
        try:
            torch.manual_seed(11)
            enc = deepgate.digae_layer.DirectMultiGCNEncoder(dim_feature=6, dim_hidden=H, s_rounds=2, t_rounds=2, layernorm=True).to(dev)
            s, t = enc(x, x, ei)
            gs = torch.randn(n, H, device=dev, generator=torch.Generator(dev).manual_seed(5))
            gt = torch.randn(n, H, device=dev, generator=torch.Generator(dev).manual_seed(6))
            ((s * gs).sum() + (t * gt).sum()).backward()
            out = [s.detach(), t.detach()] + [p.grad.detach().clone() for p in enc.parameters()]
            if flag and flag in res:
                if H == 64:      # (the H = 32 backward adds its weight gradients with float atomics)
                    assert all(torch.equal(a, b) for a, b in zip(out, res[True]))
            else:
                res[flag] = out
            names = ['s', 't'] + [k for k, _ in enc.named_parameters()]
        finally:
            ops.QUOTIENT = old
    for nm, a, b in zip(names, res[True], res[False]):
        scale = float(b.abs().max())
        if scale > 1e-6:
            tol = 2e-5 if nm in ('s', 't') else 1e-4
            assert float((a - b).abs().max()) <= tol * scale, (nm, float((a - b).abs().max()) / scale)
