"""GPU checks of the bf16x3 level-sweep kernels (csrc/func_level_x3.hip) against the exact-fp32 sweep kernels
(csrc/func_level.hip, themselves checked against the reference fixtures in test_hip_model.py) on graphs built to hit
the paths the BASELINE shapes do not: fan-in beyond the gathered-together rows (kInRegs = 3) and beyond the staged
in-edge list (kInCap = 4), fan-out beyond the staged out-edge list (kOutCap = 16), partial tiles, five gate types,
nodes no aggregator updates.  Tolerance: 2e-4 of the largest entry (bf16x3 products, fp32 everything else)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _graph(rng, n_in, levels, per, T, hub_fanout, big_fanin):
    """Levelised DAG: node k of a level takes its first parent from the previous level, the rest from anywhere
    earlier; node 0 (an input) additionally feeds `hub_fanout` nodes, one node per level has `big_fanin` parents."""
    n = n_in + levels * per
    src, dst = [], []
    gate = np.zeros(n, dtype=np.int64)
    level = np.zeros(n, dtype=np.int64)
    for lv in range(1, levels + 1):
        lo = n_in + (lv - 1) * per
        prev_lo = 0 if lv == 1 else n_in + (lv - 2) * per
        prev_hi = lo
        for k in range(per):
            v = lo + k
            level[v] = lv
            gate[v] = 1 + (k % T) if k % 7 != 6 else 9             # gate id 9: not updated by any aggregator
            fanin = big_fanin if k == 1 else int(rng.integers(1, 4))
            ps = {int(rng.integers(prev_lo, prev_hi))}
            while len(ps) < min(fanin, lo):
                ps.add(int(rng.integers(0, lo)))
            for p in ps:
                src.append(p); dst.append(v)
    hubs = rng.choice(np.arange(n_in, n), size=hub_fanout, replace=False)
    for v in hubs:
        if not any(s == 0 and d == v for s, d in zip(src[-200:], dst[-200:])):
            src.append(0); dst.append(int(v))
    ei = np.unique(np.array([src, dst], dtype=np.int64), axis=1)
    return ei, gate, level, n


@pytest.mark.parametrize('H,T', [(64, 5), (32, 2)])
def test_sweep_x3_matches_fp32_sweep_on_irregular_graphs(H, T):
    dev = _dev()
    from deepgate import ops
    from deepgate.graph_plan import GraphPlan
    if ops.PRECISION != 'x3':
        pytest.skip('bf16x3 mode only')
    rng = np.random.default_rng(5 + H)
    ei, gate, level, n = _graph(rng, n_in=40, levels=9, per=75, T=T, hub_fanout=60, big_fanin=7)   # 75 nodes/level: partial tiles
    plan = GraphPlan(torch.from_numpy(ei).to(dev), n)
    plan.set_levels(torch.from_numpy(gate).to(dev), torch.from_numpy(level).to(dev), list(range(1, T + 1)))
    assert int((plan.out_ptr[1:] - plan.out_ptr[:-1]).max()) > 16 and int((plan.in_ptr[1:] - plan.in_ptr[:-1]).max()) > 4
    torch.manual_seed(H)
    hs0 = torch.randn(n, H, device=dev)
    par0 = [torch.randn(T, 2 * H, device=dev) * 0.3, torch.randn(T, 3 * H, 2 * H, device=dev) * 0.15,
            torch.randn(T, 3 * H, device=dev) * 0.1, torch.randn(T, 3 * H, device=dev) * 0.1, torch.randn(T, 3 * H, device=dev) * 0.1]
    ghf = torch.randn(n, H, device=dev)
    results = []
    import os
    for x3 in ('1', '0'):
        os.environ['MGV_SWEEP_X3'] = x3
        hs = hs0.clone().requires_grad_(True)
        par = [p.clone().requires_grad_(True) for p in par0]
        hf = ops.FuncSweepFn.apply(plan, hs, *par)
        (hf * ghf).sum().backward()
        results.append([hf.detach()] + [hs.grad] + [p.grad for p in par])
    os.environ.pop('MGV_SWEEP_X3', None)
    names = ['hf', 'd hs', 'd attn_u', 'd Wvc', 'd bvc', 'd bih', 'd bhh']
    for name, a, b in zip(names, *results):
        scale = float(b.abs().max())
        assert scale > 0, name
        assert float((a - b).abs().max()) <= 2e-4 * scale, (name, float((a - b).abs().max()), scale)
    # nodes of the unknown gate type and the inputs keep hf = 0
    idle = torch.from_numpy((gate == 9) | (level == 0)).to(dev)
    assert float(results[0][0][idle].abs().max()) == 0.0


@pytest.mark.parametrize('T,levels,per', [(5, 9, 75), (2, 40, 200), (3, 3, 10)])
def test_persistent_sweep_kernels_match_the_per_level_kernels(T, levels, per):
    """csrc/sweep_persist_x3.hip (one persistent kernel per direction: slot-dedicated workgroups, grid barrier between levels,
    in-register weight gradient; opt-in, ops.PERSIST) against the per-level kernels on the same irregular graphs: forward
    bit-identical (same arithmetic in the same order), backward to the rounding of its sums (another summation order of the
    parameter gradients), its barrier never gave up, and two identical calls give identical bits."""
    dev = _dev()
    from deepgate import ops
    from deepgate.graph_plan import GraphPlan
    if ops.PRECISION != 'x3':
        pytest.skip('bf16x3 mode only')
    H = 64
    rng = np.random.default_rng(11 + T)
    ei, gate, level, n = _graph(rng, n_in=40, levels=levels, per=per, T=T, hub_fanout=min(50, levels * per // 2), big_fanin=min(7, 30))
    plan = GraphPlan(torch.from_numpy(ei).to(dev), n)
    plan.set_levels(torch.from_numpy(gate).to(dev), torch.from_numpy(level).to(dev), list(range(1, T + 1)))
    torch.manual_seed(H + T)
    hs0 = torch.randn(n, H, device=dev)
    par0 = [torch.randn(T, 2 * H, device=dev) * 0.3, torch.randn(T, 3 * H, 2 * H, device=dev) * 0.15,
            torch.randn(T, 3 * H, device=dev) * 0.1, torch.randn(T, 3 * H, device=dev) * 0.1, torch.randn(T, 3 * H, device=dev) * 0.1]
    ghf = torch.randn(n, H, device=dev)
    results = []
    old = ops.PERSIST
    try:
        for persist in (True, True, False):
            ops.PERSIST = persist
            hs = hs0.clone().requires_grad_(True)
            par = [p.clone().requires_grad_(True) for p in par0]
            hf = ops.FuncSweepFn.apply(plan, hs, *par)
            (hf * ghf).sum().backward()
            results.append([hf.detach()] + [hs.grad] + [p.grad for p in par])
            ops.persist_check(dev)
    finally:
        ops.PERSIST = old
    assert ops._persist_roles(plan, H, n) is None                       # (switched off again: the product default)
    names = ['hf', 'd hs', 'd attn_u', 'd Wvc', 'd bvc', 'd bih', 'd bhh']
    assert torch.equal(results[0][0], results[2][0])                    # forward: bit-identical to the per-level kernels
    for name, a, b, c in zip(names, *results):
        assert torch.equal(a, b), name                                  # twice the same: identical bits
        scale = float(c.abs().max())
        assert scale > 0, name
        assert float((a - c).abs().max()) <= 2e-5 * scale, (name, float((a - c).abs().max()), scale)


@pytest.mark.parametrize('T,levels,per', [(5, 9, 75), (2, 40, 200)])
def test_packed_sweep_rows_and_span_rows_give_the_same_sweep(T, levels, per):
    """The level kernels read a tile's lists either from packed 128-byte rows (GraphPlan.order_rows, the product default: spans,
    first 4 sources, first 8 consumers) or from 16-byte span rows and the CSR lists behind them (MGV_PACKED_ROWS=0: 16 consumers
    staged).  Same graph with fan-outs past both caps: the forward is bit-identical, the backward sums a node's consumers in the
    same order up to where the tail begins — to the rounding of those sums."""
    dev = _dev()
    from deepgate import ops
    from deepgate.graph_plan import GraphPlan
    if ops.PRECISION != 'x3':
        pytest.skip('bf16x3 mode only')
    H = 64
    rng = np.random.default_rng(5 + T)
    ei, gate, level, n = _graph(rng, n_in=40, levels=levels, per=per, T=T, hub_fanout=min(50, levels * per // 2), big_fanin=min(7, 30))
    # two UPDATED gates of level 1 that drive many later gates: consumer lists past the 8 a packed row holds and past the 16 staged otherwise
    later = np.arange(40 + 2 * per, n)
    extra = []
    for v, want in ((40, 33), (42, 12)):
        has = set(ei[1][ei[0] == v].tolist())
        pool = np.array([c for c in later if c not in has])
        extra.append(np.stack([np.full(want - len(has), v), rng.choice(pool, size=want - len(has), replace=False)]))
    ei = np.unique(np.concatenate([ei] + extra, axis=1), axis=1)
    plan = GraphPlan(torch.from_numpy(ei).to(dev), n)
    plan.set_levels(torch.from_numpy(gate).to(dev), torch.from_numpy(level).to(dev), list(range(1, T + 1)))
    deg = (plan.out_ptr[1:] - plan.out_ptr[:-1])[plan.order.long()]
    assert int((deg > 16).sum()) > 0 and int(((deg > 8) & (deg <= 16)).sum()) > 0      # both tails are exercised
    torch.manual_seed(H + T)
    hs0 = torch.randn(n, H, device=dev)
    par0 = [torch.randn(T, 2 * H, device=dev) * 0.3, torch.randn(T, 3 * H, 2 * H, device=dev) * 0.15,
            torch.randn(T, 3 * H, device=dev) * 0.1, torch.randn(T, 3 * H, device=dev) * 0.1, torch.randn(T, 3 * H, device=dev) * 0.1]
    ghf = torch.randn(n, H, device=dev)
    results = []
    old = ops.PACKED_ROWS
    try:
        for packed in (2, 0):                # 2: packed rows from a plan's first step on (the default builds them at its second step)
            ops.PACKED_ROWS = packed
            plan.__dict__.pop('_order_rows', None)
            hs = hs0.clone().requires_grad_(True)
            par = [p.clone().requires_grad_(True) for p in par0]
            hf = ops.FuncSweepFn.apply(plan, hs, *par)
            (hf * ghf).sum().backward()
            results.append([hf.detach()] + [hs.grad] + [p.grad for p in par])
    finally:
        ops.PACKED_ROWS = old
    assert torch.equal(results[0][0], results[1][0])
    # the default: span rows on a plan's first step, packed rows from its second step on
    ops.PACKED_ROWS = 1
    try:
        plan.__dict__.pop('_order_rows', None)
        plan.__dict__.pop('_sweep_steps', None)
        for step in range(2):
            hs = hs0.clone().requires_grad_(True)
            par = [p.clone().requires_grad_(True) for p in par0]
            hf = ops.FuncSweepFn.apply(plan, hs, *par)
            assert (plan.__dict__.get('_order_rows') is not None) == (step == 1)
            (hf * ghf).sum().backward()
            assert torch.equal(hf.detach(), results[0][0])
    finally:
        ops.PACKED_ROWS = old
    for name, a, b in zip(['hf', 'd hs', 'd attn_u', 'd Wvc', 'd bvc', 'd bih', 'd bhh'], *results):
        scale = float(b.abs().max())
        assert scale > 0, name
        assert float((a - b).abs().max()) <= 2e-5 * scale, (name, float((a - b).abs().max()), scale)
