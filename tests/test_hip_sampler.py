"""GPU checks of the fused negative sampler (csrc/neg_sample.hip) and of the CSR-only reconstruction-loss backward.

The reference draws negatives with torch_geometric.utils.negative_sampling (dg_ae_model_aig.py:115-119); what can be
checked of a random draw: the pairs are never edges or self loops, there are |E| + N of them, the two CSRs describe
exactly the returned pairs, sources are spread uniformly, two calls differ, and the atomic-free backward equals the
atomic one on the same pairs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def _plan(dev, n_graphs=4, nodes=2048):
    import deepgate
    from deepgate import synthetic as syn
    graphs = [syn.make_graph('aig', nodes, 24, 300 + i, n_inputs=128) for i in range(n_graphs)]
    batch = deepgate.CircuitBatch.from_arrays(syn.collate(graphs), device=dev)
    return batch, deepgate.data.plan_of(batch, [1, 2])


def test_device_sampler_draws_valid_pairs_and_consistent_buckets():
    dev = _dev()
    from deepgate import sampling
    batch, plan = _plan(dev)
    N, E = plan.N, plan.E
    neg = sampling.negative_sampling_device(plan)
    ei = neg.edge_index.cpu().numpy()
    assert ei.shape == (2, E + N)                                   # the synthetic DAGs have no self loops
    assert ei.min() >= 0 and ei.max() < N
    assert not np.any(ei[0] == ei[1])
    pos = set((batch.edge_index[0].cpu().numpy() * N + batch.edge_index[1].cpu().numpy()).tolist())
    keys = ei[0] * N + ei[1]
    assert not any(int(k) in pos for k in keys)
    # grouped by source, and the CSRs are those pairs
    assert np.all(np.diff(ei[0]) >= 0)
    out_ptr, out_dst, in_ptr, in_src = [t.cpu().numpy() for t in neg.csr]
    assert out_ptr[0] == 0 and out_ptr[-1] == ei.shape[1] and in_ptr[-1] == ei.shape[1]
    np.testing.assert_array_equal(np.bincount(ei[0], minlength=N), np.diff(out_ptr))
    np.testing.assert_array_equal(np.bincount(ei[1], minlength=N), np.diff(in_ptr))
    np.testing.assert_array_equal(out_dst[:ei.shape[1]], ei[1])
    by_dst = np.repeat(np.arange(N), np.diff(in_ptr)) * N + in_src[:ei.shape[1]]       # (dst, src) pairs of the in-CSR
    np.testing.assert_array_equal(np.sort(by_dst), np.sort(ei[1] * N + ei[0]))
    # uniform sources: chi-square of the per-source counts against a flat expectation stays near its mean
    cnt = np.bincount(ei[0], minlength=N).astype(np.float64)
    lam = ei.shape[1] / N
    chi = ((cnt - lam) ** 2 / lam).sum()
    assert abs(chi - N) < 8 * np.sqrt(2 * N), (chi, N)
    # a second call draws other pairs
    other = sampling.negative_sampling_device(plan).edge_index.cpu().numpy()
    assert np.mean(np.sort(other[0] * N + other[1]) == np.sort(keys)) < 0.01


def test_same_seed_and_call_number_give_identical_buckets():
    """The buckets are filled through atomic cursors and then sorted per list: the same draw comes out in the same order."""
    dev = _dev()
    import itertools
    from deepgate import sampling
    batch, plan = _plan(dev, n_graphs=8, nodes=2048)
    draws = []
    for _ in range(3):
        sampling._CALLS = itertools.count()
        torch.manual_seed(1234)
        neg = sampling.negative_sampling_device(plan)
        draws.append([neg.edge_index.clone()] + [t.clone() for t in neg.csr])
    for other in draws[1:]:
        for a, b in zip(draws[0], other):
            assert torch.equal(a, b)
    ei = draws[0][0].cpu().numpy()
    out_ptr = draws[0][1].cpu().numpy()
    for n in np.random.Generator(np.random.PCG64(0)).integers(0, plan.N, size=200):
        assert np.all(np.diff(ei[1][out_ptr[n]:out_ptr[n + 1]]) >= 0)          # ascending inside a list


def test_csr_backward_equals_atomic_backward():
    dev = _dev()
    from deepgate import ops, sampling
    batch, plan = _plan(dev)
    N, H = plan.N, 64
    torch.manual_seed(3)
    st0 = (torch.randn(N, 2 * H, device=dev) * 0.3)
    neg = sampling.negative_sampling_device(plan)
    outs = []
    for csr in (neg.csr, None):
        st = st0.clone().requires_grad_(True)
        loss, counts, _ = ops.ReconLossFn.apply(st, batch.edge_index, neg.edge_index, False, plan, csr)
        (loss * 1.7).backward()
        outs.append((float(loss), st.grad.clone(), counts.cpu().numpy()))
    assert outs[0][0] == outs[1][0]
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    scale = float(outs[1][1].abs().max())
    assert float((outs[0][1] - outs[1][1]).abs().max()) <= 2e-6 * max(scale, 1e-6) + 1e-9
