"""Per-batch graph structure ("plan") consumed by the HIP kernels.

Replaces, for the whole batch at once, what the reference recomputes inside every forward:
`torch.stack([edge_index[1], edge_index[0]])` (digae_layer.py:264), the boolean level/gate masks and
`forward_index[mask]` (dg_ae_model_aig.py:72-75) and the per-node edge scans of `subgraph`
(utils/dag_utils.py:91-105, O(nodes x E) Python) — with two CSRs and a list of 64-node tiles
bucketed by (level, gate type).  Built with torch ops on whatever device the batch lives on; it is
batch construction, not part of the timed train step.
"""
import torch

TILE = 64
NO_GATE = 255


class GraphPlan:
    """in-CSR  : in_ptr[N+1], in_src[E]              sources of each node, original edge order kept
    out-CSR : out_ptr[N+1], out_dst[E], out_slot[E] destinations; out_slot = position of that edge
                                                     in the in-CSR (per-edge scratch is in-CSR ordered)
    sweep   : gslot[N] aggregator slot of a node (255 = never updated), order[] = updated nodes sorted
              by (level, slot), tiles of <= 64 consecutive `order` entries with one slot each,
              level_tile_ptr[L+1] = tile range of each level (levels >= 1)
    """

    def __init__(self, edge_index, num_nodes, device=None):
        dev = edge_index.device if device is None else torch.device(device)
        ei = edge_index.to(dev)
        N = int(num_nodes)
        E = int(ei.shape[1])
        if N >= 2 ** 31 or E >= 2 ** 31:
            raise ValueError('node/edge counts must fit int32')
        src, dst = ei[0].long(), ei[1].long()
        self.N, self.E, self.device = N, E, dev
        perm_in = torch.sort(dst, stable=True).indices
        perm_out = torch.sort(src, stable=True).indices
        self.in_src = src[perm_in].to(torch.int32).contiguous()
        self.out_dst = dst[perm_out].to(torch.int32).contiguous()
        inv_in = torch.empty(E, dtype=torch.long, device=dev)
        inv_in[perm_in] = torch.arange(E, device=dev)
        self.out_slot = inv_in[perm_out].to(torch.int32).contiguous()
        self.in_ptr = self._ptr(dst, N)
        self.out_ptr = self._ptr(src, N)
        self.in_dst = dst[perm_in].to(torch.int32).contiguous()   # destination of each in-CSR slot
        self.perm_in = perm_in
        self.has_levels = False

    @staticmethod
    def _ptr(index, n):
        cnt = torch.bincount(index, minlength=n)
        p = torch.zeros(n + 1, dtype=torch.int64, device=index.device)
        p[1:] = torch.cumsum(cnt, 0)
        return p.to(torch.int32).contiguous()

    def csr(self, reverse):
        """(ptr, idx) of the neighbours a node sums over: in-neighbours, or out-neighbours when the
        edges are flipped (`r_edge_index`, digae_layer.py:264)."""
        return (self.out_ptr, self.out_dst) if reverse else (self.in_ptr, self.in_src)

    def first_stage_classes(self, xcls, max_classes=256):
        """(degree, feature class) pairs of the forward CSR: every node enters the first half round of an encoder with
        the same state (ones, digae_layer.py:260), so its output row there depends on that pair alone.  Returns
        (class_id[N] int32, C, ptr[C+1], idx[sum deg], xcls[C]) — one representative row per pair whose `deg`
        neighbours all point at row 0 (any row: the representatives' inputs are all ones) — or None when there are
        more than `max_classes` pairs.  Cached per xcls tensor."""
        key = (xcls.data_ptr(), int(xcls.numel()))
        hit = getattr(self, '_stage1', None)
        if hit is not None and hit[0] == key:
            return hit[1]
        deg = (self.in_ptr[1:] - self.in_ptr[:-1]).long()
        pair = deg * 256 + xcls.long()
        uniq, inv = torch.unique(pair, return_inverse=True)
        C = int(uniq.numel())
        out = None
        if 0 < C <= max_classes:
            d = (uniq // 256)
            ptr = torch.zeros(C + 1, dtype=torch.int64, device=self.device)
            ptr[1:] = torch.cumsum(d, 0)
            idx = torch.zeros(max(int(ptr[-1].item()), 1), dtype=torch.int32, device=self.device)
            out = (inv.to(torch.int32).contiguous(), C, ptr.to(torch.int32).contiguous(), idx,
                   (uniq % 256).to(torch.uint8).contiguous())
        self._stage1 = (key, out)
        return out

    def set_levels(self, gate, forward_level, gate_ids):
        """Bucket the nodes a Model updates: level >= 1 and gate id in `gate_ids` (list, position =
        aggregator slot).  Mirrors `layer_mask & <gate>_mask` of the reference level loop."""
        dev = self.device
        g = gate.reshape(-1).to(dev).long()
        lv = forward_level.reshape(-1).to(dev).long()
        N = self.N
        if g.numel() != N or lv.numel() != N:
            raise ValueError('gate / forward_level must have one entry per node')
        slot_of = torch.full((256,), NO_GATE, dtype=torch.long, device=dev)
        for s, gid in enumerate(gate_ids):
            slot_of[int(gid)] = s
        gslot = slot_of[g.clamp(0, 255)]
        active = (lv >= 1) & (gslot != NO_GATE)
        gslot = torch.where(active, gslot, torch.full_like(gslot, NO_GATE))
        self.gslot = gslot.to(torch.uint8).contiguous()
        self.level = lv.to(torch.int32).contiguous()
        self.num_levels = int(lv.max().item()) + 1 if N > 0 else 0
        nodes = torch.nonzero(active).reshape(-1)
        # every source of an updated node must sit on a strictly lower level: the kernels update a
        # level in place, the reference reads the pre-level state (dg_ae_model_aig.py:97)
        if self.E > 0:
            d = self.in_dst.long()
            s = self.in_src.long()
            bad = active[d] & (lv[s] >= lv[d])
            if bool(bad.any()):
                raise ValueError('forward_level is not a topological levelisation of edge_index')
        T = len(gate_ids)
        key = lv[nodes] * T + gslot[nodes]
        o = torch.sort(key, stable=True)
        self.order = nodes[o.indices].to(torch.int32).contiguous()
        # CSR spans in sweep order {in0, in1, out0, out1}: a tile reaches its edge lists with one load per row
        on = nodes[o.indices]
        self.order_span = torch.stack([self.in_ptr[on], self.in_ptr[on + 1], self.out_ptr[on], self.out_ptr[on + 1]],
                                      1).to(torch.int32).contiguous()
        keys, counts = torch.unique_consecutive(o.values, return_counts=True)
        starts = torch.cumsum(counts, 0) - counts
        ntile = (counts + TILE - 1) // TILE
        gidx = torch.repeat_interleave(torch.arange(keys.numel(), device=dev), ntile)
        first = torch.cumsum(ntile, 0) - ntile
        k_in_group = torch.arange(int(ntile.sum()), device=dev) - first[gidx]
        t_start = starts[gidx] + k_in_group * TILE
        t_count = torch.minimum(counts[gidx] - k_in_group * TILE, torch.full_like(k_in_group, TILE))
        t_slot = keys[gidx] % T
        t_level = keys[gidx] // T
        self.tile_start = t_start.to(torch.int32).contiguous()
        self.tile_count = t_count.to(torch.int32).contiguous()
        self.tile_slot = t_slot.to(torch.int32).contiguous()
        # tiles grouped by slot, for the per-slot weight-gradient pass of the backward sweep
        self.slot_tiles = torch.sort(t_slot, stable=True).indices.to(torch.int32).contiguous()
        stp = torch.zeros(T + 1, dtype=torch.int64, device=dev)
        stp[1:] = torch.cumsum(torch.bincount(t_slot, minlength=T), 0)
        self.slot_tile_ptr = [int(v) for v in stp.tolist()]        # host copy
        per_level = torch.bincount(t_level, minlength=max(self.num_levels, 1))
        ltp = torch.zeros(max(self.num_levels, 1) + 1, dtype=torch.int64, device=dev)
        ltp[1:] = torch.cumsum(per_level, 0)
        self.level_tile_ptr = [int(v) for v in ltp.tolist()]      # host copy: launch geometry
        self.num_tiles = int(self.tile_start.numel())
        self.n_active = int(nodes.numel())
        self.num_slots = T
        self.has_levels = True
        return self

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        self.device = torch.device(device)
        return self
