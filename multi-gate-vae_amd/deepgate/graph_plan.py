"""Per-batch graph structure ("plan") consumed by the HIP kernels.

Replaces, for the whole batch at once, what the reference recomputes inside every forward:
`torch.stack([edge_index[1], edge_index[0]])` (digae_layer.py:264), the boolean level/gate masks and
`forward_index[mask]` (dg_ae_model_aig.py:72-75) and the per-node edge scans of `subgraph`
(utils/dag_utils.py:91-105, O(nodes x E) Python) — with two CSRs and a list of 64-node tiles
bucketed by (level, gate type).  On the GPU the plan is built by the HIP kernels of csrc/plan_build.hip (degree
histogram + scan + cursor fill + per-list sort for the CSRs, a stable counting sort by (level, slot) for the buckets:
SURVEY.md §8f row 1); on CPU tensors (host-logic tests, and `MGV_PLAN=torch` as the GPU tests' cross-check) by the
torch sorts / scans below.  Both give identical arrays.  It is batch construction, not part of the timed train step
(SURVEY.md §8d).
"""
import ctypes
import os

import numpy as np

import torch

TILE = 64
NO_GATE = 255


class ColourDictionary:
    """Dataset-wide, EXACT dictionary of colour signatures -> global colour ids, one per half round of the structural encoder.
    A colour of half round t is (feature class, global colour of its own row after t-1, sorted global colours of its neighbours after
    t-1); two nodes of different graphs with equal signatures have identical rows after half round t (induction over t: every node
    starts from ones, digae_layer.py:260), so a batch may compute them once.  Signatures are compared as Python tuples / bytes —
    no hashing of our own, no collisions.  Filled per graph, once, from the graph's own quotient stages (their representatives'
    lists): GraphPlan.assemble_quotient then merges the graphs' colours of a batch by their global ids."""

    def __init__(self, max_entries=20000000):
        self.maps, self.next_id, self.max_entries = [], [], int(max_entries)

    def globals_of(self, stages):
        """stages: per half round a dict of HOST arrays ptr [C+1], ent [n_ent] (previous colours, the graph's own numbering),
        own [C], xcls [C].  -> list of int64 arrays: the global id of every colour of the graph, per half round."""
        out = []
        gprev = np.zeros(1, dtype=np.int64)                   # before the first half round: one colour, global id 0
        for t, st in enumerate(stages):
            while len(self.maps) <= t:
                self.maps.append({}); self.next_id.append(0)
            table = self.maps[t]
            ptr, ent, own, xcls = (np.asarray(st[k]) for k in ('ptr', 'ent', 'own', 'xcls'))
            C = int(own.shape[0])
            deg = np.diff(ptr)
            entg = gprev[ent] if ent.size else np.zeros(0, dtype=np.int64)
            rowid = np.repeat(np.arange(C), deg)
            entg = entg[np.lexsort((entg, rowid))] if entg.size else entg      # every list sorted by global colour: a multiset
            owng = gprev[own]
            g = np.empty(C, dtype=np.int64)
            full = len(table) >= self.max_entries
            for c in range(C):
                key = (int(xcls[c]), int(owng[c]), entg[ptr[c]:ptr[c + 1]].tobytes())
                v = table.get(key)
                if v is None:
                    v = self.next_id[t]
                    self.next_id[t] += 1
                    if not full:
                        table[key] = v                        # (a full table hands out fresh ids: such colours are simply not shared)
                g[c] = v
            out.append(g)
            gprev = g
        return out


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
        self.hip = dev.type == 'cuda' and os.environ.get('MGV_PLAN', 'hip') != 'torch'
        if self.hip:
            self._build_csr_hip(src.contiguous(), dst.contiguous())
            self.has_levels = False
            return
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

    def _build_csr_hip(self, src, dst):
        from . import _hip
        from ._hip import ptr
        N, E, dev = self.N, self.E, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        self.in_ptr, self.out_ptr = torch.empty(N + 1, **i32), torch.empty(N + 1, **i32)
        self.in_src, self.in_dst = torch.empty(max(E, 1), **i32)[:E], torch.empty(max(E, 1), **i32)[:E]
        self.out_dst, self.out_slot = torch.empty(max(E, 1), **i32)[:E], torch.empty(max(E, 1), **i32)[:E]
        n_s = _hip.call_value('mgv_plan_csr_scratch_ints', N, E)
        scratch = torch.empty(n_s, **i32)
        status = torch.empty(2, **i32)
        _hip.call('mgv_plan_csr', N, E, ptr(src), ptr(dst), ptr(self.in_ptr), ptr(self.in_src), ptr(self.in_dst), ptr(self.out_ptr),
                  ptr(self.out_dst), ptr(self.out_slot), None, None, ptr(scratch), n_s, ptr(status))
        # Read lazily (with the level checks: one host round trip per batch).  The arrays are safe to launch on before that: the
        # CSR kernels skip out-of-range edges and zero the unfilled tail, so every stored id lies in [0, N).
        self._status = status
        self._keep = (src, dst)
        if N == 0 and E > 0:
            raise ValueError('edge_index holds node ids outside [0, num_nodes)')

    def asap_levels(self):
        """ASAP level of every node (the round of utils/dag_utils.top_sort in which it is evaluated = longest path from a source),
        by frontier relaxation over the out-CSR on the device.  Raises on a cycle.  -> int64 [N]"""
        N, dev = self.N, self.device
        if not self.hip:
            from .parser import forward_levels
            ei = torch.stack([self.in_src.long(), self.in_dst.long()]).cpu().numpy()
            return torch.from_numpy(forward_levels(ei, N)).to(dev)
        from . import _hip
        from ._hip import ptr
        self._check_status()
        level = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)[:N]
        if N == 0:
            return level.long()
        done = torch.zeros(1, dtype=torch.int32, device=dev)
        rounds = 512
        while True:
            scratch = torch.empty(3 * N + rounds + 2, dtype=torch.int32, device=dev)
            _hip.call('mgv_plan_levels', N, ptr(self.in_ptr), ptr(self.out_ptr), ptr(self.out_dst), ptr(level), rounds, ptr(scratch),
                      scratch.numel(), ptr(done))
            if int(done.item()) == N:
                return level.long()
            if rounds >= N + 1:              # more rounds than nodes cannot help: some node never lost its last pending parent
                raise ValueError('edge_index is not a DAG (%d of %d nodes levelised)' % (int(done.item()), N))
            rounds = min(rounds * 8, N + 1)

    def _check_status(self):
        st = getattr(self, '_status', None)
        if st is not None:
            self._status = None
            if int(st[0].item()) != 0:
                raise ValueError('edge_index holds node ids outside [0, num_nodes)')

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

    def tagged_idx(self, reverse, class_id):
        """The neighbour array of `csr(reverse)` with each neighbour's (degree, class) table row in the top byte (entry = node | row << 24):
        what the struct-stage kernels read in table mode.  Cached per direction and class-id tensor."""
        cache = self.__dict__.setdefault('_tagged', {})
        hit = cache.get(reverse)
        if hit is None or hit[0] is not class_id:            # the entry holds the tensor itself: an address can be reused, an object cannot
            idx = self.csr(reverse)[1]
            hit = cache[reverse] = (class_id, (idx | (class_id[idx.long()] << 24)).to(torch.int32).contiguous())
        return hit[1]

    def count_self_loops(self):
        """Edges (v, v): the negative sampler draws E - self loops + N pairs (sampling.negative_sampling_device).  One host read-back,
        kept on the plan."""
        if getattr(self, 'num_self_loops', None) is None:
            self.num_self_loops = int((self.in_src == self.in_dst).sum().item()) if self.E > 0 else 0
        return self.num_self_loops

    def warm(self, xcls=None, quotient_stages=0):
        """Build, on the CURRENT stream, the per-batch caches a train step would otherwise build lazily inside the step (each with a
        host read-back): the first-stage (degree, class) table and the tagged lists of the half round after it, the heavy-row lists
        and their segments.  The batch prefetcher calls this on its worker's stream, beside the previous step."""
        self.count_self_loops()
        for rev in (False, True):
            self.heavy(rev)
            self.heavy_segments(rev)
        if self.has_levels:
            self.heavy_segments(True, inactive_only=True)
            self.heavy_segments(True, active_by_level=True)
        from . import ops
        if self.has_levels and ops.PACKED_ROWS == 2:
            self.order_rows                  # (default: a plan builds its packed sweep rows when it comes back for a second step, ops._sweep_rows)
        counts = sorted({int(c) for c in (quotient_stages if isinstance(quotient_stages, (tuple, list, set)) else [quotient_stages]) if int(c) > 0})
        if xcls is not None and self.N > 0 and counts and ops.QUOTIENT:
            if all([len(self.quotient(xcls, c)) > 0 for c in counts]):
                return self                  # the quotient stages replace the first-stage table and its tagged lists
        if xcls is not None and self.N > 0:
            first = self.first_stage_classes(xcls)
            if first is not None and first[1] <= 256 and self.N < (1 << 24):
                self.tagged_idx(True, first[0])
        return self

    HEAVY_ROW = 64      # csrc/struct_stage_x3_common.h: kHeavyRow

    def heavy(self, reverse):
        """(count, node ids int32 ascending) of the nodes with more than HEAVY_ROW neighbours in `csr(reverse)` — clock/reset-like
        nets.  The struct-stage launchers sum such lists in a pre-pass (one workgroup per node).  Built once per plan and direction
        (one host round trip); empty on most batches."""
        cache = self.__dict__.setdefault('_heavy', {})
        if reverse not in cache:
            p = self.csr(reverse)[0]
            nodes = torch.nonzero((p[1:] - p[:-1]) > self.HEAVY_ROW).reshape(-1).to(torch.int32)
            cache[reverse] = (int(nodes.numel()), nodes.contiguous())
        return cache[reverse]

    HEAVY_SEG = 512     # list entries per workgroup of the heavy-list kernels

    def heavy_segments(self, reverse, inactive_only=False, active_by_level=False):
        """The heavy nodes' lists (see `heavy`) cut into segments of HEAVY_SEG entries for the per-node pull kernels whose
        lists they are (reconstruction-loss backward over the positive edges, the sweep backward's pulls):
        dict(K, nodes, node_seg_ptr[K+1], S, seg_node[S], seg_e0[S], seg_e1[S]) of int32 device arrays, or None when there
        are none.  `inactive_only`: only nodes without an aggregator slot; `active_by_level`: only nodes WITH one, ordered by
        (level, id), plus host lists lvl_k_ptr / lvl_seg_ptr [num_levels + 1] (the nodes / segments of each level); both need
        set_levels."""
        cache = self.__dict__.setdefault('_heavy_seg', {})
        key = (reverse, inactive_only, active_by_level)
        if key not in cache:
            K, nodes = self.heavy(reverse)
            out = None
            lv = None
            if K > 0 and (inactive_only or active_by_level):
                act = self.gslot[nodes.long()] != NO_GATE
                nodes = nodes[act if active_by_level else ~act].contiguous()
                K = int(nodes.numel())
                if K > 0 and active_by_level:
                    lv_dev = self.level[nodes.long()].long()
                    order = torch.sort(lv_dev * (self.N + 1) + nodes.long()).indices          # by (level, id)
                    nodes = nodes[order].contiguous()
                    lv = lv_dev[order].cpu().numpy()
            if K > 0:
                p = self.csr(reverse)[0]
                e0 = p[nodes.long()].cpu().numpy().astype(np.int64)
                e1 = p[nodes.long() + 1].cpu().numpy().astype(np.int64)
                seg_node, s0, s1, nsp = [], [], [], [0]
                for k in range(K):
                    for b in range(int(e0[k]), int(e1[k]), self.HEAVY_SEG):
                        seg_node.append(k); s0.append(b); s1.append(min(b + self.HEAVY_SEG, int(e1[k])))
                    nsp.append(len(seg_node))
                i32 = dict(dtype=torch.int32, device=self.device)
                out = dict(K=K, nodes=nodes, node_seg_ptr=torch.tensor(nsp, **i32), S=len(seg_node), seg_node=torch.tensor(seg_node, **i32),
                           seg_e0=torch.tensor(s0, **i32), seg_e1=torch.tensor(s1, **i32))
                if lv is not None:
                    L = max(self.num_levels, 1)
                    cnt = np.bincount(lv, minlength=L)
                    kp = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
                    out['lvl_k_ptr'] = [int(v) for v in kp]
                    out['lvl_seg_ptr'] = [int(nsp[int(v)]) for v in kp]
            cache[key] = out
        return cache[key]

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
        if self.hip and self.N > 0:
            out = self._first_stage_classes_hip(xcls, max_classes)
            if out is not False:
                self._stage1 = (key, out)
                return out
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

    QUOTIENT_FRACTION = 2.0    # a half round runs on distinct rows only while they are at most N / 2 (measured: DESIGN.md 4.3)
    QUOTIENT_GROWTH = 8        # colours multiply by at least this per half round (config 2: x50, x226, x48): the next one is not even tried if C * 8 would not qualify

    QUOTIENT_MIN_NODES = 131072  # below: a step is launch-bound, the extra small launches cost more than the rows save (65,536 nodes: 7.40 ms with, 7.18 without; 262,144: 10.8 / 11.3)

    def quotient(self, xcls, max_stages, force=False):
        """Quotient stages of the structural encoder.  Every node starts from the same state (ones, digae_layer.py:260), so after
        half round t a node's row depends only on its colour: (feature class, colour after t-1, multiset of its neighbours' colours
        after t-1) — colour refinement over the alternating in- / out-CSR.  While the colours are few (early half rounds of any
        netlist: three gate types, bounded fan-in), the half round is computed on ONE row per colour.  Returns a list of stages
        t = 1, 2, ... (possibly empty), each a dict:
            C        colours after the stage;  cid [N] int32 colour of every node;  rev  the stage gathers over the out-CSR
            ptr [C+1], idx int32: a representative's neighbour list, entries = C + (colour after t-1) — rows of the stacked input
                     [own rows (C) | table of stage t-1 (C_prev)] the fp32 stage kernels read;  own [C] int64: colour after t-1;
                     ent_idx / own32: the same lists and own colours as plain int32 rows of the table of stage t-1 (the bf16x3
                     kernels read that table in place: own rows through own32)
            xcls [C] uint8, heavy: (count, rows with more than HEAVY_ROW neighbours)
            own_rows/own_levels, ent_rows/ent_levels: per colour of stage t-1 the representatives that own it / whose lists name it,
                     as segment tables of mgv_seg_sum (backward)
        and the last stage also `sum_levels` (class_sum_levels) for the per-colour sums of the per-node gradient.
        Grouping is proposed by a 64-bit key (two sums of random per-colour values over the neighbour list, folded with the own
        colour, class and degree) and then CHECKED exactly: every member against its colour's representative, list entry by list
        entry (lists sorted by colour); refinement stops at the first disagreement, so a key collision costs speed, never
        correctness.  Cached per xcls tensor.  `force`: also below QUOTIENT_MIN_NODES (per-graph parts for assemble_quotient).
        Every stage also keeps the pieces a batch-level assembly needs under 'raw' (references, no extra work)."""
        cache = getattr(self, '_quotient', None)
        if cache is None or cache[0] is not xcls:      # (the tensor itself: an address can be reused)
            cache = self._quotient = (xcls, {})
        if int(max_stages) in cache[1]:                 # per stage count: the source and the target encoder may ask for different ones
            return cache[1][int(max_stages)]
        N, dev = self.N, self.device
        stages = []
        self._check_status()                 # (edge ids outside [0, N) raise here, before the lists are read)
        if self.hip and self.QUOTIENT_DEVICE and not force:
            if N >= self.QUOTIENT_MIN_NODES and self.E > 0:
                stages = self._quotient_dev(xcls, int(max_stages))
        elif (force or N >= self.QUOTIENT_MIN_NODES) and self.E > 0:
            i64 = dict(dtype=torch.int64, device=dev)
            gen = torch.Generator(device=dev)
            gen.manual_seed(0x5EED5)
            xc = xcls.long()
            prev, Cp = torch.zeros(N, **i64), 1
            node = torch.arange(N, **i64)
            owners = {}                      # the row of every CSR slot, per direction
            for t_ in range(1, max_stages + 1):
                rev = t_ % 2 == 0
                p, idx = self.csr(rev)
                pl, il = p.long(), idx.long()
                deg = pl[1:] - pl[:-1]
                f = torch.randint(1, 1 << 62, (3, Cp + 1), generator=gen, **i64)
                if self.hip:
                    # grouping key and exact check by the plan builder's kernels (csrc/plan_build.hip: mgv_colour_keys / _check)
                    from . import _hip
                    from ._hip import ptr
                    prev32 = prev.to(torch.int32)
                    pairs = self._stage1_pairs(xcls) if (t_ == 1 and not force) else None
                    if pairs is not None:
                        mix = pairs[0].long()            # stage 1 IS the (in-degree, class) grouping: its table id is the key (as _quotient_dev)
                    else:
                        mix = torch.empty(N, **i64)
                        _hip.call('mgv_colour_keys', N, ptr(p), ptr(idx), ptr(prev32), ptr(f), Cp + 1, ptr(xcls), self._key_bits(Cp), ptr(mix))
                else:
                    owner = owners.get(rev)
                    if owner is None:
                        owner = owners[rev] = torch.repeat_interleave(node, deg) if rev else self.in_dst.long()
                    pn = prev[il]
                    h1 = torch.zeros(N, **i64).index_add_(0, owner, f[0][pn])
                    h2 = torch.zeros(N, **i64).index_add_(0, owner, f[1][pn])
                    mix = h1 * 0x1E3779B97F4A7C15 + h2 + f[2][prev] + xc * 0x632BE59BD9B4E019 + deg * 0x2545F4914F6CDD1D
                # ONE stable sort of the keys groups the nodes: colour = rank of the key, representative = first member of its run
                # (a scatter-min onto a handful of addresses costs 27 ms; torch.unique + a second sort by colour 1.6 ms more)
                skey, by_colour = torch.sort(mix, stable=True)
                first = torch.ones(N, dtype=torch.bool, device=dev)
                first[1:] = skey[1:] != skey[:-1]
                starts = torch.nonzero(first).reshape(-1)
                C = int(starts.numel())
                if C * (1.1 if force else self.QUOTIENT_FRACTION) > N:
                    break                    # (force: a graph's own stages go on until it is nearly fully refined; assembly applies the batch rule)
                inv = torch.empty(N, **i64)
                inv[by_colour] = torch.cumsum(first, 0) - 1
                members = torch.diff(starts, append=torch.tensor([N], **i64))
                rep = by_colour[starts]
                # exact check of the grouping (the key above only PROPOSES it): every member has its representative's feature class,
                # previous colour, degree and neighbour-colour multiset
                n_long = 1
                if self.hip:
                    flags = torch.zeros(2, dtype=torch.int32, device=dev)
                    inv32, rep32 = inv.to(torch.int32), rep.to(torch.int32)      # (named: a temporary would be freed, and its memory reused, before the launch)
                    _hip.call('mgv_colour_check', N, ptr(p), ptr(idx), ptr(prev32), ptr(xcls), ptr(inv32), ptr(rep32), ptr(flags))
                    bad, n_long = flags.tolist()
                    if bad:
                        break
                if n_long:                   # (CPU plan, or lists too long for the kernel's pairwise comparison): both lists sorted by colour
                    owner = owners.get(rev)
                    if owner is None:
                        owner = owners[rev] = torch.repeat_interleave(node, deg) if rev else self.in_dst.long()
                    if not self._colour_lists_equal(pl, il, owner, prev, inv, rep, xc, Cp):
                        break
                dr = deg[rep]
                rptr = torch.zeros(C + 1, **i64)
                rptr[1:] = torch.cumsum(dr, 0)
                n_ent = int(rptr[-1].item())
                row = torch.repeat_interleave(torch.arange(C, **i64), dr)
                k = torch.arange(n_ent, **i64) - rptr[row]
                ent = prev[il[pl[rep][row] + k]] if n_ent else torch.zeros(0, **i64)
                own = prev[rep]
                heavy = torch.nonzero(dr > self.HEAVY_ROW).reshape(-1).to(torch.int32)

                # per colour of stage t-1: the representatives that own it / the list entries that name it, as segment tables of
                # mgv_seg_sum (a colour like "AND gate" is named by thousands of entries: balanced, and summed in a fixed order)
                own_o, own_levels = self.class_sum_levels(own, Cp)
                ent_o, ent_levels = self.class_sum_levels(ent, Cp)
                raw = dict(rptr=rptr, ent=ent, own=own, row=row, heavy=heavy, own_sorted=(own_o, own_levels['counts']),
                           ent_sorted=(ent_o, ent_levels['counts']), cid_sorted=(by_colour, members))
                stages.append(dict(C=C, raw=raw, cid=inv.to(torch.int32).contiguous(), rev=rev, ptr=rptr.to(torch.int32).contiguous(),
                                   idx=(ent + C).to(torch.int32).contiguous() if n_ent else torch.zeros(1, dtype=torch.int32, device=dev),
                                   ent_idx=ent.to(torch.int32).contiguous() if n_ent else torch.zeros(1, dtype=torch.int32, device=dev),
                                   own=own, own32=own.to(torch.int32).contiguous(), xcls=xcls[rep].contiguous(),
                                   heavy=(int(heavy.numel()), heavy.contiguous()),
                                   own_rows=own_o, own_levels=own_levels,
                                   ent_rows=row[ent_o.long()].to(torch.int32).contiguous() if n_ent else torch.zeros(1, dtype=torch.int32, device=dev),
                                   ent_levels=ent_levels))
                prev, Cp = inv, C
                last_sorted = (by_colour, members)
                if not force and C * self.QUOTIENT_GROWTH * self.QUOTIENT_FRACTION > N:
                    break                    # colours multiply per half round: the next one would not qualify
            if stages:
                stages[-1]['sum_levels'] = self.class_sum_levels(stages[-1]['cid'], stages[-1]['C'], presorted=last_sorted)
        cache[1][int(max_stages)] = stages
        return stages

    def _colour_lists_equal(self, pl, il, owner, prev, inv, rep, xc, Cp):
        """Exact check of a grouping by sorting: every node has its representative's feature class, previous colour, degree and
        neighbour-colour multiset (both lists sorted by colour and compared entry by entry).  All arguments int64."""
        deg = pl[1:] - pl[:-1]
        pn = prev[il]
        ri = rep[inv]
        same = (xc == xc[ri]) & (prev == prev[ri]) & (deg == deg[ri])
        sorted_col = torch.sort(owner * (Cp + 1) + pn).values - owner * (Cp + 1)      # each node's list, colours ascending
        k_in_list = torch.arange(owner.numel(), dtype=torch.int64, device=self.device) - pl[:-1][owner]
        lists_same = sorted_col == sorted_col[(pl[:-1][ri[owner]] + k_in_list).clamp_(max=max(owner.numel() - 1, 0))]
        return bool(same.all()) and bool(lists_same.all())

    def _stage1_pairs(self, xcls):
        """The first refinement stage without a sort: every node starts from the same state, so its colour after the first half round
        is its (in-degree, feature class) pair — the table the plan builder's pair kernels number in one pass (ids ascending in the
        pair code, every id present).  None when that table does not apply (a degree above 255, or too many pairs)."""
        first = self._first_stage_classes_hip(xcls, 1 << 15)
        return first if (first is not None and first is not False) else None

    def _key_bits(self, Cp):
        """Bits of the grouping key a refinement stage sorts on: enough that two of the colours to expect (at most 1,024 per previous
        colour, at most N) share a key with probability below 2^-20 — early stages sort 40-odd bits instead of 63."""
        expect = min(self.N, int(Cp) * 1024)
        return min(63, max(24, 2 * max(expect, 2).bit_length() + 20))

    QUOTIENT_DEVICE = True      # hip plans: a stage's tables by the plan builder's kernels (_quotient_dev); False: the torch composition below

    def _sort_by_key_dev(self, keys, n, K):
        """(order int32 [n], members per key int32 [K]) of a STABLE sort of n int32 keys in [0, K): the counting sort of the tile
        builder while its (key, block) histogram stays small, else a radix sort over the keys' bits."""
        from . import _hip
        from ._hip import ptr
        i32 = dict(dtype=torch.int32, device=self.device)
        if n == 0:
            return torch.zeros(0, **i32), torch.zeros(K, **i32)
        order = torch.empty(n, **i32)
        if K <= 8192 and K * ((n + 1023) // 1024) <= 4 * n + 65536:
            ks = torch.empty(K + 1, **i32)
            n_s = _hip.call_value('mgv_count_sort_scratch_ints', n, K)
            scratch = torch.empty(n_s, **i32)
            _hip.call('mgv_count_sort_i32', n, ptr(keys), K, ptr(order), ptr(ks), ptr(scratch), n_s)
            return order, ks[1:] - ks[:-1]
        sk = torch.empty(n, **i32)
        t_i = _hip.call_value('mgv_sort_pairs_temp_ints', 4, n)
        temp = torch.empty(t_i, **i32)
        _hip.call('mgv_sort_pairs', 4, n, ptr(keys), ptr(sk), ptr(order), max((K - 1).bit_length(), 1), ptr(temp), t_i)
        counts = torch.empty(K, **i32)
        _hip.call('mgv_sorted_key_counts', n, ptr(sk), K, ptr(counts))
        return order, counts

    def _class_sum_levels_dev(self, counts, C, seg=64):
        """class_sum_levels' tables from the members per colour (int32 [C]) by the plan builder's kernels: per level one scan pass
        (segments, members, partial rows, colours going on), one read-back of the four totals, one fill pass."""
        from . import _hip
        from ._hip import ptr
        i32 = dict(dtype=torch.int32, device=self.device)
        counts0 = counts
        levels = []
        gid, G = None, C
        base, src_row = C, 0
        while True:
            w_i = _hip.call_value('mgv_seg_level_work_ints', G)
            work = torch.empty(w_i, **i32)
            _hip.call('mgv_seg_level_scan', G, ptr(counts), seg, ptr(work), w_i)
            total, _, n_partial, g_next = work[:4].tolist()
            sp = torch.empty(total + 1, **i32)
            out_row = None if (n_partial == 0 and G == C and base == C) else torch.empty(total, **i32)
            gid_n = torch.empty(g_next, **i32) if g_next else None
            counts_n = torch.empty(g_next, **i32) if g_next else None
            _hip.call('mgv_seg_level_fill', G, total, ptr(gid), seg, base, ptr(work), ptr(sp), ptr(out_row), ptr(gid_n), ptr(counts_n))
            levels.append((total, sp, out_row, src_row))
            if n_partial == 0:
                return dict(C=C, rows=base, levels=levels, counts=counts0)
            gid, counts, G = gid_n, counts_n, g_next
            src_row, base = base, base + n_partial

    def _quotient_dev(self, xcls, max_stages):
        """The stages of `quotient` with every table built by the plan builder's kernels (csrc/plan_build.hip, radix_sort.hip): per
        stage the grouping keys, ONE radix sort of them (the permutation as int32), runs -> colours / representatives, the exact
        check, the representatives' lists in previous colours, and the segment tables of the colour-level sums (stable sort by
        previous colour + per-level scans).  Host read-backs per stage: the colour count, the check's flags, the list length, and
        four totals per table level.  Same numbering, same tables as the torch composition (GPU test, field by field)."""
        from . import _hip
        from ._hip import ptr
        N, dev = self.N, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        i64 = dict(dtype=torch.int64, device=dev)
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x5EED5)
        stages = []
        prev32, Cp = torch.zeros(N, **i32), 1
        t_sort = _hip.call_value('mgv_sort_pairs_temp_ints', 8, N)
        n_grp = 2 * N + N // 2048 + 66
        last = None
        for t_ in range(1, max_stages + 1):
            rev = t_ % 2 == 0
            p, idx = self.csr(rev)
            f = torch.randint(1, 1 << 62, (3, Cp + 1), generator=gen, **i64)
            pairs = self._stage1_pairs(xcls) if t_ == 1 else None
            if pairs is not None:
                # stage 1 = the (in-degree, class) table (exact by construction: no keys, no sort, no check); every list names the one
                # shared start row
                cid, C, rptr, _, xrep = pairs
                if C * self.QUOTIENT_FRACTION > N:
                    break
                dr = rptr[1:] - rptr[:-1]
                n_ent = int(rptr[-1].item())
                own32 = torch.zeros(C, **i32)
                ent = torch.zeros(max(n_ent, 1), **i32)
                row = torch.repeat_interleave(torch.arange(C, **i32), dr.long())
                if n_ent == 0:
                    row = torch.zeros(1, **i32)
                heavy = torch.nonzero(dr > self.HEAVY_ROW).reshape(-1).to(torch.int32)
                stages.append(self._stage_tables_dev(C, cid, rev, rptr, own32, xrep, ent, row, n_ent, int(heavy.numel()), heavy, Cp))
                prev32, Cp, last = cid, C, None
                if C * self.QUOTIENT_GROWTH * self.QUOTIENT_FRACTION > N:
                    break
                continue
            mix, skey = torch.empty(N, **i64), torch.empty(N, **i64)
            by_colour, temp = torch.empty(N, **i32), torch.empty(t_sort, **i32)
            bits = self._key_bits(Cp)
            _hip.call('mgv_colour_keys', N, ptr(p), ptr(idx), ptr(prev32), ptr(f), Cp + 1, ptr(xcls), bits, ptr(mix))
            _hip.call('mgv_sort_pairs', 8, N, ptr(mix), ptr(skey), ptr(by_colour), bits, ptr(temp), t_sort)
            cid, starts, rep = torch.empty(N, **i32), torch.empty(N + 1, **i32), torch.empty(N, **i32)
            flags = torch.zeros(3, **i32)                       # [grouping refuted, lists too long for the kernel's check, colours]
            scratch = torch.empty(n_grp, **i32)
            _hip.call('mgv_colour_groups', N, ptr(skey), ptr(by_colour), ptr(cid), ptr(starts), ptr(rep), ptr(flags[2:]), ptr(scratch), n_grp)
            C = int(flags[2].item())
            if C * self.QUOTIENT_FRACTION > N:
                break
            _hip.call('mgv_colour_check', N, ptr(p), ptr(idx), ptr(prev32), ptr(xcls), ptr(cid), ptr(rep), ptr(flags))
            bad, n_long = flags[:2].tolist()
            if bad:
                break
            if n_long:                       # lists longer than the kernel compares pairwise: the sort-based check
                pl, il = p.long(), idx.long()
                owner = torch.repeat_interleave(torch.arange(N, **i64), pl[1:] - pl[:-1]) if rev else self.in_dst.long()
                if not self._colour_lists_equal(pl, il, owner, prev32.long(), cid.long(), rep[:C].long(), xcls.long(), Cp):
                    break
            small = torch.empty(C + 2, **i32)                   # rptr [C + 1], then the heavy-row count
            rptr, own32, xrep = small[:C + 1], torch.empty(C, **i32), torch.empty(C, dtype=torch.uint8, device=dev)
            n_rr = C + C // 2048 + 66
            scratch = torch.empty(n_rr, **i32)
            _hip.call('mgv_colour_rep_rows', C, ptr(rep), ptr(p), ptr(prev32), ptr(xcls), self.HEAVY_ROW, ptr(rptr), ptr(own32), ptr(xrep),
                      ptr(small[C + 1:]), ptr(scratch), n_rr)
            n_ent, n_heavy = small[C:].tolist()
            ent, row = torch.empty(max(n_ent, 1), **i32), torch.empty(max(n_ent, 1), **i32)
            if n_ent:
                _hip.call('mgv_colour_rep_lists', C, ptr(rep), ptr(p), ptr(idx), ptr(prev32), ptr(rptr), ptr(ent), ptr(row))
            else:
                ent.zero_()
            heavy = (torch.nonzero((rptr[1:] - rptr[:-1]) > self.HEAVY_ROW).reshape(-1).to(torch.int32) if n_heavy
                     else torch.zeros(0, **i32))
            stages.append(self._stage_tables_dev(C, cid, rev, rptr, own32, xrep, ent, row, n_ent, int(n_heavy), heavy, Cp))
            prev32, Cp = cid, C
            last = (by_colour, starts)
            if C * self.QUOTIENT_GROWTH * self.QUOTIENT_FRACTION > N:
                break                        # colours multiply per half round: the next one would not qualify
        if stages:
            C = stages[-1]['C']
            if last is None:                 # (the pair-table stage is the last one: its members by colour through the counting sort)
                by_colour, counts = self._sort_by_key_dev(stages[-1]['cid'], N, C)
            else:
                by_colour, counts = last[0], last[1][1:C + 1] - last[1][:C]
            stages[-1]['sum_levels'] = (by_colour, self._class_sum_levels_dev(counts, C))
        return stages

    def _stage_tables_dev(self, C, cid, rev, rptr, own32, xrep, ent, row, n_ent, n_heavy, heavy, Cp):
        """A stage's dict from its colours and representatives' lists: the stable sorts by previous colour and the segment tables."""
        own_o, own_counts = self._sort_by_key_dev(own32, C, Cp)
        ent_o, ent_counts = self._sort_by_key_dev(ent, n_ent, Cp)
        return dict(C=C, cid=cid, rev=rev, ptr=rptr, idx=(ent + C) if n_ent else ent, ent_idx=ent,
                    own=own32.long(), own32=own32, xcls=xrep, heavy=(int(n_heavy), heavy),
                    own_rows=own_o, own_levels=self._class_sum_levels_dev(own_counts, Cp),
                    ent_rows=row.index_select(0, ent_o) if n_ent else ent, ent_levels=self._class_sum_levels_dev(ent_counts, Cp))

    def assemble_quotient(self, parts, node_off, max_stages):
        """Quotient stages of a BATCH from its graphs' own stages (`parts[g]` = GraphPlan(graph g).quotient(..., force=True), cached by
        the loader: a dataset's graphs come back every epoch in other batches, their colour refinement need not).  Graphs share
        nothing, so the disjoint union of their colourings is a valid colouring of the batch (colours are never merged ACROSS graphs:
        a few more rows than the batch-level refinement finds, the same exactness — the check ran per graph).  Pure index
        arithmetic: concatenation with offsets (node ids by `node_off`, colour ids by the running sums of the graphs' colour counts;
        every graph's stage-1 lists name the ONE shared all-ones row) and the segment tables of mgv_seg_sum rebuilt from the
        concatenated, already colour-sorted orders (class_sum_levels(presorted=...): no sort).  Stages are cut where a graph has
        none left, or where the batch's colours exceed N / QUOTIENT_FRACTION; nothing below QUOTIENT_MIN_NODES.  Installs the result
        as this plan's quotient cache for `max_stages` and returns it."""
        N, dev = self.N, self.device
        i64 = dict(dtype=torch.int64, device=dev)
        counts = sorted({int(c) for c in (max_stages if isinstance(max_stages, (tuple, list, set)) else [max_stages]) if int(c) > 0})
        S = min([len(p) for p in parts] + [max(counts + [0])])
        G = len(parts)
        stages = []
        if N >= self.QUOTIENT_MIN_NODES and self.E > 0:
            prev_off = [0] * G               # offset of graph g's previous-stage colours (stage 1: the single shared row)
            Cp = 1
            for t in range(S):
                st = [p[t] for p in parts]
                Cs = [s_['C'] for s_ in st]
                C = sum(Cs)
                if C * self.QUOTIENT_FRACTION > N:
                    break
                coff = [0] * G
                for g in range(1, G):
                    coff[g] = coff[g - 1] + Cs[g - 1]
                n_ents = [int(s_['raw']['ent'].numel()) for s_ in st]
                eoff = [0] * G
                for g in range(1, G):
                    eoff[g] = eoff[g - 1] + n_ents[g - 1]
                n_ent = sum(n_ents)
                n_nodes = [int(node_off[g + 1]) - int(node_off[g]) for g in range(G)]

                def spread(offs, lens, total):
                    # the per-graph offsets, one per element of the concatenation (ONE launch, no read-back: the size is known here)
                    return torch.repeat_interleave(torch.tensor(offs, **i64), torch.tensor(lens, **i64), output_size=total)

                def catoff(ts, off_vec):
                    flat = (torch.cat(ts) if len(ts) > 1 else ts[0]).long()
                    return flat + off_vec if off_vec is not None else flat
                by_node_c = spread(coff, n_nodes, N)
                by_rep_c, by_rep_e = spread(coff, Cs, C), spread(eoff, Cs, C)
                by_rep_p = spread(prev_off, Cs, C) if t > 0 else None
                by_ent_c = spread(coff, n_ents, n_ent)
                by_ent_p = spread(prev_off, n_ents, n_ent) if t > 0 else None
                cid = catoff([s_['cid'] for s_ in st], by_node_c)
                rptr = torch.cat([catoff([s_['raw']['rptr'][:-1] for s_ in st], by_rep_e), torch.tensor([n_ent], **i64)])
                ent = catoff([s_['raw']['ent'] for s_ in st], by_ent_p)
                own = catoff([s_['raw']['own'] for s_ in st], by_rep_p)
                row = catoff([s_['raw']['row'] for s_ in st], by_ent_c)
                n_heavy = [int(s_['raw']['heavy'].numel()) for s_ in st]
                heavy = catoff([s_['raw']['heavy'] for s_ in st], spread(coff, n_heavy, sum(n_heavy)) if sum(n_heavy) else None).to(torch.int32)
                if t == 0:
                    # one previous colour for everybody: every representative / entry belongs to group 0, in index order
                    own_pre = (torch.arange(C, dtype=torch.int32, device=dev), torch.tensor([C], **i64))
                    ent_o = torch.arange(n_ent, dtype=torch.int32, device=dev)
                    ent_pre = (ent_o, torch.tensor([n_ent], **i64))
                else:
                    own_pre = (catoff([s_['raw']['own_sorted'][0] for s_ in st], by_rep_c), torch.cat([s_['raw']['own_sorted'][1] for s_ in st]))
                    ent_o = catoff([s_['raw']['ent_sorted'][0] for s_ in st], spread(eoff, n_ents, n_ent))
                    ent_pre = (ent_o, torch.cat([s_['raw']['ent_sorted'][1] for s_ in st]))
                own_o, own_levels = self.class_sum_levels(None, Cp, presorted=own_pre)
                ent_o, ent_levels = self.class_sum_levels(None, Cp, presorted=ent_pre)
                zero1 = torch.zeros(1, dtype=torch.int32, device=dev)
                cid_sorted = (catoff([s_['raw']['cid_sorted'][0] for s_ in st], spread([int(v) for v in node_off[:G]], n_nodes, N)), torch.cat([s_['raw']['cid_sorted'][1] for s_ in st]))
                stages.append(dict(C=C, cid=cid.to(torch.int32).contiguous(), rev=st[0]['rev'], ptr=rptr.to(torch.int32).contiguous(),
                                   idx=(ent + C).to(torch.int32).contiguous() if n_ent else zero1,
                                   ent_idx=ent.to(torch.int32).contiguous() if n_ent else zero1,
                                   own=own, own32=own.to(torch.int32).contiguous(), xcls=torch.cat([s_['xcls'] for s_ in st]).contiguous(),
                                   heavy=(int(heavy.numel()), heavy.contiguous()), own_rows=own_o, own_levels=own_levels,
                                   ent_rows=row[ent_o.long()].to(torch.int32).contiguous() if n_ent else zero1, ent_levels=ent_levels,
                                   _cid_sorted=cid_sorted))
                prev_off, Cp = coff, C
        out = {}
        for c in counts:
            lst = list(stages[:c])
            if lst:
                last = dict(lst[-1])
                last['sum_levels'] = self.class_sum_levels(last['cid'], last['C'], presorted=last['_cid_sorted'])
                lst[-1] = last
            out[c] = lst
        xc = getattr(self, 'xcls', None)
        cache = getattr(self, '_quotient', None)
        if cache is None or cache[0] is not xc:
            cache = self._quotient = (xc, {})
        cache[1].update(out)
        return out

    def assemble_quotient_merged(self, parts, gcols, node_off, max_stages):
        """Like assemble_quotient, but colours are MERGED across the batch's graphs through their dataset-wide global ids
        (`gcols[g][t]` [C_g] int64 from ColourDictionary.globals_of): the result is the batch-level colour refinement itself (same
        partition as GraphPlan.quotient finds), at the price of one `unique` + one argsort over the graphs' colours per half round
        (not over the nodes), two sorts for the segment tables and one sort of the nodes for the last stage's sums."""
        N, dev = self.N, self.device
        i64 = dict(dtype=torch.int64, device=dev)
        counts = sorted({int(c) for c in (max_stages if isinstance(max_stages, (tuple, list, set)) else [max_stages]) if int(c) > 0})
        S = min([len(p) for p in parts] + [max(counts + [0])])
        G = len(parts)
        stages = []
        if N >= self.QUOTIENT_MIN_NODES and self.E > 0:
            n_nodes = [int(node_off[g + 1]) - int(node_off[g]) for g in range(G)]
            inv_prev, prev_off, Cp = None, [0] * G, 1
            zero1 = torch.zeros(1, dtype=torch.int32, device=dev)

            def spread(offs, lens, total):
                return torch.repeat_interleave(torch.tensor(offs, **i64), torch.tensor(lens, **i64), output_size=total)
            for t in range(S):
                st = [p[t] for p in parts]
                Cs = [s_['C'] for s_ in st]
                Ctot = sum(Cs)
                coff = [0] * G
                for g in range(1, G):
                    coff[g] = coff[g - 1] + Cs[g - 1]
                n_ents = [int(s_['raw']['ent'].numel()) for s_ in st]
                eoff = [0] * G
                for g in range(1, G):
                    eoff[g] = eoff[g - 1] + n_ents[g - 1]
                uniq, inv = torch.unique(torch.cat([gcols[g][t] for g in range(G)]), return_inverse=True)
                C = int(uniq.numel())
                if C * self.QUOTIENT_FRACTION > N:
                    break
                # the first colour (in concatenation order) of every merged colour represents it
                order = torch.sort(inv, stable=True).indices
                members = torch.bincount(inv, minlength=C)
                rep = order[torch.cumsum(members, 0) - members]
                cid = inv[torch.cat([s_['cid'] for s_ in st]).long() + spread(coff, n_nodes, N)]
                start_cc = torch.cat([s_['raw']['rptr'][:-1] for s_ in st]) + spread(eoff, Cs, Ctot)
                len_cc = torch.cat([s_['raw']['rptr'][1:] - s_['raw']['rptr'][:-1] for s_ in st])
                ent_cc = torch.cat([s_['raw']['ent'] for s_ in st])
                own_cc = torch.cat([s_['raw']['own'] for s_ in st])
                if t > 0:                    # previous colours: the graph's numbering -> concatenation index -> merged id of stage t-1
                    ent_cc = inv_prev[ent_cc + spread(prev_off, n_ents, sum(n_ents))]
                    own_cc = inv_prev[own_cc + spread(prev_off, Cs, Ctot)]
                dr = len_cc[rep]
                rptr = torch.zeros(C + 1, **i64)
                rptr[1:] = torch.cumsum(dr, 0)
                n_ent = int(rptr[-1].item())
                row = torch.repeat_interleave(torch.arange(C, **i64), dr, output_size=n_ent)
                ent = ent_cc[start_cc[rep][row] + (torch.arange(n_ent, **i64) - rptr[row])] if n_ent else torch.zeros(0, **i64)
                own = own_cc[rep]
                heavy = torch.nonzero(dr > self.HEAVY_ROW).reshape(-1).to(torch.int32)
                own_o, own_levels = self.class_sum_levels(own, Cp)
                ent_o, ent_levels = self.class_sum_levels(ent, Cp)
                stages.append(dict(C=C, cid=cid.to(torch.int32).contiguous(), rev=st[0]['rev'], ptr=rptr.to(torch.int32).contiguous(),
                                   idx=(ent + C).to(torch.int32).contiguous() if n_ent else zero1,
                                   ent_idx=ent.to(torch.int32).contiguous() if n_ent else zero1,
                                   own=own, own32=own.to(torch.int32).contiguous(), xcls=torch.cat([s_['xcls'] for s_ in st])[rep].contiguous(),
                                   heavy=(int(heavy.numel()), heavy.contiguous()), own_rows=own_o, own_levels=own_levels,
                                   ent_rows=row[ent_o.long()].to(torch.int32).contiguous() if n_ent else zero1, ent_levels=ent_levels))
                inv_prev, prev_off, Cp = inv, coff, C
        out = {}
        for c in counts:
            lst = list(stages[:c])
            if lst:
                last = dict(lst[-1])
                last['sum_levels'] = self.class_sum_levels(last['cid'], last['C'])
                lst[-1] = last
            out[c] = lst
        xc = getattr(self, 'xcls', None)
        cache = getattr(self, '_quotient', None)
        if cache is None or cache[0] is not xc:
            cache = self._quotient = (xc, {})
        cache[1].update(out)
        return out

    def class_sum_levels(self, cid, C, seg=64, presorted=None):
        """Segment tables of mgv_seg_sum for per-colour row sums: (order [N] int32 = nodes sorted by colour, tables) with
        tables = dict(C, rows, levels=[(n_seg, seg_ptr int32, out_row int32 | None, src_row)]).  The sums live in ONE buffer of
        `rows` rows: the first C are the colours' sums, partial rows follow.  Level 1 cuts every colour's run of members into
        segments of <= seg: a colour that fits one segment writes its final row, the others leave one partial row per segment
        (colour by colour, in order) from row C on; level l+1 does the same with the partial rows level l left at `src_row`
        — only for the colours that still have more than one — until none is left."""
        dev = self.device
        i64 = dict(dtype=torch.int64, device=dev)
        if presorted is not None:            # (items already sorted by colour, members per colour)
            order, counts = presorted[0].to(torch.int32).contiguous(), presorted[1]
        else:
            srt = torch.sort(cid.long(), stable=True)
            order = srt.indices.to(torch.int32).contiguous()
            # members per colour from the sorted run boundaries (a histogram with atomics costs ~1 ms on 2.7 M entries, this 30 us)
            bounds = torch.searchsorted(srt.values, torch.arange(C + 1, **i64))
            counts = bounds[1:] - bounds[:-1]
        levels = []
        counts0 = counts
        gid = torch.arange(C, **i64)         # the colours still being summed
        base, src_row = C, 0
        while True:
            G = int(gid.numel())
            nseg = torch.clamp((counts + seg - 1) // seg, min=1)         # (a colour nobody carries still gets its zero row)
            cls = torch.repeat_interleave(torch.arange(G, **i64), nseg)
            total = int(cls.numel())
            first = torch.cumsum(nseg, 0) - nseg
            start = torch.cumsum(counts, 0) - counts
            sp = torch.empty(total + 1, **i64)
            sp[:total] = start[cls] + seg * (torch.arange(total, **i64) - first[cls])
            sp[total] = counts.sum()
            multi = nseg > 1
            seg_multi = multi[cls]
            n_partial = int(seg_multi.sum().item())
            if n_partial == 0 and G == C and base == C:
                out_row = None                                           # every colour is one segment: segment s writes row s
            else:
                out_row = torch.where(seg_multi, base + torch.cumsum(seg_multi, 0) - 1, gid[cls]).to(torch.int32).contiguous()
            levels.append((total, sp.to(torch.int32).contiguous(), out_row, src_row))
            if n_partial == 0:
                return order, dict(C=C, rows=base, levels=levels, counts=counts0)
            gid, counts = gid[multi], nseg[multi]
            src_row, base = base, base + n_partial

    def _first_stage_classes_hip(self, xcls, max_classes):
        from . import _hip
        from ._hip import ptr
        N, dev = self.N, self.device
        i32 = dict(dtype=torch.int32, device=dev)
        work = torch.empty(2 * 65537 + 64, **i32)
        cid = torch.empty(N, **i32)
        cls_deg = torch.zeros(65536, **i32)
        cls_x = torch.zeros(65536, dtype=torch.uint8, device=dev)
        status = torch.empty(1, **i32)
        xc = xcls.contiguous()
        _hip.call('mgv_plan_pairs', N, ptr(self.in_ptr), ptr(xc), ptr(work), ptr(work[65537:]), ptr(work[2 * 65537:]), ptr(cid), ptr(cls_deg),
                  ptr(cls_x), ptr(status))
        C, bad = int(work[65537 + 65536].item()), int(status.item())
        if bad:
            return False                       # a degree above 255: the torch path handles it
        if not 0 < C <= max_classes:
            return None
        d = cls_deg[:C].long()
        p = torch.zeros(C + 1, dtype=torch.int64, device=dev)
        p[1:] = torch.cumsum(d, 0)             # C entries: a dozen
        idx = torch.zeros(max(int(p[-1].item()), 1), **i32)
        return (cid, C, p.to(torch.int32).contiguous(), idx, cls_x[:C].contiguous())

    def _drop_level_caches(self):
        """Caches derived from gslot / level (and the tagged neighbour arrays): stale once the levels are set again, which
        `data.plan_of` does when a batch meets a model with another gate set."""
        self.__dict__.pop('_heavy_seg', None)
        self.__dict__.pop('_tagged', None)
        self.__dict__.pop('_slot_nodes', None)
        self.__dict__.pop('_persist_roles', None)
        self.__dict__.pop('_order_rows', None)
        self.__dict__.pop('_sweep_steps', None)

    def _set_levels_hip(self, gate, forward_level, gate_ids):
        from . import _hip
        from ._hip import ptr
        self._check_status()                 # before anything consumes the CSR
        self._drop_level_caches()
        N, E, dev, T = self.N, self.E, self.device, len(gate_ids)
        i32 = dict(dtype=torch.int32, device=dev)
        g = gate.reshape(-1).to(dev, torch.float32).contiguous()
        lv = forward_level.reshape(-1).to(dev, torch.int64).contiguous()
        if g.numel() != N or lv.numel() != N:
            raise ValueError('gate / forward_level must have one entry per node')
        tab = (ctypes.c_uint8 * 256)(*([NO_GATE] * 256))
        for s_, gid in enumerate(gate_ids):
            tab[int(gid)] = s_
        self.gslot = torch.empty(N, dtype=torch.uint8, device=dev)
        self.level = torch.empty(N, **i32)
        key = torch.empty(max(N, 1), **i32)
        flags = torch.zeros(2, **i32)                       # [max level, level-order violation]
        _hip.call('mgv_plan_keys', N, T, ptr(g), ptr(lv), tab, ptr(self.gslot), ptr(self.level), ptr(key), ptr(flags))
        _hip.call('mgv_plan_check_levels', E, ptr(self.in_src), ptr(self.in_dst), ptr(self.gslot), ptr(self.level), ptr(flags[1:]))
        maxlevel, bad = (int(v) for v in flags.tolist())    # the batch's first host round trip
        if bad:
            raise ValueError('forward_level is not a topological levelisation of edge_index')
        L = maxlevel + 1 if N > 0 else 0
        self.num_levels = L
        K = max(L, 1) * T
        if K > 8192:
            return False
        order = torch.empty(max(N, 1), **i32)
        key_start = torch.empty(K + 1, **i32)
        n_s = _hip.call_value('mgv_count_sort_scratch_ints', N, K)
        scratch = torch.empty(n_s, **i32)
        _hip.call('mgv_count_sort_i32', N, ptr(key), K, ptr(order), ptr(key_start), ptr(scratch), n_s)
        small = torch.empty(2 * (K + 1) + K // 2048 + 64 + max(L, 1) + 1, **i32)
        ntile, tile_first, scan_s = small[:K + 1], small[K + 1:2 * (K + 1)], small[2 * (K + 1):2 * (K + 1) + K // 2048 + 64]
        ltp_dev = small[2 * (K + 1) + K // 2048 + 64:]
        _hip.call('mgv_plan_tile_counts', K, ptr(key_start), ptr(ntile), ptr(tile_first), ptr(scan_s))
        n_active, num_tiles = int(key_start[K].item()), int(tile_first[K].item())         # second round trip
        self.order = order[:n_active]
        self.order_span = torch.empty(max(n_active, 1), 4, **i32)[:n_active]
        self.tile_start, self.tile_count, self.tile_slot = (torch.empty(max(num_tiles, 1), **i32)[:num_tiles] for _ in range(3))
        _hip.call('mgv_plan_tiles', K, T, max(L, 1), n_active, ptr(key_start), ptr(tile_first), ptr(self.order), ptr(self.in_ptr), ptr(self.out_ptr),
                  ptr(self.tile_start), ptr(self.tile_count), ptr(self.tile_slot), ptr(self.order_span), ptr(ltp_dev))
        # tiles grouped by slot (stable), for the per-slot weight-gradient pass of the backward sweep
        self.slot_tiles = torch.empty(max(num_tiles, 1), **i32)[:num_tiles]
        slot_start = torch.empty(T + 1, **i32)
        n_s2 = _hip.call_value('mgv_count_sort_scratch_ints', num_tiles, T)
        scratch2 = torch.empty(n_s2, **i32)
        _hip.call('mgv_count_sort_i32', num_tiles, ptr(self.tile_slot), T, ptr(self.slot_tiles), ptr(slot_start), ptr(scratch2), n_s2)
        self.key_tile_ptr = tile_first.clone()               # [L * T + 1]: tile range of every (level, slot) key (persistent sweep)
        host = torch.cat([ltp_dev[:max(L, 1) + 1], slot_start, self.key_tile_ptr]).tolist()  # third (last) round trip
        self.level_tile_ptr = host[:max(L, 1) + 1]
        self.slot_tile_ptr = host[max(L, 1) + 1:max(L, 1) + 1 + T + 1]
        self._key_tile_ptr_host = host[max(L, 1) + 1 + T + 1:]
        self.num_tiles, self.n_active, self.num_slots = num_tiles, n_active, T
        self.has_levels = True
        return self

    ROW_INTS = 32           # func_level_x3_common.h kRowInts
    ROW_OUT = 8

    @property
    def order_rows(self):
        """Packed sweep rows [n_active, 32] int32 (one 128-byte line per updated node, sweep order): {in0, in1, out0, out1}, the first 4
        in-edge sources (-1: none), the first 8 consumers as (node, in-CSR slot) pairs (node -1: none), their gate slots as bytes.
        The level kernels reach a tile's lists with this ONE load level instead of span -> CSR lists -> gslot."""
        rows = self.__dict__.get('_order_rows')
        if rows is not None:
            return rows
        n, dev = self.n_active, self.device
        if self.hip:
            from . import _hip
            from ._hip import ptr
            rows = torch.empty(max(n, 1), self.ROW_INTS, dtype=torch.int32, device=dev)[:n]
            _hip.call('mgv_plan_order_rows', n, ptr(self.order), ptr(self.in_ptr), ptr(self.in_src), ptr(self.out_ptr), ptr(self.out_dst),
                      ptr(self.out_slot), ptr(self.gslot), ptr(rows))
        else:
            on = self.order.long()
            rows = torch.full((n, self.ROW_INTS), -1, dtype=torch.int32, device=dev)
            rows[:, 0:4] = self.order_span
            if self.E > 0 and n > 0:
                in0, in1 = self.in_ptr[on].long(), self.in_ptr[on + 1].long()
                idx = in0[:, None] + torch.arange(4, device=dev)
                ok = idx < in1[:, None]
                rows[:, 4:8] = torch.where(ok, self.in_src[idx.clamp(max=self.E - 1)], torch.full_like(idx, -1, dtype=torch.int32))
                o0, o1 = self.out_ptr[on].long(), self.out_ptr[on + 1].long()
                oidx = o0[:, None] + torch.arange(self.ROW_OUT, device=dev)
                ook = oidx < o1[:, None]
                oc = oidx.clamp(max=self.E - 1)
                c = torch.where(ook, self.out_dst[oc], torch.full_like(oc, -1, dtype=torch.int32))
                sl = torch.where(ook, self.out_slot[oc], torch.full_like(oc, -1, dtype=torch.int32))
                rows[:, 8:24] = torch.stack([c, sl], 2).reshape(n, 2 * self.ROW_OUT)
                gc = torch.where(ook, self.gslot[c.clamp(min=0).long()], torch.full_like(oc, NO_GATE, dtype=torch.uint8))
                rows[:, 24:26] = gc.contiguous().view(torch.int32)
        self.__dict__['_order_rows'] = rows
        return rows

    def set_levels(self, gate, forward_level, gate_ids):
        """Bucket the nodes a Model updates: level >= 1 and gate id in `gate_ids` (list, position =
        aggregator slot).  Mirrors `layer_mask & <gate>_mask` of the reference level loop."""
        if self.hip:
            done = self._set_levels_hip(gate, forward_level, gate_ids)
            if done is not False:
                return self
        self._drop_level_caches()
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
        per_key = torch.bincount(t_level * T + t_slot, minlength=max(self.num_levels, 1) * T)
        ktp = torch.zeros(max(self.num_levels, 1) * T + 1, dtype=torch.int64, device=dev)
        ktp[1:] = torch.cumsum(per_key, 0)
        self.key_tile_ptr = ktp.to(torch.int32).contiguous()
        self._key_tile_ptr_host = [int(v) for v in ktp.tolist()]
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

    def persist_roles(self, max_grid):
        """Workgroup sets of the persistent sweep kernels (csrc/sweep_persist_x3.hip): wg_begin[T + 1] — workgroups
        [wg_begin[g], wg_begin[g + 1]) stay with aggregator slot g for the whole sweep.  A slot gets as many workgroups as its
        widest level has tiles when all slots fit `max_grid` (one tile per workgroup and level), else a share of `max_grid`
        proportional to its tile total (at least one).  None when the plan has no tiles.  Cached per max_grid."""
        cache = self.__dict__.setdefault('_persist_roles', {})
        if max_grid in cache:
            return cache[max_grid]
        T, L = self.num_slots, max(self.num_levels, 1)
        ktp = np.asarray(self._key_tile_ptr_host, dtype=np.int64)
        cnt = (ktp[1:] - ktp[:-1]).reshape(L, T)
        tot, widest = cnt.sum(0), cnt.max(0)
        roles = None
        if int(tot.sum()) > 0 and max_grid >= int((tot > 0).sum()):
            if int(widest.sum()) <= max_grid:
                w = widest.copy()
            else:
                share = tot * (max_grid / float(tot.sum()))
                w = np.minimum(np.maximum(np.floor(share).astype(np.int64), (tot > 0).astype(np.int64)), widest)
                # hand the remaining workgroups to the slots with the most tiles per workgroup; take back from the richest if over
                while int(w.sum()) < max_grid and bool((w < widest).any()):
                    load = np.where(w < widest, tot / np.maximum(w, 1), -1.0)
                    w[int(np.argmax(load))] += 1
                while int(w.sum()) > max_grid:
                    load = np.where(w > 1, tot / np.maximum(w, 1), np.inf)
                    w[int(np.argmin(load))] -= 1
            roles = [0] + [int(v) for v in np.cumsum(w)]
        cache[max_grid] = roles
        return roles

    def slot_nodes(self):
        """Node ids (int64, ascending) of every aggregator slot's updated nodes: what a round >= 2 of the sweep gathers to form
        W_hh h_prev per gate type.  Cached per levelisation."""
        if '_slot_nodes' not in self.__dict__:
            assert self.has_levels
            self._slot_nodes = [torch.nonzero(self.gslot == s_).reshape(-1) for s_ in range(self.num_slots)]
        return self._slot_nodes
