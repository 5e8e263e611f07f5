"""One-process-per-GPU plumbing for the counting path.

The path shards by independent units (genome ranges / chains / samples): every
rank stages its own records and counts its own intervals; count vectors are never
exchanged.  The only collective is a tiny all-reduce of summary totals (RCCL over
xGMI with the ``nccl`` backend; ``gloo`` on CPU in the tests).  Float totals are
reduced in fixed rank order so the result does not depend on the ring schedule.
"""
import os

import numpy as np


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init(backend, device=None):
    """Join the process group described by the torchrun environment (no-op for world size 1)."""
    import torch.distributed as dist
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        kwargs = {}
        if backend == "nccl" and device is not None:
            kwargs["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


def shard_chains(n_chains, rank, world):
    """Contiguous, balanced shard of chain indices for `rank` (chains are independent units)."""
    bounds = np.linspace(0, n_chains, world + 1).astype(np.int64)
    return np.arange(bounds[rank], bounds[rank + 1])


def allreduce_int_totals(values, device="cpu"):
    """Sum a small vector of int64 totals over all ranks (exact)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def reduce_float_totals_ordered(values, device="cpu"):
    """Sum float64 totals over ranks in FIXED rank order (all-gather, then a left-to-right sum),
    so center-mapping totals are reproducible bit for bit."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return t.tolist()
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    acc = torch.zeros_like(t)
    for p in parts:  # rank order
        acc = acc + p
    return acc.tolist()


def max_over_ranks(x, device="cpu"):
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class GenomePartition(object):
    """Cut one job into `world` contiguous genome ranges balanced by record count.

    Rank ``r`` gets (a) from every file the records that start inside its range, extended to the
    left by the largest reference span of any record (read-only duplication: a read that starts
    before the cut can still map inside the range), and (b) every queried segment *cut at the
    range borders*.  Cutting is exact for all five mapping rules: a read is counted at a position
    it aligns to, so whenever it counts inside a piece it also overlaps that piece, i.e. the
    reference's ``fetch`` would have returned it for the piece (genome_array.py:800-823), and
    records keep their relative order inside a rank (center sums stay bit-identical).
    No count vector crosses ranks; per-chain statistics are completed with one small all-reduce
    (:func:`allreduce_chain_sums`).

    `segments` is a dict with ``tid, start, end, strand, out_off, out_step, row_stride`` (the
    arguments of ``pc_plan_create``, e.g. ``IntervalTable.plan_arrays()``).
    """

    def __init__(self, files, segments, world, sample=1 << 20):
        self.files = list(files)
        self.world = int(world)
        seg = {k: np.asarray(v) for k, v in segments.items() if k in
               ("tid", "start", "end", "strand", "out_off", "out_step", "row_stride")}
        self.seg = seg
        ntid = len(self.files[0].references) if self.files else 0
        self.ntid = ntid
        # linear genome coordinate: contigs laid end to end, each as long as anything that touches it
        span_max = 1
        extent = np.zeros(ntid, np.int64)
        for f in self.files:
            if f.lengths is not None:
                extent = np.maximum(extent, np.asarray(f.lengths, np.int64)[:ntid])
            if f.n:
                end = f.ref_end()
                span_max = max(span_max, int((end - f.pos).max()))
                np.maximum.at(extent, f.tid, end)
        known = (seg["tid"] >= 0) & (seg["tid"] < ntid)
        if known.any():
            np.maximum.at(extent, seg["tid"][known], seg["end"][known].astype(np.int64))
        self.halo = span_max
        self.tid_off = np.zeros(ntid + 1, np.int64)
        np.cumsum(extent + 1, out=self.tid_off[1:])
        self._keys = [self.tid_off[f.tid] + f.pos for f in self.files]
        # cut points at record-count quantiles (of a sample of every file's keys)
        parts = []
        for k in self._keys:
            step = max(1, len(k) // sample)
            parts.append(k[::step])
        allk = np.sort(np.concatenate(parts)) if parts and sum(len(x) for x in parts) else np.zeros(0, np.int64)
        if len(allk) and self.world > 1:
            q = (np.arange(1, self.world) * len(allk)) // self.world
            self.cuts = allk[q].astype(np.int64)
        else:
            self.cuts = np.full(max(self.world - 1, 0), self.tid_off[-1], np.int64)
        self._build_pieces(known)

    @classmethod
    def from_cuts(cls, segments, world, tid_off, cuts, halo):
        """The same partition from GIVEN cut points (linear genome coordinates under `tid_off`) and halo, without
        any alignment file: what a rank builds when it only ever holds its own range of the job
        (``synth.JobLayout.cuts``; bench.py --gpus N).  Everything that concerns the segments -- pieces, rank-local
        layouts, owned elements, chain-sum plans -- works as usual; ``records`` / ``record_ranges`` need files."""
        self = cls.__new__(cls)
        self.files = []
        self.world = int(world)
        self.seg = {k: np.asarray(v) for k, v in segments.items() if k in
                    ("tid", "start", "end", "strand", "out_off", "out_step", "row_stride")}
        self.tid_off = np.asarray(tid_off, np.int64)
        self.ntid = len(self.tid_off) - 1
        self.halo = int(halo)
        self._keys = []
        self.cuts = np.asarray(cuts, np.int64)
        if len(self.cuts) != max(self.world - 1, 0):
            raise ValueError("GenomePartition.from_cuts: %d cut points for %d ranks" % (len(self.cuts), self.world))
        self._build_pieces((self.seg["tid"] >= 0) & (self.seg["tid"] < self.ntid))
        return self

    def cut_coordinates(self):
        """The cut points as ``(tid, pos)`` pairs."""
        t = np.searchsorted(self.tid_off, self.cuts, side="right") - 1
        t = np.clip(t, 0, max(self.ntid - 1, 0))
        return [(int(a), int(c - self.tid_off[a])) for a, c in zip(t, self.cuts)]

    def _build_pieces(self, known):
        seg, cuts = self.seg, self.cuts
        nseg = len(seg["tid"])
        ks = np.zeros(nseg, np.int64)
        ke = np.zeros(nseg, np.int64)
        ks[known] = self.tid_off[seg["tid"][known]] + seg["start"][known]
        ke[known] = self.tid_off[seg["tid"][known]] + seg["end"][known]
        r0 = np.searchsorted(cuts, ks, side="right")
        r1 = np.where(ke > ks, np.searchsorted(cuts, ke, side="left"), r0)
        r0[~known] = 0   # segments on contigs the files do not have are all-zero: rank 0 keeps them whole
        r1[~known] = 0
        npiece = (r1 - r0 + 1).astype(np.int64)
        owner = np.repeat(np.arange(nseg), npiece)
        first = np.zeros(nseg + 1, np.int64)
        np.cumsum(npiece, out=first[1:])
        rank = r0[owner] + (np.arange(len(owner)) - first[:-1][owner])
        lo_cut = np.where(rank > 0, cuts[np.clip(rank - 1, 0, max(len(cuts) - 1, 0))] if len(cuts) else 0, np.iinfo(np.int64).min)
        hi_cut = np.where(rank < self.world - 1, cuts[np.clip(rank, 0, max(len(cuts) - 1, 0))] if len(cuts) else 0,
                          np.iinfo(np.int64).max)
        a = np.maximum(ks[owner], lo_cut)
        b = np.minimum(ke[owner], hi_cut)
        kn = known[owner]
        delta = np.where(kn, a - ks[owner], 0)
        start = seg["start"][owner].astype(np.int64) + delta
        end = np.where(kn, start + (b - a), seg["end"][owner].astype(np.int64))
        step = seg["out_step"][owner].astype(np.int64)
        self.piece = dict(
            rank=rank.astype(np.int64), owner=owner, tid=seg["tid"][owner].astype(np.int32), start=start, end=end,
            strand=seg["strand"][owner].astype(np.uint8), out_off=seg["out_off"][owner].astype(np.int64) + step * delta,
            out_step=seg["out_step"][owner].astype(np.int8), row_stride=seg["row_stride"][owner].astype(np.int64))

    # ------------------------------------------------------------------ per rank
    def record_ranges(self, rank):
        """``[(i0, i1), ...]`` per file: the records rank `rank` stages."""
        out = []
        for k in self._keys:
            lo = 0 if rank == 0 else int(np.searchsorted(k, self.cuts[rank - 1] - self.halo, side="left"))
            hi = len(k) if rank == self.world - 1 else int(np.searchsorted(k, self.cuts[rank], side="left"))
            out.append((lo, max(hi, lo)))
        return out

    def records(self, rank):
        return [f.slice(i0, i1) for f, (i0, i1) in zip(self.files, self.record_ranges(rank))]

    def rank_regions(self, rank, references):
        """The genome range of `rank` -- its cut interval extended to the left by the halo -- as ``(chrom, start, end)``
        regions over `references` (the contig names): what the rank hands to a region read of ONE shared BAM file
        (``Engine.add_bam(path, regions=...)`` / ``read_bam(path, regions=...)``: only the BGZF members of the span the BAI index
        points to are read) instead of every rank decoding the whole file.  A read is returned for a region it OVERLAPS,
        so the rank gets at least the records :meth:`record_ranges` names (every record that starts inside the range or
        within a halo before it) plus those that reach in from further left: more read-only duplication, same counts
        (a read only counts at positions it aligns to, and the rank only counts positions of its own pieces)."""
        lo = 0 if rank == 0 else int(self.cuts[rank - 1]) - self.halo
        hi = int(self.tid_off[-1]) if rank == self.world - 1 else int(self.cuts[rank])
        lo = max(lo, 0)
        out = []
        for t in range(self.ntid):
            a, b = int(self.tid_off[t]), int(self.tid_off[t + 1])
            s, e = max(lo, a), min(hi, b)
            if e > s:
                out.append((references[t], s - a, e - a))
        return out

    def segments(self, rank, layout="global"):
        """Segment pieces of `rank` as ``pc_plan_create`` arrays.  ``layout="global"`` keeps the
        caller's output coordinates (every rank fills its part of one global layout: right for
        summed slices); ``"local"`` packs the pieces into a rank-local buffer (forward order, one
        block of ``rows x len`` per piece) -- see :meth:`scatter_local`."""
        m = np.nonzero(self.piece["rank"] == rank)[0]
        out = {k: self.piece[k][m] for k in ("tid", "start", "end", "strand", "out_off", "out_step", "row_stride")}
        out["piece_index"] = m
        if layout == "local":
            ln = out["end"] - out["start"]
            out["local_len"] = ln
        return out

    def local_plan_arrays(self, rank, rows=1):
        """Rank-local layout: pieces back to back, ``[rows, len]`` each, genome order inside."""
        sg = self.segments(rank, "local")
        ln = sg["local_len"]
        off = np.zeros(len(ln) + 1, np.int64)
        np.cumsum(ln * rows, out=off[1:])
        return dict(tid=sg["tid"], start=sg["start"], end=sg["end"], strand=sg["strand"], out_off=off[:-1].copy(),
                    out_step=np.ones(len(ln), np.int8), row_stride=ln.astype(np.int64), out_elems=int(off[-1]),
                    piece_index=sg["piece_index"])

    def _element_maps(self, rank, rows, keep=None):
        """Flat ``(local_index, global_index, piece_of_element)`` of every element of rank `rank`'s local output
        (pieces `keep`: boolean over the rank's pieces) -- built with repeats and running offsets, no loop over
        pieces (C4: 479 k exons x cuts)."""
        lp = self.local_plan_arrays(rank, rows)
        pi = lp["piece_index"]
        n = (lp["end"] - lp["start"]).astype(np.int64)
        sel = n > 0
        if keep is not None:
            sel &= keep
        j = np.nonzero(sel)[0]
        z = np.zeros(0, np.int64)
        if not len(j):
            return z, z, z, lp
        nj = n[j]
        cnt = nj * rows                                  # elements of every kept piece: rows x len, row-major
        tot = int(cnt.sum())
        first = np.zeros(len(j) + 1, np.int64)
        np.cumsum(cnt, out=first[1:])
        pj = np.repeat(np.arange(len(j)), cnt)           # kept piece of every element
        e = np.arange(tot, dtype=np.int64) - first[:-1][pj]   # element index inside its piece
        r, i = e // nj[pj], e % nj[pj]                   # row, position
        pc = self.piece
        g0 = pc["out_off"][pi[j]].astype(np.int64)
        st = pc["out_step"][pi[j]].astype(np.int64)
        rs = pc["row_stride"][pi[j]].astype(np.int64)
        li = lp["out_off"][j][pj] + e
        gi = g0[pj] + r * rs[pj] + st[pj] * i
        return li, gi, j[pj], lp

    def owned_elements(self, rank, rows=1, segments=None):
        """Where rank `rank`'s local output (see :meth:`local_plan_arrays`) sits in the caller's global
        layout: ``(local_index, global_index)`` element arrays, optionally only for the pieces of the
        given segment indices.  Summed slices (``out_step`` 0) are not covered."""
        keep = None
        if segments is not None:
            want = np.zeros(len(self.seg["tid"]), bool)
            want[np.asarray(segments, np.int64)] = True
            pi = np.nonzero(self.piece["rank"] == rank)[0]
            keep = want[self.piece["owner"][pi]]
        li, gi, _, _ = self._element_maps(rank, rows, keep)
        return li, gi

    def chain_sum_plan_arrays(self, rank, seg_chain, n_chains, rows=1):
        """``pc_plan_create`` arrays that make rank `rank` SUM each of its pieces into slot
        ``chain * rows + row`` of an ``int64[n_chains * rows]`` buffer (``out_step`` 0); one
        all-reduce of that buffer over the ranks completes the chains that straddle a cut
        (:func:`allreduce_device_sums`).  `seg_chain[s]` = chain of segment ``s``."""
        sg = self.segments(rank)
        owner = self.piece["owner"][sg["piece_index"]]
        k = len(owner)
        return dict(tid=sg["tid"], start=sg["start"], end=sg["end"], strand=sg["strand"],
                    out_off=np.asarray(seg_chain, np.int64)[owner] * rows, out_step=np.zeros(k, np.int8),
                    row_stride=np.ones(k, np.int64), out_elems=int(n_chains) * rows)

    def scatter_local(self, global_out, rank, local_out, rows=1):
        """Place a rank-local result (see :meth:`local_plan_arrays`) into the caller's global layout
        (host-side assembly of chains whose exons straddle a cut)."""
        li, gi, pj, lp = self._element_maps(rank, rows)
        if not len(li):
            return global_out
        st = self.piece["out_step"][lp["piece_index"][pj]]
        laid = st != 0
        global_out[gi[laid]] = np.asarray(local_out)[li[laid]]
        if not laid.all():                               # summed slices: every element of a (piece, row) adds to ONE slot
            np.add.at(global_out, gi[~laid], np.asarray(local_out)[li[~laid]])
        return global_out


def allreduce_chain_sums(values, device="cpu"):
    """Complete per-chain statistics whose chains straddle a range border: one all-reduce (RCCL
    ``ncclSum`` over xGMI on GPUs) of an int64 vector with one slot per chain."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.array(values, dtype=np.int64, copy=True)).to(device)  # never aliases the caller's array
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


class _DevicePointer(object):
    """``__cuda_array_interface__`` view of memory some engine owns (no copy, no ownership)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def device_tensor(ptr, n, dtype="int64"):
    """A ``torch`` tensor over `n` elements at device address `ptr` (e.g. ``Plan.device_ptr``): lets
    RCCL reduce the engine's output in place, without a host round trip."""
    import torch
    typestr = {"int64": "<i8", "float64": "<f8"}[dtype]
    return torch.as_tensor(_DevicePointer(ptr, n, typestr), device="cuda")


def allreduce_device_sums(ptr, n, dtype="int64", force=False):
    """In-place all-reduce (RCCL ``ncclSum`` over xGMI) of `n` int64 / float64 values at device
    address `ptr`.  The caller synchronises the engine's stream first: RCCL runs on torch's stream.
    Returns the tensor view.  `force`: run the collective in a world of one too (the hardware test of
    this path on a one-GPU box, tests/test_gpu_rccl.py)."""
    import torch.distributed as dist
    t = device_tensor(ptr, n, dtype)
    if dist.is_initialized() and (dist.get_world_size() > 1 or force):
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
