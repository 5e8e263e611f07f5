"""Length-stratified consumers of the counting path.

The reference's ``psite`` and ``phase_by_size`` scripts ask, per region and per read length
``k``, for ``ga.map_fn([reads of aligned length k], segment)`` (bin/psite.py:176-197,
bin/phase_by_size.py:183-198): one Python pass over the reads per segment and length.  Here the
mapping rule of the |BAMGenomeArray| is re-expressed as a
:class:`~plastid_amd.map_factories.StratifiedVariableFivePrimeMapFactory` (rows = read lengths)
and ALL regions and lengths are counted in one launch of the tile kernel, laid out directly in
the array the script wants (``[length, region, window column]`` for ``psite``).
"""
import numpy as np

from . import _lib
from .map_factories import (TABLE_LEN, FivePrimeMapFactory, StratifiedVariableFivePrimeMapFactory,
                            ThreePrimeMapFactory, VariableFivePrimeMapFactory)


def stratify_mapping(map_fn, min_len, max_len):
    """``(factory, row_valid, nrows)``: a stratified factory whose row ``k - min_len`` equals `map_fn`
    applied to the reads of aligned length ``k`` only, for the built-in point rules.

    ``row_valid[r]`` is False for lengths `map_fn` does not map at all (offset >= length, no
    usable table entry): the reference skips those reads, so their rows are zero -- the
    stratified rule itself would put them on the read's last base (map_factories.pyx:773-774),
    which is why the caller zeroes those rows."""
    min_len, max_len = int(min_len), int(max_len)
    if min_len < 1 or max_len < min_len or max_len >= TABLE_LEN - 1:
        raise ValueError("bad read length range [%s, %s]" % (min_len, max_len))
    lengths = np.arange(min_len, max_len + 1)
    if isinstance(map_fn, StratifiedVariableFivePrimeMapFactory) or isinstance(map_fn, VariableFivePrimeMapFactory):
        fw = np.asarray(map_fn.forward_offsets)[lengths]
        if isinstance(map_fn, StratifiedVariableFivePrimeMapFactory):
            inside = (lengths >= map_fn.min_length) & (lengths <= map_fn.max_length)
            off = np.where(fw < 0, lengths - 1, fw)   # its own "last base" rule, kept
            valid = inside
        else:
            off, valid = fw, fw >= 0
    elif isinstance(map_fn, FivePrimeMapFactory):
        off = np.full(len(lengths), map_fn.offset)
        valid = off < lengths
    elif isinstance(map_fn, ThreePrimeMapFactory):
        off = lengths - 1 - map_fn.offset
        valid = map_fn.offset < lengths
    else:
        raise TypeError("length-stratified counting needs a five-prime, three-prime or variable-offset mapping rule")
    od = {int(L): int(o) for L, o, v in zip(lengths, off, valid) if v}
    hi = max_len if max_len > min_len else max_len + 1   # the factory wants max > min (map_factories.pyx:716-717)
    return StratifiedVariableFivePrimeMapFactory(od, min_len, hi), np.asarray(valid, bool), hi - min_len + 1


def count_stratified(ga, min_len, max_len, tid, start, end, strand, out_off, out_step, out_elems):
    """One launch: int64 ``[max_len-min_len+1, out_elems]``; within a row, position ``start+i`` of a
    segment goes to ``out_off + out_step*i`` (the layout of ``pc_plan_create`` with the row stride
    set to `out_elems`).  Filters of `ga` apply as usual; normalisation does not."""
    if not ga._native() or ga.map_fn._kind == _lib.MAP_CENTER:
        raise TypeError("length-stratified counting needs a built-in point mapping rule")
    fac, valid, nrows = stratify_mapping(ga.map_fn, min_len, max_len)
    norm = ga._normalize
    ga._normalize = False
    try:
        ga._sync_engine()                     # filters -> exclusion bits, size filter
    finally:
        ga._normalize = norm
    eng = ga._engine
    fac._configure(eng)                       # the next ga call re-installs ga.map_fn (_sync_engine)
    nseg = len(tid)
    plan = eng.plan(tid, start, end, strand, out_off, out_step, np.full(nseg, out_elems, np.int64),
                    int(out_elems) * nrows, nrows)
    out = plan.count(np.int64).reshape(nrows, int(out_elems))
    plan.close()
    ga.map_fn._configure(eng)
    out = out[:int(max_len) - int(min_len) + 1]
    out[~valid] = 0
    return out


def _chain_layout(ga, chains, base_of_chain):
    """Segment arrays laying chain ``c`` out 5'->3' from ``base_of_chain[c]`` (roitools.pyx:3259-3271)."""
    tid, start, end, strand, off, step = [], [], [], [], [], []
    for ci, c in enumerate(chains):
        t = ga._chrom_index.get(c.chrom, -1) if len(c) else -1
        rev = c.c_strand == 2
        done = 0
        for seg in c:
            n = seg.end - seg.start
            tid.append(t); start.append(seg.start); end.append(seg.end); strand.append(c.c_strand)
            off.append(base_of_chain[ci] + (c.length - 1 - done if rev else done))
            step.append(-1 if rev else 1)
            done += n
    return (np.array(tid, np.int32), np.array(start, np.int64), np.array(end, np.int64), np.array(strand, np.uint8),
            np.array(off, np.int64), np.array(step, np.int8))


def counts_by_length(ga, chains, min_len, max_len):
    """Per chain an int64 ``[n_lengths, chain.length]`` array, 5'->3': row ``k - min_len`` is the
    count vector of the reads of aligned length ``k`` under ``ga``'s mapping rule."""
    lens = np.array([c.length for c in chains], np.int64)
    base = np.zeros(len(chains) + 1, np.int64)
    np.cumsum(lens, out=base[1:])
    tid, start, end, strand, off, step = _chain_layout(ga, chains, base[:-1])
    flat = count_stratified(ga, min_len, max_len, tid, start, end, strand, off, step, int(base[-1]))
    return [flat[:, base[i]:base[i + 1]] for i in range(len(chains))]


# ------------------------------------------------------------------------------ psite
def psite_raw_counts(ga, rois, alignment_offsets, window_size, min_len, max_len):
    """``raw_count_dict`` of ``psite.do_count`` (bin/psite.py:153-197): for every read length a
    masked float array ``[n_rois, window_size]``; row ``i`` holds, from column
    ``alignment_offsets[i]``, the 5'->3' count vector of ROI ``i`` for that length (NaN and masked
    elsewhere; ROI positions under the ROI's own masks are masked too).  `rois` are
    |SegmentChains| with their masks added."""
    n, W = len(rois), int(window_size)
    offs = [int(round(x)) for x in alignment_offsets]
    for i, roi in enumerate(rois):
        assert offs[i] + roi.length <= W                                   # bin/psite.py:174
    base = np.array([i * W + offs[i] for i in range(n)], np.int64)
    tid, start, end, strand, off, step = _chain_layout(ga, rois, base)
    flat = count_stratified(ga, min_len, max_len, tid, start, end, strand, off, step, n * W)
    out = {}
    covered = np.zeros((n, W), bool)
    roimask = np.zeros((n, W), bool)
    for i, roi in enumerate(rois):
        covered[i, offs[i]:offs[i] + roi.length] = True
        if roi._position_mask is not None:
            m = np.asarray(roi._position_mask, bool)
            roimask[i, offs[i]:offs[i] + roi.length] = m[::-1] if roi.c_strand == 2 else m
    for r, k in enumerate(range(int(min_len), int(max_len) + 1)):
        data = np.where(covered, flat[r].reshape(n, W).astype(float), np.nan)
        out[k] = np.ma.MaskedArray(data, mask=(~covered) | roimask, dtype=float)
    return out


def psite_profiles(raw_count_dict, norm_start, norm_end, min_counts, upstream_flank, aggregate=False):
    """Normalised matrices and the per-length metagene profile of ``psite.do_count``
    (bin/psite.py:199-241).  Returns ``(norm_count_dict, profile)`` where `profile` is a dict of
    columns: ``x``, ``"<k>-mers"`` and ``"<k>_regions_counted"``."""
    import warnings
    norm_count_dict = {}
    first = next(iter(raw_count_dict.values()))
    window_size = first.shape[1]
    profile = {"x": np.arange(-upstream_flank, window_size - upstream_flank)}
    for k, k_raw in raw_count_dict.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            denominator = np.nansum(k_raw[:, norm_start:norm_end], axis=1)
            norm_count_dict[k] = (k_raw.T.astype(float) / denominator).T
        norm_counts = np.ma.MaskedArray(norm_count_dict[k], mask=k_raw.mask)
        norm_counts.mask[np.isnan(norm_counts)] = True
        norm_counts.mask[np.isinf(norm_counts)] = True
        with warnings.catch_warnings():
            warnings.filterwarnings("ignore", ".*mean of empty.*", RuntimeWarning)
            try:
                if not aggregate:
                    prof = np.ma.median(norm_counts[denominator >= min_counts], axis=0)
                else:
                    prof = np.nansum(k_raw[denominator >= min_counts], axis=0)
            except (IndexError, ValueError):
                prof = np.zeros_like(profile["x"], dtype=float)
        profile["%s-mers" % k] = prof
        profile["%s_regions_counted" % k] = ((~norm_counts.mask)[denominator >= min_counts]).sum(0)
    return norm_count_dict, profile


# ------------------------------------------------------------------------------ phase_by_size
def phase_by_size(ga, cds_chains, read_lengths, codon_buffer=5, back_buffer=None, batch_positions=1 << 24):
    """Sub-codon phasing per read length (bin/phase_by_size.py:166-215): for every coding chain the
    5'->3' count vector of each length is cut into codons (a trailing partial codon is ignored),
    codons ``[codon_buffer:back_buffer]`` are kept (``back_buffer`` defaults to ``-codon_buffer`` as
    for annotation input, :162; ROI input uses -1, :145 -- and, as there, a buffer of 0 keeps
    nothing because ``-0 == 0``), and the three columns are summed over all chains.
    Returns ``{length: array([n0, n1, n2])}``."""
    read_lengths = sorted(int(k) for k in read_lengths)
    lo, hi = read_lengths[0], read_lengths[-1]
    sums = {k: np.zeros(3) for k in read_lengths}
    chains = [c for c in cds_chains if len(c) > 0]          # :176 only coding regions
    back = -int(codon_buffer) if back_buffer is None else int(back_buffer)
    i = 0
    while i < len(chains):
        j, tot = i, 0
        while j < len(chains) and (j == i or tot + chains[j].length <= batch_positions):
            tot += chains[j].length
            j += 1
        for c, mat in zip(chains[i:j], counts_by_length(ga, chains[i:j], lo, hi)):
            ncodon = c.length // 3
            cod = mat[:, :3 * ncodon].reshape(mat.shape[0], ncodon, 3)[:, int(codon_buffer):back, :].sum(1)
            for k in read_lengths:
                sums[k] += cod[k - lo]
        i = j
    return sums


def phase_table(phase_sums):
    """The columns of the script's phasing table (bin/phase_by_size.py:217-236): ``read_length``,
    ``reads_counted``, ``fraction_reads_counted``, ``phase0..2``."""
    lengths = np.array(sorted(phase_sums))
    counted = np.array([phase_sums[k].sum() for k in lengths]).astype(int)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = {"read_length": lengths, "reads_counted": counted,
               "fraction_reads_counted": counted.astype(float) / counted.sum()}
        vec = np.array([phase_sums[k].astype(float) / phase_sums[k].astype(float).sum() for k in lengths])
    for i in range(3):
        out["phase%s" % i] = vec[:, i]
    return out
