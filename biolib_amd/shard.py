"""Multi-GPU sharding of a batch of reads and the final count reduction.

The scan is embarrassingly parallel over sequences (every read / contig is scanned independently,
SURVEY.md §8e), so N GPUs = N independent shards: contiguous ranges of reads balanced by count,
one process per GPU, no data-path collective.  The only exchange is the optional reduction of the
per-shard digests at the end: counts are summed with one all-reduce (RCCL over xGMI when the
backend is "nccl", gloo in the CPU tests); XOR digests, for which RCCL has no reduction op, are
all-gathered (8 bytes each) and folded locally.
"""
import numpy as np


def shard_reads(n_reads, world_size, rank):
    """Contiguous, balanced range of reads for `rank`: (first_read, n_reads_of_rank)."""
    base, extra = divmod(int(n_reads), int(world_size))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def file_shard_reader(path, rank=None, world_size=None):
    """A Reader of this rank's part of a plain or bgzip'ed FASTA / FASTQ file that all ranks read between them (bl_reader_open_shard): no
    rank reads the whole file, none talks to another, and the parts' records in rank order are the file's records.  rank /
    world_size default to the initialised torch.distributed group (1 rank otherwise)."""
    from .scan import Reader

    if rank is None or world_size is None:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            rank, world_size = dist.get_rank(), dist.get_world_size()
        else:
            rank, world_size = 0, 1
    return Reader(path, shard=(int(rank), int(world_size)))


def shard_ranges(offsets, world_size, rank):
    """For ragged batches: split the sequences so that every rank gets about the same number of BASES.
    offsets: uint64[n_seqs+1].  Returns (first_seq, end_seq)."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    total = int(offsets[-1])
    lo = np.searchsorted(offsets, np.uint64(total * rank // world_size), side="left")
    hi = np.searchsorted(offsets, np.uint64(total * (rank + 1) // world_size), side="left")
    n_seqs = len(offsets) - 1
    lo = 0 if rank == 0 else min(int(lo), n_seqs)
    hi = n_seqs if rank == world_size - 1 else min(int(hi), n_seqs)
    return lo, max(hi, lo)


# ---- cutting ONE concatenation of sequences across ranks by BASES: a contig longer than a shard is split ----------
#
# The reference streams a contig of any length through one view (kmer_view.hpp:46-54, 181-202); a chromosome must therefore be
# able to use every GPU.  Rank r OWNS the windows (k-mers) whose FIRST base lies in [lo, hi) = its share of the bases.  To decide
# them it uploads the PIECE [lo - 1, hi + unit + w - 2): the (unit-1)+(w-1) bases behind hi that its last window reads, and the
# one base in front of lo, because whether the window at lo OPENS a minimizer occurrence depends on the window before it
# (the scan's range rule: a record belongs to the range that holds the first window of its occurrence).  The piece is an
# ordinary batch — sequence starts inside it keep their meaning, its first base counts as a start — whose origin is set to
# piece_lo (bl_batch_set_origin), so every reported position is a position in the whole and the ranks' records, concatenated
# in rank order, ARE the records of one scan over the whole.  No rank talks to another.

def base_range(total, world_size, rank):
    """[lo, hi) of the `total` bases of the concatenation that `rank` owns: contiguous, balanced to one base."""
    total, world_size, rank = int(total), int(world_size), int(rank)
    return total * rank // world_size, total * (rank + 1) // world_size


def contig_piece(offsets, lo, hi, unit, w, guard=0, tail=0):
    """What a rank uploads and scans to own the windows that start in [lo, hi).  offsets: uint64[n_seqs + 1] of the whole
    concatenation.  guard: windows scanned in front of lo and behind hi as well (super-k-mers: a range cuts groups, so groups
    are scanned whole and kept by their first window, see super_kmers_of_shard).
    tail: bases kept behind the last window's last base (1 for scans that drop the k-mer that ENDS its sequence, quirk Q1: the
    piece's own end must not look like the end of the sequence to the last owned k-mer).  Returns a dict:
      piece_lo, piece_hi   bases [piece_lo, piece_hi) of the whole form the batch; piece_lo is its origin
      offsets              uint64 sequence offsets of the batch (local; the piece's first base opens a sequence)
      first, n             the range of the batch to scan (local): the windows starting in [lo - guard, hi + guard)"""
    offsets = np.asarray(offsets, dtype=np.uint64)
    total = int(offsets[-1])
    lo, hi = int(lo), int(hi)
    span = int(unit) + int(w) - 1                       # bases of one window
    scan_lo, scan_hi = max(0, lo - int(guard)), min(total, hi + int(guard))
    piece_lo = max(0, scan_lo - 1)                      # the window in front of the first one decides whether that one opens an occurrence
    piece_hi = min(total, scan_hi + span - 1 + int(tail))
    if hi <= lo:
        piece_lo = piece_hi = scan_lo = scan_hi = min(lo, total)
    a = int(np.searchsorted(offsets, np.uint64(piece_lo), side="right"))
    b = int(np.searchsorted(offsets, np.uint64(piece_hi), side="left"))
    inner = offsets[a:b].astype(np.int64) - piece_lo     # sequence starts strictly inside the piece
    local = np.concatenate([[0], inner, [piece_hi - piece_lo]]).astype(np.uint64)
    return dict(piece_lo=piece_lo, piece_hi=piece_hi, offsets=local, first=scan_lo - piece_lo, n=scan_hi - scan_lo)


_RANGE = 1_500_000_000  # positions per scan call (a range holds at most 2^31)


def _piece_batch(ctx, bases, plan):
    """the rank's piece as a batch with its origin set; bases: the whole concatenation (uint8 array) or a callable(lo, hi) -> array"""
    piece = bases(plan["piece_lo"], plan["piece_hi"]) if callable(bases) else np.asarray(bases)[plan["piece_lo"]:plan["piece_hi"]]
    return ctx.upload(piece, plan["offsets"]).set_origin(plan["piece_lo"])


def _ranges(plan):
    a, end = plan["first"], plan["first"] + plan["n"]
    while a < end:
        yield a, min(_RANGE, end - a)
        a += _RANGE


def _cat(parts, keys):
    if not parts:
        return {k: np.zeros(0, np.uint8 if k in ("mm_pos", "sizes") else np.uint64) for k in keys} | {"count": 0}
    out = {k: np.concatenate([p[k] for p in parts]) for k in keys}
    out["count"] = int(sum(int(p["count"]) for p in parts))
    return out


def minimizers_of_shard(ctx, bases, offsets, unit, w, seed=0, canonical=False, rank=0, world_size=1):
    """This rank's minimizer records of the whole concatenation (values, positions — global —, hashes): rank order = position order."""
    lo, hi = base_range(int(np.asarray(offsets)[-1]), world_size, rank)
    plan = contig_piece(offsets, lo, hi, unit, w)
    if plan["n"] == 0:
        return _cat([], ("values", "positions", "hashes"))
    b = _piece_batch(ctx, bases, plan)
    parts = [b.minimizers(unit, w, seed=seed, canonical=canonical, first=a, n=n) for a, n in _ranges(plan)]
    b.close()
    return _cat(parts, ("values", "positions", "hashes"))


def syncmers_of_shard(ctx, bases, offsets, k, s, start_offset, end_offset, canonical=False, drop_last=False, rank=0, world_size=1):
    """This rank's syncmer positions (global).  A k-mer is decided by its own k bases: unit = s, w = k - s + 1 give the same piece."""
    lo, hi = base_range(int(np.asarray(offsets)[-1]), world_size, rank)
    plan = contig_piece(offsets, lo, hi, s, k - s + 1, tail=1 if drop_last else 0)
    if plan["n"] == 0:
        return _cat([], ("positions",))
    b = _piece_batch(ctx, bases, plan)
    parts = [b.syncmers(k, s, start_offset, end_offset, canonical=canonical, drop_last=drop_last, first=a, n=n) for a, n in _ranges(plan)]
    b.close()
    return _cat(parts, ("positions",))


def super_kmers_of_shard(ctx, bases, offsets, k, m, seed=0, canonical=False, rank=0, world_size=1):
    """This rank's super-k-mers (minimizers, first_pos — global —, mm_pos, sizes, hashes).  A scan range CUTS the groups at its
    ends (include/biolib_amd.h), so the rank scans w = k - m + 1 windows beyond both ends of its share — no group is longer —
    and keeps the groups whose first k-mer starts inside [lo, hi): every group is then whole on exactly one rank."""
    w = k - m + 1
    lo, hi = base_range(int(np.asarray(offsets)[-1]), world_size, rank)
    plan = contig_piece(offsets, lo, hi, m, w, guard=w)
    keys = ("minimizers", "first_pos", "mm_pos", "sizes", "hashes")
    if hi <= lo:
        return _cat([], keys)
    b = _piece_batch(ctx, bases, plan)
    # one call per <= 1.5 G positions would cut groups between the calls as well: overlap the calls by w and keep by first k-mer
    parts = []
    a, end = plan["first"], plan["first"] + plan["n"]
    while a < end:
        n = min(_RANGE, end - a)
        g = b.super_kmers(k, m, seed=seed, canonical=canonical, first=a, n=n)
        own_lo = max(lo, plan["piece_lo"] + a + (w if a > plan["first"] else 0))
        own_hi = min(hi, plan["piece_lo"] + a + n - (w if a + n < end else 0))
        keep = (g["first_pos"] >= np.uint64(own_lo)) & (g["first_pos"] < np.uint64(max(own_hi, own_lo)))
        parts.append({key: g[key][keep] for key in keys} | {"count": int(keep.sum())})
        if a + n >= end:
            break
        a += n - 2 * w
    b.close()
    return _cat(parts, keys)


def reduce_digests(local, device=None, group=None):
    """local: dict with integer 'count' (and optionally other *_count / sum_* keys, summed) and xor_* keys
    (folded with XOR).  Returns the whole-job digest on every rank.  Without an initialised process
    group the input is returned unchanged."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return dict(local)
    sums = sorted(k for k in local if not k.startswith("xor_"))
    xors = sorted(k for k in local if k.startswith("xor_"))
    out = {}
    if sums:
        # uint64 wrap-around sums travel as int64 bit patterns
        t = torch.tensor(np.array([local[k] for k in sums], dtype=np.uint64).view(np.int64), dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        vals = t.cpu().numpy().view(np.uint64)
        out.update({k: int(v) for k, v in zip(sums, vals)})
    if xors:
        t = torch.tensor(np.array([local[k] for k in xors], dtype=np.uint64).view(np.int64), dtype=torch.int64, device=device)
        world = dist.get_world_size(group)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t, group=group)
        acc = np.zeros(len(xors), dtype=np.uint64)
        for g in gathered:
            acc ^= g.cpu().numpy().view(np.uint64)
        out.update({k: int(v) for k, v in zip(xors, acc)})
    return out


def exchange_and_count(keys, partition, count, group=None):
    """Partitioned k-mer counting across ranks (the step after the scan in a distributed counter).

    keys       this rank's k-mers / minimizers: 1-D int64 tensor holding uint64 bit patterns (device tensor with RCCL,
               CPU tensor with gloo), duplicates allowed
    partition  callable(keys, parts) -> (bucketed_keys, counts[parts]): keys regrouped so that bucket b — the keys owned
               by rank b, hash64(key) % parts == b — is contiguous and buckets follow each other in rank order
               (Context.partition on the GPU)
    count      callable(keys) -> (distinct_keys, multiplicities) (Context.sort_count on the GPU)

    Every key travels at most once: one 8-byte-per-rank all-to-all of bucket sizes, then ONE variable-size all-to-all of
    the keys themselves (RCCL over xGMI: each GPU pair uses its own direct link, so the exchange is bound by
    keys_per_rank * 8 B * (world-1)/world over 7 links, not by a ring).  Returns this rank's (distinct_keys,
    multiplicities): the exact global multiplicity of every key this rank owns."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return count(keys)
    world = dist.get_world_size(group)
    bucketed, counts = partition(keys, world)
    return count(exchange(bucketed, counts, group))


def exchange(bucketed, counts, group=None):
    """The all-to-all itself.  bucketed: tensor whose rows (dim 0; 8-byte keys or 16-byte records as int64 pairs) are
    grouped by destination rank, counts[r] rows for rank r.  Returns the rows this rank received, grouped by source."""
    import torch
    import torch.distributed as dist

    counts = [int(c) for c in counts]
    # RCCL ("nccl") moves device tensors directly over xGMI; under gloo (CPU tests, one-GPU rehearsals) device tensors are
    # staged through the host
    staged = bucketed.is_cuda and dist.get_backend(group) == "gloo"
    wire = bucketed[: sum(counts)].contiguous()
    if staged:
        wire = wire.cpu()
    send = torch.tensor(counts, dtype=torch.int64, device=wire.device)
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)
    recv_sizes = [int(x) for x in recv.cpu().tolist()]
    inbox = torch.empty((sum(recv_sizes),) + tuple(wire.shape[1:]), dtype=wire.dtype, device=wire.device)
    dist.all_to_all_single(inbox, wire, output_split_sizes=recv_sizes, input_split_sizes=counts, group=group)
    return inbox.to(bucketed.device) if staged else inbox


def count_kmers_via_super_kmers(ctx, batch, k, m, seed=0, canonical=True, group=None, force_exchange=False):
    """Distributed exact k-mer counting the way super-k-mers are meant to be used:

      scan  -> super-k-mers of this rank's reads (bl_scan_super_kmers)
      pack  -> 16-byte sequence records (bl_pack_super_kmers)
      route -> bucket by minimizer hash % world (bl_partition_records), one all-to-all over RCCL / xGMI
      count -> bucket the received records by minimizer hash and count every bucket's k-mers in an LDS hash table
               (bl_count_super_kmers; the expand + sort + run-length path remains as its fallback for oversized buckets)

    All occurrences of a canonical k-mer share their minimizer value, so they meet on one rank and its local count is
    the global one.  ~1.8 bytes per input base cross the links instead of 8 bytes per k-mer.  Works without a process
    group (single GPU); force_exchange runs the all-to-all even at world size 1 (tests).  Returns (distinct k-mers,
    multiplicities) owned by this rank, as device tensors, in no particular order."""
    import torch.distributed as dist

    recs, hashes = batch.super_kmer_records(k, m, seed=seed, canonical=canonical)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_exchange):
        bucketed, counts = ctx.partition_records(hashes, recs, dist.get_world_size(group))
        recs = exchange(bucketed, counts, group)
    # minimizer buckets counted in LDS hash tables (bl_count_super_kmers): no global sort; the pairs come in no particular order
    return ctx.count_super_kmers(recs, k, m, seed=seed, canonical=canonical)
