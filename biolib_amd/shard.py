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
