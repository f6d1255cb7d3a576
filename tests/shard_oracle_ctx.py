"""TEST INFRASTRUCTURE: what biolib_amd.Context / Batch do for biolib_amd.shard, restated on the CPU oracle — so that the
host logic of the contig split (which bases a rank uploads, which range it scans, which records it keeps) runs where there
is no GPU.  The RANGE semantics of the C ABI are restated here from include/biolib_amd.h:
  * a scan of [first, first + n) reports the windows / k-mers whose first base lies in the range; bases behind it are read;
  * minimizers: a record belongs to the range that holds the FIRST WINDOW of its occurrence;
  * super-k-mers: a range CUTS groups at its ends;
  * reported positions = origin (bl_batch_set_origin) + position inside the batch.
The `-m gpu` twin of every test that uses this runs the same shard functions on real batches."""
import numpy as np

import oracle_lib as O


class OracleBatch:
    def __init__(self, bases, offsets):
        self.seq = np.ascontiguousarray(bases, dtype=np.uint8)
        self.offs = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.origin = 0

    def set_origin(self, origin):
        self.origin = int(origin)
        return self

    def close(self):
        pass

    def _end(self, first, n):
        return len(self.seq) if not n else min(len(self.seq), first + n)

    def _groups(self, k, m, seed, canonical):
        if len(self.seq) == 0:
            z = np.zeros(0, np.uint64)
            return z, z, np.zeros(0, np.uint8), np.zeros(0, np.uint8), z
        return O.super_kmers(self.seq, self.offs, k, m, seed, canonical)

    def minimizers(self, unit, w, seed=0, canonical=False, first=0, n=0):
        # one super-k-mer group per minimizer occurrence: its first k-mer is the occurrence's first window
        mn, fp, mp, sz, hs = self._groups(unit + w - 1, unit, seed, canonical)
        keep = (fp >= first) & (fp < self._end(first, n))
        pos = fp + mp.astype(np.uint64) + np.uint64(self.origin)
        return dict(count=int(keep.sum()), values=mn[keep], positions=pos[keep], hashes=hs[keep])

    def super_kmers(self, k, m, seed=0, canonical=False, first=0, n=0):
        mn, fp, mp, sz, hs = self._groups(k, m, seed, canonical)
        end = self._end(first, n)
        fp = fp.astype(np.int64)
        last = fp + sz.astype(np.int64) - 1
        nf, nl = np.maximum(fp, first), np.minimum(last, end - 1)
        keep = nf <= nl
        mpos = fp + mp.astype(np.int64)  # position of the minimizer: unchanged by the cut
        return dict(count=int(keep.sum()), minimizers=mn[keep], first_pos=(nf[keep] + self.origin).astype(np.uint64),
                    mm_pos=(mpos - nf)[keep].astype(np.uint8), sizes=(nl - nf + 1)[keep].astype(np.uint8), hashes=hs[keep])

    def syncmers(self, k, s, start_offset, end_offset, seed=0, canonical=False, drop_last=False, first=0, n=0):
        if len(self.seq) < k:
            return dict(count=0, positions=np.zeros(0, np.uint64))
        cnt, pos = O.syncmers(self.seq, self.offs, k, s, start_offset, end_offset, canonical, drop_last=drop_last)
        keep = (pos >= first) & (pos < self._end(first, n))
        return dict(count=int(keep.sum()), positions=pos[keep] + np.uint64(self.origin))


class OracleContext:
    def upload(self, bases, offsets=None):
        bases = np.asarray(bases, dtype=np.uint8)
        return OracleBatch(bases, np.array([0, len(bases)], np.uint64) if offsets is None else offsets)
