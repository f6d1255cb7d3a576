"""Host-side binding of the scan entry points (thin; all work happens in the HIP library).

Device memory for outputs comes from torch (plumbing only): int64 CUDA tensors whose storage the
kernels fill with uint64 records; `.view(np.uint64)` on the host copy restores the type.
"""
import ctypes as C
import weakref

import numpy as np

from . import capi
from .capi import FLAG_CANONICAL, FLAG_DROP_LAST, FLAG_SYNC, BiolibError, Result, check


def hash64(value, seed=0):
    """hash::hash64::hash<uint64_t>(value, seed) (include/hash.hpp:55-59), bit-exact, on the host."""
    return int(capi.lib().bl_hash64_u64(int(value) & (2**64 - 1), int(seed) & (2**64 - 1)))


def _flags(canonical=False, drop_last=False, sync=False):
    return (FLAG_CANONICAL if canonical else 0) | (FLAG_DROP_LAST if drop_last else 0) | (FLAG_SYNC if sync else 0)


class Context:
    """One context per (process, GPU): owns the stream-ordered workspace of the scans."""

    def __init__(self, device=0, torch_stream=True, lanes=1):
        import torch

        if not torch.cuda.is_available():
            raise BiolibError(capi.BL_ERR_NO_DEVICE, "no GPU visible: biolib_amd has no CPU fallback")
        self._lib = capi.lib()
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        h = C.c_void_p()
        check(self._lib.bl_ctx_create(self.device, C.byref(h)))
        self._h = h
        self._batches = weakref.WeakSet()
        self._pending = []  # Result structs of asynchronous scans: the library fills them at the next sync
        self._borrowed = bool(torch_stream)
        if torch_stream:
            self.use_torch_stream()
        elif lanes != 1:
            # own streams only: consecutive asynchronous scans overlap (they must not share output arrays)
            check(self._lib.bl_ctx_set_lanes(self._h, int(lanes)))

    def use_torch_stream(self):
        """Run on torch's CURRENT stream (call again after switching torch streams): scans, set operations, torch kernels
        and RCCL collectives are then ordered with each other.  torch's default stream is the legacy stream, handle 0,
        which bl_ctx_set_stream honours as a stream (it does not mean "unset")."""
        import torch

        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.bl_ctx_set_stream(self._h, C.c_void_p(s)))
        self._borrowed = True

    def _inputs_ready(self):
        """A context on its own streams is not ordered with torch: tensors produced by pending torch work (kernels,
        all-to-all payloads) must be complete before the library reads them."""
        if not self._borrowed:
            import torch

            torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        if getattr(self, "_h", None):
            for b in list(self._batches):
                b.close()
            self._lib.bl_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.bl_ctx_sync(self._h))
        self._pending.clear()

    def last_scan_ms(self):
        ms = C.c_float()
        check(self._lib.bl_ctx_last_scan_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def set_exact_windows(self, on=True):
        """Later scans decide their windows on the 64-bit hashes themselves (bl_ctx_set_exact_windows): same records, for checks and A/B runs.
        Result.redone counts the tiles a default scan had to decide a second time."""
        check(self._lib.bl_ctx_set_exact_windows(self._h, 1 if on else 0))

    def set_option(self, name, value):
        """tuning / test switches by name (bl_ctx_set_option: "exact_windows", "lanes", "position_tiled", "emit_lds_bytes")"""
        check(self._lib.bl_ctx_set_option(self._h, name.encode(), int(value)))

    def kernel_timing(self, enable=True):
        check(self._lib.bl_ctx_kernel_timing(self._h, 1 if enable else 0))

    def _hold(self, result, flags):
        """an asynchronous scan fills `result` at a later sync: keep it alive until then (bounded)"""
        if flags & FLAG_SYNC:
            return
        if len(self._pending) >= 4096:
            self.sync()
        self._pending.append(result)

    def kernel_time(self):
        """(total ms, launches) of the scan kernels alone since kernel_timing(True)."""
        ms, n = C.c_double(), C.c_uint64()
        check(self._lib.bl_ctx_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)

    def mark(self):
        """a marker on the device's timeline behind everything issued so far (bl_ctx_mark): does not stop the device"""
        check(self._lib.bl_ctx_mark(self._h))

    def mark_times(self):
        """synchronises; milliseconds after the first marker at which the work in front of each marker had finished; forgets the markers"""
        n = C.c_uint32()
        check(self._lib.bl_ctx_mark_times(self._h, None, 0, C.byref(n)))  # capacity 0: how many there are (they stay)
        k = int(n.value)
        buf = (C.c_double * max(k, 1))()
        check(self._lib.bl_ctx_mark_times(self._h, buf, k, C.byref(n)))
        return [float(buf[i]) for i in range(k)]

    # ---- batches
    def upload(self, bases, offsets=None, read_len=0):
        """host bases -> device batch; sequences by `offsets`, or reads of one length `read_len` laid end to end (bl_batch_upload_reads)"""
        if isinstance(bases, str):
            bases = bases.encode()
        arr = np.frombuffer(bytes(bases), dtype=np.uint8) if isinstance(bases, (bytes, bytearray)) else np.ascontiguousarray(bases, dtype=np.uint8)
        if read_len:
            h = C.c_void_p()
            check(self._lib.bl_batch_upload_reads(self._h, arr.ctypes.data_as(C.c_void_p) if arr.size else None, arr.size, int(read_len), C.byref(h)))
            return Batch(self, h)
        offs = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        check(self._lib.bl_batch_upload(self._h, arr.ctypes.data_as(C.c_void_p) if arr.size else None, arr.size,
                                        None if offs is None else offs.ctypes.data_as(C.c_void_p), 0 if offs is None else len(offs) - 1, C.byref(h)))
        return Batch(self, h)

    def synth(self, seed, n_bases, read_len=0):
        h = C.c_void_p()
        check(self._lib.bl_batch_synth(self._h, int(seed), int(n_bases), int(read_len), C.byref(h)))
        return Batch(self, h)

    def from_text(self, text):
        """Parse raw FASTA / 4-line FASTQ text on the device (bl_batch_from_text).  text: bytes or a uint8 numpy array."""
        arr = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray, memoryview)) else np.ascontiguousarray(text, dtype=np.uint8)
        h, ns, nb = C.c_void_p(), C.c_uint64(), C.c_uint64()
        check(self._lib.bl_batch_from_text(self._h, arr.ctypes.data_as(C.c_void_p) if arr.size else None, arr.size, C.byref(h), C.byref(ns), C.byref(nb)))
        return Batch(self, h)

    def from_tensor(self, t, offsets=None, read_len=0):
        """Wrap a uint8 CUDA tensor of ASCII bases (not copied; must stay alive)."""
        self._inputs_ready()
        assert t.is_cuda and t.dtype.itemsize == 1 and t.is_contiguous()
        offs = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint64)
        h = C.c_void_p()
        check(self._lib.bl_batch_from_device(self._h, C.c_void_p(t.data_ptr()), t.numel(),
                                             None if offs is None else offs.ctypes.data_as(C.c_void_p), 0 if offs is None else len(offs) - 1,
                                             int(read_len), C.byref(h)))
        b = Batch(self, h)
        b._keep = t
        return b

    # ---- set operations on device k-mer lists (int64 CUDA tensors holding uint64 keys)
    def sort_unique(self, keys, n=None):
        """in-place sort + unique of keys[:n]; returns the number of distinct keys now at the front"""
        n = keys.numel() if n is None else int(n)
        self._inputs_ready()
        out = C.c_uint64()
        check(self._lib.bl_sort_unique_u64(self._h, C.c_void_p(keys.data_ptr()), n, C.byref(out)))
        return int(out.value)

    def jaccard(self, a, na, b, nb):
        """(|A n B|, |A u B|) of two sorted duplicate-free key arrays"""
        self._inputs_ready()
        i, u = C.c_uint64(), C.c_uint64()
        check(self._lib.bl_jaccard_sorted_u64(self._h, C.c_void_p(a.data_ptr()), int(na), C.c_void_p(b.data_ptr()), int(nb), C.byref(i), C.byref(u)))
        return int(i.value), int(u.value)

    def partition(self, keys, parts, seed=0, n=None):
        """(bucketed copy of keys[:n], sizes[parts]): bucket b = keys with hash64(key, seed) % parts == b, contiguous, in bucket order"""
        n = keys.numel() if n is None else int(n)
        self._inputs_ready()
        out = self.empty_u64(n)
        counts = (C.c_uint64 * int(parts))()
        check(self._lib.bl_partition_u64(self._h, C.c_void_p(keys.data_ptr()), n, int(parts), int(seed), C.c_void_p(out.data_ptr()), counts))
        return out[:n], [int(c) for c in counts]

    def sort_count(self, keys, n=None):
        """(distinct keys ascending, multiplicities) of keys[:n]; keys is sorted in place"""
        import torch

        n = keys.numel() if n is None else int(n)
        self._inputs_ready()
        check(self._lib.bl_sort_u64(self._h, C.c_void_p(keys.data_ptr()), n))
        uniq = self.empty_u64(n)
        mult = torch.empty(max(n, 1), dtype=torch.int32, device=self.torch_device)
        runs = C.c_uint64()
        check(self._lib.bl_count_sorted_u64(self._h, C.c_void_p(keys.data_ptr()), n, C.c_void_p(uniq.data_ptr()), C.c_void_p(mult.data_ptr()), C.byref(runs)))
        return uniq[: runs.value], mult[: runs.value]

    def partition_records(self, hashes, records, parts, n=None):
        """(bucketed copy of the 16-byte records[:n], sizes[parts]): bucket b = records with hashes[i] % parts == b"""
        import torch

        n = records.shape[0] if n is None else int(n)
        self._inputs_ready()
        out = torch.empty((max(n, 1), 2), dtype=torch.int64, device=self.torch_device)
        counts = (C.c_uint64 * int(parts))()
        check(self._lib.bl_partition_records(self._h, C.c_void_p(hashes.data_ptr()), C.c_void_p(records.data_ptr()), n, int(parts), C.c_void_p(out.data_ptr()), counts))
        return out[:n], [int(c) for c in counts]

    def expand_super_kmers(self, records, k, canonical=True, n=None):
        """the k-mers (device tensor) the packed super-k-mer records stand for, group after group"""
        n = records.shape[0] if n is None else int(n)
        self._inputs_ready()
        need = C.c_uint64()
        flags = FLAG_CANONICAL if canonical else 0
        rc = self._lib.bl_expand_super_kmers(self._h, C.c_void_p(records.data_ptr()), n, int(k), flags, None, 0, C.byref(need))
        if rc not in (0, capi.BL_ERR_CAPACITY):
            check(rc)
        out = self.empty_u64(need.value)
        if need.value:
            check(self._lib.bl_expand_super_kmers(self._h, C.c_void_p(records.data_ptr()), n, int(k), flags, C.c_void_p(out.data_ptr()), need.value, C.byref(need)))
        return out[: need.value]

    def read_file_u64(self, path, with_count=False):
        """a run file (raw sorted u64) or, with_count, an io::basic_store'd vector<uint64_t> of biolib -> device tensor"""
        n = C.c_uint64()
        check(self._lib.bl_file_count_u64(str(path).encode(), 1 if with_count else 0, C.byref(n)))
        out = self.empty_u64(n.value)
        check(self._lib.bl_read_file_u64(self._h, str(path).encode(), 1 if with_count else 0, C.c_void_p(out.data_ptr()), n.value, C.byref(n)))
        return out[: n.value]

    def merge_runs(self, paths):
        """the sorted union (duplicates kept) of biolib run files, on the device: what iterating the reference's
        external_memory_vector yields"""
        arr = (C.c_char_p * len(paths))(*[str(p).encode() for p in paths])
        total = C.c_uint64()
        rc = self._lib.bl_merge_runs_u64(self._h, arr, len(paths), None, 0, C.byref(total))
        if rc not in (0, capi.BL_ERR_CAPACITY):
            check(rc)
        out = self.empty_u64(total.value)
        if total.value:
            check(self._lib.bl_merge_runs_u64(self._h, arr, len(paths), C.c_void_p(out.data_ptr()), total.value, C.byref(total)))
        return out[: total.value]

    def count_super_kmers(self, records, k, m, seed=0, canonical=True, n=None, out=None):
        """(distinct k-mers, multiplicities) of the packed super-k-mer records, in no particular order: buckets by minimizer
        hash counted in LDS hash tables (bl_count_super_kmers), no global sort"""
        import torch

        n = records.shape[0] if n is None else int(n)
        self._inputs_ready()
        flags = FLAG_CANONICAL if canonical else 0
        need = C.c_uint64()
        cap = int(n * (k - m + 1) * 0.62) + 4096  # groups average ~ (w+1)/2 k-mers; retried with the exact need if short
        while True:
            if out is not None:  # caller-owned (keys int64, counts int32) tensors: no allocation on this path
                keys, cnts = out
                cap = min(keys.numel(), cnts.numel())
                out = None
            else:
                keys = self.empty_u64(cap)
                cnts = torch.empty(max(cap, 1), dtype=torch.int32, device=self.torch_device)
            rc = self._lib.bl_count_super_kmers(self._h, C.c_void_p(records.data_ptr()), n, int(k), int(m), int(seed), flags, C.c_void_p(keys.data_ptr()),
                                                C.c_void_p(cnts.data_ptr()), cap, C.byref(need))
            if rc == capi.BL_ERR_CAPACITY:
                cap = int(need.value) + 64
                continue
            check(rc)
            return keys[: need.value], cnts[: need.value]

    def probe_hbm(self, n_bytes=8 << 30, iters=5):
        """(read GB/s, copy GB/s) sustained by this device: read-only stream kernel and DtoD copy"""
        r, c = C.c_double(), C.c_double()
        check(self._lib.bl_probe_hbm(self._h, int(n_bytes), int(iters), C.byref(r), C.byref(c)))
        return r.value, c.value

    def bgzf_inflate(self, data):
        """inflate a buffer of whole BGZF members on the device (bl_bgzf_walk + bl_bgzf_inflate): (text as a numpy uint8 array,
        per-member status codes — 0 = sound)"""
        data = bytes(data)
        cap = len(data) // 26 + 1
        members = np.zeros(cap * 4, np.uint64)  # 32 bytes per member
        n, used, text = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(self._lib.bl_bgzf_walk(data, len(data), 0, 0, members.ctypes.data, cap, C.byref(n), C.byref(used), C.byref(text)))
        if used.value != len(data):
            raise BiolibError(capi.BL_ERR_INVALID, "the buffer ends inside a BGZF member")
        n, text = n.value, text.value
        ptrs = []
        try:
            for size in (len(data) + 8, 32 * max(n, 1), text + 16, 4 * max(n, 1)):
                p = C.c_void_p()
                check(self._lib.bl_device_alloc(self._h, size, C.byref(p)))
                ptrs.append(p)
            d_packed, d_members, d_text, d_status = ptrs
            check(self._lib.bl_copy_to_device(self._h, d_packed, data, len(data)))
            if n:
                check(self._lib.bl_copy_to_device(self._h, d_members, members.ctypes.data, 32 * n))
            check(self._lib.bl_bgzf_inflate(self._h, d_packed, len(data), d_members, n, d_text, text, d_status))
            self.sync()
            out = np.zeros(text, np.uint8)
            st = np.zeros(max(n, 1), np.uint32)
            if text:
                check(self._lib.bl_copy_to_host(self._h, out.ctypes.data, d_text, text))
            if n:
                check(self._lib.bl_copy_to_host(self._h, st.ctypes.data, d_status, 4 * n))
            return out, st[:n]
        finally:
            for p in ptrs:
                self._lib.bl_device_free(self._h, p)

    def clock_probe_start(self, duration_ms):
        """start measuring the shader clock the chip holds over the next duration_ms (beside whatever else runs)"""
        h = C.c_void_p()
        check(self._lib.bl_clock_probe_start(self._h, int(duration_ms), C.byref(h)))
        return h

    def clock_probe_finish(self, probe):
        ghz = C.c_double()
        check(self._lib.bl_clock_probe_finish(probe, C.byref(ghz)))
        return float(ghz.value)

    # ---- device arrays (torch plumbing)
    def empty_u64(self, n):
        import torch

        return torch.empty(max(int(n), 1), dtype=torch.int64, device=self.torch_device)

    def empty_u8(self, n):
        import torch

        return torch.empty(max(int(n), 1), dtype=torch.uint8, device=self.torch_device)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _host_u64(t, n):
    return t[:n].cpu().numpy().view(np.uint64).copy()


class Batch:
    """Device-resident sequences (bl_batch)."""

    def __init__(self, ctx, handle):
        self.ctx = ctx
        self._h = handle
        self._lib = ctx._lib
        self._keep = None
        ctx._batches.add(self)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self.ctx, "_h", None):  # a closed context has already destroyed its batches
                self._lib.bl_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def n_bases(self):
        return int(self._lib.bl_batch_n_bases(self._h))

    @property
    def n_seqs(self):
        return int(self._lib.bl_batch_n_seqs(self._h))

    def set_origin(self, origin):
        """This batch is a piece of a longer concatenation whose base `origin` is the batch's base 0: every position a scan reports
        (and folds into xor_pos) becomes origin + position inside the batch (bl_batch_set_origin; biolib_amd.shard cuts contigs with it)."""
        check(self._lib.bl_batch_set_origin(self._h, int(origin)))
        return self

    def download(self, first=0, n=None):
        n = self.n_bases - first if n is None else n
        out = np.empty(n, dtype=np.uint8)
        check(self._lib.bl_batch_download(self._h, first, n, out.ctypes.data_as(C.c_void_p)))
        return out

    # ---- raw (asynchronous) entry points: device tensors in, Result filled after ctx.sync()
    def kmers_raw(self, k, seed, flags, first=0, n=0, values=None, hashes=None, valid=None, result=None):
        result = result if result is not None else Result()
        self.ctx._hold(result, flags)
        check(self._lib.bl_scan_kmers(self.ctx._h, self._h, first, n, k, seed, flags, _ptr(values), _ptr(hashes), _ptr(valid), C.byref(result)))
        return result

    def minimizers_raw(self, unit, w, seed, flags, first=0, n=0, values=None, positions=None, hashes=None, capacity=0, result=None):
        result = result if result is not None else Result()
        self.ctx._hold(result, flags)
        check(self._lib.bl_scan_minimizers(self.ctx._h, self._h, first, n, unit, w, seed, flags, _ptr(values), _ptr(positions), _ptr(hashes),
                                           capacity, C.byref(result)))
        return result

    def hash_sample_raw(self, k, seed, threshold, flags, first=0, n=0, values=None, positions=None, hashes=None, capacity=0, result=None):
        result = result if result is not None else Result()
        self.ctx._hold(result, flags)
        check(self._lib.bl_scan_hash_sample(self.ctx._h, self._h, first, n, k, seed, threshold, flags, _ptr(values), _ptr(positions), _ptr(hashes),
                                            capacity, C.byref(result)))
        return result

    def super_kmers_raw(self, k, m, seed, flags, first=0, n=0, minimizers=None, first_pos=None, mm_pos=None, sizes=None, hashes=None,
                        capacity=0, result=None):
        result = result if result is not None else Result()
        self.ctx._hold(result, flags)
        check(self._lib.bl_scan_super_kmers(self.ctx._h, self._h, first, n, k, m, seed, flags, _ptr(minimizers), _ptr(first_pos), _ptr(mm_pos),
                                            _ptr(sizes), _ptr(hashes), capacity, C.byref(result)))
        return result

    def syncmers_raw(self, k, s, soff, eoff, seed, flags, first=0, n=0, positions=None, capacity=0, result=None):
        result = result if result is not None else Result()
        self.ctx._hold(result, flags)
        check(self._lib.bl_scan_syncmers(self.ctx._h, self._h, first, n, k, s, soff, eoff, seed, flags, _ptr(positions), capacity, C.byref(result)))
        return result

    # ---- convenience wrappers returning host numpy arrays
    def _span(self, first, n):
        end = self.n_bases if not n else min(self.n_bases, first + n)
        return max(end - first, 0)

    def kmers(self, k, seed=0, canonical=False, drop_last=False, first=0, n=0, arrays=True):
        span = self._span(first, n)
        c = self.ctx
        v = c.empty_u64(span) if arrays else None
        h = c.empty_u64(span) if arrays else None
        ok = c.empty_u8(span) if arrays else None
        r = self.kmers_raw(k, seed, _flags(canonical, drop_last, True), first, n, v, h, ok)
        out = r.as_dict()
        out["sum_hash"] = out.pop("xor_pos")
        if arrays:
            out.update(values=_host_u64(v, span), hashes=_host_u64(h, span), valid=ok[:span].cpu().numpy().copy())
        return out

    def _with_capacity(self, guess, run):
        cap = max(int(guess), 64)
        while True:
            try:
                return run(cap)
            except BiolibError as e:
                if e.code != capi.BL_ERR_CAPACITY:
                    raise
                cap = int(self._last_count) + 64

    def minimizers(self, unit, w, seed=0, canonical=False, first=0, n=0, capacity=None):
        span = self._span(first, n)
        guess = capacity if capacity is not None else int(span * 2.4 / (w + 1)) + 4096

        def run(cap):
            c = self.ctx
            v, p, h = c.empty_u64(cap), c.empty_u64(cap), c.empty_u64(cap)
            r = Result()
            try:
                self.minimizers_raw(unit, w, seed, _flags(canonical, False, True), first, n, v, p, h, cap, r)
            finally:
                self._last_count = r.count
            cnt = int(r.count)
            out = r.as_dict()
            out.update(values=_host_u64(v, cnt), positions=_host_u64(p, cnt), hashes=_host_u64(h, cnt))
            return out

        return self._with_capacity(guess, run)

    def hash_sample(self, k, seed=0, threshold=2**64 - 1, canonical=False, drop_last=False, first=0, n=0, device=False):
        """k-mers with hash64(value, seed) < threshold (hash_sampler over kmer_view).  device=True keeps the value tensor on the GPU."""
        span = self._span(first, n)
        cap = span + 64
        c = self.ctx
        v, p, h = c.empty_u64(cap), c.empty_u64(cap), c.empty_u64(cap)
        r = self.hash_sample_raw(k, seed, threshold, _flags(canonical, drop_last, True), first, n, v, p, h, cap)
        cnt = int(r.count)
        out = r.as_dict()
        if device:
            out.update(values_device=v, n=cnt)
        else:
            out.update(values=_host_u64(v, cnt), positions=_host_u64(p, cnt), hashes=_host_u64(h, cnt))
        return out

    def super_kmer_records(self, k, m, seed=0, canonical=False, first=0, n=0, fused=True):
        """(records int64[n,2], minimizer hashes int64[n]) on the device: the packed super-k-mers of the range, straight from the
        scan (bl_scan_super_kmer_records; fused=False: bl_scan_super_kmers + bl_pack_super_kmers, the same records)"""
        import torch

        span = self._span(first, n)
        c = self.ctx

        def run(cap):
            hs = c.empty_u64(cap)
            r = Result()
            if fused:
                recs = torch.empty((max(cap, 1), 2), dtype=torch.int64, device=c.torch_device)
                c._hold(r, _flags(canonical, False, True))
                try:
                    check(self._lib.bl_scan_super_kmer_records(c._h, self._h, int(first), int(n), int(k), int(m), int(seed), _flags(canonical, False, True),
                                                               C.c_void_p(recs.data_ptr()), C.c_void_p(hs.data_ptr()), int(cap), C.byref(r)))
                finally:
                    self._last_count = r.count
                cnt = int(r.count)
                return recs[:cnt], hs[:cnt]
            fp, sz, mp = c.empty_u64(cap), c.empty_u8(cap), c.empty_u8(cap)
            try:
                self.super_kmers_raw(k, m, seed, _flags(canonical, False, True), first, n, None, fp, mp, sz, hs, cap, r)
            finally:
                self._last_count = r.count
            cnt = int(r.count)
            recs = torch.empty((max(cnt, 1), 2), dtype=torch.int64, device=c.torch_device)
            check(self._lib.bl_pack_super_kmers(c._h, self._h, C.c_void_p(fp.data_ptr()), C.c_void_p(sz.data_ptr()), C.c_void_p(mp.data_ptr()), cnt, int(k), int(m),
                                                C.c_void_p(recs.data_ptr())))
            c.sync()
            return recs[:cnt], hs[:cnt]

        return self._with_capacity(int(span * 2.4 / (k - m + 2)) + 4096, run)

    def super_kmers(self, k, m, seed=0, canonical=False, first=0, n=0, capacity=None):
        span = self._span(first, n)
        guess = capacity if capacity is not None else int(span * 2.4 / (k - m + 2)) + 4096

        def run(cap):
            c = self.ctx
            mn, fp, hs = c.empty_u64(cap), c.empty_u64(cap), c.empty_u64(cap)
            mp, sz = c.empty_u8(cap), c.empty_u8(cap)
            r = Result()
            try:
                self.super_kmers_raw(k, m, seed, _flags(canonical, False, True), first, n, mn, fp, mp, sz, hs, cap, r)
            finally:
                self._last_count = r.count
            cnt = int(r.count)
            out = r.as_dict()
            out.update(minimizers=_host_u64(mn, cnt), first_pos=_host_u64(fp, cnt), mm_pos=mp[:cnt].cpu().numpy().copy(),
                       sizes=sz[:cnt].cpu().numpy().copy(), hashes=_host_u64(hs, cnt))
            return out

        return self._with_capacity(guess, run)

    def syncmers(self, k, s, start_offset, end_offset, seed=0, canonical=False, drop_last=False, first=0, n=0, positions=True, capacity=None):
        span = self._span(first, n)
        if not positions:
            r = self.syncmers_raw(k, s, start_offset, end_offset, seed, _flags(canonical, drop_last, True), first, n)
            return r.as_dict()
        guess = capacity if capacity is not None else int(span * 2.6 / (k - s + 1)) + 4096

        def run(cap):
            p = self.ctx.empty_u64(cap)
            r = Result()
            try:
                self.syncmers_raw(k, s, start_offset, end_offset, seed, _flags(canonical, drop_last, True), first, n, p, cap, r)
            finally:
                self._last_count = r.count
            out = r.as_dict()
            out.update(positions=_host_u64(p, int(r.count)))
            return out

        return self._with_capacity(guess, run)


class Reader:
    """FASTA / FASTQ (plain or gzip) reader of the library (bl_reader_*): host records or device batches."""

    def __init__(self, path, threads=0, shard=None):
        """threads: inflate workers for gzip and BGZF input (0 = one per core, at most 16); plain files use one read-ahead thread.
        shard=(rank, world): this reader takes part `rank` of a plain or BGZF file that `world` readers read between them (BGZF:
        device batches only; the parts' records in rank order are the file's records)"""
        self._lib = capi.lib()
        h = C.c_void_p()
        if shard is not None:
            check(self._lib.bl_reader_open_shard(str(path).encode(), int(shard[0]), int(shard[1]), C.byref(h)))
        else:
            check(self._lib.bl_reader_open_threads(str(path).encode(), int(threads), C.byref(h)))
        self._h = h

    @property
    def shard_range(self):
        """(first byte, end byte) of the file whose members are this reader's own (end = 2^64 - 1: to the end of the file)"""
        a, b = C.c_uint64(), C.c_uint64()
        check(self._lib.bl_reader_shard_range(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    @property
    def kind(self):
        """'plain', 'gzip' or 'bgzf'"""
        return self._lib.bl_reader_kind(self._h).decode()

    def close(self):
        if getattr(self, "_h", None):
            self._lib.bl_reader_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def records(self):
        """yield (name, sequence bytes) — host only, no GPU needed"""
        name, seq, n = C.c_char_p(), C.c_void_p(), C.c_uint64()
        while True:
            rc = self._lib.bl_reader_next_record(self._h, C.byref(name), C.byref(seq), C.byref(n))
            if rc == 1:
                return
            check(rc)
            yield name.value.decode("latin1"), C.string_at(seq.value, n.value) if n.value else b""

    def text_spans(self, max_bytes=0):
        """yield the decompressed text as bytes objects cut at record boundaries (what the device-side parser takes)"""
        p, n = C.c_void_p(), C.c_uint64()
        while True:
            rc = self._lib.bl_reader_next_text(self._h, int(max_bytes), C.byref(p), C.byref(n))
            if rc == 1:
                return
            check(rc)
            yield C.string_at(p.value, n.value)

    def device_batches(self, ctx, max_text_bytes=0):
        """yield Batch objects parsed ON THE DEVICE from spans of the decompressed text (regular FASTA / 4-line FASTQ only;
        names are not kept): parallel inflate -> one H2D copy -> bl_batch_from_text"""
        while True:
            b, ns, nb = C.c_void_p(), C.c_uint64(), C.c_uint64()
            check(self._lib.bl_reader_next_batch_device(ctx._h, self._h, int(max_text_bytes), C.byref(b), C.byref(ns), C.byref(nb)))
            if not b.value:
                return
            yield Batch(ctx, b)

    def batches(self, ctx, max_bases=0, names=True):
        """yield (Batch, names, offsets) of whole records holding at most max_bases bases each (names=False: the list stays
        empty — one call per name is what a Python loop over millions of short reads spends its time on)"""
        while True:
            b, ns, nb = C.c_void_p(), C.c_uint64(), C.c_uint64()
            check(self._lib.bl_reader_next_batch(ctx._h, self._h, int(max_bases), C.byref(b), C.byref(ns), C.byref(nb)))
            if not b.value:
                return
            offs_p, n2 = C.c_void_p(), C.c_uint64()
            check(self._lib.bl_reader_last_batch(self._h, None, C.byref(offs_p), C.byref(n2)))
            offs = np.ctypeslib.as_array(C.cast(offs_p, C.POINTER(C.c_uint64)), shape=(n2.value + 1,)).copy()
            nm = [self._lib.bl_reader_last_name(self._h, i).decode("latin1") for i in range(n2.value)] if names else []
            yield Batch(ctx, b), nm, offs
