// bl_capi.hip — implementation of the C ABI declared in include/biolib_amd.h (host side).
// Contexts own a HIP stream and a small device workspace; batches own (or borrow) the base buffer
// plus a 1-bit-per-base sequence-start vector; scans enqueue  memset -> tile kernel -> digest fold
// -> async copy of the 72-byte result into a pinned slot, all on the context's stream.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/biolib_amd.h"
#include "bl_launch.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg)
{
    g_err = msg;
    return code;
}

}  // namespace

// used by the host-only translation units of the library (bl_ingest.cpp)
int bl_set_error(int code, const char* msg) { return fail(code, msg ? msg : ""); }
struct bl_ctx;
hipStream_t bl_ctx_stream(bl_ctx* ctx);
int bl_ctx_device(bl_ctx* ctx);

namespace {

#define BL_HIP(call)                                                                                       \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail(e_ == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP,                               \
                        std::string(#call) + ": " + hipGetErrorString(e_));                                \
    } while (0)

constexpr int RESULT_WORDS = 16;   // device/pinned result record (u64 words)
constexpr int RING = 256;          // pinned result slots (recycled one by one behind their own events, never by draining the context)
constexpr size_t HDR_BYTES = 256;  // ticket, error
constexpr size_t SHARD_BYTES = (size_t)bl::NSHARD * 8 * sizeof(unsigned long long);

struct Pending {
    bl_result* user;
    int slot;
    uint64_t capacity;
    bool has_capacity;
};

}  // namespace

// One execution lane: a stream plus the scratch its scans use.  With two lanes (bl_ctx_set_lanes) and the
// context's own streams, consecutive asynchronous scans alternate between the lanes and are staggered by an event: a
// scan's pass 1 starts when the previous scan's pass 1 has finished, i.e. beside that scan's pass 2.  Pass 1 is bound by
// VALU issue and leaves HBM idle, pass 2 is bound by HBM writes: together they run at the VALU rate of their combined
// instruction count (measured on MI355X: 346 -> 387 Gbp/s on the C3 workload; pass-2 residency is capped, see launch_scan_emit).  Consecutive scans then run concurrently,
// so they must not share output arrays.  Default: one lane.  A borrowed caller stream always uses lane 0.
struct Lane {
    hipStream_t own = nullptr;
    unsigned char* ws = nullptr;  // [hdr 256 B][shards][result]
    size_t ws_bytes = 0;
    unsigned char* tile_buf = nullptr;  // tile counts / prefixes
    size_t tile_buf_bytes = 0;
    uint16_t* slot_buf = nullptr;  // per-tile u16 record lists
    size_t slot_buf_bytes = 0;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_count_done = nullptr;  // two-lane mode: recorded after this lane's pass-1 kernel
    bool count_recorded = false;
    bool scan_recorded = false;          // ev_stop has been recorded at least once

    unsigned long long* shards() const { return reinterpret_cast<unsigned long long*>(ws + HDR_BYTES); }
    unsigned long long* result() const { return reinterpret_cast<unsigned long long*>(ws + HDR_BYTES + SHARD_BYTES); }
};

struct bl_ctx {
    int device = 0;
    Lane lanes[2];
    Lane* cur = nullptr;               // lane of the scan being issued / issued last
    int next_lane = 0;
    int n_lanes = 1;                   // 2: consecutive async scans alternate lanes, staggered (bl_ctx_set_lanes)
    // two-lane mode: LDS footprint pass-2 workgroups are padded to, which caps how many of them a CU holds beside the next scan's pass 1
    // (uncapped, the memory-bound record pass crowds the ALU-bound one out of the SIMDs: 214 Gbp/s at C3 in round 1).  0 = by mode
    // (emit_lds_default); bl_ctx_set_option("emit_lds_bytes") overrides.  Swept on MI355X with the round-3 kernels — minimizers on 150-bp reads (pass 1 holds
    // 12.4 KB per workgroup): 16 K -> 497, 22-26 K -> 517, 28 K -> 495, 40 K -> 460 Gbp/s; super-k-mers on 10-kbp reads (pass 1: 28.7 KB):
    // 16 K -> 400, 24 K -> 383.  Round 4 (four-tile record kernel for minimizers): see scan_windows.
    uint32_t emit_lds_per_wg = 0;
    hipStream_t user_stream = nullptr; // borrowed (bl_ctx_set_stream); NULL while borrowed = the legacy default stream
    bool borrowed = false;             // scans run on user_stream instead of the lanes' own streams
    hipStream_t stream = nullptr;      // stream of `cur`
    unsigned long long* pinned = nullptr;  // RING * RESULT_WORDS
    hipEvent_t slot_ev[RING] = {};         // recorded behind the copy into the slot
    int next_slot = 0;
    std::deque<Pending> pending;           // oldest first; at most RING entries
    bool timed = false;
    // optional per-launch timing of the main scan kernel alone (bl_ctx_kernel_timing)
    bool exact_windows = false;  // bl_ctx_set_exact_windows
    bool position_tiled = false; // bl_ctx_set_option("position_tiled"): never the read-tiled layout
    bool ktiming = false;
    std::vector<hipEvent_t> ev_pool;                       // free events
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_open;  // recorded, not yet read
    double kernel_ms = 0.0;
    uint64_t kernel_launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> marks;  // bl_ctx_mark: one event per lane stream (second: null with one stream)
    std::vector<bl_batch*> batches;  // live batches: destroyed with the context if the caller did not
    void* scratch[8] = {};  // grow-only device scratch of the non-scan entry points (bl_ctx_scratch)
    size_t scratch_bytes[8] = {};
    // device buffers of batches that came and went (bl_ctx_pool_*): a file is read as hundreds of batches of the same few
    // sizes, and every hipFree waits for the whole device — including the reader's inflate of the next span on its own stream
    std::vector<std::pair<void*, size_t>> pool_idle;
    std::vector<std::pair<void*, size_t>> pool_live;

    unsigned long long* shards() const { return cur->shards(); }
    unsigned long long* result() const { return cur->result(); }
    static size_t fixed_bytes() { return HDR_BYTES + SHARD_BYTES + RESULT_WORDS * sizeof(unsigned long long); }
};

struct bl_batch {
    bl_ctx* ctx = nullptr;
    uint8_t* bases = nullptr;
    bool owns_bases = false;
    uint32_t* start_bits = nullptr;  // nullptr: single sequence, or fixed-length reads whose vector has not been needed yet
    uint64_t n_bases = 0;
    uint64_t n_seqs = 0;
    uint64_t read_len = 0;           // != 0: every sequence has this length (the last one may be shorter)
    uint64_t origin = 0;             // position of base 0 in the caller's whole (bl_batch_set_origin): added to reported positions
};

namespace {

int grow(bl_ctx* c, void** buf, size_t* have, size_t need)
{
    if (need <= *have) return BL_OK;
    BL_HIP(hipStreamSynchronize(c->stream));  // the old buffer may still be in use by queued work
    if (*buf) BL_HIP(hipFree(*buf));
    *buf = nullptr;
    *have = 0;
    const size_t bytes = need + need / 8;
    hipError_t e = hipMalloc(buf, bytes);
    if (e != hipSuccess) return fail(BL_ERR_OOM, std::string("hipMalloc(scan scratch): ") + hipGetErrorString(e));
    *have = bytes;
    return BL_OK;
}

int ensure_workspace(bl_ctx* c)
{
    return grow(c, reinterpret_cast<void**>(&c->cur->ws), &c->cur->ws_bytes, bl_ctx::fixed_bytes());
}

// pick the lane (and stream) of the next scan
void select_lane(bl_ctx* c)
{
    if (c->borrowed) {
        c->cur = &c->lanes[0];
        c->stream = c->user_stream;
    } else {
        c->cur = &c->lanes[c->next_lane];
        if (c->n_lanes == 2) c->next_lane ^= 1;
        c->stream = c->cur->own;
    }
}

void deliver(bl_ctx* c, const Pending& p)
{
    const unsigned long long* r = c->pinned + (size_t)p.slot * RESULT_WORDS;
    bl_result out;
    std::memset(&out, 0, sizeof(out));
    out.count = r[0];
    out.xor_value = r[1];
    out.xor_hash = r[2];
    out.xor_pos = r[3];
    out.aux = r[4];
    out.redone = (int32_t)(r[5] > 0x7fffffffull ? 0x7fffffffull : r[5]);
    out.status = BL_OK;
    if (p.has_capacity && r[0] > p.capacity) out.status = BL_ERR_CAPACITY;
    if (p.user) *p.user = out;
}

int flush_pending(bl_ctx* c)
{
    for (const Pending& p : c->pending) deliver(c, p);
    c->pending.clear();
    return BL_OK;
}

int flush_kernel_events(bl_ctx* c)
{
    for (auto& pr : c->ev_open) {
        float ms = 0.f;
        BL_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
        c->kernel_ms += ms;
        c->kernel_launches += 1;
        c->ev_pool.push_back(pr.first);
        c->ev_pool.push_back(pr.second);
    }
    c->ev_open.clear();
    return BL_OK;
}

int sync_ctx(bl_ctx* c)
{
    for (Lane& l : c->lanes) BL_HIP(hipStreamSynchronize(l.own));
    if (c->borrowed) BL_HIP(hipStreamSynchronize(c->user_stream));
    int rc = flush_kernel_events(c);
    if (rc != BL_OK) return rc;
    return flush_pending(c);
}

// event pair around the main kernel of one scan (only when kernel timing is on)
int kernel_event(bl_ctx* c, bool start)
{
    if (!c->ktiming) return BL_OK;
    if (start) {
        hipEvent_t ev[2];
        for (int i = 0; i < 2; ++i) {
            if (!c->ev_pool.empty()) { ev[i] = c->ev_pool.back(); c->ev_pool.pop_back(); }
            else BL_HIP(hipEventCreate(&ev[i]));
        }
        c->ev_open.emplace_back(ev[0], ev[1]);
        BL_HIP(hipEventRecord(ev[0], c->stream));
    } else {
        BL_HIP(hipEventRecord(c->ev_open.back().second, c->stream));
    }
    return BL_OK;
}

// clear the digest shards, record the start event
int begin_scan(bl_ctx* c)
{
    BL_HIP(hipSetDevice(c->device));
    select_lane(c);
    int rc = ensure_workspace(c);
    if (rc != BL_OK) return rc;
    BL_HIP(hipEventRecord(c->cur->ev_start, c->stream));
    BL_HIP(hipMemsetAsync(c->cur->ws, 0, bl_ctx::fixed_bytes(), c->stream));
    return BL_OK;
}

// fold the shards, copy the result record to a pinned slot, register the user's bl_result
int end_scan(bl_ctx* c, uint32_t add_mask, bl_result* user, bool has_capacity, uint64_t capacity, uint32_t flags,
             bool already_folded = false)
{
    if (!already_folded) {
        hipError_t e = bl::launch_reduce_shards(c->shards(), c->result(), add_mask, reinterpret_cast<unsigned long long*>(c->cur->ws) + 1, c->stream);
        if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("reduce_shards: ") + hipGetErrorString(e));
    }
    if ((int)c->pending.size() >= RING) {
        // the ring is full: hand over the OLDEST result alone (its scan was issued RING scans ago and has normally
        // finished long since) — the streams keep running, nothing is drained
        const Pending old = c->pending.front();
        BL_HIP(hipEventSynchronize(c->slot_ev[old.slot]));
        deliver(c, old);
        c->pending.pop_front();
    }
    const int slot = c->next_slot;
    c->next_slot = (c->next_slot + 1) % RING;
    BL_HIP(hipMemcpyAsync(c->pinned + (size_t)slot * RESULT_WORDS, c->result(), RESULT_WORDS * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, c->stream));
    BL_HIP(hipEventRecord(c->slot_ev[slot], c->stream));
    BL_HIP(hipEventRecord(c->cur->ev_stop, c->stream));
    c->cur->scan_recorded = true;
    c->timed = true;
    c->pending.push_back(Pending{user, slot, capacity, has_capacity});
    if (flags & BL_FLAG_SYNC) {
        int rc = sync_ctx(c);
        if (rc != BL_OK) return rc;
        if (user && user->status != BL_OK)
            return fail(user->status, "output capacity too small for the records found");
    }
    return BL_OK;
}

int zero_result(bl_ctx* c, bl_result* user, uint32_t flags)
{
    if (user) std::memset(user, 0, sizeof(*user));
    (void)c;
    (void)flags;
    return BL_OK;
}

int check_range(const bl_batch* b, uint64_t first, uint64_t n, uint64_t& end)
{
    if (first > b->n_bases) return fail(BL_ERR_INVALID, "range starts beyond the batch");
    end = (n == 0 || n > b->n_bases - first) ? b->n_bases : first + n;  // first + n cannot wrap here
    if (end - first > (1ull << 31)) return fail(BL_ERR_INVALID, "a scan range may hold at most 2^31 positions; split it");
    return BL_OK;
}

int make_start_bits(bl_ctx* c, bl_batch* b, const uint64_t* offsets, uint64_t n_seqs, uint64_t read_len)
{
    const uint64_t n_words = (b->n_bases + 31) / 32 + 4;
    if (!offsets && (read_len == 0 || read_len >= b->n_bases)) {
        b->start_bits = nullptr;  // one sequence starting at 0
        b->n_seqs = b->n_bases ? 1 : 0;
        return BL_OK;
    }
    BL_HIP(hipMalloc(&b->start_bits, n_words * sizeof(uint32_t)));
    BL_HIP(hipMemsetAsync(b->start_bits, 0, n_words * sizeof(uint32_t), c->stream));
    if (offsets) {
        for (uint64_t q = 0; q < n_seqs; ++q)
            if (offsets[q] > offsets[q + 1]) return fail(BL_ERR_INVALID, "offsets must be non-decreasing");
        if (offsets[0] != 0 || offsets[n_seqs] != b->n_bases) return fail(BL_ERR_INVALID, "offsets must span [0, n_bases]");
        uint64_t* d_off = nullptr;
        BL_HIP(hipMalloc(&d_off, (n_seqs + 1) * sizeof(uint64_t)));
        hipError_t e = hipMemcpyAsync(d_off, offsets, (n_seqs + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = bl::launch_start_bits_offsets(b->start_bits, d_off, n_seqs, b->n_bases, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        (void)hipFree(d_off);
        if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("start_bits_offsets: ") + hipGetErrorString(e));
        b->n_seqs = n_seqs;
    } else {
        hipError_t e = bl::launch_start_bits_fixed(b->start_bits, n_words, b->n_bases, read_len, c->stream);
        if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("start_bits_fixed: ") + hipGetErrorString(e));
        b->n_seqs = (b->n_bases + read_len - 1) / read_len;
        BL_HIP(hipStreamSynchronize(c->stream));  // the batch may be scanned from either lane
    }
    return BL_OK;
}

// Fixed-length reads: where a read starts is arithmetic, and the read-tiled scans (bl_scan_frl.hpp) never look at the
// 1-bit-per-base vector (6.25 GB for a 50-Gbp batch).  It is only built when a position-tiled scan first needs it.
int ensure_start_bits(bl_ctx* c, const bl_batch* cb)
{
    bl_batch* b = const_cast<bl_batch*>(cb);  // the handle is the library's own object
    if (b->start_bits || b->read_len == 0 || b->read_len >= b->n_bases) return BL_OK;
    int rc = sync_ctx(c);  // no scan of this batch is in flight while its description changes
    if (rc != BL_OK) return rc;
    return make_start_bits(c, b, nullptr, 0, b->read_len);
}

}  // namespace

// The stream every NON-scan entry point (set operations, super-k-mer packing, the text parser ...) enqueues its work on:
// the borrowed stream, or lane 0's.  With two lanes a scan may still be running on lane 1's stream, so lane 0's stream is
// first made to wait for the end of the last scan issued on the other lane: callers may chain scan -> pack / sort / ...
// without bl_ctx_sync, as the header promises.
hipStream_t bl_ctx_stream(bl_ctx* c)
{
    if (c->borrowed) return c->user_stream;
    if (c->n_lanes == 2 && c->lanes[1].scan_recorded) (void)hipStreamWaitEvent(c->lanes[0].own, c->lanes[1].ev_stop, 0);
    return c->lanes[0].own;
}
int bl_batch_adopt_device(bl_ctx* ctx, void* d_bases, uint64_t n_bases, uint64_t* d_offsets, uint64_t n_seqs, uint64_t fixed_len, bl_batch** out);
int bl_ctx_device(bl_ctx* c) { return c->device; }

// Device scratch that lives with the context (slot 0..7), grown on demand and never shrunk: the set operations and the
// bucketed counter need gigabytes of temporary space per call, and hipMalloc / hipFree of that size costs more than their
// kernels.  The caller has synchronised its previous use (these entry points are synchronous).  nullptr on failure.
// Who uses what: 0-3 the k-mer counter and the text parser (which call the set operations while holding them), 4-6 the set
// operations themselves (4 data, 5 library workspace, 6 counters).
void* bl_ctx_scratch(bl_ctx* c, int slot, size_t bytes)
{
    if (!c || slot < 0 || slot >= 8) return nullptr;
    if (bytes <= c->scratch_bytes[slot]) return c->scratch[slot];
    (void)hipSetDevice(c->device);
    if (c->scratch[slot]) {
        (void)hipDeviceSynchronize();
        (void)hipFree(c->scratch[slot]);
        c->scratch[slot] = nullptr;
        c->scratch_bytes[slot] = 0;
    }
    const size_t want = bytes + bytes / 8;
    if (hipMalloc(&c->scratch[slot], want) != hipSuccess) {
        c->scratch[slot] = nullptr;
        return nullptr;
    }
    c->scratch_bytes[slot] = want;
    return c->scratch[slot];
}

// Device memory for short-lived batch buffers: taken from the idle list when something of about the right size is there,
// returned to it instead of freed (up to 16 buffers of at most 1 GiB each, 4 GiB together).  bl_ctx_pool_free of a pointer
// that did not come from bl_ctx_pool_alloc is a plain hipFree.  The caller has synchronised the buffer's last use.
void* bl_ctx_pool_alloc(bl_ctx* c, size_t bytes)
{
    if (!c) return nullptr;
    if (bytes < 256) bytes = 256;
    size_t best = c->pool_idle.size();
    for (size_t i = 0; i < c->pool_idle.size(); ++i) {
        const size_t have = c->pool_idle[i].second;
        if (have >= bytes && have <= 2 * bytes + 65536 && (best == c->pool_idle.size() || have < c->pool_idle[best].second)) best = i;
    }
    if (best < c->pool_idle.size()) {
        const auto entry = c->pool_idle[best];
        c->pool_idle[best] = c->pool_idle.back();
        c->pool_idle.pop_back();
        c->pool_live.push_back(entry);
        return entry.first;
    }
    void* p = nullptr;
    const size_t want = bytes + bytes / 8;  // (the next span is rarely exactly as long as this one)
    if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    c->pool_live.emplace_back(p, want);
    return p;
}

void bl_ctx_pool_free(bl_ctx* c, void* p)
{
    if (!p) return;
    if (c)
        for (size_t i = 0; i < c->pool_live.size(); ++i)
            if (c->pool_live[i].first == p) {
                const auto entry = c->pool_live[i];
                c->pool_live[i] = c->pool_live.back();
                c->pool_live.pop_back();
                size_t total = 0;
                for (const auto& e : c->pool_idle) total += e.second;
                if (c->pool_idle.size() < 16 && entry.second <= ((size_t)1 << 30) && total + entry.second <= ((size_t)4 << 30)) c->pool_idle.push_back(entry);
                else (void)hipFree(p);
                return;
            }
    (void)hipFree(p);
}

extern "C" {

const char* bl_last_error(void) { return g_err.c_str(); }
int bl_version(void) { return BL_VERSION; }

int bl_device_count(int* n)
{
    if (!n) return fail(BL_ERR_INVALID, "n is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return fail(BL_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *n = c;
    return BL_OK;
}

int bl_ctx_create(int device, bl_ctx** out)
{
    if (!out) return fail(BL_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) return fail(BL_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= n) return fail(BL_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    BL_HIP(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(BL_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 (MI355X) only");
    BL_HIP(hipSetDevice(device));
    bl_ctx* c = new (std::nothrow) bl_ctx();
    if (!c) return fail(BL_ERR_OOM, "host allocation failed");
    c->device = device;
    hipError_t e = hipSuccess;
    for (Lane& l : c->lanes) {
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&l.own, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreate(&l.ev_start);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&l.ev_count_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreate(&l.ev_stop);
    }
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&c->pinned), (size_t)RING * RESULT_WORDS * sizeof(unsigned long long), hipHostMallocDefault);
    for (int i = 0; i < RING && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->slot_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        bl_ctx_destroy(c);
        return fail(BL_ERR_HIP, std::string("context setup: ") + hipGetErrorString(e));
    }
    for (int i = 0; i < 2; ++i) {  // fixed workspace of both lanes
        c->cur = &c->lanes[i];
        c->stream = c->cur->own;
        int rc = ensure_workspace(c);
        if (rc != BL_OK) { bl_ctx_destroy(c); return rc; }
    }
    c->cur = &c->lanes[0];
    c->stream = c->lanes[0].own;
    *out = c;
    return BL_OK;
}

// 1 or 2 execution lanes for contexts that run on their own streams (see Lane above); takes effect from the next scan
int bl_ctx_set_lanes(bl_ctx* c, int n)
{
    if (!c || (n != 1 && n != 2)) return fail(BL_ERR_INVALID, "bl_ctx_set_lanes: need a context and n = 1 or 2");
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    c->n_lanes = n;
    c->next_lane = 0;
    c->lanes[0].count_recorded = c->lanes[1].count_recorded = false;
    return BL_OK;
}

int bl_ctx_destroy(bl_ctx* c)
{
    if (!c) return BL_OK;
    (void)hipSetDevice(c->device);
    for (Lane& l : c->lanes)
        if (l.own) (void)hipStreamSynchronize(l.own);
    if (c->borrowed) (void)hipStreamSynchronize(c->user_stream);
    while (!c->batches.empty()) bl_batch_destroy(c->batches.back());  // handles of leftover batches become invalid
    for (Lane& l : c->lanes) {
        if (l.ws) (void)hipFree(l.ws);
        if (l.tile_buf) (void)hipFree(l.tile_buf);
        if (l.slot_buf) (void)hipFree(l.slot_buf);
        if (l.ev_start) (void)hipEventDestroy(l.ev_start);
        if (l.ev_count_done) (void)hipEventDestroy(l.ev_count_done);
        if (l.ev_stop) (void)hipEventDestroy(l.ev_stop);
        if (l.own) (void)hipStreamDestroy(l.own);
    }
    for (void* p : c->scratch)
        if (p) (void)hipFree(p);
    for (const auto& e : c->pool_idle) (void)hipFree(e.first);
    for (const auto& e : c->pool_live) (void)hipFree(e.first);  // (none: every batch has been destroyed above)
    if (c->pinned) (void)hipHostFree(c->pinned);
    for (hipEvent_t ev : c->slot_ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto& pr : c->ev_open) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t ev : c->ev_pool) (void)hipEventDestroy(ev);
    for (auto& pr : c->marks) { (void)hipEventDestroy(pr.first); if (pr.second) (void)hipEventDestroy(pr.second); }
    delete c;
    return BL_OK;
}

int bl_ctx_set_stream(bl_ctx* c, void* hip_stream)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    c->borrowed = true;
    c->user_stream = static_cast<hipStream_t>(hip_stream);  // NULL is a stream too: the legacy default stream (torch's default)
    c->cur = &c->lanes[0];
    c->stream = c->user_stream;
    return BL_OK;
}

int bl_ctx_use_own_streams(bl_ctx* c)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    c->borrowed = false;
    c->user_stream = nullptr;
    c->cur = &c->lanes[0];
    c->stream = c->lanes[0].own;
    return BL_OK;
}

int bl_ctx_sync(bl_ctx* c)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    return sync_ctx(c);
}

int bl_ctx_last_scan_ms(bl_ctx* c, float* ms)
{
    if (!c || !ms) return fail(BL_ERR_INVALID, "NULL argument");
    if (!c->timed) return fail(BL_ERR_INVALID, "no scan has been issued on this context");
    BL_HIP(hipEventSynchronize(c->cur->ev_stop));
    BL_HIP(hipEventElapsedTime(ms, c->cur->ev_start, c->cur->ev_stop));
    return BL_OK;
}

int bl_ctx_set_exact_windows(bl_ctx* c, int on)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    c->exact_windows = on != 0;
    return BL_OK;
}

int bl_ctx_set_option(bl_ctx* c, const char* name, int64_t value)
{
    if (!c || !name) return fail(BL_ERR_INVALID, "bl_ctx_set_option: NULL argument");
    const std::string n(name);
    if (n == "exact_windows" && (value == 0 || value == 1)) return bl_ctx_set_exact_windows(c, (int)value);
    if (n == "lanes") return bl_ctx_set_lanes(c, (int)value);
    if (n == "position_tiled" && (value == 0 || value == 1)) {
        c->position_tiled = value != 0;
        return BL_OK;
    }
    if (n == "emit_lds_bytes" && value >= 0 && value <= 160 * 1024) {
        c->emit_lds_per_wg = (uint32_t)value;
        return BL_OK;
    }
    return fail(BL_ERR_INVALID, "bl_ctx_set_option: unknown name or value out of range: " + n);
}

int bl_ctx_kernel_timing(bl_ctx* c, int enable)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    c->ktiming = enable != 0;
    c->kernel_ms = 0.0;
    c->kernel_launches = 0;
    return BL_OK;
}

int bl_ctx_kernel_time(bl_ctx* c, double* total_ms, uint64_t* launches)
{
    if (!c || !total_ms || !launches) return fail(BL_ERR_INVALID, "NULL argument");
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    *total_ms = c->kernel_ms;
    *launches = c->kernel_launches;
    return BL_OK;
}

// Markers on the device's own timeline, without stopping it: where a caller wants the time at which "everything issued so far" had
// finished — a step of a benchmark — but must not synchronise there, because the next step's first kernels overlap this one's last.
// at most this many markers wait for bl_ctx_mark_times (two events each): a caller that never collects them gets an error, not a leak
static constexpr size_t MAX_MARKS = 4096;

int bl_ctx_mark(bl_ctx* c)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    if (c->marks.size() >= MAX_MARKS) return fail(BL_ERR_CAPACITY, "bl_ctx_mark: 4096 markers outstanding: collect them with bl_ctx_mark_times");
    BL_HIP(hipSetDevice(c->device));
    std::pair<hipEvent_t, hipEvent_t> m{nullptr, nullptr};
    const bool two = !c->borrowed && c->n_lanes == 2 && c->lanes[1].own;
    hipError_t e = hipEventCreate(&m.first);
    if (e == hipSuccess && two) e = hipEventCreate(&m.second);
    if (e == hipSuccess) e = hipEventRecord(m.first, c->borrowed ? c->user_stream : c->lanes[0].own);
    if (e == hipSuccess && two) e = hipEventRecord(m.second, c->lanes[1].own);
    if (e != hipSuccess) {  // nothing of a failed marker stays behind
        if (m.first) (void)hipEventDestroy(m.first);
        if (m.second) (void)hipEventDestroy(m.second);
        return fail(BL_ERR_HIP, std::string("bl_ctx_mark: ") + hipGetErrorString(e));
    }
    c->marks.push_back(m);
    return BL_OK;
}

int bl_ctx_mark_times(bl_ctx* c, double* ms, uint32_t capacity, uint32_t* n)
{
    if (!c || !n || (!ms && capacity)) return fail(BL_ERR_INVALID, "NULL argument");
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    *n = (uint32_t)c->marks.size();
    if (capacity < c->marks.size()) {  // (the markers stay: ask again with room for *n)
        if (capacity == 0) return BL_OK;
        return fail(BL_ERR_CAPACITY, "bl_ctx_mark_times: more markers than the buffer holds (*n says how many)");
    }
    hipError_t e = hipSuccess;
    for (size_t i = 0; i < c->marks.size(); ++i) {
        float a = 0.f, b = 0.f;  // since marker 0's event on the first stream: timestamps are the device's, whatever the stream
        if (e == hipSuccess) e = hipEventSynchronize(c->marks[i].first);
        if (e == hipSuccess) e = hipEventElapsedTime(&a, c->marks[0].first, c->marks[i].first);
        if (e == hipSuccess && c->marks[i].second) {
            e = hipEventSynchronize(c->marks[i].second);
            if (e == hipSuccess) e = hipEventElapsedTime(&b, c->marks[0].first, c->marks[i].second);
        }
        if (i < capacity) ms[i] = a > b ? a : b;
    }
    for (auto& pr : c->marks) { (void)hipEventDestroy(pr.first); if (pr.second) (void)hipEventDestroy(pr.second); }
    c->marks.clear();
    if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("bl_ctx_mark_times: ") + hipGetErrorString(e));
    return BL_OK;
}

// ---------------------------------------------------------------------------------------- batches

static int new_batch(bl_ctx* c, uint64_t n_bases, bl_batch** out, bl_batch*& b)
{
    if (!c || !out) return fail(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    BL_HIP(hipSetDevice(c->device));
    b = new (std::nothrow) bl_batch();
    if (!b) return fail(BL_ERR_OOM, "host allocation failed");
    b->ctx = c;
    b->n_bases = n_bases;
    c->batches.push_back(b);
    return BL_OK;
}

// the two upload entry points: sequences given by offsets, or reads of one length (read_len != 0, offsets ignored)
static int upload_batch(bl_ctx* c, const char* bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, uint64_t read_len, bl_batch** out)
{
    bl_batch* b = nullptr;
    int rc = new_batch(c, n_bases, out, b);
    if (rc != BL_OK) return rc;
    if (n_bases && !bases) { bl_batch_destroy(b); return fail(BL_ERR_INVALID, "bases is NULL"); }
    // from the context's pool of batch buffers: a stream of uploads reuses a few buffers instead of paying a device-wide wait per hipFree
    b->bases = static_cast<uint8_t*>(bl_ctx_pool_alloc(c, n_bases + 64));
    if (!b->bases) { bl_batch_destroy(b); return fail(BL_ERR_OOM, "device allocation of the batch's bases failed"); }
    b->owns_bases = true;
    hipError_t e = hipMemsetAsync(b->bases + (n_bases & ~15ull), 0, 64 + (n_bases & 15ull), c->stream);
    if (e == hipSuccess && n_bases) e = hipMemcpyAsync(b->bases, bases, n_bases, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { bl_batch_destroy(b); return fail(BL_ERR_HIP, std::string("upload: ") + hipGetErrorString(e)); }
    uint64_t fixed = 0;  // offsets that describe equal-length reads (the usual short-read batch) need no start-bit vector
    if (read_len) {
        if (read_len < n_bases) {
            fixed = read_len;
            n_seqs = (n_bases + read_len - 1) / read_len;
        } else {
            offsets = nullptr;  // one read
            n_seqs = 0;
        }
    } else if (offsets && n_seqs > 1 && offsets[0] == 0 && offsets[n_seqs] == n_bases && offsets[1] > 0 && offsets[1] < n_bases) {
        fixed = offsets[1];
        for (uint64_t q = 1; q < n_seqs && fixed; ++q)
            if (offsets[q] != q * fixed) fixed = 0;
        if (fixed && (n_bases - offsets[n_seqs - 1] > fixed || n_bases == offsets[n_seqs - 1])) fixed = 0;
    }
    if (fixed) {
        b->read_len = fixed;
        b->n_seqs = n_seqs;
    } else {
        rc = make_start_bits(c, b, offsets, n_seqs, 0);
    }
    if (rc == BL_OK) {
        e = hipStreamSynchronize(c->stream);  // the host buffer may go away after we return
        if (e != hipSuccess) rc = fail(BL_ERR_HIP, std::string("upload sync: ") + hipGetErrorString(e));
    }
    if (rc != BL_OK) { bl_batch_destroy(b); return rc; }
    *out = b;
    return BL_OK;
}

int bl_batch_upload(bl_ctx* c, const char* bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, bl_batch** out)
{
    return upload_batch(c, bases, n_bases, offsets, n_seqs, 0, out);
}

int bl_batch_upload_reads(bl_ctx* c, const char* bases, uint64_t n_bases, uint64_t read_len, bl_batch** out)
{
    if (read_len == 0) return fail(BL_ERR_INVALID, "read_len must not be 0");
    return upload_batch(c, bases, n_bases, nullptr, 0, read_len, out);
}

int bl_batch_from_device(bl_ctx* c, const void* d_bases, uint64_t n_bases, const uint64_t* offsets, uint64_t n_seqs, uint64_t read_len,
                         bl_batch** out)
{
    bl_batch* b = nullptr;
    int rc = new_batch(c, n_bases, out, b);
    if (rc != BL_OK) return rc;
    if ((n_bases && !d_bases) || (reinterpret_cast<uintptr_t>(d_bases) & 15)) {
        bl_batch_destroy(b);
        return fail(BL_ERR_INVALID, "d_bases must be a non-NULL, 16-byte aligned device pointer");
    }
    b->bases = const_cast<uint8_t*>(static_cast<const uint8_t*>(d_bases));
    b->owns_bases = false;
    if (!offsets && read_len > 0 && read_len < n_bases) {  // fixed-length reads: the start-bit vector is built on demand
        b->read_len = read_len;
        b->n_seqs = (n_bases + read_len - 1) / read_len;
    } else {
        rc = make_start_bits(c, b, offsets, n_seqs, read_len);
        if (rc != BL_OK) { bl_batch_destroy(b); return rc; }
    }
    *out = b;
    return BL_OK;
}

int bl_batch_set_origin(bl_batch* b, uint64_t origin)
{
    if (!b) return fail(BL_ERR_INVALID, "batch is NULL");
    b->origin = origin;
    return BL_OK;
}

uint64_t bl_batch_origin(const bl_batch* b) { return b ? b->origin : 0; }

int bl_batch_synth(bl_ctx* c, uint64_t seed, uint64_t n_bases, uint64_t read_len, bl_batch** out)
{
    bl_batch* b = nullptr;
    int rc = new_batch(c, n_bases, out, b);
    if (rc != BL_OK) return rc;
    hipError_t e = hipMalloc(&b->bases, n_bases + 64);
    if (e != hipSuccess) { bl_batch_destroy(b); return fail(BL_ERR_OOM, std::string("hipMalloc(bases): ") + hipGetErrorString(e)); }
    b->owns_bases = true;
    e = hipMemsetAsync(b->bases + (n_bases & ~15ull), 0, 64 + (n_bases & 15ull), c->stream);
    if (e == hipSuccess) e = bl::launch_synth(b->bases, 0, n_bases, seed, c->stream);
    if (e != hipSuccess) { bl_batch_destroy(b); return fail(BL_ERR_HIP, std::string("synth: ") + hipGetErrorString(e)); }
    if (read_len > 0 && read_len < n_bases) {
        b->read_len = read_len;
        b->n_seqs = (n_bases + read_len - 1) / read_len;
    } else {
        rc = make_start_bits(c, b, nullptr, 0, read_len);
    }
    if (rc == BL_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(BL_ERR_HIP, "synth sync failed");
    if (rc != BL_OK) { bl_batch_destroy(b); return rc; }
    *out = b;
    return BL_OK;
}

}  // extern "C"

// Used by the device-side text parser (bl_parse.hip): take ownership of a device base buffer (>= n_bases + 64 bytes,
// zero padded) and of DEVICE offsets[n_seqs + 1]; builds the start bits and frees the offsets.  fixed_len != 0: the parser
// found every sequence that long (the usual short-read file): the batch is a fixed-length one — read-tiled scans, no start bits.
int bl_batch_adopt_device(bl_ctx* c, void* d_bases, uint64_t n_bases, uint64_t* d_offsets, uint64_t n_seqs, uint64_t fixed_len, bl_batch** out)
{
    bl_batch* b = nullptr;
    int rc = new_batch(c, n_bases, out, b);
    if (rc != BL_OK) { bl_ctx_pool_free(c, d_bases); bl_ctx_pool_free(c, d_offsets); return rc; }
    b->bases = static_cast<uint8_t*>(d_bases);
    b->owns_bases = true;
    b->n_seqs = n_seqs;
    if (fixed_len && n_seqs > 1 && fixed_len < n_bases) {
        b->read_len = fixed_len;
        bl_ctx_pool_free(c, d_offsets);
        *out = b;
        return BL_OK;
    }
    hipStream_t s = bl_ctx_stream(c);
    const uint64_t n_words = (n_bases + 31) / 32 + 4;
    b->start_bits = static_cast<uint32_t*>(bl_ctx_pool_alloc(c, n_words * sizeof(uint32_t)));
    hipError_t e = b->start_bits ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess) e = hipMemsetAsync(b->start_bits, 0, n_words * sizeof(uint32_t), s);
    if (e == hipSuccess) e = bl::launch_start_bits_offsets(b->start_bits, d_offsets, n_seqs, n_bases, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    bl_ctx_pool_free(c, d_offsets);
    if (e != hipSuccess) { bl_batch_destroy(b); return fail(BL_ERR_HIP, std::string("adopt: ") + hipGetErrorString(e)); }
    *out = b;
    return BL_OK;
}

extern "C" {

int bl_batch_destroy(bl_batch* b)
{
    if (!b) return BL_OK;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        for (Lane& l : b->ctx->lanes)
            if (l.own) (void)hipStreamSynchronize(l.own);
        if (b->ctx->borrowed) (void)hipStreamSynchronize(b->ctx->user_stream);
        auto& v = b->ctx->batches;
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] == b) { v[i] = v.back(); v.pop_back(); break; }
    }
    if (b->owns_bases && b->bases) bl_ctx_pool_free(b->ctx, b->bases);
    if (b->start_bits) bl_ctx_pool_free(b->ctx, b->start_bits);
    delete b;
    return BL_OK;
}

uint64_t bl_batch_n_bases(const bl_batch* b) { return b ? b->n_bases : 0; }
uint64_t bl_batch_n_seqs(const bl_batch* b) { return b ? b->n_seqs : 0; }
const void* bl_batch_device_bases(const bl_batch* b) { return b ? b->bases : nullptr; }

int bl_batch_download(bl_batch* b, uint64_t first, uint64_t n, char* out)
{
    if (!b || (!out && n)) return fail(BL_ERR_INVALID, "NULL argument");
    if (first > b->n_bases || n > b->n_bases - first) return fail(BL_ERR_INVALID, "range beyond the batch");
    BL_HIP(hipSetDevice(b->ctx->device));
    int rc = sync_ctx(b->ctx);
    if (rc != BL_OK) return rc;
    if (n) BL_HIP(hipMemcpy(out, b->bases + first, n, hipMemcpyDeviceToHost));
    return BL_OK;
}

// ---------------------------------------------------------------------------------------- scans

int bl_scan_kmers(bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t k, uint64_t seed, uint32_t flags,
                  uint64_t* d_values, uint64_t* d_hashes, uint8_t* d_valid, bl_result* result)
{
    if (!c || !b || b->ctx != c) return fail(BL_ERR_INVALID, "ctx/batch is NULL or the batch belongs to another context");
    if (k < 1 || k > bl::MAX_UNIT) return fail(BL_ERR_INVALID, "k must be in [1, 32] (KmerType = uint64_t)");
    uint64_t end;
    int rc = check_range(b, first, n, end);
    if (rc != BL_OK) return rc;
    if (end <= first) return zero_result(c, result, flags);
    rc = ensure_start_bits(c, b);
    if (rc != BL_OK) return rc;
    bl::KmerParams p{};
    p.bases = b->bases;
    p.n_bases = (int64_t)b->n_bases;
    p.start_bits = b->start_bits;
    p.first = (int64_t)first;
    p.end = (int64_t)end;
    p.origin = bl::align_down16((int64_t)first);
    p.n_tiles = (int32_t)(((int64_t)end - 1 - p.origin) / bl::H + 1);
    p.unit = (int32_t)k;
    p.seed = (uint32_t)seed;  // hash.hpp:16,50: the seed is truncated to 32 bits
    p.canonical = (flags & BL_FLAG_CANONICAL) ? 1 : 0;
    p.drop_last = (flags & BL_FLAG_DROP_LAST) ? 1 : 0;
    p.out_value = d_values;
    p.out_hash = d_hashes;
    p.out_valid = d_valid;
    rc = begin_scan(c);
    if (rc != BL_OK) return rc;
    p.shards = c->shards();
    // 2,048 workgroups striding over the tiles (1.6 resident sets at five workgroups per CU).  One resident set exactly — 1,280 — measured
    // SLOWER on one lane (532 against 567 Gbp/s, A/B on one box) and the same on two (654): the 15 % a second lane gives this scan is not its tail.
    const int n_blocks = p.n_tiles < 256 * 8 ? p.n_tiles : 256 * 8;
    rc = kernel_event(c, true);
    if (rc != BL_OK) return rc;
    hipError_t e = bl::launch_kmers(p, n_blocks, c->stream);
    if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("kmer_kernel: ") + hipGetErrorString(e));
    rc = kernel_event(c, false);
    if (rc != BL_OK) return rc;
    return end_scan(c, /*add_mask: count, sum*/ (1u << 0) | (1u << 3), result, false, 0, flags);
}

static int scan_windows(int mode, bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t unit, uint32_t w, uint64_t seed,
                        uint32_t flags, bl::ScanParams& p, uint64_t capacity, bl_result* result)
{
    if (!c || !b || b->ctx != c) return fail(BL_ERR_INVALID, "ctx/batch is NULL or the batch belongs to another context");
    if (unit < 1 || unit > bl::MAX_UNIT) return fail(BL_ERR_INVALID, "hashed unit length must be in [1, 32]");
    if (w < 1 || w > bl::MAX_W) return fail(BL_ERR_INVALID, "window must be in [1, 64]");
    uint64_t end;
    int rc = check_range(b, first, n, end);
    if (rc != BL_OK) return rc;
    if (end <= first) return zero_result(c, result, flags);
    p.bases = b->bases;
    p.n_bases = (int64_t)b->n_bases;
    p.pos_base = (int64_t)b->origin;
    p.exact_windows = c->exact_windows ? 1 : 0;
    bl::plan_scan(mode, (int64_t)first, (int64_t)end, (int)w, p);
    // fixed-length short reads, range aligned to reads: the read-tiled layout (no start bits, no hashing of positions
    // that cannot start a unit) when it pays; bl_ctx_set_option("position_tiled", 1) keeps the position-tiled kernels (A/B measurements)
    if (b->read_len && !c->position_tiled && !p.use_threshold && bl::frl_width_built(mode, (int)w)) {
        bl::plan_scan_frl_for(mode, (int64_t)first, (int64_t)end, (int64_t)b->n_bases, (int64_t)b->read_len, (int)unit, (int)w, (flags & BL_FLAG_CANONICAL) != 0, p);
    }
    if (!p.frl) {
        rc = ensure_start_bits(c, b);
        if (rc != BL_OK) return rc;
    }
    p.start_bits = b->start_bits;
    p.unit = (int32_t)unit;
    p.w = (int32_t)w;
    p.seed = (uint32_t)seed;
    p.canonical = (flags & BL_FLAG_CANONICAL) ? 1 : 0;
    p.drop_last = (flags & BL_FLAG_DROP_LAST) ? 1 : 0;
    p.capacity = capacity;
    rc = begin_scan(c);
    if (rc != BL_OK) return rc;
    // scratch: tile counts + local prefixes + scan-block totals/prefixes, and the per-tile u16 list slots
    const size_t nt = (size_t)p.n_tiles, nb = (nt + bl::SCAN_BLK - 1) / bl::SCAN_BLK;
    rc = grow(c, reinterpret_cast<void**>(&c->cur->tile_buf), &c->cur->tile_buf_bytes, (2 * nt + 2 * nb + 8) * sizeof(unsigned long long) + nt * sizeof(uint32_t));
    if (rc != BL_OK) return rc;
    const size_t n_lists = mode == bl::MODE_SUPERKMER ? 3 : 1;
    const size_t slot_entries = nt * (size_t)p.stride;
    const size_t list_bytes = n_lists * slot_entries * sizeof(uint16_t);  // multiple of 32 bytes (stride % 16 == 0)
    rc = grow(c, reinterpret_cast<void**>(&c->cur->slot_buf), &c->cur->slot_buf_bytes, list_bytes + nt * (size_t)p.slot_chunks * sizeof(uint32_t));
    if (rc != BL_OK) return rc;
    unsigned long long* tb = reinterpret_cast<unsigned long long*>(c->cur->tile_buf);
    p.tile_counts = tb;
    p.tile_base = tb + nt;
    unsigned long long* block_tot = tb + 2 * nt;
    p.block_base = block_tot + nb;
    p.redo_list = reinterpret_cast<uint32_t*>(tb + 2 * nt + 2 * nb + 8);                     // tiles a closed-syncmer pass 1 could not decide
    p.redo_count = reinterpret_cast<unsigned long long*>(c->cur->ws) + 1;                     // header word 1, zeroed by begin_scan
    p.slots_a = c->cur->slot_buf;
    p.slots_j = mode == bl::MODE_SUPERKMER ? c->cur->slot_buf + slot_entries : nullptr;
    p.slots_e = mode == bl::MODE_SUPERKMER ? c->cur->slot_buf + 2 * slot_entries : nullptr;
    p.slots_c = reinterpret_cast<uint32_t*>(reinterpret_cast<unsigned char*>(c->cur->slot_buf) + list_bytes);
    p.shards = c->shards();
    // count -> tile prefix scan -> emit, all tiles in one group (the prefix scan supports several
    // groups with a running carry; one group is what is used)
    const bl::GroupRange all{0, (uint32_t)p.n_tiles};
    unsigned long long* carry = reinterpret_cast<unsigned long long*>(c->cur->ws);  // header word, zeroed by begin_scan
    const bool staggered = c->n_lanes == 2 && !c->borrowed;
    if (staggered) {
        // two lanes: this scan's pass 1 starts when the other lane's pass 1 has finished, so that it runs beside the
        // other lane's pass 2 (HBM-write bound) instead of beside its pass 1 (both ALU bound: nothing to gain)
        Lane* other = c->cur == &c->lanes[0] ? &c->lanes[1] : &c->lanes[0];
        if (other->count_recorded) BL_HIP(hipStreamWaitEvent(c->stream, other->ev_count_done, 0));
    }
    rc = kernel_event(c, true);
    if (rc != BL_OK) return rc;
    hipError_t e = bl::launch_scan_count(mode, p, all, c->stream);  // pass 1: the dominant kernel (timed alone)
    if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("scan_count_kernel: ") + hipGetErrorString(e));
    if (staggered) {
        BL_HIP(hipEventRecord(c->cur->ev_count_done, c->stream));
        c->cur->count_recorded = true;
    }
    rc = kernel_event(c, false);
    if (rc != BL_OK) return rc;
    e = bl::launch_tile_scan(p, all, block_tot, carry, c->stream);
    if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("tile_scan: ") + hipGetErrorString(e));
    // (minimizer scans: 28 K since the record kernel takes four tiles per workgroup and carries 10 KB of its own — three of its workgroups per
    // CU; never below round 3's form in three A/B sweeps of round 4, where 24 K was once 2.7 % below it: profiles/r04_ab_summary.txt)
    // (super-k-mer scans: 8 K since their pass 1 is compiled for five waves per SIMD and holds 5 x 28.7 KB of a CU's 160: 393 / 396 / 394 / 390 / 382 Gbp/s
    // at 4 / 8 / 12 / 16 / 24 K on one box, C4 at 24 Gbp)
    const uint32_t emit_lds_default = mode == bl::MODE_SUPERKMER ? 8192u : 28672u;
    e = bl::launch_scan_emit(mode, p, all, c->stream, staggered ? (c->emit_lds_per_wg ? c->emit_lds_per_wg : emit_lds_default) : 0);  // pass 2
    if (e != hipSuccess) return fail(BL_ERR_HIP, std::string("scan_emit_kernel: ") + hipGetErrorString(e));
    return BL_OK;
}

int bl_scan_minimizers(bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t unit, uint32_t w, uint64_t seed, uint32_t flags,
                       uint64_t* d_values, uint64_t* d_positions, uint64_t* d_hashes, uint64_t capacity, bl_result* result)
{
    bl::ScanParams p{};
    p.out_value = d_values;
    p.out_pos = d_positions;
    p.out_hash = d_hashes;
    const bool wants = d_values || d_positions || d_hashes;
    int rc = scan_windows(bl::MODE_MINIMIZER, c, b, first, n, unit, w, seed, flags, p, wants ? capacity : 0, result);
    if (rc != BL_OK || p.n_tiles == 0) return rc;
    return end_scan(c, 1u << 0, result, wants, capacity, flags);
}

int bl_scan_hash_sample(bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t k, uint64_t seed, uint64_t threshold, uint32_t flags,
                        uint64_t* d_values, uint64_t* d_positions, uint64_t* d_hashes, uint64_t capacity, bl_result* result)
{
    bl::ScanParams p{};
    p.out_value = d_values;
    p.out_pos = d_positions;
    p.out_hash = d_hashes;
    p.use_threshold = 1;
    p.hash_below = threshold;
    const bool wants = d_values || d_positions || d_hashes;
    int rc = scan_windows(bl::MODE_MINIMIZER, c, b, first, n, k, 1, seed, flags, p, wants ? capacity : 0, result);
    if (rc != BL_OK || p.n_tiles == 0) return rc;
    return end_scan(c, 1u << 0, result, wants, capacity, flags);
}

int bl_scan_super_kmers(bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t k, uint32_t m, uint64_t seed, uint32_t flags,
                        uint64_t* d_minimizers, uint64_t* d_first_pos, uint8_t* d_mm_pos, uint8_t* d_sizes, uint64_t* d_hashes,
                        uint64_t capacity, bl_result* result)
{
    if (m < 1 || k < m) return fail(BL_ERR_INVALID, "need 1 <= m <= k");
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    bl::ScanParams p{};
    p.out_value = d_minimizers;
    p.out_first = d_first_pos;
    p.out_mmpos = d_mm_pos;
    p.out_hash = d_hashes;
    const bool wants = d_minimizers || d_first_pos || d_mm_pos || d_sizes || d_hashes;
    p.out_size = d_sizes;  // from the group's own end event, inside the record pass
    int rc = scan_windows(bl::MODE_SUPERKMER, c, b, first, n, m, k - m + 1, seed, flags, p, wants ? capacity : 0, result);
    if (rc != BL_OK || p.n_tiles == 0) return rc;
    return end_scan(c, (1u << 0) | (1u << 4), result, wants, capacity, flags);
}

int bl_scan_super_kmer_records(bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t k, uint32_t m, uint64_t seed, uint32_t flags,
                               uint64_t* d_records, uint64_t* d_hashes, uint64_t capacity, bl_result* result)
{
    if (m < 1 || k < m || k > 32 || 2 * k - m > 59) return fail(BL_ERR_INVALID, "need 1 <= m <= k <= 32 and 2k - m <= 59 (bases per packed record)");
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    bl::ScanParams p{};
    p.out_records = d_records;
    p.out_hash = d_hashes;
    const bool wants = d_records || d_hashes;
    int rc = scan_windows(bl::MODE_SUPERKMER, c, b, first, n, m, k - m + 1, seed, flags, p, wants ? capacity : 0, result);
    if (rc != BL_OK || p.n_tiles == 0) return rc;
    return end_scan(c, (1u << 0) | (1u << 4), result, wants, capacity, flags);
}

int bl_scan_syncmers(bl_ctx* c, const bl_batch* b, uint64_t first, uint64_t n, uint32_t k, uint32_t s, uint32_t start_offset,
                     uint32_t end_offset, uint64_t seed, uint32_t flags, uint64_t* d_positions, uint64_t capacity, bl_result* result)
{
    if (s < 1 || k < s || k > bl::MAX_UNIT) return fail(BL_ERR_INVALID, "need 1 <= s <= k <= 32");
    bl::ScanParams p{};
    p.out_pos = d_positions;
    p.soff = (int32_t)start_offset;
    p.eoff = (int32_t)end_offset;
    int rc = scan_windows(bl::MODE_SYNCMER, c, b, first, n, s, k - s + 1, seed, flags, p, d_positions ? capacity : 0, result);
    if (rc != BL_OK || p.n_tiles == 0) return rc;
    return end_scan(c, 1u << 0, result, d_positions != nullptr, capacity, flags);
}

// ---------------------------------------------------------------------------------------- helpers

int bl_device_alloc(bl_ctx* c, uint64_t bytes, void** d_ptr)
{
    if (!c || !d_ptr) return fail(BL_ERR_INVALID, "NULL argument");
    BL_HIP(hipSetDevice(c->device));
    *d_ptr = nullptr;
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(BL_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e));
    return BL_OK;
}

int bl_device_free(bl_ctx* c, void* d_ptr)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    BL_HIP(hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    if (d_ptr) BL_HIP(hipFree(d_ptr));
    return BL_OK;
}

// page-locked host memory: copies to and from it are single DMA transfers and it is touched (mapped) once, at allocation
int bl_host_alloc(bl_ctx* c, uint64_t bytes, void** ptr)
{
    if (!c || !ptr) return fail(BL_ERR_INVALID, "NULL argument");
    BL_HIP(hipSetDevice(c->device));
    *ptr = nullptr;
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 16, hipHostMallocPortable);
    if (e != hipSuccess) return fail(BL_ERR_OOM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return BL_OK;
}

int bl_host_free(bl_ctx* c, void* ptr)
{
    if (!c) return fail(BL_ERR_INVALID, "ctx is NULL");
    BL_HIP(hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    if (ptr) BL_HIP(hipHostFree(ptr));
    return BL_OK;
}

int bl_copy_to_host(bl_ctx* c, void* dst, const void* d_src, uint64_t bytes)
{
    if (!c || (bytes && (!dst || !d_src))) return fail(BL_ERR_INVALID, "NULL argument");
    BL_HIP(hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    if (bytes) BL_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return BL_OK;
}

int bl_copy_to_device(bl_ctx* c, void* d_dst, const void* src, uint64_t bytes)
{
    if (!c || (bytes && (!d_dst || !src))) return fail(BL_ERR_INVALID, "NULL argument");
    BL_HIP(hipSetDevice(c->device));
    int rc = sync_ctx(c);
    if (rc != BL_OK) return rc;
    if (bytes) BL_HIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return BL_OK;
}

uint64_t bl_hash64_u64(uint64_t value, uint64_t seed) { return bl::murmur64(value, (uint32_t)seed); }

}  // extern "C"
