// bl_ingest.cpp — host side of FASTA / FASTQ ingest: file -> decompressed byte stream -> records
// or raw text spans -> device batches.  Written from the formats, in three layers:
//
//   1. ByteSource: the decompressed bytes of the file as a sequence of chunks, produced by background threads
//        plain file      one read-ahead thread
//        BGZF            (bgzip: gzip members of <= 64 KiB whose header says how long they are) — the members are cut out
//                        by one thread and inflated by a pool, results delivered in file order: inflate scales with cores
//        other gzip      a deflate stream has no entry points, but a block start can be FOUND and the text decoded from there with
//                        place holders for what matches copy from the unknown 32 KiB before: ParallelGzip (pieces: bl_pgzip.hpp)
//                        has a pool decode the file in parts and puts them together in file order; pipes and small files: one
//                        zlib thread running ahead of the parser.  Concatenated members are followed either way.
//   2. RecordParser: a line-oriented state machine over the chunks (memchr for line ends, no per-byte loop) that returns
//      what the reader biolib's tools use returns — the reference's tests/kseq.h:185-234 is the behaviour to match and
//      tests/test_ingest.py + tests/ingest_fuzz.py (the reference reader as judge) are the gate:
//        * a record starts at the next '>' or '@' byte, wherever it stands; the name ends at the first whitespace byte
//        * sequence lines are joined until a line begins with '>', '@' or '+'; empty lines vanish; one '\r' at the end of
//          what has been gathered so far is dropped after each line, unless that is all there is
//        * after a '+' line, quality lines are gathered until they are at least as long as the sequence; a different
//          total length (or no quality at all) makes the record malformed
//      RecordBatcher: batches of whole records (bl_reader_next_batch), parsed one batch ahead of the caller by a thread.
//   3. TextCutter: text spans for the device-side parser (bl_batch_from_text) — the decompressed text cut at record boundaries
//      found from the span's own last bytes, assembled ahead of the caller in a ring of (page-locked) buffers; plain files are
//      read straight into them.
//   4. The compressed path (PackedSpans, DeviceBgzf, next_batch_packed): a BGZF file goes to the device as it is — a thread reads
//      it and walks the member headers, bl_inflate.hip inflates a span's members on the GPU, the text is cut and parsed there.
//      bl_reader_open_shard gives each of several readers its own part of one file (parts meet at record starts that every
//      reader recognises by itself).
// Bases are passed through untouched (the scan's own table decides what is a break).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "../../include/biolib_amd.h"
#include "bl_crc32.hpp"
#include "bl_pgzip.hpp"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip
extern hipStream_t bl_ctx_stream(bl_ctx* ctx);
extern int bl_ctx_device(bl_ctx* ctx);
extern int bl_bgzf_inflate_on(hipStream_t s, const void* d_packed, uint64_t packed_bytes, const bl_bgzf_member* d_members, uint64_t n_members, void* d_text,
                              uint64_t text_bytes, uint32_t* d_status);  // bl_inflate.hip
extern int bl_parse_device_text(bl_ctx* ctx, const uint8_t* d_text, uint64_t n_bytes, char first_byte, const char* ends, uint64_t ends_n, bl_batch** out,
                                uint64_t* n_seqs, uint64_t* n_bases);  // bl_parse.hip

namespace {

constexpr size_t CHUNK_BYTES = 4u << 20;  // decompressed bytes per chunk (plain / stream gzip)
constexpr int BGZF_GROUP = 48;            // BGZF members inflated per job (~3 MiB of text)

struct Chunk {
    std::vector<unsigned char> bytes;
    bool ok = true;  // false: the stream is damaged from here on
};

// Byte vectors that have been used once and keep their storage: a fresh multi-megabyte vector costs a page fault per 4 KiB on
// its first use, which is more than filling it costs.
class SpareBuffers {
public:
    std::vector<unsigned char> take()
    {
        std::lock_guard<std::mutex> lk(m_);
        if (spare_.empty()) return {};
        std::vector<unsigned char> v = std::move(spare_.back());
        spare_.pop_back();
        v.clear();
        return v;
    }
    // the same with its old contents and size: a resize() to about that size then writes nothing
    std::vector<unsigned char> take_as_is()
    {
        std::lock_guard<std::mutex> lk(m_);
        if (spare_.empty()) return {};
        std::vector<unsigned char> v = std::move(spare_.back());
        spare_.pop_back();
        return v;
    }
    void give(std::vector<unsigned char>&& v)
    {
        if (v.capacity() == 0) return;
        std::lock_guard<std::mutex> lk(m_);
        if (spare_.size() < 64) spare_.push_back(std::move(v));
    }

private:
    std::mutex m_;
    std::vector<std::vector<unsigned char>> spare_;
};

// Results of jobs handed out in order, finished in any order, consumed in order.
class OrderedQueue {
public:
    explicit OrderedQueue(size_t depth) : depth_(depth) {}
    // producer: reserve the next slot (blocks while `depth` results are waiting); nullptr once the consumer has gone
    std::shared_ptr<Chunk> reserve()
    {
        std::unique_lock<std::mutex> lk(m_);
        space_.wait(lk, [&] { return slots_.size() < depth_ || abandoned_; });
        if (abandoned_) return nullptr;
        auto c = std::make_shared<Chunk>();
        slots_.push_back({c, false});
        return c;
    }
    void finish(const std::shared_ptr<Chunk>& c)
    {
        std::lock_guard<std::mutex> lk(m_);
        for (auto& s : slots_)
            if (s.chunk == c) s.done = true;
        ready_.notify_all();
    }
    void close()
    {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        ready_.notify_all();
    }
    // consumer: next chunk in order; false at the end of the stream
    bool pop(Chunk& out)
    {
        std::unique_lock<std::mutex> lk(m_);
        ready_.wait(lk, [&] { return abandoned_ || (!slots_.empty() && slots_.front().done) || (closed_ && slots_.empty()); });
        if (abandoned_ || slots_.empty()) return false;
        out = std::move(*slots_.front().chunk);
        slots_.pop_front();
        space_.notify_all();
        return true;
    }
    void abandon()
    {
        std::lock_guard<std::mutex> lk(m_);
        abandoned_ = true;
        space_.notify_all();
        ready_.notify_all();
    }
    bool abandoned()
    {
        std::lock_guard<std::mutex> lk(m_);
        return abandoned_;
    }

private:
    struct Slot {
        std::shared_ptr<Chunk> chunk;
        bool done;
    };
    std::mutex m_;
    std::condition_variable ready_, space_;
    std::deque<Slot> slots_;
    size_t depth_;
    bool closed_ = false, abandoned_ = false;
};

// A pool that runs jobs (compressed group -> chunk) on `n` threads.
class InflatePool {
public:
    struct Job {
        std::vector<unsigned char> packed;  // whole BGZF members, back to back
        std::shared_ptr<Chunk> out;
    };
    InflatePool(int n, OrderedQueue& q, SpareBuffers& spare) : q_(q), spare_(spare)
    {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~InflatePool()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void submit(Job&& j)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            jobs_.push_back(std::move(j));
        }
        cv_.notify_one();
    }

private:
    // one BGZF member: 12-byte fixed header, XLEN bytes of extra fields, raw deflate data, CRC32, ISIZE (RFC 1952 + SAM spec §4.1)
    static bool inflate_member(const unsigned char* p, size_t len, std::vector<unsigned char>& out)
    {
        if (len < 26) return false;
        const size_t xlen = p[10] | (p[11] << 8);
        if (12 + xlen + 8 > len) return false;
        const unsigned char* data = p + 12 + xlen;
        const size_t dlen = len - 12 - xlen - 8;
        const uint32_t isize = (uint32_t)p[len - 4] | ((uint32_t)p[len - 3] << 8) | ((uint32_t)p[len - 2] << 16) | ((uint32_t)p[len - 1] << 24);
        const uint32_t want_crc = (uint32_t)p[len - 8] | ((uint32_t)p[len - 7] << 8) | ((uint32_t)p[len - 6] << 16) | ((uint32_t)p[len - 5] << 24);
        if (isize > (1u << 16)) return false;
        const size_t at = out.size();
        out.resize(at + isize);
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) return false;
        z.next_in = const_cast<unsigned char*>(data);
        z.avail_in = (uInt)dlen;
        z.next_out = out.data() + at;
        z.avail_out = isize;
        const int rc = inflate(&z, Z_FINISH);
        const bool good = (rc == Z_STREAM_END) && z.total_out == isize;
        inflateEnd(&z);
        return good && bl_crc32(0, out.data() + at, isize) == want_crc;
    }
    void run()
    {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [&] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                j = std::move(jobs_.front());
                jobs_.pop_front();
            }
            size_t at = 0;
            while (at < j.packed.size()) {
                const size_t bsize = ((size_t)j.packed[at + 16] | ((size_t)j.packed[at + 17] << 8)) + 1;  // validated by the cutter
                if (!inflate_member(j.packed.data() + at, bsize, j.out->bytes)) {
                    j.out->ok = false;
                    break;
                }
                at += bsize;
            }
            spare_.give(std::move(j.packed));
            q_.finish(j.out);
        }
    }
    OrderedQueue& q_;
    SpareBuffers& spare_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// One gzip stream inflated by many threads (the pieces: bl_pgzip.hpp).  The file is cut into parts of `part_bytes`; a pool decodes
// each part from the first block it finds in it to the first block boundary in the next part, as symbols; this thread takes the
// parts in file order, checks that a part begins at the very bit the part before ended on, turns the symbols into text (pool
// again, by pieces that become the queue's chunks) and keeps the members' CRC-32.  Whatever lies between the end of one part
// and the start of the next that fits (blocks the finder does not look for: stored, fixed, final ones; a false find; anything
// damaged) is inflated here by zlib, block by block, which also gives zlib the verdict on every odd or broken stream.
class ParallelGzip {
public:
    ParallelGzip(int fd, uint64_t size, int threads, size_t part_bytes, OrderedQueue& q, SpareBuffers& spare)
        : fd_(fd), size_(size), threads_(threads), part_bytes_(part_bytes), q_(q), spare_(spare)
    {
    }
    ~ParallelGzip()
    {
        stop_workers();
        if (map_ && map_ != MAP_FAILED) munmap(const_cast<uint8_t*>(map_), size_);
    }
    // false: nothing has been delivered and the caller should inflate the stream the plain way
    bool run()
    {
        void* m = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
        if (m == MAP_FAILED) return false;
        map_ = static_cast<const uint8_t*>(m);
        (void)madvise(m, size_, MADV_SEQUENTIAL);
        const uint64_t data = blpg::skip_member_header(map_, size_, 0);
        if (data == blpg::NPOS) return false;
        pos_ = 8 * data;
        const uint64_t n_parts = (size_ + part_bytes_ - 1) / part_bytes_;
        const int ahead = threads_ + 2;
        for (int i = 0; i < ahead + 2; ++i) free_parts_.push_back(new PartSlot());
        for (int i = 0; i < threads_; ++i) workers_.emplace_back([this] { work(); });
        uint64_t issued = 0;
        std::deque<PartSlot*> flying;
        bool done = false;
        for (uint64_t i = 0; i < n_parts && !done && !gone_; ++i) {
            while (issued < n_parts && issued < i + (uint64_t)ahead) {
                PartSlot* s = take_part(issued == i);  // (must have part i; the others only if a slot is free)
                if (!s) break;
                issue(s, issued++);
                flying.push_back(s);
            }
            PartSlot* s = flying.front();
            flying.pop_front();
            {
                const uint64_t t0 = now_us();
                wait_decoded(s);
                us_wait_part_ += now_us() - t0;
            }
            std::shared_ptr<PartSlot> ref(s, [this](PartSlot* x) { give_part(x); });
            if (!s->found || s->part.start_bit < pos_) {  // nothing found in this part, or the stream is past it already
                ++(s->found ? n_missed_ : n_none_);
                continue;
            }
            if (s->part.start_bit > pos_ && !by_zlib(s->part.start_bit, done)) break;
            if (done || failed_ || gone_) break;
            if (s->part.start_bit != pos_) {  // zlib's block boundaries stepped over the find: it was a false one
                ++n_missed_;
                continue;
            }
            if (!accept(ref)) break;
            ++n_taken_;
            pool_bytes_ += s->part.n;
            pos_ = s->part.end_bit;
            done = s->part.at_eof;
            settle(false);
        }
        cancel_ = true;
        for (PartSlot* s : flying) {
            wait_decoded(s);
            give_part(s);
        }
        if (!done && !failed_ && !gone_) by_zlib(blpg::NPOS, done);
        if (!gone_) {
            settle(true);
            if (failed_ || !done) {
                auto c = q_.reserve();
                if (c) {
                    c->ok = false;
                    q_.finish(c);
                }
            }
        }
        stop_workers();
        if (std::getenv("BL_INGEST_TRACE"))
            std::fprintf(stderr, "[pgzip] %llu parts of %zu bytes: %llu taken as found, %llu without a find, %llu not reached; text by the pool %llu bytes, by zlib %llu bytes\n",
                         (unsigned long long)n_parts, part_bytes_, (unsigned long long)n_taken_, (unsigned long long)n_none_, (unsigned long long)n_missed_,
                         (unsigned long long)pool_bytes_, (unsigned long long)zlib_bytes_);
        if (std::getenv("BL_INGEST_TRACE"))
            std::fprintf(stderr, "[pgzip] pool: find %.3f s, decode %.3f s, text + crc %.3f s; this thread waited %.3f s for parts, %.3f s for room in the queue\n",
                         us_find_ * 1e-6, us_decode_ * 1e-6, us_text_ * 1e-6, us_wait_part_ * 1e-6, us_wait_queue_ * 1e-6);
        return true;
    }

private:
    uint64_t n_taken_ = 0, n_none_ = 0, n_missed_ = 0, pool_bytes_ = 0, zlib_bytes_ = 0;  // (trace)
    std::atomic<uint64_t> us_find_{0}, us_decode_{0}, us_text_{0}, us_wait_part_{0}, us_wait_queue_{0};
    static uint64_t now_us() { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    struct PartSlot {
        blpg::Part part;
        bool found = false;
        bool decoded = false;
    };
    struct Piece {
        int state = 0;  // 0: being made, 1: good, 2: bad (guarded by m_)
        uint32_t crc = 0;
        uint64_t len = 0;
        bool is_end = false;  // not a piece: the end of a member, whose CRC-32 and length are due
        uint32_t want_crc = 0, want_isize = 0;
    };

    int fd_;
    uint64_t size_;
    int threads_;
    size_t part_bytes_;
    OrderedQueue& q_;
    SpareBuffers& spare_;
    const uint8_t* map_ = nullptr;
    uint64_t pos_ = 0;                  // the bit where the next block of the stream begins
    uint8_t window_[blpg::WINDOW];      // the last 32 KiB of text made so far
    uint32_t known_ = 0;                // how many of them exist
    bool failed_ = false, gone_ = false;  // damaged stream; the consumer has left
    std::atomic<bool> cancel_{false};

    std::mutex m_;
    std::condition_variable cv_jobs_, cv_done_;
    std::deque<std::function<void()>> urgent_, normal_;
    std::vector<std::thread> workers_;
    bool stopping_ = false;
    std::vector<PartSlot*> free_parts_;
    std::deque<std::shared_ptr<Piece>> ledger_;
    uint32_t run_crc_ = 0;
    uint64_t run_len_ = 0;

    void work()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_jobs_.wait(lk, [&] { return stopping_ || !urgent_.empty() || !normal_.empty(); });
                if (!urgent_.empty()) {
                    job = std::move(urgent_.front());
                    urgent_.pop_front();
                } else if (!normal_.empty()) {
                    job = std::move(normal_.front());
                    normal_.pop_front();
                } else {
                    return;
                }
            }
            job();
        }
    }
    void stop_workers()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            stopping_ = true;
        }
        cv_jobs_.notify_all();
        for (auto& t : workers_) t.join();
        workers_.clear();
        for (PartSlot* s : free_parts_) delete s;
        free_parts_.clear();
    }
    PartSlot* take_part(bool wait)
    {
        std::unique_lock<std::mutex> lk(m_);
        if (wait) cv_done_.wait(lk, [&] { return !free_parts_.empty(); });
        if (free_parts_.empty()) return nullptr;
        PartSlot* s = free_parts_.back();
        free_parts_.pop_back();
        return s;
    }
    void give_part(PartSlot* s)
    {
        s->part.sym.shrink_to(8ull * part_bytes_);  // (text is usually 3-6 symbols per compressed byte)
        {
            std::lock_guard<std::mutex> lk(m_);
            free_parts_.push_back(s);
        }
        cv_done_.notify_all();
    }
    void issue(PartSlot* s, uint64_t index)
    {
        s->found = false;
        s->decoded = false;
        const uint64_t first = index == 0 ? pos_ : 8 * index * (uint64_t)part_bytes_;
        const uint64_t limit = 8 * (index + 1) * (uint64_t)part_bytes_;
        auto job = [this, s, index, first, limit] {
            static thread_local blpg::SymbolDecoder decoder;
            // a part ends after 24 symbols per compressed byte of it at the latest (never less than 4 M, never more than 48 M): what
            // follows is found by the next part's search or, failing that, decoded by zlib — bounded memory, not unbounded buffers
            const uint64_t MAX_SYMBOLS = std::min<uint64_t>(48ull << 20, std::max<uint64_t>(4ull << 20, 24ull * part_bytes_));
            if (!cancel_) {
                uint64_t t0 = now_us();
                uint64_t at = index == 0 ? first : blpg::find_block(map_, size_, first, limit);
                us_find_ += now_us() - t0;
                for (int tries = 0; at != blpg::NPOS && tries < 16 && !cancel_; ++tries) {
                    t0 = now_us();
                    const bool good = decoder.run(map_, size_, at, limit, MAX_SYMBOLS, s->part);
                    us_decode_ += now_us() - t0;
                    if (good) {
                        s->found = true;
                        break;
                    }
                    at = index == 0 ? blpg::NPOS : blpg::find_block(map_, size_, at + 1, limit);
                }
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                s->decoded = true;
            }
            cv_done_.notify_all();
        };
        {
            std::lock_guard<std::mutex> lk(m_);
            normal_.push_back(std::move(job));
        }
        cv_jobs_.notify_one();
    }
    void wait_decoded(PartSlot* s)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_done_.wait(lk, [&] { return s->decoded; });
    }
    void push_window(const uint8_t* text, uint64_t n)
    {
        if (n >= blpg::WINDOW) {
            std::memcpy(window_, text + (n - blpg::WINDOW), blpg::WINDOW);
        } else {
            std::memmove(window_, window_ + n, blpg::WINDOW - n);
            std::memcpy(window_ + (blpg::WINDOW - n), text, n);
        }
        known_ = known_ + n >= blpg::WINDOW ? blpg::WINDOW : (uint32_t)(known_ + n);
    }
    void note_member_end(uint32_t crc, uint32_t isize)
    {
        auto e = std::make_shared<Piece>();
        e->is_end = true;
        e->state = 1;
        e->want_crc = crc;
        e->want_isize = isize;
        std::lock_guard<std::mutex> lk(m_);
        ledger_.push_back(e);
    }
    // fold the finished pieces at the front of the ledger into the running CRC; check it where a member ends
    void settle(bool wait_all)
    {
        std::unique_lock<std::mutex> lk(m_);
        while (!ledger_.empty()) {
            auto e = ledger_.front();
            if (e->state == 0) {
                if (!wait_all) return;
                cv_done_.wait(lk, [&] { return e->state != 0; });
            }
            ledger_.pop_front();
            if (e->state == 2) failed_ = true;
            if (e->is_end) {
                if (run_crc_ != e->want_crc || (uint32_t)run_len_ != e->want_isize) failed_ = true;
                run_crc_ = 0;
                run_len_ = 0;
            } else if (e->len) {
                run_crc_ = (uint32_t)crc32_combine(run_crc_, e->crc, (z_off_t)e->len);
                run_len_ += e->len;
            }
        }
    }
    // the symbols of a part that begins where the stream stands -> chunks of text, made by the pool
    bool accept(const std::shared_ptr<PartSlot>& ref)
    {
        const blpg::Part& part = ref->part;
        auto table = std::make_shared<blpg::SymbolTable>();
        table->set(window_);
        if (known_ < blpg::WINDOW && !blpg::markers_known(part.sym.p, part.n, known_)) {  // (the first 32 KiB of the stream only)
            failed_ = true;
            return false;
        }
        uint64_t at = 0;
        size_t next_end = 0;
        while (at < part.n || next_end < part.ends.size()) {
            while (next_end < part.ends.size() && part.ends[next_end].out_off == at) {
                note_member_end(part.ends[next_end].crc, part.ends[next_end].isize);
                ++next_end;
            }
            if (at >= part.n) break;
            uint64_t stop = at + CHUNK_BYTES < part.n ? at + CHUNK_BYTES : part.n;
            if (next_end < part.ends.size() && part.ends[next_end].out_off < stop) stop = part.ends[next_end].out_off;
            const uint64_t t0 = now_us();
            auto c = q_.reserve();
            us_wait_queue_ += now_us() - t0;
            if (!c) {
                gone_ = true;
                return false;
            }
            auto piece = std::make_shared<Piece>();
            piece->len = stop - at;
            const uint64_t from = at;
            auto job = [this, ref, table, c, piece, from] {
                const uint64_t t0 = now_us();
                c->bytes = spare_.take_as_is();
                c->bytes.resize(piece->len);  // (a new buffer is cleared by this: better here than on the one thread that orders the parts)
                blpg::resolve(ref->part.sym.p + from, piece->len, *table, c->bytes.data());
                const uint32_t crc = bl_crc32(0, c->bytes.data(), piece->len);
                us_text_ += now_us() - t0;
                q_.finish(c);
                {
                    std::lock_guard<std::mutex> lk(m_);
                    piece->crc = crc;
                    piece->state = 1;
                }
                cv_done_.notify_all();
            };
            {
                std::lock_guard<std::mutex> lk(m_);
                ledger_.push_back(piece);
                urgent_.push_back(std::move(job));
            }
            cv_jobs_.notify_one();
            at = stop;
        }
        // the window behind this part: its last 32 KiB as text
        if (part.n >= blpg::WINDOW) {
            uint8_t tail[blpg::WINDOW];
            blpg::resolve(part.sym.p + (part.n - blpg::WINDOW), blpg::WINDOW, *table, tail);
            push_window(tail, blpg::WINDOW);
        } else if (part.n) {
            std::vector<uint8_t> text(part.n);
            blpg::resolve(part.sym.p, part.n, *table, text.data());
            push_window(text.data(), part.n);
        }
        return true;
    }
    // zlib from pos_ on, block by block, until a block begins at or beyond `target` (or the stream ends: done = true).
    // false: the stream is damaged (failed_) or the consumer has left (gone_).
    bool by_zlib(uint64_t target, bool& done)
    {
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        if (inflateInit2(&z, -15) != Z_OK) {
            failed_ = true;
            return false;
        }
        uint64_t in_at = pos_ >> 3;
        if (pos_ & 7) {
            inflatePrime(&z, 8 - (int)(pos_ & 7), map_[in_at] >> (pos_ & 7));
            ++in_at;
        }
        if (known_) inflateSetDictionary(&z, window_ + (blpg::WINDOW - known_), known_);
        std::shared_ptr<Chunk> c;
        auto flush = [&]() {  // what has been written to the open chunk goes out
            if (!c) return;
            const size_t n = CHUNK_BYTES - z.avail_out;
            c->bytes.resize(n);
            auto piece = std::make_shared<Piece>();
            piece->len = n;
            piece->crc = bl_crc32(0, c->bytes.data(), n);
            piece->state = 1;
            zlib_bytes_ += n;
            push_window(c->bytes.data(), n);
            {
                std::lock_guard<std::mutex> lk(m_);
                ledger_.push_back(piece);
            }
            q_.finish(c);
            c.reset();
        };
        bool good = true;
        for (;;) {
            if (!c) {
                c = q_.reserve();
                if (!c) {
                    gone_ = true;
                    good = false;
                    break;
                }
                c->bytes = spare_.take();
                c->bytes.resize(CHUNK_BYTES);
                z.next_out = c->bytes.data();
                z.avail_out = (uInt)CHUNK_BYTES;
            }
            if (z.avail_in == 0) {
                const uint64_t left = size_ - in_at;
                const uint64_t n = left < (1ull << 30) ? left : (1ull << 30);
                z.next_in = const_cast<uint8_t*>(map_ + in_at);
                z.avail_in = (uInt)n;
            }
            const uint8_t* before = z.next_in;
            const int rc = inflate(&z, Z_BLOCK);
            in_at += (uint64_t)(z.next_in - before);
            if (rc == Z_STREAM_END) {  // a member's last block: trailer, then another member or the end of the file
                flush();
                if (in_at + 8 > size_) {
                    failed_ = true;
                    good = false;
                    break;
                }
                auto le32 = [&](uint64_t q) { return (uint32_t)map_[q] | ((uint32_t)map_[q + 1] << 8) | ((uint32_t)map_[q + 2] << 16) | ((uint32_t)map_[q + 3] << 24); };
                note_member_end(le32(in_at), le32(in_at + 4));
                in_at += 8;
                if (in_at == size_) {
                    pos_ = 8 * size_;
                    done = true;
                    break;
                }
                // what follows a complete member and does not open with the gzip magic is trailing garbage (tape / block padding):
                // the end of the stream, as for gzread, which the reference's drivers read through (tests/kseq.h over gzFile)
                if (in_at + 2 > size_ || map_[in_at] != 0x1f || map_[in_at + 1] != 0x8b) {
                    pos_ = 8 * size_;
                    done = true;
                    break;
                }
                const uint64_t data = blpg::skip_member_header(map_, size_, in_at);
                if (data == blpg::NPOS || inflateReset(&z) != Z_OK) {
                    failed_ = true;
                    good = false;
                    break;
                }
                in_at = data;
                pos_ = 8 * data;
                z.avail_in = 0;
                if (pos_ >= target) break;
                continue;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) {
                failed_ = true;
                good = false;
                break;
            }
            if ((z.data_type & 128) && !(z.data_type & 64)) {  // stopped between two blocks
                pos_ = 8 * in_at - (uint64_t)(z.data_type & 7);
                if (pos_ >= target) break;
            }
            if (z.avail_out == 0) flush();
            if (rc == Z_BUF_ERROR && z.avail_in == 0 && in_at == size_) {  // the file ends inside a member
                failed_ = true;
                good = false;
                break;
            }
        }
        if (c) {
            if (good || CHUNK_BYTES - z.avail_out) {
                flush();
            } else {
                c->bytes.clear();
                q_.finish(c);
            }
        }
        inflateEnd(&z);
        return good;
    }
};

// The decompressed bytes of one file.
class ByteSource {
public:
    ByteSource(FILE* f, int threads) : f_(f), queue_(16), threads_(threads < 1 ? 1 : threads)
    {
        unsigned char head[18];
        const size_t got = std::fread(head, 1, sizeof(head), f_);
        std::rewind(f_);
        const bool gz = got >= 2 && head[0] == 0x1f && head[1] == 0x8b;
        const bool bgzf = gz && got == 18 && head[2] == 8 && (head[3] & 4) && head[12] == 'B' && head[13] == 'C' && head[14] == 2 && head[15] == 0;
        kind_ = bgzf ? "bgzf" : gz ? "gzip" : "plain";
    }
    // the threads start with the first request for bytes: a BGZF file that goes to the device compressed never needs them
    void start()
    {
        if (started_) return;
        started_ = true;
        if (kind_[0] == 'b') {
            pool_.reset(new InflatePool(threads_, queue_, spare_));
            feeder_ = std::thread([this] { cut_bgzf(); });
        } else if (kind_[0] == 'g') {
            feeder_ = std::thread([this] { inflate_stream(); });
        } else {
            feeder_ = std::thread([this] { read_plain(); });
        }
    }
    ~ByteSource()
    {
        queue_.abandon();
        if (feeder_.joinable()) feeder_.join();
        pool_.reset();  // joins the workers
        if (f_) std::fclose(f_);
    }
    bool next(Chunk& c)
    {
        start();
        return queue_.pop(c);
    }
    void recycle(std::vector<unsigned char>&& v) { spare_.give(std::move(v)); }
    void abandon() { queue_.abandon(); }  // the consumer is leaving: next() returns false from now on
    // a plain file whose chunk threads have not been started can be read straight into the caller's buffer (the text spans
    // do: one copy less on the path of an uncompressed file).  -1 on a read error.
    bool can_read_direct() const { return kind_[0] == 'p' && !started_; }
    // a plain file of which only the bytes [first, end) are this reader's (bl_reader_open_shard); before the first read
    void set_range(uint64_t first, uint64_t end)
    {
        range_first_ = first;
        range_end_ = end;
    }
    long read_direct(void* dst, size_t want)
    {
        // large requests are cut in four and read side by side (pread at explicit offsets): one thread copies out of the page
        // cache at ~10 GB/s, which is less than the H2D link takes
        if (direct_off_ == (uint64_t)-1) {
            struct stat st;
            direct_regular_ = fstat(fileno(f_), &st) == 0 && S_ISREG(st.st_mode);  // anything else: plain fread below
            direct_size_ = !direct_regular_ ? 0 : ((uint64_t)st.st_size < range_end_ ? (uint64_t)st.st_size : range_end_);
            direct_off_ = range_first_;
        }
        if (!direct_regular_ || want < ((size_t)8 << 20)) {
            if (direct_regular_) {
                const uint64_t left = direct_off_ < direct_size_ ? direct_size_ - direct_off_ : 0;
                if (want > left) want = (size_t)left;
                if (want == 0) return 0;
                if (std::fseek(f_, (long)direct_off_, SEEK_SET) != 0) return -1;
            }
            const size_t n = std::fread(dst, 1, want, f_);
            if (n < want && std::ferror(f_)) return -1;
            direct_off_ += n;
            return (long)n;
        }
        const uint64_t left = direct_off_ < direct_size_ ? direct_size_ - direct_off_ : 0;
        const size_t n = want < left ? want : (size_t)left;
        constexpr int PARTS = 4;  // (eight measured no faster)
        const size_t part = (n + PARTS - 1) / PARTS;
        bool ok[PARTS];
        std::thread helpers[PARTS];
        const int fd = fileno(f_);
        auto work = [&](int i) {
            size_t a = (size_t)i * part, b = a + part < n ? a + part : n;
            ok[i] = true;
            while (a < b) {
                const ssize_t got = pread(fd, static_cast<char*>(dst) + a, b - a, (off_t)(direct_off_ + a));
                if (got <= 0) { ok[i] = false; break; }  // (the file shrank under us, or an I/O error)
                a += (size_t)got;
            }
        };
        for (int i = 1; i < PARTS; ++i) helpers[i] = std::thread(work, i);
        work(0);
        bool all = ok[0];
        for (int i = 1; i < PARTS; ++i) {
            helpers[i].join();
            all = all && ok[i];
        }
        if (!all) return -1;
        direct_off_ += n;
        return (long)n;
    }
    const char* kind() const { return kind_; }

private:
    void read_plain()
    {
        uint64_t left = ~0ULL;  // (a reader of one part of the file: its bytes only)
        if (range_first_ || range_end_ != ~0ULL) {
            left = range_end_ > range_first_ ? range_end_ - range_first_ : 0;
            if (std::fseek(f_, (long)range_first_, SEEK_SET) != 0) left = 0;
        }
        for (;;) {
            auto c = queue_.reserve();
            if (!c) return;
            c->bytes = spare_.take();
            c->bytes.resize(CHUNK_BYTES);
            const size_t want = left < CHUNK_BYTES ? (size_t)left : CHUNK_BYTES;
            const size_t n = want ? std::fread(c->bytes.data(), 1, want, f_) : 0;
            c->bytes.resize(n);
            if (n < want && std::ferror(f_)) c->ok = false;
            left -= n;
            const bool last = n < CHUNK_BYTES;
            queue_.finish(c);
            if (last) break;
        }
        queue_.close();
    }
    // A regular gzip file of some size goes to the many-threaded decoder; BL_PGZIP=0 keeps it off, BL_PGZIP_PART (bytes) sets
    // the size of the parts (tests: small parts make many seams).
    bool inflate_parallel()
    {
        const char* off = std::getenv("BL_PGZIP");
        if (threads_ < 2 || (off && off[0] == '0')) return false;
        struct stat st;
        if (fstat(fileno(f_), &st) != 0 || !S_ISREG(st.st_mode)) return false;
        size_t part = (size_t)2 << 20;
        if (const char* e = std::getenv("BL_PGZIP_PART")) {
            const long v = std::atol(e);
            if (v >= 1024) part = (size_t)v;
        }
        if ((uint64_t)st.st_size < 2 * (uint64_t)part) return false;
        ParallelGzip pg(fileno(f_), (uint64_t)st.st_size, threads_, part, queue_, spare_);
        return pg.run();
    }
    void inflate_stream()
    {
        if (inflate_parallel()) {
            queue_.close();
            return;
        }
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        bool ok = inflateInit2(&z, 15 + 32) == Z_OK;  // gzip or zlib wrapper, detected
        std::vector<unsigned char> in(1u << 20);
        bool input_done = false, member_open = false, at_boundary = false;  // at_boundary: a member has just ended
        while (ok) {
            auto c = queue_.reserve();
            if (!c) break;
            c->bytes = spare_.take();
            c->bytes.resize(CHUNK_BYTES);
            z.next_out = c->bytes.data();
            z.avail_out = (uInt)CHUNK_BYTES;
            bool finished = false;
            while (z.avail_out > 0) {
                if (z.avail_in == 0 && !input_done) {
                    const size_t n = std::fread(in.data(), 1, in.size(), f_);
                    if (n < in.size()) input_done = true;
                    z.next_in = in.data();
                    z.avail_in = (uInt)n;
                }
                if (at_boundary && z.avail_in == 1 && !input_done) {  // the magic may straddle two reads: keep the byte, read on
                    in[0] = *z.next_in;
                    const size_t n = std::fread(in.data() + 1, 1, in.size() - 1, f_);
                    if (n < in.size() - 1) input_done = true;
                    z.next_in = in.data();
                    z.avail_in = (uInt)(n + 1);
                }
                if (z.avail_in == 0 && input_done) {  // no more input: fine between members, a truncation inside one
                    if (member_open) c->ok = false;
                    finished = true;
                    break;
                }
                if (at_boundary) {
                    // behind a complete member: another member, or trailing garbage, which ends the stream as it does for
                    // gzread (the reference's drivers read through it: tests/kseq.h over gzFile)
                    if (z.avail_in < 2 || z.next_in[0] != 0x1f || z.next_in[1] != 0x8b) {
                        finished = true;
                        break;
                    }
                    at_boundary = false;
                }
                member_open = true;
                const int rc = inflate(&z, Z_NO_FLUSH);
                if (rc == Z_STREAM_END) {  // end of a member: another may follow (concatenated gzip)
                    member_open = false;
                    at_boundary = true;
                    if (inflateReset(&z) != Z_OK) { c->ok = false; finished = true; break; }
                } else if (rc != Z_OK && rc != Z_BUF_ERROR) {
                    c->ok = false;
                    finished = true;
                    break;
                } else if (rc == Z_BUF_ERROR && z.avail_in == 0 && input_done) {
                    c->ok = false;
                    finished = true;
                    break;
                }
            }
            c->bytes.resize(CHUNK_BYTES - z.avail_out);
            const bool stop = finished || !c->ok;
            queue_.finish(c);
            if (stop) break;
        }
        inflateEnd(&z);
        queue_.close();
    }
    void cut_bgzf()
    {
        bool more = true, any_member = false;
        while (more) {
            auto c = queue_.reserve();
            if (!c) break;
            InflatePool::Job job;
            job.out = c;
            c->bytes = spare_.take();
            job.packed = spare_.take();
            for (int b = 0; b < BGZF_GROUP; ++b) {
                unsigned char head[18];
                const size_t got = std::fread(head, 1, sizeof(head), f_);
                if (got == 0) { more = false; break; }  // clean end of file
                if (any_member && (got < 2 || head[0] != 0x1f || head[1] != 0x8b)) { more = false; break; }  // trailing garbage: the end, as for gzread
                const bool good = got == 18 && head[0] == 0x1f && head[1] == 0x8b && head[2] == 8 && (head[3] & 4) && head[12] == 'B' && head[13] == 'C' &&
                                  head[14] == 2 && head[15] == 0 && (head[10] | (head[11] << 8)) >= 6;
                const size_t bsize = good ? ((size_t)head[16] | ((size_t)head[17] << 8)) + 1 : 0;
                if (!good || bsize < 26) { c->ok = false; more = false; break; }
                const size_t at = job.packed.size();
                job.packed.resize(at + bsize);
                std::memcpy(job.packed.data() + at, head, 18);
                if (std::fread(job.packed.data() + at + 18, 1, bsize - 18, f_) != bsize - 18) {
                    job.packed.resize(at);
                    c->ok = false;
                    more = false;
                    break;
                }
                any_member = true;
            }
            if (!c->ok) job.packed.clear();
            pool_->submit(std::move(job));
        }
        queue_.close();
    }

    FILE* f_;
    OrderedQueue queue_;
    SpareBuffers spare_;
    std::unique_ptr<InflatePool> pool_;
    std::thread feeder_;
    const char* kind_ = "plain";
    int threads_;
    bool started_ = false;
    uint64_t direct_off_ = (uint64_t)-1, direct_size_ = 0;  // read_direct: where the next read starts, where the reads end (regular file)
    bool direct_regular_ = false;
    uint64_t range_first_ = 0, range_end_ = ~0ULL;          // the reader's part of a plain file
};

inline bool is_blank(int c) { return c == ' ' || (c >= '\t' && c <= '\r'); }  // isspace() of the C locale

// Forward cursor over the chunks of a ByteSource.
class Cursor {
public:
    explicit Cursor(ByteSource& s) : src_(s) {}
    bool broken() const { return broken_; }
    // make at least one byte available; false at the end of the stream
    bool more()
    {
        while (pos_ >= cur_.bytes.size()) {
            if (ended_) return false;
            Chunk c;
            if (!src_.next(c)) { ended_ = true; return false; }
            if (!c.ok) broken_ = true;
            src_.recycle(std::move(cur_.bytes));
            cur_ = std::move(c);
            pos_ = 0;
            if (broken_ && cur_.bytes.empty()) { ended_ = true; return false; }
        }
        return true;
    }
    int take() { return more() ? cur_.bytes[pos_++] : -1; }
    const unsigned char* here() const { return cur_.bytes.data() + pos_; }
    size_t left() const { return cur_.bytes.size() - pos_; }
    void skip(size_t n) { pos_ += n; }
    void abandon() { src_.abandon(); }
    ByteSource& source() { return src_; }

private:
    ByteSource& src_;
    Chunk cur_;
    size_t pos_ = 0;
    bool ended_ = false, broken_ = false;
};

enum class Step { Record, End, Malformed, Broken };

struct Record {
    std::string name, comment, seq, qual;
};

class RecordParser {
public:
    explicit RecordParser(ByteSource& s) : in_(s) {}
    void abandon() { in_.abandon(); }  // the consumer is leaving: a parser waiting for bytes sees the end of the stream

    Step next(Record& r)
    {
        if (!header_seen_ && !seek_marker()) return in_.broken() ? Step::Broken : Step::End;
        header_seen_ = false;
        r.comment.clear();
        r.seq.clear();
        r.qual.clear();
        int stop = -1;
        if (!token(r.name, stop)) return in_.broken() ? Step::Broken : Step::End;  // the marker was the last byte of the file
        if (stop != '\n') {
            r.comment.clear();
            if (line_tail(r.comment)) trim_cr(r.comment);
        }
        // sequence lines, until a line opens with a marker or with the FASTQ separator
        for (;;) {
            const int c = in_.take();
            if (c < 0) return in_.broken() ? Step::Broken : Step::Record;  // FASTA record that ends the file
            if (c == '>' || c == '@') {
                header_seen_ = true;
                return Step::Record;
            }
            if (c == '+') break;
            if (c == '\n') continue;
            r.seq.push_back((char)c);
            line_tail(r.seq);
            trim_cr(r.seq);
        }
        // the rest of the separator line carries nothing
        for (;;) {
            const int c = in_.take();
            if (c < 0) return in_.broken() ? Step::Broken : Step::Malformed;  // no quality at all
            if (c == '\n') break;
        }
        do {
            if (!line_tail(r.qual)) break;
            trim_cr(r.qual);
        } while (r.qual.size() < r.seq.size());
        if (in_.broken()) return Step::Broken;
        return r.qual.size() == r.seq.size() ? Step::Record : Step::Malformed;
    }

private:
    static void trim_cr(std::string& s)
    {
        if (s.size() > 1 && s.back() == '\r') s.pop_back();
    }
    // consume up to and including the next '>' or '@'
    bool seek_marker()
    {
        while (in_.more()) {
            const unsigned char* p = in_.here();
            const size_t n = in_.left();
            size_t i = 0;
            while (i < n && p[i] != '>' && p[i] != '@') ++i;  // normally the very next byte: nothing stands between records
            if (i < n) {
                in_.skip(i + 1);
                return true;
            }
            in_.skip(n);
        }
        return false;
    }
    // bytes up to the first whitespace byte, which is consumed and reported; false if the stream had nothing left
    bool token(std::string& out, int& stop)
    {
        out.clear();
        stop = 0;
        if (!in_.more()) return false;
        do {
            const unsigned char* p = in_.here();
            const size_t n = in_.left();
            size_t i = 0;
            while (i < n && !is_blank(p[i])) ++i;
            out.append(reinterpret_cast<const char*>(p), i);
            if (i < n) {
                stop = p[i];
                in_.skip(i + 1);
                return true;
            }
            in_.skip(n);
        } while (in_.more());
        return true;
    }
    // append the rest of the current line (the newline is consumed, not stored); false if the stream had nothing left
    bool line_tail(std::string& out)
    {
        if (!in_.more()) return false;
        do {
            const unsigned char* p = in_.here();
            const size_t n = in_.left();
            const void* nl = std::memchr(p, '\n', n);
            const size_t i = nl ? (size_t)(static_cast<const unsigned char*>(nl) - p) : n;
            out.append(reinterpret_cast<const char*>(p), i);
            if (nl) {
                in_.skip(i + 1);
                return true;
            }
            in_.skip(n);
        } while (in_.more());
        return true;
    }

    Cursor in_;
    bool header_seen_ = false;  // the marker of the next record has been consumed already
};

// Raw text cut at record boundaries, for the device-side parser.  A producer thread runs ahead of the caller and fills a ring
// of three span buffers that live as long as the reader (their pages are touched once, not once per span: first touches cost
// more than everything else here; page-locked when the spans go to a device, so that the copy is one DMA); the caller holds
// one span while the next ones are being assembled.
//
// Where a span may end is decided from its own last bytes, not by counting from the start of the stream:
//   FASTQ (the stream opens with '@'): in front of a line that begins with '@' and whose second-next line begins with '+'.
//     In a 4-line record only the header satisfies that: for a quality line beginning with '@' the second-next line is a
//     sequence line, which cannot begin with '+'.
//   FASTA: in front of a line that begins with '>'.
// is `at` (the first byte of a line) where a record begins (fmt_ 'q': 4-line FASTQ, else FASTA)?  -1: the bytes that decide are not in the buffer yet
int record_opens_at(const char* p, size_t filled, size_t at, char fmt_)
{
    if (at >= filled) return -1;
    if (fmt_ != 'q') return p[at] == '>';
    if (p[at] != '@') return 0;
    const char* nl = static_cast<const char*>(std::memchr(p + at, '\n', filled - at));
    if (!nl || (size_t)(nl + 1 - p) >= filled) return -1;
    nl = static_cast<const char*>(std::memchr(nl + 1, '\n', filled - (size_t)(nl + 1 - p)));
    if (!nl || (size_t)(nl + 1 - p) >= filled) return -1;
    return nl[1] == '+';
}
// the last record boundary in (0, limit], else the first one beyond it; 0: none in the buffer
size_t find_record_cut(const char* p, size_t filled, size_t limit, char fmt_)
{
    size_t upto = limit < filled ? limit : filled;  // a newline at a position < upto opens a line at <= limit
    if (fmt_ == 'q') {
        for (int lines = 0; upto > 0 && lines < 16; ++lines) {  // a header is among any four consecutive lines
            const char* nl = static_cast<const char*>(memrchr(p, '\n', upto));
            if (!nl) break;
            const size_t at = (size_t)(nl - p) + 1;
            if (record_opens_at(p, filled, at, fmt_) == 1) return at;
            upto = (size_t)(nl - p);
        }
    } else {
        size_t last = filled ? (limit < filled - 1 ? limit : filled - 1) : 0;  // the last position a cut may take
        while (last > 0) {  // the last '>' that follows a newline
            const char* gt = static_cast<const char*>(memrchr(p + 1, '>', last));  // positions 1 .. last
            if (!gt) break;
            const size_t at = (size_t)(gt - p);
            if (p[at - 1] == '\n') return at;
            last = at - 1;
        }
    }
    size_t from = limit < filled ? limit : filled;  // a record longer than the limit: the first boundary after it
    while (from < filled) {
        const char* nl = static_cast<const char*>(std::memchr(p + from, '\n', filled - from));
        if (!nl) break;
        const size_t at = (size_t)(nl - p) + 1;
        const int v = record_opens_at(p, filled, at, fmt_);
        if (v == 1) return at;
        if (v < 0) break;
        from = at;
    }
    return 0;
}

// The first place in p[0 .. n) where a record begins that can be RECOGNISED from the text alone: the first byte of a line (a
// newline stands in front of it, inside the text) that opens a record.  0: none (offsets are >= 1); (size_t)-1: the text ends
// before the candidate in hand can be decided.  This is how two readers that take neighbouring parts of one file agree on where
// one's records end and the other's begin without talking to each other.
size_t first_record_start(const char* p, size_t n, char fmt_)
{
    size_t from = 0;
    while (from < n) {
        const char* nl = static_cast<const char*>(std::memchr(p + from, '\n', n - from));
        if (!nl) return 0;
        const size_t at = (size_t)(nl - p) + 1;
        const int v = record_opens_at(p, n, at, fmt_);
        if (v == 1) return at;
        if (v < 0) return (size_t)-1;
        from = at;
    }
    return 0;
}

struct SpanMemory {  // where span buffers come from
    void* (*get)(size_t bytes);
    void (*put)(void* p);
};

struct SpanBuf {
    char* p = nullptr;
    size_t cap = 0;
    size_t n = 0;     // bytes of the span handed to the caller; what follows them opens the next span
    int state = 0;    // 0 free, 1 ready, 2 held by the caller
    int verdict = 1;  // 1 span, 0 end of stream, -1 damaged stream
};

class TextCutter {
public:
    static constexpr int RING = 3;
    static constexpr size_t LOOK = 64u << 10;  // read this far past the limit before looking for the cut
    TextCutter(ByteSource& s, size_t limit, SpanMemory mem) : in_(s), limit_(limit), mem_(mem) { worker_ = std::thread([this] { produce(); }); }
    ~TextCutter()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        in_.abandon();  // a producer waiting for bytes wakes up with "end of stream"
        if (worker_.joinable()) worker_.join();
        for (auto& b : ring_)
            if (b.p) mem_.put(b.p);
    }
    size_t limit() const { return limit_; }
    // next span (valid until the following call): 1 = span ready, 0 = end of stream, -1 = damaged stream
    int next(const char** text, size_t* n)
    {
        std::unique_lock<std::mutex> lk(m_);
        if (held_ >= 0) {
            ring_[held_].state = 0;
            held_ = -1;
            cv_.notify_all();
        }
        if (ended_) return ended_verdict_;
        SpanBuf& b = ring_[take_];
        cv_.wait(lk, [&] { return b.state == 1; });
        if (b.verdict != 1) {
            ended_ = true;
            ended_verdict_ = b.verdict;
            return b.verdict;
        }
        b.state = 2;
        held_ = take_;
        take_ = (take_ + 1) % RING;
        *text = b.p;
        *n = b.n;
        return 1;
    }

private:
    bool grow(SpanBuf& b, size_t keep, size_t need)
    {
        if (need <= b.cap) return true;
        size_t want = b.cap ? b.cap : (size_t)1 << 20;
        while (want < need) want *= 4;
        const size_t full = limit_ + LOOK + CHUNK_BYTES;  // what a span of ordinary records needs at most
        if (want > full && need <= full) want = full;
        char* q = static_cast<char*>(mem_.get(want));
        if (!q) return false;
        if (keep) std::memcpy(q, b.p, keep);
        if (b.p) mem_.put(b.p);
        b.p = q;
        b.cap = want;
        return true;
    }
    void produce()
    {
        int put = 0;
        const char* carry = nullptr;  // the bytes behind the previous span's cut (they stay in the previous buffer until copied)
        size_t carry_n = 0;
        bool eof = false;
        const bool direct = in_.source().can_read_direct();
        for (;;) {
            SpanBuf* b;
            {
                std::unique_lock<std::mutex> lk(m_);
                b = &ring_[put];
                cv_.wait(lk, [&] { return quit_ || b->state == 0; });
                if (quit_) return;
            }
            int verdict = 1;
            size_t filled = 0, cut = 0, want = limit_ + LOOK;
            if (!grow(*b, 0, carry_n + 1)) verdict = -1;
            if (verdict == 1 && carry_n) std::memcpy(b->p, carry, carry_n);
            if (verdict == 1) filled = carry_n;
            while (verdict == 1) {
                if (filled && !fmt_) fmt_ = b->p[0] == '@' ? 'q' : 'a';
                if (eof && filled == 0) { verdict = 0; break; }
                if (eof && filled <= limit_) { cut = filled; break; }
                if (filled >= want || eof) {
                    cut = find_record_cut(b->p, filled, limit_, fmt_);
                    if (cut == 0 && eof) cut = filled;  // the tail of the file, whatever it is
                    if (cut) break;
                    want = filled + LOOK;  // one record longer than all this: keep reading
                }
                if (direct) {  // an uncompressed file: the bytes go from the page cache straight into the span
                    // doubling steps while the buffer is still growing (a small file gets a small buffer); once it has its full
                    // size, everything that is missing in one (parallel) read
                    size_t step = b->cap >= want ? want - filled : (filled > ((size_t)1 << 20) ? filled : (size_t)1 << 20);
                    if (step > want - filled) step = want - filled;
                    if (!grow(*b, filled, filled + step)) { verdict = -1; break; }
                    const long got = in_.source().read_direct(b->p + filled, step);
                    if (got < 0) { verdict = -1; break; }
                    if ((size_t)got < step) eof = true;
                    filled += (size_t)got;
                    continue;
                }
                if (!in_.more()) {
                    if (in_.broken()) { verdict = -1; break; }
                    eof = true;
                    continue;
                }
                if (in_.broken()) { verdict = -1; break; }
                size_t add = in_.left();
                if (add > want - filled) add = want - filled;
                if (!grow(*b, filled, filled + add)) { verdict = -1; break; }
                std::memcpy(b->p + filled, in_.here(), add);
                in_.skip(add);
                filled += add;
            }
            if (verdict == 1 && in_.broken()) verdict = -1;
            b->n = cut;
            b->verdict = verdict;
            carry = b->p + cut;
            carry_n = verdict == 1 ? filled - cut : 0;
            {
                std::lock_guard<std::mutex> lk(m_);
                b->state = 1;
            }
            cv_.notify_all();
            if (verdict != 1) return;
            put = (put + 1) % RING;
        }
    }

    Cursor in_;
    const size_t limit_;
    const SpanMemory mem_;
    char fmt_ = 0;
    SpanBuf ring_[RING];
    std::mutex m_;
    std::condition_variable cv_;
    std::thread worker_;
    int take_ = 0, held_ = -1;
    bool quit_ = false, ended_ = false;
    int ended_verdict_ = 0;
};

// BGZF that goes to the device compressed: a thread reads the file into a ring of page-locked buffers and walks the member
// headers (that is all the host does: bl_inflate.hip inflates on the GPU); a span holds whole members with at most `limit`
// bytes of text between them.
struct PackedSpan {
    char* p = nullptr;
    size_t cap = 0;
    size_t n = 0;  // bytes of whole members; what follows them opens the next span
    std::vector<bl_bgzf_member> members;
    uint64_t text_bytes = 0;
    uint64_t own_text = 0;  // text of the members that lie inside the reader's own part of the file (a prefix of the span's text;
                            // less than text_bytes only for a sharded reader, whose last record ends in its neighbour's part)
    int state = 0;    // 0 free, 1 ready, 2 held by the caller
    int verdict = 1;  // 1 span, 0 end of file, -1 damaged file
};

class PackedSpans {
public:
    static constexpr int RING = 3;
    static constexpr size_t READ = 4u << 20;
    // start / own_end: the part of the file that is this reader's own (whole file: 0 / ~0); members behind own_end are still
    // delivered (the caller needs them to finish its last record) but in small spans
    PackedSpans(FILE* f, size_t limit, SpanMemory mem, uint64_t start = 0, uint64_t own_end = ~0ULL) : f_(f), limit_(limit), mem_(mem), start_(start), own_end_(own_end)
    {
        worker_ = std::thread([this] { produce(); });
    }
    ~PackedSpans()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
        for (auto& b : ring_)
            if (b.p) mem_.put(b.p);
        if (f_) std::fclose(f_);
    }
    int next(PackedSpan** out)  // the span stays valid until the following call
    {
        std::unique_lock<std::mutex> lk(m_);
        if (held_ >= 0) {
            ring_[held_].state = 0;
            held_ = -1;
            cv_.notify_all();
        }
        if (ended_) return ended_verdict_;
        PackedSpan& b = ring_[take_];
        cv_.wait(lk, [&] { return b.state == 1; });
        if (b.verdict != 1) {
            ended_ = true;
            ended_verdict_ = b.verdict;
            return b.verdict;
        }
        b.state = 2;
        held_ = take_;
        take_ = (take_ + 1) % RING;
        *out = &b;
        return 1;
    }

private:
    bool grow(PackedSpan& b, size_t keep, size_t need)
    {
        if (need <= b.cap) return true;
        size_t want = b.cap ? b.cap : 2 * READ;
        while (want < need) want *= 2;
        char* q = static_cast<char*>(mem_.get(want));
        if (!q) return false;
        if (keep) std::memcpy(q, b.p, keep);
        if (b.p) mem_.put(b.p);
        b.p = q;
        b.cap = want;
        return true;
    }
    void produce()
    {
        int put = 0;
        const char* carry = nullptr;
        size_t carry_n = 0;
        bool eof = false;
        uint64_t file_off = start_;  // where in the file the current span's first byte lies
        if (start_ && fseeko(f_, (off_t)start_, SEEK_SET) != 0) eof = true;
        for (;;) {
            PackedSpan* b;
            {
                std::unique_lock<std::mutex> lk(m_);
                b = &ring_[put];
                cv_.wait(lk, [&] { return quit_ || b->state == 0; });
                if (quit_) return;
            }
            int verdict = 1;
            size_t filled = 0, walked = 0;
            uint64_t text = 0, own_text = 0;
            unsigned borrowed = 0;
            b->members.clear();
            if (!grow(*b, 0, carry_n + READ)) verdict = -1;
            if (verdict == 1 && carry_n) std::memcpy(b->p, carry, carry_n);
            if (verdict == 1) filled = carry_n;
            bool full = false;
            while (verdict == 1 && !full) {
                for (;;) {  // the whole members that are here
                    bl_bgzf_member m;
                    uint64_t got = 0, used = 0, tb = 0;
                    if (seen_member_ && filled > walked) {
                        // behind a complete member, bytes that do not open with the gzip magic are trailing garbage: the end of
                        // the stream, as for gzread (the reference's drivers read through it: tests/kseq.h over gzFile)
                        const unsigned char* q = reinterpret_cast<const unsigned char*>(b->p) + walked;
                        const size_t have = filled - walked;
                        if (q[0] != 0x1f || (have >= 2 ? q[1] != 0x8b : eof)) {
                            filled = walked;
                            eof = true;
                            break;
                        }
                    }
                    if (bl_bgzf_walk(b->p + walked, filled - walked, walked, text, &m, 1, &got, &used, &tb) != BL_OK) { verdict = -1; break; }
                    if (got == 0) break;
                    // a span is full when its text would pass the limit — or when it holds as many members as the inflate kernel
                    // keeps resident per 64 MiB of limit (1,024: four waves on each of 256 CUs): one member more would cost a
                    // second round of the kernel for that member alone
                    const bool own = file_off + walked < own_end_;
                    if (!b->members.empty() && (text + m.isize > limit_ || b->members.size() >= max_members_)) { full = true; break; }
                    if (!own && borrowed >= 16) { full = true; break; }  // (members of the neighbour's part: a few at a time)
                    b->members.push_back(m);
                    seen_member_ = true;
                    walked += used;
                    text += m.isize;
                    if (own) own_text = text;
                    else ++borrowed;
                }
                if (verdict != 1 || full || eof) break;
                if (!grow(*b, filled, filled + READ)) { verdict = -1; break; }
                const size_t n = std::fread(b->p + filled, 1, READ, f_);
                if (n < READ) {
                    if (std::ferror(f_)) { verdict = -1; break; }
                    eof = true;
                }
                filled += n;
            }
            if (verdict == 1 && b->members.empty()) verdict = filled > walked ? -1 : 0;  // a member cut short by the end of the file / the end
            b->n = walked;
            b->text_bytes = text;
            b->own_text = own_text;
            b->verdict = verdict;
            carry = b->p + walked;
            carry_n = verdict == 1 ? filled - walked : 0;
            file_off += walked;
            {
                std::lock_guard<std::mutex> lk(m_);
                b->state = 1;
            }
            cv_.notify_all();
            if (verdict != 1) return;
            put = (put + 1) % RING;
        }
    }

    FILE* f_;
    bool seen_member_ = false;  // produce(): at least one whole member has been read
    const size_t limit_;
    const size_t max_members_ = (limit_ >> 16) ? (limit_ >> 16) : 1;
    const SpanMemory mem_;
    const uint64_t start_, own_end_;
    PackedSpan ring_[RING];
    std::mutex m_;
    std::condition_variable cv_;
    std::thread worker_;
    int take_ = 0, held_ = -1;
    bool quit_ = false, ended_ = false;
    int ended_verdict_ = 0;
};

}  // namespace

// Batches of whole records for bl_reader_next_batch, assembled one batch AHEAD of the caller by a thread of the reader: while
// the caller uploads, scans or walks batch i (include/compat/read_pool.hpp walks it k-mer by k-mer), batch i + 1 is being parsed.
struct HostBatch {
    std::string bases;
    std::vector<uint64_t> offsets;
    std::vector<std::string> names;
    Step end = Step::Record;  // Record: a batch; End: nothing left; Malformed / Broken: the file failed here
    int state = 0;            // 0 free, 1 ready, 2 held by the caller
};

class RecordBatcher {
public:
    RecordBatcher(ByteSource& src, uint64_t max_bases) : parser_(src), max_bases_(max_bases) { worker_ = std::thread([this] { produce(); }); }
    ~RecordBatcher()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            quit_ = true;
        }
        cv_.notify_all();
        parser_source_abandon();
        if (worker_.joinable()) worker_.join();
    }
    uint64_t max_bases() const { return max_bases_; }
    // the next batch (valid until the following call); its `end` says whether it is one
    HostBatch* next()
    {
        std::unique_lock<std::mutex> lk(m_);
        if (held_ >= 0) {
            slot_[held_].state = 0;
            held_ = -1;
            cv_.notify_all();
        }
        if (last_) return last_;
        HostBatch& b = slot_[take_];
        cv_.wait(lk, [&] { return b.state == 1; });
        if (b.end != Step::Record) {
            last_ = &b;  // the end (or the failure) stays the answer
            return last_;
        }
        b.state = 2;
        held_ = take_;
        take_ ^= 1;
        return &b;
    }

private:
    void parser_source_abandon() { parser_.abandon(); }
    void produce()
    {
        int put = 0;
        bool have_pending = false;
        Record rec;
        for (;;) {
            HostBatch* b;
            {
                std::unique_lock<std::mutex> lk(m_);
                b = &slot_[put];
                cv_.wait(lk, [&] { return quit_ || b->state == 0; });
                if (quit_) return;
            }
            b->bases.clear();
            b->offsets.assign(1, 0);
            b->names.clear();
            b->end = Step::Record;
            if (max_bases_ && max_bases_ <= ((uint64_t)1 << 30) && b->bases.capacity() < max_bases_) b->bases.reserve((size_t)max_bases_);  // (no regrowth copies)
            for (;;) {
                if (!have_pending) {
                    const Step st = parser_.next(rec);
                    if (st != Step::Record) {
                        if (b->names.empty() || st != Step::End) b->end = st;  // (records in front of a clean end are a last batch)
                        else end_after_ = true;
                        break;
                    }
                }
                have_pending = false;
                if (!b->names.empty() && max_bases_ && b->bases.size() + rec.seq.size() > max_bases_) {
                    have_pending = true;  // the record opens the next batch
                    break;
                }
                b->bases += rec.seq;
                b->offsets.push_back(b->bases.size());
                b->names.push_back(rec.name);
            }
            const bool stop = b->end != Step::Record;
            {
                std::lock_guard<std::mutex> lk(m_);
                b->state = 1;
            }
            cv_.notify_all();
            if (stop) return;
            put ^= 1;
            if (end_after_) {  // the file ended behind that batch: the next answer is "nothing left"
                HostBatch* e;
                {
                    std::unique_lock<std::mutex> lk(m_);
                    e = &slot_[put];
                    cv_.wait(lk, [&] { return quit_ || e->state == 0; });
                    if (quit_) return;
                }
                e->bases.clear();
                e->offsets.assign(1, 0);
                e->names.clear();
                e->end = Step::End;
                {
                    std::lock_guard<std::mutex> lk(m_);
                    e->state = 1;
                }
                cv_.notify_all();
                return;
            }
        }
    }

    RecordParser parser_;
    const uint64_t max_bases_;
    HostBatch slot_[2];
    std::mutex m_;
    std::condition_variable cv_;
    std::thread worker_;
    int take_ = 0, held_ = -1;
    HostBatch* last_ = nullptr;
    bool quit_ = false, end_after_ = false;
};

// Device side of the compressed path: the members of a span are inflated into `text[cur]` behind the bytes the previous span
// left over (the part of its last record that was not complete yet), the text is cut at its last record boundary, parsed there,
// and what lies behind the cut moves to the front of the other buffer.
struct DeviceBgzf {
    std::unique_ptr<PackedSpans> spans;
    bl_ctx* ctx = nullptr;
    void *d_packed = nullptr, *d_members = nullptr;
    uint32_t* d_status = nullptr;
    uint8_t* d_text[2] = {nullptr, nullptr};
    size_t packed_cap = 0, members_cap = 0, text_cap[2] = {0, 0};
    int cur = 0;
    size_t carry_n = 0;
    bool eof = false;
    char first_byte = 0;
    // a reader of one part of the file (bl_reader_open_shard): the text in front of the first recognisable record start belongs
    // to the part before (trim_front, parts 1 ..), and this part's last record is finished from the members of the next one:
    // ext_start = where in the current text those borrowed members begin (npos: not reached yet)
    bool trim_front = false, finished = false;
    size_t ext_start = (size_t)-1;
    // what comes back from the device lands in page-locked memory: a copy into pageable memory would make the host wait for
    // everything queued in front of it (the inflate kernel), and nothing would overlap
    char* window = nullptr;
    uint32_t* status = nullptr;
    char* first_back = nullptr;
    bl_bgzf_member* members_up = nullptr;  // (and the member table goes up from page-locked memory, for the same reason)
    size_t window_cap = 0, status_cap = 0, members_up_cap = 0;
    hipStream_t stream = nullptr;  // upload + inflate of the NEXT span run here while the current text is parsed on the context's stream
    struct {
        bool active = false;
        size_t n_members = 0, filled = 0;
        int buf = 0;
    } pending;
    double t_launch = 0, t_wait = 0, t_parse = 0;  // seconds spent starting spans, waiting for inflate, parsing (BL_INGEST_TRACE=1 prints them)
    unsigned n_batches = 0;
    ~DeviceBgzf()
    {
        if (std::getenv("BL_INGEST_TRACE"))
            std::fprintf(stderr, "[bl ingest] compressed path: %u batches, launch %.1f ms, wait for inflate %.1f ms, parse %.1f ms\n", n_batches, t_launch * 1e3,
                         t_wait * 1e3, t_parse * 1e3);
        if (stream) {
            (void)hipStreamSynchronize(stream);
            (void)hipStreamDestroy(stream);
        }
        spans.reset();
        for (void* p : {d_packed, d_members, (void*)d_text[0], (void*)d_text[1]})  // (d_status lies inside d_members)
            if (p) (void)hipFree(p);
        for (void* p : {(void*)window, (void*)status, (void*)first_back, (void*)members_up})
            if (p) (void)hipHostFree(p);
    }
};

struct bl_reader {
    std::string path;
    // a reader of one part of a BGZF file (bl_reader_open_shard): the part's members lie in [shard_start, shard_end) of the file
    bool sharded = false, shard_first = true;
    uint64_t shard_start = 0, shard_end = ~0ULL;
    char shard_first_byte = 0;  // first byte of the FILE's text: tells FASTQ from FASTA to a part that does not begin with a record
    std::unique_ptr<DeviceBgzf> packed;
    std::unique_ptr<ByteSource> source;
    std::unique_ptr<RecordBatcher> batcher;  // bl_reader_next_batch: batches parsed one ahead
    HostBatch* batch = nullptr;              // the one handed out last
    std::unique_ptr<RecordParser> records;
    std::unique_ptr<TextCutter> text;
    Record rec;  // bl_reader_next_record: the record handed out last
};

namespace {

int step_error(Step s)
{
    if (s == Step::Malformed) return bl_set_error(BL_ERR_INVALID, "truncated or mismatched FASTQ quality string");
    return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
}

}  // namespace

extern "C" {

int bl_reader_open_threads(const char* path, int threads, bl_reader** out)
{
    if (!path || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    FILE* f = std::fopen(path, "rb");
    if (!f) return bl_set_error(BL_ERR_INVALID, (std::string("cannot open ") + path).c_str());
    if (threads <= 0) {
        const unsigned hw = std::thread::hardware_concurrency();
        threads = hw == 0 ? 4 : (hw > 16 ? 16 : (int)hw);
    }
    bl_reader* r = new (std::nothrow) bl_reader();
    if (!r) {
        std::fclose(f);
        return bl_set_error(BL_ERR_OOM, "host allocation failed");
    }
    r->path = path;
    r->source.reset(new ByteSource(f, threads));
    *out = r;
    return BL_OK;
}

int bl_reader_open(const char* path, bl_reader** out) { return bl_reader_open_threads(path, 0, out); }

namespace {

bool bgzf_header_ok(const unsigned char* h, uint64_t* bsize)
{
    const bool good = h[0] == 0x1f && h[1] == 0x8b && h[2] == 8 && (h[3] & 4) && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0 &&
                      ((unsigned)h[10] | ((unsigned)h[11] << 8)) >= 6;
    *bsize = ((uint64_t)h[16] | ((uint64_t)h[17] << 8)) + 1;
    return good && *bsize >= 26;
}

// The first member boundary at or behind `from`: a BGZF header whose successor (where its BSIZE says) is a header too, or the
// end of the file.  Members are at most 64 KiB long, so one lies within 64 KiB; `size` when there is none.
uint64_t member_boundary_from(int fd, uint64_t from, uint64_t size)
{
    if (from >= size) return size;
    std::vector<unsigned char> buf((size_t)((size - from) < (131072 + 18) ? (size - from) : (131072 + 18)));
    size_t have = 0;
    while (have < buf.size()) {
        const ssize_t got = pread(fd, buf.data() + have, buf.size() - have, (off_t)(from + have));
        if (got <= 0) break;
        have += (size_t)got;
    }
    for (size_t i = 0; i + 18 <= have; ++i) {
        uint64_t bsize = 0;
        if (buf[i] != 0x1f || !bgzf_header_ok(buf.data() + i, &bsize)) continue;
        const uint64_t next = from + i + bsize;
        if (next == size) return from + i;
        unsigned char h[18];
        uint64_t b2 = 0;
        if (next + 18 <= size && pread(fd, h, 18, (off_t)next) == 18 && bgzf_header_ok(h, &b2) && next + b2 <= size) return from + i;
    }
    return size;
}

}  // namespace

int bl_reader_open_shard(const char* path, uint32_t rank, uint32_t world, bl_reader** out)
{
    if (!path || !out || world == 0 || rank >= world) return bl_set_error(BL_ERR_INVALID, "bad argument (rank < world)");
    int rc = bl_reader_open_threads(path, 0, out);
    if (rc != BL_OK) return rc;
    bl_reader* r = *out;
    auto fail_close = [&](const char* msg) {
        bl_reader_close(r);
        *out = nullptr;
        return bl_set_error(BL_ERR_INVALID, msg);
    };
    if (r->source->kind()[0] == 'g') return fail_close("reading a file in parts needs plain text or BGZF (bgzip): one gzip stream has no entry points");
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
        if (fd >= 0) close(fd);
        return fail_close("cannot stat the file (a regular file is needed)");
    }
    const uint64_t size = (uint64_t)st.st_size;
    if (r->source->kind()[0] == 'p') {
        // plain text: the parts are byte ranges that meet where a record begins — the first one a reader can recognise from
        // the text behind size / world * rank alone (first_record_start), which both neighbours work out the same way
        char first = 0;
        const bool has_text = pread(fd, &first, 1, 0) == 1;
        const char fmt = first == '@' ? 'q' : 'a';
        bool io_ok = true;
        auto boundary = [&](uint64_t from) -> uint64_t {
            if (from == 0) return 0;
            std::vector<char> w;
            for (size_t look = (size_t)1 << 20; from < size; look *= 4) {
                const size_t n = (size_t)(size - from < look ? size - from : look);
                w.resize(n);
                size_t got = 0;
                while (got < n) {
                    const ssize_t g = pread(fd, w.data() + got, n - got, (off_t)(from + got));
                    if (g <= 0) { io_ok = false; return size; }
                    got += (size_t)g;
                }
                const size_t at = first_record_start(w.data(), n, fmt);
                if (at != 0 && at != (size_t)-1) return from + at;
                if (n == size - from) break;  // the rest of the file holds no further record start
            }
            return size;
        };
        r->shard_start = has_text ? boundary(size / world * rank) : 0;
        r->shard_end = !has_text ? 0 : (rank + 1 == world ? size : boundary(size / world * (rank + 1)));
        close(fd);
        if (!io_ok) return fail_close("cannot read the file");
        r->shard_first = rank == 0;
        r->source->set_range(r->shard_start, r->shard_end);
        return BL_OK;
    }
    r->sharded = true;
    r->shard_first = rank == 0;
    r->shard_start = rank == 0 ? 0 : member_boundary_from(fd, size / world * rank, size);
    r->shard_end = rank + 1 == world ? ~0ULL : member_boundary_from(fd, size / world * (rank + 1), size);
    // the first byte of the file's text (FASTQ or FASTA?): the first member that holds text, inflated here
    {
        std::vector<unsigned char> m(65536 + 64);
        unsigned char first = 0;
        uint64_t at = 0;
        for (int tries = 0; tries < 64 && !first && at < size; ++tries) {
            const ssize_t got = pread(fd, m.data(), m.size(), (off_t)at);
            uint64_t bsize = 0;
            if (got < 28 || !bgzf_header_ok(m.data(), &bsize) || (uint64_t)got < bsize) break;
            const size_t xlen = (size_t)m[10] | ((size_t)m[11] << 8);
            z_stream z;
            std::memset(&z, 0, sizeof(z));
            if (12 + xlen + 8 <= bsize && inflateInit2(&z, -15) == Z_OK) {
                z.next_in = m.data() + 12 + xlen;
                z.avail_in = (uInt)(bsize - 12 - xlen - 8);
                z.next_out = &first;
                z.avail_out = 1;
                (void)inflate(&z, Z_SYNC_FLUSH);
                if (z.avail_out != 0) first = 0;
                inflateEnd(&z);
            }
            at += bsize;  // (a member without text: look at the next one)
        }
        r->shard_first_byte = (char)first;
    }
    close(fd);
    if (!r->shard_first_byte) return fail_close("cannot read the first member of the file");
    return BL_OK;
}

int bl_reader_shard_range(bl_reader* r, uint64_t* first_byte, uint64_t* end_byte)
{
    if (!r || !first_byte || !end_byte) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *first_byte = r->shard_start;
    *end_byte = r->shard_end;
    return BL_OK;
}

int bl_reader_close(bl_reader* r)
{
    if (!r) return BL_OK;
    r->batcher.reset();
    r->records.reset();
    r->text.reset();
    r->packed.reset();
    r->source.reset();
    delete r;
    return BL_OK;
}

const char* bl_reader_kind(bl_reader* r) { return r && r->source ? r->source->kind() : ""; }

int bl_reader_next_record(bl_reader* r, const char** name, const char** seq, uint64_t* seq_len)
{
    if (!r || !seq_len) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (r->sharded) return bl_set_error(BL_ERR_INVALID, "a reader of one part of a file delivers device batches only");
    if (r->text || r->packed) return bl_set_error(BL_ERR_INVALID, "this reader is delivering text spans: records and spans cannot be mixed");
    if (r->batcher) return bl_set_error(BL_ERR_INVALID, "this reader is delivering batches: records and batches cannot be mixed");
    if (!r->records) r->records.reset(new RecordParser(*r->source));
    const Step s = r->records->next(r->rec);
    if (s == Step::End) {
        *seq_len = 0;
        if (name) *name = nullptr;
        if (seq) *seq = nullptr;
        return 1;
    }
    if (s != Step::Record) return step_error(s);
    if (name) *name = r->rec.name.c_str();
    if (seq) *seq = r->rec.seq.data();
    *seq_len = r->rec.seq.size();
    return BL_OK;
}

int bl_reader_next_batch(bl_ctx* ctx, bl_reader* r, uint64_t max_bases, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases)
{
    if (!ctx || !r || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (r->sharded) return bl_set_error(BL_ERR_INVALID, "a reader of one part of a file delivers device batches only");
    *out = nullptr;
    if (r->text || r->packed) return bl_set_error(BL_ERR_INVALID, "this reader is delivering text spans: records and spans cannot be mixed");
    if (r->records) return bl_set_error(BL_ERR_INVALID, "this reader is delivering single records: records and batches cannot be mixed");
    if (!r->batcher) r->batcher.reset(new RecordBatcher(*r->source, max_bases));
    if (r->batcher->max_bases() != max_bases) return bl_set_error(BL_ERR_INVALID, "the batch size of a reader is fixed by its first batch call");
    r->batch = r->batcher->next();
    const HostBatch& b = *r->batch;
    if (b.end == Step::Malformed || b.end == Step::Broken) return step_error(b.end);
    if (n_seqs) *n_seqs = b.names.size();
    if (n_bases) *n_bases = b.bases.size();
    if (b.end == Step::End || b.names.empty()) return BL_OK;  // end of file: *out stays NULL
    return bl_batch_upload(ctx, b.bases.data(), b.bases.size(), b.offsets.data(), b.names.size(), out);
}

namespace {

const SpanMemory HOST_MEMORY = {[](size_t n) { return std::malloc(n); }, [](void* p) { std::free(p); }};
// page-locked, visible to every device: the H2D copy of a span is then a single DMA transfer at link speed
const SpanMemory PINNED_MEMORY = {[](size_t n) {
                                      void* p = nullptr;
                                      return hipHostMalloc(&p, n, hipHostMallocPortable) == hipSuccess ? p : nullptr;
                                  },
                                  [](void* p) { (void)hipHostFree(p); }};

int next_span(bl_reader* r, uint64_t max_bytes, const SpanMemory& mem, const char** text, uint64_t* n_bytes)
{
    if (r->sharded) return bl_set_error(BL_ERR_INVALID, "a reader of one part of a file delivers device batches only");
    if (r->records || r->packed || r->batcher) return bl_set_error(BL_ERR_INVALID, "this reader is delivering records: records and spans cannot be mixed");
    const size_t limit = max_bytes ? (size_t)max_bytes : (size_t)64 << 20;
    if (!r->text) r->text.reset(new TextCutter(*r->source, limit, mem));
    if (r->text->limit() != limit) return bl_set_error(BL_ERR_INVALID, "the span size of a reader is fixed by its first span call");
    *text = nullptr;
    *n_bytes = 0;
    size_t n = 0;
    const int rc = r->text->next(text, &n);
    if (rc < 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
    if (rc == 0) return 1;  // end of file
    *n_bytes = n;
    return BL_OK;
}

}  // namespace

int bl_reader_next_text(bl_reader* r, uint64_t max_bytes, const char** text, uint64_t* n_bytes)
{
    if (!r || !text || !n_bytes) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    return next_span(r, max_bytes, HOST_MEMORY, text, n_bytes);
}

namespace {

#define R_HIP(call)                                                                                        \
    do {                                                                                                   \
        const hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) return bl_set_error(e_ == hipErrorOutOfMemory ? BL_ERR_OOM : BL_ERR_HIP, hipGetErrorString(e_)); \
    } while (0)

// page-locked host memory of at least `need` bytes (contents not kept)
int pinned_reserve_raw(void** p, size_t* cap, size_t need)
{
    if (need <= *cap) return BL_OK;
    size_t want = *cap ? *cap : 4096;
    while (want < need) want *= 2;
    if (*p) R_HIP(hipHostFree(*p));
    *p = nullptr;
    *cap = 0;
    void* q = nullptr;
    R_HIP(hipHostMalloc(&q, want, hipHostMallocPortable));
    *p = q;
    *cap = want;
    return BL_OK;
}
#define pinned_reserve(pp, cap, need) pinned_reserve_raw(reinterpret_cast<void**>(pp), cap, need)

// a device buffer of at least `need` bytes that keeps its first `keep` bytes
int device_reserve(void** p, size_t* cap, size_t need, size_t keep, hipStream_t s)
{
    if (need <= *cap) return BL_OK;
    size_t want = *cap ? *cap : (size_t)1 << 20;
    while (want < need) want *= 2;
    void* q = nullptr;
    R_HIP(hipMalloc(&q, want));
    if (keep) {
        const hipError_t e = hipMemcpyAsync(q, *p, keep, hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) { (void)hipFree(q); return bl_set_error(BL_ERR_HIP, hipGetErrorString(e)); }
    }
    R_HIP(hipStreamSynchronize(s));  // nothing queued may still use the old buffer
    if (*p) R_HIP(hipFree(*p));
    *p = q;
    *cap = want;
    return BL_OK;
}

// Take the next packed span and start inflating it into text[buf] behind the `offset` bytes that are there already.
int launch_packed(DeviceBgzf& d, int buf, size_t offset)
{
    hipStream_t s = d.stream;
    PackedSpan* sp = nullptr;
    if (!d.eof) {
        const int rc = d.spans->next(&sp);  // (releases the previous span's buffer: its upload was waited for)
        if (rc < 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
        if (rc == 0) d.eof = true;
    }
    const size_t add = sp ? (size_t)sp->text_bytes : 0, filled = offset + add;
    if (sp && d.ext_start == (size_t)-1 && sp->own_text < sp->text_bytes) d.ext_start = offset + (size_t)sp->own_text;
    {
        void* p = d.d_text[buf];
        const int rc = device_reserve(&p, &d.text_cap[buf], filled + 64, offset, s);
        d.d_text[buf] = static_cast<uint8_t*>(p);
        if (rc != BL_OK) return rc;
    }
    const size_t n_members = sp ? sp->members.size() : 0;
    if (n_members) {
        int rc = device_reserve(&d.d_packed, &d.packed_cap, sp->n + 8, 0, s);
        if (rc == BL_OK) rc = device_reserve(&d.d_members, &d.members_cap, n_members * (sizeof(bl_bgzf_member) + sizeof(uint32_t)), 0, s);
        if (rc != BL_OK) return rc;
        d.d_status = reinterpret_cast<uint32_t*>(static_cast<char*>(d.d_members) + n_members * sizeof(bl_bgzf_member));
        R_HIP(hipMemcpyAsync(d.d_packed, sp->p, sp->n, hipMemcpyHostToDevice, s));
        rc = pinned_reserve(&d.members_up, &d.members_up_cap, n_members * sizeof(bl_bgzf_member));
        if (rc != BL_OK) return rc;
        std::memcpy(d.members_up, sp->members.data(), n_members * sizeof(bl_bgzf_member));  // (the previous table's copy was waited for)
        R_HIP(hipMemcpyAsync(d.d_members, d.members_up, n_members * sizeof(bl_bgzf_member), hipMemcpyHostToDevice, s));
        rc = bl_bgzf_inflate_on(s, d.d_packed, sp->n, static_cast<const bl_bgzf_member*>(d.d_members), n_members, d.d_text[buf] + offset, add, d.d_status);
        if (rc != BL_OK) return rc;
        rc = pinned_reserve(&d.status, &d.status_cap, n_members * sizeof(uint32_t));
        if (rc != BL_OK) return rc;
        R_HIP(hipMemcpyAsync(d.status, d.d_status, n_members * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    }
    d.pending.active = true;
    d.pending.n_members = n_members;
    d.pending.filled = filled;
    d.pending.buf = buf;
    return BL_OK;
}

// One batch from the compressed path (*out NULL at the end of the file).  The span after the one returned is uploaded and
// inflated on the reader's own stream before this call parses its text: the upload, the launch and the host's part overlap the
// parse and whatever the caller does with the batch.  (The kernels themselves take turns: four members per CU hold all of its
// LDS, so a scan queued behind a span's inflate starts when that has drained — tests/perf/ingest_trace.py.)
int next_batch_packed(bl_ctx* ctx, bl_reader* r, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases)
{
    DeviceBgzf& d = *r->packed;
    if (d.ctx != ctx) return bl_set_error(BL_ERR_INVALID, "a reader's device batches must all go to the same context");
    R_HIP(hipSetDevice(bl_ctx_device(ctx)));
    if (!d.stream) R_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    hipStream_t s = d.stream;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (;;) {
        if (d.finished) return BL_OK;  // (a part of a file: its last record has been delivered)
        if (!d.pending.active) {
            const double t0 = now();
            const int rc = launch_packed(d, d.cur, d.carry_n);
            d.t_launch += now() - t0;
            if (rc != BL_OK) return rc;
        }
        d.pending.active = false;
        const size_t filled = d.pending.filled, n_members = d.pending.n_members;
        uint8_t* const text = d.d_text[d.pending.buf];
        d.cur = d.pending.buf;
        if (filled == 0) {
            R_HIP(hipStreamSynchronize(s));  // (members without text: the end-of-file marker)
            for (size_t i = 0; i < n_members; ++i)
                if (d.status[i] != 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
            if (d.eof) return BL_OK;  // end of file
            continue;
        }
        auto fetch = [&](size_t from, size_t n) -> int {  // text[from, from + n) -> d.window, waited for; member statuses checked
            int rc = pinned_reserve(&d.window, &d.window_cap, n + 1);
            if (rc == BL_OK && !d.first_back) {
                size_t one = 0;
                rc = pinned_reserve(&d.first_back, &one, 64);
            }
            if (rc != BL_OK) return rc;
            R_HIP(hipMemcpyAsync(d.window, text + from, n, hipMemcpyDeviceToHost, s));
            if (!d.first_byte) R_HIP(hipMemcpyAsync(d.first_back, text, 1, hipMemcpyDeviceToHost, s));
            const double t0 = now();
            R_HIP(hipStreamSynchronize(s));
            d.t_wait += now() - t0;
            if (!d.first_byte) d.first_byte = d.first_back[0];
            for (size_t i = 0; i < n_members; ++i)
                if (d.status[i] != 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
            return BL_OK;
        };
        size_t win = 0, cut = 0, win_from = 0;
        bool last_batch = false;
        if (d.trim_front) {
            // a reader of a later part of the file: its text begins at the first record start that can be recognised
            size_t n = filled < ((size_t)1 << 20) ? filled : (size_t)1 << 20, at = 0;
            for (;;) {
                const int rc = fetch(0, n);
                if (rc != BL_OK) return rc;
                at = first_record_start(d.window, n, d.first_byte == '@' ? 'q' : 'a');
                if ((at != 0 && at != (size_t)-1) || n == filled) break;
                n = filled;
            }
            if (at == 0 || at == (size_t)-1) {  // none here
                if (d.eof) { d.finished = true; return BL_OK; }  // nor anywhere: every byte of this part belongs to the part before
                d.carry_n = filled;
                continue;
            }
            // the borrowed members begin in front of that record start: the next part's reader finds the same start, so no record
            // begins in this part at all
            if (d.ext_start != (size_t)-1 && d.ext_start < at) { d.finished = true; return BL_OK; }
            const size_t rest = filled - at;
            const int other = d.cur ^ 1;
            void* p = d.d_text[other];
            const int rc = device_reserve(&p, &d.text_cap[other], rest + 64, 0, s);
            d.d_text[other] = static_cast<uint8_t*>(p);
            if (rc != BL_OK) return rc;
            if (rest) R_HIP(hipMemcpyAsync(d.d_text[other], text + at, rest, hipMemcpyDeviceToDevice, s));
            R_HIP(hipStreamSynchronize(s));
            d.cur = other;
            d.carry_n = rest;
            if (d.ext_start != (size_t)-1) d.ext_start = d.ext_start > at ? d.ext_start - at : 0;
            d.trim_front = false;
            continue;  // (the next span is appended to what is left; this reader's first batch is that much longer)
        }
        if (d.ext_start != (size_t)-1) {
            // the members of this reader's own part end inside this text: its last record ends at the first record start that
            // can be recognised in what was borrowed from the next part (where that part's reader begins)
            win_from = d.ext_start > 64 ? d.ext_start - 64 : 0;
            win = filled - win_from;
            const int rc = fetch(win_from, win);
            if (rc != BL_OK) return rc;
            const size_t off = d.ext_start - win_from;
            const size_t at = first_record_start(d.window + off, win - off, d.first_byte == '@' ? 'q' : 'a');
            if (at == 0 || at == (size_t)-1) {
                if (!d.eof) {  // not in sight yet: borrow more
                    d.carry_n = filled;
                    continue;
                }
                cut = filled;  // the file ends first: everything is this part's
            } else {
                cut = d.ext_start + at;
            }
            last_batch = true;
        } else {
            // the end of the text comes back to the host: the cut is decided there
            win = filled < ((size_t)256 << 10) ? filled : (size_t)256 << 10;
            for (;;) {
                win_from = filled - win;
                const int rc = fetch(win_from, win);
                if (rc != BL_OK) return rc;
                if (d.eof) { cut = filled; break; }
                const size_t at = find_record_cut(d.window, win, win, d.first_byte == '@' ? 'q' : 'a');
                if (at >= 64 || (at > 0 && win == filled)) { cut = win_from + at; break; }  // (64: the parser wants to see the bytes in front of the cut)
                if (win == filled) break;  // no boundary in all of it: one record longer than the span
                win = win * 8 < filled ? win * 8 : filled;
            }
        }
        if (cut == 0) {  // keep everything and read on
            d.carry_n = filled;
            continue;
        }
        const size_t rest = last_batch ? 0 : filled - cut;
        const int other = d.cur ^ 1;
        if (rest) {
            void* p = d.d_text[other];
            const int rc = device_reserve(&p, &d.text_cap[other], rest + 64, 0, s);
            d.d_text[other] = static_cast<uint8_t*>(p);
            if (rc != BL_OK) return rc;
            R_HIP(hipMemcpyAsync(d.d_text[other], text + cut, rest, hipMemcpyDeviceToDevice, s));
        }
        d.cur = other;
        d.carry_n = rest;
        if (last_batch) d.finished = true;
        if (!d.eof && !last_batch) {  // the next span starts now, behind the bytes just moved
            const double t0 = now();
            const int rc = launch_packed(d, d.cur, d.carry_n);
            d.t_launch += now() - t0;
            if (rc != BL_OK) return rc;
        }
        const size_t ends_n = cut < 64 ? cut : 64;
        const char* ends = d.window + (cut - win_from) - ends_n;
        const double t0 = now();
        const int rc = bl_parse_device_text(ctx, text, cut, d.first_byte, ends, ends_n, out, n_seqs, n_bases);
        d.t_parse += now() - t0;
        ++d.n_batches;
        return rc;
    }
}

}  // namespace

int bl_reader_next_batch_device(bl_ctx* ctx, bl_reader* r, uint64_t max_text_bytes, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases)
{
    if (!ctx || !r || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_seqs) *n_seqs = 0;
    if (n_bases) *n_bases = 0;
    // BGZF goes to the device compressed (BL_HOST_INFLATE=1: through the host's inflate pool instead — for comparisons)
    if (!r->packed && !r->text && !r->records && r->source->kind()[0] == 'b' && (r->sharded || !std::getenv("BL_HOST_INFLATE"))) {
        FILE* f = std::fopen(r->path.c_str(), "rb");
        if (!f) return bl_set_error(BL_ERR_INVALID, (std::string("cannot open ") + r->path).c_str());
        r->packed.reset(new DeviceBgzf());
        r->packed->ctx = ctx;
        r->packed->trim_front = r->sharded && !r->shard_first;
        if (r->sharded) r->packed->first_byte = r->shard_first_byte;
        r->packed->spans.reset(new PackedSpans(f, max_text_bytes ? (size_t)max_text_bytes : (size_t)64 << 20, PINNED_MEMORY, r->shard_start, r->shard_end));
    }
    if (r->packed) {
        if (r->records || r->text) return bl_set_error(BL_ERR_INVALID, "records, spans and device batches cannot be mixed on one reader");
        return next_batch_packed(ctx, r, out, n_seqs, n_bases);
    }
    const char* text = nullptr;
    uint64_t n = 0;
    const int rc = next_span(r, max_text_bytes, PINNED_MEMORY, &text, &n);
    if (rc != BL_OK) return rc == 1 ? BL_OK : rc;  // end of file: *out stays NULL
    return bl_batch_from_text(ctx, text, n, out, n_seqs, n_bases);
}

int bl_reader_last_batch(bl_reader* r, const char** bases, const uint64_t** offsets, uint64_t* n_seqs)
{
    if (!r) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    static const uint64_t zero = 0;
    const HostBatch* b = r->batch;
    if (bases) *bases = b ? b->bases.data() : "";
    if (offsets) *offsets = b && !b->offsets.empty() ? b->offsets.data() : &zero;
    if (n_seqs) *n_seqs = b ? b->names.size() : 0;
    return BL_OK;
}

const char* bl_reader_last_name(bl_reader* r, uint64_t i) { return (r && r->batch && i < r->batch->names.size()) ? r->batch->names[i].c_str() : nullptr; }

}  // extern "C"
