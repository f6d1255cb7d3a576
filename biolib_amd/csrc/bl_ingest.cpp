// bl_ingest.cpp — host side of FASTA / FASTQ ingest: file -> decompressed byte stream -> records
// or raw text spans -> device batches.  Written from the formats, in three layers:
//
//   1. ByteSource: the decompressed bytes of the file as a sequence of chunks, produced by background threads
//        plain file      one read-ahead thread
//        BGZF            (bgzip: gzip members of <= 64 KiB whose header says how long they are) — the members are cut out
//                        by one thread and inflated by a pool, results delivered in file order: inflate scales with cores
//        other gzip      one inflate thread running ahead of the parser (a deflate stream has no entry points; concatenated
//                        members are followed)
//   2. RecordParser: a line-oriented state machine over the chunks (memchr for line ends, no per-byte loop) that returns
//      what the reader biolib's tools use returns — the reference's tests/kseq.h:185-234 is the behaviour to match and
//      tests/test_ingest.py + tests/ingest_fuzz.py (the reference reader as judge) are the gate:
//        * a record starts at the next '>' or '@' byte, wherever it stands; the name ends at the first whitespace byte
//        * sequence lines are joined until a line begins with '>', '@' or '+'; empty lines vanish; one '\r' at the end of
//          what has been gathered so far is dropped after each line, unless that is all there is
//        * after a '+' line, quality lines are gathered until they are at least as long as the sequence; a different
//          total length (or no quality at all) makes the record malformed
//   3. Text spans for the device-side parser (bl_batch_from_text): the decompressed text cut at record boundaries, so that
//      .gz input takes  parallel inflate -> one H2D copy -> parse on the GPU  instead of the host record loop.
// Bases are passed through untouched (the scan's own table decides what is a break).
#include <zlib.h>

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/biolib_amd.h"

extern int bl_set_error(int code, const char* msg);  // bl_capi.hip

namespace {

constexpr size_t CHUNK_BYTES = 4u << 20;  // decompressed bytes per chunk (plain / stream gzip)
constexpr int BGZF_GROUP = 48;            // BGZF members inflated per job (~3 MiB of text)

struct Chunk {
    std::vector<unsigned char> bytes;
    bool ok = true;  // false: the stream is damaged from here on
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
        ready_.wait(lk, [&] { return (!slots_.empty() && slots_.front().done) || (closed_ && slots_.empty()); });
        if (slots_.empty()) return false;
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
    InflatePool(int n, OrderedQueue& q) : q_(q)
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
        return good && (uint32_t)crc32(crc32(0L, Z_NULL, 0), out.data() + at, isize) == want_crc;
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
            q_.finish(j.out);
        }
    }
    OrderedQueue& q_;
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// The decompressed bytes of one file.
class ByteSource {
public:
    ByteSource(FILE* f, int threads) : f_(f), queue_(16)
    {
        unsigned char head[18];
        const size_t got = std::fread(head, 1, sizeof(head), f_);
        std::rewind(f_);
        const bool gz = got >= 2 && head[0] == 0x1f && head[1] == 0x8b;
        const bool bgzf = gz && got == 18 && head[2] == 8 && (head[3] & 4) && head[12] == 'B' && head[13] == 'C' && head[14] == 2 && head[15] == 0;
        if (bgzf) {
            kind_ = "bgzf";
            pool_.reset(new InflatePool(threads < 1 ? 1 : threads, queue_));
            feeder_ = std::thread([this] { cut_bgzf(); });
        } else if (gz) {
            kind_ = "gzip";
            feeder_ = std::thread([this] { inflate_stream(); });
        } else {
            kind_ = "plain";
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
    bool next(Chunk& c) { return queue_.pop(c); }
    const char* kind() const { return kind_; }

private:
    void read_plain()
    {
        for (;;) {
            auto c = queue_.reserve();
            if (!c) return;
            c->bytes.resize(CHUNK_BYTES);
            const size_t n = std::fread(c->bytes.data(), 1, CHUNK_BYTES, f_);
            c->bytes.resize(n);
            if (n < CHUNK_BYTES && std::ferror(f_)) c->ok = false;
            const bool last = n < CHUNK_BYTES;
            queue_.finish(c);
            if (last) break;
        }
        queue_.close();
    }
    void inflate_stream()
    {
        z_stream z;
        std::memset(&z, 0, sizeof(z));
        bool ok = inflateInit2(&z, 15 + 32) == Z_OK;  // gzip or zlib wrapper, detected
        std::vector<unsigned char> in(1u << 20);
        bool input_done = false, member_open = false;
        while (ok) {
            auto c = queue_.reserve();
            if (!c) break;
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
                if (z.avail_in == 0 && input_done) {  // no more input: fine between members, a truncation inside one
                    if (member_open) c->ok = false;
                    finished = true;
                    break;
                }
                member_open = true;
                const int rc = inflate(&z, Z_NO_FLUSH);
                if (rc == Z_STREAM_END) {  // end of a member: another may follow (concatenated gzip)
                    member_open = false;
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
        bool more = true;
        while (more) {
            auto c = queue_.reserve();
            if (!c) break;
            InflatePool::Job job;
            job.out = c;
            for (int b = 0; b < BGZF_GROUP; ++b) {
                unsigned char head[18];
                const size_t got = std::fread(head, 1, sizeof(head), f_);
                if (got == 0) { more = false; break; }  // clean end of file
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
            }
            if (!c->ok) job.packed.clear();
            pool_->submit(std::move(job));
        }
        queue_.close();
    }

    FILE* f_;
    OrderedQueue queue_;
    std::unique_ptr<InflatePool> pool_;
    std::thread feeder_;
    const char* kind_ = "plain";
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

// Raw text cut at record boundaries, for the device-side parser.
class TextCutter {
public:
    explicit TextCutter(ByteSource& s) : in_(s) {}
    // next span of at most `limit` bytes (more if a single record is longer) that ends at a record boundary:
    // FASTQ (first byte '@'): after every 4th line, counted from the start of the stream; FASTA: before a line-initial '>'
    // 1 = span ready, 0 = end of stream, -1 = damaged stream
    int next(size_t limit, std::string& span)
    {
        std::string buf;
        buf.swap(carry_);  // always starts at a record boundary
        size_t scanned = 0, best = 0, cut = 0;
        uint64_t lines = 0;
        bool have_best = false;
        for (;;) {
            if (!buf.empty() && !fmt_) fmt_ = buf[0] == '@' ? 'q' : 'a';
            bool decided = false;
            if (fmt_ == 'q') {
                // FASTQ: only the line COUNT matters (a record is 4 lines): count the newlines of what is new in one pass
                // (the compiler vectorises the loop), then walk back from the last one to the last multiple of four
                const size_t upto = buf.size() < limit ? buf.size() : limit;
                if (scanned < upto) {
                    const char* q = buf.data();
                    uint64_t add = 0;
                    for (size_t i = scanned; i < upto; ++i) add += q[i] == '\n';
                    lines += add;
                    scanned = upto;
                    size_t back = (size_t)(lines % 4), at = upto;  // drop `back` newlines from the end, then cut after the one before them
                    bool found = lines >= 4;
                    for (size_t drop = 0; found && drop <= back; ++drop) {
                        const void* nl = at ? memrchr(q, '\n', at) : nullptr;
                        if (!nl) { found = false; break; }
                        at = (size_t)(static_cast<const char*>(nl) - q);
                        if (drop == back) { best = at + 1; have_best = true; }
                    }
                }
                if (buf.size() > limit && !have_best) {  // a single record longer than the limit: the first boundary beyond it
                    while (scanned < buf.size()) {
                        const void* nl = std::memchr(buf.data() + scanned, '\n', buf.size() - scanned);
                        if (!nl) { scanned = buf.size(); break; }
                        scanned = (size_t)(static_cast<const char*>(nl) - buf.data()) + 1;
                        if (++lines % 4 == 0) { cut = scanned; decided = true; break; }
                    }
                } else if (buf.size() > limit) {
                    scanned = limit + 1;  // everything up to the limit has been counted: `best` stands
                }
            }
            while (fmt_ != 'q' && scanned < buf.size()) {
                const void* nl = std::memchr(buf.data() + scanned, '\n', buf.size() - scanned);
                if (!nl) {
                    scanned = buf.size();
                    break;
                }
                const size_t p = (size_t)(static_cast<const char*>(nl) - buf.data()) + 1;  // first byte of the next line
                if (p >= buf.size()) break;  // the byte that decides is not here yet: this newline is looked at again
                const bool boundary = buf[p] == '>';
                scanned = p;
                if (!boundary) continue;
                if (p <= limit) {
                    best = p;
                    have_best = true;
                } else {
                    cut = have_best ? best : p;
                    decided = true;
                    break;
                }
            }
            if (decided) break;
            if (have_best && scanned > limit) {
                cut = best;
                break;
            }
            if (!in_.more()) {
                if (in_.broken()) return -1;
                if (buf.empty()) return 0;
                cut = buf.size();  // the tail of the file
                break;
            }
            buf.append(reinterpret_cast<const char*>(in_.here()), in_.left());
            in_.skip(in_.left());
        }
        if (in_.broken()) return -1;
        carry_.assign(buf, cut, std::string::npos);
        buf.resize(cut);
        span.swap(buf);
        return 1;
    }

private:
    Cursor in_;
    std::string carry_;
    char fmt_ = 0;
};

}  // namespace

struct bl_reader {
    std::unique_ptr<ByteSource> source;
    std::unique_ptr<RecordParser> records;
    std::unique_ptr<TextCutter> text;
    Record rec;
    bool have_pending = false;  // a record was parsed but did not fit the previous batch
    // last batch
    std::string bases, span;
    std::vector<uint64_t> offsets;
    std::vector<std::string> names;
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
    r->source.reset(new ByteSource(f, threads));
    *out = r;
    return BL_OK;
}

int bl_reader_open(const char* path, bl_reader** out) { return bl_reader_open_threads(path, 0, out); }

int bl_reader_close(bl_reader* r)
{
    if (!r) return BL_OK;
    r->records.reset();
    r->text.reset();
    r->source.reset();
    delete r;
    return BL_OK;
}

const char* bl_reader_kind(bl_reader* r) { return r && r->source ? r->source->kind() : ""; }

int bl_reader_next_record(bl_reader* r, const char** name, const char** seq, uint64_t* seq_len)
{
    if (!r || !seq_len) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (r->text) return bl_set_error(BL_ERR_INVALID, "this reader is delivering text spans: records and spans cannot be mixed");
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
    *out = nullptr;
    if (r->text) return bl_set_error(BL_ERR_INVALID, "this reader is delivering text spans: records and spans cannot be mixed");
    if (!r->records) r->records.reset(new RecordParser(*r->source));
    r->bases.clear();
    r->offsets.assign(1, 0);
    r->names.clear();
    for (;;) {
        if (!r->have_pending) {
            const Step s = r->records->next(r->rec);
            if (s == Step::End) break;
            if (s != Step::Record) return step_error(s);
        }
        r->have_pending = false;
        if (!r->names.empty() && max_bases && r->bases.size() + r->rec.seq.size() > max_bases) {
            r->have_pending = true;  // keep the parsed record for the next batch
            break;
        }
        r->bases += r->rec.seq;
        r->offsets.push_back(r->bases.size());
        r->names.push_back(r->rec.name);
    }
    if (n_seqs) *n_seqs = r->names.size();
    if (n_bases) *n_bases = r->bases.size();
    if (r->names.empty()) return BL_OK;  // end of file: *out stays NULL
    return bl_batch_upload(ctx, r->bases.data(), r->bases.size(), r->offsets.data(), r->names.size(), out);
}

int bl_reader_next_text(bl_reader* r, uint64_t max_bytes, const char** text, uint64_t* n_bytes)
{
    if (!r || !text || !n_bytes) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (r->records) return bl_set_error(BL_ERR_INVALID, "this reader is delivering records: records and spans cannot be mixed");
    if (!r->text) r->text.reset(new TextCutter(*r->source));
    *text = nullptr;
    *n_bytes = 0;
    const int rc = r->text->next(max_bytes ? (size_t)max_bytes : (size_t)256 << 20, r->span);
    if (rc < 0) return bl_set_error(BL_ERR_INVALID, "error reading the (compressed) stream");
    if (rc == 0) return 1;  // end of file
    *text = r->span.data();
    *n_bytes = r->span.size();
    return BL_OK;
}

int bl_reader_next_batch_device(bl_ctx* ctx, bl_reader* r, uint64_t max_text_bytes, bl_batch** out, uint64_t* n_seqs, uint64_t* n_bases)
{
    if (!ctx || !r || !out) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_seqs) *n_seqs = 0;
    if (n_bases) *n_bases = 0;
    const char* text = nullptr;
    uint64_t n = 0;
    const int rc = bl_reader_next_text(r, max_text_bytes, &text, &n);
    if (rc != BL_OK) return rc == 1 ? BL_OK : rc;  // end of file: *out stays NULL
    return bl_batch_from_text(ctx, text, n, out, n_seqs, n_bases);
}

int bl_reader_last_batch(bl_reader* r, const char** bases, const uint64_t** offsets, uint64_t* n_seqs)
{
    if (!r) return bl_set_error(BL_ERR_INVALID, "NULL argument");
    if (bases) *bases = r->bases.data();
    if (offsets) *offsets = r->offsets.data();
    if (n_seqs) *n_seqs = r->names.size();
    return BL_OK;
}

const char* bl_reader_last_name(bl_reader* r, uint64_t i) { return (r && i < r->names.size()) ? r->names[i].c_str() : nullptr; }

}  // extern "C"
