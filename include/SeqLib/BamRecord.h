// BamRecord.h -- the subset of SeqLib::BamRecord / Cigar / CigarField that the BWAAligner path writes
// and that a caller needs to read its output.  Same names, signatures and
// semantics as /root/reference/SeqLib/BamRecord.h:49-192,202-675 and src/BamRecord.cpp:33-106,
// 255-274,646-664,861-917,960-970,1039-1054; the rest of that 70-method class (mate/pair fields,
// interval algebra, pile-up helpers) is BAM utility API outside this path.
#pragma once
#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <new>
#include <regex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>
#include "SeqLib/hts_compat.h"

namespace SeqLib {

namespace detail {
// Record memory of the BATCH path (BWAAligner::alignSequences): a chunk yields millions of records, and three mallocs per record (the shared_ptr<BamRecord>, the
// bam1_t with its control block, the data blob) plus the first touch of ~400 fresh bytes each were most of a record's cost on the host.  A slab is one large
// block (2 MB) that a builder thread fills front to back with the shells and blobs of a few thousand consecutive records; every allocation
// made in it holds one reference, and the block is freed when the last of them is gone -- so a single surviving record keeps its slab (up to 2 MB) alive.  The
// per-read entry points (alignSequence) never use slabs.
struct Slab {
    std::atomic<long> refs{1};          // 1 = the builder's own hold while it fills the slab
    size_t cap = 0, used = 0;
    static constexpr size_t kAlign = 16, kHeader = 64, kBytes = (size_t)2 << 20;
    // full-size slabs are recycled through a process-wide pool (warm pages: the first touch of fresh memory costs more than everything else a record needs);
    // at most pool_cap() of them are kept -- SEQLIB_AMD_SLAB_POOL_MB, default 2 048 MB -- and trim_pool() returns them to the system
    struct Pool { std::mutex mu; std::vector<Slab *> free_; size_t cap; Pool() { const char *e = std::getenv("SEQLIB_AMD_SLAB_POOL_MB"); cap = (size_t)(e && *e ? std::atol(e) : 2048) * ((size_t)1 << 20) / kBytes; }
 };
    static Pool &pool() { static Pool *p = new Pool; return *p; }          // (never destroyed: records of static lifetime may still release slabs while the process exits)
    static void trim_pool() { Pool &P = pool(); std::lock_guard<std::mutex> g(P.mu); for (Slab *s : P.free_) std::free(s); P.free_.clear(); }
    static Slab *make(size_t bytes)
    {
        bytes = (bytes + 4095) & ~(size_t)4095;
        void *p = nullptr;
        if (bytes == kBytes) {
            Pool &P = pool();
            std::lock_guard<std::mutex> g(P.mu);
            if (!P.free_.empty()) { p = P.free_.back(); P.free_.pop_back(); }
        }
        if (!p && (::posix_memalign(&p, 4096, bytes) != 0 || !p)) throw std::bad_alloc();
        Slab *s = new (p) Slab();
        s->cap = bytes; s->used = kHeader;
        return s;
    }
    void destroy()
    {
        const size_t c = cap;
        this->~Slab();
        if (c == kBytes) {
            Pool &P = pool();
            std::lock_guard<std::mutex> g(P.mu);
            if (P.free_.size() < P.cap) { P.free_.push_back(this); return; }
        }
        std::free(this);
    }
    size_t room() const { return cap - used; }
    void *take(size_t n) { void *p = reinterpret_cast<char *>(this) + used; used += (n + kAlign - 1) & ~(kAlign - 1); return p; }          // the caller checked room()
    void retain() { refs.fetch_add(1, std::memory_order_relaxed); }
    void release() { if (refs.fetch_sub(1, std::memory_order_acq_rel) == 1) destroy(); }
};
static_assert(sizeof(Slab) <= Slab::kHeader, "slab header");
// std::allocate_shared's allocator over the builder's current slab: one reference per allocation, returned by deallocate
template <class T> struct SlabAlloc {
    using value_type = T;
    Slab *slab;
    explicit SlabAlloc(Slab *s) : slab(s) {}
    template <class U> SlabAlloc(const SlabAlloc<U> &o) : slab(o.slab) {}
    T *allocate(size_t n) { slab->retain(); return static_cast<T *>(slab->take(n * sizeof(T))); }
    void deallocate(T *, size_t) { slab->release(); }
    template <class U> bool operator==(const SlabAlloc<U> &o) const { return slab == o.slab; }
    template <class U> bool operator!=(const SlabAlloc<U> &o) const { return slab != o.slab; }
};
// one builder thread's view: the slab it is filling, replaced when a record does not fit
struct SlabWriter {
    Slab *cur = nullptr;
    size_t hint_bytes = Slab::kBytes;          // what the builder still expects to write (sizes its last slab)
    ~SlabWriter() { if (cur) cur->release(); }
    SlabWriter() = default;
    SlabWriter(const SlabWriter &) = delete;
    SlabWriter &operator=(const SlabWriter &) = delete;
    Slab *ensure(size_t bytes)
    {
        if (cur && cur->room() >= bytes) return cur;
        if (cur) cur->release();
        cur = nullptr;
        const size_t want = std::max(bytes + Slab::kHeader, std::min(Slab::kBytes, hint_bytes + Slab::kHeader + 4096));
        cur = Slab::make(want);
        return cur;
    }
};
}  // namespace detail

struct Bam1Deleter {            // /root/reference/SeqLib/BamWalker.h:19-24
    void operator()(bam1_t *b) const { if (b) bam_destroy1(b); }
};
// A default-constructed BamRecord keeps its bam1_t and the shared_ptr control block in ONE allocation (the reference pays
// bam_init1's calloc plus the control block): the record's shared_ptr<bam1_t> aliases the member of this box.
struct Bam1Box {
    bam1_t b;
    detail::Slab *slab = nullptr;          // the slab b.data lies in (BAM_USER_OWNS_DATA set), if any: this box holds one reference
    Bam1Box() { std::memset(&b, 0, sizeof b); }
    ~Bam1Box() { if (!(b.mempolicy & BAM_USER_OWNS_DATA)) std::free(b.data); if (slab) slab->release(); }
    Bam1Box(const Bam1Box &) = delete;
    Bam1Box &operator=(const Bam1Box &) = delete;
};
inline std::shared_ptr<bam1_t> make_bam1()
{
    auto box = std::make_shared<Bam1Box>();
    return std::shared_ptr<bam1_t>(box, &box->b);
}

constexpr char BASES[16] = {' ', 'A', 'C', ' ', 'G', ' ', ' ', ' ', 'T', ' ', ' ', ' ', ' ', ' ', ' ', 'N'};

class CigarField {
    friend class Cigar;
public:
    CigarField(char opChr, uint32_t len)
    {
        static const char ops[] = BAM_CIGAR_STR;
        int op = -1;
        for (int i = 0; i < 9; ++i) if (ops[i] == opChr) op = i;
        if (op < 0) throw std::invalid_argument("Cigar type must be one of MIDSHPN=X");
        data = (len << BAM_CIGAR_SHIFT) | static_cast<uint32_t>(op);
    }
    explicit CigarField(uint32_t f) : data(f) {}
    constexpr uint32_t raw() const noexcept { return data; }
    constexpr char Type() const noexcept { return bam_cigar_opchr(data); }
    constexpr uint8_t RawType() const noexcept { return bam_cigar_op(data); }
    constexpr uint32_t Length() const noexcept { return bam_cigar_oplen(data); }
    constexpr bool ConsumesReference() const noexcept { return (bam_cigar_type(bam_cigar_op(data)) & 2) != 0; }
    constexpr bool ConsumesQuery() const noexcept { return (bam_cigar_type(bam_cigar_op(data)) & 1) != 0; }
    constexpr bool operator==(const CigarField &o) const noexcept { return data == o.data; }
    constexpr bool operator!=(const CigarField &o) const noexcept { return !(*this == o); }
    friend std::ostream &operator<<(std::ostream &out, const CigarField &c) noexcept { return out << c.Length() << c.Type(); }
private:
    uint32_t data;
};

class Cigar {
public:
    Cigar() = default;
    explicit Cigar(const std::string &cig)
    {
        static const std::regex token{R"((\d+)([MIDNSHP=X]))"};
        std::smatch m;
        auto s = cig;
        while (std::regex_search(s, m, token)) {
            add(CigarField(m[2].str()[0], static_cast<uint32_t>(std::stoi(m[1].str()))));
            s = m.suffix().str();
        }
    }
    using iterator = std::vector<CigarField>::iterator;
    using const_iterator = std::vector<CigarField>::const_iterator;
    iterator begin() noexcept { return m_data.begin(); }
    iterator end() noexcept { return m_data.end(); }
    const_iterator begin() const noexcept { return m_data.begin(); }
    const_iterator end() const noexcept { return m_data.end(); }
    const CigarField &back() const noexcept { return m_data.back(); }
    const CigarField &front() const noexcept { return m_data.front(); }
    size_t size() const noexcept { return m_data.size(); }
    void reserve(size_t n) { m_data.reserve(n); }
    CigarField &operator[](size_t i) noexcept { return m_data[i]; }
    const CigarField &operator[](size_t i) const noexcept { return m_data[i]; }
    int NumQueryConsumed() const noexcept { int t = 0; for (auto &c : m_data) if (c.ConsumesQuery()) t += (int)c.Length(); return t; }
    int NumReferenceConsumed() const noexcept { int t = 0; for (auto &c : m_data) if (c.ConsumesReference()) t += (int)c.Length(); return t; }
    void add(const CigarField &c) { m_data.push_back(c); }
    bool operator==(const Cigar &c) const noexcept { return m_data == c.m_data; }
    bool operator!=(const Cigar &c) const { return !(c == *this); }
    friend std::ostream &operator<<(std::ostream &out, const Cigar &c) noexcept { for (auto &f : c) out << f; return out; }
private:
    std::vector<CigarField> m_data;
};

class BamRecord;
typedef std::shared_ptr<BamRecord> BamRecordPtr;
typedef std::vector<BamRecordPtr> BamRecordPtrVector;

class BamRecord {
    friend class BWAAligner;
public:
    BamRecord() : b(make_bam1()) {}
    explicit BamRecord(bam1_t *raw) : b(raw, Bam1Deleter()) {}
    explicit BamRecord(std::shared_ptr<bam1_t> owned) : b(std::move(owned)) {}          // (the batch path's slab-backed records)
    BamRecord(const BamRecord &) = delete;
    BamRecord &operator=(const BamRecord &) = delete;
    BamRecord(BamRecord &&) = default;
    BamRecord &operator=(BamRecord &&) = default;

    bool isEmpty() const { return !b; }
    bool ReverseFlag() const { return b ? ((b->core.flag & BAM_FREVERSE) != 0) : false; }
    bool SecondaryFlag() const { return b ? ((b->core.flag & BAM_FSECONDARY) != 0) : false; }
    bool SupplementaryFlag() const { return b ? ((b->core.flag & BAM_FSUPPLEMENTARY) != 0) : false; }
    bool MappedFlag() const { return b ? ((b->core.flag & BAM_FUNMAP) == 0) : false; }
    int32_t Position() const { return b ? (int32_t)b->core.pos : -1; }
    int32_t PositionEnd() const { return b ? (int32_t)bam_endpos(b.get()) : -1; }
    int32_t ChrID() const { return b ? b->core.tid : -1; }
    int32_t MateChrID() const { return b ? b->core.mtid : -1; }
    int32_t MapQuality() const { return b ? b->core.qual : -1; }
    std::string Qname() const { return std::string(bam_get_qname(b)); }
    uint32_t AlignmentFlag() const { return b->core.flag; }
    int32_t Length() const { return b->core.l_qseq; }
    int32_t CigarSize() const { return b ? (int32_t)b->core.n_cigar : -1; }
    void SetID(int32_t id) { if (b) b->core.tid = id; }

    std::string Sequence() const
    {
        if (!b) return {};
        const uint8_t *s = bam_get_seq(b.get());
        std::string out;
        out.reserve((size_t)b->core.l_qseq);
        for (int i = 0; i < b->core.l_qseq; ++i) out.push_back(BASES[bam_seqi(s, i)]);
        return out;
    }
    /** the quality string, `offset` added to every value (src/BamRecord.cpp:1073-1083) */
    std::string Qualities(int offset = 33) const
    {
        if (!b) return {};
        const uint8_t *q = bam_get_qual(b.get());
        std::string out;
        out.reserve((size_t)b->core.l_qseq);
        for (int i = 0; i < b->core.l_qseq; ++i) out.push_back((char)(q[i] + offset));
        return out;
    }
    Cigar GetCigar() const
    {
        Cigar cig;
        cig.reserve(b->core.n_cigar);
        const uint8_t *raw = reinterpret_cast<const uint8_t *>(bam_get_cigar(b.get()));
        for (size_t i = 0; i < b->core.n_cigar; ++i) { uint32_t w; std::memcpy(&w, raw + i * 4, 4); cig.add(CigarField{w}); }
        return cig;
    }
    std::string CigarString() const
    {
        if (!b) return {};
        std::string out;
        const uint8_t *raw = reinterpret_cast<const uint8_t *>(bam_get_cigar(b.get()));
        for (uint32_t i = 0; i < b->core.n_cigar; ++i) {
            uint32_t w; std::memcpy(&w, raw + i * 4, 4);
            out += std::to_string(bam_cigar_oplen(w));
            out.push_back(BAM_CIGAR_STR[w & BAM_CIGAR_MASK]);
        }
        return out;
    }
    bool GetIntTag(std::string_view tag, int32_t &out) const
    {
        if (!b) return false;
        uint8_t *aux = bam_aux_get(b.get(), tag.data());
        if (!aux) return false;
        char type = (char)*aux;
        if (type != 'i' && type != 'I' && type != 'c' && type != 'C' && type != 's' && type != 'S') return false;
        out = (int32_t)bam_aux2i(aux);
        return true;
    }
    bool GetZTag(std::string_view tag, std::string &out) const
    {
        if (!b) return false;
        uint8_t *aux = bam_aux_get(b.get(), tag.data());
        if (!aux) return false;
        char *z = bam_aux2Z(aux);
        if (!z) return false;
        out = z;
        return true;
    }
    void AddIntTag(std::string_view tag, int32_t val)
    {
        if (!b) return;
        bam_aux_append(b.get(), tag.data(), 'i', sizeof(val), reinterpret_cast<const uint8_t *>(&val));
    }
    void AddZTag(std::string_view tag, std::string_view val)
    {
        if (tag.empty() || val.empty()) return;
        uint8_t *existing = bam_aux_get(b.get(), tag.data());
        if (existing) bam_aux_del(b.get(), existing);
        std::string z(val);
        bam_aux_append(b.get(), tag.data(), 'Z', static_cast<int>(z.size() + 1), reinterpret_cast<const uint8_t *>(z.c_str()));
    }
    bam1_t *raw() const { return b.get(); }

private:
    std::shared_ptr<bam1_t> b;
};

typedef std::vector<BamRecord> BamRecordVector;

}  // namespace SeqLib
