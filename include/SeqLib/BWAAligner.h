// BWAAligner.h -- drop-in for SeqLib::BWAAligner (/root/reference/SeqLib/BWAAligner.h:12-69,
// /root/reference/src/BWAAligner.cpp).  Header-only C++ over the C-ABI of libseqlib_amd.so: the
// seed-and-extend work (mem_align1 + mem_reg2aln + hit sort/filters, src/BWAAligner.cpp:104-146) runs in
// HIP kernels on the MI355X; this class keeps the option setters (:14-87), turns error codes back into the
// reference's exceptions, and materialises the BamRecords exactly as src/BWAAligner.cpp:151-248 does.
//
// Besides the reference's per-read alignSequence (one GPU round trip per call -- correct but slow, like any
// per-item GPU call) there is the batch entry the GPU needs:
//     alignSequences(const UnalignedSequenceVector&, std::vector<BamRecordPtrVector>&, hardclip, keepSecFrac, maxSecondary)
// where read i behaves exactly as the i-th successive alignSequence call, including its lrand48() draw
// (the process's real libc stream is read with seed48 and advanced by the number of reads).
// The north-star spelling AlignSequence(...) and a BamRecordVector overload are provided as aliases.
//
// Threads: as in the reference, alignSequence is const and one BWAAligner may be shared by any number of host
// threads.  The device handle is created once (std::call_once), each call reserves its lrand48() draws under a
// process-wide lock (peek + advance of the libc state is one step), and the C-ABI serves the calls of one handle
// one after another (the aligner's own lock): concurrent callers get correct results, not concurrent GPU work --
// use alignSequences for throughput.
// Limit of the GPU path, reported by an exception: reads longer than SLX_MAX_READ_LEN.
#pragma once
#include <algorithm>
#include <atomic>
#include <cassert>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#if defined(__linux__)
#include <sched.h>
#endif
#include "seqlib_amd.h"
#include "SeqLib/BWAIndex.h"
#include "SeqLib/BamRecord.h"
#include "SeqLib/UnalignedSequence.h"

namespace SeqLib {

namespace detail {
// 4-bit sequence of a record from the ASCII read (src/BWAAligner.cpp:208-220): dst[(sl + 1) / 2] bytes, high nibble first.  rev: the read backwards with the
// reference's reverse map -- A -> 8, T -> 1, C and G unchanged.  Only the exact letters A C G T are bases (lower case is 15, as in the reference's switch).
inline void pack_seq4_scalar(const uint8_t *cp, int from, int sl, bool rev, uint8_t *dst)
{
    for (int j = from; j < sl; ++j) {
        const uint8_t c = rev ? cp[sl - 1 - j] : cp[j];
        uint8_t v = 15;
        switch (c) { case 'A': v = rev ? 8 : 1; break; case 'C': v = 2; break; case 'G': v = 4; break; case 'T': v = rev ? 1 : 8; break; }
        dst[j >> 1] |= (uint8_t)(v << ((~j & 1) << 2));
    }
}
inline void pack_seq4(const uint8_t *cp, int sl, bool rev, uint8_t *dst)
{
    int done = 0;
#if defined(__SSE2__)
    const __m128i cA = _mm_set1_epi8('A'), cC = _mm_set1_epi8('C'), cG = _mm_set1_epi8('G'), cT = _mm_set1_epi8('T');
    const __m128i vA = _mm_set1_epi8(rev ? 8 : 1), vC = _mm_set1_epi8(2), vG = _mm_set1_epi8(4), vT = _mm_set1_epi8(rev ? 1 : 8), vN = _mm_set1_epi8(15);
    const __m128i lo = _mm_set1_epi16(0x00ff), zero = _mm_setzero_si128();
    auto block = [&](int at) {                               // bases [at, at + 16) of the output, at even
        __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i *>(rev ? cp + sl - 16 - at : cp + at));
        if (rev) {                                           // byte order reversed: swap inside the 16-bit words, then reverse the eight words
            c = _mm_or_si128(_mm_srli_epi16(c, 8), _mm_slli_epi16(c, 8));
            c = _mm_shufflelo_epi16(c, _MM_SHUFFLE(0, 1, 2, 3));
            c = _mm_shufflehi_epi16(c, _MM_SHUFFLE(0, 1, 2, 3));
            c = _mm_shuffle_epi32(c, _MM_SHUFFLE(1, 0, 3, 2));
        }
        __m128i v = _mm_or_si128(_mm_or_si128(_mm_and_si128(_mm_cmpeq_epi8(c, cA), vA), _mm_and_si128(_mm_cmpeq_epi8(c, cC), vC)),
                                 _mm_or_si128(_mm_and_si128(_mm_cmpeq_epi8(c, cG), vG), _mm_and_si128(_mm_cmpeq_epi8(c, cT), vT)));
        v = _mm_or_si128(v, _mm_and_si128(_mm_cmpeq_epi8(v, zero), vN));
        const __m128i w = _mm_or_si128(_mm_slli_epi16(_mm_and_si128(v, lo), 4), _mm_srli_epi16(v, 8));          // word k = base 2k << 4 | base 2k + 1
        _mm_storel_epi64(reinterpret_cast<__m128i *>(dst + (at >> 1)), _mm_packus_epi16(w, w));
    };
    for (; done + 16 <= sl; done += 16) block(done);
    if (sl >= 16 && done < sl) { const int at = (sl - 16) & ~1; block(at); done = at + 16; }          // the tail as one more block that overlaps the last (same bytes again)
#endif
    std::memset(dst + (done >> 1), 0, (size_t)(((sl + 1) >> 1) - (done >> 1)));
    pack_seq4_scalar(cp, done, sl, rev, dst);
}

// CPUs this process may actually use: hardware threads, cut down to the CPU affinity mask, to the cgroup's CPU quota and to this rank's share (a container
// that shows 256 hardware threads with a quota of 16 CPUs runs 256 busy threads at a sixteenth of their speed each).
inline unsigned effective_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
#if defined(__linux__)
    {   // the affinity mask (taskset, a launcher's binding)
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0) n = std::min(n, (unsigned)a); }
    }
#endif
    auto quota = [](const char *path, const char *path_period) -> double {
        FILE *f = std::fopen(path, "r");
        if (!f) return 0.0;
        char a[64] = {0}, b[64] = {0};
        double q = 0.0;
        const int got = std::fscanf(f, "%63s %63s", a, b);
        std::fclose(f);
        if (got >= 1 && std::strcmp(a, "max") != 0 && std::atof(a) > 0) {
            double period = got >= 2 ? std::atof(b) : 0.0;
            if (period <= 0 && path_period) { FILE *g = std::fopen(path_period, "r"); if (g) { if (std::fscanf(g, "%63s", b) == 1) period = std::atof(b); std::fclose(g); } }
            if (period > 0) q = std::atof(a) / period;
        }
        return q;
    };
    double q = quota("/sys/fs/cgroup/cpu.max", nullptr);                                         // cgroup v2: "<quota> <period>" or "max <period>"
    if (q <= 0) q = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");   // cgroup v1
    if (q > 0) n = std::min(n, (unsigned)std::max(1.0, std::ceil(q)));
    // one process per GPU on a shared node: every rank takes its share of the CPUs, not all of them (LOCAL_WORLD_SIZE is what
    // torch.distributed.run / torchrun export; SEQLIB_AMD_LOCAL_RANKS for any other launcher)
    for (const char *name : {"SEQLIB_AMD_LOCAL_RANKS", "LOCAL_WORLD_SIZE"}) {
        const char *e = std::getenv(name);
        if (e && std::atoi(e) > 1) { n = std::max(1u, n / (unsigned)std::atoi(e)); break; }
    }
    return n;
}

// Host threads of one alignSequences call: task groups run to completion in submission order within a priority class; packing
// the next chunk's reads (high) goes before building the previous chunk's records (low), so the GPU never waits for its input.
class TaskPool {
public:
    struct Group {
        std::function<void(int)> fn;
        int n = 0;
        std::atomic<int> next{0};
        int done = 0;
        std::exception_ptr err;
    };
    explicit TaskPool(unsigned n_threads)
    {
        for (unsigned t = 0; t < n_threads; ++t) th_.emplace_back([this]() { loop(); });
    }
    ~TaskPool()
    {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    std::shared_ptr<Group> submit(int n_tasks, std::function<void(int)> fn, bool high)
    {
        auto g = std::make_shared<Group>();
        g->fn = std::move(fn); g->n = n_tasks;
        if (n_tasks > 0) {
            { std::lock_guard<std::mutex> l(mu_); (high ? hi_ : lo_).push_back(g); }
            cv_.notify_all();
        }
        return g;
    }
    void wait(const std::shared_ptr<Group> &g)
    {
        std::unique_lock<std::mutex> l(mu_);
        done_cv_.wait(l, [&]() { return g->done >= g->n; });
        if (g->err) std::rethrow_exception(g->err);
    }
private:
    void loop()
    {
        std::unique_lock<std::mutex> l(mu_);
        for (;;) {
            cv_.wait(l, [&]() { return stop_ || !hi_.empty() || !lo_.empty(); });
            if (stop_) return;
            auto &q = !hi_.empty() ? hi_ : lo_;
            std::shared_ptr<Group> g = q.front();
            const int i = g->next.fetch_add(1);
            if (i >= g->n) { if (!q.empty() && q.front() == g) q.pop_front(); continue; }
            if (i == g->n - 1 && !q.empty() && q.front() == g) q.pop_front();
            l.unlock();
            std::exception_ptr e;
            try { g->fn(i); } catch (...) { e = std::current_exception(); }
            l.lock();
            if (e && !g->err) g->err = e;
            if (++g->done >= g->n) done_cv_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::deque<std::shared_ptr<Group>> hi_, lo_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    bool stop_ = false;
};
}  // namespace detail

class BWAAligner {
public:
    explicit BWAAligner(BWAIndexPtr idx) : index_(std::move(idx)) { slx_opt_init(&memopt_); }   // mem_opt_init + MEM_F_SOFTCLIP
    ~BWAAligner()
    {
        try { Flush(); } catch (...) {}          // calls still queued by alignSequenceAsync
        for (Staging &st : stage_) { slx_host_free(st.bases); slx_host_free(st.offs); }
        if (al_) slx_aligner_free(al_);
        if (al2_) slx_aligner_free(al2_);
    }
    BWAAligner(const BWAAligner &) = delete;
    BWAAligner &operator=(const BWAAligner &) = delete;

    void SetGapOpen(int gap_open)
    {
        Flush();
        if (gap_open < 0) throw std::invalid_argument{"SetGapOpen: gap_open must be >= 0"};
        memopt_.o_ins = memopt_.o_del = gap_open;
    }
    void SetGapExtension(int gap_ext)
    {
        Flush();
        if (gap_ext < 0) throw std::invalid_argument{"SetGapExtension: gap_ext must be >= 0"};
        memopt_.e_ins = memopt_.e_del = gap_ext;
    }
    void SetMismatchPenalty(int mismatch)
    {
        Flush();
        if (mismatch < 0) throw std::invalid_argument{"SetMismatchPenalty: mismatch must be >= 0"};
        memopt_.b = mismatch;
        slx_fill_scmat(memopt_.a, memopt_.b, memopt_.mat);
    }
    void SetZDropoff(int zdrop)
    {
        Flush();
        if (zdrop < 0) throw std::invalid_argument{"SetZDropoff: zdrop must be >= 0"};
        memopt_.zdrop = zdrop;
    }
    void SetAScore(int a)
    {
        Flush();   // scales the penalties but leaves the score matrix as it was (reference behaviour, src/BWAAligner.cpp:43-59)
        if (a < 0) throw std::invalid_argument{"SetAScore: a must be >= 0"};
        memopt_.a = a;
        memopt_.b *= a; memopt_.T *= a; memopt_.o_ins *= a; memopt_.o_del *= a; memopt_.e_ins *= a; memopt_.e_del *= a;
        memopt_.zdrop *= a; memopt_.pen_clip5 *= a; memopt_.pen_clip3 *= a; memopt_.pen_unpaired *= a;
    }
    void Set3primeClippingPenalty(int penalty)
    {
        Flush();
        if (penalty < 0) throw std::invalid_argument{"Set3primeClippingPenalty: penalty must be >= 0"};
        memopt_.pen_clip3 = penalty;
    }
    void Set5primeClippingPenalty(int penalty)
    {
        Flush();
        if (penalty < 0) throw std::invalid_argument{"Set5primeClippingPenalty: penalty must be >= 0"};
        memopt_.pen_clip5 = penalty;
    }
    void SetBandwidth(int bw)
    {
        Flush();
        if (bw < 0) throw std::invalid_argument{"SetBandwidth: bandwidth must be >= 0"};
        memopt_.w = bw;
    }
    void SetReseedTrigger(float trigger)
    {
        Flush();
        if (trigger < 0.0f) throw std::invalid_argument{"SetReseedTrigger: trigger must be >= 0"};
        memopt_.split_factor = trigger;
    }
    // ---- not in the reference: bwa's own output rules ---------------------------------
    // The reference's glue emits every region bwa finds, sorted by mapq, with its own secondary filters and without the score
    // threshold, the supplementary flag and the XA / SA tags `bwa mem` would give (src/BWAAligner.cpp:136-146, :240: h.XA is
    // always NULL).  With UseBwaMemRecords(true) a read yields what bwa's mem_reg2sam prints for it instead: its primaries
    // scoring >= T in bwa's order (0x800 on all but the first, mapq capped at the first's), XA:Z from mem_gen_alt through the
    // branch of :240, MD:Z, XS:i and SA:Z; a read without such a record yields one unmapped record (flag 4).  keepSecFrac and
    // maxSecondary are not used then.
    void UseBwaMemRecords(bool on = true) { Flush(); if (on) memopt_.flag |= SLX_F_REG2SAM; else memopt_.flag &= ~SLX_F_REG2SAM; }
    void SetOutputScoreThreshold(int T)
    {
        Flush();
        if (T < 0) throw std::invalid_argument{"SetOutputScoreThreshold: T must be >= 0"};
        memopt_.T = T;
    }

    // ---- the reference's entry points -------------------------------------------------------------
    void alignSequence(const std::string &seq, const std::string &name, BamRecordPtrVector &out, bool hardclip, double keepSecFrac,
                       int maxSecondary) const
    {
        if (index_->IsEmpty()) return;                       // nothing to do if no index (src/BWAAligner.cpp:101)
        Flush();                                             // queued calls came first: they keep the earlier lrand48 draws
        combine_call(seq, name, out, hardclip, keepSecFrac, maxSecondary);
    }
    void alignSequence(const UnalignedSequence &us, BamRecordPtrVector &out, bool hardclip, double keepSecFrac, int maxSecondary) const
    {
        alignSequence(us.Seq, us.Name, out, hardclip, keepSecFrac, maxSecondary);
        if (!copyComment_) return;
        for (auto &rec : out) rec->AddZTag("BC", us.Com);
    }
    // ---- batch entry (new): the whole vector in GPU-sized passes -------------------------------------
    // A large batch is cut into chunks: the GPU aligns chunk k while the host threads build the BamRecords of chunk k-1
    // and pack the bases of chunk k+1 straight into pinned staging memory (no concatenated copy of the reads).
    void alignSequences(const UnalignedSequenceVector &reads, std::vector<BamRecordPtrVector> &out, bool hardclip, double keepSecFrac,
                        int maxSecondary) const
    {
        out.clear();
        out.resize(reads.size());
        if (index_->IsEmpty() || reads.empty()) return;
        Flush();
        run_batch(reads, out, hardclip, keepSecFrac, maxSecondary);
    }
    // ---- deferred per-read calls (new): the reference's calling convention at batch speed ----------------
    // Every reference caller loops `alignSequence` over its reads (README.md:174-180, src/seqtools/seqtools.cpp:198-210).  Here one such
    // call is one GPU round trip (~1 ms: dozens of launches for one read), so a loop over 10^6 reads is slower than the CPU library.
    // alignSequenceAsync queues the call instead -- same arguments, the caller's own output vector -- and Flush() runs everything queued
    // as ONE batch: the i-th queued read is the i-th successive alignSequence call (same lrand48 draw, records appended to its vector in
    // call order).  Queued calls flush by themselves when the arguments after the name change, when 2 M reads are waiting, before any
    // synchronous alignSequence / alignSequences call and before any option setter takes effect (so a queued read is aligned with the
    // options and the draw it would have had as a plain call), and in the destructor.  The output vectors must stay alive until then.
    // If the batch throws, the queue is put back.  Not thread-safe (one queue per aligner).
    void alignSequenceAsync(const std::string &seq, const std::string &name, BamRecordPtrVector &out, bool hardclip, double keepSecFrac, int maxSecondary)
    {
        if (index_->IsEmpty()) return;
        if (!q_reads_.empty() && (hardclip != q_hardclip_ || keepSecFrac != q_ksf_ || maxSecondary != q_maxsec_)) Flush();
        q_hardclip_ = hardclip; q_ksf_ = keepSecFrac; q_maxsec_ = maxSecondary;
        q_reads_.emplace_back(name, seq);
        q_outs_.push_back(&out);
        if (q_reads_.size() >= (size_t)2 << 20) Flush();
    }
    void Flush() const
    {
        if (q_reads_.empty()) return;
        std::vector<BamRecordPtrVector> res;
        UnalignedSequenceVector reads;
        std::vector<BamRecordPtrVector *> outs;
        reads.swap(q_reads_); outs.swap(q_outs_);          // (the batch below finds the queue empty)
        try { alignSequences(reads, res, q_hardclip_, q_ksf_, q_maxsec_); }
        catch (...) { reads.swap(q_reads_); outs.swap(q_outs_); throw; }          // nothing was delivered: the calls stay queued
        for (size_t i = 0; i < res.size(); ++i)
            for (auto &r : res[i]) outs[i]->push_back(std::move(r));
    }
    size_t Pending() const { return q_reads_.size(); }
    // The GPUs this aligner drives (HIP ordinals; before the first alignment).  Default: the current device, or what the
    // environment variable SEQLIB_AMD_DEVICES names ("all", or a comma-separated list).  With several devices a batch is
    // sharded over them by contiguous read ranges inside the C-ABI (slx_aligner_create with n_dev > 1); results are identical.
    void UseDevices(const std::vector<int> &devices) { devices_ = devices; }
    // ---- north-star spellings ---------------------------------------------------------------------
    void AlignSequence(const std::string &seq, const std::string &name, BamRecordPtrVector &out, bool hardclip, double keepSecFrac,
                       int maxSecondary) const { alignSequence(seq, name, out, hardclip, keepSecFrac, maxSecondary); }
    void AlignSequence(const std::string &seq, const std::string &name, BamRecordVector &out, bool hardclip, double keepSecFrac,
                       int maxSecondary) const
    {
        BamRecordPtrVector tmp;
        alignSequence(seq, name, tmp, hardclip, keepSecFrac, maxSecondary);
        for (auto &p : tmp) out.push_back(std::move(*p));
    }

private:
    mutable UnalignedSequenceVector q_reads_;       // alignSequenceAsync's queue (mutable: the const entry points flush it first)
    mutable std::vector<BamRecordPtrVector *> q_outs_;
    bool q_hardclip_ = false; double q_ksf_ = 0; int q_maxsec_ = 0;
    BWAIndexPtr index_;
    slx_opt memopt_;
    mutable slx_aligner *al_ = nullptr;
    mutable std::once_flag al_once_;
    mutable slx_aligner *al2_ = nullptr;        // a second device handle (its own copy of the index in HBM, its own workers): the batch path keeps two chunks' calls in flight
    mutable std::once_flag al2_once_;
    bool copyComment_ = false;
    std::vector<int> devices_;
    mutable std::vector<std::string> names_;   // contig names, for the XA / SA strings
    mutable std::once_flag names_once_;
    struct Staging { char *bases = nullptr; uint64_t *offs = nullptr; size_t cap_bases = 0, cap_reads = 0; };
    mutable Staging stage_[4];                 // pinned staging of the chunked batch path (kept between calls): two slots per call in flight
    mutable std::mutex batch_mu_;              // one chunked batch at a time per aligner (they share the staging)

    static long env_long(const char *name, long dflt)
    {
        const char *v = std::getenv(name);
        return v && *v ? std::atol(v) : dflt;
    }

    slx_aligner *handle() const
    {   // created on first use, once, whichever thread gets here first (a failed creation throws and may be retried)
        std::call_once(al_once_, [this]() {
            std::vector<int> dev = devices_;
            if (dev.empty()) {
                const char *e = std::getenv("SEQLIB_AMD_DEVICES");
                if (e && *e) {
                    if (!std::strcmp(e, "all")) { for (int d = 0; d < slx_device_count(); ++d) dev.push_back(d); }
                    else for (const char *p = e; *p;) {
                        char *end = nullptr;
                        const long d = std::strtol(p, &end, 10);
                        if (end == p) throw std::invalid_argument(std::string("SEQLIB_AMD_DEVICES: cannot parse '") + e + "'");
                        dev.push_back((int)d);
                        p = end;
                        while (*p == ',' || *p == ' ') ++p;
                    }
                }
            }
            slx_aligner *al = nullptr;
            const int rc = slx_aligner_create(index_->idx_, dev.empty() ? nullptr : dev.data(), (int)dev.size(), &al);
            if (rc == SLX_ENOMEM) throw std::bad_alloc();
            if (rc != SLX_OK) throw std::runtime_error(std::string("BWAAligner: ") + slx_last_error());
            al_ = al;
            n_dev_ = dev.empty() ? 1 : (int)dev.size();
            dev_used_ = dev;
        });
        return al_;
    }
    mutable int n_dev_ = 1;
    mutable std::vector<int> dev_used_;

    slx_aligner *handle2() const
    {   // the same devices once more; a failure (no memory for a second copy of the index) leaves the batch path with one call at a time
        handle();
        std::call_once(al2_once_, [this]() {
            slx_aligner *al = nullptr;
            if (slx_aligner_create(index_->idx_, dev_used_.empty() ? nullptr : dev_used_.data(), (int)dev_used_.size(), &al) == SLX_OK) al2_ = al;
        });
        return al2_;
    }

    static std::mutex &rng_mutex() { static std::mutex m; return m; }

    static void throw_rc(int rc)
    {
        if (rc == SLX_ENOMEM) throw std::bad_alloc();
        if (rc == SLX_EINVAL) throw std::invalid_argument(slx_last_error());
        if (rc != SLX_OK) throw std::runtime_error(std::string("BWAAligner::alignSequence: ") + slx_last_error());
    }

    // record construction of src/BWAAligner.cpp:151-248 for hit k of `h`.  Same bytes as the reference builds; one allocation
    // for the data blob, sized for the tags it ends with (the reference reallocs at each bam_aux_append).  xa / sa / xs: the
    // extra tags of a UseBwaMemRecords record (XA:Z where :240 puts it, XS:i and SA:Z after AS:i).  With a slab writer (the batch
    // path) the record's two shells and its blob are carved out of the builder thread's current slab instead of three mallocs.
    static BamRecordPtr make_record(const slx_hits &h, int64_t k, const std::string_view seq, const char *name, size_t l_name, bool hardclip,
                                    const std::string *xa = nullptr, const std::string *sa = nullptr, const int32_t *xs = nullptr, const std::string *md = nullptr,
                                    detail::SlabWriter *sw = nullptr)
    {
        const uint32_t *cig = h.cigar + h.cig_off[k];
        const int n_cigar = h.n_cigar_ops[k];
        size_t tstart = 0, clen = seq.size();
        if (hardclip) {                                     // :164-177 (clips arrive as H here; bwa's op 3 there)
            clen = 0;
            for (int c = 0; c < n_cigar; ++c) {
                const uint32_t op = bam_cigar_op(cig[c]);
                if (c == 0 && op == BAM_CHARD_CLIP) tstart = bam_cigar_oplen(cig[c]);
                else if (op != BAM_CHARD_CLIP && (bam_cigar_type(op) & 1)) clen += bam_cigar_oplen(cig[c]);
            }
            assert(clen && tstart + clen <= seq.size());
        }
        const std::string_view clipped = seq.substr(tstart, clen);
        const int sl = (int)clipped.size();
        const int l_core = (int)(l_name + 1) + (n_cigar << 2) + ((sl + 1) >> 1) + sl;
        const int l_tags = 3 * 7 + (xa && !xa->empty() ? 4 + (int)xa->size() : 0) + (sa && !sa->empty() ? 4 + (int)sa->size() : 0) + (xs ? 7 : 0) +
                           (md && !md->empty() ? 4 + (int)md->size() : 0);
        BamRecordPtr b;
        bam1_t *r;
        if (sw) {
            detail::Slab *slab = sw->ensure((size_t)(l_core + l_tags) + 384);          // the blob + the two shells with their control blocks
            auto box = std::allocate_shared<Bam1Box>(detail::SlabAlloc<Bam1Box>(slab));
            r = &box->b;
            r->data = static_cast<uint8_t *>(slab->take((size_t)(l_core + l_tags)));
            slab->retain();
            box->slab = slab;
            r->mempolicy = BAM_USER_OWNS_DATA;          // never realloc'ed or freed: a record that outgrows it moves to memory of its own (hts_compat.h)
            b = std::allocate_shared<BamRecord>(detail::SlabAlloc<BamRecord>(slab), std::shared_ptr<bam1_t>(box, r));
        } else {
            b = std::make_shared<BamRecord>();
            r = b->b.get();
            r->data = static_cast<uint8_t *>(std::malloc((size_t)(l_core + l_tags)));
            if (!r->data) throw std::bad_alloc();
        }
        r->core.tid = h.rid[k];
        r->core.pos = h.pos[k];
        r->core.qual = h.mapq[k];
        r->core.flag = h.flag[k];                          // reverse (0x10) and secondary (0x100) already folded in
        r->core.n_cigar = (uint32_t)n_cigar;
        r->core.mtid = -1; r->core.mpos = -1; r->core.isize = 0;
        r->core.l_qname = (uint16_t)(l_name + 1);
        r->core.l_qseq = (int32_t)sl;
        r->m_data = (uint32_t)(l_core + l_tags);
        r->l_data = l_core;
        std::memcpy(r->data, name, l_name);
        r->data[l_name] = 0;
        std::memcpy(r->data + r->core.l_qname, cig, (size_t)n_cigar << 2);
        uint8_t *seqbuf = r->data + r->core.l_qname + (r->core.n_cigar << 2);
        // :208-220 -- 4-bit codes A 1, C 2, G 4, T 8, anything else 15; on the reverse strand the read backwards with A<->T swapped and C, G left as they
        // are (reference behaviour); the qualities: [0] = 0xff, the rest zero here (the reference leaves its malloc'ed bytes)
        detail::pack_seq4(reinterpret_cast<const uint8_t *>(clipped.data()), sl, (h.flag[k] & BAM_FREVERSE) != 0, seqbuf);
        std::memset(seqbuf + ((sl + 1) >> 1), 0, (size_t)sl);
        if (sl > 0) bam_get_qual(r)[0] = 0xff;
        b->AddIntTag("NA", h.na[k]);
        b->AddIntTag("NM", h.nm[k]);
        if (md && !md->empty()) b->AddZTag("MD", *md);       // bwa prints MD:Z right after NM:i (mem_aln2sam)
        if (xa && !xa->empty()) b->AddZTag("XA", *xa);       // `if (h.XA) b->AddZTag("XA", ...)` (:240)
        b->AddIntTag("AS", h.score[k]);
        if (xs && *xs >= 0) b->AddIntTag("XS", *xs);
        if (sa && !sa->empty()) b->AddZTag("SA", *sa);
        return b;
    }

    // what `bwa mem` prints for a read nothing of which scores >= T: flag 4, no position, the read as it came (mem_aln2sam on mem_reg2aln(0))
    static BamRecordPtr make_unmapped(const std::string_view seq, const char *name, size_t l_name)
    {
        auto b = std::make_shared<BamRecord>();
        bam1_t *r = b->b.get();
        r->core.tid = -1; r->core.pos = -1; r->core.qual = 0; r->core.flag = BAM_FUNMAP; r->core.n_cigar = 0;
        r->core.mtid = -1; r->core.mpos = -1; r->core.isize = 0;
        r->core.l_qname = (uint16_t)(l_name + 1);
        r->core.l_qseq = (int32_t)seq.size();
        const int sl = (int)seq.size();
        const int l_core = r->core.l_qname + ((sl + 1) >> 1) + sl;
        r->data = static_cast<uint8_t *>(std::calloc((size_t)(l_core + 3 * 7), 1));
        if (!r->data) throw std::bad_alloc();
        r->m_data = (uint32_t)(l_core + 3 * 7);
        r->l_data = l_core;
        std::memcpy(r->data, name, l_name);
        uint8_t *seqbuf = r->data + r->core.l_qname;
        for (int p = 0; p < sl; ++p) {
            uint8_t v = 15;
            switch (seq[(size_t)p]) { case 'A': v = 1; break; case 'C': v = 2; break; case 'G': v = 4; break; case 'T': v = 8; break; }
            seqbuf[p >> 1] |= (uint8_t)(v << ((~p & 1) << 2));
        }
        if (sl > 0) bam_get_qual(r)[0] = 0xff;
        b->AddIntTag("NA", 0);
        b->AddIntTag("AS", 0);
        b->AddIntTag("XS", 0);
        return b;
    }

    const std::string &contig_name(int rid) const
    {
        std::call_once(names_once_, [this]() { for (int i = 0; i < index_->NumSequences(); ++i) names_.push_back(index_->ChrIDToName(i)); });
        return names_[(size_t)rid];
    }

    // MD:Z of entry k as bwa_gen_cigar2 builds it next to NM: match run lengths, a mismatch as the reference base, a deletion as ^ + the
    // deleted bases; from the finished record (position, CIGAR, strand) and the forward strand of the index (slx_index_fetch)
    std::string md_string(const slx_hits &h, int64_t k, const std::string_view seq) const
    {
        const uint32_t *cig = h.cigar + h.cig_off[k];
        const int n_cigar = h.n_cigar_ops[k];
        int64_t rlen = 0;
        for (int c = 0; c < n_cigar; ++c) if (bam_cigar_type(bam_cigar_op(cig[c])) & 2) rlen += bam_cigar_oplen(cig[c]);
        std::string ref((size_t)rlen, 'N');
        if (rlen && slx_index_fetch(index_->idx_, h.rid[k], h.pos[k], rlen, &ref[0]) != SLX_OK) return std::string();
        const bool rev = (h.flag[k] & BAM_FREVERSE) != 0;
        const int len = (int)seq.size();
        auto qbase = [&](int x) -> char {          // base x of the read as the record shows it; anything but ACGT never equals a reference base
            char c = rev ? seq[(size_t)(len - 1 - x)] : seq[(size_t)x];
            c = (char)(c & 0xdf);
            if (!rev) return c;
            return c == 'A' ? 'T' : c == 'C' ? 'G' : c == 'G' ? 'C' : c == 'T' ? 'A' : c;
        };
        std::string md;
        int x = 0, u = 0;
        int64_t y = 0;
        for (int c = 0; c < n_cigar; ++c) {
            const uint32_t op = bam_cigar_op(cig[c]);
            const int l = (int)bam_cigar_oplen(cig[c]);
            if (op == BAM_CMATCH) {
                for (int i = 0; i < l; ++i) {
                    if (qbase(x + i) != ref[(size_t)(y + i)]) { md += std::to_string(u); md.push_back(ref[(size_t)(y + i)]); u = 0; }
                    else ++u;
                }
                x += l; y += l;
            } else if (op == BAM_CDEL) {
                md += std::to_string(u); md.push_back('^');
                md.append(ref, (size_t)y, (size_t)l);
                u = 0; y += l;
            } else x += l;          // insertion, soft clip, hard clip (the walk is over the whole read)
        }
        md += std::to_string(u);
        return md;
    }

    // all records of read i of a result: the glue's (src/BWAAligner.cpp:136-248), or -- UseBwaMemRecords -- bwa's own, from the entries
    // the C-ABI hands over (SLX_F_REG2SAM): records (sub >= 0) and the alternatives (xa_parent) that make up their XA:Z
    void build_read(const slx_hits &h, int64_t i, const std::string_view seq, const char *name, size_t l_name, bool hardclip, const std::string *com,
                    BamRecordPtrVector &dst, detail::SlabWriter *sw = nullptr) const
    {
        const int64_t k0 = h.hit_off[i], k1 = h.hit_off[i + 1];
        if (!h.xa_parent) {
            if (k1 > k0) dst.reserve(dst.size() + (size_t)(k1 - k0));
            for (int64_t k = k0; k < k1; ++k) {
                BamRecordPtr rec = make_record(h, k, seq, name, l_name, hardclip, nullptr, nullptr, nullptr, nullptr, sw);
                if (com) rec->AddZTag("BC", *com);
                dst.push_back(std::move(rec));              // appended: `out` is never cleared (src/BWAAligner.cpp:97-98)
            }
            return;
        }
        std::vector<int64_t> rec;                               // entries that are records (sub >= 0), in bwa's order
        for (int64_t k = k0; k < k1; ++k) if (h.sub[k] >= 0) rec.push_back(k);
        const int64_t n_rec = (int64_t)rec.size();
        if (n_rec == 0) {
            BamRecordPtr r = make_unmapped(seq, name, l_name);
            if (com) r->AddZTag("BC", *com);
            dst.push_back(std::move(r));
            return;
        }
        auto put_cigar = [&](std::string &o, int64_t k) {       // XA:Z and SA:Z show bwa's own cigar: clips are S whatever the record's clipping
            const uint32_t *cig = h.cigar + h.cig_off[k];
            for (int c = 0; c < h.n_cigar_ops[k]; ++c) {
                o += std::to_string(bam_cigar_oplen(cig[c]));
                const uint32_t op = bam_cigar_op(cig[c]);
                o.push_back(op == BAM_CHARD_CLIP ? 'S' : BAM_CIGAR_STR[op]);
            }
        };
        std::vector<std::string> xa((size_t)n_rec), sa((size_t)n_rec);
        for (int64_t k = k0; k < k1; ++k) {                      // mem_gen_alt: "chr,<strand><pos>,<CIGAR>,<NM>;" in region order
            if (h.xa_parent[k] < 0) continue;
            std::string &o = xa[(size_t)h.xa_parent[k]];
            o += contig_name(h.rid[k]); o.push_back(',');
            o.push_back((h.flag[k] & BAM_FREVERSE) ? '-' : '+'); o += std::to_string((long long)h.pos[k] + 1); o.push_back(',');
            put_cigar(o, k);
            o.push_back(','); o += std::to_string(h.nm[k]); o.push_back(';');
        }
        if (n_rec > 1)                                        // mem_aln2sam: "chr,<pos>,<strand>,<CIGAR>,<mapq>,<NM>;" of every other record
            for (int64_t j = 0; j < n_rec; ++j)
                for (int64_t q = 0; q < n_rec; ++q) {
                    if (q == j) continue;
                    const int64_t k = rec[(size_t)q];
                    std::string &o = sa[(size_t)j];
                    o += contig_name(h.rid[k]); o.push_back(','); o += std::to_string((long long)h.pos[k] + 1); o.push_back(',');
                    o.push_back((h.flag[k] & BAM_FREVERSE) ? '-' : '+'); o.push_back(',');
                    put_cigar(o, k);
                    o.push_back(','); o += std::to_string((int)h.mapq[k]); o.push_back(','); o += std::to_string(h.nm[k]); o.push_back(';');
                }
        dst.reserve(dst.size() + (size_t)n_rec);
        for (int64_t j = 0; j < n_rec; ++j) {
            const std::string md = md_string(h, rec[(size_t)j], seq);
            BamRecordPtr r = make_record(h, rec[(size_t)j], seq, name, l_name, hardclip, &xa[(size_t)j], &sa[(size_t)j], &h.sub[rec[(size_t)j]], &md, sw);
            if (com) r->AddZTag("BC", *com);
            dst.push_back(std::move(r));
        }
    }

    // BamRecords of reads [a, b) of a result whose read 0 is read `base` of the call
    void materialise(const slx_hits &h, int64_t a, int64_t b, const char *bases, const uint64_t *offs, const char *const *names,
                     const UnalignedSequenceVector *reads, int64_t base, bool hardclip, BamRecordPtrVector *single_out,
                     std::vector<BamRecordPtrVector> *batch_out) const
    {
        for (int64_t i = a; i < b; ++i) {
            const std::string_view seq(bases + offs[i], (size_t)(offs[i + 1] - offs[i]));
            BamRecordPtrVector &dst = single_out ? *single_out : (*batch_out)[(size_t)(base + i)];
            const char *nm = reads ? (*reads)[(size_t)(base + i)].Name.c_str() : names[i];
            const size_t l_name = reads ? (*reads)[(size_t)(base + i)].Name.size() : std::strlen(nm);
            build_read(h, i, seq, nm, l_name, hardclip, (reads && copyComment_) ? &(*reads)[(size_t)(base + i)].Com : nullptr, dst);
        }
    }

    // ---- concurrent callers of the per-read entry: flat combining ----------------------------------------
    // The reference's alignSequence is const, lock-free and re-entrant (SeqLib/BWAAligner.h:51-63): T threads that share one aligner each align
    // their own read at the same time.  Here a call is a GPU round trip of ~450 us whatever it holds, and the device handle serves one call at
    // a time -- so T callers would get one round trip each, one after another.  Instead the first caller to arrive becomes the LEADER of a round:
    // it takes every call that is waiting with the same glue arguments (its own included), reserves their lrand48 draws in one step in ARRIVAL
    // order (the order in which the reference's threads would have drawn from the process-global stream), aligns them as ONE batch and builds
    // every caller's records into that caller's vector; callers that arrive while a round runs wait and are served by the next round, led by
    // one of them.  A lone caller is a leader with a round of one: the single-thread behaviour (records, draws, exceptions) is unchanged.
    struct CombReq {
        const std::string *seq, *name; BamRecordPtrVector *out; bool hardclip; double ksf; int maxsec;
        bool done = false; std::exception_ptr err;
    };
    struct Combiner { std::mutex mu; std::condition_variable cv, cv_leader; std::vector<CombReq *> waiting; bool leader = false; uint64_t rounds = 0, calls = 0; size_t last_round = 1; };
    mutable Combiner comb_;
public:
    // rounds led and calls served by them since the aligner was created (calls / rounds = how many concurrent callers shared a GPU round trip)
    void CombinedCallStats(uint64_t &rounds, uint64_t &calls) const { std::lock_guard<std::mutex> g(comb_.mu); rounds = comb_.rounds; calls = comb_.calls; }
private:

    void combine_call(const std::string &seq, const std::string &name, BamRecordPtrVector &out, bool hardclip, double ksf, int maxsec) const
    {
        CombReq me{&seq, &name, &out, hardclip, ksf, maxsec};
        std::unique_lock<std::mutex> lk(comb_.mu);
        comb_.waiting.push_back(&me);
        if (comb_.leader) comb_.cv_leader.notify_one();          // (a leader may be holding its round open for the callers of the last one)
        for (;;) {
            if (me.done) break;
            if (comb_.leader) { comb_.cv.wait(lk); continue; }
            // lead one round: the waiting calls whose arguments equal the first one's, in arrival order (the others wait for the next round)
            comb_.leader = true;
            // The callers of a round are released together and come back together: the first one back would lead a round of one while the others queue up behind
            // it.  When the last round served several callers the leader holds its round open until as many are waiting again -- for at most a tenth of what a
            // GPU round trip costs.  A lone caller never waits.
            if (comb_.last_round > 1 && comb_.waiting.size() < comb_.last_round)
                comb_.cv_leader.wait_for(lk, std::chrono::microseconds(env_long("SEQLIB_AMD_COMBINE_WAIT_US", 40)), [&]() { return comb_.waiting.size() >= comb_.last_round; });
            std::vector<CombReq *> round, rest;
            const CombReq &f = *comb_.waiting.front();
            for (CombReq *r : comb_.waiting) (r->hardclip == f.hardclip && r->ksf == f.ksf && r->maxsec == f.maxsec ? round : rest).push_back(r);
            comb_.waiting.swap(rest);
            lk.unlock();
            std::exception_ptr err;
            try { run_round(round); } catch (...) { err = std::current_exception(); }
            lk.lock();
            for (CombReq *r : round) { r->err = err; r->done = true; }
            ++comb_.rounds; comb_.calls += round.size();
            comb_.last_round = round.size();
            comb_.leader = false;
            comb_.cv.notify_all();
        }
        lk.unlock();
        if (me.err) std::rethrow_exception(me.err);
    }

    void run_round(const std::vector<CombReq *> &round) const
    {
        const int64_t n = (int64_t)round.size();
        if (n == 1) {                                        // the lone caller: no copy of the read
            const uint64_t offs[2] = {0, (uint64_t)round[0]->seq->size()};
            const char *names[1] = {round[0]->name->c_str()};
            run(round[0]->seq->data(), offs, 1, names, nullptr, round[0]->hardclip, round[0]->ksf, round[0]->maxsec, round[0]->out, nullptr);
            return;
        }
        std::string bases;
        std::vector<uint64_t> offs((size_t)n + 1, 0);
        for (int64_t i = 0; i < n; ++i) { bases += *round[(size_t)i]->seq; offs[(size_t)i + 1] = bases.size(); }
        slx_aligner *al = handle();
        uint64_t state;
        {
            std::lock_guard<std::mutex> g(rng_mutex());
            state = slx_lrand48_peek_libc();
            slx_lrand48_skip_libc((uint64_t)n);
        }
        slx_hits h;
        throw_rc(slx_align_batch(al, &memopt_, bases.data(), offs.data(), n, state, 0, round[0]->hardclip ? 1 : 0, round[0]->ksf, round[0]->maxsec, &h));
        try {
            for (int64_t i = 0; i < n; ++i) {
                const CombReq &r = *round[(size_t)i];
                build_read(h, i, std::string_view(bases.data() + offs[(size_t)i], r.seq->size()), r.name->c_str(), r.name->size(), r.hardclip, nullptr, *r.out);
            }
        } catch (...) { slx_hits_free(&h); throw; }
        slx_hits_free(&h);
    }

    // one read (the reference's calling convention): one GPU round trip, records built on the calling thread
    void run(const char *bases, const uint64_t *offs, int64_t n, const char *const *names, const UnalignedSequenceVector *reads, bool hardclip,
             double keepSecFrac, int maxSecondary, BamRecordPtrVector *single_out, std::vector<BamRecordPtrVector> *batch_out) const
    {
        // the reference's mem_align1 draws lrand48() once per call from the process-global libc stream: this call reserves
        // its n draws in one step, so that concurrent callers never share an ordinal
        slx_aligner *al = handle();
        uint64_t state;
        {
            std::lock_guard<std::mutex> g(rng_mutex());
            state = slx_lrand48_peek_libc();
            slx_lrand48_skip_libc((uint64_t)n);
        }
        slx_hits h;
        throw_rc(slx_align_batch(al, &memopt_, bases, offs, n, state, 0, hardclip ? 1 : 0, keepSecFrac, maxSecondary, &h));
        try { materialise(h, 0, n, bases, offs, names, reads, 0, hardclip, single_out, batch_out); } catch (...) { slx_hits_free(&h); throw; }
        slx_hits_free(&h);
    }

    // the batch: chunks of the read vector through pack (host threads) -> align (GPU) -> records (host threads), overlapped.
    // Tunables (environment): SEQLIB_AMD_THREADS host threads (default: the CPUs the process may use -- hardware threads cut down to the
    // cgroup CPU quota), SEQLIB_AMD_CHUNK reads per chunk and device (default 8 M), SEQLIB_AMD_TRACE=1 per-chunk timings on stderr.
    void run_batch(const UnalignedSequenceVector &reads, std::vector<BamRecordPtrVector> &out, bool hardclip, double keepSecFrac, int maxSecondary) const
    {
        slx_aligner *al = handle();
        const int64_t n = (int64_t)reads.size();
        uint64_t state;
        {
            std::lock_guard<std::mutex> g(rng_mutex());
            state = slx_lrand48_peek_libc();
            slx_lrand48_skip_libc((uint64_t)n);
        }
        std::lock_guard<std::mutex> batch(batch_mu_);
        const int64_t chunk = std::max<int64_t>(1024, env_long("SEQLIB_AMD_CHUNK", 8000000) * n_dev_);
        const int64_t n_chunks = (n + chunk - 1) / chunk;
        // host threads: the CPUs this process may use -- but not fewer than 16 threads when there are fewer than 8 CPUs: measured with the whole process confined to two CPUs
        // (taskset), 2 / 4 / 16 threads build 9.9 / 11.1 / 14.4 M reads/s (the pool's threads also block -- page faults of fresh record memory, hand-overs -- and the
        // GPU workers' waiting threads take their share of two CPUs whatever the pool does; 256 threads on a 16-CPU quota, on the other hand, ran at 6.3 M against 14.8 M)
        const unsigned cpus = detail::effective_cpus();
        unsigned T = (unsigned)env_long("SEQLIB_AMD_THREADS", (long)(cpus < 8 ? std::min(16u, cpus * 8) : cpus));
        T = std::max(1u, std::min(T, 512u));
        if (n < 8192) T = 1;
        const bool use_slabs = n >= env_long("SEQLIB_AMD_SLAB_MIN_READS", 8192) && env_long("SEQLIB_AMD_SLABS", 1) != 0;
        detail::TaskPool pool(T);
        // tasks per stage of a chunk: at least 64 however few threads there are -- a pack request (high priority) waits for the build tasks that are RUNNING, and with two
        // threads and eight tasks per chunk a build task was a million reads (~120 ms): the next GPU call started that much late (2 threads: 10 -> see profiles/NOTES_r06.md)
        const int parts = (int)std::min<int64_t>(std::max<int64_t>((int64_t)T * 4, 64), std::max<int64_t>(1, chunk / 2048));

        struct ChunkJob { int64_t lo = 0, hi = 0; std::vector<uint64_t> part_bytes; std::shared_ptr<detail::TaskPool::Group> packed, built; slx_hits h; bool have_h = false; };
        std::vector<ChunkJob> jobs((size_t)n_chunks);
        for (int64_t c = 0; c < n_chunks; ++c) { jobs[(size_t)c].lo = c * chunk; jobs[(size_t)c].hi = std::min(n, (c + 1) * chunk); }

        // Two chunks' GPU calls in flight (SEQLIB_AMD_CALLS_IN_FLIGHT, default 2; needs a second device handle = a second copy of the index in HBM): a call ends in the
        // single-read critical paths of its heaviest reads -- tens of milliseconds whatever the chunk holds -- and begins with its upload; with one call at a time a
        // 4 M-read chunk took 100 ms where the kernels need 65.  Lane L (a host thread) takes chunks L, L + lanes, ...: packs them, calls its own handle, hands the result
        // to the record builders.  Read i keeps draw `state + i` whichever lane aligns it.
        slx_aligner *als[2] = {al, nullptr};
        int n_lanes = 1;
        // (not with fewer than eight host threads: a thread inside a GPU call spins on the stream, and two of them take the builders' CPUs -- at 2 threads 9.6 M reads/s
        // against 13.4 M with one call at a time)
        if (n_chunks >= 2 && env_long("SEQLIB_AMD_CALLS_IN_FLIGHT", (env_long("SEQLIB_AMD_THREADS", 0) > 0 ? T : cpus) >= 8 ? 2 : 1) >= 2 && (als[1] = handle2()) != nullptr) n_lanes = 2;
        const int n_slot = 2 * n_lanes;
        // pack chunk c into staging slot c % n_slot: lengths per part -> exclusive scan -> offsets + bases, all on the pool
        auto submit_pack = [&](int64_t c) {
            ChunkJob &J = jobs[(size_t)c];
            Staging &S = stage_[c % n_slot];
            const int64_t m = J.hi - J.lo;
            if ((size_t)m + 1 > S.cap_reads) {
                slx_host_free(S.offs);
                S.cap_reads = 0;
                S.offs = static_cast<uint64_t *>(slx_host_alloc(((size_t)m + 1) * 8));
                if (!S.offs) throw std::bad_alloc();
                S.cap_reads = (size_t)m + 1;
            }
            J.part_bytes.assign((size_t)parts + 1, 0);
            auto sized = pool.submit(parts, [&, c](int t) {
                const ChunkJob &Jc = jobs[(size_t)c];
                const int64_t mm = Jc.hi - Jc.lo, a = Jc.lo + mm * t / parts, b = Jc.lo + mm * (t + 1) / parts;
                uint64_t tot = 0;
                for (int64_t i = a; i < b; ++i) tot += reads[(size_t)i].Seq.size();
                jobs[(size_t)c].part_bytes[(size_t)t + 1] = tot;
            }, true);
            pool.wait(sized);
            for (int t = 0; t < parts; ++t) J.part_bytes[(size_t)t + 1] += J.part_bytes[(size_t)t];
            const uint64_t total = J.part_bytes[(size_t)parts];
            if (total + 64 > S.cap_bases) {
                slx_host_free(S.bases);
                S.cap_bases = 0;
                const size_t want = (size_t)(total + total / 8 + 4096);
                S.bases = static_cast<char *>(slx_host_alloc(want));
                if (!S.bases) throw std::bad_alloc();
                S.cap_bases = want;
            }
            J.packed = pool.submit(parts, [&, c](int t) {
                const ChunkJob &Jc = jobs[(size_t)c];
                Staging &Sc = stage_[c % n_slot];
                const int64_t mm = Jc.hi - Jc.lo, a = Jc.lo + mm * t / parts, b = Jc.lo + mm * (t + 1) / parts;
                uint64_t o = Jc.part_bytes[(size_t)t];
                for (int64_t i = a; i < b; ++i) {
                    const std::string &sq = reads[(size_t)i].Seq;
                    Sc.offs[i - Jc.lo] = o;
                    std::memcpy(Sc.bases + o, sq.data(), sq.size());
                    o += sq.size();
                }
                if (b == Jc.hi) Sc.offs[mm] = o;
            }, true);
        };

        auto cleanup = [&]() {
            for (ChunkJob &J : jobs) {
                if (J.built) { try { pool.wait(J.built); } catch (...) {} }
                if (J.packed) { try { pool.wait(J.packed); } catch (...) {} }
                if (J.have_h) { slx_hits_free(&J.h); J.have_h = false; }
            }
        };
        const bool trace = env_long("SEQLIB_AMD_TRACE", 0) != 0;
        const auto t_begin = std::chrono::steady_clock::now();
        auto ms_since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        std::atomic<bool> failed{false};
        auto lane = [&](int L) {
            // this lane's chunks: c = L, L + n_lanes, ...; chunk c's staging slot was chunk c - n_slot's, a chunk of this lane whose call has returned
            if (L < n_chunks) submit_pack(L);
            if (L + n_lanes < n_chunks) submit_pack(L + n_lanes);
            for (int64_t c = L; c < n_chunks && !failed.load(); c += n_lanes) {
                ChunkJob &J = jobs[(size_t)c];
                const auto t_c = std::chrono::steady_clock::now();
                pool.wait(J.packed);
                const double ms_packwait = ms_since(t_c);
                // at most two result blocks per lane are alive: the records of this lane's chunk before last are done before the next call is made
                if (c >= 2 * n_lanes) { ChunkJob &P = jobs[(size_t)(c - 2 * n_lanes)]; pool.wait(P.built); slx_hits_free(&P.h); P.have_h = false; }
                Staging &S = stage_[c % n_slot];
                const double ms_pre = ms_since(t_c);
                throw_rc(slx_align_batch(als[L], &memopt_, S.bases, S.offs, J.hi - J.lo, state, (uint64_t)J.lo, hardclip ? 1 : 0, keepSecFrac, maxSecondary, &J.h));
                J.have_h = true;
                if (c + (int64_t)n_slot < n_chunks) submit_pack(c + n_slot);          // (the slot is free again: the call has read it)
                if (trace) std::fprintf(stderr, "[alignSequences] chunk %lld (%lld reads, lane %d): waited %.1f ms for its pack, %.1f ms until the GPU call, GPU call %.1f ms, t = %.1f ms, %u host threads\n",
                                        (long long)c, (long long)(J.hi - J.lo), L, ms_packwait, ms_pre, ms_since(t_c) - ms_pre, ms_since(t_begin), T);
                J.built = pool.submit(parts, [&, c](int t) {
                    const ChunkJob &Jc = jobs[(size_t)c];
                    const int64_t mm = Jc.hi - Jc.lo;
                    // the records of this stretch of reads come out of slabs this task fills (BamRecord.h, detail::Slab): no malloc per record.  SEQLIB_AMD_SLABS=0
                    // keeps every record in allocations of its own.
                    detail::SlabWriter writer;
                    detail::SlabWriter *sw = use_slabs ? &writer : nullptr;
                    const int64_t i0 = mm * t / parts, i1 = mm * (t + 1) / parts;
                    // the sequence of a record comes from the caller's reads (the staging slot is reused by a later chunk)
                    for (int64_t i = i0; i < i1; ++i) {
                        const UnalignedSequence &us = reads[(size_t)(Jc.lo + i)];
                        if (sw && !(i & 255)) {               // what is left of the stretch sizes the task's last slab: records still to come x (shells + blob of a read like this one)
                            const int64_t k_left = Jc.h.hit_off[i1] - Jc.h.hit_off[i];
                            writer.hint_bytes = (size_t)k_left * (us.Seq.size() + (us.Seq.size() >> 1) + us.Name.size() + 256);
                        }
                        build_read(Jc.h, i, us.Seq, us.Name.c_str(), us.Name.size(), hardclip, copyComment_ ? &us.Com : nullptr, out[(size_t)(Jc.lo + i)], sw);
                    }
                }, false);
            }
        };
        try {
            std::exception_ptr err2;
            std::thread second;
            if (n_lanes > 1) second = std::thread([&]() { try { lane(1); } catch (...) { err2 = std::current_exception(); failed.store(true); } });
            try { lane(0); } catch (...) { failed.store(true); if (second.joinable()) second.join(); throw; }
            if (second.joinable()) second.join();
            if (err2) std::rethrow_exception(err2);
            const auto t_tail = std::chrono::steady_clock::now();
            for (ChunkJob &J : jobs) if (J.built) { pool.wait(J.built); if (J.have_h) { slx_hits_free(&J.h); J.have_h = false; } }
            if (trace) std::fprintf(stderr, "[alignSequences] records of the last chunks: %.1f ms after the last GPU call; total %.1f ms\n", ms_since(t_tail), ms_since(t_begin));
        } catch (...) { cleanup(); throw; }
    }
};

}  // namespace SeqLib
