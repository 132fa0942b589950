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
// Limits of the GPU path, both reported by exceptions: reads longer than SLX_MAX_READ_LEN; indexes with ALT contigs.
#pragma once
#include <algorithm>
#include <cassert>
#include <exception>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>
#include "seqlib_amd.h"
#include "SeqLib/BWAIndex.h"
#include "SeqLib/BamRecord.h"
#include "SeqLib/UnalignedSequence.h"

namespace SeqLib {

class BWAAligner {
public:
    explicit BWAAligner(BWAIndexPtr idx) : index_(std::move(idx)) { slx_opt_init(&memopt_); }   // mem_opt_init + MEM_F_SOFTCLIP
    ~BWAAligner() { if (al_) slx_aligner_free(al_); }
    BWAAligner(const BWAAligner &) = delete;
    BWAAligner &operator=(const BWAAligner &) = delete;

    void SetGapOpen(int gap_open)
    {
        if (gap_open < 0) throw std::invalid_argument{"SetGapOpen: gap_open must be >= 0"};
        memopt_.o_ins = memopt_.o_del = gap_open;
    }
    void SetGapExtension(int gap_ext)
    {
        if (gap_ext < 0) throw std::invalid_argument{"SetGapExtension: gap_ext must be >= 0"};
        memopt_.e_ins = memopt_.e_del = gap_ext;
    }
    void SetMismatchPenalty(int mismatch)
    {
        if (mismatch < 0) throw std::invalid_argument{"SetMismatchPenalty: mismatch must be >= 0"};
        memopt_.b = mismatch;
        slx_fill_scmat(memopt_.a, memopt_.b, memopt_.mat);
    }
    void SetZDropoff(int zdrop)
    {
        if (zdrop < 0) throw std::invalid_argument{"SetZDropoff: zdrop must be >= 0"};
        memopt_.zdrop = zdrop;
    }
    void SetAScore(int a)
    {   // scales the penalties but leaves the score matrix as it was (reference behaviour, src/BWAAligner.cpp:43-59)
        if (a < 0) throw std::invalid_argument{"SetAScore: a must be >= 0"};
        memopt_.a = a;
        memopt_.b *= a; memopt_.T *= a; memopt_.o_ins *= a; memopt_.o_del *= a; memopt_.e_ins *= a; memopt_.e_del *= a;
        memopt_.zdrop *= a; memopt_.pen_clip5 *= a; memopt_.pen_clip3 *= a; memopt_.pen_unpaired *= a;
    }
    void Set3primeClippingPenalty(int penalty)
    {
        if (penalty < 0) throw std::invalid_argument{"Set3primeClippingPenalty: penalty must be >= 0"};
        memopt_.pen_clip3 = penalty;
    }
    void Set5primeClippingPenalty(int penalty)
    {
        if (penalty < 0) throw std::invalid_argument{"Set5primeClippingPenalty: penalty must be >= 0"};
        memopt_.pen_clip5 = penalty;
    }
    void SetBandwidth(int bw)
    {
        if (bw < 0) throw std::invalid_argument{"SetBandwidth: bandwidth must be >= 0"};
        memopt_.w = bw;
    }
    void SetReseedTrigger(float trigger)
    {
        if (trigger < 0.0f) throw std::invalid_argument{"SetReseedTrigger: trigger must be >= 0"};
        memopt_.split_factor = trigger;
    }

    // ---- the reference's entry points -------------------------------------------------------------
    void alignSequence(const std::string &seq, const std::string &name, BamRecordPtrVector &out, bool hardclip, double keepSecFrac,
                       int maxSecondary) const
    {
        if (index_->IsEmpty()) return;                       // nothing to do if no index (src/BWAAligner.cpp:101)
        const uint64_t offs[2] = {0, (uint64_t)seq.size()};
        const char *names[1] = {name.c_str()};
        run(seq.data(), offs, 1, names, nullptr, hardclip, keepSecFrac, maxSecondary, &out, nullptr);
    }
    void alignSequence(const UnalignedSequence &us, BamRecordPtrVector &out, bool hardclip, double keepSecFrac, int maxSecondary) const
    {
        alignSequence(us.Seq, us.Name, out, hardclip, keepSecFrac, maxSecondary);
        if (!copyComment_) return;
        for (auto &rec : out) rec->AddZTag("BC", us.Com);
    }
    // ---- batch entry (new): one GPU pass for the whole vector ---------------------------------------
    void alignSequences(const UnalignedSequenceVector &reads, std::vector<BamRecordPtrVector> &out, bool hardclip, double keepSecFrac,
                        int maxSecondary) const
    {
        out.clear();
        out.resize(reads.size());
        if (index_->IsEmpty() || reads.empty()) return;
        std::string bases;
        std::vector<uint64_t> offs(reads.size() + 1, 0);
        std::vector<const char *> names(reads.size());
        size_t tot = 0;
        for (auto &r : reads) tot += r.Seq.size();
        bases.reserve(tot);
        for (size_t i = 0; i < reads.size(); ++i) { bases += reads[i].Seq; offs[i + 1] = bases.size(); names[i] = reads[i].Name.c_str(); }
        run(bases.data(), offs.data(), (int64_t)reads.size(), names.data(), &reads, hardclip, keepSecFrac, maxSecondary, nullptr, &out);
    }
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
    BWAIndexPtr index_;
    slx_opt memopt_;
    mutable slx_aligner *al_ = nullptr;
    mutable std::once_flag al_once_;
    bool copyComment_ = false;

    slx_aligner *handle() const
    {   // created on first use, once, whichever thread gets here first (a failed creation throws and may be retried)
        std::call_once(al_once_, [this]() {
            slx_aligner *al = nullptr;
            const int rc = slx_aligner_create(index_->idx_, nullptr, 0, &al);
            if (rc == SLX_ENOMEM) throw std::bad_alloc();
            if (rc != SLX_OK) throw std::runtime_error(std::string("BWAAligner: ") + slx_last_error());
            al_ = al;
        });
        return al_;
    }

    static std::mutex &rng_mutex() { static std::mutex m; return m; }

    // record construction of src/BWAAligner.cpp:151-248 for hit k of `h`
    static BamRecordPtr make_record(const slx_hits &h, int64_t k, const std::string_view seq, const char *name, bool hardclip)
    {
        auto b = std::make_shared<BamRecord>();
        bam1_t *r = b->b.get();
        const uint32_t *cig = h.cigar + h.cig_off[k];
        const int n_cigar = h.n_cigar_ops[k];
        r->core.tid = h.rid[k];
        r->core.pos = h.pos[k];
        r->core.qual = h.mapq[k];
        r->core.flag = h.flag[k];                          // reverse (0x10) and secondary (0x100) already folded in
        r->core.n_cigar = (uint32_t)n_cigar;
        r->core.mtid = -1; r->core.mpos = -1; r->core.isize = 0;
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
        const size_t l_name = std::strlen(name);
        r->core.l_qname = (uint16_t)(l_name + 1);
        r->core.l_qseq = (int32_t)clipped.size();
        r->l_data = r->core.l_qname + (n_cigar << 2) + ((r->core.l_qseq + 1) >> 1) + r->core.l_qseq;
        r->data = static_cast<uint8_t *>(std::calloc((size_t)r->l_data ? (size_t)r->l_data : 1, 1));   // reference: malloc; quals past [0] are zero here
        if (!r->data) throw std::bad_alloc();
        r->m_data = 0;                                      // as the reference leaves it: the first tag append reallocs
        std::memcpy(r->data, name, l_name + 1);
        std::memcpy(r->data + r->core.l_qname, cig, (size_t)n_cigar << 2);
        uint8_t *seqbuf = r->data + r->core.l_qname + (r->core.n_cigar << 2);
        const int sl = (int)clipped.size();
        if (h.flag[k] & BAM_FREVERSE) {                     // :208-220 -- A<->T swapped, C and G left as they are (reference behaviour)
            int j = 0;
            for (int p = sl - 1; p >= 0; --p, ++j) {
                uint8_t v = 15;
                switch (clipped[(size_t)p]) { case 'A': v = 8; break; case 'C': v = 2; break; case 'G': v = 4; break; case 'T': v = 1; break; }
                seqbuf[j >> 1] &= (uint8_t)~(0xF << ((~j & 1) << 2));
                seqbuf[j >> 1] |= (uint8_t)(v << ((~j & 1) << 2));
            }
        } else {
            for (int p = 0; p < sl; ++p) {
                uint8_t v = 15;
                switch (clipped[(size_t)p]) { case 'A': v = 1; break; case 'C': v = 2; break; case 'G': v = 4; break; case 'T': v = 8; break; }
                seqbuf[p >> 1] &= (uint8_t)~(0xF << ((~p & 1) << 2));
                seqbuf[p >> 1] |= (uint8_t)(v << ((~p & 1) << 2));
            }
        }
        if (sl > 0) bam_get_qual(r)[0] = 0xff;
        b->AddIntTag("NA", h.na[k]);
        b->AddIntTag("NM", h.nm[k]);
        b->AddIntTag("AS", h.score[k]);
        return b;
    }

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
        const int rc = slx_align_batch(al, &memopt_, bases, offs, n, state, 0, hardclip ? 1 : 0, keepSecFrac, maxSecondary, &h);
        if (rc == SLX_ENOMEM) throw std::bad_alloc();
        if (rc == SLX_EINVAL) throw std::invalid_argument(slx_last_error());
        if (rc != SLX_OK) throw std::runtime_error(std::string("BWAAligner::alignSequence: ") + slx_last_error());
        // record construction (src/BWAAligner.cpp:151-248): reads [a, b) -- every read owns its own output vector, so a large
        // batch is materialised by several host threads (the step is malloc- and shared_ptr-bound, SURVEY hard part 7)
        auto materialise = [&](int64_t a, int64_t b) {
            for (int64_t i = a; i < b; ++i) {
                const std::string_view seq(bases + offs[i], (size_t)(offs[i + 1] - offs[i]));
                BamRecordPtrVector &dst = single_out ? *single_out : (*batch_out)[(size_t)i];
                for (int64_t k = h.hit_off[i]; k < h.hit_off[i + 1]; ++k) {
                    BamRecordPtr rec = make_record(h, k, seq, names[i], hardclip);
                    if (reads && copyComment_) rec->AddZTag("BC", (*reads)[(size_t)i].Com);
                    dst.push_back(rec);                     // appended: `out` is never cleared (src/BWAAligner.cpp:97-98)
                }
            }
        };
        try {
            unsigned T = (single_out || n < 8192) ? 1u : std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
            if (T <= 1) materialise(0, n);
            else {
                std::vector<std::thread> th;
                std::vector<std::exception_ptr> err(T);
                for (unsigned t = 0; t < T; ++t)
                    th.emplace_back([&, t]() {
                        try { materialise(n * (int64_t)t / (int64_t)T, n * (int64_t)(t + 1) / (int64_t)T); } catch (...) { err[t] = std::current_exception(); }
                    });
                for (auto &x : th) x.join();
                for (auto &e : err) if (e) std::rethrow_exception(e);
            }
        } catch (...) { slx_hits_free(&h); throw; }
        slx_hits_free(&h);
    }
};

}  // namespace SeqLib
