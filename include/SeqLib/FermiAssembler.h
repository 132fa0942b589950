// FermiAssembler.h -- MI355X-native mirror of SeqLib::FermiAssembler (/root/reference/SeqLib/FermiAssembler.h:20-150,
// /root/reference/src/FermiAssembler.cpp): same class name, methods, argument meaning and defaults, header-only over the C-ABI of
// include/seqlib_amd_fml.h.  The reads live in the object as the reference's fseq1_t array does; CorrectReads / CorrectAndFilterReads /
// PerformAssembly / DirectAssemble hand them to the GPU path as ONE window (slx_fml_correct / slx_fml_assemble /
// slx_fml_direct_assemble).  AssembleWindows() is the batch entry the reference lacks: many windows in one call, each assembled as a
// FermiAssembler holding only its reads would (window-parallel).
// Errors: the reference's fermi-lite calls cannot fail; here a C-ABI error (no GPU, unsupported option) is a std::runtime_error.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "SeqLib/BamRecord.h"
#include "SeqLib/UnalignedSequence.h"
#include "SeqLib/fml_compat.h"

namespace SeqLib {

namespace detail {
struct FmlContext {          // one slx_fml per process and device, created on first use
    slx_fml *h = nullptr;
    ~FmlContext() { if (h) slx_fml_free(h); }
    static slx_fml *get()
    {
        static FmlContext c;
        if (!c.h && slx_fml_create(-1, &c.h) != SLX_OK) throw std::runtime_error(std::string("seqlib_amd: ") + slx_last_error());
        return c.h;
    }
};
inline void fml_check(int rc) { if (rc != SLX_OK) throw std::runtime_error(std::string("seqlib_amd: ") + slx_last_error()); }

struct FlatReads {
    std::string bases, quals;
    std::vector<uint64_t> offs;
    bool has_qual = false;
};
// The reads as flat text.  Qualities are kept PER READ as the reference keeps them (fseq1_t::qual is NULL for a read without, src/FermiAssembler.cpp:52-62,
// and fermi-lite then counts every base of that read as high quality): a read without a quality string of its sequence's length gets a stretch of 'I'
// (Phred 40, above every threshold of bfc) in the flat buffer, and its own quality string is never written back.  No read with qualities: no buffer.
inline FlatReads flat_reads(const std::vector<std::string> &seq, const std::vector<std::string> &qual, const std::vector<char> *hasq)
{
    FlatReads f;
    f.offs.push_back(0);
    for (size_t i = 0; i < seq.size(); ++i) if ((!hasq || (*hasq)[i]) && !seq[i].empty() && qual[i].size() == seq[i].size()) { f.has_qual = true; break; }
    for (size_t i = 0; i < seq.size(); ++i) {
        f.bases += seq[i];
        if (f.has_qual) {
            if ((!hasq || (*hasq)[i]) && qual[i].size() == seq[i].size()) f.quals += qual[i]; else f.quals.append(seq[i].size(), 'I');
        }
        f.offs.push_back(f.bases.size());
    }
    return f;
}
}  // namespace detail

class FermiAssembler {
public:
    /** Create an empty FermiAssembler with default parameters (src/FermiAssembler.cpp:6-8) */
    FermiAssembler() { fml_opt_init(&opt); }
    /** Create an empty FermiAssembler with the provided parameters (:10-13) */
    explicit FermiAssembler(fml_opt_t &_opt) : opt(_opt) {}
    ~FermiAssembler() { ClearReads(); ClearContigs(); }
    FermiAssembler(const FermiAssembler &) = delete;
    FermiAssembler &operator=(const FermiAssembler &) = delete;

    /** Provide a set of reads to be assembled (:82-101: names, sequences and qualities are copied) */
    void AddReads(const BamRecordVector &brv)
    {
        for (BamRecordVector::const_iterator r = brv.begin(); r != brv.end(); ++r) push(r->Qname(), r->Sequence(), r->Qualities(), true);
    }
    /** Add a set of unaligned sequences (:64-81: every one, empty or not; the quality string is taken as it is) */
    void AddReads(const UnalignedSequenceVector &v)
    {
        for (UnalignedSequenceVector::const_iterator r = v.begin(); r != v.end(); ++r) push(r->Name, r->Seq, r->Qual, true);
    }
    /** Add a single sequence (:52-62: ignored when its name or sequence is empty; no quality string = NULL) */
    void AddRead(const UnalignedSequence &r)
    {
        if (r.Seq.empty() || r.Name.empty()) return;
        push(r.Name, r.Seq, r.Qual, !r.Qual.empty());
    }
    /** Add a single aligned read (:46-50) */
    void AddRead(const BamRecord &r) { AddRead(UnalignedSequence(r.Qname(), r.Sequence(), r.Qualities())); }

    void ClearReads() { m_names.clear(); m_seq.clear(); m_qual.clear(); m_hasq.clear(); }
    void ClearContigs() { if (m_utgs) slx_fml_utgs_free(n_utg, m_utgs); m_utgs = nullptr; n_utg = 0; }

    /** Error correction of the reads in place (:133 fml_correct) */
    void CorrectReads() { run_correct(0); }
    /** Error correction by trimming at / dropping reads with unique k-mers (:135-138 fml_fltuniq) */
    void CorrectAndFilterReads() { run_correct(1); }

    /** The sequences in this object, possibly corrected (:165-177: names and sequences; a dropped read has an empty sequence) */
    UnalignedSequenceVector GetSequences() const
    {
        UnalignedSequenceVector r;
        for (size_t i = 0; i < m_seq.size(); ++i) {
            UnalignedSequence read;
            read.Seq = m_seq[i];
            read.Name = m_names[i];
            r.push_back(read);
        }
        return r;
    }

    /** String graph assembly (:140-143 fml_assemble: correction, unique-k-mer filter, unitigs, graph cleaning).  The reference's
     * fml_assemble frees the reads it is given; here they stay as they were (GetSequences still works). */
    void PerformAssembly()
    {
        ClearContigs();
        detail::FlatReads f = flat();
        const int64_t win_off[2] = {0, (int64_t)m_seq.size()};
        detail::fml_check(slx_fml_assemble(detail::FmlContext::get(), &opt, f.bases.data(), f.has_qual ? f.quals.data() : nullptr, f.offs.data(), (int64_t)m_seq.size(),
                                           win_off, 1, &m_utgs, &n_utg));
    }

    /** Assembly without error correction (:26-44); kcov as BFC::GetKCov() reports it */
    void DirectAssemble(float kcov)
    {
        ClearContigs();
        detail::FlatReads f = flat();
        detail::fml_check(slx_fml_direct_assemble(detail::FmlContext::get(), &opt, kcov, f.bases.data(), f.offs.data(), (int64_t)m_seq.size(), &m_utgs, &n_utg));
    }

    /** The assembled contigs, upper case ACGTN (:145-151) */
    std::vector<std::string> GetContigs() const
    {
        std::vector<std::string> c;
        for (int i = 0; i < n_utg; ++i) c.push_back(std::string(m_utgs[i].seq));
        return c;
    }

    void SetMinOverlap(uint32_t m) { opt.min_asm_ovlp = (int)m; }
    void SetAggressiveTrim() { opt.mag_opt.flag |= MAG_F_AGGRESSIVE; }
    void SetSimplifyBubble() { opt.mag_opt.flag &= ~MAG_F_NO_SIMPL; }          // closed bubbles cut down to their two best-supported paths before mag_g_pop_simple (fml_graph.h: simplify_bubble)
    void SetDropOverlapRatio(double ratio) { opt.mag_opt.min_dratio1 = (float)ratio; }
    void SetKmerMinThreshold(int min) { opt.min_cnt = min; }
    void SetKmerMaxThreshold(int max) { opt.max_cnt = max; }
    uint32_t GetMinOverlap() const { return (uint32_t)opt.min_asm_ovlp; }
    size_t NumSequences() const { return m_seq.size(); }

    /** GFA of the unitig graph (:180-204) */
    void WriteGFA(std::ostream &out)
    {
        out << "H\tVN:Z:1.0" << std::endl;
        for (int i = 0; i < n_utg; ++i) {
            const fml_utg_t *u = m_utgs + i;
            out << "S\t" << i << "\t";
            out << u->seq << "\tLN:i:" << u->len << "\tRC:i:" << u->nsr << "\tPD:Z:";
            out << u->cov << std::endl;
            for (int j = 0; j < u->n_ovlp[0] + u->n_ovlp[1]; ++j) {
                fml_ovlp_t *o = &u->ovlp[j];
                if (i < (int)o->id) {
                    out << "L\t" << i << "\t" << "+-"[!o->from] << "\t" << o->id << "\t" << "+-"[o->to] << "\t" << o->len << "M" << std::endl;
                }
            }
        }
    }

    /** (new) Many windows in one call: window w holds reads[win_off[w] .. win_off[w + 1]); contigs[w] = what a FermiAssembler holding
     * only those reads returns from PerformAssembly() + GetContigs().  The windows run through the GPU stages together. */
    static void AssembleWindows(const UnalignedSequenceVector &reads, const std::vector<int64_t> &win_off, std::vector<std::vector<std::string> > &contigs,
                                const fml_opt_t *options = nullptr)
    {
        fml_opt_t o;
        if (options) o = *options; else fml_opt_init(&o);
        const int n_win = (int)win_off.size() - 1;
        contigs.assign((size_t)std::max(n_win, 0), std::vector<std::string>());
        if (n_win <= 0) return;
        std::vector<std::string> sq, ql;
        sq.reserve(reads.size()); ql.reserve(reads.size());
        for (const UnalignedSequence &r : reads) { sq.push_back(r.Seq); ql.push_back(r.Qual); }
        detail::FlatReads f = detail::flat_reads(sq, ql, nullptr);
        std::vector<fml_utg_t *> utgs((size_t)n_win, nullptr);
        std::vector<int> n_utg((size_t)n_win, 0);
        detail::fml_check(slx_fml_assemble(detail::FmlContext::get(), &o, f.bases.data(), f.has_qual ? f.quals.data() : nullptr, f.offs.data(), (int64_t)reads.size(),
                                           win_off.data(), n_win, utgs.data(), n_utg.data()));
        for (int w = 0; w < n_win; ++w) {
            for (int i = 0; i < n_utg[(size_t)w]; ++i) contigs[(size_t)w].push_back(std::string(utgs[(size_t)w][i].seq));
            slx_fml_utgs_free(n_utg[(size_t)w], utgs[(size_t)w]);
        }
    }

private:
    void push(const std::string &name, const std::string &seq, const std::string &qual, bool has_qual)
    {
        m_names.push_back(name); m_seq.push_back(seq); m_qual.push_back(qual); m_hasq.push_back(has_qual ? 1 : 0);
    }
    detail::FlatReads flat() const { return detail::flat_reads(m_seq, m_qual, &m_hasq); }
    void run_correct(int flt_uniq)
    {
        detail::FlatReads f = flat();
        const int64_t n = (int64_t)m_seq.size(), win_off[2] = {0, n};
        std::vector<int32_t> ns((size_t)n + 1), nl((size_t)n + 1);
        fml_opt_t o = opt;          // fml_correct adjusts a copy: the object's options keep ec_k = 0 (auto)
        detail::fml_check(slx_fml_correct(detail::FmlContext::get(), &o, f.bases.data(), f.has_qual ? f.quals.data() : nullptr, f.offs.data(), n, win_off, 1, flt_uniq,
                                          ns.data(), nl.data(), nullptr, nullptr));
        for (size_t i = 0; i < m_seq.size(); ++i) {
            if (flt_uniq) {
                m_seq[i] = m_seq[i].substr((size_t)ns[i], (size_t)nl[i]);
                if (m_hasq[i] && m_qual[i].size() >= (size_t)(ns[i] + nl[i])) m_qual[i] = m_qual[i].substr((size_t)ns[i], (size_t)nl[i]);
            } else {
                m_seq[i].assign(f.bases, (size_t)f.offs[i], (size_t)(f.offs[i + 1] - f.offs[i]));
                if (f.has_qual && m_hasq[i] && m_qual[i].size() == m_seq[i].size()) m_qual[i].assign(f.quals, (size_t)f.offs[i], (size_t)(f.offs[i + 1] - f.offs[i]));
            }
        }
    }

    std::vector<std::string> m_names, m_seq, m_qual;
    std::vector<char> m_hasq;
    int n_utg = 0;
    fml_opt_t opt;
    fml_utg_t *m_utgs = nullptr;
};

}  // namespace SeqLib
