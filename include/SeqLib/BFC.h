// BFC.h -- MI355X-native mirror of SeqLib::BFC (/root/reference/SeqLib/BFC.h:22-116, /root/reference/src/BFC.cpp): k-mer-spectrum error
// correction split the way the reference splits it -- Train() counts the k-mers of the stored reads (fml_count, src/BFC.cpp:208-286) and
// keeps the table (here: in HBM, in a device context this object owns -- slx_fml_count), ErrorCorrect() derives kcov / min_cov from its histogram and corrects the stored reads in
// place (src/BFC.cpp:289-362, slx_fml_error_correct).  Header-only over include/seqlib_amd_fml.h.
#pragma once
#include <algorithm>
#include <cassert>
#include <string>
#include <string_view>
#include <vector>
#include "SeqLib/FermiAssembler.h"

namespace SeqLib {

class BFC {
public:
    BFC() { fml_opt_init(&fml_opt); }
    ~BFC() { if (own) slx_fml_free(own); }
    BFC(const BFC &) = delete;          // the object owns its k-mer table (the reference's BFC owns its bfc_ch_t, src/BFC.cpp:208-286)
    BFC &operator=(const BFC &) = delete;

    /** Add a sequence for training or correction (src/BFC.cpp:59-86: refuses an empty sequence or a quality string of another length) */
    bool AddSequence(std::string_view seq, std::string_view qual, std::string_view name)
    {
        if (seq.empty() || (!qual.empty() && qual.size() != seq.size())) return false;
        m_seq.emplace_back(seq); m_qual.emplace_back(qual); m_names.emplace_back(name);
        return true;
    }

    /** k for training; 0 = chosen from the total length of the reads (fml_opt_adjust) */
    void SetKmer(int k) { kmer = k; }

    /** Count the k-mers of the stored reads (src/BFC.cpp:208-286).  The table stays on the device until the next Train(). */
    void Train()
    {
        fml_opt_init(&fml_opt);
        if (kmer <= 0) {
            std::vector<int32_t> lens;
            for (const std::string &s : m_seq) lens.push_back((int32_t)s.size());
            slx_fml_opt_adjust(&fml_opt, (int64_t)lens.size(), lens.data());
            kmer = fml_opt.ec_k;
        }
        detail::FlatReads f = flat();
        if (!own && slx_fml_create(-1, &own) != SLX_OK) throw std::runtime_error(std::string("seqlib_amd: ") + slx_last_error());
        detail::fml_check(slx_fml_count(own, f.bases.data(), f.has_qual ? f.quals.data() : nullptr, f.offs.data(), (int64_t)m_seq.size(), kmer, 20));
        trained = true;
    }

    /** Correct the stored reads against the trained table, in place (src/BFC.cpp:289-362) */
    void ErrorCorrect()
    {
        assert(kmer > 0);
        if (!trained) throw std::runtime_error("BFC::ErrorCorrect: Train() first");
        detail::FlatReads f = flat();
        int min_cov = 0;
        detail::fml_check(slx_fml_error_correct(own, &fml_opt, f.bases.data(), f.has_qual ? f.quals.data() : nullptr, f.offs.data(), (int64_t)m_seq.size(),
                                                0, nullptr, nullptr, &kcov, &min_cov));
        for (size_t i = 0; i < m_seq.size(); ++i) {
            m_seq[i].assign(f.bases, (size_t)f.offs[i], (size_t)(f.offs[i + 1] - f.offs[i]));
            if (f.has_qual && m_qual[i].size() == m_seq[i].size()) m_qual[i].assign(f.quals, (size_t)f.offs[i], (size_t)(f.offs[i + 1] - f.offs[i]));
        }
    }

    /** Clear the stored reads, but not the training outcome (src/BFC.cpp:195-205) */
    void ClearReads() { m_seq.clear(); m_qual.clear(); m_names.clear(); m_idx = 0; }

    float GetKCov() const { return kcov; }
    int GetKMer() const { return kmer; }
    int NumSequences() const { return (int)m_seq.size(); }

    /** The next stored sequence, upper case, and its name (src/BFC.cpp:141-151) */
    bool GetSequence(std::string &s, std::string &q)
    {
        if (m_idx >= m_seq.size()) return false;
        s = m_seq[m_idx];
        q = m_names[m_idx];
        std::transform(s.begin(), s.end(), s.begin(), ::toupper);
        ++m_idx;
        return true;
    }
    void ResetGetSequence() { m_idx = 0; }

private:
    detail::FlatReads flat() const { return detail::flat_reads(m_seq, m_qual, nullptr); }
    size_t m_idx = 0;
    std::vector<std::string> m_seq, m_qual, m_names;
    fml_opt_t fml_opt;
    int kmer = 0;
    float kcov = 0;
    bool trained = false;
    slx_fml *own = nullptr;          // this object's device context: its count table lives there from Train() to the next Train()
};

}  // namespace SeqLib
