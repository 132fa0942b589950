// BWAIndex.h -- drop-in for SeqLib::BWAIndex (/root/reference/SeqLib/BWAIndex.h:27-76,
// /root/reference/src/BWAIndex.cpp): same public methods, same exceptions, over the C-ABI of
// libseqlib_amd.so.  The index lives in bwa's on-disk layout on the host (LoadIndex/WriteIndex are
// byte-compatible with `bwa index` files) and is staged into HBM when a BWAAligner is created;
// ConstructIndex sorts suffixes and builds BWT/Occ/SA on the GPU.
#pragma once
#include <memory>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "seqlib_amd.h"
#include "SeqLib/BamHeader.h"
#include "SeqLib/UnalignedSequence.h"

namespace SeqLib {

class BWAIndex {
public:
    BWAIndex() = default;
    ~BWAIndex() { if (idx_) slx_index_free(idx_); }
    BWAIndex(const BWAIndex &) = delete;
    BWAIndex &operator=(const BWAIndex &) = delete;

    bool IsEmpty() const noexcept { return idx_ == nullptr; }

    BamHeader HeaderFromIndex() const { return BamHeader(printSamHeader()); }

    int NumSequences() const { return idx_ ? slx_index_nseq(idx_) : 0; }

    std::string ChrIDToName(int id) const
    {
        if (!idx_) throw std::runtime_error("Index has not be loaded / constructed");
        if (id < 0 || id >= slx_index_nseq(idx_))
            throw std::out_of_range("BWAIndex::ChrIDToName - id out of bounds of refs in index for id of " + std::to_string(id) +
                                    " on IDX of size " + std::to_string(slx_index_nseq(idx_)));
        return std::string(slx_index_name(idx_, id));
    }

    std::string printSamHeader() const
    {
        if (!idx_) return "";
        std::ostringstream out;
        for (int i = 0; i < slx_index_nseq(idx_); ++i)
            out << "@SQ\tSN:" << slx_index_name(idx_, i) << "\tLN:" << slx_index_len(idx_, i) << "\n";
        return out.str();
    }

    void ConstructIndex(const UnalignedSequenceVector &refs)
    {
        if (refs.empty()) return;
        for (auto const &r : refs)
            if (r.Name.empty() || r.Seq.empty())
                throw std::invalid_argument("BWAIndex::Construct each reference must have non-empty Name and Seq");
        std::vector<const char *> names, seqs;
        std::vector<int64_t> lens;
        for (auto const &r : refs) { names.push_back(r.Name.c_str()); seqs.push_back(r.Seq.c_str()); lens.push_back((int64_t)r.Seq.size()); }
        slx_index *ni = nullptr;
        const int rc = slx_index_build(names.data(), seqs.data(), lens.data(), (int)refs.size(), &ni);
        if (rc == SLX_EINVAL) throw std::invalid_argument(slx_last_error());
        if (rc == SLX_ENOMEM) throw std::bad_alloc();
        if (rc != SLX_OK) throw std::runtime_error(std::string("BWAIndex::Construct ") + slx_last_error());
        if (idx_) slx_index_free(idx_);
        idx_ = ni;
    }

    void LoadIndex(const std::string &prefix)
    {
        slx_index *ni = nullptr;
        if (slx_index_load(prefix.c_str(), &ni) != SLX_OK || !ni) throw std::runtime_error("Failed to load BWA index");
        if (idx_) slx_index_free(idx_);
        idx_ = ni;
    }

    void WriteIndex(const std::string &prefix) const
    {
        if (!idx_) throw std::runtime_error("BWAIndex::writeIndex: no index loaded");
        if (slx_index_write(idx_, prefix.c_str()) != SLX_OK) throw std::runtime_error(slx_last_error());
    }

    friend std::ostream &operator<<(std::ostream &os, const BWAIndex &idx)
    {
        if (!idx.idx_) os << "[BWAIndex] <no index loaded>";
        else os << "[BWAIndex] #seqs=" << slx_index_nseq(idx.idx_) << " pac_len=" << slx_index_l_pac(idx.idx_) << " holes=" << slx_index_n_holes(idx.idx_);
        return os;
    }

private:
    slx_index *idx_ = nullptr;
    friend class BWAAligner;
};

using BWAIndexPtr = std::shared_ptr<BWAIndex>;

}  // namespace SeqLib
