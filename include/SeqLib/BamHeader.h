// BamHeader.h -- minimal stand-in for SeqLib::BamHeader, enough for BWAIndex::HeaderFromIndex
// (/root/reference/src/BWAIndex.cpp:35-42): built from SAM header text, answers NumSequences / IDtoName /
// Name2ID / GetSequenceLength / AsString.  The htslib-backed remainder of the class
// (/root/reference/SeqLib/BamHeader.h) is BAM I/O, a "next" row of the scope table.
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace SeqLib {

class BamHeader {
public:
    BamHeader() = default;
    explicit BamHeader(const std::string &hdr) : text_(hdr)
    {
        std::istringstream in(hdr);
        std::string line;
        while (std::getline(in, line)) {
            if (line.rfind("@SQ", 0) != 0) continue;
            std::string name; int64_t len = 0;
            std::istringstream f(line);
            std::string tok;
            while (std::getline(f, tok, '\t')) {
                if (tok.rfind("SN:", 0) == 0) name = tok.substr(3);
                else if (tok.rfind("LN:", 0) == 0) len = std::stoll(tok.substr(3));
            }
            names_.push_back(name); lens_.push_back(len);
        }
    }
    bool isEmpty() const { return names_.empty(); }
    int NumSequences() const { return (int)names_.size(); }
    std::string IDtoName(int id) const
    {
        if (id < 0) throw std::invalid_argument("BamHeader::IDtoName - ID must be >= 0");
        if (id >= (int)names_.size()) throw std::out_of_range("BamHeader::IDtoName - Requested ID is higher than number of sequences");
        return names_[(size_t)id];
    }
    int Name2ID(const std::string &name) const
    {
        for (size_t i = 0; i < names_.size(); ++i) if (names_[i] == name) return (int)i;
        return -1;
    }
    int GetSequenceLength(int id) const { return id >= 0 && id < (int)lens_.size() ? (int)lens_[(size_t)id] : -1; }
    std::string AsString() const { return text_; }
private:
    std::string text_;
    std::vector<std::string> names_;
    std::vector<int64_t> lens_;
};

}  // namespace SeqLib
