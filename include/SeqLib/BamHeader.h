// BamHeader.h -- SeqLib::BamHeader without htslib: the dictionary of reference sequences behind BWAIndex::HeaderFromIndex
// (/root/reference/src/BWAIndex.cpp:35-42) and BamWriter::SetHeader.  Same public surface as /root/reference/SeqLib/BamHeader.h:36-107
// (HeaderSequence / HeaderSequenceVector, the three constructors' worth of inputs that need no htslib type, NumSequences, both
// GetSequenceLength, IsOpen / isEmpty, AsString, IDtoName, Name2ID, GetHeaderSequenceVector) with the behaviour of
// /root/reference/src/BamHeader.cpp:12-140; get() / get_() and the constructor from a raw header hand out htslib's bam_hdr_t, which this
// image does not have (with htslib present the reference's own class is the one to keep).
#pragma once
#include <cstdint>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

namespace SeqLib {

// a reference sequence and its length: the SQ line of a header (/root/reference/SeqLib/BamHeader.h:18-28)
struct HeaderSequence {
    HeaderSequence(const std::string &n, uint32_t l) : Name(n), Length(l) {}
    std::string Name;
    uint32_t Length;
};
typedef std::vector<HeaderSequence> HeaderSequenceVector;

class BamHeader {
public:
    BamHeader() = default;
    // from reference names and lengths: the text is "@HD\tVN:1.4" and one @SQ line each (src/BamHeader.cpp:20-46)
    BamHeader(const HeaderSequenceVector &hsv) : open_(true)
    {
        std::ostringstream text;
        text << "@HD\tVN:1.4" << "\n";
        for (const HeaderSequence &q : hsv) {
            add(q.Name, q.Length);
            text << "@SQ\tSN:" << q.Name << "\tLN:" << q.Length << "\n";
        }
        text_ = text.str();
    }
    // from header text, lines separated by newlines (src/BamHeader.cpp:12-18: sam_hdr_parse takes SN / LN from every @SQ line)
    explicit BamHeader(const std::string &hdr) : text_(hdr), open_(true)
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
            add(name, len);
        }
    }
    bool IsOpen() const { return open_; }                    // "has been opened": anything but a default-constructed header
    bool isEmpty() const { return !open_; }
    int NumSequences() const { return (int)names_.size(); }   // 0 for an uninitialised header
    std::string IDtoName(int id) const
    {
        if (id < 0) throw std::invalid_argument("BamHeader::IDtoName - ID must be >= 0");
        if (!open_) throw std::out_of_range("BamHeader::IDtoName - Header is uninitialized");
        if (id >= (int)names_.size()) throw std::out_of_range("BamHeader::IDtoName - Requested ID is higher than number of sequences");
        return names_[(size_t)id];
    }
    // -1 when the name is not in the dictionary; of two sequences of one name the first keeps it (the reference's hash map does not overwrite)
    int Name2ID(const std::string &name) const
    {
        const auto it = n2i_.find(name);
        return it == n2i_.end() ? -1 : it->second;
    }
    int GetSequenceLength(int id) const { return id >= 0 && id < (int)lens_.size() ? (int)lens_[(size_t)id] : -1; }
    int GetSequenceLength(const std::string &id) const { return GetSequenceLength(Name2ID(id)); }
    std::string AsString() const { return text_; }
    HeaderSequenceVector GetHeaderSequenceVector() const
    {
        HeaderSequenceVector out;
        for (size_t i = 0; i < names_.size(); ++i) out.push_back(HeaderSequence(names_[i], (uint32_t)lens_[i]));
        return out;
    }
private:
    void add(const std::string &name, int64_t len)
    {
        n2i_.insert(std::make_pair(name, (int)names_.size()));
        names_.push_back(name); lens_.push_back(len);
    }
    std::string text_;
    std::vector<std::string> names_;
    std::vector<int64_t> lens_;
    std::unordered_map<std::string, int> n2i_;
    bool open_ = false;
};

}  // namespace SeqLib
