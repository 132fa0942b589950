// UnalignedSequence.h -- input record of the BWAAligner path; same members and constructors as
// /root/reference/SeqLib/UnalignedSequence.h:12-56 so that callers compile unchanged.
#pragma once
#include <iostream>
#include <string>
#include <vector>

namespace SeqLib {

struct UnalignedSequence {
    UnalignedSequence() {}
    UnalignedSequence(const std::string &n, const std::string &s) : Name(n), Seq(s), Strand('*') {}
    UnalignedSequence(const std::string &n, const std::string &s, const std::string &q) : Name(n), Seq(s), Qual(q), Strand('*') {}
    UnalignedSequence(const std::string &n, const std::string &s, const std::string &q, char t) : Name(n), Seq(s), Qual(q), Strand(t) {}

    std::string Name;   ///< name of the read / contig
    std::string Com;    ///< comment
    std::string Seq;    ///< bases (ACGTN)
    std::string Qual;   ///< qualities
    char Strand = '*';  ///< '*', '+' or '-'

    friend std::ostream &operator<<(std::ostream &os, const UnalignedSequence &us)
    {
        return os << "@" << us.Name << " " << us.Com << "\n" << us.Seq << "\n+\n" << us.Qual << "\n";
    }
};

typedef std::vector<UnalignedSequence> UnalignedSequenceVector;

}  // namespace SeqLib
