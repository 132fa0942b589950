// FastqReader.h -- SeqLib::FastqReader for the MI355X drop-in: FASTA / FASTQ (plain or gzip) -> UnalignedSequence,
// the producer in front of BWAAligner::alignSequences (first "next" row).
// Same interface and return conventions as /root/reference/SeqLib/FastqReader.h:26-74 and
// /root/reference/src/FastqReader.cpp:8-59.  The reference delegates the parsing to bwa's kseq.h (un-vendored
// submodule); the record grammar is restated here as one small pull parser over a zlib stream:
//   * anything before the first '>' or '@' is skipped;
//   * the name runs to the first whitespace byte, the comment (if the name did not end the line) to the end of line;
//   * sequence lines are concatenated until a line STARTS with '>', '@' or '+'; empty lines are skipped;
//   * after '+', quality lines are concatenated until they are at least as long as the sequence; a length mismatch or
//     a missing quality block is an error and ends the stream (kseq_read() < 0  =>  GetNextSequence() == false);
//   * a '\r' before the line feed is dropped from comment/sequence/quality lines longer than one byte.
// Like the reference, a field of `s` is only assigned once the parser has a buffer for it: Com stays untouched until
// the first record with a comment line has been seen, Qual until the first '+' block (kseq's lazily allocated kstring).
#pragma once
#include <cctype>
#include <iostream>
#include <string>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "SeqLib/UnalignedSequence.h"

namespace SeqLib {

class FastqReader {
public:
    FastqReader() = default;
    explicit FastqReader(const std::string &file) : m_file(file) { Open(m_file); }
    FastqReader(const FastqReader &) = delete;
    FastqReader &operator=(const FastqReader &) = delete;
    ~FastqReader() { if (fp_) gzclose(fp_); }

    // src/FastqReader.cpp:8-31: false (with a message on stderr) when the path does not stat or zlib cannot open it
    bool Open(const std::string &f)
    {
        m_file = f;
        struct stat st;
        if (stat(m_file.c_str(), &st) != 0) {
            std::cerr << "FastqReader: Failed to read non-existant file " << m_file << std::endl;
            return false;
        }
        if (fp_) { gzclose(fp_); fp_ = nullptr; }
        fp_ = (m_file != "-") ? gzopen(m_file.c_str(), "r") : gzdopen(fileno(stdin), "r");
        if (!fp_) {
            std::cerr << "FastqReader: Failed to read " << m_file << std::endl;
            return false;
        }
        gzbuffer(fp_, 1u << 18);
        begin_ = end_ = 0; eof_ = false; pending_ = 0;
        have_com_ = have_qual_ = false;
        return true;
    }

    // src/FastqReader.cpp:37-57
    bool GetNextSequence(UnalignedSequence &s)
    {
        if (!fp_) return false;
        if (parse_record() < 0) return false;
        s.Name = name_;
        if (have_com_) s.Com = com_;
        s.Seq = seq_;
        if (have_qual_) s.Qual = qual_;
        return true;
    }

private:
    static constexpr int BUF = 1 << 16;

    int getc_()
    {
        if (begin_ >= end_) {
            if (eof_) return -1;
            begin_ = 0;
            end_ = gzread(fp_, buf_, BUF);
            if (end_ < BUF) eof_ = true;
            if (end_ <= 0) { end_ = 0; return -1; }
        }
        return (unsigned char)buf_[begin_++];
    }
    // appends bytes up to (not including) the next delimiter; line mode stops at '\n', otherwise at any whitespace.
    // Returns the delimiter, or -1 when the input ended before any byte was seen.
    int until_(bool line, std::string &dst)
    {
        bool any = false;
        int c;
        while ((c = getc_()) != -1) {
            any = true;
            if (line ? c == '\n' : std::isspace(c) != 0) break;
            dst.push_back((char)c);
        }
        if (!any) return -1;
        if (line && dst.size() > 1 && dst.back() == '\r') dst.pop_back();
        return c == -1 ? 0 : c;
    }
    // >=0 sequence length; -1 end of input; -2 truncated / inconsistent quality
    int parse_record()
    {
        int c;
        if (pending_ == 0) {
            while ((c = getc_()) != -1 && c != '>' && c != '@') {}
            if (c == -1) return -1;
            pending_ = c;
        }
        name_.clear(); com_.clear(); seq_.clear(); qual_.clear();
        const int d = until_(false, name_);
        if (d == -1) return -1;
        if (d != '\n' && until_(true, com_) != -1) have_com_ = true;
        while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
            if (c == '\n') continue;
            seq_.push_back((char)c);
            until_(true, seq_);
        }
        if (c == '>' || c == '@') pending_ = c;
        if (c != '+') return (int)seq_.size();
        have_qual_ = true;
        while ((c = getc_()) != -1 && c != '\n') {}
        if (c == -1) return -2;
        while (until_(true, qual_) >= 0 && qual_.size() < seq_.size()) {}
        pending_ = 0;
        if (qual_.size() != seq_.size()) return -2;
        return (int)seq_.size();
    }

    std::string m_file;
    gzFile fp_ = nullptr;
    char buf_[BUF];
    int begin_ = 0, end_ = 0;
    bool eof_ = false;
    int pending_ = 0;           // header byte of the next record already consumed ('>' or '@'), 0 if none
    bool have_com_ = false, have_qual_ = false;
    std::string name_, com_, seq_, qual_;
};

}  // namespace SeqLib
