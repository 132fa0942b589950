// BamWriter.h -- SeqLib::BamWriter for the MI355X drop-in: the consumer behind BWAAligner ("next" row):
// writes the records alignSequence(s) produced as SAM text or as BAM (BGZF-framed, zlib deflate).
// Same interface and return conventions as /root/reference/SeqLib/BamWriter.h:10-136 and
// /root/reference/src/BamWriter.cpp:10-113 (false instead of exceptions; messages on stderr; Open refuses to
// reopen; WriteHeader needs a non-empty header and an open file).  The reference hands the bytes to htslib
// (hts_open / sam_hdr_write / sam_write1), which is not part of this image; the two encodings are restated from the
// SAM/BAM specification (SAMv1 sections 1.4, 4.1, 4.2):
//   SAM  : QNAME FLAG RNAME POS+1 MAPQ CIGAR RNEXT PNEXT+1 TLEN SEQ QUAL [TAG:TYPE:VALUE ...]; '*' for an absent
//          CIGAR / SEQ / QUAL (qual[0] == 0xff), '=' for RNEXT == RNAME, integer aux types all printed as 'i'.
//   BAM  : "BAM\1", header text, reference dictionary, then block_size-prefixed records whose variable part is the
//          bam1_t::data image as is; 64 KiB BGZF blocks and the 28-byte EOF block.  The bin field is computed from
//          the alignment span (reg2bin); compressed bytes depend on the deflate implementation, the inflated stream
//          does not.
// CRAM needs htslib's codec stack and reference access: Open() fails loudly for it; BuildIndex() (BAI) likewise.
#pragma once
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>
#include "SeqLib/BamHeader.h"
#include "SeqLib/BamRecord.h"

namespace SeqLib {

const int BAM = 4;      // /root/reference/SeqLib/BamWriter.h:10-12
const int SAM = 3;
const int CRAM = 6;

class BamWriter {
public:
    BamWriter() : output_format("wb") {}
    explicit BamWriter(int o)
    {
        switch (o) {
        case BAM: output_format = "wb"; break;
        case CRAM: output_format = "wc"; break;
        case SAM: output_format = "w"; break;
        default: throw std::invalid_argument("Invalid writer type");
        }
    }
    BamWriter(const BamWriter &) = delete;
    BamWriter &operator=(const BamWriter &) = delete;
    ~BamWriter() { Close(); }

    void SetHeader(const BamHeader &h) { hdr = h; }
    BamHeader Header() const { return hdr; }
    bool IsOpen() const { return fop != nullptr; }

    bool Open(const std::string &f)
    {
        if (fop) return false;                       // don't reopen
        m_out = f;
        if (output_format == "wc") {
            std::cerr << "BamWriter::Open - CRAM output needs htslib; not available in the MI355X drop-in" << std::endl;
            return false;
        }
        fop = (f == "-") ? stdout : std::fopen(f.c_str(), "wb");
        if (!fop) return false;
        blk.clear();
        return true;
    }

    bool WriteHeader() const
    {
        if (hdr.isEmpty()) {
            std::cerr << "BamWriter::WriteHeader - No header supplied. Provide with SetWriteHeader" << std::endl;
            return false;
        }
        if (!fop) {
            std::cerr << "BamWriter::WriteHeader - Output not open for writing. Open with Open()" << std::endl;
            return false;
        }
        const std::string text = hdr.AsString();
        if (output_format == "w") return std::fwrite(text.data(), 1, text.size(), fop) == text.size();
        std::string h("BAM\1", 4);
        put32(h, (uint32_t)text.size());
        h += text;
        put32(h, (uint32_t)hdr.NumSequences());
        for (int i = 0; i < hdr.NumSequences(); ++i) {
            const std::string nm = hdr.IDtoName(i);
            put32(h, (uint32_t)nm.size() + 1);
            h += nm; h.push_back('\0');
            put32(h, (uint32_t)hdr.GetSequenceLength(i));
        }
        if (!bgzf_write(h.data(), h.size())) return false;
        return bgzf_flush();                         // htslib starts the records in a fresh block
    }

    bool WriteRecord(const BamRecord &r)
    {
        if (!fop) return false;
        const bam1_t *b = r.raw();
        if (!b) return false;
        if (output_format == "w") {
            std::string line;
            if (!format_sam(b, line)) return false;
            return std::fwrite(line.data(), 1, line.size(), fop) == line.size();
        }
        const bam1_core_t &c = b->core;
        std::string rec;
        rec.reserve(36 + (size_t)b->l_data);
        put32(rec, (uint32_t)(32 + b->l_data));
        put32(rec, (uint32_t)c.tid);
        put32(rec, (uint32_t)c.pos);
        put32(rec, (uint32_t)reg2bin(c.pos, bam_endpos(b)) << 16 | (uint32_t)c.qual << 8 | (uint32_t)(c.l_qname & 0xff));
        put32(rec, (uint32_t)c.flag << 16 | (uint32_t)(c.n_cigar & 0xffff));
        put32(rec, (uint32_t)c.l_qseq);
        put32(rec, (uint32_t)c.mtid);
        put32(rec, (uint32_t)c.mpos);
        put32(rec, (uint32_t)c.isize);
        rec.append((const char *)b->data, (size_t)b->l_data);
        return bgzf_write(rec.data(), rec.size());
    }

    bool Close()
    {
        if (!fop) return false;
        bool ok = true;
        if (output_format == "wb") {
            ok = bgzf_flush();
            static const unsigned char eof_block[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            ok = std::fwrite(eof_block, 1, 28, fop) == 28 && ok;
        }
        if (fop == stdout) std::fflush(stdout); else ok = std::fclose(fop) == 0 && ok;
        fop = nullptr;
        return ok;
    }

    bool BuildIndex() const
    {
        if (fop) { std::cerr << "Trying to index open BAM. Close first with Close()" << std::endl; return false; }
        if (m_out.empty()) { std::cerr << "Trying to make index, but no BAM specified" << std::endl; return false; }
        std::cerr << "Failed to create index" << std::endl;     // BAI construction is htslib's; not in this drop-in
        return false;
    }
    bool SetCramReference(const std::string &) { return false; }

    friend std::ostream &operator<<(std::ostream &out, const BamWriter &b)
    {
        if (b.fop) out << "Write format: " << (b.output_format == "w" ? "SAM" : "BAM");
        return out << " Write file " << b.m_out;
    }

    // one SAM text line for a record ('\n'-terminated); false when a reference id is outside the header
    bool format_sam(const bam1_t *b, std::string &o) const
    {
        const bam1_core_t &c = b->core;
        auto rname = [&](int32_t tid, std::string &dst) {
            if (tid < 0) { dst += '*'; return true; }
            if (tid >= hdr.NumSequences()) return false;
            dst += hdr.IDtoName(tid);
            return true;
        };
        o.append(bam_get_qname(b));
        o += '\t'; o += std::to_string(c.flag); o += '\t';
        if (!rname(c.tid, o)) return false;
        o += '\t'; o += std::to_string(c.pos + 1);
        o += '\t'; o += std::to_string((int)c.qual); o += '\t';
        if (c.n_cigar == 0) o += '*';
        else {
            const uint8_t *raw = reinterpret_cast<const uint8_t *>(bam_get_cigar(b));
            for (uint32_t i = 0; i < c.n_cigar; ++i) {
                uint32_t w; std::memcpy(&w, raw + i * 4, 4);
                o += std::to_string(bam_cigar_oplen(w)); o += bam_cigar_opchr(w);
            }
        }
        o += '\t';
        if (c.mtid < 0) o += '*';
        else if (c.mtid == c.tid) o += '=';
        else if (!rname(c.mtid, o)) return false;
        o += '\t'; o += std::to_string(c.mpos + 1);
        o += '\t'; o += std::to_string(c.isize); o += '\t';
        if (c.l_qseq) {
            const uint8_t *s = bam_get_seq(b);
            for (int i = 0; i < c.l_qseq; ++i) o += "=ACMGRSVTWYHKDBN"[bam_seqi(s, i)];
            o += '\t';
            const uint8_t *q = bam_get_qual(b);
            if (q[0] == 0xff) o += '*';
            else for (int i = 0; i < c.l_qseq; ++i) o += (char)(q[i] + 33);
        } else o += "*\t*";
        const uint8_t *s = bam_get_aux(b), *end = b->data + b->l_data;
        while (end - s >= 4) {
            o += '\t'; o += (char)s[0]; o += (char)s[1]; o += ':';
            const uint8_t t = s[2];
            s += 3;
            if (!format_aux(t, s, end, o)) return false;
        }
        o += '\n';
        return true;
    }

private:
    static void put32(std::string &s, uint32_t v) { char b[4] = {(char)v, (char)(v >> 8), (char)(v >> 16), (char)(v >> 24)}; s.append(b, 4); }
    template <typename T> static T rd(const uint8_t *p) { T v; std::memcpy(&v, p, sizeof(T)); return v; }
    static int reg2bin(int64_t beg, int64_t end)     // SAMv1 5.3
    {
        --end;
        if (beg >> 14 == end >> 14) return (int)(((1 << 15) - 1) / 7 + (beg >> 14));
        if (beg >> 17 == end >> 17) return (int)(((1 << 12) - 1) / 7 + (beg >> 17));
        if (beg >> 20 == end >> 20) return (int)(((1 << 9) - 1) / 7 + (beg >> 20));
        if (beg >> 23 == end >> 23) return (int)(((1 << 6) - 1) / 7 + (beg >> 23));
        if (beg >> 26 == end >> 26) return (int)(((1 << 3) - 1) / 7 + (beg >> 26));
        return 0;
    }
    static bool format_num(uint8_t t, const uint8_t *&s, const uint8_t *end, std::string &o)
    {
        char buf[64];
        switch (t) {
        case 'c': if (end - s < 1) return false; o += std::to_string((int)rd<int8_t>(s)); s += 1; return true;
        case 'C': if (end - s < 1) return false; o += std::to_string((unsigned)rd<uint8_t>(s)); s += 1; return true;
        case 's': if (end - s < 2) return false; o += std::to_string((int)rd<int16_t>(s)); s += 2; return true;
        case 'S': if (end - s < 2) return false; o += std::to_string((unsigned)rd<uint16_t>(s)); s += 2; return true;
        case 'i': if (end - s < 4) return false; o += std::to_string(rd<int32_t>(s)); s += 4; return true;
        case 'I': if (end - s < 4) return false; o += std::to_string(rd<uint32_t>(s)); s += 4; return true;
        case 'f': if (end - s < 4) return false; std::snprintf(buf, sizeof buf, "%g", rd<float>(s)); o += buf; s += 4; return true;
        case 'd': if (end - s < 8) return false; std::snprintf(buf, sizeof buf, "%g", rd<double>(s)); o += buf; s += 8; return true;
        }
        return false;
    }
    static bool format_aux(uint8_t t, const uint8_t *&s, const uint8_t *end, std::string &o)
    {
        switch (t) {
        case 'A': if (end - s < 1) return false; o += "A:"; o += (char)*s++; return true;
        case 'c': case 'C': case 's': case 'S': case 'i': case 'I': o += "i:"; return format_num(t, s, end, o);
        case 'f': o += "f:"; return format_num(t, s, end, o);
        case 'd': o += "d:"; return format_num(t, s, end, o);
        case 'Z': case 'H':
            o += (char)t; o += ':';
            while (s < end && *s) o += (char)*s++;
            if (s >= end) return false;
            ++s;
            return true;
        case 'B': {
            if (end - s < 5) return false;
            const uint8_t sub = *s++;
            const uint32_t n = rd<uint32_t>(s); s += 4;
            o += "B:"; o += (char)sub;
            for (uint32_t i = 0; i < n; ++i) { o += ','; if (!format_num(sub, s, end, o)) return false; }
            return true;
        }
        }
        return false;
    }

    // ---- BGZF (SAMv1 4.1): gzip members of <= 64 KiB carrying their compressed size in a "BC" extra field
    static constexpr size_t BGZF_BLOCK = 0xff00;
    bool bgzf_write(const char *p, size_t n) const
    {
        while (n) {
            const size_t take = std::min(n, BGZF_BLOCK - blk.size());
            blk.append(p, take); p += take; n -= take;
            if (blk.size() == BGZF_BLOCK && !bgzf_flush()) return false;
        }
        return true;
    }
    bool bgzf_flush() const
    {
        if (blk.empty()) return true;
        unsigned char out[0x10000];
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
        zs.next_in = (Bytef *)blk.data(); zs.avail_in = (uInt)blk.size();
        zs.next_out = out + 18; zs.avail_out = sizeof out - 18 - 8;
        const int rc = deflate(&zs, Z_FINISH);
        const size_t clen = zs.total_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) return false;
        static const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(out, head, 16);
        const size_t total = 18 + clen + 8;
        out[16] = (unsigned char)((total - 1) & 0xff); out[17] = (unsigned char)((total - 1) >> 8);
        const uint32_t crc = (uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)blk.data(), (uInt)blk.size());
        const uint32_t isz = (uint32_t)blk.size();
        for (int i = 0; i < 4; ++i) { out[18 + clen + i] = (unsigned char)(crc >> (8 * i)); out[22 + clen + i] = (unsigned char)(isz >> (8 * i)); }
        blk.clear();
        return std::fwrite(out, 1, total, fop) == total;
    }

    std::string m_out;
    std::string output_format;
    FILE *fop = nullptr;
    BamHeader hdr;
    mutable std::string blk;        // uncompressed bytes of the BGZF block being filled
};

}  // namespace SeqLib
