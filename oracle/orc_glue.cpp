/*
 * orc_glue.cpp -- ORACLE (test infrastructure only; see orc.h).
 *
 * Restates the SeqLib glue of /root/reference/src/BWAAligner.cpp:89-250: region filtering
 * (:117-129), std::sort by (mapq desc, rid, pos) (:7-11,:133), the secondary filters (:136-146),
 * record construction (:151-236) and the NA/NM/AS tags (:238-241, src/BamRecord.cpp:960-970).
 * C++ only because the reference sorts with std::sort and ties must fall the same way.
 */
#include "orc.h"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <climits>
#include <string>
#include <thread>
#include <vector>

extern "C" orc_counters *orc_counters_ptr(void);

namespace {
/* src/BWAAligner.cpp:7-11 */
bool aln_sort(const orc_aln &a, const orc_aln &b)
{
    if (a.mapq != b.mapq) return a.mapq > b.mapq;
    if (a.rid != b.rid) return a.rid < b.rid;
    return a.pos < b.pos;
}

void append_int_tag(std::vector<uint8_t> &d, const char *tag, int32_t v)
{ /* bam_aux_append(b, tag, 'i', 4, &v) */
    d.push_back((uint8_t)tag[0]); d.push_back((uint8_t)tag[1]); d.push_back('i');
    uint8_t raw[4]; std::memcpy(raw, &v, 4);
    d.insert(d.end(), raw, raw + 4);
}
} // namespace

extern "C" int orc_align_sequence(const orc_opt *opt, const orc_index *idx, const char *seq, int len,
                                  const char *name, int hardclip, double keepSecFrac, int maxSecondary,
                                  uint64_t rng_base, uint64_t ordinal, orc_hit **out)
{
    *out = nullptr;
    if (!idx) return 0;                                     /* :101 */
    /* mem_align1 passes lrand48() as the hash salt: draw number `ordinal` (0-based) of the stream */
    uint64_t salt = orc_lrand48_nth(rng_base, ordinal + 1);
    orc_reg *regs = nullptr;
    int n_regs = orc_align1(opt, idx, len, seq, salt, &regs); /* :104-109 */

    double primaryScore = 0;
    std::vector<orc_aln> hits;
    hits.reserve((size_t)n_regs);
    for (int i = 0; i < n_regs; ++i) {                      /* :117-129 */
        orc_reg &r = regs[i];
        /* `r.secondary` is an int used as a truth value: -1 (primary) is true, 0 (secondary to region 0) is false */
        if (r.secondary && (keepSecFrac < 0.0 || keepSecFrac > 1.0)) continue;
        hits.push_back(orc_reg2aln(opt, idx, len, seq, &r));
    }
    std::free(regs);
    std::sort(hits.begin(), hits.end(), aln_sort);          /* :133 */

    std::vector<orc_hit> recs;
    for (size_t i = 0; i < hits.size(); ++i) {              /* :136-248 */
        orc_aln &h = hits[i];
        bool isSec = (h.flag & 0x100);
        bool tooLow = isSec && (primaryScore * keepSecFrac > h.score);
        bool tooMany = isSec && (int(i) > maxSecondary);
        if (tooLow || tooMany) { std::free(h.cigar); continue; }
        if (!isSec) primaryScore = h.score;

        orc_hit rec;
        std::memset(&rec, 0, sizeof rec);
        rec.rid = h.rid; rec.pos = h.pos; rec.mapq = (uint8_t)h.mapq;
        rec.flag = (uint16_t)(h.flag | (h.is_rev ? 0x10 : 0));
        rec.n_cigar = h.n_cigar;
        rec.score = h.score; rec.nm = (int32_t)h.NM; rec.na = n_regs;

        /* optional hard clip (:164-177): leading op-3 length and query-consuming total */
        size_t tstart = 0, clen = (size_t)len;
        if (hardclip) {
            clen = 0;
            for (int c = 0; c < h.n_cigar; ++c) {
                uint32_t op = h.cigar[c] & 0xf, ol = h.cigar[c] >> 4;
                if (c == 0 && op == 3) tstart = ol;
                else if ((0x3C1A7 >> (op << 1)) & 1) clen += ol; /* bam_cigar_type(op)&1 */
            }
        }
        const char *cl = seq + tstart;
        int sl = (int)clen;

        size_t l_qname = std::strlen(name) + 1;
        std::vector<uint8_t> d;
        d.insert(d.end(), (const uint8_t *)name, (const uint8_t *)name + l_qname);
        /* cigar with op 3 rewritten to S(4) or H(5) (:191-202) */
        rec.cigar = (uint32_t *)std::malloc(4 * (size_t)(h.n_cigar ? h.n_cigar : 1));
        for (int k = 0; k < h.n_cigar; ++k) {
            uint32_t w = h.cigar[k];
            if ((w & 0xf) == 3) w = (w & ~0xfu) | (hardclip ? 5u : 4u);
            rec.cigar[k] = w;
            uint8_t raw[4]; std::memcpy(raw, &w, 4);
            d.insert(d.end(), raw, raw + 4);
        }
        std::free(h.cigar);
        /* 4-bit packed sequence (:206-233); reverse strand swaps A<->T only, C/G untouched */
        size_t seq_off = d.size();
        d.resize(seq_off + (size_t)((sl + 1) >> 1), 0);
        if (h.is_rev) {
            int j = 0;
            for (int p = sl - 1; p >= 0; --p, ++j) {
                uint8_t v = 15;
                switch (cl[p]) { case 'A': v = 8; break; case 'C': v = 2; break; case 'G': v = 4; break; case 'T': v = 1; break; }
                d[seq_off + (size_t)(j >> 1)] |= (uint8_t)(v << ((~j & 1) << 2));
            }
        } else {
            for (int p = 0; p < sl; ++p) {
                uint8_t v = 15;
                switch (cl[p]) { case 'A': v = 1; break; case 'C': v = 2; break; case 'G': v = 4; break; case 'T': v = 8; break; }
                d[seq_off + (size_t)(p >> 1)] |= (uint8_t)(v << ((~p & 1) << 2));
            }
        }
        /* qualities (:235-236): byte 0 = 0xff, rest uninitialised in the reference -> zero here */
        size_t q_off = d.size();
        d.resize(q_off + (size_t)sl, 0);
        if (sl > 0) d[q_off] = 0xff;
        append_int_tag(d, "NA", n_regs);                    /* :238 */
        append_int_tag(d, "NM", (int32_t)h.NM);             /* :239 */
        append_int_tag(d, "AS", h.score);                   /* :241 (XA never fires) */
        rec.l_qname = (int32_t)l_qname; rec.l_qseq = sl;
        rec.l_data = (int32_t)d.size();
        rec.data = (uint8_t *)std::malloc(d.size());
        std::memcpy(rec.data, d.data(), d.size());
        recs.push_back(rec);
    }
    orc_counters *cnt = orc_counters_ptr();
    cnt->n_reads += 1; cnt->read_bases += (uint64_t)len; cnt->n_hits += recs.size();
    for (auto &r : recs) cnt->n_cigar_ops += (uint64_t)r.n_cigar;
    if (!recs.empty()) {
        *out = (orc_hit *)std::malloc(recs.size() * sizeof(orc_hit));
        std::memcpy(*out, recs.data(), recs.size() * sizeof(orc_hit));
    }
    return (int)recs.size();
}

extern "C" void orc_hits_free(orc_hit *h, int n)
{
    if (!h) return;
    for (int i = 0; i < n; ++i) { std::free(h[i].cigar); std::free(h[i].data); }
    std::free(h);
}

extern "C" int orc_align_batch(const orc_opt *opt, const orc_index *idx, const char *bases, const uint64_t *offs,
                               int64_t n_reads, int hardclip, double keepSecFrac, int maxSecondary,
                               uint64_t rng_base, uint64_t first_ordinal, orc_batch_out *o)
{
    std::vector<int32_t> read_idx, rid, score, nm, na, n_cigar;
    std::vector<int64_t> pos, cig_off, hit_off;
    std::vector<uint16_t> flag;
    std::vector<uint8_t> mapq;
    std::vector<uint32_t> cigar;
    hit_off.push_back(0); cig_off.push_back(0);
    for (int64_t r = 0; r < n_reads; ++r) {
        orc_hit *h = nullptr;
        int n = orc_align_sequence(opt, idx, bases + offs[r], (int)(offs[r + 1] - offs[r]), "r", hardclip, keepSecFrac,
                                   maxSecondary, rng_base, first_ordinal + (uint64_t)r, &h);
        for (int i = 0; i < n; ++i) {
            read_idx.push_back((int32_t)r); rid.push_back(h[i].rid); pos.push_back(h[i].pos); flag.push_back(h[i].flag);
            mapq.push_back(h[i].mapq); score.push_back(h[i].score); nm.push_back(h[i].nm); na.push_back(h[i].na);
            n_cigar.push_back(h[i].n_cigar);
            cigar.insert(cigar.end(), h[i].cigar, h[i].cigar + h[i].n_cigar);
            cig_off.push_back((int64_t)cigar.size());
        }
        hit_off.push_back((int64_t)rid.size());
        orc_hits_free(h, n);
    }
    auto dup = [](const void *p, size_t bytes) { void *q = std::malloc(bytes ? bytes : 1); std::memcpy(q, p, bytes); return q; };
    o->n_hits = (int64_t)rid.size();
    o->read_idx = (int32_t *)dup(read_idx.data(), read_idx.size() * 4);
    o->rid = (int32_t *)dup(rid.data(), rid.size() * 4);
    o->score = (int32_t *)dup(score.data(), score.size() * 4);
    o->nm = (int32_t *)dup(nm.data(), nm.size() * 4);
    o->na = (int32_t *)dup(na.data(), na.size() * 4);
    o->n_cigar = (int32_t *)dup(n_cigar.data(), n_cigar.size() * 4);
    o->pos = (int64_t *)dup(pos.data(), pos.size() * 8);
    o->flag = (uint16_t *)dup(flag.data(), flag.size() * 2);
    o->mapq = (uint8_t *)dup(mapq.data(), mapq.size());
    o->cig_off = (int64_t *)dup(cig_off.data(), cig_off.size() * 8);
    o->cigar = (uint32_t *)dup(cigar.data(), cigar.size() * 4);
    o->hit_off = (int64_t *)dup(hit_off.data(), hit_off.size() * 8);
    return (int)o->n_hits;
}

extern "C" void orc_batch_free(orc_batch_out *o)
{
    std::free(o->read_idx); std::free(o->rid); std::free(o->score); std::free(o->nm); std::free(o->na);
    std::free(o->n_cigar); std::free(o->pos); std::free(o->flag); std::free(o->mapq); std::free(o->cig_off);
    std::free(o->cigar); std::free(o->hit_off);
    std::memset(o, 0, sizeof *o);
}

/* The CPU baseline of bench.py as SURVEY 8d(ii) defines it: ONE process, n_threads std::threads over chunks of 64 reads (handed
 * out by an atomic counter) sharing one read-only index, each read one orc_align_sequence call (the reference's calling convention).  Returns the
 * wall time from the start of the first thread to the join of the last; thread_secs[t] (if not null) = thread t's own time. */
extern "C" double orc_time_batch_mt(const orc_opt *opt, const orc_index *idx, const char *bases, const uint64_t *offs, int64_t n_reads,
                                    int n_threads, int hardclip, double keepSecFrac, int maxSecondary, uint64_t rng_base,
                                    uint64_t first_ordinal, int64_t *n_hits_out, double *thread_secs)
{
    if (n_threads < 1) n_threads = 1;
    std::vector<int64_t> hits((size_t)n_threads, 0);
    std::vector<double> secs((size_t)n_threads, 0.0);
    // dynamic chunks of 64 reads from one atomic counter: repeat reads cluster, and with static ranges the slowest thread set the time
    std::atomic<int64_t> next(0);
    auto work = [&](int t) {
        const auto t0 = std::chrono::steady_clock::now();
        int64_t nh = 0;
        for (;;) {
            const int64_t a = next.fetch_add(64), b = a + 64 < n_reads ? a + 64 : n_reads;
            if (a >= n_reads) break;
            for (int64_t r = a; r < b; ++r) {
                orc_hit *h = nullptr;
                const int n = orc_align_sequence(opt, idx, bases + offs[r], (int)(offs[r + 1] - offs[r]), "r", hardclip, keepSecFrac, maxSecondary,
                                                 rng_base, first_ordinal + (uint64_t)r, &h);
                nh += n;
                orc_hits_free(h, n);
            }
        }
        hits[(size_t)t] = nh;
        secs[(size_t)t] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    const auto t0 = std::chrono::steady_clock::now();
    if (n_threads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t tot = 0;
    for (int t = 0; t < n_threads; ++t) { tot += hits[(size_t)t]; if (thread_secs) thread_secs[t] = secs[(size_t)t]; }
    if (n_hits_out) *n_hits_out = tot;
    return wall;
}

/* ---------------------------------------------------------------------------------------------- mem_reg2sam semantics
 * Restates bwamem.c:mem_reg2sam (without MEM_F_ALL / MEM_F_NO_MULTI, which SeqLib's API cannot set), bwamem_extra.c:mem_gen_alt
 * and the SA tag of bwamem.c:mem_aln2sam. */
static int get_pri_idx(double XA_drop_ratio, const orc_reg *a, int i)
{
    int k = a[i].secondary_all;
    if (k >= 0 && a[i].score >= a[k].score * XA_drop_ratio) return k;
    return -1;
}

/* bwa.c: bwa_gen_cigar2's MD string, from the finished alignment: match run lengths, a mismatch as the reference base, a deletion as
 * ^ + the deleted reference bases; insertions and clips leave no trace; a deletion that was the first or the last operation is not in
 * the CIGAR any more (mem_reg2aln squeezed it out, as bwa skips it in MD).  bwa walks the reverse-strand alignments in reversed
 * coordinates with complemented letters: the same comparisons and the same letters as the forward strand against the read as SAM shows it. */
static std::string md_string(const orc_index *idx, const char *seq, int len, int rid, int64_t pos, bool is_rev, const uint32_t *cig, int n_cig)
{
    auto ref_at = [&](int64_t p) { return (int)(idx->pac[p >> 2] >> ((~p & 3) << 1) & 3); };
    auto qcode = [&](int x) {          /* base x of the read as the record shows it (reverse strand: the reverse complement) */
        const int c = is_rev ? seq[len - 1 - x] : seq[x];
        int v;
        switch (c) { case 'A': case 'a': v = 0; break; case 'C': case 'c': v = 1; break; case 'G': case 'g': v = 2; break; case 'T': case 't': v = 3; break; default: v = 4; }
        return is_rev && v < 4 ? 3 - v : v;
    };
    std::string md;
    int x = 0, u = 0;
    int64_t y = idx->anns[rid].offset + pos;
    for (int k = 0; k < n_cig; ++k) {
        const int op = (int)(cig[k] & 0xf), l = (int)(cig[k] >> 4);
        if (op == 0) {
            for (int i = 0; i < l; ++i) {
                if (qcode(x + i) != ref_at(y + i)) { md += std::to_string(u); md.push_back("ACGT"[ref_at(y + i)]); u = 0; }
                else ++u;
            }
            x += l; y += l;
        } else if (op == 2) {
            md += std::to_string(u); md.push_back('^');
            for (int i = 0; i < l; ++i) md.push_back("ACGT"[ref_at(y + i)]);
            u = 0; y += l;
        } else if (op == 1 || op == 3 || op == 4) x += l;          /* insertion, soft clip (bwa's op 3 / BAM 4); a hard clip (5) consumes nothing of the record's sequence -- but `seq` here is the whole read */
        else if (op == 5) x += l;
    }
    md += std::to_string(u);
    return md;
}

static void put_cigar(std::string &s, const uint32_t *cig, int n, const char *ops)
{
    for (int k = 0; k < n; ++k) { s += std::to_string(cig[k] >> 4); s.push_back(ops[cig[k] & 0xf]); }
}

extern "C" int orc_align_sequence_sam(const orc_opt *opt, const orc_index *idx, const char *seq, int len, int hardclip,
                                      uint64_t rng_base, uint64_t ordinal, orc_samhit **out)
{
    *out = nullptr;
    if (!idx) return 0;
    const uint64_t salt = orc_lrand48_nth(rng_base, ordinal + 1);
    orc_reg *a = nullptr;
    const int n = orc_align1(opt, idx, len, seq, salt, &a);
    /* mem_gen_alt: which secondaries become XA alternatives of which primary */
    std::vector<int> cnt((size_t)n, 0), pri((size_t)n, -1);
    std::vector<char> has_alt((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        const int r = get_pri_idx(opt->XA_drop_ratio, a, i);
        pri[(size_t)i] = r;
        if (r >= 0) { ++cnt[(size_t)r]; if (a[i].is_alt) has_alt[(size_t)r] = 1; }
    }
    /* mem_reg2sam: the records */
    struct Rec { int k; orc_aln aln; int flag; int mapq; int sub; };
    std::vector<Rec> recs;
    std::vector<int> rec_of((size_t)n, -1);
    for (int k = 0; k < n; ++k) {
        const orc_reg *p = &a[k];
        if (p->score < opt->T) continue;
        if (p->secondary >= 0) continue;                     /* (p->is_alt || !(opt->flag & MEM_F_ALL)) holds: MEM_F_ALL is never set */
        Rec q;
        q.k = k;
        q.aln = orc_reg2aln(opt, idx, len, seq, p);
        q.flag = q.aln.flag | (q.aln.is_rev ? 0x10 : 0);
        q.mapq = (int)q.aln.mapq;
        q.sub = q.aln.sub;
        if (!recs.empty()) q.flag |= 0x800;                  /* supplementary */
        if (!recs.empty() && !p->is_alt && q.mapq > recs[0].mapq) q.mapq = recs[0].mapq;
        rec_of[(size_t)k] = (int)recs.size();
        recs.push_back(q);
    }
    /* which regions are XA alternatives, of which record, and the string bwa appends to XA[r] for them (region order) */
    std::vector<std::string> xa(recs.size());
    std::vector<int> alt_parent((size_t)n, -1);
    for (int i = 0; i < n; ++i) {
        const int r = pri[(size_t)i];
        if (r < 0) continue;
        if (cnt[(size_t)r] > opt->max_XA_hits_alt || (!has_alt[(size_t)r] && cnt[(size_t)r] > opt->max_XA_hits)) continue;
        if (rec_of[(size_t)r] < 0) continue;                 /* its primary is not printed: neither is its XA */
        alt_parent[(size_t)i] = rec_of[(size_t)r];
        orc_aln t = orc_reg2aln(opt, idx, len, seq, &a[i]);
        std::string &x = xa[(size_t)rec_of[(size_t)r]];
        x += idx->anns[t.rid].name;
        x.push_back(',');
        x.push_back("+-"[t.is_rev]);
        x += std::to_string((long long)t.pos + 1);
        x.push_back(',');
        put_cigar(x, t.cigar, t.n_cigar, "MIDSHN");
        x.push_back(',');
        x += std::to_string((int)t.NM);
        x.push_back(';');
        std::free(t.cigar);
    }
    /* entries in region order: a region that is a record, an alternative, or (ALT-aware index) both */
    std::vector<orc_samhit> o;
    for (int k = 0; k < n; ++k) {
        const int j = rec_of[(size_t)k];
        if (j < 0 && alt_parent[(size_t)k] < 0) continue;
        orc_aln al = j >= 0 ? recs[(size_t)j].aln : orc_reg2aln(opt, idx, len, seq, &a[k]);
        orc_samhit h;
        std::memset(&h, 0, sizeof h);
        h.rid = al.rid; h.pos = al.pos; h.score = al.score; h.nm = (int32_t)al.NM; h.na = n; h.n_cigar = al.n_cigar;
        h.flag = (uint16_t)(j >= 0 ? recs[(size_t)j].flag : (al.flag | (al.is_rev ? 0x10 : 0)));
        h.mapq = (uint8_t)(j >= 0 ? recs[(size_t)j].mapq : (int)al.mapq);
        h.sub = j >= 0 ? recs[(size_t)j].sub : -1;
        h.xa_parent = alt_parent[(size_t)k];
        h.cigar = (uint32_t *)std::malloc(4 * (size_t)(al.n_cigar ? al.n_cigar : 1));
        for (int c = 0; c < al.n_cigar; ++c) {
            uint32_t w = al.cigar[c];
            if ((w & 0xf) == 3) w = (w & ~0xfu) | (hardclip ? 5u : 4u);
            h.cigar[c] = w;
        }
        if (j >= 0) {
            h.md = strdup(md_string(idx, seq, len, al.rid, al.pos, al.is_rev, h.cigar, al.n_cigar).c_str());
            if (!xa[(size_t)j].empty()) h.xa = strdup(xa[(size_t)j].c_str());
            if (recs.size() > 1) {                           /* mem_aln2sam: the other non-secondary hits of the list */
                std::string sa;
                for (size_t i = 0; i < recs.size(); ++i) {
                    if ((int)i == j) continue;
                    const orc_aln &r = recs[i].aln;
                    sa += idx->anns[r.rid].name; sa.push_back(',');
                    sa += std::to_string((long long)r.pos + 1); sa.push_back(',');
                    sa.push_back("+-"[r.is_rev]); sa.push_back(',');
                    put_cigar(sa, r.cigar, r.n_cigar, "MIDSH");
                    sa.push_back(','); sa += std::to_string(recs[i].mapq);
                    sa.push_back(','); sa += std::to_string((int)r.NM);
                    sa.push_back(';');
                }
                h.sa = strdup(sa.c_str());
            }
        } else std::free(al.cigar);
        o.push_back(h);
    }
    for (Rec &q : recs) std::free(q.aln.cigar);
    std::free(a);
    if (!o.empty()) {
        *out = (orc_samhit *)std::malloc(o.size() * sizeof(orc_samhit));
        std::memcpy(*out, o.data(), o.size() * sizeof(orc_samhit));
    }
    return (int)o.size();
}

extern "C" void orc_samhits_free(orc_samhit *h, int n)
{
    if (!h) return;
    for (int i = 0; i < n; ++i) { std::free(h[i].cigar); std::free(h[i].xa); std::free(h[i].sa); std::free(h[i].md); }
    std::free(h);
}
