// hts_compat.h -- the slice of htslib's sam.h that the BWAAligner path touches, so that the drop-in
// headers build without htslib (not installed in this image).  Layout and macro
// names follow htslib >= 1.10 so that code written against <htslib/sam.h> for these members compiles
// unchanged.  If the real htslib is on the include path, define SEQLIB_AMD_USE_HTSLIB to use it instead.
#pragma once
#ifdef SEQLIB_AMD_USE_HTSLIB
extern "C" {
#include "htslib/sam.h"
}
#else
#include <cstdint>
#include <cstdlib>
#include <cstring>

typedef int64_t hts_pos_t;
typedef struct bam1_core_t {
    hts_pos_t pos;
    int32_t tid;
    uint16_t bin;
    uint8_t qual;
    uint8_t l_extranul;
    uint16_t flag;
    uint16_t l_qname;
    uint32_t n_cigar;
    int32_t l_qseq;
    int32_t mtid;
    hts_pos_t mpos;
    hts_pos_t isize;
} bam1_core_t;
typedef struct bam1_t {
    bam1_core_t core;
    uint64_t id;
    uint8_t *data;
    int l_data;
    uint32_t m_data;
    uint32_t mempolicy : 2, : 30;
} bam1_t;

#define BAM_CMATCH 0
#define BAM_CINS 1
#define BAM_CDEL 2
#define BAM_CREF_SKIP 3
#define BAM_CSOFT_CLIP 4
#define BAM_CHARD_CLIP 5
#define BAM_CPAD 6
#define BAM_CEQUAL 7
#define BAM_CDIFF 8
#define BAM_CBACK 9
#define BAM_CIGAR_STR "MIDNSHP=XB"
#define BAM_CIGAR_SHIFT 4
#define BAM_CIGAR_MASK 0xf
#define BAM_CIGAR_TYPE 0x3C1A7
#define bam_cigar_op(c) ((c) & BAM_CIGAR_MASK)
#define bam_cigar_oplen(c) ((c) >> BAM_CIGAR_SHIFT)
#define bam_cigar_opchr(c) (BAM_CIGAR_STR "??????"[bam_cigar_op(c)])
#define bam_cigar_gen(l, o) ((l) << BAM_CIGAR_SHIFT | (o))
#define bam_cigar_type(o) (BAM_CIGAR_TYPE >> ((o) << 1) & 3)

#define BAM_FPAIRED 1
#define BAM_FPROPER_PAIR 2
#define BAM_FUNMAP 4
#define BAM_FMUNMAP 8
#define BAM_FREVERSE 16
#define BAM_FMREVERSE 32
#define BAM_FREAD1 64
#define BAM_FREAD2 128
#define BAM_FSECONDARY 256
#define BAM_FQCFAIL 512
#define BAM_FDUP 1024
#define BAM_FSUPPLEMENTARY 2048

#define bam_get_qname(b) ((char *)(b)->data)
#define bam_get_cigar(b) ((uint32_t *)((b)->data + (b)->core.l_qname))
#define bam_get_seq(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname)
#define bam_get_qual(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1))
#define bam_get_aux(b) ((b)->data + ((b)->core.n_cigar << 2) + (b)->core.l_qname + (((b)->core.l_qseq + 1) >> 1) + (b)->core.l_qseq)
#define bam_get_l_aux(b) ((b)->l_data - ((b)->core.n_cigar << 2) - (b)->core.l_qname - (b)->core.l_qseq - (((b)->core.l_qseq + 1) >> 1))
#define bam_seqi(s, i) ((s)[(i) >> 1] >> ((~(i) & 1) << 2) & 0xf)

static inline bam1_t *bam_init1(void) { return (bam1_t *)calloc(1, sizeof(bam1_t)); }
/* htslib's bam1_t::mempolicy bits (sam.h: bam_set_mempolicy): memory the caller owns is not freed by bam_destroy1, and data the caller owns is never
 * realloc'ed -- a record that outgrows it moves to malloc'ed memory of its own (sam_realloc_bam_data).  The batch path hands out records whose data lie
 * in slabs shared by thousands of records (BamRecord.h, detail::Slab). */
#define BAM_USER_OWNS_STRUCT 1
#define BAM_USER_OWNS_DATA 2
static inline void bam_destroy1(bam1_t *b)
{
    if (!b) return;
    if (!(b->mempolicy & BAM_USER_OWNS_DATA)) free(b->data);
    if (!(b->mempolicy & BAM_USER_OWNS_STRUCT)) free(b);
}
static inline int bam_aux_type2size(uint8_t t)
{
    switch (t) { case 'A': case 'c': case 'C': return 1; case 's': case 'S': return 2; case 'i': case 'I': case 'f': return 4; case 'd': return 8; default: return 0; }
}
static inline uint8_t *bam_aux_skip(uint8_t *s, uint8_t *end)
{   // s points at the type byte
    if (s >= end) return end;
    uint8_t t = *s++;
    int sz = bam_aux_type2size(t);
    if (sz) return s + sz <= end ? s + sz : end;
    if (t == 'Z' || t == 'H') { while (s < end && *s) ++s; return s < end ? s + 1 : end; }
    if (t == 'B') {
        if (end - s < 5) return end;
        int esz = bam_aux_type2size(*s);
        uint32_t n; memcpy(&n, s + 1, 4);
        s += 5 + (size_t)esz * n;
        return s <= end ? s : end;
    }
    return end;
}
static inline uint8_t *bam_aux_get(const bam1_t *b, const char tag[2])
{
    uint8_t *s = bam_get_aux(b), *end = b->data + b->l_data;
    while (s && end - s >= 3) {
        if (s[0] == (uint8_t)tag[0] && s[1] == (uint8_t)tag[1]) return s + 2;   // pointer to the type byte
        s = bam_aux_skip(s + 2, end);
    }
    return 0;
}
static inline int bam_aux_del(bam1_t *b, uint8_t *s)
{   // s = pointer returned by bam_aux_get
    uint8_t *end = b->data + b->l_data, *p = s - 2, *next = bam_aux_skip(s, end);
    memmove(p, next, (size_t)(end - next));
    b->l_data -= (int)(next - p);
    return 0;
}
static inline int bam_aux_append(bam1_t *b, const char tag[2], char type, int len, const uint8_t *data)
{
    uint32_t need = (uint32_t)b->l_data + 3 + (uint32_t)len;
    if (b->m_data < need) {
        uint32_t m = need; --m; m |= m >> 1; m |= m >> 2; m |= m >> 4; m |= m >> 8; m |= m >> 16; ++m;   // kroundup32
        uint8_t *nd;
        if (b->mempolicy & BAM_USER_OWNS_DATA) {          /* sam_realloc_bam_data: the caller's memory is left alone, the record gets its own */
            nd = (uint8_t *)malloc(m);
            if (!nd) return -1;
            if (b->l_data > 0) memcpy(nd, b->data, (size_t)b->l_data);
            b->mempolicy &= ~(uint32_t)BAM_USER_OWNS_DATA;
        } else {
            nd = (uint8_t *)realloc(b->data, m);
            if (!nd) return -1;
        }
        b->data = nd; b->m_data = m;
    }
    b->data[b->l_data] = (uint8_t)tag[0]; b->data[b->l_data + 1] = (uint8_t)tag[1]; b->data[b->l_data + 2] = (uint8_t)type;
    memcpy(b->data + b->l_data + 3, data, (size_t)len);
    b->l_data = (int)need;
    return 0;
}
static inline int64_t bam_aux2i(const uint8_t *s)
{
    uint8_t t = *s++;
    switch (t) {
    case 'c': return (int8_t)*s;
    case 'C': return *s;
    case 's': { int16_t v; memcpy(&v, s, 2); return v; }
    case 'S': { uint16_t v; memcpy(&v, s, 2); return v; }
    case 'i': { int32_t v; memcpy(&v, s, 4); return v; }
    case 'I': { uint32_t v; memcpy(&v, s, 4); return v; }
    default: return 0;
    }
}
static inline char *bam_aux2Z(const uint8_t *s) { return (*s == 'Z' || *s == 'H') ? (char *)(s + 1) : 0; }
static inline hts_pos_t bam_endpos(const bam1_t *b)
{
    hts_pos_t r = b->core.pos;
    const uint32_t *c = bam_get_cigar(b);
    int any = 0;
    for (uint32_t k = 0; k < b->core.n_cigar; ++k)
        if (bam_cigar_type(bam_cigar_op(c[k])) & 2) { r += bam_cigar_oplen(c[k]); any = 1; }
    return any ? r : b->core.pos + 1;
}
#endif
