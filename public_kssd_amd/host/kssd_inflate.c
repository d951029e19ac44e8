/* kssd_inflate.c -- gzip members in memory -> their bytes (RFC 1951 / 1952), written for the inputs of `kssd dist`.
 *
 * The reference reads every input through popen("zcat -fc <file>") (iseq2comem.c:187,196-208): one zcat process per file, a pipe, 64 KiB
 * freads.  Round 1 - 4 of this build used zlib's gzread instead, and a directory of .fasta.gz genomes -- the form the reference's own
 * test data comes in -- then spends 75 % of the command inside zlib's byte-at-a-time inflate on the host threads (0.3 GB/s of text
 * per thread).  Sequence text is the worst case for that loop and the best case for a table-driven one: four or five symbols with
 * codes of two or three bits, hardly any match.  This decoder reads the stream through a 64-bit bit buffer that is refilled
 * eight bytes at a time, and its first-level table (11 bits) answers up to THREE literals per lookup; lengths and distances take
 * one lookup each (second-level tables for the few codes longer than the first level).  The check value of a member (CRC-32) is
 * computed with carry-less multiplications where the CPU has them (zlib 1.2.11's table loop runs at 1 GB/s: slower than the decoder).
 * Own code throughout; what is reused are the published formats and the published folding constants of the CRC-32 polynomial. */
#include "kssd_host.h"

#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h> /* crc32() as the fallback check */

#if defined(__x86_64__)
#include <immintrin.h>
#endif

/* ---- table entries ------------------------------------------------------------------------------------------------------------
 * Laid out for the instructions the symbol loop spends on an entry (two files in step run at the core's instruction rate):
 * bits 0-7   ALL the bits the entry consumes, a byte of its own: the bit buffer is shifted by the entry as it is (the shift takes its
 *            count from the low six bits), the bit count subtracts the entry as it is (only its low byte is ever looked at), and the
 *            mask of a length's or distance's extra bits takes its width from it -- a literal's code(s); a length's or distance's code
 *            AND the extra bits behind it; behind a first-level pointer the second-level part of the code (+ extra bits); for the
 *            pointer itself, the first level's bits
 * bits 8-11  a length / distance: the code's own bits (its extra bits are what bits 0-7 count beyond them); a pointer: the second
 *            level's index bits
 * bit 12     F_LIT: one literal byte in bits 16-23 -- bit 13, F_LIT2: and a second one in bits 24-31
 * bit 14     F_SUB: pointer to a second-level table, its first entry in bits 16-31
 * bit 15     F_EXC: the end-of-block code (bits 16-31 = 1) or a code nobody owns (0)
 * none of bits 12-15: a length or distance, its base in bits 16-31 */
#define F_LIT 0x1000u
#define F_LIT2 0x2000u
#define F_SUB 0x4000u
#define F_EXC 0x8000u
#define ENTRY_LIT1(nbits, b0) ((uint32_t)(nbits) | F_LIT | ((uint32_t)(b0) << 16))
#define ENTRY_LIT2(nbits, b0, b1) ((uint32_t)(nbits) | F_LIT | F_LIT2 | ((uint32_t)(b0) << 16) | ((uint32_t)(b1) << 24))
#define ENTRY_BASE(total, code_bits, base) ((uint32_t)(total) | ((uint32_t)(code_bits) << 8) | ((uint32_t)(base) << 16))
#define ENTRY_SUB(main_bits, index_bits, first) ((uint32_t)(main_bits) | ((uint32_t)(index_bits) << 8) | F_SUB | ((uint32_t)(first) << 16))
#define ENTRY_EOB(nbits) ((uint32_t)(nbits) | F_EXC | (1u << 16))
#define ENTRY_BAD F_EXC
#define E_TOTAL(e) ((e) & 255u)
#define E_CODE(e) (((e) >> 8) & 15u)
#define E_PAYLOAD(e) ((e) >> 16)
#define E_IS_BASE(e) (((e) & 0xF000u) == 0)
/* a length's / distance's value: base + the extra bits, which sit behind the code in the low E_TOTAL bits of the bit buffer (the
 * shift by `(e >> 8) & 63` is the shift by E_CODE: bits 12 and 13 of such an entry are clear) */
#define E_VALUE(e, bb) (E_PAYLOAD(e) + (uint32_t)(((bb) & (((uint64_t)1 << E_TOTAL(e)) - 1)) >> (((e) >> 8) & 63u)))

#define LIT_BITS 11
#define DST_BITS 8
#define LIT_ROOM ((1 << LIT_BITS) + (1 << 15)) /* first level + every second level a set of codes can ask for */
#define DST_ROOM ((1 << DST_BITS) + (1 << 12))

typedef struct {
    uint32_t lit[LIT_ROOM];
    uint32_t lit_one[1 << LIT_BITS];   /* the first level without literal groups: what the careful path (one symbol at a time) looks up */
    uint32_t dst[DST_ROOM];
    uint32_t fixed_lit[1 << LIT_BITS]; /* the fixed codes (at most 9 bits: no second level) */
    uint32_t fixed_lit_one[1 << LIT_BITS];
    uint32_t fixed_dst[1 << DST_BITS];
    int fixed_ready;
} inflate_tabs;

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dst_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dst_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static uint32_t rev_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1 - i);
    return r;
}

/* what symbol `sym` of the literal/length (dist = 0) or distance (dist = 1) alphabet decodes to, in a code of nbits bits */
static uint32_t leaf(int dist, int sym, int nbits)
{
    if (dist) return sym < 30 ? ENTRY_BASE(nbits + dst_extra[sym], nbits, dst_base[sym]) : ENTRY_BAD;
    if (sym < 256) return ENTRY_LIT1(nbits, sym);
    if (sym == 256) return ENTRY_EOB(nbits);
    if (sym < 286) return ENTRY_BASE(nbits + len_extra[sym - 257], nbits, len_base[sym - 257]);
    return ENTRY_BAD;
}

/* canonical code lengths -> decoding tables.  -1: the lengths oversubscribe the code space (or need more room than there is) */
static int build_tables(const uint8_t *lens, int n_sym, int dist, uint32_t *tab, int main_bits, int room, uint32_t *single /* literal/length: the first level before grouping */)
{
    int count[16] = {0};
    for (int s = 0; s < n_sym; s++) count[lens[s]]++;
    count[0] = 0;
    int left = 1;
    uint32_t next_code[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; l++) {
        left = left * 2 - count[l];
        if (left < 0) return -1;
        code = (code + (uint32_t)count[l - 1]) << 1;
        next_code[l] = code;
    }
    const int main_n = 1 << main_bits;
    for (int i = 0; i < main_n; i++) tab[i] = ENTRY_BAD; /* (an incomplete set: a code nobody owns is an error when it is met) */
    uint8_t sub_bits[1 << LIT_BITS];
    memset(sub_bits, 0, (size_t)main_n);
    uint16_t rev[288];
    for (int s = 0; s < n_sym; s++) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev_bits(next_code[l]++, l);
        rev[s] = (uint16_t)r;
        if (l <= main_bits) {
            const uint32_t e = leaf(dist, s, l);
            for (uint32_t i = r; i < (uint32_t)main_n; i += 1u << l) tab[i] = e;
        } else {
            const uint32_t p = r & (uint32_t)(main_n - 1);
            if (l - main_bits > sub_bits[p]) sub_bits[p] = (uint8_t)(l - main_bits);
        }
    }
    int next = main_n;
    for (int p = 0; p < main_n; p++) {
        if (!sub_bits[p]) continue;
        if (next + (1 << sub_bits[p]) > room) return -1;
        tab[p] = ENTRY_SUB(main_bits, sub_bits[p], next);
        for (int i = 0; i < (1 << sub_bits[p]); i++) tab[next + i] = ENTRY_BAD;
        next += 1 << sub_bits[p];
    }
    for (int s = 0; s < n_sym; s++) {
        const int l = lens[s];
        if (l <= main_bits) continue;
        const uint32_t p = rev[s] & (uint32_t)(main_n - 1), hi = rev[s] >> main_bits;
        const uint32_t base = E_PAYLOAD(tab[p]), sb = E_CODE(tab[p]);
        const uint32_t e = leaf(dist, s, l - main_bits);
        for (uint32_t i = hi; i < (1u << sb); i += 1u << (l - main_bits)) tab[base + i] = e;
    }
    if (!dist) {
        /* literal pairs: a first-level slot whose bits hold two whole literal codes answers both at once */
        memcpy(single, tab, (size_t)main_n * sizeof(uint32_t));
        for (int i = 0; i < main_n; i++) {
            const uint32_t e1 = single[i];
            if (!(e1 & F_LIT)) continue;
            const uint32_t n1 = E_TOTAL(e1);
            const uint32_t e2 = single[(uint32_t)i >> n1]; /* the unknown bits above read as zeros: right for every code that fits the known ones */
            const uint32_t n2 = E_TOTAL(e2);
            if ((e2 & F_LIT) && n1 + n2 <= (uint32_t)main_bits) tab[i] = ENTRY_LIT2(n1 + n2, E_PAYLOAD(e1) & 255u, E_PAYLOAD(e2) & 255u);
        }
    }
    return 0;
}

/* ---- the bit reader -------------------------------------------------------------------------------------------------------------- */
typedef struct {
    const unsigned char *in, *in_end;
    uint64_t bb; /* bits not consumed yet, the next one at bit 0 */
    int bc;      /* how many of them are real */
} bitr;

static inline uint64_t load64(const unsigned char *p)
{
    uint64_t v;
    memcpy(&v, p, 8);
    return v; /* (x86-64, little endian: the build's only host) */
}

/* careful refill (the stream's last bytes): whole bytes while they fit */
static inline void refill_safe(bitr *b)
{
    while (b->bc < 56 && b->in < b->in_end) { /* (at most 63 bits: the fast path shifts by the count) */
        b->bb |= (uint64_t)*b->in++ << b->bc;
        b->bc += 8;
    }
}
static inline int take_bits(bitr *b, int n, uint32_t *v)
{
    if (b->bc < n) {
        refill_safe(b);
        if (b->bc < n) return -1; /* the stream ends inside a field */
    }
    *v = (uint32_t)(b->bb & ((1ull << n) - 1));
    b->bb >>= n;
    b->bc -= n;
    return 0;
}

typedef struct {
    unsigned char *out; /* grown with realloc */
    size_t cap, len;
} outbuf;

static int out_room(outbuf *o, size_t more)
{
    if (o->len + more <= o->cap) return 0;
    size_t nc = o->cap + o->cap / 2 + more + (1u << 20);
    unsigned char *q = (unsigned char *)realloc(o->out, nc);
    if (!q) return -1;
    o->out = q;
    o->cap = nc;
    return 0;
}

#define FAST_OUT_MARGIN 336 /* the longest match (258) + what the copies and the literal stores may write beyond their last byte */

/* one block's symbols (its tables are built): returns 0 at the end-of-block code, -1 on a corrupt stream, -2 out of memory.
 *
 * Sequence text deflates into MATCHES, not literals: zlib -1 turns a bacterial genome into 1.3 M matches of four bases on average
 * and a few dozen literals, -6 into 0.7 M matches of seven and 0.2 M literals.  What a match costs is the chain of dependent
 * steps from one table entry to the next: refill -> look up -> shift.  So a match is ONE refill (56 bits hold the longest length
 * and distance: 15 + 5 + 15 + 13), each entry is ONE shift (its count covers the extra bits too, the value is picked off the side),
 * and the copy hangs off the chain (nothing it produces is needed to decode on). */
/* the symbol loops are built twice, with and without BMI2 (shifts and masks by a register's count in one instruction, no detour
 * through %cl), and the loader picks by the CPU */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define HOT_CLONES __attribute__((target_clones("bmi2", "default")))
#else
#define HOT_CLONES
#endif

/* one turn of the fast loop on the stream whose locals end in S: literals (up to three lookups of up to two each) or a match.
 * `break` leaves the turn; the end-of-block code and a corrupt stream set rc##S and leave the loop (label `stop`).  bc##S: only its
 * low byte is the bit count (whole entries are subtracted from it). */
#define REFILL(S)                                       \
    do {                                                \
        bb##S |= load64(in##S) << (bc##S & 63);         \
        in##S += 7 - ((bc##S >> 3) & 7);                \
        bc##S |= 56;                                    \
    } while (0)
#define CONSUME(S, e)                                   \
    do {                                                \
        bb##S >>= (e) & 63;                             \
        bc##S -= (e);                                   \
    } while (0)
#define PUT_GROUP(S)                                    \
    do {                                                \
        const uint16_t w = (uint16_t)E_PAYLOAD(e);      \
        memcpy(out##S, &w, 2);                          \
        out##S += 1 + ((e >> 13) & 1);                  \
        CONSUME(S, e);                                  \
    } while (0)
#define STEP(S)                                                                                                                     \
    do {                                                                                                                            \
        REFILL(S); /* eight bytes over the top of what is left, whole bytes accepted: 56 - 63 real bits afterwards */               \
        uint32_t e = lit##S[bb##S & lmask];                                                                                         \
        int behind_literals = 0;                                                                                                    \
        if (e & F_LIT) {                                                                                                            \
            PUT_GROUP(S);                                                                                                           \
            e = lit##S[bb##S & lmask];                                                                                              \
            if (e & F_LIT) {                                                                                                        \
                PUT_GROUP(S);                                                                                                       \
                e = lit##S[bb##S & lmask];                                                                                          \
                if (e & F_LIT) {                                                                                                    \
                    PUT_GROUP(S);                                                                                                   \
                    break;                                                                                                          \
                }                                                                                                                   \
            }                                                                                                                       \
            behind_literals = 1; /* up to 22 bits are gone: a length (20) still fits, the distance behind it (28) may not */        \
        }                                                                                                                           \
        if (__builtin_expect(!E_IS_BASE(e), 0)) { /* (no literal here: a pointer, the end of the block, nobody's code) */           \
            if (e & F_SUB) {                                                                                                        \
                CONSUME(S, e);                                                                                                      \
                e = lit##S[E_PAYLOAD(e) + (bb##S & ((1u << E_CODE(e)) - 1))];                                                       \
                if (e & F_LIT) { /* (a literal with a long code) */                                                                 \
                    *out##S++ = (unsigned char)E_PAYLOAD(e);                                                                        \
                    CONSUME(S, e);                                                                                                  \
                    break;                                                                                                          \
                }                                                                                                                   \
            }                                                                                                                       \
            if (e & F_EXC) {                                                                                                        \
                CONSUME(S, e);                                                                                                      \
                rc##S = E_PAYLOAD(e) ? 0 : -1;                                                                                      \
                goto stop;                                                                                                          \
            }                                                                                                                       \
        }                                                                                                                           \
        const uint32_t len = E_VALUE(e, bb##S);                                                                                     \
        CONSUME(S, e);                                                                                                              \
        if (behind_literals) REFILL(S);                                                                                             \
        uint32_t d = dst##S[bb##S & dmask];                                                                                         \
        if (__builtin_expect(!E_IS_BASE(d), 0)) {                                                                                   \
            if (d & F_SUB) {                                                                                                        \
                CONSUME(S, d);                                                                                                      \
                d = dst##S[E_PAYLOAD(d) + (bb##S & ((1u << E_CODE(d)) - 1))];                                                       \
            }                                                                                                                       \
            if (!E_IS_BASE(d)) { rc##S = -1; goto stop; } /* (the distance tables hold no literals and no end of block) */          \
        }                                                                                                                           \
        const uint32_t dist = E_VALUE(d, bb##S);                                                                                    \
        CONSUME(S, d);                                                                                                              \
        if (__builtin_expect((size_t)(out##S - out_min##S) < dist, 0)) { rc##S = -1; goto stop; } /* before the member's first byte */ \
        const unsigned char *src = out##S - dist;                                                                                   \
        unsigned char *const end = out##S + len;                                                                                    \
        if (__builtin_expect(dist >= 16, 1)) { /* the common case first, without a loop: a match of up to 16 bytes is one copy */    \
            memcpy(out##S, src, 16);                                                                                                \
            if (__builtin_expect(len > 16, 0)) {                                                                                    \
                out##S += 16;                                                                                                       \
                src += 16;                                                                                                          \
                do {                                                                                                                \
                    memcpy(out##S, src, 16);                                                                                        \
                    out##S += 16;                                                                                                   \
                    src += 16;                                                                                                      \
                } while (out##S < end);                                                                                             \
            }                                                                                                                       \
        } else if (dist >= 8) {                                                                                                     \
            do {                                                                                                                    \
                memcpy(out##S, src, 8);                                                                                             \
                out##S += 8;                                                                                                        \
                src += 8;                                                                                                           \
            } while (out##S < end);                                                                                                 \
        } else if (dist == 1) {                                                                                                     \
            memset(out##S, *src, len);                                                                                              \
        } else {                                                                                                                    \
            do { *out##S++ = *src++; } while (out##S < end);                                                                        \
        }                                                                                                                           \
        out##S = end;                                                                                                               \
    } while (0)
/* the locals of a stream inside a fast loop, and their way back into the stream's reader and output */
#define FAST_LOCALS(S, b, o, lit_, dst_, member_start)                                                                               \
    const unsigned char *in##S = (b)->in, *const in_fast##S = (b)->in_end - 16;                                                     \
    unsigned char *out##S = (o)->out + (o)->len, *const out_fast##S = (o)->out + (o)->cap - FAST_OUT_MARGIN;                        \
    unsigned char *const out_min##S = (o)->out + (member_start);                                                                    \
    const uint32_t *const lit##S = (lit_), *const dst##S = (dst_);                                                                  \
    uint64_t bb##S = (b)->bb;                                                                                                       \
    uint32_t bc##S = (uint32_t)(b)->bc
#define FAST_STORE(S, b, o) /* (the bit buffer may hold bytes beyond what the symbols used: bc counts the real bits only) */       \
    do {                                                                                                                            \
        (b)->in = in##S;                                                                                                            \
        (b)->bb = bb##S;                                                                                                            \
        (b)->bc = (int)(bc##S & 255u);                                                                                                      \
        (o)->len = (size_t)(out##S - (o)->out);                                                                                     \
    } while (0)
#define FAST_FITS(b, o) ((b)->in_end - (b)->in >= 16 && (o)->cap - (o)->len >= FAST_OUT_MARGIN)

HOT_CLONES static int inflate_symbols(bitr *b, outbuf *o, const uint32_t *lit, const uint32_t *lit_one, const uint32_t *dst, size_t member_start)
{
    const uint64_t lmask = (1u << LIT_BITS) - 1, dmask = (1u << DST_BITS) - 1;
    for (;;) {
        /* ---- fast: at least 16 input bytes and FAST_OUT_MARGIN output bytes to spare, no check per symbol ---- */
        if (FAST_FITS(b, o)) {
            FAST_LOCALS(A, b, o, lit, dst, member_start);
            int rcA = 1; /* 1: ran out of margin, 0: end of block, -1: corrupt */
            while (inA <= in_fastA && outA <= out_fastA) STEP(A);
        stop:
            FAST_STORE(A, b, o);
            if (rcA <= 0) return rcA;
        }
        /* ---- careful: one symbol, every read and write checked ---- */
        if (out_room(o, FAST_OUT_MARGIN + 8) != 0) return -2;
        refill_safe(b); /* 56 bits and more unless the stream ends: whatever one entry consumes */
        uint32_t e = lit_one[b->bb & lmask]; /* (second levels are shared: they live behind the first level of `lit`) */
        if (e & F_SUB) {
            if (b->bc < LIT_BITS) return -1;
            b->bb >>= LIT_BITS;
            b->bc -= LIT_BITS;
            refill_safe(b);
            e = lit[E_PAYLOAD(e) + (b->bb & ((1u << E_CODE(e)) - 1))];
        }
        if ((int)E_TOTAL(e) > b->bc) return -1; /* the stream ends inside a code or its extra bits */
        if (e & F_LIT) {
            o->out[o->len++] = (unsigned char)E_PAYLOAD(e);
            b->bb >>= E_TOTAL(e);
            b->bc -= (int)E_TOTAL(e);
            continue;
        }
        if (e & F_EXC) {
            b->bb >>= E_TOTAL(e);
            b->bc -= (int)E_TOTAL(e);
            return E_PAYLOAD(e) ? 0 : -1;
        }
        const uint32_t len = E_VALUE(e, b->bb);
        b->bb >>= E_TOTAL(e);
        b->bc -= (int)E_TOTAL(e);
        refill_safe(b);
        uint32_t d = dst[b->bb & dmask];
        if (d & F_SUB) {
            if (b->bc < DST_BITS) return -1;
            b->bb >>= DST_BITS;
            b->bc -= DST_BITS;
            refill_safe(b);
            d = dst[E_PAYLOAD(d) + (b->bb & ((1u << E_CODE(d)) - 1))];
        }
        if (!E_IS_BASE(d) || (int)E_TOTAL(d) > b->bc) return -1;
        const uint32_t dist = E_VALUE(d, b->bb);
        b->bb >>= E_TOTAL(d);
        b->bc -= (int)E_TOTAL(d);
        if (o->len - member_start < dist) return -1;
        for (uint32_t i = 0; i < len; i++, o->len++) o->out[o->len] = o->out[o->len - dist];
    }
}

/* The same fast loop over TWO streams, a turn of each in turn.  A stream's turns form one chain of dependent steps (the next table
 * index is in the bits behind this entry's), some 25 cycles a match at half the instructions the core could retire in them; a
 * second, independent chain fills the other half.  Runs while BOTH streams have their margins, and stops when either leaves its
 * block: rc 0 = end of block, -1 = corrupt, 1 = out of margin (its caller goes on through inflate_symbols), 2 = only the other one stopped. */
HOT_CLONES static void inflate_symbols2(bitr *b0, outbuf *o0, const uint32_t *lit0, const uint32_t *dst0, size_t member_start0, int *rc0,
                             bitr *b1, outbuf *o1, const uint32_t *lit1, const uint32_t *dst1, size_t member_start1, int *rc1)
{
    const uint64_t lmask = (1u << LIT_BITS) - 1, dmask = (1u << DST_BITS) - 1;
    *rc0 = FAST_FITS(b0, o0) ? 2 : 1;
    *rc1 = FAST_FITS(b1, o1) ? 2 : 1;
    if (*rc0 == 1 || *rc1 == 1) return;
    FAST_LOCALS(A, b0, o0, lit0, dst0, member_start0);
    FAST_LOCALS(B, b1, o1, lit1, dst1, member_start1);
    int rcA = 2, rcB = 2;
    while (inA <= in_fastA && outA <= out_fastA && inB <= in_fastB && outB <= out_fastB) {
        STEP(A);
        STEP(B);
    }
    if (inA > in_fastA || outA > out_fastA) rcA = 1;
    if (inB > in_fastB || outB > out_fastB) rcB = 1;
stop:
    FAST_STORE(A, b0, o0);
    FAST_STORE(B, b1, o1);
    *rc0 = rcA;
    *rc1 = rcB;
}
#undef REFILL
#undef CONSUME
#undef PUT_GROUP
#undef STEP
#undef FAST_LOCALS
#undef FAST_STORE
#undef FAST_FITS

/* ---- blocks -------------------------------------------------------------------------------------------------------------------- */
static void fixed_tables(inflate_tabs *t)
{
    if (t->fixed_ready) return;
    uint8_t lens[288];
    for (int i = 0; i < 144; i++) lens[i] = 8;
    for (int i = 144; i < 256; i++) lens[i] = 9;
    for (int i = 256; i < 280; i++) lens[i] = 7;
    for (int i = 280; i < 288; i++) lens[i] = 8;
    build_tables(lens, 288, 0, t->fixed_lit, LIT_BITS, 1 << LIT_BITS, t->fixed_lit_one);
    for (int i = 0; i < 32; i++) lens[i] = 5;
    build_tables(lens, 32, 1, t->fixed_dst, DST_BITS, 1 << DST_BITS, NULL);
    t->fixed_ready = 1;
}

/* the code lengths of a dynamic block (RFC 1951 3.2.7) */
static int read_dynamic(bitr *b, inflate_tabs *t)
{
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint32_t hlit, hdist, hclen, v;
    if (take_bits(b, 5, &hlit) || take_bits(b, 5, &hdist) || take_bits(b, 4, &hclen)) return -1;
    hlit += 257;
    hdist += 1;
    hclen += 4;
    if (hlit > 286 || hdist > 30) return -1;
    uint8_t cl[19] = {0};
    for (uint32_t i = 0; i < hclen; i++) {
        if (take_bits(b, 3, &v)) return -1;
        cl[order[i]] = (uint8_t)v;
    }
    /* the code-length code: at most 7 bits, one flat table */
    uint16_t ct[128]; /* symbol << 4 | bits; 0 = no code */
    {
        int count[8] = {0};
        for (int s = 0; s < 19; s++) count[cl[s]]++;
        count[0] = 0;
        uint32_t next[8], code = 0;
        int left = 1;
        for (int l = 1; l <= 7; l++) {
            left = left * 2 - count[l];
            if (left < 0) return -1;
            code = (code + (uint32_t)count[l - 1]) << 1;
            next[l] = code;
        }
        memset(ct, 0, sizeof ct);
        for (int s = 0; s < 19; s++) {
            const int l = cl[s];
            if (!l) continue;
            const uint32_t r = rev_bits(next[l]++, l);
            for (uint32_t i = r; i < 128; i += 1u << l) ct[i] = (uint16_t)((s << 4) | l);
        }
    }
    uint8_t lens[286 + 30];
    uint32_t n = 0;
    const uint32_t total = hlit + hdist;
    while (n < total) {
        refill_safe(b);
        const uint16_t c = ct[b->bb & 127];
        const int l = c & 15, s = c >> 4;
        if (!l || l > b->bc) return -1;
        b->bb >>= l;
        b->bc -= l;
        if (s < 16) {
            lens[n++] = (uint8_t)s;
            continue;
        }
        uint32_t rep, val = 0;
        if (s == 16) {
            if (n == 0 || take_bits(b, 2, &rep)) return -1;
            rep += 3;
            val = lens[n - 1];
        } else if (s == 17) {
            if (take_bits(b, 3, &rep)) return -1;
            rep += 3;
        } else {
            if (take_bits(b, 7, &rep)) return -1;
            rep += 11;
        }
        if (n + rep > total) return -1;
        while (rep--) lens[n++] = (uint8_t)val;
    }
    if (lens[256] == 0) return -1; /* no end-of-block code */
    if (build_tables(lens, (int)hlit, 0, t->lit, LIT_BITS, LIT_ROOM, t->lit_one)) return -1;
    if (build_tables(lens + hlit, (int)hdist, 1, t->dst, DST_BITS, DST_ROOM, NULL)) return -1;
    return 0;
}

/* ---- CRC-32 (gzip's check value) -------------------------------------------------------------------------------------------------
 * Folding by carry-less multiplication (V. Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ
 * Instruction", Intel 2009): four 128-bit lanes folded 512 bits at a time, then down to 128, 64 and 32 bits; the constants are
 * x^n mod P for the bit-reflected polynomial 0x1DB710641.  Whatever the fold does not take (the first bytes up to a multiple of
 * 16, a buffer below 64 bytes) goes through zlib's crc32().  Checked against zlib's on random buffers by tests/test_inflate.py. */
#if defined(__x86_64__)
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold(uint32_t crc, const unsigned char *p, size_t len /* >= 64, a multiple of 16 */)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4);
    const __m128i k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124);
    const __m128i poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x1 = _mm_loadu_si128((const __m128i *)(p + 0)), x2 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i *)(p + 32)), x4 = _mm_loadu_si128((const __m128i *)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    p += 64;
    len -= 64;
    while (len >= 64) {
        __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_clmulepi64_si128(x1, k1k2, 0x11);
        x2 = _mm_clmulepi64_si128(x2, k1k2, 0x11);
        x3 = _mm_clmulepi64_si128(x3, k1k2, 0x11);
        x4 = _mm_clmulepi64_si128(x4, k1k2, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, a1), _mm_loadu_si128((const __m128i *)(p + 0)));
        x2 = _mm_xor_si128(_mm_xor_si128(x2, a2), _mm_loadu_si128((const __m128i *)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(x3, a3), _mm_loadu_si128((const __m128i *)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(x4, a4), _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64;
        len -= 64;
    }
    /* four lanes into one */
    __m128i a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x2);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x3);
    a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
    x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), x4);
    while (len >= 16) {
        a = _mm_clmulepi64_si128(x1, k3k4, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k3k4, 0x11), a), _mm_loadu_si128((const __m128i *)p));
        p += 16;
        len -= 16;
    }
    /* 128 -> 64 bits */
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i x = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), x);
    /* 64 -> 32 bits */
    x = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, mask32);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(x1, k5, 0x00), x);
    /* Barrett reduction */
    x = _mm_and_si128(x1, mask32);
    x = _mm_clmulepi64_si128(x, poly, 0x10);
    x = _mm_and_si128(x, mask32);
    x = _mm_clmulepi64_si128(x, poly, 0x00);
    x1 = _mm_xor_si128(x1, x);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

uint32_t kssd_crc32(uint32_t crc, const unsigned char *p, size_t len)
{
#if defined(__x86_64__)
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (have && len >= 64) {
        const size_t body = len & ~(size_t)15;
        crc = ~crc32_fold(~crc, p, body);
        p += body;
        len -= body;
    }
#endif
    while (len) { /* (zlib takes its lengths as 32-bit numbers) */
        const size_t n = len > (1u << 30) ? (1u << 30) : len;
        crc = (uint32_t)crc32(crc, p, (uInt)n);
        p += n;
        len -= n;
    }
    return crc;
}

/* ---- one file's state: gzip members > deflate blocks > symbols -------------------------------------------------------------------
 * A state machine rather than nested loops, so that two files can be decoded in step (kssd_gunzip_mem2): z_advance() runs a file
 * up to the next Huffman block's symbols (member headers, block headers, stored blocks, code lengths -- none of it hot), the symbol
 * loops take it to the block's end, z_block_end() closes the block and, behind a member's last, checks its CRC-32 and length. */
enum { ZS_MEMBER, ZS_BLOCK, ZS_SYMBOLS, ZS_DONE, ZS_FAILED };
typedef struct {
    const unsigned char *src; /* the file's bytes */
    size_t src_len, at;       /* ... and where its next member starts */
    int members;
    bitr b;
    outbuf o;
    inflate_tabs *t;
    const uint32_t *lit, *lit_one, *dst; /* ZS_SYMBOLS: the block's tables */
    size_t member_start;
    int last, state, rc; /* last: the block is the member's last; rc: the KSSD_HOST_ERR_* of a ZS_FAILED file */
} zstate;

static void z_fail(zstate *z, int rc)
{
    z->state = ZS_FAILED;
    z->rc = rc;
}

static __thread inflate_tabs *t_tabs[2];
static pthread_key_t t_tabs_key; /* (the tables go back when the thread ends) */
static pthread_once_t t_tabs_once = PTHREAD_ONCE_INIT;
static void t_tabs_free(void *unused)
{
    (void)unused;
    for (int i = 0; i < 2; i++) { free(t_tabs[i]); t_tabs[i] = NULL; }
}
static void t_tabs_key_make(void) { pthread_key_create(&t_tabs_key, t_tabs_free); }
static int z_init(zstate *z, int slot, const unsigned char *in, size_t in_len, unsigned char *out, size_t cap)
{
    memset(z, 0, sizeof *z);
    z->src = in;
    z->src_len = in_len;
    z->o.out = out;
    z->o.cap = cap;
    z->state = ZS_MEMBER;
    z->rc = KSSD_HOST_OK;
    z->t = t_tabs[slot]; /* (the thread's own, kept from file to file: 0.2 MB that malloc() would map and unmap every time) */
    if (!z->t) {
        pthread_once(&t_tabs_once, t_tabs_key_make);
        pthread_setspecific(t_tabs_key, (void *)1);
        z->t = t_tabs[slot] = (inflate_tabs *)malloc(sizeof *z->t);
        if (!z->t) { z_fail(z, KSSD_HOST_ERR_NOMEM); return -1; }
        z->t->fixed_ready = 0;
    }
    /* the last member's trailer states its length modulo 2^32: room for it up front (a single member below 4 GiB: exact) -- as far as
     * the file's own size makes it plausible (sequence text packs 3 - 5 : 1; a damaged trailer must not reserve gigabytes per thread;
     * text that packs better than 16 : 1 grows its buffer on the way) */
    if (in_len >= 18) {
        size_t isize = (size_t)in[in_len - 4] | ((size_t)in[in_len - 3] << 8) | ((size_t)in[in_len - 2] << 16) | ((size_t)in[in_len - 1] << 24);
        if (isize > in_len * 16 + ((size_t)1 << 20)) isize = in_len * 16 + ((size_t)1 << 20);
        if (out_room(&z->o, isize + FAST_OUT_MARGIN + 64)) { z_fail(z, KSSD_HOST_ERR_NOMEM); return -1; }
    }
    return 0;
}

/* behind a member's last block: whole unread bytes go back to the input, the bits of the last byte are dropped; CRC-32, ISIZE */
static void z_member_end(zstate *z)
{
    bitr *b = &z->b;
    b->bb >>= b->bc & 7;
    b->bc -= b->bc & 7;
    b->in -= b->bc >> 3;
    b->bb = 0;
    b->bc = 0;
    if ((size_t)(b->in_end - b->in) < 8) { z_fail(z, KSSD_HOST_ERR_IO); return; }
    const uint32_t want_crc = (uint32_t)b->in[0] | ((uint32_t)b->in[1] << 8) | ((uint32_t)b->in[2] << 16) | ((uint32_t)b->in[3] << 24);
    const uint32_t want_len = (uint32_t)b->in[4] | ((uint32_t)b->in[5] << 8) | ((uint32_t)b->in[6] << 16) | ((uint32_t)b->in[7] << 24);
    const size_t start = z->member_start;
    if ((uint32_t)(z->o.len - start) != want_len || kssd_crc32(0, z->o.out + start, z->o.len - start) != want_crc) { z_fail(z, KSSD_HOST_ERR_IO); return; }
    z->at = (size_t)(b->in - z->src) + 8;
    z->members++;
    z->state = ZS_MEMBER;
}

static void z_block_end(zstate *z)
{
    if (z->last) z_member_end(z);
    else z->state = ZS_BLOCK;
}

/* up to the symbols of the next Huffman block (ZS_SYMBOLS), the end of the file (ZS_DONE) or a failure */
static void z_advance(zstate *z)
{
    for (;;) {
        if (z->state == ZS_MEMBER) {
            const unsigned char *in = z->src;
            const size_t in_len = z->src_len;
            size_t at = z->at;
            while (z->members && at < in_len && in[at] == 0) at++; /* (zcat accepts zero bytes behind the last member -- tape blocks -- and so does this) */
            if (at >= in_len) {
                if (!z->members) z_fail(z, KSSD_HOST_ERR_IO);
                else z->state = ZS_DONE;
                return;
            }
            if (in_len - at < 18 || in[at] != 0x1f || in[at + 1] != 0x8b || in[at + 2] != 8 || (in[at + 3] & 0xE0)) { z_fail(z, KSSD_HOST_ERR_IO); return; }
            const int flg = in[at + 3];
            size_t p = at + 10;
            if (flg & 4) { /* FEXTRA */
                if (p + 2 > in_len) { z_fail(z, KSSD_HOST_ERR_IO); return; }
                p += 2 + ((size_t)in[p] | ((size_t)in[p + 1] << 8));
            }
            for (int k = 0; k < 2; k++) /* FNAME, FCOMMENT: zero-terminated */
                if (flg & (k ? 16 : 8)) {
                    while (p < in_len && in[p]) p++;
                    p++;
                }
            if (flg & 2) p += 2; /* FHCRC */
            if (p + 8 > in_len) { z_fail(z, KSSD_HOST_ERR_IO); return; }
            z->b.in = in + p;
            z->b.in_end = in + in_len;
            z->b.bb = 0;
            z->b.bc = 0;
            z->member_start = z->o.len;
            z->state = ZS_BLOCK;
        } else if (z->state == ZS_BLOCK) {
            bitr *b = &z->b;
            outbuf *o = &z->o;
            uint32_t last, type;
            if (take_bits(b, 1, &last) || take_bits(b, 2, &type)) { z_fail(z, KSSD_HOST_ERR_IO); return; }
            z->last = (int)last;
            if (type == 0) { /* stored: back to a byte boundary, LEN / NLEN, the bytes */
                b->bb >>= b->bc & 7;
                b->bc -= b->bc & 7;
                /* the fast loop's refill leaves bits of *in above the count (harmless while `in` stands still: the next refill ORs
                 * the same bits again) -- the copy below moves `in`, so nothing above the count may stay */
                b->bb = b->bc ? b->bb & ((1ull << b->bc) - 1) : 0;
                uint32_t len, nlen;
                if (take_bits(b, 16, &len) || take_bits(b, 16, &nlen) || (len ^ nlen) != 0xFFFFu) { z_fail(z, KSSD_HOST_ERR_IO); return; }
                if (out_room(o, len)) { z_fail(z, KSSD_HOST_ERR_NOMEM); return; }
                uint32_t got = 0;
                while (got < len && b->bc >= 8) { /* (whole bytes the bit buffer holds already) */
                    o->out[o->len++] = (unsigned char)b->bb;
                    b->bb >>= 8;
                    b->bc -= 8;
                    got++;
                }
                if ((size_t)(b->in_end - b->in) < len - got) { z_fail(z, KSSD_HOST_ERR_IO); return; }
                memcpy(o->out + o->len, b->in, len - got);
                b->in += len - got;
                o->len += len - got;
                z_block_end(z);
                if (z->state == ZS_FAILED) return;
            } else if (type == 1) {
                fixed_tables(z->t);
                z->lit = z->t->fixed_lit;
                z->lit_one = z->t->fixed_lit_one;
                z->dst = z->t->fixed_dst;
                z->state = ZS_SYMBOLS;
                return;
            } else if (type == 2) {
                if (read_dynamic(b, z->t)) { z_fail(z, KSSD_HOST_ERR_IO); return; }
                z->lit = z->t->lit;
                z->lit_one = z->t->lit_one;
                z->dst = z->t->dst;
                z->state = ZS_SYMBOLS;
                return;
            } else {
                z_fail(z, KSSD_HOST_ERR_IO);
                return;
            }
        } else {
            return;
        }
    }
}

/* a file to its end, on its own */
static void z_run(zstate *z)
{
    for (;;) {
        z_advance(z);
        if (z->state != ZS_SYMBOLS) return;
        const int r = inflate_symbols(&z->b, &z->o, z->lit, z->lit_one, z->dst, z->member_start);
        if (r) { z_fail(z, r == -2 ? KSSD_HOST_ERR_NOMEM : KSSD_HOST_ERR_IO); return; }
        z_block_end(z);
    }
}

static int z_finish(zstate *z, unsigned char **out, size_t *cap, size_t *len)
{
    *out = z->o.out; /* (the caller's buffer may have moved even when the stream turns out corrupt) */
    *cap = z->o.cap;
    *len = z->state == ZS_DONE ? z->o.len : 0;
    return z->state == ZS_DONE ? KSSD_HOST_OK : z->rc != KSSD_HOST_OK ? z->rc : KSSD_HOST_ERR_IO;
}

/* ---- gzip members ---------------------------------------------------------------------------------------------------------------- */
int kssd_gunzip_mem(const unsigned char *in, size_t in_len, unsigned char **out, size_t *cap, size_t *len)
{
    if (!in || !out || !cap || !len) return KSSD_HOST_ERR_PARAM;
    zstate z;
    if (z_init(&z, 0, in, in_len, *out, *cap) == 0) z_run(&z);
    return z_finish(&z, out, cap, len);
}

/* two files at once, their Huffman blocks decoded in step by one thread (inflate_symbols2); whatever one file has beyond the other
 * -- and whatever is no symbol loop -- runs on its own.  rc[f]: what kssd_gunzip_mem would have returned for file f. */
void kssd_gunzip_mem2(const unsigned char *const in[2], const size_t in_len[2], unsigned char **out[2], size_t *cap[2], size_t *len[2], int rc[2])
{
    zstate z[2];
    for (int f = 0; f < 2; f++) {
        if (!in[f] || !out[f] || !cap[f] || !len[f]) { /* (nothing is decoded when a file's arguments are missing) */
            rc[0] = rc[1] = KSSD_HOST_ERR_PARAM;
            return;
        }
    }
    for (int f = 0; f < 2; f++)
        if (z_init(&z[f], f, in[f], in_len[f], *out[f], *cap[f]) == 0) z_advance(&z[f]);
    while (z[0].state == ZS_SYMBOLS && z[1].state == ZS_SYMBOLS) {
        int r[2];
        inflate_symbols2(&z[0].b, &z[0].o, z[0].lit, z[0].dst, z[0].member_start, &r[0], &z[1].b, &z[1].o, z[1].lit, z[1].dst, z[1].member_start, &r[1]);
        for (int f = 0; f < 2; f++) {
            if (r[f] == 2) continue;                          /* (only the other one stopped) */
            if (r[f] == 1)                                    /* out of its margins: to the block's end through the careful loop */
                r[f] = inflate_symbols(&z[f].b, &z[f].o, z[f].lit, z[f].lit_one, z[f].dst, z[f].member_start);
            if (r[f]) { z_fail(&z[f], r[f] == -2 ? KSSD_HOST_ERR_NOMEM : KSSD_HOST_ERR_IO); continue; }
            z_block_end(&z[f]);
            z_advance(&z[f]);
        }
    }
    for (int f = 0; f < 2; f++) {
        if (z[f].state == ZS_SYMBOLS || z[f].state == ZS_BLOCK || z[f].state == ZS_MEMBER) z_run(&z[f]);
        rc[f] = z_finish(&z[f], out[f], cap[f], len[f]);
    }
}
