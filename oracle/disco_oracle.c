/*
 * disco_oracle.c — CPU restatement of DISCO's BuildGraph hot path (see disco_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP path, never the thing measured or shipped.
 * Parity status: PINNED against the reference golden vector and against the real reference binary
 * (tests/test_oracle_golden.py, tests/golden/).
 *
 * Plain C99, one base per byte, no bit tricks: deliberately a different style from the HIP kernels
 * so that the two implementations do not share bugs.  BG/ = /root/reference/src/BuildGraph/src/.
 */
#include "disco_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAX_EDGE_PER_KMER 4 /* BG/Common.h:62 */
#define NONE UINT64_MAX

/* ------------------------------------------------------------------------------------------------
 * small helpers
 * ---------------------------------------------------------------------------------------------- */
typedef struct vec64 {
    uint64_t *p;
    size_t n, cap;
} vec64;

static void vec64_push(vec64 *v, uint64_t x)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? v->cap * 2 : 16;
        v->p = (uint64_t *)realloc(v->p, v->cap * sizeof(uint64_t));
    }
    v->p[v->n++] = x;
}

int oracle_encode(const char *ascii, size_t n, uint8_t *codes)
{
    for (size_t i = 0; i < n; i++) {
        switch (ascii[i]) {
        case 'A': codes[i] = 0; break;
        case 'C': codes[i] = 1; break;
        case 'G': codes[i] = 2; break;
        case 'T': codes[i] = 3; break;
        default: return -1; /* BG/HashTable.cpp:474-475 throws */
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a-2  Dataset::testRead  — BG/Dataset.cpp:403-452
 * ---------------------------------------------------------------------------------------------- */
static const char *const FILTER_STRINGS[] = { /* BG/Dataset.cpp:48-85 (duplicates kept as listed) */
    "ACACACACACACACACACACACACACACA", "AGAGAGAGAGAGAGAGAGAGAGAGAGAGA", "ATATATATATATATATATATATATATATA",
    "CGCGCGCGCGCGCGCGCGCGCGCGCGCGC", "CTCTCTCTCTCTCTCTCTCTCTCTCTCTC", "AAGAAGAAGAAGAAGAAGAAGAAGAAGAA",
    "ATAATAATAATAATAATAATAATAATAAT", "TAATAATAATAATAATAATAATAATAATA", "AACAACAACAACAACAACAACAACAACAA",
    "ACAACAACAACAACAACAACAACAACAAC", "CAACAACAACAACAACAACAACAACAACA", "AAGAAGAAGAAGAAGAAGAAGAAGAAGAA",
    "AGAAGAAGAAGAAGAAGAAGAAGAAGAAG", "GAAGAAGAAGAAGAAGAAGAAGAAGAAGA", "TTCTTCTTCTTCTTCTTCTTCTTCTTCTT",
    "AAATAAATAAATAAATAAATAAATAAATA", "TAAATAAATAAATAAATAAATAAATAAAT", "ATAAATAAATAAATAAATAAATAAATAAA",
    "AATAAATAAATAAATAAATAAATAAATAA", "AATTAATTAATTAATTAATTAATTAATTA", "ATTAATTAATTAATTAATTAATTAATTAA",
    "TTAATTAATTAATTAATTAATTAATTAAT", "TAATTAATTAATTAATTAATTAATTAATT", "AAAGAAAGAAAGAAAGAAAGAAAGAAAGA",
    "AAAGAAAGAAAGAAAGAAAGAAAGAAAGA", "AGAAAGAAAGAAAGAAAGAAAGAAAGAAA", "GAAAGAAAGAAAGAAAGAAAGAAAGAAAG",
    "TACATACATACATACATACATACATACAT", "ACATACATACATACATACATACATACATA", "CATACATACATACATACATACATACATAC",
    "ATACATACATACATACATACATACATACA", "GTTTGTTTGTTTGTTTGTTTGTTTGTTTG", "TGTTTGTTTGTTTGTTTGTTTGTTTGTTT",
    "TTTGTTTGTTTGTTTGTTTGTTTGTTTGT", "AGGGAGGGAGGGAGGGAGGGAGGGAGGGA", "GAGGGAGGGAGGGAGGGAGGGAGGGAGGG",
    "GGAGGGAGGGAGGGAGGGAGGGAGGGAGG", "GGGAGGGAGGGAGGGAGGGAGGGAGGGAG"};
static const char *const MER_CHECK[] = { /* BG/Dataset.cpp:87 */
    "AC", "AG", "AT", "CG", "CT", "GT", "AAT", "ATA", "TAA", "AAC", "ACA", "CAA", "AAG", "AGA", "GAA", "GGGGCC"};

/* countSubstring — BG/Common.h:173-183 : non-overlapping occurrences, scanning left to right */
static size_t count_substring(const char *s, size_t n, const char *sub, size_t m)
{
    size_t count = 0, pos = 0;
    if (m == 0 || n < m) return 0;
    while (pos + m <= n) {
        if (memcmp(s + pos, sub, m) == 0) {
            count++;
            pos += m;
        } else
            pos++;
    }
    return count;
}

int oracle_test_read(const char *read, size_t len)
{
    size_t cnt[4] = {0, 0, 0, 0};
    if (len < 30) return 0; /* MIN_READ_SIZE, BG/Dataset.h:15, BG/Dataset.cpp:407 */
    for (size_t i = 0; i < len; i++) { /* :409-414 */
        char c = read[i];
        if (c != 'A' && c != 'C' && c != 'G' && c != 'T') return 0;
        cnt[(c >> 1) & 3]++;
    }
    size_t threshold = (size_t)((double)len * .7); /* :415 */
    if (cnt[0] >= threshold || cnt[1] >= threshold || cnt[2] >= threshold || cnt[3] >= threshold) return 0;
    for (size_t i = 0; i < sizeof FILTER_STRINGS / sizeof *FILTER_STRINGS; i++) { /* :420-429 */
        size_t fl = strlen(FILTER_STRINGS[i]);
        if (len < fl) return 0;
        if (memcmp(FILTER_STRINGS[i], read, fl) == 0) return 0;
        if (memcmp(FILTER_STRINGS[i], read + len - fl, fl) == 0) return 0;
    }
    threshold = (size_t)((double)len * .5); /* :431 */
    for (size_t i = 0; i < sizeof MER_CHECK / sizeof *MER_CHECK; i++) { /* :432-438 */
        size_t m = strlen(MER_CHECK[i]);
        size_t rep = count_substring(read, len, MER_CHECK[i], m) * m;
        if (rep >= threshold) return 0;
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------------
 * a-1  record splitting of Dataset::readDataset — BG/Dataset.cpp:255-294
 *   first byte of the file decides FASTA ('>') or FASTQ ('@'); FASTA record = header line, then
 *   everything up to the next '>' with '\n' removed ('\r' is kept); FASTQ record = 4 lines.
 * ---------------------------------------------------------------------------------------------- */
long oracle_parse_records(const char *buf, size_t n, void (*cb)(void *, const char *, size_t), void *user)
{
    long nrec = 0;
    size_t p = 0;
    int fasta;
    char *tmp;
    if (n == 0) return 0;
    if (buf[0] == '>') fasta = 1;
    else if (buf[0] == '@') fasta = 0;
    else return -1;
    tmp = (char *)malloc(n + 1);
    while (p < n) {
        /* getline(myFile,text): header line (may be empty -> the reference would still proceed) */
        size_t e = p;
        while (e < n && buf[e] != '\n') e++;
        p = (e < n) ? e + 1 : n; /* past the header line */
        if (fasta) {
            /* getline(myFile,text,'>') : up to and excluding the next '>' (consumed) */
            size_t q = p, m = 0;
            while (q < n && buf[q] != '>') {
                if (buf[q] != '\n') tmp[m++] = buf[q];
                q++;
            }
            cb(user, tmp, m);
            nrec++;
            p = (q < n) ? q + 1 : n;
            /* an empty remainder after the last '>' would make getline fail -> loop ends */
            if (p >= n) break;
        } else {
            size_t line_s[3], line_e[3];
            for (int l = 0; l < 3; l++) {
                line_s[l] = p;
                e = p;
                while (e < n && buf[e] != '\n') e++;
                line_e[l] = e;
                p = (e < n) ? e + 1 : n;
            }
            cb(user, buf + line_s[0], line_e[0] - line_s[0]);
            nrec++;
        }
    }
    free(tmp);
    return nrec;
}

/* ------------------------------------------------------------------------------------------------
 * index: a-5 getHashIndex, a-6/a-7 two-pass CSR table — BG/HashTable.cpp:46-114,383-416,423-514
 * The reference's hash value never reaches the output; only the in-bucket order
 * (ascending read id, prefix record before suffix record) does, and that is preserved here by
 * filling in read order.
 * ---------------------------------------------------------------------------------------------- */
typedef struct oracle_index {
    const uint8_t *codes;
    const uint64_t *off;
    uint64_t n;
    uint32_t k;
    uint64_t nbuckets; /* power of two */
    uint64_t *start;   /* nbuckets + 1 */
    uint64_t *rec;     /* 2n records: id << 1 | isSuffix */
    uint64_t *super;   /* superReadID per read, NONE = not contained (BG/Read.h:24) */
    uint32_t max_subs; /* 0 = the reference (exact compares); > 0 = the inexact-overlap EXTENSION, see disco_oracle.h */
} oracle_index;

static uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* getHashIndex — BG/HashTable.cpp:383-391 : min(hash(kmer), hash(revcomp(kmer))) */
static uint64_t kmer_bucket(const oracle_index *ix, const uint8_t *s)
{
    uint64_t hf = 1469598103934665603ull, hr = 1469598103934665603ull;
    for (uint32_t i = 0; i < ix->k; i++) {
        hf = (hf ^ s[i]) * 1099511628211ull;
        hr = (hr ^ (uint64_t)(3 - s[ix->k - 1 - i])) * 1099511628211ull;
    }
    hf = mix64(hf);
    hr = mix64(hr);
    return (hf <= hr ? hf : hr) & (ix->nbuckets - 1);
}

static uint32_t read_len(const oracle_index *ix, uint64_t id) { return (uint32_t)(ix->off[id + 1] - ix->off[id]); }
static const uint8_t *read_ptr(const oracle_index *ix, uint64_t id) { return ix->codes + ix->off[id]; }

static void index_build(oracle_index *ix)
{
    uint64_t nb = 1024;
    while (nb < 4 * ix->n) nb <<= 1;
    ix->nbuckets = nb;
    ix->start = (uint64_t *)calloc(nb + 1, sizeof(uint64_t));
    ix->rec = (uint64_t *)malloc((2 * ix->n + 1) * sizeof(uint64_t));
    /* populateReadLengths — BG/HashTable.cpp:77-91,341-356 */
    for (uint64_t i = 0; i < ix->n; i++) {
        const uint8_t *r = read_ptr(ix, i);
        uint32_t L = read_len(ix, i);
        ix->start[kmer_bucket(ix, r) + 1]++;
        ix->start[kmer_bucket(ix, r + L - ix->k) + 1]++;
    }
    /* exclusive prefix sum — BG/HashTable.cpp:58-67 */
    for (uint64_t b = 0; b < nb; b++) ix->start[b + 1] += ix->start[b];
    /* populateReadData — BG/HashTable.cpp:97-114,451-454,486-489 : prefix record, then suffix record */
    uint64_t *cursor = (uint64_t *)calloc(nb, sizeof(uint64_t));
    for (uint64_t i = 0; i < ix->n; i++) {
        const uint8_t *r = read_ptr(ix, i);
        uint32_t L = read_len(ix, i);
        uint64_t b = kmer_bucket(ix, r);
        ix->rec[ix->start[b] + cursor[b]++] = i << 1;
        b = kmer_bucket(ix, r + L - ix->k);
        ix->rec[ix->start[b] + cursor[b]++] = (i << 1) | 1;
    }
    free(cursor);
}

static int eq_fwd(const uint8_t *a, const uint8_t *b, uint32_t n) { return memcmp(a, b, n) == 0; }
/* a[0..n) == revcomp(b[0..n)) */
static int eq_rev(const uint8_t *a, const uint8_t *b, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++)
        if (a[i] != 3 - b[n - 1 - i]) return 0;
    return 1;
}

/* the compare of checkOverlap* : memcmp in the reference; with the extension, at most max_subs differing bases */
static uint32_t mismatches(const uint8_t *a, const uint8_t *b, uint32_t n)
{
    uint32_t c = 0;
    for (uint32_t i = 0; i < n; i++) c += a[i] != b[i];
    return c;
}
static int same(const oracle_index *ix, const uint8_t *a, const uint8_t *b, uint32_t n)
{
    if (ix->max_subs == 0) return memcmp(a, b, n) == 0;
    return mismatches(a, b, n) <= ix->max_subs;
}

/* getListOfReads — BG/HashTable.cpp:521-571.  Appends readID | type << 62 in bucket order. */
static void list_of_reads(const oracle_index *ix, const uint8_t *q, vec64 *hits, int skip_contained)
{
    uint64_t b = kmer_bucket(ix, q);
    hits->n = 0;
    for (uint64_t p = ix->start[b]; p < ix->start[b + 1]; p++) {
        uint64_t id = ix->rec[p] >> 1;
        int is_suffix = (int)(ix->rec[p] & 1);
        const uint8_t *r = read_ptr(ix, id);
        uint32_t L = read_len(ix, id);
        if (skip_contained && ix->super[id] != NONE) continue; /* :533 */
        if (!is_suffix) {                                      /* :535-550 */
            if (eq_fwd(q, r, ix->k)) vec64_push(hits, id | (0ull << 62));
            else if (eq_rev(q, r, ix->k)) vec64_push(hits, id | (3ull << 62));
        } else { /* :551-566 */
            const uint8_t *s = r + L - ix->k;
            if (eq_fwd(q, s, ix->k)) vec64_push(hits, id | (1ull << 62));
            else if (eq_rev(q, s, ix->k)) vec64_push(hits, id | (2ull << 62));
        }
    }
}

/* string2 of checkOverlap* : read2 forward (orient 0,1) or reverse complement (orient 2,3) */
static void oriented(const oracle_index *ix, uint64_t id, uint32_t orient, uint8_t *buf)
{
    const uint8_t *r = read_ptr(ix, id);
    uint32_t L = read_len(ix, id);
    if (orient == 0 || orient == 1) memcpy(buf, r, L);
    else
        for (uint32_t i = 0; i < L; i++) buf[i] = (uint8_t)(3 - r[L - 1 - i]);
}

/* checkOverlapForContainedRead — BG/OverlapGraph.cpp:517-554 */
static int check_contained(const oracle_index *ix, const uint8_t *read1, uint32_t len1, uint64_t read2,
                           uint32_t orient, uint32_t start, uint8_t *buf)
{
    uint32_t k = ix->k, len2 = read_len(ix, read2);
    oriented(ix, read2, orient, buf);
    if (orient == 0 || orient == 2) {
        uint32_t rem1 = len1 - start - k, rem2 = len2 - k;
        if (rem1 >= rem2) return same(ix, read1 + start + k, buf + k, rem2);
    } else {
        uint32_t rem1 = start, rem2 = len2 - k;
        if (rem1 >= rem2) return same(ix, read1 + start - rem2, buf, rem2);
    }
    return 0;
}

/* checkOverlap — BG/OverlapGraph.cpp:567-595 */
static int check_overlap(const oracle_index *ix, const uint8_t *read1, uint32_t len1, uint64_t read2,
                         uint32_t orient, uint32_t start, uint8_t *buf)
{
    uint32_t k = ix->k, len2 = read_len(ix, read2);
    oriented(ix, read2, orient, buf);
    if (orient == 0 || orient == 2) {
        if (len1 - start - k >= len2 - k) return 0; /* :579 */
        return same(ix, read1 + start + k, buf + k, len1 - (start + k));
    } else {
        if (len2 - k < start) return 0; /* :591 */
        return same(ix, read1, buf + (len2 - k - start), start);
    }
}

/* orientation / overlap-length map shared by BG/OverlapGraph.cpp:428-434 and :660-666 */
static void map_type(uint32_t type, uint32_t len1, uint32_t k, uint32_t j, uint32_t *orient, uint32_t *ovl)
{
    switch (type) {
    case 0: *orient = 3; *ovl = len1 - j; break;
    case 1: *orient = 0; *ovl = k + j; break;
    case 2: *orient = 2; *ovl = len1 - j; break;
    default: *orient = 1; *ovl = k + j; break;
    }
}

/* twinEdgeOrientation — BG/OverlapGraph.cpp:770-784 */
static uint32_t twin_orient(uint32_t o) { return o == 0 ? 3 : (o == 3 ? 0 : o); }

/* ------------------------------------------------------------------------------------------------
 * a-9  markContainedReads — BG/OverlapGraph.cpp:333-505, literal sequential (-t 1) order
 * ---------------------------------------------------------------------------------------------- */
static void mark_contained(oracle_index *ix, uint32_t maxlen, oracle_result *out)
{
    vec64 hits = {0, 0, 0};
    uint8_t *buf = (uint8_t *)malloc(maxlen + 1);
    size_t cap = 0, n = 0;
    oracle_contained_row *rows = NULL;
    for (uint64_t i = 0; i < ix->n; i++) {          /* :391 */
        /* extension only: with substitutions containment is not transitive, so which reads are contained would depend on the
         * order in which containers are themselves marked; the extension defines the order-free form — a read is contained
         * iff ANY read contains it within the threshold, recorded with its first container in (read, j, bucket) order */
        if (ix->max_subs == 0 && ix->super[i] != NONE) continue; /* :395 */
        const uint8_t *read1 = read_ptr(ix, i);
        uint32_t len1 = read_len(ix, i);
        for (uint32_t j = 0; j < len1 - ix->k; j++) { /* :401 */
            list_of_reads(ix, read1 + j, &hits, ix->max_subs == 0);  /* :404 */
            for (size_t h = 0; h < hits.n; h++) {    /* :407 */
                uint64_t read2 = hits.p[h] & 0x3FFFFFFFFFFFFFFFull;
                uint32_t type = (uint32_t)(hits.p[h] >> 62);
                if (ix->super[read2] != NONE) continue; /* :417 */
                uint32_t len2 = read_len(ix, read2);
                if (i != read2 && check_contained(ix, read1, len1, read2, type, j, buf)) { /* :421 */
                    if (len1 > len2 || (len1 == len2 && i < read2)) {                       /* :424,:449 */
                        uint32_t orient, ovl;
                        map_type(type, len1, ix->k, j, &orient, &ovl);
                        if (ix->super[read2] == NONE) ix->super[read2] = i; /* :435-436 */
                        if (n == cap) {
                            cap = cap ? cap * 2 : 1024;
                            rows = (oracle_contained_row *)realloc(rows, cap * sizeof *rows);
                        }
                        rows[n].contained = read2;
                        rows[n].super = i;
                        rows[n].orient = orient;
                        rows[n].len2 = len2;
                        rows[n].len1 = len1;
                        rows[n].start = len1 - ovl;
                        rows[n].j = j;
                        rows[n].type = type;
                        n++;
                    }
                }
            }
        }
    }
    free(buf);
    free(hits.p);
    out->contained = rows;
    out->c.n_contained = n;
}

/* ------------------------------------------------------------------------------------------------
 * a-12  insertAllEdgesOfRead — BG/OverlapGraph.cpp:631-678, for one read, without the
 * explored-skip (:652-653): every non-contained read discovers its own edges.
 * A find is packed as  offset << 44 | dst << 2 | orient  so that sorting orders by offset first.
 * ---------------------------------------------------------------------------------------------- */
#define FIND_PACK(off, dst, o) (((uint64_t)(off) << 44) | ((uint64_t)(dst) << 2) | (uint64_t)(o))
#define FIND_OFF(f) ((uint32_t)((f) >> 44))
#define FIND_DST(f) (((f) >> 2) & ((1ull << 42) - 1))
#define FIND_ORI(f) ((uint32_t)((f)&3))

static void edges_of_read(const oracle_index *ix, uint64_t read1id, vec64 *finds, vec64 *hits, vec64 *inserted,
                          uint8_t *buf, uint64_t *cap_sites)
{
    const uint8_t *read1 = read_ptr(ix, read1id);
    uint32_t len1 = read_len(ix, read1id), k = ix->k;
    finds->n = 0;
    inserted->n = 0;
    for (uint32_t j = 1; j < len1 - k; j++) { /* :638 */
        list_of_reads(ix, read1 + j, hits, 1); /* :641 */
        int ctr = 0;
        size_t h = 0;
        for (; h < hits->n && ctr < MAX_EDGE_PER_KMER; h++) { /* :645 */
            uint64_t read2 = hits->p[h] & 0x3FFFFFFFFFFFFFFFull;
            uint32_t type = (uint32_t)(hits->p[h] >> 62);
            int seen = 0;
            for (size_t t = 0; t < inserted->n; t++)
                if (inserted->p[t] == read2) seen = 1; /* :656 */
            if (read1id != read2 && !seen && ix->super[read1id] == NONE && ix->super[read2] == NONE &&
                check_overlap(ix, read1, len1, read2, type, j, buf)) { /* :655-658 */
                uint32_t orient, ovl;
                map_type(type, len1, k, j, &orient, &ovl); /* :660-666 */
                vec64_push(finds, FIND_PACK(len1 - ovl, read2, orient)); /* :667 */
                vec64_push(inserted, read2);
                ctr++;
            }
        }
        /* cap-bind diagnostic: did the cap cut off a hit that would have been accepted? */
        for (; h < hits->n; h++) {
            uint64_t read2 = hits->p[h] & 0x3FFFFFFFFFFFFFFFull;
            uint32_t type = (uint32_t)(hits->p[h] >> 62);
            int seen = 0;
            for (size_t t = 0; t < inserted->n; t++)
                if (inserted->p[t] == read2) seen = 1;
            if (read1id != read2 && !seen && ix->super[read2] == NONE &&
                check_overlap(ix, read1, len1, read2, type, j, buf)) {
                (*cap_sites)++;
                break;
            }
        }
    }
}

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static int cmp_edge(const void *a, const void *b)
{
    const oracle_edge *x = (const oracle_edge *)a, *y = (const oracle_edge *)b;
    if (x->src != y->src) return x->src < y->src ? -1 : 1;
    if (x->dst != y->dst) return x->dst < y->dst ? -1 : 1;
    if (x->orient != y->orient) return x->orient < y->orient ? -1 : 1;
    if (x->offset != y->offset) return x->offset < y->offset ? -1 : 1;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * the whole path
 * ---------------------------------------------------------------------------------------------- */
/* extension: substitutions of the overlap an edge stands for, from the edge's geometry alone (BG/Edge.h:30-34 orientations,
 * offsets as insertEdge stores them, BG/OverlapGraph.cpp:614-626): orient 2,3 — string2 starts at `offset` of src;
 * orient 0,1 — string2 ends where the first len_src - offset bases of src end; string2 = dst (0,3) or its reverse complement */
static uint32_t edge_substitutions(const oracle_index *ix, uint64_t src, uint64_t dst, uint32_t orient, uint32_t offset, uint8_t *buf)
{
    uint32_t l1 = read_len(ix, src), l2 = read_len(ix, dst), ovl = l1 - offset;
    const uint8_t *a = read_ptr(ix, src);
    oriented(ix, dst, (orient == 0 || orient == 3) ? 0 : 2, buf);
    if (orient >= 2) return mismatches(a + offset, buf, ovl);
    return mismatches(a, buf + (l2 - ovl), ovl);
}

int oracle_build_graph(const uint8_t *codes, const uint64_t *off, uint64_t n_reads, uint32_t min_overlap,
                       uint32_t flags, oracle_result *out)
{
    return oracle_build_graph_inexact(codes, off, n_reads, min_overlap, flags, 0, out);
}

int oracle_build_graph_inexact(const uint8_t *codes, const uint64_t *off, uint64_t n_reads, uint32_t min_overlap,
                               uint32_t flags, uint32_t max_subs, oracle_result *out)
{
    oracle_index ix;
    uint32_t maxlen = 0;
    int geometry_errors = 0;
    if (!out || min_overlap < 2) return -1;
    memset(out, 0, sizeof *out);
    memset(&ix, 0, sizeof ix);
    ix.codes = codes;
    ix.off = off;
    ix.n = n_reads;
    ix.k = min_overlap - 1; /* hashStringLength, BG/HashTable.cpp:50 */
    ix.max_subs = max_subs;
    out->c.n_reads = n_reads;
    for (uint64_t i = 0; i < n_reads; i++) {
        uint64_t L = off[i + 1] - off[i];
        if (L <= min_overlap || L > 32767) return -1; /* BG/Dataset.cpp:305 ; 15-bit length field BG/HashTable.cpp:531 */
        if (L > maxlen) maxlen = (uint32_t)L;
        out->c.probes += L - ix.k;
    }
    ix.super = (uint64_t *)malloc((n_reads + 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < n_reads; i++) ix.super[i] = NONE;
    index_build(&ix);

    if (flags & ORACLE_COUNT_HITS) { /* H: exact k-mer matches over every probe, self excluded */
        vec64 hits = {0, 0, 0};
        for (uint64_t i = 0; i < n_reads; i++) {
            uint32_t L = read_len(&ix, i);
            for (uint32_t j = 0; j < L - ix.k; j++) {
                list_of_reads(&ix, read_ptr(&ix, i) + j, &hits, 0);
                for (size_t h = 0; h < hits.n; h++)
                    if ((hits.p[h] & 0x3FFFFFFFFFFFFFFFull) != i) out->c.kmer_hits++;
            }
        }
        free(hits.p);
    }

    mark_contained(&ix, maxlen, out);

    /* every non-contained read discovers its edges; F[v] = finds of v */
    vec64 *F = (vec64 *)calloc(n_reads + 1, sizeof(vec64));
    {
        vec64 finds = {0, 0, 0}, hits = {0, 0, 0}, inserted = {0, 0, 0};
        uint8_t *buf = (uint8_t *)malloc(maxlen + 1);
        for (uint64_t v = 0; v < n_reads; v++) {
            if (ix.super[v] != NONE) continue;
            edges_of_read(&ix, v, &finds, &hits, &inserted, buf, &out->c.cap_bind_sites);
            for (size_t t = 0; t < finds.n; t++) vec64_push(&F[v], finds.p[t]);
        }
        free(buf);
        free(finds.p);
        free(hits.p);
        free(inserted.p);
    }
    /* insertEdge adds the twin to read2's list (BG/OverlapGraph.cpp:614-626): adj = finds ∪ twins */
    vec64 *adj = (vec64 *)calloc(n_reads + 1, sizeof(vec64));
    for (uint64_t v = 0; v < n_reads; v++) {
        uint32_t lv = read_len(&ix, v);
        for (size_t t = 0; t < F[v].n; t++) {
            uint64_t f = F[v].p[t], w = FIND_DST(f);
            uint32_t lw = read_len(&ix, w);
            uint32_t off_rev = lw + FIND_OFF(f) - lv; /* :617 */
            uint64_t twin = FIND_PACK(off_rev, v, twin_orient(FIND_ORI(f)));
            int found = 0;
            for (size_t u = 0; u < F[w].n; u++)
                if (F[w].p[u] == twin) found = 1;
            if (!found) out->c.asymmetric_pairs++;
            vec64_push(&adj[v], f);
            vec64_push(&adj[w], twin);
        }
    }
    for (uint64_t v = 0; v < n_reads; v++) { /* sort by offset (:675-676) with a total tie-break, unique */
        if (!adj[v].n) continue;
        qsort(adj[v].p, adj[v].n, sizeof(uint64_t), cmp_u64);
        size_t m = 1;
        for (size_t t = 1; t < adj[v].n; t++)
            if (adj[v].p[t] != adj[v].p[m - 1]) adj[v].p[m++] = adj[v].p[t];
        adj[v].n = m;
        out->c.e_pre += m;
    }
    out->c.e_pre /= 2;

    /* a-14 markTransitiveEdges — BG/OverlapGraph.cpp:687-723 ; flag[v][t] = edge t of v marked from v */
    uint8_t **flag = (uint8_t **)calloc(n_reads + 1, sizeof(uint8_t *));
    for (uint64_t v = 0; v < n_reads; v++) {
        size_t d = adj[v].n;
        if (!d) continue;
        flag[v] = (uint8_t *)calloc(d, 1);
        uint8_t *elim = (uint8_t *)calloc(d, 1); /* ELIMINATED per list slot; slots of one dst share the state */
        for (size_t i = 0; i < d; i++) {          /* :693 */
            if (elim[i]) continue;                /* :696 INPLAY test */
            uint64_t u = FIND_DST(adj[v].p[i]);
            uint32_t type1 = FIND_ORI(adj[v].p[i]);
            for (size_t j = 0; j < adj[u].n; j++) { /* :698 */
                uint64_t w = FIND_DST(adj[u].p[j]);
                uint32_t type2 = FIND_ORI(adj[u].p[j]);
                int ok = ((type1 == 0 || type1 == 2) && (type2 == 0 || type2 == 1)) ||
                         ((type1 == 1 || type1 == 3) && (type2 == 2 || type2 == 3)); /* :705-708 */
                if (!ok) continue;
                for (size_t s = 0; s < d; s++) /* markedNodes is keyed by read id (:689-691,:701) */
                    if (FIND_DST(adj[v].p[s]) == w && !elim[s]) elim[s] = 1;
            }
        }
        memcpy(flag[v], elim, d); /* :713-720 (the twin flag is looked up from the other side below) */
        free(elim);
    }
    /* a-15 removeTransitiveEdges + a-16 canonical emission: an edge survives iff it is flagged from neither end */
    {
        size_t cap = 0, n = 0;
        oracle_edge *E = NULL;
        for (uint64_t v = 0; v < n_reads; v++) {
            uint32_t lv = read_len(&ix, v);
            for (size_t t = 0; t < adj[v].n; t++) {
                uint64_t f = adj[v].p[t], w = FIND_DST(f);
                if (!(v < w)) continue; /* BG/OverlapGraph.cpp:808 : column 1 is the smaller read number */
                uint32_t lw = read_len(&ix, w);
                uint64_t twin = FIND_PACK(lw + FIND_OFF(f) - lv, v, twin_orient(FIND_ORI(f)));
                int twin_flag = 0;
                for (size_t u = 0; u < adj[w].n; u++)
                    if (adj[w].p[u] == twin) twin_flag = flag[w][u];
                if (flag[v][t] || twin_flag) continue;
                if (n == cap) {
                    cap = cap ? cap * 2 : 1024;
                    E = (oracle_edge *)realloc(E, cap * sizeof *E);
                }
                E[n].src = v;
                E[n].dst = w;
                E[n].orient = FIND_ORI(f);
                E[n].offset = FIND_OFF(f);
                E[n].len_src = lv;
                E[n].len_dst = lw;
                n++;
            }
        }
        if (n) qsort(E, n, sizeof *E, cmp_edge);
        out->edges = E;
        out->c.e_out = n;
        out->edge_subs = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
        if (max_subs) {
            uint8_t *buf = (uint8_t *)malloc(maxlen + 1);
            for (size_t t = 0; t < n; t++) {
                out->edge_subs[t] = edge_substitutions(&ix, E[t].src, E[t].dst, E[t].orient, E[t].offset, buf);
                if (out->edge_subs[t] > max_subs) geometry_errors++; /* an edge is an accepted find or its twin */
            }
            free(buf);
        }
    }
    for (uint64_t v = 0; v < n_reads; v++) {
        free(F[v].p);
        free(adj[v].p);
        free(flag[v]);
    }
    free(F);
    free(adj);
    free(flag);
    free(ix.start);
    free(ix.rec);
    free(ix.super);
    return geometry_errors ? -2 : 0;
}

void oracle_free_result(oracle_result *r)
{
    if (!r) return;
    free(r->contained);
    free(r->edges);
    free(r->edge_subs);
    r->contained = NULL;
    r->edges = NULL;
    r->edge_subs = NULL;
}
