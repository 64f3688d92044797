/*
 * disco_oracle.h — CPU restatement of DISCO's BuildGraph hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under disco_amd/ (the product) may include, link, load or
 * execute this.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, and
 * only as the checker.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement against
 *   (1) the reference's only in-tree golden vector, src/BuildGraph/bench_test_0_parGraph.txt, and
 *   (2) outputs of the real reference `buildG` (oracle/_ref/buildG_ref, built from
 *       /root/reference by oracle/Makefile) on seeded inputs, committed under tests/golden/.
 *
 * The inexact-overlap entry point at the end (oracle_build_graph_inexact, max_subs > 0) is an EXTENSION with no counterpart in the
 * reference: for that mode parity with the reference is UNPINNED by construction (there is no reference output to pin it to); what
 * pins it is max_subs = 0 being the pinned restatement and tests/test_oracle_inexact.py (a brute-force statement of the rule).
 *
 * Every function cites the reference file:line it follows (BG/ = src/BuildGraph/src/).
 * Read IDs here are 0-based ranks among good reads in file order (reference readNumber - 1).
 */
#ifndef DISCO_ORACLE_H_
#define DISCO_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_contained_row {
    uint64_t contained;  /* read2: the contained / duplicate read                              */
    uint64_t super;      /* read1: the containing read                                         */
    uint32_t orient;     /* file orientation, BG/OverlapGraph.cpp:428-434                       */
    uint32_t len2;       /* length of the contained read                                       */
    uint32_t len1;       /* length of the containing read                                      */
    uint32_t start;      /* len1 - overlapLen  (column 9 of the row)                           */
    uint32_t j;          /* k-mer position in read1 at which the hit was found                 */
    uint32_t type;       /* hash-hit type 0..3, BG/HashTable.cpp:535-566                        */
} oracle_contained_row;

typedef struct oracle_edge {
    uint64_t src;        /* smaller read id (column 1)                                         */
    uint64_t dst;        /* larger read id  (column 2)                                         */
    uint32_t orient;     /* 0..3 as seen from src, BG/Edge.h:30-34                              */
    uint32_t offset;     /* overlap offset in src (start1)                                     */
    uint32_t len_src;
    uint32_t len_dst;
} oracle_edge;

typedef struct oracle_counters {
    uint64_t n_reads;
    uint64_t probes;           /* Q = sum over reads of (L - k)                                 */
    uint64_t kmer_hits;        /* H = (probe, record) pairs with exact k-mer match, self excluded;
                                  only filled when ORACLE_COUNT_HITS is set                      */
    uint64_t n_contained;      /* C                                                             */
    uint64_t e_pre;            /* undirected overlaps entering the pre-reduction graph          */
    uint64_t e_out;            /* undirected edges after transitive reduction                   */
    uint64_t cap_bind_sites;   /* (read, j) sites where MAX_EDGE_PER_KMER cut off a valid hit   */
    uint64_t asymmetric_pairs; /* directed finds whose twin was not found from the other read   */
} oracle_counters;

typedef struct oracle_result {
    oracle_contained_row *contained;  /* discovery order: super ascending, then j, bucket order */
    oracle_edge *edges;               /* sorted by (src, dst, orient, offset), unique           */
    oracle_counters c;
    uint32_t *edge_subs;              /* per edge: substitutions of its overlap (all 0 unless the extension below is on) */
} oracle_result;

enum { ORACLE_COUNT_HITS = 1 };

/* ASCII -> base codes A0 C1 G2 T3 (BG/HashTable.h:16-24); returns -1 on a non-ACGT byte. */
int oracle_encode(const char *ascii, size_t n, uint8_t *codes);

/* a-2  Dataset::testRead  (BG/Dataset.cpp:403-452) on an upper-cased read. 1 = keep. */
int oracle_test_read(const char *read, size_t len);

/* a-1  Dataset::readDataset record splitting (BG/Dataset.cpp:255-294) for one FASTA/FASTQ buffer.
 * Calls cb(user, seq, len) once per record, in file order, with the raw (not upper-cased) sequence.
 * Returns the number of records, or -1 for an unknown format. */
long oracle_parse_records(const char *buf, size_t n,
                          void (*cb)(void *user, const char *seq, size_t len), void *user);

/* The path a-5 .. a-15 in bulk form:
 * index (BG/HashTable.cpp:46-114,423-514) -> markContainedReads (BG/OverlapGraph.cpp:333-505,
 * literal sequential order) -> insertAllEdgesOfRead for every non-contained read
 * (BG/OverlapGraph.cpp:631-678, no explored-skip) -> union of both sides' finds ->
 * markTransitiveEdges / removeTransitiveEdges per node (BG/OverlapGraph.cpp:687-761).
 *   codes : concatenated base codes of all good reads; read i = codes[off[i] .. off[i+1])
 *   min_overlap : MinOverlap4BuildGraph; k = min_overlap - 1 (BG/HashTable.cpp:50)
 * Returns 0, or -1 on bad arguments.  Free with oracle_free_result. */
int oracle_build_graph(const uint8_t *codes, const uint64_t *off, uint64_t n_reads,
                       uint32_t min_overlap, uint32_t flags, oracle_result *out);

/* EXTENSION, not a restatement: the reference compares exactly and writes 0 into the substitutions column
 * (BG/OverlapGraph.cpp:815-816).  With max_subs > 0 the two compares of checkOverlapForContainedRead / checkOverlap accept up to
 * max_subs differing bases outside the seed k-mer (the seed itself is still an exact hash-table hit); everything else follows
 * the reference, except that containment is taken in its order-free form (a read is contained iff any read contains it within
 * the threshold — with substitutions containment is not transitive, so the reference's skip of already-contained containers,
 * :395, would make the result depend on the processing order).  Pairs whose seed k-mer carries a substitution on one side are
 * found from the other side only; insertEdge's twin (:614-626) completes them, as it does for cap-bound pairs.
 * max_subs = 0 is oracle_build_graph.  Returns -2 if an emitted edge's overlap has more than max_subs substitutions. */
int oracle_build_graph_inexact(const uint8_t *codes, const uint64_t *off, uint64_t n_reads,
                               uint32_t min_overlap, uint32_t flags, uint32_t max_subs, oracle_result *out);

void oracle_free_result(oracle_result *r);

#ifdef __cplusplus
}
#endif
#endif /* DISCO_ORACLE_H_ */
