/*
 * disco_hip.h — C ABI of the MI355X-native BuildGraph hot path (libdisco_hip.so).
 *
 * This is the drop-in boundary INSIDE the buildG process.  The reference has no FFI for
 * this path; its nearest analogue is the HashTable / OverlapGraph surface that main.cpp drives
 * (/root/reference/src/BuildGraph/src/main.cpp:55-62):
 *
 *     Dataset  -> HashTable::insertDataset -> OverlapGraph ctor (markContainedReads, BFS edge discovery,
 *                                             markTransitiveEdges/removeTransitiveEdges, saveParGraphToFile)
 *
 * Each entry point below names the reference interface it replaces (BG/ = src/BuildGraph/src/).
 *
 * Conventions
 *   - every function returns 0 on success or a negative DISCO_E_* code; disco_last_error() gives the message
 *   - the caller owns host buffers, the library owns device buffers (except disco_adopt_reads)
 *   - one context per GPU; a context is not thread-safe; different contexts are independent
 *   - read ids are 0-based ranks of the good reads in file order (reference readNumber - 1, BG/Dataset.cpp:133-134)
 *   - reads are 2-bit packed, 32 bases per 64-bit word, MSB first, A0 C1 G2 T3 (BG/HashTable.cpp:456-477,
 *     BG/HashTable.h:16-24), one read per row of a fixed-stride [n][stride_words] array, unused bits zero
 *   - no C++ types, exceptions or torch types cross this boundary
 */
#ifndef DISCO_HIP_H_
#define DISCO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): disco_dist_info grew in the MIDDLE in round 5 (DISCO_X_COUNT 11 -> 14: every field behind bytes_sent[] moved) under the
 * number 1; the number now says so, and buildG / disco_amd/buildgraph.py refuse a library whose number is not the header's they were
 * written against. The bench / test-only entry points (synthetic reads, substitution errors, the two bandwidth probes) moved to
 * include/disco_hip_test.h — same library, not part of what a BuildGraph host binds. */
#define DISCO_ABI_VERSION 2

enum {
    DISCO_OK = 0,
    DISCO_E_ARG = -1,      /* bad argument                                               */
    DISCO_E_HIP = -2,      /* a HIP runtime call failed                                  */
    DISCO_E_NOMEM = -3,    /* device or host allocation failed                           */
    DISCO_E_STATE = -4,    /* phases called out of order                                 */
    DISCO_E_CAPACITY = -5, /* an internal buffer could not be grown to the needed size   */
    DISCO_E_UNSUPPORTED = -6
};

/* disco_params.flags */
#define DISCO_FLAG_TWO_PASS_VERIFY 1u /* read sets of mixed lengths (shortest < 0.9 x longest), single-GPU passes: verify the
                                         containment-type candidates first, fix the contained flags, then fetch rows only for the
                                         overlap-type candidates of non-contained reads — about half the row fetches where most
                                         reads are contained (metagenomes). Results are unchanged; disco_counters.kmer_hits and
                                         .raw_hits then count the compared candidates only (the reference has no such counters).
                                         Round 4: since the single pass sends one containment key per alignment instead of two it is
                                         the faster form on the shapes measured (50 M reads of 100-250 bases: 31 against 39 ms of
                                         verify); the flag stays for read sets it still helps and is ignored on a table with two
                                         classes of rows (disco_long_rows) */

typedef struct disco_ctx disco_ctx;

typedef struct disco_params {
    uint32_t min_overlap;        /* MinOverlap4BuildGraph (disco.cfg:9); k = min_overlap - 1 (BG/HashTable.cpp:50) */
    uint32_t max_edges_per_kmer; /* MAX_EDGE_PER_KMER (BG/Common.h:62); 0 -> 4                                       */
    uint32_t flags;              /* DISCO_FLAG_*                                                                     */
    uint32_t max_substitutions;  /* 0 = the reference: exact overlaps. > 0 = EXTENSION (the reference always writes
                                    0 into the substitutions column, BG/OverlapGraph.cpp:815-816): a candidate seeded by an exact
                                    end-k-mer hit is an overlap / a containment if its aligned region differs in at most this many
                                    bases; pairs found from one side only are completed by the twin pass (order-dependent regime) */
} disco_params;


/* one row of <prefix>_<t>_containedReads.txt (BG/OverlapGraph.cpp:438-447) in read ids */
typedef struct disco_contained_row {
    uint64_t contained; /* read2                                        */
    uint64_t super;     /* read1, the containing read                   */
    uint32_t orient;    /* BG/OverlapGraph.cpp:428-434                   */
    uint32_t len2;
    uint32_t len1;
    uint32_t start;     /* len1 - overlapLen                            */
    uint32_t j;         /* k-mer position in read1                      */
    uint32_t type;      /* hash-hit type 0..3 (BG/HashTable.cpp:535-566) */
} disco_contained_row;

/* one line of <prefix>_<t>_parGraph.txt (BG/OverlapGraph.cpp:808-867) in read ids, src < dst */
typedef struct disco_edge {
    uint64_t src;
    uint64_t dst;
    uint32_t orient; /* BG/Edge.h:30-34, as seen from src */
    uint32_t offset; /* overlap offset in src              */
    uint32_t len_src;
    uint32_t len_dst;
} disco_edge;

typedef struct disco_counters {
    uint64_t n_reads;
    uint64_t probes;           /* Q : k-mer probes issued                                              */
    uint64_t kmer_hits;        /* H : (probe, index record) pairs with an exact k-mer match, self excluded */
    uint64_t n_contained;      /* C                                                                    */
    uint64_t raw_hits;         /* verified overlap hits before the contained filter / cap             */
    uint64_t e_pre;            /* undirected overlaps in the pre-reduction graph (the metric's unit)   */
    uint64_t e_out;            /* undirected edges after transitive reduction                          */
    uint64_t cap_bind_sites;   /* (read, j) sites where max_edges_per_kmer cut off a valid hit         */
    uint64_t asymmetric_pairs; /* directed finds whose twin was not found from the other read         */
    uint64_t big_rows;         /* reads that took the large-row path in the probe                      */
    uint64_t index_buckets;    /* bucket table size                                                    */
    uint64_t hbm_bytes;        /* device bytes currently allocated by this context                     */
} disco_counters;

/* ---- lifetime -------------------------------------------------------------------------------------------------- */
int disco_abi_version(void);
/* replaces: `new HashTable()` + parameters read by main (BG/main.cpp:42,57) */
int disco_create(int device, const disco_params *p, disco_ctx **out);
void disco_destroy(disco_ctx *ctx);
const char *disco_last_error(const disco_ctx *ctx); /* ctx may be NULL: error of the last failed disco_create */
/* run on a caller-provided hipStream_t (e.g. torch's current stream); NULL restores the context's own stream */
int disco_set_stream(disco_ctx *ctx, void *hip_stream);
int disco_synchronize(disco_ctx *ctx);

/* ---- reads: replaces HashTable::populateReadData / insertIntoTable packing (BG/HashTable.cpp:97-114,423-514) ---- */
/* host helper: pack one upper-case ACGT read into ceil(len/32) words; returns DISCO_E_ARG on a non-ACGT byte
 * (the reference throws std::invalid_argument, BG/HashTable.cpp:474-475) */
int disco_pack_ascii(const char *seq, uint32_t len, uint64_t *out_words);
/* page-locked host memory for the packed reads (so that disco_upload_reads runs at PCIe rate); NULL on failure */
void *disco_host_alloc(size_t bytes);
void disco_host_free(void *p);
/* copy host reads into HBM. packed = [n][stride_words], len[i] in (min_overlap, 32767] (BG/Dataset.cpp:305,
 * 15-bit length field BG/HashTable.cpp:531) */
int disco_upload_reads(disco_ctx *ctx, const uint64_t *packed, uint32_t stride_words, const uint16_t *len, uint64_t n);
/* the same with the reads BACK TO BACK, as the reference itself keeps them (every read packed at its own length,
 * BG/HashTable.cpp:456-477): read i occupies the ceil(len[i] / 32) words that follow those of read i - 1 (disco_pack_ascii's output,
 * one read after the other). A set with a few long reads then costs neither n rows as wide as the longest one on the host nor that
 * many bytes over the link; on the device such a set gets two classes of rows (disco_long_rows) straight from the chunks.
 * disco_stride_words afterwards: ceil(longest / 32) rounded up to a multiple of 8 — the stride disco_download_reads writes. */
int disco_upload_reads_ragged(disco_ctx *ctx, const uint64_t *words, const uint16_t *len, uint64_t n);
/* use reads that are already resident in HBM (caller-owned device pointers, must outlive the context's use) */
int disco_adopt_reads(disco_ctx *ctx, const void *d_packed, uint32_t stride_words, const void *d_len, uint64_t n);
/* copy the packed reads / lengths back to the host (tests, writer) */
int disco_download_reads(disco_ctx *ctx, uint64_t *packed, uint16_t *len);
uint32_t disco_stride_words(const disco_ctx *ctx);
uint64_t disco_num_reads(const disco_ctx *ctx);
/* Two classes of rows. The reference packs every read at its own length (BG/HashTable.cpp:456-477: no stride); a table with one stride
 * pays for its longest read in every row. When a set handed over with a stride of more than 8 words has only a few reads of more
 * than 256 bases (at most one in five: beyond that the rows of one stride are no worse; single GPU, exact overlaps, a window length the minimizer runs
 * are built for), the table is laid out in two classes: 64-byte rows for everybody (a long read's row holds its head; its tail gets a row
 * of its own) plus full rows for the long reads only — the short reads then run exactly the kernels of a pure short set. The uploads
 * and the device input stage pack the classes directly (the table of one stride — n rows as wide as the longest read — is never made
 * on the device); a table that was generated on the device is re-laid by the first disco_build_index. Nothing
 * changes at this interface (disco_stride_words and disco_download_reads keep speaking of the table that was handed over; results
 * are those of the one-stride table, bit for bit); this call tells whether it happened: the number of long reads, 0 if the table
 * has one stride. DISCO_NO_TWO_CLASS=1 keeps one stride. */
uint64_t disco_long_rows(const disco_ctx *ctx);

/* restrict the QUERY side of the probe / edge / reduction phases to reads [lo, hi) (multi-GPU sharding, replaces the
 * per-rank read ranges of MPI/OverlapGraph.cpp:524-528). The index always covers all reads. Default: [0, n). */
int disco_set_query_range(disco_ctx *ctx, uint64_t lo, uint64_t hi);

/* ---- phases ---------------------------------------------------------------------------------------------------- */
/* replaces HashTable::insertDataset: populateReadLengths -> prefix sum -> populateReadData (BG/HashTable.cpp:46-114) */
int disco_build_index(disco_ctx *ctx);
/* fused probe + verify: HashTable::getListOfReads (BG/HashTable.cpp:521-571) for every k-mer of every query read,
 * checkOverlapForContainedRead (BG/OverlapGraph.cpp:517-554) and checkOverlap (:567-595) on every candidate */
int disco_probe(disco_ctx *ctx);
/* replaces OverlapGraph::markContainedReads (BG/OverlapGraph.cpp:333-505); closed form of SURVEY.md §8c-7 */
int disco_mark_contained(disco_ctx *ctx, uint64_t *n_contained);
/* replaces insertAllEdgesOfRead for every non-contained read (BG/OverlapGraph.cpp:631-678) + twin insertion (:614-626)
 * = disco_select_edges + disco_symmetrize(full) + disco_merge_extras */
int disco_build_edges(disco_ctx *ctx, uint64_t *n_pre);
/*   the three steps separately (multi-GPU: select -> export/all-gather/import -> symmetrize -> merge):            */
/*   per-read edge selection: cap of max_edges_per_kmer per k-mer, one edge per destination, sort by offset          */
int disco_select_edges(disco_ctx *ctx);
/*   every find's twin must be in the other read's list (insertEdge, BG/OverlapGraph.cpp:614-626); counts the finds
 *   whose twin is missing (asymmetric pairs) for targets in the query range (full = 0) or for all nodes (full != 0) */
int disco_symmetrize(disco_ctx *ctx, int full, uint64_t *n_asym);
/*   add the missing twins collected by disco_symmetrize to the lists */
int disco_merge_extras(disco_ctx *ctx);
/* replaces markTransitiveEdges / removeTransitiveEdges (BG/OverlapGraph.cpp:687-761)
 * = disco_transitive_mark + disco_emit_edges */
int disco_transitive_reduce(disco_ctx *ctx, uint64_t *n_out);
int disco_transitive_mark(disco_ctx *ctx);
int disco_emit_edges(disco_ctx *ctx, uint64_t *n_out);
/* all five phases back to back on the context's stream */
int disco_run_graph(disco_ctx *ctx);

/* ---- adjacency in and out (order-dependent regime of the multi-GPU flow; tools) ------------------------------------ */
/* adjacency of the local query range after disco_select_edges: deg[i] (uint32) for i in [lo,hi) and the concatenated
 * rows (uint64 entries), exported to caller-provided device buffers / imported from arrays covering ALL nodes
 * (replaces the full-replica partial graphs of MPI/OverlapGraph.cpp:244-279). */
int disco_adjacency_size(disco_ctx *ctx, uint64_t *n_entries);
int disco_export_adjacency(disco_ctx *ctx, void *d_deg_u32, void *d_entries_u64);
int disco_import_adjacency(disco_ctx *ctx, const void *d_deg_u32_all, const void *d_entries_u64_all, uint64_t n_entries_all);

/* ---- multi-GPU flow: one context per GPU (rank), RCCL underneath ------------------------------------------------------
 * Replaces buildG-MPI / buildG-MPIRMA (MPI/main.cpp:29-37 MPI_Init_thread + rank ranges, RMA/HashTable.cpp:95-116 range split of
 * hashData, :422-435 RMA window, :644-653,694-705 MPI_Get per bucket, :1066-1087 needsProcessing ownership,
 * MPI/OverlapGraph.cpp:218-246,473-506 gossip of marked / contained ids). The reads ARRIVE range-partitioned (rank r uploads the ids
 * [r*per, (r+1)*per)); inside a pass they — the graph nodes — are dealt to the ranks by their read-level minimizer (round 5: ranks own
 * loci, disco_dist_info.placement; id ranges for inexact overlaps, the partitioned index and the gather-everything regime);
 * the index is BUILT hash-partitioned (records routed by all-to-all to the owner of their
 * bucket range) and its shards exchanged; containment keys are min-reduced to the owner; the transitive reduction fetches the
 * two or three neighbour rows a node's marking sweeps ON REQUEST from their owners (all-to-all), and surviving half-edges are
 * pushed to the owner of the smaller endpoint. Every call below is COLLECTIVE: all ranks, same order. */
#define DISCO_UNIQUE_ID_BYTES 128
enum { /* exchanges of one pass (disco_dist_info.bytes_sent) */
    DISCO_X_READS = 0,     /* all-gather of the packed reads (only with DISCO_DIST_GATHER_READS)      */
    DISCO_X_INDEX_RECORDS, /* all-to-all: index records to the owner of their bucket range            */
    DISCO_X_INDEX_SHARDS,  /* all-gather-v: built bucket-table and record shards                       */
    DISCO_X_CONTAIN,       /* reduce-scatter(MIN) of the containment keys + all-gather of the bitmap   */
    DISCO_X_ROW_REQUESTS,  /* all-to-all: (node, class) row requests + degrees back                    */
    DISCO_X_ROW_DATA,      /* all-to-all: the requested neighbour rows, 4-byte entries                 */
    DISCO_X_PUSH,          /* all-to-all: surviving half-edges to the owner of the smaller endpoint    */
    DISCO_X_ADJACENCY,     /* regime 1 only: all-gather of the whole adjacency                          */
    DISCO_X_TWINS,         /* regime 2 only: drop bitmap all-gather + all-to-all of {node, twin entry} into reads that dropped a hit */
    DISCO_X_QUERIES,       /* partitioned index only: all-to-all of the lookups (one per minimizer run) to the bucket's owner */
    DISCO_X_HITS,          /* partitioned index only: all-to-all of the matching records back to the read's owner             */
    DISCO_X_KEYS,          /* ranks own loci: all-gather of the 4-byte read-level minimizer keys the reads are dealt by         */
    DISCO_X_READS_DEALT,   /* ranks own loci: all-to-all of the own reads' rows, ahead of the all-gather of all reads            */
    DISCO_X_CONTAIN_KEYS,  /* reduce-scatter(MIN) of the containment keys when it runs BEHIND the pass, on the second communicator
                              (DISCO_X_CONTAIN then counts the bitmaps only: all-gather of every rank's "has a key" bits + the result) */
    DISCO_X_COUNT
};
/* disco_dist_run_graph flags. GATHER_READS: the pass starts from range-partitioned reads. KEEP_INDEX_PARTITIONED: the index is
 * built hash-partitioned and STAYS so — every rank holds its slice of the bucket table and of the records only, the lookups of a
 * rank's reads travel to the owners and the matching records back (replaces the RMA window over the split hashData with its
 * MPI_Get per bucket, RMA/HashTable.cpp:95-116,644-705); default: the built slices are replicated by one all-gather, which is the
 * faster exchange whenever the index fits every GPU. Every rank must pass the same flags
 * (DISCO_DIST_PARTITIONED_INDEX=1 in the environment sets the second one). */
enum { DISCO_DIST_GATHER_READS = 1, DISCO_DIST_KEEP_INDEX_PARTITIONED = 2 };
typedef struct disco_dist_info {
    uint32_t world, rank;
    uint64_t n_reads, own_lo, own_hi;
    uint64_t n_contained, e_pre, e_out;       /* whole job                                                    */
    uint64_t e_out_local;                     /* edges this rank emitted (disco_fetch_edges)                  */
    uint64_t n_contained_local;               /* contained rows this rank holds (disco_fetch_contained)       */
    uint64_t cap_bind_sites, asymmetric_pairs, dropped_hits, probes, kmer_hits; /* whole job                  */
    uint32_t regime;                          /* 0: regular (rows on request); 2: the same after completing the lists of the reads
                                                 that dropped a hit across ranks (per-k-mer cap, second hit to a destination: real
                                                 data at their repeats); 1: adjacency gathered, every rank finishes on its own
                                                 (too many one-sided pairs for 2, inexact overlaps, >= 2^30 reads) */
    uint32_t tr_rounds;                       /* request rounds of the transitive reduction (1 or 2)         */
    uint64_t tr_deferred;                     /* nodes redone after the second round, whole job               */
    uint64_t bytes_sent[DISCO_X_COUNT];       /* this rank's payload bytes to OTHER ranks, last pass          */
    float ms[DISCO_X_COUNT];                  /* host wall time of each exchange on this rank, last pass      */
    float ms_total;
    /* round 4 (appended): what the last pass did on this rank besides its kernels */
    float kernel_ms;                          /* sum of the phase timers (HIP events around the kernels of each phase)              */
    uint32_t comm_ops;                        /* operations issued on the communicator(s): collectives and small host exchanges       */
    uint32_t host_syncs;                      /* blocking waits of the host on the device (stream / event synchronisations)           */
    uint32_t device_allocs, device_frees;     /* requests that reached the HIP runtime DURING the pass (0 with the arena in place)    */
    uint64_t arena_bytes, arena_peak;         /* the context's arena (one allocation before the first collective) and its high water  */
    uint64_t hbm_peak;                        /* most device memory the context's buffers held during the pass (arena or not)         */
    /* round 5 (appended) */
    uint64_t own_reads;                       /* reads (graph nodes) this rank processed: own_hi - own_lo over id ranges; its loci's reads otherwise */
    uint32_t placement;                       /* 0: ranks own id ranges [own_lo, own_hi); 1: ranks own loci — reads dealt by their read-level
                                                 minimizer (own_lo / own_hi then only name the range the rank's reads ARRIVED in)      */
    uint32_t reserved_;
} disco_dist_info;
/* fills out[0..DISCO_UNIQUE_ID_BYTES) on ONE rank (ncclGetUniqueId); the caller hands it to the others (MPI_Bcast-like, any
 * side channel) */
int disco_comm_unique_id(void *out, size_t cap);
/* joins the RCCL communicator of nranks ranks (ncclCommInitRank on the context's device) */
int disco_comm_init(disco_ctx *ctx, const void *unique_id, int nranks, int rank);
/* the ranks are contexts of THIS process driven by one host thread each, on one device or on peers: collectives become
 * device-to-device copies fenced by host barriers (RCCL refuses two ranks on one GPU; this runs the identical multi-rank
 * code path on a single-GPU box: tests, buildG --gpus N --same-device) */
int disco_comm_init_local(disco_ctx *const *ctxs, int nranks);
int disco_comm_rank(const disco_ctx *ctx);  /* 0 without a communicator */
int disco_comm_world(const disco_ctx *ctx); /* 1 without a communicator */
/* the transport behind the context's communicator: "rccl" (disco_comm_init), "loop" (disco_comm_init_local) or "none" */
const char *disco_comm_kind(const disco_ctx *ctx);
/* id range this rank owns of n_total reads */
int disco_dist_range(const disco_ctx *ctx, uint64_t n_total, uint64_t *lo, uint64_t *hi);
/* this rank's reads = rows [lo, hi) of the job's n_total reads (replaces the per-rank file pass of MPI/Dataset.cpp:153-170) */
int disco_dist_upload_reads(disco_ctx *ctx, const uint64_t *packed_own, uint32_t stride_words, const uint16_t *len_own, uint64_t n_total);
/* one whole pass; afterwards disco_fetch_edges / disco_fetch_contained / disco_fetch_edge_files return THIS rank's share */
int disco_dist_run_graph(disco_ctx *ctx, uint32_t flags);
int disco_dist_get_info(disco_ctx *ctx, disco_dist_info *out);

/* ---- input stage on the GPU (SURVEY.md section 8 a-1 … a-3, f-2) ------------------------------------------------------------- */
/* Reads FASTA / FASTQ files itself (parallel pread through pinned staging), finds the records, cleans and filters every read
 * (Dataset::readDataset / testRead, BG/Dataset.cpp:161-380,403-452) and packs the good ones into the context's read table, all on the
 * device: replaces parse + filter + pack on the host cores AND the upload. Read ids = rank among the good reads in file order over the
 * files in the order given (pass the -pe files, then the -se files). The graph results of a previous pass on the context are discarded
 * by the call, whatever it returns. Returns DISCO_E_UNSUPPORTED — the context's reads are unchanged — when a file is not
 * of a form the device stage accepts (FASTA: it must start with '>' and every '>' must begin a line; sequences may be wrapped — at one
 * width per record, or irregularly up to 4096 bases; FASTQ: it starts with '@', records of four lines; not accepted: .gz, empty or
 * unreadable files): the caller then runs its host stage (disco_amd/host/fastx.cpp follows the reference's
 * getline calls literally and produces its error messages) and disco_upload_reads. */
typedef struct disco_ingest_file {
    uint64_t first_index, last_index; /* 1-based file indices of the file's first / last record (every record counts, BG/Dataset.cpp:294) */
    uint64_t good, bad;
} disco_ingest_file;
typedef struct disco_ingest_info {
    uint64_t n_reads, total_records, too_long; /* too_long: otherwise good reads beyond 32767 bases (dropped; BG/HashTable.cpp:531) */
    uint32_t stride_words;                     /* words per read the longest good read needs                                         */
    uint32_t shortest, longest;
    float read_s, device_s;                    /* host wall: files into HBM / everything after                                        */
} disco_ingest_info;
int disco_ingest_fasta(disco_ctx *ctx, const char *const *paths, int n_files, uint32_t host_threads, disco_ingest_info *info, disco_ingest_file *files);
/* lengths and 1-based file indices (for <prefix>_ReadIDMap.txt and the id columns of every output line) of the reads the last
 * disco_ingest_fasta kept: len[n_reads], file_index[n_reads]. The ONE call that may run on a host thread of its own while another
 * thread drives a pass on the same context (it works on the copy stream and shares only the mirrored lengths, under a lock); its
 * error text, if any, goes to the context's error buffer like everybody's — read it after joining the thread */
int disco_ingest_fetch(disco_ctx *ctx, uint16_t *len, uint64_t *file_index);

/* ---- results --------------------------------------------------------------------------------------------------- */
/* optional: the rows start their way to the host NOW, on a side stream (grouped != 0: also in the contained-read files' order),
 * for callers with device work between disco_mark_contained and the fetch (it costs that work 1-3 ms; the fetch then only waits) */
int disco_start_contained_rows(disco_ctx *ctx, int grouped);
/* rows in ascending contained-read id; returns the number of rows written, or a negative error */
int64_t disco_fetch_contained(disco_ctx *ctx, disco_contained_row *out, uint64_t cap);
/* the same rows in the order the contained-read files are written in — grouped by containing read (SG/DataSet.cpp:316-335 needs the
 * rows of one containing read adjacent), ascending (containing read, j, contained read) = the reference's emission order per
 * containing read (BG/OverlapGraph.cpp:438-447). Sorted on the device while the pass goes on; returns DISCO_E_UNSUPPORTED when that
 * did not happen (multi-GPU contexts, more rows than DISCO_EAGER_ROWS_MAX, a containing read with more than 256 rows): the caller
 * then sorts what disco_fetch_contained returns. */
int64_t disco_fetch_contained_grouped(disco_ctx *ctx, disco_contained_row *out, uint64_t cap);
/* edges of the local query range (src < dst), in no particular order; returns count or negative error */
int64_t disco_fetch_edges(disco_ctx *ctx, disco_edge *out, uint64_t cap);
/* substitutions of every edge's overlap, in the order of disco_fetch_edges — the third number of an edge line, "no substitutions"
 * in the reference (BG/OverlapGraph.cpp:815); all 0 unless disco_params.max_substitutions > 0. Computed on the device from the
 * packed reads and the edge geometry. */
int64_t disco_fetch_edge_substitutions(disco_ctx *ctx, uint16_t *out, uint64_t cap);
/* one file index in [0, n_files) per edge, in the order of disco_fetch_edges: the connected components of the reduced graph
 * dealt out to n_files files (large components by size, small ones by hash), so that every node has ALL its edges in one
 * file — what lets the consumer pre-simplify the files independently (SG/OverlapGraphSimple.cpp:344,636-644; the reference
 * gets it from the locality of its BFS batches, BG/OverlapGraph.cpp:100-325) */
int64_t disco_fetch_edge_files(disco_ctx *ctx, uint32_t n_files, uint16_t *out, uint64_t cap);
/* the same partition for edges the HOST holds (buildG --gpus N: the edges all ranks emitted, concatenated; n_nodes = reads of
 * the whole job): out[i] = file of edges[i]. Any context will do; its graph state is not touched. */
int64_t disco_partition_edges(disco_ctx *ctx, const disco_edge *edges, uint64_t n_edges, uint64_t n_nodes, uint32_t n_files, uint16_t *out);
int disco_get_counters(disco_ctx *ctx, disco_counters *out);
/* milliseconds of the last run of each phase, measured with HIP events on the stream the kernels were launched on
 * (index = DISCO_PH_*). DISCO_PH_PROBE_KERNEL / DISCO_PH_VERIFY each bracket exactly one launch of the two longest kernels. */
enum {
    DISCO_PH_INDEX = 0,    /* memset + count + scan + fill                       */
    DISCO_PH_PROBE_KERNEL, /* one launch of probe_kernel<false> (candidate generation) */
    DISCO_PH_VERIFY,       /* one launch of verify_kernel                         */
    DISCO_PH_CONTAIN,
    DISCO_PH_SELECT,       /* edge_select_kernel<false>                          */
    DISCO_PH_CSR,          /* degree scan + row copy                             */
    DISCO_PH_TWIN,         /* twin_check_kernel                                  */
    DISCO_PH_TRMARK,       /* transitive_mark_kernel<false>                      */
    DISCO_PH_EMIT,         /* emit mark + scan + emit fill                       */
    DISCO_PH_ORDER,        /* count + scan + scatter of the grouped processing order (the keys come from the index pass) */
    DISCO_PH_COUNT
};
int disco_phase_ms(disco_ctx *ctx, float *ms, int n);
/* processing order of the query range for the probe and verify passes: a device array of q_hi - q_lo read ids (a permutation
 * of the range; caller-owned, must stay valid), or NULL: the library's own grouping (reads with the same read-level minimizer
 * next to each other; file order for small inputs or with DISCO_NO_ORDER=1). Results do not depend on it; reads that overlap
 * each other processed back to back find the same index buckets and candidate rows in the cache. */
int disco_set_query_order(disco_ctx *ctx, const void *d_order_u64);
/* the order the last disco_probe walked: device pointer to q_hi - q_lo entries owned by the context, read id in bits 31..0 and
 * the read's length in bits 47..32 (the kernels carry the length with the id); NULL = file order */
int disco_get_query_order(disco_ctx *ctx, const void **d_order_u64);
/* ---- the edge lines of the text files, formatted on the GPU ------------------------------------------------------------------ */
/* saveParGraphToFile's lines (BG/OverlapGraph.cpp:808-867: "src \t dst \t orient,ovl,0,0,len1,start1,len1-1,len2,0,ovl-1,NA,flag") for the
 * edges still resident after disco_transitive_reduce, file after file: edge_file = the file of every edge in the order of
 * disco_fetch_edges (disco_fetch_edge_files; may be null for one file), file_index = the 1-based file index of every read (the
 * ids of the text files, SG/DataSet.cpp:103-107; null: read id + 1), flag 2 on every line (the files are cut along connected
 * components). The text of file t is bytes [file_offsets[t], file_offsets[t + 1]) of what disco_fetch_edge_text copies out; returns the
 * total number of bytes. At most 256 files, exact overlaps only (otherwise DISCO_E_UNSUPPORTED: the host writer formats). */
int64_t disco_format_edges(disco_ctx *ctx, uint32_t n_files, const uint16_t *edge_file, const uint64_t *file_index, uint64_t *file_offsets);
int disco_fetch_edge_text(disco_ctx *ctx, char *out, uint64_t cap);
/* ... or straight into the caller's open files: fds[f] receives the bytes of file f (file_offsets[f] .. file_offsets[f + 1] of the
 * last disco_format_edges for n_files files) from its offset 0; host_threads pwrite while the next piece leaves the device */
int disco_write_edge_text(disco_ctx *ctx, const int *fds, uint32_t n_files, uint32_t host_threads);

/* ---- chains of the reduced graph as composite edges ---------------------------------------------------- */
/* The consumer's first step on the files this stage writes is parsimplify: every maximal chain of nodes with exactly two edges that
 * leave them from opposite ends becomes one composite edge carrying the reads inside it (contractParCompositeEdges,
 * SG/OverlapGraphSimple.cpp:69-109,313-500; is_mergeable / mergeEdges, SG/EdgeSimple.cpp:214-272). disco_contract_chains does that
 * contraction on the edges still resident after disco_transitive_reduce — by ranking the chains (pointer jumping), not by walking
 * them — for the overlaps of at least min_overlap_simplify bases (the consumer's load filter, SG/OverlapGraphSimple.cpp:589).
 * Rings made of such nodes only are left alone. disco_amd/host/parsimple.cpp takes the result as its starting point. */
typedef struct disco_chain_edge {
    uint64_t a, b;       /* end nodes (read ids), stored direction a -> b                                  */
    uint64_t offset;     /* sum of the links' offsets                                                      */
    uint32_t orient;     /* (first link & 2) | (last link & 1): mergedEdgeOrientation, SG/EdgeSimple.cpp:272 */
    uint32_t n_links;    /* >= 2                                                                           */
    uint64_t first_link; /* its links are links[first_link .. first_link + n_links)                        */
} disco_chain_edge;
typedef struct disco_chain_link { /* one simple overlap of the chain, in its direction: into read `to` */
    uint32_t to, offset, orient;
} disco_chain_link;
int disco_contract_chains(disco_ctx *ctx, uint32_t min_overlap_simplify, uint64_t *n_composite, uint64_t *n_links);
/* the same for edges held by the host (buildG --gpus N: all ranks' edges, concatenated); the context must hold the reads */
int disco_contract_chains_of(disco_ctx *ctx, const disco_edge *edges, uint64_t n_edges, uint32_t min_overlap_simplify, uint64_t *n_composite,
                             uint64_t *n_links);
/* composite edges, their links, and one byte per edge — in the order of disco_fetch_edges, or of the array given to
 * disco_contract_chains_of — that is 1 if the edge went into a composite edge; null pointers skip an output */
int disco_fetch_chains(disco_ctx *ctx, disco_chain_edge *comp, disco_chain_link *links, uint8_t *edge_absorbed);

/* device-to-device copy on the context's stream (staging for caller-side collectives) */
int disco_memcpy_d2d(disco_ctx *ctx, void *dst, const void *src, uint64_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* DISCO_HIP_H_ */
