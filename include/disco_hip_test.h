/*
 * disco_hip_test.h — entry points of libdisco_hip.so that exist for bench.py and the tests only: synthetic reads generated in HBM
 * (replaces bbmap/randomreads.sh for the BASELINE configurations), substitution errors, and the two bandwidth probes the bench line
 * quotes beside the nominal HBM peak. Same library, same C rules as include/disco_hip.h; NOT part of the boundary a BuildGraph host
 * binds — the reference has no counterpart for any of them.
 */
#ifndef DISCO_HIP_TEST_H_
#define DISCO_HIP_TEST_H_

#include "disco_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* synthetic reads, see disco_amd/csrc/readgen.h (replaces bbmap/randomreads.sh for the BASELINE configs) */
typedef struct disco_genspec_abi {
    uint64_t seed, n_reads, contig_len;
    uint32_t n_contigs, len_min, len_max, skew; /* skew bit 0: metagenome-like contig abundances; bits 1-15 / 16-31: length of a tail of
                                                   long reads and their share of the reads in 1 / 65536 (csrc/readgen.h) */
} disco_genspec_abi;

/* generate synthetic reads directly in HBM (bench / tests) */
int disco_generate_reads(disco_ctx *ctx, const disco_genspec_abi *spec);
/* substitution errors into the resident reads, in place (bench / tests of the inexact-overlap extension): every base of this
 * context's reads — of the rank's own range in the multi-GPU flow — is replaced by another one with probability rate_ppm / 10^6,
 * a pure function of (seed, read, position) (csrc/readgen.h; numpy twin: disco_amd/readgen.py). Before disco_build_index; not on an
 * uploaded / ingested table that got two classes of rows (disco_long_rows: DISCO_E_UNSUPPORTED). */
int disco_substitute_bases(disco_ctx *ctx, uint64_t seed, uint32_t rate_ppm);
/* every rank generates ITS range of the job's reads (multi-GPU bench: the inputs are range-partitioned in HBM when a step starts) */
int disco_dist_generate_reads(disco_ctx *ctx, const disco_genspec_abi *spec);

/* which probe the last disco_build_index prepared: the 32-bit words of minimizer runs a read got (16 or 32: probe_runs_kernel walks the run
 * lists the index pass left), or 0: no run lists for this shape — round 2's probe derives the windows' minimizers itself. Tests only
 * (round 6: every window of 2 .. 64 m-mers over reads of up to 256 bases has run lists; before, min-overlap 30 / 35 / 40 / 45 / 50 only) */
int disco_probe_run_words(const disco_ctx *ctx);

/* attainable HBM bandwidth on this device (SURVEY.md §8d "Roofline that bounds the path": nominal AND measured): a
 * streaming copy kernel over two scratch buffers of `bytes` each, `reps` timed launches after one warm-up;
 * *gb_per_s = read + written bytes per second / 1e9 of the best launch. Measurement aid for bench.py, no reference
 * counterpart. */
int disco_measure_hbm(disco_ctx *ctx, uint64_t bytes, int reps, double *gb_per_s);
/* attainable bandwidth of the access pattern that dominates the path — one random, 64-byte aligned 64-byte row per lane
 * out of a table of `bytes` (the candidate-row fetch of verify, the bucket walk of probe): *gb_per_s = 64 B x rows
 * fetched per second / 1e9 of the best of `reps` launches. The ceiling the gather-bound kernels are priced against in
 * DESIGN.md, next to the nominal and the streaming figure. */
int disco_measure_gather(disco_ctx *ctx, uint64_t bytes, int reps, double *gb_per_s);

#ifdef __cplusplus
}
#endif
#endif /* DISCO_HIP_TEST_H_ */
