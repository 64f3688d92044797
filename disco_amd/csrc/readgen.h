/*
 * readgen.h — deterministic, counter-based synthetic read generator.
 *
 * Replaces bbmap/randomreads.sh (Java, absent on the build/GPU boxes) for the
 * BASELINE configs.  Every quantity is a pure function of (seed, index) built from splitmix64,
 * so the same reads come out of the host C/C++ twin, the numpy twin (disco_amd/readgen.py) and
 * the HIP kernel that writes 2-bit-packed reads straight into HBM (generate_reads_kernel).
 *
 * Model ("uniform-random genome, error-free, both strands"):
 *   genome  = n_contigs contigs of contig_len bases, base g = 2-bit field of mix(seed_g + (g>>5))
 *   read r  : h0 = mix(seed_r + 4r), h1 = mix(seed_r + 4r + 1), h2 = mix(seed_r + 4r + 2)
 *             len    = len_min + h2 % (len_max - len_min + 1)
 *             contig = h0 % n_contigs            (reads never span contigs)
 *             pos    = h1 % (contig_len - len + 1)
 *             strand = h0 >> 63                  (1 = reverse complement of the genome window)
 *   skew = 1 : contig = ((h0 % n_contigs) * (h3 % n_contigs)) / n_contigs with h3 = mix(seed_r + 4r + 3): the product of two
 *             uniform draws, i.e. P(contig <= x n) ~ x (1 - ln x): the first contigs are covered tens of times deeper than the
 *             average, the last ones a fraction of it (abundance spread of a metagenome), integer
 *             arithmetic only so that all twins agree bit for bit.
 * Bases are coded A0 C1 G2 T3 (reference packing, BG/HashTable.h:16-24).
 */
#ifndef DISCO_READGEN_H_
#define DISCO_READGEN_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define DISCO_HD __host__ __device__ __forceinline__
#else
#define DISCO_HD static inline
#endif

typedef struct disco_genspec {
    uint64_t seed;        /* dataset seed                                        */
    uint64_t n_reads;     /* number of reads                                     */
    uint64_t contig_len;  /* bases per contig                                    */
    uint32_t n_contigs;   /* number of contigs                                   */
    uint32_t len_min;     /* shortest read                                       */
    uint32_t len_max;     /* longest read (== len_min for fixed length)          */
    uint32_t skew;        /* bit 0: "metagenome-like" contig abundances (see disco_read_location); 0: contigs equally abundant.
                             bits 1-15 / 16-31: a tail of LONG reads — their length, and their share of the reads in 1 / 65536
                             (DISCO_GEN_LONG_*): which reads are long is a pure function of (seed, read) like everything else */
} disco_genspec;
#define DISCO_GEN_SKEWED(s) ((s)->skew & 1u)
#define DISCO_GEN_LONG_LEN(s) (((s)->skew >> 1) & 0x7FFFu)
#define DISCO_GEN_LONG_SHARE(s) ((s)->skew >> 16)
#define DISCO_GEN_SKEW_WORD(skewed, long_len, long_share) (((skewed) ? 1u : 0u) | ((uint32_t)(long_len) << 1) | ((uint32_t)(long_share) << 16))

DISCO_HD uint64_t disco_mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/* 32 genome bases (MSB-first 2-bit fields) of genome word w */
DISCO_HD uint64_t disco_genome_word(uint64_t seed, uint64_t w)
{
    return disco_mix64((seed * 0xD1342543DE82EF95ull) ^ (w + 0x632BE59BD9B4E019ull));
}

DISCO_HD uint32_t disco_genome_base(uint64_t seed, uint64_t g)
{
    return (uint32_t)(disco_genome_word(seed, g >> 5) >> (62 - 2 * (g & 31))) & 3u;
}

typedef struct disco_readloc {
    uint64_t gpos;    /* absolute genome coordinate of the window start */
    uint32_t len;
    uint32_t strand;  /* 1 = read is the reverse complement of the window */
} disco_readloc;

DISCO_HD disco_readloc disco_read_location(const disco_genspec *s, uint64_t r)
{
    const uint64_t sr = s->seed ^ 0xA5A5A5A55A5A5A5Aull;
    uint64_t h0 = disco_mix64(sr + 4 * r);
    uint64_t h1 = disco_mix64(sr + 4 * r + 1);
    uint64_t h2 = disco_mix64(sr + 4 * r + 2);
    disco_readloc loc;
    loc.len = s->len_min + (uint32_t)(h2 % (uint64_t)(s->len_max - s->len_min + 1));
    if (DISCO_GEN_LONG_SHARE(s) && (uint32_t)(disco_mix64(h2 ^ 0x6C6F6E6772656164ull) & 0xFFFFu) < DISCO_GEN_LONG_SHARE(s)) loc.len = DISCO_GEN_LONG_LEN(s);
    uint64_t contig = (h0 & 0x7FFFFFFFFFFFFFFFull) % s->n_contigs;
    if (DISCO_GEN_SKEWED(s)) contig = (contig * (disco_mix64(sr + 4 * r + 3) % s->n_contigs)) / s->n_contigs;
    uint64_t pos = h1 % (s->contig_len - loc.len + 1);
    loc.gpos = contig * s->contig_len + pos;
    loc.strand = (uint32_t)(h0 >> 63);
    return loc;
}

/* base i (0-based, read orientation) of read r, code 0..3 */
DISCO_HD uint32_t disco_read_base(const disco_genspec *s, const disco_readloc *loc, uint32_t i)
{
    if (!loc->strand)
        return disco_genome_base(s->seed, loc->gpos + i);
    return 3u - disco_genome_base(s->seed, loc->gpos + (loc->len - 1 - i));
}

/* sequencing-error model for the inexact-overlap extension (bench / tests): base p of read r is substituted with probability
 * rate_ppm / 10^6, by one of the three other bases — a pure function of (seed, r, p), like everything above */
DISCO_HD uint32_t disco_substituted_base(uint64_t seed, uint32_t rate_ppm, uint64_t r, uint32_t p, uint32_t base)
{
    const uint64_t h = disco_mix64((seed * 0x9FB21C651E98DF25ull) ^ ((r << 15) | p));
    if ((uint32_t)(h % 1000000u) >= rate_ppm) return base;
    return (base + 1u + (uint32_t)((h >> 40) % 3u)) & 3u;
}

#endif /* DISCO_READGEN_H_ */
