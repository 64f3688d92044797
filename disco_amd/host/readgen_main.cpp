/* readgen — writes the deterministic synthetic reads of disco_amd/csrc/readgen.h as FASTA (stand-in for bbmap/randomreads.sh,
 * which needs Java). Same reads as disco_generate_reads / disco_amd.readgen for the same spec.
 *   readgen <out.fasta> <n_reads> [read_len=150] [coverage=30] [seed=42] [len_max=read_len] [contig_len=5000000]           */
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../csrc/readgen.h"

int main(int argc, char **argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: readgen <out.fasta> <n_reads> [read_len=150] [coverage=30] [seed=42] [len_max=read_len] [contig_len=5000000] [skew=0]\n");
        return 1;
    }
    disco_genspec s;
    s.n_reads = strtoull(argv[2], nullptr, 10);
    s.len_min = argc > 3 ? (uint32_t)atoi(argv[3]) : 150;
    const double cov = argc > 4 ? atof(argv[4]) : 30.0;
    s.seed = argc > 5 ? strtoull(argv[5], nullptr, 10) : 42;
    s.len_max = argc > 6 ? (uint32_t)atoi(argv[6]) : s.len_min;
    const uint64_t want_contig = argc > 7 ? strtoull(argv[7], nullptr, 10) : 5000000ull;
    s.skew = argc > 8 ? (uint32_t)atoi(argv[8]) : 0;
    /* same sizing rule as disco_amd.readgen.GenSpec.coverage / bench.py */
    const double mean = (s.len_min + s.len_max) / 2.0;
    uint64_t genome = (uint64_t)(s.n_reads * mean / cov);
    uint64_t nc = genome / want_contig;
    if (nc < 1) nc = 1;
    s.n_contigs = (uint32_t)nc;
    uint64_t total = genome;
    if (total < nc * (uint64_t)(s.len_max + 1)) total = nc * (uint64_t)(s.len_max + 1);
    s.contig_len = total / nc;
    if (s.contig_len < s.len_max + 1) s.contig_len = s.len_max + 1;
    FILE *f = fopen(argv[1], "wb");
    if (!f) {
        perror(argv[1]);
        return 2;
    }
    std::vector<char> buf(1 << 22);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::string line;
    for (uint64_t r = 0; r < s.n_reads; r++) {
        disco_readloc loc = disco_read_location(&s, r);
        line.assign(">r");
        line += std::to_string(r + 1);
        line += '\n';
        for (uint32_t i = 0; i < loc.len; i++) line += "ACGT"[disco_read_base(&s, &loc, i)];
        line += '\n';
        fwrite(line.data(), 1, line.size(), f);
    }
    fclose(f);
    fprintf(stderr, "readgen: %llu reads, %u contigs x %llu bp, seed %llu\n", (unsigned long long)s.n_reads, s.n_contigs,
            (unsigned long long)s.contig_len, (unsigned long long)s.seed);
    return 0;
}
