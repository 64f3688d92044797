/* fastx.cpp — see fastx.h. Single pass over each input: split -> filter (parallel) -> pack (parallel). */
#include "fastx.h"

#include <omp.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstring>

namespace disco {

namespace {

/* BG/Dataset.cpp:48-85 — 29-mers that disqualify a read when they are its prefix or suffix */
const char *const kEndRepeats[] = {
    "ACACACACACACACACACACACACACACA", "AGAGAGAGAGAGAGAGAGAGAGAGAGAGA", "ATATATATATATATATATATATATATATA", "CGCGCGCGCGCGCGCGCGCGCGCGCGCGC",
    "CTCTCTCTCTCTCTCTCTCTCTCTCTCTC", "AAGAAGAAGAAGAAGAAGAAGAAGAAGAA", "ATAATAATAATAATAATAATAATAATAAT", "TAATAATAATAATAATAATAATAATAATA",
    "AACAACAACAACAACAACAACAACAACAA", "ACAACAACAACAACAACAACAACAACAAC", "CAACAACAACAACAACAACAACAACAACA", "AGAAGAAGAAGAAGAAGAAGAAGAAGAAG",
    "GAAGAAGAAGAAGAAGAAGAAGAAGAAGA", "TTCTTCTTCTTCTTCTTCTTCTTCTTCTT", "AAATAAATAAATAAATAAATAAATAAATA", "TAAATAAATAAATAAATAAATAAATAAAT",
    "ATAAATAAATAAATAAATAAATAAATAAA", "AATAAATAAATAAATAAATAAATAAATAA", "AATTAATTAATTAATTAATTAATTAATTA", "ATTAATTAATTAATTAATTAATTAATTAA",
    "TTAATTAATTAATTAATTAATTAATTAAT", "TAATTAATTAATTAATTAATTAATTAATT", "AAAGAAAGAAAGAAAGAAAGAAAGAAAGA", "AGAAAGAAAGAAAGAAAGAAAGAAAGAAA",
    "GAAAGAAAGAAAGAAAGAAAGAAAGAAAG", "TACATACATACATACATACATACATACAT", "ACATACATACATACATACATACATACATA", "CATACATACATACATACATACATACATAC",
    "ATACATACATACATACATACATACATACA", "GTTTGTTTGTTTGTTTGTTTGTTTGTTTG", "TGTTTGTTTGTTTGTTTGTTTGTTTGTTT", "TTTGTTTGTTTGTTTGTTTGTTTGTTTGT",
    "AGGGAGGGAGGGAGGGAGGGAGGGAGGGA", "GAGGGAGGGAGGGAGGGAGGGAGGGAGGG", "GGAGGGAGGGAGGGAGGGAGGGAGGGAGG", "GGGAGGGAGGGAGGGAGGGAGGGAGGGAG"};
/* BG/Dataset.cpp:87 — motifs whose non-overlapping occurrences may not cover half of the read */
const char *const kMotifs[] = {"AC", "AG", "AT", "CG", "CT", "GT", "AAT", "ATA", "TAA", "AAC", "ACA", "CAA", "AAG", "AGA", "GAA", "GGGGCC"};

size_t covered_by(const char *s, size_t n, const char *motif, size_t m)
{
    size_t hits = 0;
    for (size_t p = 0; p + m <= n;) { /* left-to-right, non-overlapping: BG/Common.h:173-183 */
        if (memcmp(s + p, motif, m) == 0) {
            hits++;
            p += m;
        } else
            p++;
    }
    return hits * m;
}

bool slurp(const std::string &path, std::string &data, std::string &err)
{
    const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0; /* BG/Dataset.cpp:167 */
    if (gz) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) {
            err = "Unable to open file: " + path;
            return false;
        }
        char buf[1 << 16];
        int got;
        while ((got = gzread(f, buf, sizeof buf)) > 0) data.append(buf, (size_t)got);
        gzclose(f);
        return true;
    }
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "Unable to open file: " + path;
        return false;
    }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(sz > 0 ? (size_t)sz : 0);
    size_t rd = sz > 0 ? fread(&data[0], 1, (size_t)sz, f) : 0;
    fclose(f);
    data.resize(rd);
    return true;
}

struct Span {
    size_t off;
    uint32_t len;
};

/* split one file into records; sequences are copied (newline-free) into arena */
bool split_records(const std::string &d, std::string &arena, std::vector<Span> &recs, std::string &err)
{
    const size_t n = d.size();
    if (n == 0) return true;
    bool fasta;
    if (d[0] == '>') fasta = true;
    else if (d[0] == '@') fasta = false;
    else {
        err = "Unknown input file format."; /* BG/Dataset.cpp:267 */
        return false;
    }
    size_t p = 0;
    while (p < n) {
        const char *nl = (const char *)memchr(d.data() + p, '\n', n - p); /* header line */
        p = nl ? (size_t)(nl - d.data()) + 1 : n;
        if (fasta) {
            const size_t start = arena.size();
            size_t q = p;
            while (q < n && d[q] != '>') {
                const char *e = (const char *)memchr(d.data() + q, '\n', n - q);
                size_t line_end = e ? (size_t)(e - d.data()) : n;
                const char *gt = (const char *)memchr(d.data() + q, '>', line_end - q);
                if (gt) line_end = (size_t)(gt - d.data());
                arena.append(d, q, line_end - q);
                q = line_end;
                if (q < n && d[q] == '\n') q++;
            }
            recs.push_back(Span{start, (uint32_t)(arena.size() - start)});
            p = q < n ? q + 1 : n; /* consume the '>' */
            if (p >= n) break;
        } else {
            size_t s[3], e[3];
            for (int l = 0; l < 3; l++) {
                s[l] = p;
                const char *x = p < n ? (const char *)memchr(d.data() + p, '\n', n - p) : nullptr;
                e[l] = x ? (size_t)(x - d.data()) : n;
                p = x ? e[l] + 1 : n;
            }
            const size_t start = arena.size();
            arena.append(d, s[0], e[0] - s[0]);
            recs.push_back(Span{start, (uint32_t)(e[0] - s[0])});
        }
    }
    return true;
}

} // namespace

bool test_read(const char *s, size_t n)
{
    if (n < 30) return false; /* MIN_READ_SIZE */
    size_t cnt[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) {
        switch (s[i]) {
        case 'A': cnt[0]++; break;
        case 'C': cnt[1]++; break;
        case 'G': cnt[2]++; break;
        case 'T': cnt[3]++; break;
        default: return false;
        }
    }
    size_t thr = (size_t)((double)n * .7);
    for (size_t c : cnt)
        if (c >= thr) return false;
    for (const char *r : kEndRepeats) {
        const size_t m = strlen(r);
        if (n < m) return false;
        if (memcmp(r, s, m) == 0 || memcmp(r, s + n - m, m) == 0) return false;
    }
    thr = (size_t)((double)n * .5);
    for (const char *m : kMotifs)
        if (covered_by(s, n, m, strlen(m)) >= thr) return false;
    return true;
}

bool load_reads(const std::vector<std::string> &pe, const std::vector<std::string> &se, uint32_t min_overlap, int threads,
                ReadSet &out, std::string &err)
{
    std::string arena;
    std::vector<Span> recs;
    std::vector<std::pair<std::string, bool>> inputs;
    for (auto &f : pe) inputs.push_back({f, true});
    for (auto &f : se) inputs.push_back({f, false});
    std::vector<size_t> file_first;
    for (auto &in : inputs) {
        std::string data;
        if (!slurp(in.first, data, err)) return false;
        const size_t before = recs.size();
        if (!split_records(data, arena, recs, err)) return false;
        if (recs.size() == before) {
            err = "File empty. No reads loaded from " + in.first; /* BG/Dataset.cpp:113-114 */
            return false;
        }
        FileRange fr;
        fr.name = in.first;
        fr.paired = in.second;
        fr.first_index = before + 1;
        fr.last_index = recs.size();
        fr.good = fr.bad = 0;
        out.files.push_back(fr);
    }
    const size_t nrec = recs.size();
    out.total_records = nrec;
    std::vector<uint8_t> good(nrec, 0);
#pragma omp parallel for schedule(dynamic, 4096) num_threads(threads)
    for (size_t i = 0; i < nrec; i++) {
        char *s = &arena[recs[i].off];
        const uint32_t L = recs[i].len;
        for (uint32_t t = 0; t < L; t++) s[t] = (char)toupper((unsigned char)s[t]);
        good[i] = (L > min_overlap && L <= 32767 && test_read(s, L)) ? 1 : 0;
    }
    std::vector<uint64_t> rank(nrec + 1, 0);
    uint32_t lo = UINT32_MAX, hi = 0;
    for (size_t i = 0; i < nrec; i++) {
        rank[i + 1] = rank[i] + good[i];
        if (good[i]) {
            lo = std::min(lo, recs[i].len);
            hi = std::max(hi, recs[i].len);
        }
    }
    const uint64_t n = rank[nrec];
    for (auto &fr : out.files) {
        fr.good = rank[fr.last_index] - rank[fr.first_index - 1];
        fr.bad = (fr.last_index - fr.first_index + 1) - fr.good;
    }
    out.shortest = n ? lo : 0;
    out.longest = hi;
    out.stride_words = std::max<uint32_t>(1, (hi + 31) / 32);
    out.packed.assign((size_t)n * out.stride_words, 0);
    out.len.resize(n);
    out.file_index.resize(n);
    const uint32_t S = out.stride_words;
#pragma omp parallel for schedule(dynamic, 4096) num_threads(threads)
    for (size_t i = 0; i < nrec; i++) {
        if (!good[i]) continue;
        const uint64_t id = rank[i];
        const char *s = &arena[recs[i].off];
        const uint32_t L = recs[i].len;
        uint64_t *w = &out.packed[(size_t)id * S];
        for (uint32_t t = 0; t < L; t++) {
            const uint64_t b = (s[t] == 'A') ? 0 : (s[t] == 'C') ? 1 : (s[t] == 'G') ? 2 : 3;
            w[t >> 5] |= b << (62 - 2 * (t & 31)); /* BG/HashTable.cpp:456-477 */
        }
        out.len[id] = (uint16_t)L;
        out.file_index[id] = i + 1;
    }
    return true;
}

} // namespace disco
