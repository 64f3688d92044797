/* fastx.cpp — see fastx.h.
 *
 * One parallel pass per input file (the reference reads every input three times, serially:
 * BG/Dataset.cpp:161-380, BG/HashTable.cpp:119-337):
 *   the file is mapped, cut into byte ranges, every thread finds the records that START in its range, cleans them
 *   (newlines out, upper case), runs the read filter and remembers (good, length); a prefix sum over the threads gives
 *   every good read its id (= rank in file order) and every record its file index; a second parallel sweep packs the good
 *   reads 2-bit into pinned host memory at their final place.
 * Exactness: the FASTA fast path requires every '>' of the file to be the first byte of a line (then "record = header line
 * + everything up to the next '>'", BG/Dataset.cpp:270-281, is decidable locally); any other file falls back to the
 * sequential splitter below, which follows the reference's getline calls literally.
 */
#include "fastx.h"

#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
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

/* an input file in memory: mmap for plain files, a heap buffer for .gz */
struct Blob {
    const char *data = nullptr;
    size_t n = 0;
    std::string owned;
    void *map = nullptr;
    size_t map_len = 0;
    ~Blob()
    {
        if (map) munmap(map, map_len);
    }
};

bool load_blob(const std::string &path, Blob &b, std::string &err, int threads)
{
    const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0; /* BG/Dataset.cpp:167 */
    if (gz) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) {
            err = "Unable to open file: " + path;
            return false;
        }
        gzbuffer(f, 1 << 20);
        std::vector<char> buf(1 << 22);
        int got;
        while ((got = gzread(f, buf.data(), (unsigned)buf.size())) > 0) b.owned.append(buf.data(), (size_t)got);
        gzclose(f);
        b.data = b.owned.data();
        b.n = b.owned.size();
        return true;
    }
    int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) {
        err = "Unable to open file: " + path;
        return false;
    }
    struct stat st;
    if (fstat(fd, &st) != 0) {
        close(fd);
        err = "Unable to open file: " + path;
        return false;
    }
    b.n = (size_t)st.st_size;
    if (b.n) {
        /* parallel pread into anonymous memory: faulting a file mapping from many threads serialises on the mapping lock */
        void *m = mmap(nullptr, b.n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (m == MAP_FAILED) {
            close(fd);
            err = "Unable to allocate memory for file: " + path;
            return false;
        }
        madvise(m, b.n, MADV_HUGEPAGE);
        b.map = m;
        b.map_len = b.n;
        b.data = (const char *)m;
        const int nt = std::max(1, std::min(threads, 16));
        bool ok = true;
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++) {
            size_t p0 = b.n * (size_t)t / nt, p1 = b.n * (size_t)(t + 1) / nt;
            while (p0 < p1) {
                ssize_t got = pread(fd, (char *)m + p0, std::min<size_t>(p1 - p0, (size_t)1 << 26), (off_t)p0);
                if (got <= 0) {
#pragma omp atomic write
                    ok = false;
                    break;
                }
                p0 += (size_t)got;
            }
        }
        if (!ok) {
            close(fd);
            err = "Unable to read file: " + path;
            return false;
        }
    }
    close(fd);
    return true;
}

/* a record's sequence: up to two pieces of the file (the bytes between them are skipped newlines are handled by clean()) */
struct Rec {
    size_t s, e; /* raw sequence bytes [s, e) in the blob; '\n' inside is dropped when cleaning */
};

/* literal sequential splitter: the reference's getline calls (BG/Dataset.cpp:255-293) */
bool split_sequential(const Blob &b, std::vector<Rec> &recs, std::string &err)
{
    const char *d = b.data;
    const size_t n = b.n;
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
        const char *nl = (const char *)memchr(d + p, '\n', n - p); /* header line */
        p = nl ? (size_t)(nl - d) + 1 : n;
        if (fasta) {
            const char *gt = p < n ? (const char *)memchr(d + p, '>', n - p) : nullptr;
            const size_t q = gt ? (size_t)(gt - d) : n;
            recs.push_back(Rec{p, q});
            p = q < n ? q + 1 : n; /* consume the '>' */
            if (p >= n) break;
        } else {
            size_t s0 = p;
            const char *x = p < n ? (const char *)memchr(d + p, '\n', n - p) : nullptr;
            size_t e0 = x ? (size_t)(x - d) : n;
            p = x ? e0 + 1 : n;
            for (int l = 0; l < 2; l++) { /* '+' line and quality line */
                const char *y = p < n ? (const char *)memchr(d + p, '\n', n - p) : nullptr;
                p = y ? (size_t)(y - d) + 1 : n;
            }
            recs.push_back(Rec{s0, e0});
        }
    }
    return true;
}

/* upper-casing table (BG/Dataset.cpp:303-304 uses toupper in the "C" locale) */
struct UpperTable {
    unsigned char t[256];
    UpperTable()
    {
        for (int i = 0; i < 256; i++) t[i] = (unsigned char)((i >= 'a' && i <= 'z') ? i - 32 : i);
    }
};
const UpperTable kUpper;

/* clean a record into buf (newlines dropped, upper case); returns the length */
inline uint32_t clean(const char *d, const Rec &r, std::string &buf)
{
    buf.resize(r.e - r.s);
    char *o = &buf[0];
    size_t m = 0, p = r.s;
    while (p < r.e) { /* copy line by line (BG/Dataset.cpp:276 removes the newlines) */
        const char *nl = (const char *)memchr(d + p, '\n', r.e - p);
        const size_t q = nl ? (size_t)(nl - d) : r.e;
        for (size_t i = p; i < q; i++) o[m++] = (char)kUpper.t[(unsigned char)d[i]];
        p = q + 1;
    }
    buf.resize(m);
    return (uint32_t)m;
}

/* record starts of a FASTA blob whose every '>' begins a line, found in parallel; false if the precondition fails */
bool fasta_starts_parallel(const Blob &b, int threads, std::vector<std::vector<size_t>> &starts)
{
    const char *d = b.data;
    const size_t n = b.n;
    starts.assign(threads, {});
    bool ok = true;
#pragma omp parallel for schedule(static, 1) num_threads(threads)
    for (int t = 0; t < threads; t++) {
        const size_t b0 = n * (size_t)t / threads, b1 = n * (size_t)(t + 1) / threads;
        size_t p = b0;
        while (p < b1) {
            const char *gt = (const char *)memchr(d + p, '>', b1 - p);
            if (!gt) break;
            const size_t q = (size_t)(gt - d);
            if (q != 0 && d[q - 1] != '\n') {
#pragma omp atomic write
                ok = false;
                break;
            }
            starts[t].push_back(q);
            p = q + 1;
        }
    }
    return ok;
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
        const size_t m = 29;
        if (n < m) return false;
        if ((s[0] == r[0] && memcmp(r, s, m) == 0) || (s[n - m] == r[0] && memcmp(r, s + n - m, m) == 0)) return false;
    }
    thr = (size_t)((double)n * .5);
    for (const char *mo : kMotifs) {
        const size_t m = strlen(mo);
        /* the non-overlapping occurrences of a motif cannot outnumber what the read's base counts allow: if even that bound
         * stays under the threshold the scan is pointless (exact: an upper bound on countSubstring, BG/Common.h:173-183) */
        size_t need[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < m; i++) need[mo[i] == 'A' ? 0 : mo[i] == 'C' ? 1 : mo[i] == 'G' ? 2 : 3]++;
        size_t bound = n;
        for (int b = 0; b < 4; b++)
            if (need[b]) bound = std::min(bound, cnt[b] / need[b]);
        if (bound * m < thr) continue;
        if (covered_by(s, n, mo, m) >= thr) return false;
    }
    return true;
}

bool load_reads(const std::vector<std::string> &pe, const std::vector<std::string> &se, uint32_t min_overlap, int threads,
                ReadSet &out, std::string &err, HostAlloc alloc)
{
    if (threads < 1) threads = 1;
    const bool verbose = getenv("DISCO_VERBOSE") != nullptr;
    double t_last = omp_get_wtime();
    auto lap = [&](const char *what) {
        const double t = omp_get_wtime();
        if (verbose) fprintf(stderr, "[disco host] %-28s %.3f s\n", what, t - t_last);
        t_last = t;
    };
    std::vector<std::pair<std::string, bool>> inputs;
    for (auto &f : pe) inputs.push_back({f, true});
    for (auto &f : se) inputs.push_back({f, false});

    /* ---- pass A: records + (good, length) of every file ---------------------------------------------------------- */
    std::vector<Blob> blobs(inputs.size());
    std::vector<std::vector<Rec>> recs(inputs.size());
    std::vector<std::vector<uint16_t>> glen(inputs.size()); /* 0 = rejected, else the read length */
    std::vector<std::vector<std::vector<uint64_t>>> arenas(inputs.size()); /* [file][thread] packed good reads */
    uint64_t total_records = 0;
    for (size_t fi = 0; fi < inputs.size(); fi++) {
        Blob &b = blobs[fi];
        if (!load_blob(inputs[fi].first, b, err, threads)) return false;
        lap("read file");
        std::vector<Rec> &R = recs[fi];
        bool done = false;
        if (b.n && b.data[0] == '>' && threads > 1) { /* FASTA fast path */
            std::vector<std::vector<size_t>> starts;
            if (fasta_starts_parallel(b, threads, starts)) {
                size_t cnt = 0;
                for (auto &v : starts) cnt += v.size();
                std::vector<size_t> all;
                all.reserve(cnt + 1);
                for (auto &v : starts) all.insert(all.end(), v.begin(), v.end());
                /* a '>' that is the very last byte starts nothing (the reference's next getline fails) unless it is the only
                 * one; it still ends the sequence of the record before it */
                const std::vector<size_t> orig = all;
                if (all.size() > 1 && all.back() == b.n - 1) all.pop_back();
                R.resize(all.size());
#pragma omp parallel for schedule(static) num_threads(threads)
                for (size_t i = 0; i < all.size(); i++) {
                    const size_t end = (i + 1 < orig.size()) ? orig[i + 1] : b.n;
                    const char *nl = (const char *)memchr(b.data + all[i], '\n', end - all[i]);
                    R[i] = Rec{nl ? (size_t)(nl - b.data) + 1 : end, end};
                }
                done = true;
            }
        }
        if (!done) {
            R.clear();
            if (!split_sequential(b, R, err)) return false;
        }
        if (R.empty()) {
            err = "File empty. No reads loaded from " + inputs[fi].first; /* BG/Dataset.cpp:113-114 */
            return false;
        }
        lap("split records");
        glen[fi].assign(R.size(), 0);
        std::vector<uint16_t> &G = glen[fi];
        /* thread t owns the records [nr*t/T, nr*(t+1)/T) in BOTH passes; the good reads are packed right away into the
         * thread's arena (ceil(L/32) words each) so that pass B only has to move words */
        arenas[fi].assign(threads, {});
        const size_t nr0 = R.size();
#pragma omp parallel num_threads(threads)
        {
            std::string buf;
#pragma omp for schedule(static, 1)
            for (int t = 0; t < threads; t++) {
                std::vector<uint64_t> &ar = arenas[fi][t];
                for (size_t i = nr0 * (size_t)t / threads; i < nr0 * (size_t)(t + 1) / threads; i++) {
                    const uint32_t L = clean(b.data, R[i], buf);
                    if (!(L > min_overlap && L <= 32767 && test_read(buf.data(), L))) continue; /* BG/Dataset.cpp:305 */
                    G[i] = (uint16_t)L;
                    const size_t w0 = ar.size(), W = (L + 31) / 32;
                    ar.resize(w0 + W, 0);
                    uint64_t *w = &ar[w0];
                    for (uint32_t x = 0; x < L; x++) {
                        const char ch = buf[x];
                        const uint64_t bb = (ch == 'A') ? 0 : (ch == 'C') ? 1 : (ch == 'G') ? 2 : 3;
                        w[x >> 5] |= bb << (62 - 2 * (x & 31)); /* BG/HashTable.cpp:456-477 */
                    }
                }
            }
        }
        lap("clean + filter");
        FileRange fr;
        fr.name = inputs[fi].first;
        fr.paired = inputs[fi].second;
        fr.first_index = total_records + 1;
        fr.last_index = total_records + R.size();
        fr.good = fr.bad = 0;
        for (uint16_t g : G) (g ? fr.good : fr.bad)++;
        out.files.push_back(fr);
        total_records += R.size();
    }
    out.total_records = total_records;

    /* ---- ids: rank among the good reads in file order; stride from the longest good read ------------------------- */
    uint64_t n = 0;
    uint32_t lo = UINT32_MAX, hi = 0;
    for (auto &G : glen)
        for (uint16_t g : G)
            if (g) {
                n++;
                lo = std::min<uint32_t>(lo, g);
                hi = std::max<uint32_t>(hi, g);
            }
    out.shortest = n ? lo : 0;
    out.longest = hi;
    out.stride_words = std::max<uint32_t>(1, (hi + 31) / 32);
    const uint32_t S = out.stride_words;
    out.n_reads = n;
    out.alloc = alloc;
    const size_t words = (size_t)n * S;
    out.packed = nullptr;
    if (alloc.alloc) out.packed = (uint64_t *)alloc.alloc(std::max<size_t>(words, 1) * 8);
    if (!out.packed) {
        out.packed_fallback.assign(std::max<size_t>(words, 1), 0);
        out.packed = out.packed_fallback.data();
        out.alloc = HostAlloc{};
    }
    out.len.resize(n);
    out.file_index.resize(n);
    lap("allocate packed reads");

    /* ---- pass B: pack the good reads at their final place -------------------------------------------------------- */
    uint64_t id_base = 0, rec_base = 0;
    for (size_t fi = 0; fi < inputs.size(); fi++) {
        const std::vector<Rec> &R = recs[fi];
        const std::vector<uint16_t> &G = glen[fi];
        const int nt = threads;
        std::vector<uint64_t> tbase(nt + 1, 0);
        const size_t nr = R.size();
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++) {
            uint64_t c = 0;
            for (size_t i = nr * (size_t)t / nt; i < nr * (size_t)(t + 1) / nt; i++) c += G[i] != 0;
            tbase[t + 1] = c;
        }
        for (int t = 0; t < nt; t++) tbase[t + 1] += tbase[t];
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++) {
            uint64_t id = id_base + tbase[t];
            const uint64_t *src = arenas[fi][t].data();
            for (size_t i = nr * (size_t)t / nt; i < nr * (size_t)(t + 1) / nt; i++) {
                if (!G[i]) continue;
                const uint32_t L = G[i], W = (L + 31) / 32;
                uint64_t *w = out.packed + (size_t)id * S;
                memcpy(w, src, W * 8);
                for (uint32_t x = W; x < S; x++) w[x] = 0;
                src += W;
                out.len[id] = (uint16_t)L;
                out.file_index[id] = rec_base + i + 1; /* BG/Dataset.cpp:294: every record counts */
                id++;
            }
            std::vector<uint64_t>().swap(arenas[fi][t]);
        }
        id_base += tbase[nt];
        rec_base += nr;
    }
    lap("pack");
    return true;
}

ReadSet::~ReadSet()
{
    if (packed && alloc.free && packed != packed_fallback.data()) alloc.free(packed);
}

} // namespace disco
