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

/* kEndRepeats / kMotifs: the read filter's pattern tables (BG/Dataset.cpp:48-87), shared with the GPU input stage */
#include "../csrc/read_filter_tables.h"

size_t covered_by(const char *s, size_t n, const char *motif, size_t m)
{
    size_t hits = 0;
    if (m == 2 && motif[0] != motif[1]) { /* occurrences of two different letters cannot overlap: a plain count */
        const char x = motif[0], y = motif[1];
        for (size_t p = 0; p + 1 < n; p++) hits += (size_t)((s[p] == x) & (s[p + 1] == y));
        return hits * 2;
    }
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
/* a vector whose resize() does not zero-fill (50 M records are written right away by the threads that found them) */
template <typename T>
struct DefaultInit : std::allocator<T> {
    template <typename U>
    struct rebind {
        using other = DefaultInit<U>;
    };
    template <typename U>
    void construct(U *p)
    {
        ::new ((void *)p) U;
    }
    template <typename U, typename... A>
    void construct(U *p, A &&...a)
    {
        ::new ((void *)p) U(std::forward<A>(a)...);
    }
};
using RecVec = std::vector<Rec, DefaultInit<Rec>>;

bool split_sequential(const Blob &b, RecVec &recs, std::string &err)
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

/* base code of an (upper-case) character: A0 C1 G2 T3 (BG/HashTable.h:16-24), anything else 4 */
struct CodeTable {
    unsigned char t[256];
    uint64_t inc[256]; /* one 16-bit counter per base in a 64-bit word (reads are shorter than 32768), other characters: 0 */
    CodeTable()
    {
        for (int i = 0; i < 256; i++) {
            t[i] = 4;
            inc[i] = 0;
        }
        t['A'] = 0;
        t['C'] = 1;
        t['G'] = 2;
        t['T'] = 3;
        inc['A'] = 1ull;
        inc['C'] = 1ull << 16;
        inc['G'] = 1ull << 32;
        inc['T'] = 1ull << 48;
    }
};
const CodeTable kCode;

/* clean a record into buf (newlines dropped, upper case) and count its bases on the way (cnt[4] = characters that are not
 * ACGT); returns the length */
inline uint32_t clean(const char *d, const Rec &r, std::string &buf, uint32_t cnt[5])
{
    buf.resize(r.e - r.s);
    char *o = &buf[0];
    size_t m = 0, p = r.s;
    uint64_t acc = 0; /* four 16-bit base counters; a record of 32768 or more characters is rejected by its length anyway */
    while (p < r.e) { /* copy line by line (BG/Dataset.cpp:276 removes the newlines) */
        const char *nl = (const char *)memchr(d + p, '\n', r.e - p);
        const size_t q = nl ? (size_t)(nl - d) : r.e;
        if (m + (q - p) < 32768) {
            for (size_t i = p; i < q; i++) {
                const unsigned char c = kUpper.t[(unsigned char)d[i]];
                o[m++] = (char)c;
                acc += kCode.inc[c];
            }
        } else {
            for (size_t i = p; i < q; i++) o[m++] = (char)kUpper.t[(unsigned char)d[i]];
        }
        p = q + 1;
    }
    buf.resize(m);
    cnt[0] = (uint32_t)(acc & 0xFFFF);
    cnt[1] = (uint32_t)((acc >> 16) & 0xFFFF);
    cnt[2] = (uint32_t)((acc >> 32) & 0xFFFF);
    cnt[3] = (uint32_t)(acc >> 48);
    cnt[4] = (uint32_t)(m - std::min<size_t>(m, (size_t)cnt[0] + cnt[1] + cnt[2] + cnt[3])); /* characters that are not ACGT */
    return (uint32_t)m;
}

/* ---- the common record — one line of upper-case ACGT — without the byte-at-a-time passes (x86-64 with AVX2 + BMI2, chosen at run
 * time; anything else, and every record that is not of that form, takes clean() and the table-driven packer) ---------------------- */
#if defined(__x86_64__)
#include <immintrin.h>
#define DISCO_FAST_RECORDS 1
/* base counts of d[s, e) if every byte is one of ACGT; false otherwise (newline inside, lower case, N, ...). Reads up to 31 bytes
 * beyond e: the caller guarantees they exist. */
__attribute__((target("avx2,bmi2,popcnt"))) inline bool count_acgt(const char *d, size_t s, size_t e, uint32_t cnt[5])
{
    const __m256i vA = _mm256_set1_epi8('A'), vC = _mm256_set1_epi8('C'), vG = _mm256_set1_epi8('G'), vT = _mm256_set1_epi8('T');
    uint32_t a = 0, c = 0, g = 0, t = 0;
    for (size_t p = s; p < e; p += 32) {
        const __m256i x = _mm256_loadu_si256((const __m256i *)(d + p));
        const uint32_t in = e - p >= 32 ? 0xFFFFFFFFu : ((1u << (e - p)) - 1u);
        const uint32_t ma = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vA)) & in, mc = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vC)) & in;
        const uint32_t mg = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vG)) & in, mt = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(x, vT)) & in;
        if ((ma | mc | mg | mt) != in) return false;
        a += (uint32_t)_mm_popcnt_u32(ma);
        c += (uint32_t)_mm_popcnt_u32(mc);
        g += (uint32_t)_mm_popcnt_u32(mg);
        t += (uint32_t)_mm_popcnt_u32(mt);
    }
    cnt[0] = a;
    cnt[1] = c;
    cnt[2] = g;
    cnt[3] = t;
    cnt[4] = 0;
    return true;
}

/* 2-bit packing of L upper-case ACGT characters, MSB first (BG/HashTable.cpp:456-477): eight characters per step — (c >> 1) & 3 is
 * A0 C1 T2 G3, one xor swaps the last two — gathered with pext. Reads up to 31 bytes beyond the read. */
__attribute__((target("avx2,bmi2,popcnt"))) inline void pack_acgt(const char *sq, uint32_t L, uint64_t *w)
{
    for (uint32_t x0 = 0; x0 < L; x0 += 32) {
        uint64_t acc = 0;
        for (int q = 0; q < 4; q++) {
            uint64_t x;
            memcpy(&x, sq + x0 + 8 * q, 8);
            x = __builtin_bswap64(x);
            uint64_t t = (x >> 1) & 0x0303030303030303ull;
            t ^= (t >> 1) & 0x0101010101010101ull;
            acc = (acc << 16) | _pext_u64(t, 0x0303030303030303ull);
        }
        const uint32_t nb = std::min<uint32_t>(32, L - x0);
        if (nb < 32) acc &= ~0ull << (2 * (32 - nb)); /* what lies behind the read */
        w[x0 >> 5] = acc;
    }
}
/* positions p < n - 1 with s[p] == x and s[p + 1] == y (the occurrences of a motif of two different letters cannot overlap: a plain
 * count, BG/Common.h:173-183); reads up to 32 bytes beyond s + n */
__attribute__((target("avx2,bmi2,popcnt"))) inline size_t dimer_hits_padded(const char *s, size_t n, char x, char y)
{
    const __m256i vx = _mm256_set1_epi8(x), vy = _mm256_set1_epi8(y);
    size_t hits = 0;
    for (size_t p = 0; p + 1 < n; p += 32) {
        const __m256i a = _mm256_loadu_si256((const __m256i *)(s + p)), b = _mm256_loadu_si256((const __m256i *)(s + p + 1));
        uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(a, vx), _mm256_cmpeq_epi8(b, vy)));
        const size_t left = n - 1 - p; /* pairs that start in this block and end inside the read */
        if (left < 32) m &= (1u << left) - 1u;
        hits += (size_t)_mm_popcnt_u32(m);
    }
    return hits;
}
#else
#define DISCO_FAST_RECORDS 0
#endif

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

/* the patterns of testRead prepared once: first 8 bytes of every end repeat as an integer (a read end is compared with the
 * 38 integers, the 29-byte memcmp only runs on a match), base content and length of every motif */
struct FilterTables {
    uint64_t rep8[sizeof(kEndRepeats) / sizeof(kEndRepeats[0])];
    struct Motif {
        const char *s;
        uint32_t len, need[4];
    } motif[sizeof(kMotifs) / sizeof(kMotifs[0])];
    FilterTables()
    {
        size_t i = 0;
        for (const char *r : kEndRepeats) memcpy(&rep8[i++], r, 8);
        i = 0;
        for (const char *mo : kMotifs) {
            Motif &m = motif[i++];
            m.s = mo;
            m.len = (uint32_t)strlen(mo);
            m.need[0] = m.need[1] = m.need[2] = m.need[3] = 0;
            for (uint32_t x = 0; x < m.len; x++) m.need[kCode.t[(unsigned char)mo[x]]]++;
        }
    }
};
const FilterTables kFilter;

/* Dataset::testRead (BG/Dataset.cpp:403-452) on a cleaned read whose base counts are known (cnt[4] = non-ACGT characters) */
bool test_read_counted(const char *s, size_t n, const uint32_t cnt[5], bool padded)
{
    if (n < 30) return false; /* MIN_READ_SIZE */
    if (cnt[4]) return false; /* :411 */
    size_t thr = (size_t)((double)n * .7);
    for (int b = 0; b < 4; b++)
        if (cnt[b] >= thr) return false;
    {
        const size_t m = 29; /* n >= 30 > m */
        uint64_t head, tail;
        memcpy(&head, s, 8);
        memcpy(&tail, s + n - m, 8);
        size_t i = 0;
        for (const char *r : kEndRepeats) {
            const uint64_t r8 = kFilter.rep8[i++];
            if ((head == r8 && memcmp(r, s, m) == 0) || (tail == r8 && memcmp(r, s + n - m, m) == 0)) return false;
        }
    }
    thr = (size_t)((double)n * .5);
    uint32_t q[7] = {0, 0, 0, 0, 0, 0, 0}; /* q[m] = ceil(thr / m) for the motif lengths 2, 3, 6 */
    q[2] = (uint32_t)((thr + 1) / 2);
    q[3] = (uint32_t)((thr + 2) / 3);
    q[6] = (uint32_t)((thr + 5) / 6);
    for (const FilterTables::Motif &mo : kFilter.motif) {
        /* the non-overlapping occurrences of a motif cannot outnumber what the read's base counts allow
         * (min_b floor(cnt[b] / need[b])): if even that bound times the motif length stays under the threshold the scan is
         * pointless (exact: an upper bound on countSubstring, BG/Common.h:173-183). floor(c / d) < q  <=>  c < d q. */
        const uint32_t qq = mo.len <= 6 ? q[mo.len] : (uint32_t)((thr + mo.len - 1) / mo.len);
        bool skip = false;
        for (int b = 0; b < 4; b++)
            if (mo.need[b] && cnt[b] < mo.need[b] * qq) skip = true;
        if (skip) continue;
#if DISCO_FAST_RECORDS
        if (padded && mo.len == 2 && mo.s[0] != mo.s[1]) { /* 32 readable bytes behind the read: the dimer count 32 positions at a time */
            if (dimer_hits_padded(s, n, mo.s[0], mo.s[1]) * 2 >= thr) return false;
            continue;
        }
#endif
        if (covered_by(s, n, mo.s, mo.len) >= thr) return false;
    }
    return true;
}

bool test_read(const char *s, size_t n)
{
    uint32_t cnt[5] = {0, 0, 0, 0, 0};
    for (size_t i = 0; i < n; i++) cnt[kCode.t[(unsigned char)s[i]]]++;
    return test_read_counted(s, n, cnt, false);
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
#if DISCO_FAST_RECORDS
    const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("popcnt") && !getenv("DISCO_NO_FAST_RECORDS");
#endif
    std::vector<std::pair<std::string, bool>> inputs;
    for (auto &f : pe) inputs.push_back({f, true});
    for (auto &f : se) inputs.push_back({f, false});

    /* ---- pass A: records + (good, length) of every file ---------------------------------------------------------- */
    std::vector<Blob> blobs(inputs.size());
    std::vector<RecVec> recs(inputs.size());
    std::vector<std::vector<uint16_t>> glen(inputs.size()); /* 0 = rejected, else the read length */
    std::vector<std::vector<std::vector<uint64_t>>> arenas(inputs.size()); /* [file][thread] packed good reads */
    uint64_t total_records = 0;
    for (size_t fi = 0; fi < inputs.size(); fi++) {
        Blob &b = blobs[fi];
        if (!load_blob(inputs[fi].first, b, err, threads)) return false;
        lap("read file");
        RecVec &R = recs[fi];
        bool done = false;
        if (b.n && b.data[0] == '>' && threads > 1) { /* FASTA fast path */
            std::vector<std::vector<size_t>> starts;
            if (fasta_starts_parallel(b, threads, starts)) {
                /* every thread turns the starts it found into records, at their place in file order; the end of a record is the next
                 * start, wherever it was found. A '>' that is the very last byte starts nothing (the reference's next getline fails)
                 * unless it is the only one; it still ends the sequence of the record before it */
                std::vector<size_t> base(starts.size() + 1, 0), next_first(starts.size() + 1, b.n);
                for (size_t t = 0; t < starts.size(); t++) base[t + 1] = base[t] + starts[t].size();
                for (size_t t = starts.size(); t-- > 0;) next_first[t] = starts[t].empty() ? next_first[t + 1] : starts[t][0];
                const size_t total = base[starts.size()];
                size_t last_start = 0;
                for (size_t t = starts.size(); t-- > 0;)
                    if (!starts[t].empty()) {
                        last_start = starts[t].back();
                        break;
                    }
                const size_t n_rec = (total > 1 && last_start == b.n - 1) ? total - 1 : total;
                R.resize(n_rec);
#pragma omp parallel for schedule(static, 1) num_threads(threads)
                for (size_t t = 0; t < starts.size(); t++) {
                    const std::vector<size_t> &st = starts[t];
                    for (size_t j = 0; j < st.size(); j++) {
                        const size_t i = base[t] + j;
                        if (i >= n_rec) break;
                        const size_t end = (j + 1 < st.size()) ? st[j + 1] : next_first[t + 1];
                        const char *nl = (const char *)memchr(b.data + st[j], '\n', end - st[j]);
                        R[i] = Rec{nl ? (size_t)(nl - b.data) + 1 : end, end};
                    }
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
                    uint32_t cnt[5];
                    uint32_t L = 0;
                    const char *seq = nullptr;
                    bool fast_rec = false;
#if DISCO_FAST_RECORDS
                    if (fast) { /* one line of upper-case ACGT, used where it lies in the file */
                        const size_t rs0 = R[i].s, re0 = (R[i].e > rs0 && b.data[R[i].e - 1] == '\n') ? R[i].e - 1 : R[i].e;
                        if (re0 - rs0 <= 32767 && re0 + 32 <= b.n && count_acgt(b.data, rs0, re0, cnt)) {
                            L = (uint32_t)(re0 - rs0);
                            seq = b.data + rs0;
                            fast_rec = true;
                        }
                    }
#endif
                    if (!fast_rec) {
                        L = clean(b.data, R[i], buf, cnt);
                        seq = buf.data();
                    }
                    if (L > 32767) { /* the packed layout has the reference's 15-bit length field; the reference itself keeps such reads */
                        if (L > min_overlap && test_read_counted(seq, L, cnt, fast_rec)) {
#pragma omp atomic
                            out.too_long++;
                        }
                        continue;
                    }
                    if (!(L > min_overlap && test_read_counted(seq, L, cnt, fast_rec))) continue; /* BG/Dataset.cpp:305 */
                    G[i] = (uint16_t)L;
                    const size_t w0 = ar.size(), W = (L + 31) / 32;
                    ar.resize(w0 + W, 0);
                    uint64_t *w = &ar[w0];
#if DISCO_FAST_RECORDS
                    if (fast_rec) {
                        pack_acgt(seq, L, w);
                        continue;
                    }
#endif
                    const unsigned char *bs = (const unsigned char *)seq;
                    for (uint32_t x0 = 0; x0 < L; x0 += 32) { /* MSB first, 2 bits per base: BG/HashTable.cpp:456-477 */
                        const uint32_t nb = std::min<uint32_t>(32, L - x0);
                        uint64_t acc = 0;
                        for (uint32_t x = 0; x < nb; x++) acc = (acc << 2) | kCode.t[bs[x0 + x]];
                        w[x0 >> 5] = acc << (2 * (32 - nb));
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
    out.n_reads = n;
    out.alloc = alloc;
    /* the good reads were packed at ceil(L / 32) words each into the arenas, in id order (file, thread): back to back is just that */
    size_t words = 0;
    for (auto &fa : arenas)
        for (auto &ar : fa) words += ar.size();
    out.n_words = words;
    out.packed = nullptr;
    if (alloc.alloc) out.packed = (uint64_t *)alloc.alloc(std::max<size_t>(words, 1) * 8);
    if (!out.packed) {
        out.packed_fallback.reset(new uint64_t[std::max<size_t>(words, 1)]);
        out.packed = out.packed_fallback.get();
        out.alloc = HostAlloc{};
    }
    out.len.resize(n);
    out.file_index.resize(n);
    lap("allocate packed reads");

    /* ---- pass B: the arenas one behind the other; lengths and file indices by id ------------------------------------ */
    uint64_t id_base = 0, rec_base = 0;
    size_t word_base = 0;
    for (size_t fi = 0; fi < inputs.size(); fi++) {
        const RecVec &R = recs[fi];
        const std::vector<uint16_t> &G = glen[fi];
        const int nt = threads;
        std::vector<uint64_t> tbase(nt + 1, 0), wbase(nt + 1, 0);
        const size_t nr = R.size();
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++) {
            uint64_t c = 0;
            for (size_t i = nr * (size_t)t / nt; i < nr * (size_t)(t + 1) / nt; i++) c += G[i] != 0;
            tbase[t + 1] = c;
        }
        for (int t = 0; t < nt; t++) {
            tbase[t + 1] += tbase[t];
            wbase[t + 1] = wbase[t] + arenas[fi][t].size();
        }
#pragma omp parallel for schedule(static, 1) num_threads(nt)
        for (int t = 0; t < nt; t++) {
            uint64_t id = id_base + tbase[t];
            if (!arenas[fi][t].empty()) memcpy(out.packed + word_base + wbase[t], arenas[fi][t].data(), arenas[fi][t].size() * 8);
            for (size_t i = nr * (size_t)t / nt; i < nr * (size_t)(t + 1) / nt; i++) {
                if (!G[i]) continue;
                out.len[id] = G[i];
                out.file_index[id] = rec_base + i + 1; /* BG/Dataset.cpp:294: every record counts */
                id++;
            }
            std::vector<uint64_t>().swap(arenas[fi][t]);
        }
        id_base += tbase[nt];
        rec_base += nr;
        word_base += wbase[nt];
    }
    lap("pack");
    return true;
}

std::vector<uint64_t> ReadSet::word_offsets() const
{
    std::vector<uint64_t> o(n_reads + 1, 0);
    for (uint64_t i = 0; i < n_reads; i++) o[i + 1] = o[i] + ((uint64_t)len[i] + 31) / 32;
    return o;
}

std::vector<uint64_t> ReadSet::rows(uint64_t lo, uint64_t hi, const std::vector<uint64_t> &woff) const
{
    std::vector<uint64_t> r((hi - lo) * (uint64_t)stride_words, 0);
    for (uint64_t i = lo; i < hi; i++) memcpy(r.data() + (i - lo) * stride_words, packed + woff[i], (woff[i + 1] - woff[i]) * 8);
    return r;
}

ReadSet::~ReadSet()
{
    if (packed && alloc.free && packed != packed_fallback.get()) alloc.free(packed);
}

} // namespace disco
