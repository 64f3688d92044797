/* writer.cpp — see writer.h */
#include "writer.h"
#include <fcntl.h>
#include <unistd.h>

#include <omp.h>
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace disco {

namespace {

/* threads of the loops that get no count from their caller: what buildG's -t says (set_writer_threads), else the CPUs this process may
 * run on — NOT omp_get_max_threads(): on a box whose cgroup grants 16 of 256 hardware threads that oversubscribes every loop */
int g_writer_threads = 0;
int writer_threads()
{
    if (g_writer_threads > 0) return g_writer_threads;
    cpu_set_t set;
    CPU_ZERO(&set);
    int n = omp_get_max_threads();
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
    return std::max(n, 1);
}

/* decimal digits two at a time, written in place (the edge files are 2.5 GB of numbers at 45 M edges) */
struct DigitPairs {
    char t[200];
    DigitPairs()
    {
        for (int i = 0; i < 100; i++) {
            t[2 * i] = (char)('0' + i / 10);
            t[2 * i + 1] = (char)('0' + i % 10);
        }
    }
};
const DigitPairs kPairs;

inline char *put_u64(char *p, uint64_t v)
{
    if (v < 10) {
        *p++ = (char)('0' + v);
        return p;
    }
    if (v < 100) {
        memcpy(p, kPairs.t + 2 * v, 2);
        return p + 2;
    }
    if (v < 1000) {
        const uint32_t x = (uint32_t)v;
        *p = (char)('0' + x / 100);
        memcpy(p + 1, kPairs.t + 2 * (x % 100), 2);
        return p + 3;
    }
    int n = 4;
    for (uint64_t x = v / 10000; x; x /= 10) n++;
    char *q = p + n;
    if (v <= 0xFFFFFFFFull) { /* 32-bit divisions */
        uint32_t x = (uint32_t)v;
        while (x >= 100) {
            q -= 2;
            memcpy(q, kPairs.t + 2 * (x % 100), 2);
            x /= 100;
        }
        if (x >= 10) memcpy(q - 2, kPairs.t + 2 * x, 2);
        else *(q - 1) = (char)('0' + x);
        return p + n;
    }
    while (v >= 100) {
        q -= 2;
        memcpy(q, kPairs.t + 2 * (v % 100), 2);
        v /= 100;
    }
    if (v >= 10) memcpy(q - 2, kPairs.t + 2 * v, 2);
    else *(q - 1) = (char)('0' + v);
    return p + n;
}

/* node v belongs to file owner(v): contiguous id ranges. A node is "marked" in its owner's file, where ALL its edges are
 * written (parsimplify contracts marked nodes that have exactly two edges in that file, SG/OverlapGraphSimple.cpp:344) */
inline int owner_of(uint64_t v, uint64_t n, int files) { return (int)((__uint128_t)v * (unsigned)files / (n ? n : 1)); }

bool flush(const std::string &path, const std::string &data, std::string &err)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        err = "Unable to open file: " + path;
        return false;
    }
    if (!data.empty() && fwrite(data.data(), 1, data.size(), f) != data.size()) {
        fclose(f);
        err = "Short write: " + path;
        return false;
    }
    fclose(f);
    return true;
}

} // namespace

void set_writer_threads(int n) { g_writer_threads = n; }

bool write_read_id_map(const std::string &prefix, const ReadSet &rs, std::string &err)
{
    std::string s;
    int npe = 0, nse = 0;
    for (auto &fr : rs.files) { /* BG/Dataset.cpp:115-116,126-127 */
        s += fr.name + (fr.paired ? ": Paired-end file " : ": Singleton file ") + std::to_string(fr.paired ? ++npe : ++nse) + "\nReadID Range: (" +
             std::to_string(fr.first_index) + "," + std::to_string(fr.last_index) + ")\n";
    }
    return flush(prefix + "_ReadIDMap.txt", s, err);
}

FileTags FileTags::plain(int n_files)
{
    FileTags t;
    for (int i = 0; i < n_files; i++) t.tag.push_back(std::to_string(i));
    return t;
}

FileTags FileTags::mpi_edges(int ranks, int threads)
{
    FileTags t;
    for (int r = 0; r < ranks; r++)
        for (int i = (threads > 1 ? 1 : 0); i < std::max(threads, 1); i++) t.tag.push_back(std::to_string(r) + "_" + std::to_string(i));
    return t;
}

FileTags FileTags::mpi_contained(int ranks, int threads)
{
    FileTags t;
    for (int r = 0; r < ranks; r++)
        for (int i = 0; i < std::max(threads, 1); i++) t.tag.push_back(std::to_string(r) + "_" + std::to_string(i));
    return t;
}

static std::string tag_of(const FileTags *tags, int t) { return (tags && (size_t)t < tags->tag.size()) ? tags->tag[(size_t)t] : std::to_string(t); }

/* rows by (containing read, j, contained read): the containing reads are spread evenly over the ids, so a counting pass into 1024
 * id ranges and one std::sort per range — all threads busy — replaces one sort of 4.7 M 40-byte rows (0.42 s at config 3) */
static void sort_contained(std::vector<disco_contained_row> &rows, uint64_t n)
{
    auto less = [](const disco_contained_row &a, const disco_contained_row &b) {
        if (a.super != b.super) return a.super < b.super;
        if (a.j != b.j) return a.j < b.j;
        return a.contained < b.contained;
    };
    const size_t nr = rows.size();
    if (nr < (1u << 16) || n == 0) {
        std::sort(rows.begin(), rows.end(), less);
        return;
    }
    const uint32_t B = 1024;
    auto bucket = [&](const disco_contained_row &r) { return (uint32_t)std::min<uint64_t>((uint64_t)((__uint128_t)r.super * B / n), B - 1); };
    std::vector<uint64_t> start(B + 1, 0);
    std::vector<uint32_t> bk(nr);
#pragma omp parallel for schedule(static) num_threads(writer_threads())
    for (size_t i = 0; i < nr; i++) {
        bk[i] = bucket(rows[i]);
        __atomic_fetch_add(&start[bk[i] + 1], 1ull, __ATOMIC_RELAXED);
    }
    for (uint32_t b = 0; b < B; b++) start[b + 1] += start[b];
    std::vector<disco_contained_row> tmp(nr);
    std::vector<uint64_t> cur(start.begin(), start.end() - 1);
#pragma omp parallel for schedule(static) num_threads(writer_threads())
    for (size_t i = 0; i < nr; i++) tmp[__atomic_fetch_add(&cur[bk[i]], 1ull, __ATOMIC_RELAXED)] = rows[i];
#pragma omp parallel for schedule(dynamic, 4) num_threads(writer_threads())
    for (uint32_t b = 0; b < B; b++) std::sort(tmp.begin() + (ptrdiff_t)start[b], tmp.begin() + (ptrdiff_t)start[b + 1], less);
    rows.swap(tmp);
}

bool write_contained(const std::string &prefix, int n_files, std::vector<disco_contained_row> &rows, const ReadSet &rs, std::string &err,
                     const FileTags *tags, bool grouped)
{
    const bool verbose = getenv("DISCO_VERBOSE") != nullptr;
    double t_last = omp_get_wtime();
    auto lap = [&](const char *what) {
        if (verbose) fprintf(stderr, "[disco host]   write_contained %-18s %.3f s\n", what, omp_get_wtime() - t_last);
        t_last = omp_get_wtime();
    };
    /* rows of one containing read must be contiguous (SG/DataSet.cpp:316-335); the reference emits them per super read in
     * (j, bucket order) = ascending (j, contained id, record kind) */
    if (!grouped) sort_contained(rows, rs.size());
    lap("sort");
    const uint64_t n = rs.size();
    /* rows are sorted by containing read, so every file owns one contiguous run of them: format fixed-size chunks in
     * parallel, then one writer per file */
    const size_t nr = rows.size();
    std::vector<size_t> fbeg((size_t)n_files + 1, nr);
    {
        size_t i = 0;
        for (int t = 0; t < n_files; t++) {
            while (i < nr && owner_of(rows[i].super, n, n_files) < t) i++;
            fbeg[t] = i;
        }
        fbeg[n_files] = nr;
    }
    const size_t CH = 1 << 15;
    std::vector<std::pair<size_t, size_t>> range;
    std::vector<int> chunk_file;
    for (int t = 0; t < n_files; t++)
        for (size_t b = fbeg[t]; b < fbeg[t + 1]; b += CH) {
            range.push_back({b, std::min(b + CH, fbeg[t + 1])});
            chunk_file.push_back(t);
        }
    std::vector<std::string> text(range.size());
#pragma omp parallel for schedule(dynamic, 1) num_threads(writer_threads())
    for (size_t c = 0; c < range.size(); c++) {
        std::string &o = text[c];
        o.reserve((range[c].second - range[c].first) * 48);
        char buf[160];
        for (size_t k = range[c].first; k < range[c].second; k++) {
            const disco_contained_row &r = rows[k];
            char *p = buf;
            p = put_u64(p, rs.file_index[r.contained]); *p++ = '\t';
            p = put_u64(p, rs.file_index[r.super]); *p++ = '\t';
            p = put_u64(p, r.orient); *p++ = ',';
            p = put_u64(p, r.len2); memcpy(p, ",0,0,", 5); p += 5;
            p = put_u64(p, r.len2); memcpy(p, ",0,", 3); p += 3;
            p = put_u64(p, r.len2); *p++ = ',';
            p = put_u64(p, r.len1); *p++ = ',';
            p = put_u64(p, r.start); *p++ = ',';
            p = put_u64(p, (uint64_t)r.start + r.len2); *p++ = '\n';
            o.append(buf, (size_t)(p - buf));
        }
    }
    lap("format");
    std::vector<size_t> first_chunk((size_t)n_files + 1, 0);
    for (size_t c = 0; c < chunk_file.size(); c++) first_chunk[chunk_file[c] + 1] = c + 1;
    for (int t = 0; t < n_files; t++)
        if (first_chunk[t + 1] < first_chunk[t]) first_chunk[t + 1] = first_chunk[t];
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(writer_threads())
    for (int t = 0; t < n_files; t++) {
        const std::string path = prefix + "_" + tag_of(tags, t) + "_containedReads.txt";
        FILE *f = fopen(path.c_str(), "wb");
        bool good = f != nullptr;
        for (size_t c = first_chunk[t]; good && c < first_chunk[t + 1]; c++)
            good = text[c].empty() || fwrite(text[c].data(), 1, text[c].size(), f) == text[c].size();
        if (f) fclose(f);
        if (!good) {
#pragma omp critical
            {
                ok = false;
                err = "Unable to write file: " + path;
            }
        }
    }
    lap("write");
    return ok;
}

bool write_edges(const std::string &prefix, int n_files, const disco_edge *edges, size_t n_edges, const ReadSet &rs, int threads, std::string &err,
                 const uint16_t *edge_file, const FileTags *tags, const uint16_t *edge_subs)
{
    const uint64_t n = rs.size();
    const size_t ne = n_edges;
    if (threads < 1) threads = 1;
    const bool verbose = getenv("DISCO_VERBOSE") != nullptr;
    double t_last = omp_get_wtime();
    auto lap = [&](const char *what) {
        if (verbose) fprintf(stderr, "[disco host]   write_edges %-22s %.3f s\n", what, omp_get_wtime() - t_last);
        t_last = omp_get_wtime();
    };
    /* an edge goes to the file that owns its source and, if different, to the file that owns its destination: bucket the
     * (edge, file, flag) items by file with a counting sort, then format fixed-size chunks of items in parallel */
    std::vector<uint64_t> cnt((size_t)n_files + 1, 0);
    std::vector<int32_t> os(ne), od(ne);
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t i = 0; i < ne; i++) {
        if (edge_file) os[i] = od[i] = std::min<int32_t>(edge_file[i], n_files - 1);
        else {
            os[i] = owner_of(edges[i].src, n, n_files);
            od[i] = owner_of(edges[i].dst, n, n_files);
        }
    }
    for (size_t i = 0; i < ne; i++) {
        cnt[os[i] + 1]++;
        if (od[i] != os[i]) cnt[od[i] + 1]++;
    }
    for (int t = 0; t < n_files; t++) cnt[t + 1] += cnt[t];
    const uint64_t n_items = cnt[n_files];
    std::vector<uint64_t> item(n_items); /* edge index << 2 | flag */
    {
        std::vector<uint64_t> cur(cnt.begin(), cnt.end() - 1);
        for (size_t i = 0; i < ne; i++) {
            /* 2: both ends marked in this file; 0: only column 1; 1: only column 2 (BG/OverlapGraph.cpp:826-833,852-859) */
            if (od[i] == os[i]) item[cur[os[i]]++] = ((uint64_t)i << 2) | 2;
            else {
                item[cur[os[i]]++] = ((uint64_t)i << 2) | 0;
                item[cur[od[i]]++] = ((uint64_t)i << 2) | 1;
            }
        }
    }
    lap("bucket by file");
    const uint64_t CH = 1 << 16;
    const uint64_t n_chunks = (n_items + CH - 1) / CH;
    std::vector<std::string> text(n_chunks);
    /* chunks never straddle files: cut at file boundaries */
    std::vector<std::pair<uint64_t, uint64_t>> range;
    std::vector<int> chunk_file;
    for (int t = 0; t < n_files; t++)
        for (uint64_t b = cnt[t]; b < cnt[t + 1]; b += CH) {
            range.push_back({b, std::min<uint64_t>(b + CH, cnt[t + 1])});
            chunk_file.push_back(t);
        }
    text.assign(range.size(), std::string());
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
    for (size_t c = 0; c < range.size(); c++) {
        std::string &o = text[c];
        o.reserve((range[c].second - range[c].first) * 56);
        char buf[200];
        for (uint64_t k = range[c].first; k < range[c].second; k++) {
            if (k + 16 < range[c].second) { /* the destination's file index is a random 8-byte fetch out of 8 n bytes */
                const disco_edge &nx = edges[item[k + 16] >> 2];
                __builtin_prefetch(&rs.file_index[nx.dst]);
                __builtin_prefetch(&rs.file_index[nx.src]);
            }
            const disco_edge &e = edges[item[k] >> 2];
            const int flag = (int)(item[k] & 3);
            const uint64_t ovl = (uint64_t)e.len_src - e.offset; /* :814 */
            char *p = buf;
            p = put_u64(p, rs.file_index[e.src]); *p++ = '\t';
            p = put_u64(p, rs.file_index[e.dst]); *p++ = '\t';
            p = put_u64(p, e.orient); *p++ = ',';
            p = put_u64(p, ovl); *p++ = ',';
            p = put_u64(p, edge_subs ? edge_subs[item[k] >> 2] : 0); memcpy(p, ",0,", 3); p += 3; /* :815-816 substitutions, edits */
            p = put_u64(p, e.len_src); *p++ = ',';
            p = put_u64(p, e.offset); *p++ = ',';
            p = put_u64(p, e.len_src - 1); *p++ = ',';
            p = put_u64(p, e.len_dst); memcpy(p, ",0,", 3); p += 3;
            p = put_u64(p, ovl - 1); memcpy(p, ",NA,", 4); p += 4;
            *p++ = (char)('0' + flag); *p++ = '\n';
            o.append(buf, (size_t)(p - buf));
        }
    }
    lap("format");
    /* one writer per file, chunks in order */
    std::vector<size_t> first_chunk((size_t)n_files + 1, 0);
    for (size_t c = 0; c < chunk_file.size(); c++) first_chunk[chunk_file[c] + 1] = c + 1;
    for (int t = 0; t < n_files; t++)
        if (first_chunk[t + 1] < first_chunk[t]) first_chunk[t + 1] = first_chunk[t];
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(threads, n_files))
    for (int t = 0; t < n_files; t++) {
        const std::string path = prefix + "_" + tag_of(tags, t) + "_parGraph.txt";
        FILE *f = fopen(path.c_str(), "wb");
        bool good = f != nullptr;
        for (size_t c = first_chunk[t]; good && c < first_chunk[t + 1]; c++)
            good = text[c].empty() || fwrite(text[c].data(), 1, text[c].size(), f) == text[c].size();
        if (f) fclose(f);
        if (!good) {
#pragma omp critical
            {
                ok = false;
                err = "Unable to write file: " + path;
            }
        }
        /* layout compatibility: one start id per file (BG/OverlapGraph.cpp:211) */
        std::string e2;
        uint64_t first = n_files ? (uint64_t)(((__uint128_t)n * (unsigned)t + n_files - 1) / n_files) + 1 : 1;
        flush(prefix + "_" + tag_of(tags, t) + "_startRead.txt", std::to_string(first) + "\n", e2);
    }
    lap("write");
    return ok;
}

bool write_edge_text(const std::string &prefix, int n_files, const char *text, const uint64_t *offsets, uint64_t n_reads, std::string &err, const FileTags *tags)
{
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(writer_threads(), std::max(n_files, 1)))
    for (int t = 0; t < n_files; t++) {
        const std::string path = prefix + "_" + tag_of(tags, t) + "_parGraph.txt";
        FILE *f = fopen(path.c_str(), "wb");
        const uint64_t nb = offsets[t + 1] - offsets[t];
        bool good = f != nullptr && (nb == 0 || fwrite(text + offsets[t], 1, nb, f) == nb);
        if (f) fclose(f);
        if (!good) {
#pragma omp critical
            {
                ok = false;
                err = "Unable to write file: " + path;
            }
        }
        std::string e2; /* layout compatibility: one start id per file (BG/OverlapGraph.cpp:211) */
        const uint64_t first = n_files ? (uint64_t)(((__uint128_t)n_reads * (unsigned)t + n_files - 1) / n_files) + 1 : 1;
        flush(prefix + "_" + tag_of(tags, t) + "_startRead.txt", std::to_string(first) + "\n", e2);
    }
    return ok;
}

/* the edge files of write_edge_text, created empty and handed back open (disco_write_edge_text fills them), with their
 * <prefix>_<tag>_startRead.txt (layout compatibility: one start id per file, BG/OverlapGraph.cpp:211) */
bool open_edge_files(const std::string &prefix, int n_files, const FileTags *tags, uint64_t n_reads, int *fds, std::string &err)
{
    for (int t = 0; t < n_files; t++) {
        const std::string path = prefix + "_" + tag_of(tags, t) + "_parGraph.txt";
        fds[t] = open(path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
        if (fds[t] < 0) {
            err = "Unable to write file: " + path;
            for (int x = 0; x < t; x++) close(fds[x]);
            return false;
        }
        std::string e2;
        const uint64_t first = n_files ? (uint64_t)(((__uint128_t)n_reads * (unsigned)t + n_files - 1) / n_files) + 1 : 1;
        flush(prefix + "_" + tag_of(tags, t) + "_startRead.txt", std::to_string(first) + "\n", e2);
    }
    return true;
}

namespace {
struct BinHeader {
    char magic[8];
    uint32_t version, record_bytes;
    uint64_t n_records;
    uint32_t n_files, reserved;
};
struct BinEdge {
    uint64_t src, dst;
    uint32_t orient, offset, len_src, len_dst;
    uint16_t file, flag;
    uint32_t subs;
};
struct BinContained {
    uint64_t contained, super;
    uint32_t orient, len2, len1, start;
    uint16_t file, pad0;
    uint32_t pad1;
};
static_assert(sizeof(BinHeader) == 32 && sizeof(BinEdge) == 40 && sizeof(BinContained) == 40, "binary side-output layout");

template <typename R>
bool write_records(const std::string &path, const char *magic, const std::vector<R> &rec, int n_files, std::string &err)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) {
        err = "Unable to open file: " + path;
        return false;
    }
    BinHeader h;
    memcpy(h.magic, magic, 8);
    h.version = 1;
    h.record_bytes = (uint32_t)sizeof(R);
    h.n_records = rec.size();
    h.n_files = (uint32_t)n_files;
    h.reserved = 0;
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && (rec.empty() || fwrite(rec.data(), sizeof(R), rec.size(), f) == rec.size());
    fclose(f);
    if (!ok) err = "Short write: " + path;
    return ok;
}
} // namespace

bool write_binary(const std::string &prefix, int n_edge_files, int n_contained_files, const disco_edge *edges, size_t n_edges, const uint16_t *edge_file,
                  std::vector<disco_contained_row> &rows, const ReadSet &rs, std::string &err, const uint16_t *edge_subs)
{
    const uint64_t n = rs.size();
    std::vector<BinEdge> be(n_edges);
#pragma omp parallel for schedule(static) num_threads(writer_threads())
    for (size_t i = 0; i < n_edges; i++) {
        const disco_edge &e = edges[i];
        BinEdge &o = be[i];
        o.src = rs.file_index[e.src];
        o.dst = rs.file_index[e.dst];
        o.orient = e.orient;
        o.offset = e.offset;
        o.len_src = e.len_src;
        o.len_dst = e.len_dst;
        o.file = edge_file ? (uint16_t)std::min<int>(edge_file[i], n_edge_files - 1) : (uint16_t)owner_of(e.src, n, n_edge_files);
        o.flag = 2; /* the files are cut along connected components: both ends have all their edges in this file */
        o.subs = edge_subs ? edge_subs[i] : 0;
    }
    if (!write_records(prefix + "_edges.bin", "DISCOEDG", be, n_edge_files, err)) return false;
    std::vector<BinEdge>().swap(be);
    /* same order as the text files: by containing read, then (j, contained id) */
    sort_contained(rows, n);
    std::vector<BinContained> bc(rows.size());
#pragma omp parallel for schedule(static) num_threads(writer_threads())
    for (size_t i = 0; i < rows.size(); i++) {
        const disco_contained_row &r = rows[i];
        BinContained &o = bc[i];
        o.contained = rs.file_index[r.contained];
        o.super = rs.file_index[r.super];
        o.orient = r.orient;
        o.len2 = r.len2;
        o.len1 = r.len1;
        o.start = r.start;
        o.file = (uint16_t)owner_of(r.super, n, n_contained_files);
        o.pad0 = 0;
        o.pad1 = 0;
    }
    return write_records(prefix + "_contained.bin", "DISCOCON", bc, n_contained_files, err);
}

bool write_checkpoint(const std::string &prefix, bool ccr, bool gc, bool append, std::string &err)
{
    std::ofstream f(prefix + "_CheckpointInfo.txt", append ? std::ios::app : std::ios::trunc);
    if (!f) {
        err = "Unable to open file: " + prefix + "_CheckpointInfo.txt";
        return false;
    }
    if (ccr) f << "CCR=Complete\n";
    if (gc) f << "GC=Complete\n";
    return true;
}

void read_checkpoint(const std::string &prefix, bool &ccr, bool &gc)
{
    ccr = gc = false;
    std::ifstream f(prefix + "_CheckpointInfo.txt");
    std::string line;
    auto trim = [](std::string s) {
        size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
        return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
    };
    while (std::getline(f, line)) { /* BG/main.cpp:178-203 */
        size_t eq = line.find('=');
        if (eq == std::string::npos) continue;
        std::string k = trim(line.substr(0, eq)), v = trim(line.substr(eq + 1));
        if (k == "CCR" && v == "Complete") ccr = true;
        if (k == "GC" && v == "Complete") gc = true;
    }
}

} // namespace disco
