/*
 * fastx.h — host-side read input of the buildG drop-in: FASTA/FASTQ(/gz) record splitting, the read-quality filter and
 * 2-bit packing, every read at its own length and the reads back to back (what disco_upload_reads_ragged takes: a long outlier read
 * costs its own words, not a wider row for everybody).
 *
 * Behaviour follows the reference's Dataset (cited as BG/ = /root/reference/src/BuildGraph/src/):
 *   record splitting   BG/Dataset.cpp:255-294   (file type from the first byte; FASTA record = header line + everything up
 *                                                to the next '>' with '\n' removed; FASTQ record = 4 lines)
 *   file index         BG/Dataset.cpp:294       (1-based over ALL records of all -pe files then all -se files)
 *   keep rule          BG/Dataset.cpp:303-305   (upper-case, len > min_overlap, testRead)
 *   testRead           BG/Dataset.cpp:403-452
 *   read ids           file order among the kept reads (== buildG-MPI, MPI/Dataset.cpp:153-170)
 */
#ifndef DISCO_FASTX_H_
#define DISCO_FASTX_H_

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace disco {

struct FileRange {
    std::string name;
    bool paired;
    uint64_t first_index; /* 1-based file index of its first record */
    uint64_t last_index;  /* file index of its last record          */
    uint64_t good, bad;
};

/* optional allocator for the packed reads (buildG passes pinned host memory so that the upload runs at PCIe rate) */
struct HostAlloc {
    void *(*alloc)(size_t bytes) = nullptr;
    void (*free)(void *p) = nullptr;
};

struct ReadSet {
    uint32_t stride_words = 0;        /* ceil(longest / 32): the stride a table of these reads would have */
    uint64_t n_reads = 0;
    uint64_t n_words = 0;             /* sum of ceil(len / 32) */
    uint64_t *packed = nullptr;       /* [n_words]: read i = the ceil(len[i] / 32) words behind those of read i - 1; from `alloc` or packed_fallback */
    std::unique_ptr<uint64_t[]> packed_fallback; /* NOT zero-filled: the packer writes every word */
    /* word offset of every read, [n + 1] (for the consumers that address single reads: the multi-GPU upload, fastx_dump) */
    std::vector<uint64_t> word_offsets() const;
    /* reads [lo, hi) as rows of stride_words words (zero behind the read): what disco_dist_upload_reads takes */
    std::vector<uint64_t> rows(uint64_t lo, uint64_t hi, const std::vector<uint64_t> &woff) const;
    HostAlloc alloc;
    std::vector<uint16_t> len;        /* [n]                                  */
    std::vector<uint64_t> file_index; /* [n] 1-based index over all records   */
    std::vector<FileRange> files;
    uint64_t total_records = 0;
    uint64_t too_long = 0; /* otherwise good reads dropped only because they exceed 32767 bp (counted among the bad ones) */
    uint32_t shortest = 0, longest = 0;
    uint64_t size() const { return n_reads; }
    ReadSet() = default;
    ReadSet(const ReadSet &) = delete;
    ReadSet &operator=(const ReadSet &) = delete;
    ~ReadSet();
};

/* Dataset::testRead on an upper-cased read */
bool test_read(const char *s, size_t n);

/* Reads every file, filters, packs. Returns false and sets err on failure (unreadable file, unknown format, empty file:
 * BG/Dataset.cpp:113-114,244-245,267). */
bool load_reads(const std::vector<std::string> &pe, const std::vector<std::string> &se, uint32_t min_overlap, int threads,
                ReadSet &out, std::string &err, HostAlloc alloc = HostAlloc());

} // namespace disco
#endif
