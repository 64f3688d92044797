/*
 * writer.h — on-disk outputs of the BuildGraph stage, the drop-in contract with SimplifyGraph:
 *   <prefix>_<t>_parGraph.txt         BG/OverlapGraph.cpp:790-907
 *   <prefix>_<t>_containedReads.txt   BG/OverlapGraph.cpp:438-447
 *   <prefix>_<t>_startRead.txt        BG/OverlapGraph.cpp:201-211
 *   <prefix>_ReadIDMap.txt            BG/Dataset.cpp:103-129
 *   <prefix>_CheckpointInfo.txt       BG/OverlapGraph.cpp:487-493, BG/main.cpp:64-70
 */
#ifndef DISCO_WRITER_H_
#define DISCO_WRITER_H_

#include <cstdint>
#include <string>
#include <vector>

#include "disco_hip.h"
#include "fastx.h"

namespace disco {

/* the "<t>" of <prefix>_<t>_parGraph.txt for every file: "0", "1", ... (buildG, BG/OverlapGraph.cpp:385,899) or "<rank>_<thread>"
 * (the multi-process binaries, MPI/OverlapGraph.cpp:127,370,419,518; runDisco-MPI.sh:165-186 lists edge files for threads
 * 1..t-1 of every rank — thread 0 is the communication thread there — and contained-read files for threads 0..t-1) */
struct FileTags {
    std::vector<std::string> tag;
    static FileTags plain(int n_files);
    static FileTags mpi_edges(int ranks, int threads);     /* ranks * max(threads - 1, 1) files */
    static FileTags mpi_contained(int ranks, int threads); /* ranks * threads files             */
};

/* threads of the writer's loops that take no count of their own (buildG: -t) */
void set_writer_threads(int n);
bool write_read_id_map(const std::string &prefix, const ReadSet &rs, std::string &err);
/* every file gets written, empty ones included (the consumer aborts on a missing file) */
/* grouped: the rows already are in the files' order (containing read, j, contained read): disco_fetch_contained_grouped */
bool write_contained(const std::string &prefix, int n_files, std::vector<disco_contained_row> &rows, const ReadSet &rs, std::string &err,
                     const FileTags *tags = nullptr, bool grouped = false);
/* edge_file: file of every edge when both of its ends have ALL their edges there (disco_fetch_edge_files: connected components
 * dealt out to the files) — every line then carries flag 2; nullptr: files own contiguous id ranges, an edge between two
 * files is written to both with flags 0 / 1 */
/* edge_subs: substitutions of every edge's overlap (disco_fetch_edge_substitutions) for the third number of the line — the
 * reference always writes 0 there (BG/OverlapGraph.cpp:815); nullptr: 0 */
bool write_edges(const std::string &prefix, int n_files, const disco_edge *edges, size_t n_edges, const ReadSet &rs, int threads, std::string &err,
                 const uint16_t *edge_file = nullptr, const FileTags *tags = nullptr, const uint16_t *edge_subs = nullptr);
/* the same files from text the GPU has formatted (disco_format_edges): file t = text[offsets[t], offsets[t + 1]) */
bool open_edge_files(const std::string &prefix, int n_files, const FileTags *tags, uint64_t n_reads, int *fds, std::string &err);
bool write_edge_text(const std::string &prefix, int n_files, const char *text, const uint64_t *offsets, uint64_t n_reads, std::string &err,
                     const FileTags *tags = nullptr);
/* Binary side output: the same content as the text files without the formatting on this side and the parsing on
 * the consumer's — <prefix>_edges.bin / <prefix>_contained.bin, little endian:
 *   header  : char magic[8] ("DISCOEDG" / "DISCOCON"), u32 version = 1, u32 record_bytes, u64 n_records, u32 n_files, u32 reserved
 *   edges   : n x { u64 src, dst (1-based file indices, src < dst); u32 orient, offset (= start1), len_src, len_dst; u16 file; u16 flag; u32 substitutions }  40 B
 *             — the line "src\tdst\torient,len_src-offset,substitutions,0,len_src,offset,len_src-1,len_dst,0,len_src-offset-1,NA,flag" of file `file`
 *   contained: n x { u64 contained, super (file indices); u32 orient, len2, len1, start; u16 file; u16 pad; u32 pad }                  40 B
 *             — the line "contained\tsuper\torient,len2,0,0,len2,0,len2,len1,start,start+len2" of file `file`; rows of one super read adjacent
 * disco_amd/edgefile.py reads them and re-creates the text files byte for byte (tests/test_host.py). */
bool write_binary(const std::string &prefix, int n_edge_files, int n_contained_files, const disco_edge *edges, size_t n_edges, const uint16_t *edge_file,
                  std::vector<disco_contained_row> &rows, const ReadSet &rs, std::string &err, const uint16_t *edge_subs = nullptr);
bool write_checkpoint(const std::string &prefix, bool ccr, bool gc, bool append, std::string &err);
void read_checkpoint(const std::string &prefix, bool &ccr, bool &gc);

} // namespace disco
#endif
