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

bool write_read_id_map(const std::string &prefix, const ReadSet &rs, std::string &err);
/* every t in [0, n_files) gets a file, empty ones included (the consumer aborts on a missing file) */
bool write_contained(const std::string &prefix, int n_files, std::vector<disco_contained_row> &rows, const ReadSet &rs, std::string &err);
/* edge_file: file of every edge when both of its ends have ALL their edges there (disco_fetch_edge_files: connected components
 * dealt out to the files) — every line then carries flag 2; nullptr: files own contiguous id ranges, an edge between two
 * files is written to both with flags 0 / 1 */
bool write_edges(const std::string &prefix, int n_files, const disco_edge *edges, size_t n_edges, const ReadSet &rs, int threads, std::string &err,
                 const uint16_t *edge_file = nullptr);
bool write_checkpoint(const std::string &prefix, bool ccr, bool gc, bool append, std::string &err);
void read_checkpoint(const std::string &prefix, bool &ccr, bool &gc);

} // namespace disco
#endif
