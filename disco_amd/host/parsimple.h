/*
 * parsimple.h — the first step of the reference's SimplifyGraph on the graph the stage has just built.
 *
 * fullsimplify starts by running `parsimplify <edge file> <prefix>_<i>_ParSimpleEdges.txt <minOvl> <threads>` on every edge file
 * (SG/OverlapGraph.cpp:1051-1110) — unless all those files already exist, in which case it loads them and skips the step
 * (SG/OverlapGraph.cpp:1027-1049). parsimplify re-parses the text the stage wrote, contracts every chain of nodes with one way in
 * and one way out into a composite edge and removes weak dead ends, to a fixpoint (SG/OverlapGraphSimple.cpp:224-253:
 * contractParCompositeEdges, then { contractParCompositeEdges_Serial; removeParDeadEndNodes } until nothing changes).
 * write_par_simple does the same on the edges while they are still in memory and writes the files in parsimplify's format
 * (printEdge, SG/OverlapGraphSimple.cpp:658-690):
 *     src \t dst \t orient,offset,offset+len(dst),0,0 \t (read,orientation bit,offset)(…)…        source < destination
 * The result is the same SET of lines parsimplify writes for the same edge file (tests/test_host.py compares them sorted; the
 * reference's own line order depends on its container history, and a ring of absorbable nodes ends in a loop whose direction
 * the reference decides by pointer comparison).
 */
#ifndef DISCO_PARSIMPLE_H_
#define DISCO_PARSIMPLE_H_

#include <cstdint>
#include <string>
#include <vector>

#include "disco_hip.h"
#include "fastx.h"
#include "writer.h"

namespace disco {

struct ParSimpleStats {
    uint64_t edges_in, edges_out, nodes_absorbed, dead_end_nodes, dead_end_edges, rounds;
};

/* the chains of the graph already contracted on the GPU (disco_contract_chains + disco_fetch_chains): the edges flagged in
 * `absorbed` are not loaded, the composite edges are; what remains for the host is the rings, the dead ends and the later rounds */
struct ChainSeed {
    const disco_chain_edge *comp = nullptr;
    uint64_t n_comp = 0;
    const disco_chain_link *links = nullptr;
    const uint8_t *absorbed = nullptr; /* [n_edges] */
};

/* edges / edge_file as for write_edges (file = connected component set, so no chain crosses files); min_ovl_simplify =
 * MinOverlap4SimplifyGraph (disco.cfg:38; edges below it are dropped at load, SG/OverlapGraphSimple.cpp:589) */
bool write_par_simple(const std::string &prefix, int n_files, const disco_edge *edges, size_t n_edges, const uint16_t *edge_file, const ReadSet &rs,
                      uint32_t min_ovl_simplify, int threads, std::string &err, ParSimpleStats *stats = nullptr, const FileTags *tags = nullptr,
                      const std::vector<std::string> *paths = nullptr /* explicit output file names, one per file */,
                      const uint8_t *marked = nullptr /* [n reads] nodes all of whose edges are in their file (flags 0 / 1 / 2 of the edge
                                                         lines, SG/OverlapGraphSimple.cpp:591-644); null: every node, as in buildG's files */,
                      const ChainSeed *seed = nullptr /* only with marked == null (the GPU contraction knows no marks) */);

} // namespace disco
#endif
