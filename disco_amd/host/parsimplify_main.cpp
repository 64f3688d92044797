/*
 * parsimplify — drop-in for the reference's partial-simplification executable (SG/mainParSimplify.cpp:10-22), same command line:
 *     parsimplify <edge file> <output file> <minOvl> <threads>
 * It reads one <prefix>_<t>_parGraph.txt (format: SG/OverlapGraphSimple.cpp:535-650; the flag of every line says which of its nodes have ALL their edges in this
 * file and may therefore be changed; buildG's files carry flag 2 on every line) and writes the composite-edge file. The work is
 * disco::write_par_simple (parsimple.h), the same code buildG --par-simple runs on the edges while they are still in memory; this
 * executable exists so that the contraction can be checked against the REAL parsimplify on CPU, file against file
 * (tests/test_host.py), and for pipelines that keep the text round trip.
 */
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "parsimple.h"

int main(int argc, char **argv)
{
    if (argc < 5) {
        std::cerr << "usage: parsimplify <edge file> <output file> <minOvl> <threads>\n";
        return 1;
    }
    std::ifstream f(argv[1]);
    if (!f.is_open()) {
        std::cerr << "Unable to open file: " << argv[1] << "\n"; /* SG/OverlapGraphSimple.cpp:535-536 */
        return 1;
    }
    std::vector<disco_edge> edges;
    std::vector<unsigned> flags;
    uint64_t max_id = 0;
    std::string line;
    while (std::getline(f, line)) {
        unsigned long long a, b;
        unsigned o, ovl, s0, s1, l1, st1, sp1, l2;
        if (sscanf(line.c_str(), "%llu\t%llu\t%u,%u,%u,%u,%u,%u,%u,%u", &a, &b, &o, &ovl, &s0, &s1, &l1, &st1, &sp1, &l2) != 10) continue;
        disco_edge e;
        e.src = a;
        e.dst = b;
        e.orient = o;
        e.offset = st1;
        e.len_src = l1;
        e.len_dst = l2;
        edges.push_back(e);
        /* trailing field: 0 = only the source is marked in this file, 1 = only the destination, 2 or absent = both (SG/OverlapGraphSimple.cpp:591-600) */
        unsigned fl = 2;
        const size_t na = line.find(",NA,");
        if (na != std::string::npos) fl = (unsigned)atoi(line.c_str() + na + 4);
        flags.push_back(fl);
        max_id = std::max<uint64_t>(max_id, std::max(a, b));
    }
    disco::ReadSet rs; /* node ids ARE the file indices here */
    rs.n_reads = max_id + 1;
    rs.len.assign(max_id + 1, 0);
    rs.file_index.resize(max_id + 1);
    for (uint64_t i = 0; i <= max_id; i++) rs.file_index[i] = i;
    for (const disco_edge &e : edges) {
        rs.len[e.src] = (uint16_t)e.len_src;
        rs.len[e.dst] = (uint16_t)e.len_dst;
    }
    std::vector<uint8_t> marked(max_id + 1, 0);
    for (size_t i = 0; i < edges.size(); i++) {
        if (flags[i] != 1) marked[edges[i].src] = 1;
        if (flags[i] != 0) marked[edges[i].dst] = 1;
    }
    std::vector<std::string> paths{argv[2]};
    std::string err;
    disco::ParSimpleStats st;
    if (!disco::write_par_simple("", 1, edges.data(), edges.size(), nullptr, rs, (uint32_t)strtoul(argv[3], nullptr, 10), atoi(argv[4]), err, &st, nullptr, &paths, marked.data())) {
        std::cerr << err << "\n";
        return 1;
    }
    std::cout << st.edges_in << " edges loaded, " << st.nodes_absorbed << " nodes absorbed, " << st.dead_end_nodes << " dead-end nodes removed, " << st.edges_out
              << " edges written in " << st.rounds << " rounds\n";
    return 0;
}
