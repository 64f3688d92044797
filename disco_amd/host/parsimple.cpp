/* parsimple.cpp — see parsimple.h */
#include "parsimple.h"

#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>
#include <tuple>

namespace disco {

namespace {

/* one simple overlap along the direction an edge is stored in: from the previous node of the path to `to` */
struct Link {
    uint32_t to;
    uint32_t offset;
    uint8_t orient; /* bit 1: strand of the link's source, bit 0: strand of its destination (SG/OverlapGraphSimple.cpp:600-609) */
};

struct PEdge {
    uint32_t a, b;             /* stored direction a -> b                                                         */
    uint32_t offset;           /* sum of the links' offsets                                                       */
    uint8_t orient;            /* (first link & 2) | (last link & 1): mergedEdgeOrientation, SG/EdgeSimple.cpp:272 */
    std::vector<Link> links;   /* empty: a simple edge (one implicit link a -> b with orient / offset)            */
};

inline uint8_t twin(uint8_t o) { return (uint8_t)(((o >> 1) ^ 1) | (((o & 1) ^ 1) << 1)); } /* get_twin_orient, SG/EdgeSimple.cpp:277 */

struct Graph {
    const uint16_t *len;       /* read length by id */
    std::vector<PEdge> e;
    /* half-edge h = 2 * edge + dir (dir 1: the reverse b -> a) */
    uint32_t src(uint64_t h) const { return (h & 1) ? e[h >> 1].b : e[h >> 1].a; }
    uint32_t dst(uint64_t h) const { return (h & 1) ? e[h >> 1].a : e[h >> 1].b; }
    uint8_t orient(uint64_t h) const { return (h & 1) ? twin(e[h >> 1].orient) : e[h >> 1].orient; }
    /* reverse offset: dstLen + offset - srcLen (make_nonComposite_reverseEdge, SG/EdgeSimple.cpp:119-120; telescopes for composites) */
    uint32_t offset(uint64_t h) const
    {
        const PEdge &x = e[h >> 1];
        return (h & 1) ? (uint32_t)((int64_t)len[x.b] + x.offset - len[x.a]) : x.offset;
    }
    /* the links of half-edge h in ITS direction, appended to out (prev = the node the walk comes from) */
    void append_links(uint64_t h, std::vector<Link> &out) const
    {
        const PEdge &x = e[h >> 1];
        if (!(h & 1)) {
            if (x.links.empty()) out.push_back(Link{x.b, x.offset, x.orient});
            else out.insert(out.end(), x.links.begin(), x.links.end());
            return;
        }
        if (x.links.empty()) {
            out.push_back(Link{x.a, (uint32_t)((int64_t)len[x.b] + x.offset - len[x.a]), twin(x.orient)});
            return;
        }
        /* reverse a path a -> v1 -> ... -> b : links b -> vk, ..., v1 -> a */
        const size_t k = x.links.size();
        for (size_t i = k; i-- > 0;) {
            const uint32_t from = i ? x.links[i - 1].to : x.a, to = x.links[i].to; /* forward link from -> to */
            out.push_back(Link{from, (uint32_t)((int64_t)len[to] + x.links[i].offset - len[from]), twin(x.links[i].orient)});
        }
    }
};

} // namespace

bool write_par_simple(const std::string &prefix, int n_files, const disco_edge *edges, size_t n_edges, const uint16_t *edge_file, const ReadSet &rs,
                      uint32_t min_ovl_simplify, int threads, std::string &err, ParSimpleStats *stats, const FileTags *tags,
                      const std::vector<std::string> *paths, const uint8_t *marked, const ChainSeed *seed)
{
    if (marked) seed = nullptr;
    const uint64_t n = rs.size();
    if (threads < 1) threads = 1;
    const double t_begin = omp_get_wtime();
    Graph g;
    g.len = rs.len.data();
    std::vector<uint16_t> node_file(n, 0);
    {   /* the edges that pass the overlap filter, in their order: counted and placed by blocks in parallel (45 M at config 3) */
        std::vector<uint64_t> first((size_t)threads + 1, 0);
#pragma omp parallel num_threads(threads)
        {
            const int t = omp_get_thread_num(), nt = omp_get_num_threads();
            const size_t lo = n_edges * (size_t)t / nt, hi = n_edges * (size_t)(t + 1) / nt;
            uint64_t c = 0;
            for (size_t i = lo; i < hi; i++)
                c += edges[i].len_src - edges[i].offset >= min_ovl_simplify && !(seed && seed->absorbed[i]); /* SG/OverlapGraphSimple.cpp:589 */
            first[(size_t)t + 1] = c;
#pragma omp barrier
#pragma omp single
            {
                for (int k = 0; k < nt; k++) first[(size_t)k + 1] += first[(size_t)k];
                g.e.resize(first[(size_t)nt] + (seed ? seed->n_comp : 0));
            }
            uint64_t at = first[(size_t)t];
            for (size_t i = lo; i < hi; i++) {
                const disco_edge &x = edges[i];
                /* all edges of a node lie in one file (connected components), so concurrent writers agree */
                const uint16_t f = edge_file ? (uint16_t)std::min<int>(edge_file[i], n_files - 1) : 0;
                node_file[x.src] = node_file[x.dst] = f;
                if (x.len_src - x.offset < min_ovl_simplify || (seed && seed->absorbed[i])) continue;
                PEdge &p = g.e[at++];
                p.a = (uint32_t)x.src;
                p.b = (uint32_t)x.dst;
                p.offset = x.offset;
                p.orient = (uint8_t)x.orient;
            }
        }
    }
    ParSimpleStats st{};
    st.edges_in = g.e.size();
    if (seed && seed->n_comp) { /* the chains arrive contracted: the composite edges behind the simple ones */
        const uint64_t base = g.e.size() - seed->n_comp;
        uint64_t absorbed_edges = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : absorbed_edges) num_threads(threads)
        for (uint64_t k = 0; k < seed->n_comp; k++) {
            const disco_chain_edge &ce = seed->comp[k];
            PEdge &p = g.e[base + k];
            p.a = (uint32_t)ce.a;
            p.b = (uint32_t)ce.b;
            p.offset = (uint32_t)ce.offset;
            p.orient = (uint8_t)ce.orient;
            p.links.resize(ce.n_links);
            for (uint32_t i = 0; i < ce.n_links; i++) {
                const disco_chain_link &l = seed->links[ce.first_link + i];
                p.links[i] = Link{l.to, l.offset, (uint8_t)l.orient};
            }
            absorbed_edges += ce.n_links;
        }
        st.edges_in += absorbed_edges - seed->n_comp; /* as loaded by the stand-alone step: every simple edge */
        st.nodes_absorbed += absorbed_edges - seed->n_comp;
    }
    const bool verbose = getenv("DISCO_VERBOSE") != nullptr;
    double t_last = t_begin;
    auto lap = [&](const char *what) {
        if (verbose) fprintf(stderr, "[disco host]   parsimple %-24s %.3f s\n", what, omp_get_wtime() - t_last);
        t_last = omp_get_wtime();
    };
    lap("load");
    /* thresholds of the stand-alone parsimplify (it reads no cfg: the compiled defaults, SG/Config.cpp:43-44) */
    const size_t kMinReads = 5;
    const uint32_t kMinLength = 500;
    /* NB the stock reference executable is not deterministic here: EdgeSimple::copyEdge (SG/EdgeSimple.cpp:50-71) copies every member
     * but the two read lengths, so the composite edges its parallel contraction round builds from copies carry uninitialised
     * lengths into this test (2 M-edge file: 571 of 6 000 output lines differ between its -t 1 and -t 8 runs). With that one
     * defect repaired (oracle/Makefile: parsimplify_ref_initlen) it writes exactly the lines this function writes. */

    std::vector<uint64_t> start; /* CSR over half-edges by source */
    std::vector<uint64_t> half;
    std::vector<uint32_t> cnt;
    auto build_csr = [&]() {
        const uint64_t ne = g.e.size();
        cnt.assign(n, 0);
#pragma omp parallel for schedule(static) num_threads(threads)
        for (uint64_t i = 0; i < ne; i++) {
            __atomic_fetch_add(&cnt[g.e[i].a], 1u, __ATOMIC_RELAXED);
            __atomic_fetch_add(&cnt[g.e[i].b], 1u, __ATOMIC_RELAXED);
        }
        start.resize(n + 1);
        start[0] = 0;
        for (uint64_t v = 0; v < n; v++) start[v + 1] = start[v] + cnt[v];
        half.resize(2 * ne);
        std::fill(cnt.begin(), cnt.end(), 0u);
#pragma omp parallel for schedule(static) num_threads(threads)
        for (uint64_t i = 0; i < ne; i++) {
            half[start[g.e[i].a] + __atomic_fetch_add(&cnt[g.e[i].a], 1u, __ATOMIC_RELAXED)] = 2 * i;
            half[start[g.e[i].b] + __atomic_fetch_add(&cnt[g.e[i].b], 1u, __ATOMIC_RELAXED)] = 2 * i + 1;
        }
        /* the order of a node's half-edges decides nothing below (a node's two half-edges are told apart by "not the way back") */
    };
    auto deg = [&](uint32_t v) { return start[v + 1] - start[v]; };

    for (;;) {
        st.rounds++;
        /* ---- contraction (contractParCompositeEdges + _Serial, SG/OverlapGraphSimple.cpp:69-109,313-500): a node with exactly two
         * edges that leave it from opposite ends (is_mergeable, SG/EdgeSimple.cpp:254-270), neither of them a loop, is absorbed;
         * every maximal chain of such nodes becomes ONE composite edge between its two other nodes ------------------------------- */
        build_csr();
        lap("csr");
        std::vector<uint8_t> internal(n, 0);
#pragma omp parallel for schedule(static) num_threads(threads)
        for (uint64_t v = 0; v < n; v++) {
            if (deg((uint32_t)v) != 2) continue;
            const uint64_t h0 = half[start[v]], h1 = half[start[v] + 1];
            if ((h0 >> 1) == (h1 >> 1)) continue; /* the two ends of one loop edge */
            if (g.e[h0 >> 1].a == g.e[h0 >> 1].b || g.e[h1 >> 1].a == g.e[h1 >> 1].b) continue;
            /* "all three nodes must be marked" (SG/OverlapGraphSimple.cpp:87,358-360): marked = all of the node's edges are in this file */
            if (marked && !(marked[v] && marked[g.dst(h0)] && marked[g.dst(h1)])) continue;
            if (((g.orient(h0) >> 1) & 1) != ((g.orient(h1) >> 1) & 1)) internal[v] = 1;
        }
        std::vector<uint8_t> dead_edge(g.e.size(), 0), seen(n, 0);
        std::vector<PEdge> fresh;
        uint64_t merged = 0;
        /* one 32-byte record per absorbable node: its two half-edges, where they lead and whether the walk goes on there — a step of a
         * chain walk then costs one cache miss instead of four (CSR start, half-edge list, edge record, flag): fifty 900 000-node chains
         * (BASELINE config 3: one per contig) are pointer chases that 16 threads cannot split */
        struct Hop {
            uint64_t h[2];
            uint32_t to[2];
            uint8_t go_on[2];
        };
        std::unique_ptr<Hop[]> hop(new Hop[n]); /* not zero-filled: only the records of absorbable nodes are ever read */
#pragma omp parallel for schedule(static) num_threads(threads)
        for (uint64_t v = 0; v < n; v++) {
            if (!internal[v]) continue;
            Hop &p = hop[v];
            for (int i = 0; i < 2; i++) {
                p.h[i] = half[start[v] + i];
                p.to[i] = g.dst(p.h[i]);
                p.go_on[i] = internal[p.to[i]];
            }
        }
        /* from half-edge h (out of a non-internal node, or of a ring's anchor) through internal nodes: the half-edges of the chain in
         * walk order; returns the last one */
        auto trace = [&](uint64_t h, uint32_t origin, std::vector<uint64_t> *path) {
            uint64_t cur = h;
            uint32_t v = g.dst(h);
            bool go = internal[v] && v != origin;
            if (path) path->push_back(cur);
            while (go) {
                const Hop &p = hop[v];
                const int i = ((p.h[0] ^ 1) == cur) ? 1 : 0; /* the half-edge of v that is not the way back */
                cur = p.h[i];
                go = p.go_on[i] && p.to[i] != origin;
                v = p.to[i];
                if (path) path->push_back(cur);
            }
            return cur;
        };
        /* the composite edge of a traced chain; retires what it is made of */
        auto compose = [&](const std::vector<uint64_t> &path, PEdge &out) {
            out.a = g.src(path.front());
            out.b = g.dst(path.back());
            out.links.clear();
            size_t nl = 0;
            for (uint64_t h : path) nl += std::max<size_t>(g.e[h >> 1].links.size(), 1);
            out.links.reserve(nl);
            for (size_t i = 0; i < path.size(); i++) {
                g.append_links(path[i], out.links);
                dead_edge[path[i] >> 1] = 1;
                if (i + 1 < path.size()) seen[g.dst(path[i])] = 1;
            }
            out.offset = 0;
            for (const Link &l : out.links) out.offset += l.offset;
            out.orient = (uint8_t)((out.links.front().orient & 2) | (out.links.back().orient & 1));
        };
#pragma omp parallel num_threads(threads)
        {
            std::vector<PEdge> mine;
            std::vector<uint64_t> path;
            uint64_t my_merged = 0;
#pragma omp for schedule(dynamic, 4096)
            for (uint64_t v = 0; v < n; v++) {
                if (internal[v]) continue;
                for (uint64_t q = start[v]; q < start[v + 1]; q++) {
                    const uint64_t h = half[q];
                    if (!internal[g.dst(h)]) continue;
                    /* each chain is found from both of its ends and built from one: the end with the smaller half-edge */
                    path.clear();
                    const uint64_t last = trace(h, (uint32_t)v, &path);
                    if (h > (last ^ 1)) continue; /* the walk from the other end builds it */
                    PEdge c;
                    compose(path, c);
                    my_merged += path.size() - 1; /* nodes absorbed NOW: the edges of the chain minus one (their links may hold nodes of earlier rounds) */
                    mine.push_back(std::move(c));
                }
            }
#pragma omp critical
            {
                merged += my_merged;
                for (PEdge &c : mine) fresh.push_back(std::move(c));
            }
        }
        lap("chains");
        /* rings made of absorbable nodes only: the reference's sweep in ascending id leaves the largest id holding a loop */
        for (uint64_t v = n; v-- > 0;) {
            if (!internal[v] || seen[v]) continue;
            internal[v] = 0; /* anchor */
            PEdge c;
            std::vector<uint64_t> path;
            /* which way round: the reference decides by comparing pointers (SG/OverlapGraphSimple.cpp:456-470), i.e. arbitrarily; here by
             * the CONTENT of the anchor's two half-edges, so that the order in which the edges arrived (threads, GPU-contracted chains
             * or not) cannot show in the files */
            uint64_t h_first = half[start[v]], h_other = half[start[v] + 1];
            {
                auto key = [&](uint64_t h) { return std::make_tuple(g.dst(h), g.offset(h), g.orient(h), g.e[h >> 1].links.size()); };
                if (key(h_other) < key(h_first)) std::swap(h_first, h_other);
            }
            trace(h_first, (uint32_t)v, &path);
            compose(path, c);
            merged += path.size() - 1;
            seen[v] = 1;
            fresh.push_back(std::move(c));
        }
        if (merged) {
            std::vector<PEdge> next;
            next.reserve(g.e.size() - merged); /* the chains' edges (merged + one per chain) go, one composite edge per chain comes */
            for (uint64_t i = 0; i < g.e.size(); i++)
                if (!dead_edge[i]) next.push_back(std::move(g.e[i]));
            for (PEdge &c : fresh) next.push_back(std::move(c));
            g.e.swap(next);
            st.nodes_absorbed += merged;
        }
        lap("rings + rebuild");
        /* ---- dead ends (removeParDeadEndNodes, SG/OverlapGraphSimple.cpp:136-221): a node all of whose edges are weak — fewer than
         * 5 reads inside, shorter than 500 bp, not a loop — and all enter it or all leave it loses its edges ------------------------ */
        build_csr();
        std::vector<uint8_t> dead_node(n, 0);
        uint64_t n_dead = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_dead) num_threads(threads)
        for (uint64_t v = 0; v < n; v++) {
            if (start[v + 1] == start[v] || (marked && !marked[v])) continue;
            bool weak = true;
            uint64_t in = 0, out = 0;
            for (uint64_t q = start[v]; q < start[v + 1] && weak; q++) {
                const uint64_t h = half[q];
                const PEdge &x = g.e[h >> 1];
                if ((marked && !marked[g.dst(h)]) || x.links.size() >= kMinReads + 1 || g.offset(h) + g.len[g.dst(h)] >= kMinLength || x.a == x.b) weak = false;
                else if ((g.orient(h) >> 1) & 1) out++;
                else in++;
            }
            if (weak && in * out == 0 && in + out > 0) {
                dead_node[v] = 1;
                n_dead++;
            }
        }
        uint64_t removed = 0;
        if (n_dead) {
            std::vector<PEdge> next;
            next.reserve(g.e.size());
            for (PEdge &x : g.e) {
                if (dead_node[x.a] || dead_node[x.b]) removed++;
                else next.push_back(std::move(x));
            }
            g.e.swap(next);
            st.dead_end_nodes += n_dead;
            st.dead_end_edges += removed;
        }
        lap("dead ends");
        if (!merged && !removed) break;
    }
    st.edges_out = g.e.size();
    if (stats) *stats = st;

    /* ---- <prefix>_<t>_ParSimpleEdges.txt (printEdge, SG/OverlapGraphSimple.cpp:658-690): the direction with source < destination --- */
    std::vector<std::vector<uint64_t>> by_file((size_t)n_files);
    for (uint64_t i = 0; i < g.e.size(); i++) by_file[node_file[g.e[i].a]].push_back(i);
    bool ok = true;
#pragma omp parallel for schedule(dynamic, 1) num_threads(std::min(threads, n_files))
    for (int t = 0; t < n_files; t++) {
        std::vector<uint64_t> &ids = by_file[(size_t)t];
        auto key = [&](uint64_t i) {
            const PEdge &x = g.e[i];
            return std::make_pair(std::min(x.a, x.b), std::max(x.a, x.b));
        };
        std::sort(ids.begin(), ids.end(), [&](uint64_t p, uint64_t q) {
            const auto kp = key(p), kq = key(q);
            if (kp != kq) return kp < kq;
            const PEdge &x = g.e[p], &y = g.e[q]; /* parallel edges between the same two nodes: by content, not by the order the threads made them */
            if (x.offset != y.offset) return x.offset < y.offset;
            if (x.links.size() != y.links.size()) return x.links.size() < y.links.size();
            return x.orient < y.orient;
        });
        const std::string path = (paths && (size_t)t < paths->size()) ? (*paths)[(size_t)t]
                                 : prefix + "_" + ((tags && (size_t)t < tags->tag.size()) ? tags->tag[(size_t)t] : std::to_string(t)) + "_ParSimpleEdges.txt";
        FILE *f = fopen(path.c_str(), "wb");
        bool good = f != nullptr;
        std::vector<Link> links;
        std::vector<char> text;
        auto put = [](char *p, uint64_t v) {
            char tmp[24];
            int m = 0;
            do {
                tmp[m++] = (char)('0' + v % 10);
                v /= 10;
            } while (v);
            while (m) *p++ = tmp[--m];
            return p;
        };
        for (size_t k = 0; good && k < ids.size(); k++) {
            const PEdge &x = g.e[ids[k]];
            const uint64_t h = 2 * ids[k] + (x.a <= x.b ? 0 : 1);
            links.clear();
            g.append_links(h, links);
            const uint32_t s = g.src(h), d = g.dst(h), off = g.offset(h);
            text.resize(160 + links.size() * 36);
            char *p = text.data();
            p = put(p, rs.file_index[s]); *p++ = '\t';
            p = put(p, rs.file_index[d]); *p++ = '\t';
            p = put(p, g.orient(h)); *p++ = ',';
            p = put(p, off); *p++ = ',';
            p = put(p, (uint64_t)off + g.len[d]);
            memcpy(p, ",0,0\t", 5); p += 5;
            /* inner reads: the common node of consecutive links with the orientation bit and the offset of the link INTO it
             * (mergeList, SG/EdgeSimple.cpp:214-246) */
            for (size_t i = 0; i + 1 < links.size(); i++) {
                *p++ = '(';
                p = put(p, rs.file_index[links[i].to]); *p++ = ',';
                *p++ = (char)('0' + (links[i].orient & 1)); *p++ = ',';
                p = put(p, links[i].offset); *p++ = ')';
            }
            *p++ = '\n';
            good = fwrite(text.data(), 1, (size_t)(p - text.data()), f) == (size_t)(p - text.data());
        }
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

} // namespace disco
