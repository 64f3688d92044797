/*
 * disco_chains.h — the first step of the consumer's pre-simplification on the graph that is still resident:
 * every maximal chain of nodes with exactly two edges that leave them from opposite ends becomes ONE composite edge with the list
 * of the reads inside it — contractParCompositeEdges (SG/OverlapGraphSimple.cpp:69-109,313-500), mergeEdges / mergeList
 * (SG/EdgeSimple.cpp:214-272). The reference walks chain after chain; a genome assembled from error-free reads IS one chain per
 * contig (BASELINE config 3: fifty chains of 900 000 nodes), which is a pointer chase that neither 16 host threads nor 16 000
 * wavefronts can split by walking. Here the chains are RANKED instead (Wyllie's pointer jumping over the directed half-edges: every
 * half-edge learns the end of its chain, its distance to it and the sum of the offsets on the way, in log2(length) rounds), after
 * which every edge of a chain knows the composite edge it belongs to and its place in the list, and writes its link there.
 *
 * Edges come in the emission's slot layout (src[i], entry[i] = offset | dst | orient | len(dst), valid[i], pos[i] = rank among the
 * valid ones). Half-edge h = 2 * slot + dir; dir 1 is the reverse (dst -> src) with the twin orientation and the offset
 * len(dst) + offset - len(src) (make_nonComposite_reverseEdge, SG/EdgeSimple.cpp:119-120). Rings made of absorbable nodes only never
 * terminate and are left alone: the host pass that follows (disco_amd/host/parsimple.cpp) anchors them at their largest id as the
 * reference's sweep does, and runs the rounds after the first (dead ends, further contraction) on what is by then a small graph.
 */
#ifndef DISCO_CHAINS_H_
#define DISCO_CHAINS_H_

#include "disco_kernels.h"

#define CH_NIL 0xFFFFFFFFu

struct ChainView {
    const u64 *src, *ent;
    const u8 *valid;
    const u64 *pos;
    const u16 *len;
    u64 n_slots;
    u32 min_ovl; /* edges with a shorter overlap are not loaded (SG/OverlapGraphSimple.cpp:589) */
};

__device__ __forceinline__ bool ch_kept(const ChainView &g, u64 s)
{
    if (!g.valid[s]) return false;
    return (u32)g.len[g.src[s]] - ADJ_OFF(g.ent[s]) >= g.min_ovl;
}
__device__ __forceinline__ u32 ch_src(const ChainView &g, u32 h) { return (h & 1) ? (u32)ADJ_DST(g.ent[h >> 1]) : (u32)g.src[h >> 1]; }
__device__ __forceinline__ u32 ch_dst(const ChainView &g, u32 h) { return (h & 1) ? (u32)g.src[h >> 1] : (u32)ADJ_DST(g.ent[h >> 1]); }
__device__ __forceinline__ u32 ch_twin(u32 o) { return ((o >> 1) ^ 1u) | (((o & 1u) ^ 1u) << 1); } /* get_twin_orient, SG/EdgeSimple.cpp:277 */
__device__ __forceinline__ u32 ch_orient(const ChainView &g, u32 h)
{
    const u32 o = ADJ_ORI(g.ent[h >> 1]);
    return (h & 1) ? ch_twin(o) : o;
}
__device__ __forceinline__ u32 ch_offset(const ChainView &g, u32 h)
{
    const u64 e = g.ent[h >> 1];
    return (h & 1) ? (u32)((int)ADJ_DLEN(e) + (int)ADJ_OFF(e) - (int)g.len[g.src[h >> 1]]) : ADJ_OFF(e);
}

/* degrees and the first two half-edges of every node (all a node of degree 2 has; in no particular order: nothing depends on it) */
__global__ void ch_degree_kernel(ChainView g, u32 *__restrict__ deg, u32 *__restrict__ he)
{
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s < g.n_slots; s += (u64)gridDim.x * blockDim.x) {
        if (!ch_kept(g, s)) continue;
        const u32 a = (u32)g.src[s], b = (u32)ADJ_DST(g.ent[s]);
        const u32 ka = atomicAdd(&deg[a], 1u), kb = atomicAdd(&deg[b], 1u);
        if (ka < 2) he[2ull * a + ka] = (u32)(2 * s);
        if (kb < 2) he[2ull * b + kb] = (u32)(2 * s + 1);
    }
}

/* is_mergeable (SG/EdgeSimple.cpp:254-270) for the node's two edges: distinct, and one enters where the other leaves */
__global__ void ch_internal_kernel(ChainView g, const u32 *__restrict__ deg, const u32 *__restrict__ he, u64 n, u8 *__restrict__ internal)
{
    u64 v = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; v < n; v += (u64)gridDim.x * blockDim.x) {
        u8 in = 0;
        if (deg[v] == 2) {
            const u32 h0 = he[2 * v], h1 = he[2 * v + 1];
            if ((h0 >> 1) != (h1 >> 1) && ch_dst(g, h0) != (u32)v && ch_dst(g, h1) != (u32)v)
                in = (((ch_orient(g, h0) >> 1) & 1u) != ((ch_orient(g, h1) >> 1) & 1u)) ? 1 : 0;
        }
        internal[v] = in;
    }
}

/* one element of the ranking per half-edge: successor, elements and offsets from here to the successor (exclusive), last element seen */
struct ChRank {
    u32 nxt, cnt, last, pad;
    u64 sum;
};

__global__ void ch_init_kernel(ChainView g, const u8 *__restrict__ internal, const u32 *__restrict__ he, ChRank *__restrict__ r)
{
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s < g.n_slots; s += (u64)gridDim.x * blockDim.x) {
        const bool kept = ch_kept(g, s);
        for (u32 d = 0; d < 2; d++) {
            const u32 h = (u32)(2 * s + d);
            ChRank x;
            x.nxt = CH_NIL;
            x.cnt = 1;
            x.last = h;
            x.pad = 0;
            x.sum = 0;
            if (kept) {
                const u32 v = ch_dst(g, h);
                x.sum = ch_offset(g, h);
                if (internal[v]) x.nxt = (he[2ull * v] == (h ^ 1u)) ? he[2ull * v + 1] : he[2ull * v]; /* not the way back */
            }
            r[h] = x;
        }
    }
}

/* one round of pointer jumping, old -> fresh; *live counts the elements that still jumped */
__global__ void ch_jump_kernel(const ChRank *__restrict__ old, ChRank *__restrict__ fresh, u64 n_half, u32 *__restrict__ live)
{
    u64 h = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 mine = 0;
    for (; h < n_half; h += (u64)gridDim.x * blockDim.x) {
        ChRank x = old[h];
        if (x.nxt != CH_NIL) {
            const ChRank y = old[x.nxt];
            x.cnt += y.cnt;
            x.sum += y.sum;
            x.last = y.last;
            x.nxt = y.nxt;
            mine++;
        }
        fresh[h] = x;
    }
    if (mine) atomicAdd(live, mine);
}

/* which direction of the chain a slot lies in builds the composite edge: the one whose FIRST half-edge is the smaller of the chain's
 * two end half-edges, numbered by the edge's rank in fetch order — the rule of the host pass, whose edges are in that order */
__device__ __forceinline__ u64 ch_rankid(const ChainView &g, u32 h) { return 2 * g.pos[h >> 1] + (h & 1u); }

struct ChSlot { /* what a slot of a chain knows after the ranking */
    u32 head;  /* first half-edge of the chain in the building direction */
    u32 mine;  /* this slot's half-edge in that direction               */
    u32 index; /* its place in the list                                  */
    bool in_chain;
};
__device__ __forceinline__ ChSlot ch_slot(const ChainView &g, const u8 *internal, const ChRank *r, u64 s)
{
    ChSlot o;
    o.in_chain = false;
    o.head = o.mine = o.index = 0;
    if (!ch_kept(g, s)) return o;
    const u32 h0 = (u32)(2 * s), h1 = h0 + 1;
    if (!internal[ch_dst(g, h0)] && !internal[ch_dst(g, h1)]) return o;
    const ChRank r0 = r[h0], r1 = r[h1];
    if (r0.nxt != CH_NIL || r1.nxt != CH_NIL) return o; /* a ring of absorbable nodes: never terminates, left to the host pass */
    const u32 H0 = r1.last ^ 1u, T0 = r0.last; /* direction 0: from H0 ... h0 ... to T0 */
    const bool dir0 = !(ch_rankid(g, H0) > ch_rankid(g, T0 ^ 1u));
    o.in_chain = true;
    o.head = dir0 ? H0 : (T0 ^ 1u);
    o.mine = dir0 ? h0 : h1;
    o.index = (dir0 ? r1.cnt : r0.cnt) - 1u; /* elements from the reverse of this one to the reverse list's end = place from the head */
    return o;
}

/* per slot: 1 and the chain's length if the slot holds the first link of a composite edge; the edge dies either way */
__global__ void ch_heads_kernel(ChainView g, const u8 *__restrict__ internal, const ChRank *__restrict__ r, u8 *__restrict__ is_head, u32 *__restrict__ links_of,
                                u8 *__restrict__ dead)
{
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s < g.n_slots; s += (u64)gridDim.x * blockDim.x) {
        const ChSlot c = ch_slot(g, internal, r, s);
        const bool head = c.in_chain && c.index == 0;
        is_head[s] = head ? 1 : 0;
        links_of[s] = head ? r[c.head].cnt : 0u;
        if (c.in_chain) dead[g.pos[s]] = 1;
    }
}

struct ChainEdgeOut { /* = disco_chain_edge */
    u64 a, b, offset;
    u32 orient, n_links;
    u64 first_link;
};
struct ChainLinkOut { /* = disco_chain_link: the overlap INTO read `to` along the composite edge */
    u32 to, offset, orient;
};

__global__ void ch_emit_kernel(ChainView g, const u8 *__restrict__ internal, const ChRank *__restrict__ r, const u64 *__restrict__ comp_id,
                               const u64 *__restrict__ link_start, ChainEdgeOut *__restrict__ comp, ChainLinkOut *__restrict__ links)
{
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s < g.n_slots; s += (u64)gridDim.x * blockDim.x) {
        const ChSlot c = ch_slot(g, internal, r, s);
        if (!c.in_chain) continue;
        const u64 hs = c.head >> 1; /* the slot that holds the composite's first link: its scan values name the composite */
        const u64 first = link_start[hs];
        ChainLinkOut l;
        l.to = ch_dst(g, c.mine);
        l.offset = ch_offset(g, c.mine);
        l.orient = ch_orient(g, c.mine);
        links[first + c.index] = l;
        if (c.index == 0) {
            const ChRank rh = r[c.head];
            ChainEdgeOut e;
            e.a = ch_src(g, c.head);
            e.b = ch_dst(g, rh.last);
            e.offset = rh.sum;
            e.orient = (ch_orient(g, c.head) & 2u) | (ch_orient(g, rh.last) & 1u); /* mergedEdgeOrientation, SG/EdgeSimple.cpp:272 */
            e.n_links = rh.cnt;
            e.first_link = first;
            comp[comp_id[hs]] = e;
        }
    }
}

#endif /* DISCO_CHAINS_H_ */
