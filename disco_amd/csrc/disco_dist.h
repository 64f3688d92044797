/*
 * disco_dist.h — kernels of the multi-GPU BuildGraph flow (gfx950): routing of items to the rank that owns them, the
 * hash-partitioned index build, on-request exchange of neighbour rows for the transitive reduction, and the push of
 * surviving half-edges for the emission. Host orchestration: disco_hip.hip ("multi-GPU flow"), transport: disco_comm.h.
 *
 * Ownership (both replace RMA/HashTable.cpp:95-116, the range split of hashData, and :1066-1087, needsProcessing):
 *   reads / graph nodes : rank r owns ids [r * per, (r + 1) * per), per = ceil(n / G) rounded up to a multiple of 64
 *   index buckets       : rank r owns the buckets b with (b * G) >> log2(T) == r — a contiguous range of the bucket table
 */
#ifndef DISCO_DIST_H_
#define DISCO_DIST_H_

#include "disco_kernels.h"

#define DIST_MAX_WORLD 64

/* ---- routing: partition a flat list into one contiguous segment per destination rank --------------------------------- */
struct RouteByBucket { /* index records {bucket << 32 | slot, record} */
    int logT;
    u32 G;
    __device__ __forceinline__ u32 operator()(const ulonglong2 &r) const { return (u32)(((r.x >> 32) * (u64)G) >> logT); }
};
/* (otab: ranks own loci — the owner of a node comes out of the table; null: id ranges of `per` nodes) */
struct RouteByRowRequest { /* the node whose row is asked for */
    u64 per;
    const u8 *otab;
    __device__ __forceinline__ u32 operator()(const u32 &r) const { return otab ? (u32)otab[r] : (u32)((u64)r / per); }
};
struct RouteByNode { /* {node id, payload} */
    u64 per;
    const u8 *otab;
    __device__ __forceinline__ u32 operator()(const ulonglong2 &r) const { return otab ? (u32)otab[r.x] : (u32)(r.x / per); }
};

#define ROUTE_ITEMS 8 /* items per thread and tile */
template <typename T, typename F>
__global__ void __launch_bounds__(256) route_count_kernel(const T *__restrict__ items, u64 n, F owner, u32 G, u64 *__restrict__ cnt)
{
    __shared__ u32 s_h[DIST_MAX_WORLD];
    if (threadIdx.x < DIST_MAX_WORLD) s_h[threadIdx.x] = 0;
    __syncthreads();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) atomicAdd(&s_h[owner(items[i])], 1u);
    __syncthreads();
    if (threadIdx.x < G && s_h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], (u64)s_h[threadIdx.x]);
}

/* cursor[g] starts at the first slot of segment g; a tile reserves its share of every segment with one atomic per segment */
template <typename T, typename F>
__global__ void __launch_bounds__(256) route_scatter_kernel(const T *__restrict__ items, u64 n, F owner, u32 G, u64 *__restrict__ cursor, T *__restrict__ out)
{
    __shared__ u32 s_h[DIST_MAX_WORLD];
    __shared__ u64 s_base[DIST_MAX_WORLD];
    const u64 tile = (u64)256 * ROUTE_ITEMS;
    for (u64 t0 = (u64)blockIdx.x * tile; t0 < n; t0 += (u64)gridDim.x * tile) {
        if (threadIdx.x < DIST_MAX_WORLD) s_h[threadIdx.x] = 0;
        __syncthreads();
        T my[ROUTE_ITEMS];
        u32 o[ROUTE_ITEMS], r[ROUTE_ITEMS];
#pragma unroll
        for (int q = 0; q < ROUTE_ITEMS; q++) {
            const u64 i = t0 + (u64)q * 256 + threadIdx.x;
            o[q] = 0xFFFFFFFFu;
            if (i < n) {
                my[q] = items[i];
                o[q] = owner(my[q]);
                r[q] = atomicAdd(&s_h[o[q]], 1u);
            }
        }
        __syncthreads();
        if (threadIdx.x < G) s_base[threadIdx.x] = s_h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], (u64)s_h[threadIdx.x]) : 0ull;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < ROUTE_ITEMS; q++)
            if (o[q] != 0xFFFFFFFFu) out[s_base[o[q]] + r[q]] = my[q];
        __syncthreads();
    }
}

/* first slot of every segment from the counts (one wavefront; G <= 64): the routing needs no host round trip between its two kernels */
__global__ void route_cursor_kernel(const u64 *__restrict__ cnt, u32 G, u64 *__restrict__ cursor)
{
    const u32 l = threadIdx.x;
    u64 x = l < G ? cnt[l] : 0ull, incl = x;
    for (int o = 1; o < 64; o <<= 1) {
        const u64 y = ((u64)(u32)__shfl_up((int)(u32)(incl >> 32), o) << 32) | (u32)__shfl_up((int)(u32)incl, o);
        if ((int)l >= o) incl += y;
    }
    if (l < G) cursor[l] = incl - x;
}

/* ---- reads: rows travel at the words they use, not at the 64-byte stride of the table ---------------------------------- */
/* rows [r0, r0 + nrows) of the S-stride table <-> a dense array of W words per row (W = words of the longest read of the job) */
__global__ void pack_rows_kernel(const u64 *__restrict__ table, int S, int W, u64 r0, u64 nrows, u64 *__restrict__ dense)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = nrows * (u64)W;
    for (; i < total; i += (u64)gridDim.x * blockDim.x) dense[i] = table[(r0 + i / W) * S + i % W];
}

/* all rows but [skip_lo, skip_hi) (the rank's own, already in place); the words beyond W stay zero (the table was cleared once) */
/* (per / block_words: rank p's rows start at word p * block_words of `dense` — its block also carries its lengths behind the rows;
 * block_words = per * W: the rows of all ranks back to back) */
__global__ void unpack_rows_kernel(const u64 *__restrict__ dense, int S, int W, u64 nrows, u64 skip_lo, u64 skip_hi, u64 *__restrict__ table, u64 per, u64 block_words)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = nrows * (u64)W;
    for (; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 r = i / W;
        if (r >= skip_lo && r < skip_hi) continue;
        table[r * S + i % W] = dense[(r / per) * block_words + (r % per) * (u64)W + i % W];
    }
}
/* the lengths behind the rows of every rank's block -> len[] (all ranks but `skip`) */
__global__ void unpack_lens_kernel(const u64 *__restrict__ dense, u64 per, u64 block_words, u64 rows_words, u32 G, u32 skip, u16 *__restrict__ len)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = per * (u64)G;
    for (; i < total; i += (u64)gridDim.x * blockDim.x) {
        const u64 p = i / per;
        if (p == skip) continue;
        len[i] = ((const u16 *)(dense + p * block_words + rows_words))[i % per];
    }
}

__global__ void add_u64_kernel(u64 *__restrict__ p, u64 n, u64 val)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) p[i] += val;
}

/* the long reads of [lo, hi) — more than DISCO_SHORT_MAX bases — and the longest of the others: what decides whether the job gets two classes
 * of rows (every rank reduces these over the ranks and decides alike) */
__global__ void long_stats_kernel(const u16 *__restrict__ len, u64 lo, u64 hi, u64 *__restrict__ out)
{
    u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 n_long = 0, smax = 0;
    for (; i < hi; i += (u64)gridDim.x * blockDim.x) {
        const u32 L = len[i];
        if (L > (u32)DISCO_SHORT_MAX) n_long++;
        else smax = L > smax ? L : smax;
    }
    for (int o = 32; o > 0; o >>= 1) {
        n_long += (u32)__shfl_down((int)n_long, o);
        smax = max(smax, (u32)__shfl_down((int)smax, o));
    }
    if ((threadIdx.x & 63) == 0) {
        if (n_long) atomicAdd(&out[0], (u64)n_long);
        atomicMax(&out[1], (u64)smax);
    }
}

/* ---- ranks own loci: reads dealt to the ranks by their read-level minimizer ------------------------------------------------- */
/* the read-level minimizer keys of the reads [lo, hi) alone (okey[i] as index_count_kernel / index_runs_kernel compute it: the smallest
 * 32-bit order hash among all m-mers of the read). The reads are dealt by this key before anything else is computed from them: the
 * records, runs and rows of a read are the business of the rank that gets it. One thread per read, one rolling pass. */
template <bool FIXED_M>
__global__ void __launch_bounds__(256) read_keys_kernel(DiscoView v, u64 lo, u64 hi, u32 *__restrict__ okey)
{
    const u64 i = lo + (u64)blockIdx.x * 256u + threadIdx.x;
    if (i >= hi) return;
    const u64 *__restrict__ p = v.reads + i * (u64)v.S;
    const int L = v.len[i];
    u32 best = 0xFFFFFFFFu;
    if (FIXED_M) { /* m = RUNS_M (every k from 23 to 86): index_runs_kernel's rolling pass on 32-bit halves */
        constexpr int m = RUNS_M;
        const int nmm = L - m + 1;
        MmerRoll<m> roll(p, v.S);
        for (int q = 0; q < m - 1; ++q) roll.step();
        int nmax = nmm; /* every thread of the wavefront walks the same positions (the longest read's); a shorter read masks what it keeps */
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
        nmax = (int)uniform_u32((u32)nmax);
        for (int q = 0; q < nmax; ++q) {
            roll.step();
            const u32 h = roll.hash();
            best = min(best, q < nmm ? h : 0xFFFFFFFFu);
        }
    } else if (v.m > 16) { /* any other length of two dwords: the same pass with the length as a run-time value */
        const int m = v.m, nmm = L - m + 1;
        MmerRoll<0> roll(p, v.S, m);
        for (int q = 0; q < m - 1; ++q) roll.step();
        int nmax = nmm;
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
        nmax = (int)uniform_u32((u32)nmax);
        for (int q = 0; q < nmax; ++q) {
            roll.step();
            const u32 h = roll.hash();
            best = min(best, q < nmm ? h : 0xFFFFFFFFu);
        }
    } else {
        const int m = v.m, nmm = L - m + 1;
        const u64 mask = (1ull << (2 * m)) - 1ull;
        const int rsh = 2 * (m - 1);
        u64 f = 0, r = 0, word = 0;
        int pos = 0;
        auto next = [&]() {
            if ((pos & 31) == 0) word = p[pos >> 5];
            const u32 b = (u32)(word >> 62);
            word <<= 2;
            ++pos;
            f = ((f << 2) | b) & mask;
            r = (r >> 2) | ((u64)(3u - b) << rsh);
        };
        for (int q = 0; q < m - 1; ++q) next();
        for (int q = 0; q < nmm; ++q) {
            next();
            best = min(best, order_hash32(r < f ? r : f));
        }
    }
    okey[i] = best;
}

/* the rows of the own reads ahead of the all-gather of all reads (which only verify waits for): every rank sends the reads of its home
 * range to the ranks that got them — {id | length << 32, the W words the job's longest read uses} — one all-to-all-v of about
 * n / G x (W + 1) x 8 bytes per rank; the receiver puts them where the all-gather will put them again */
template <int W>
struct ReadItem {
    u64 hdr;
    u64 w[W];
};
template <int W>
struct RouteByReadItem {
    const u8 *otab;
    __device__ __forceinline__ u32 operator()(const ReadItem<W> &r) const { return (u32)otab[(u32)r.hdr]; }
};
template <int W>
__global__ void read_items_kernel(const u64 *__restrict__ reads, const u16 *__restrict__ len, int S, u64 lo, u64 hi, ReadItem<W> *__restrict__ out)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = (hi - lo) * (u64)(W + 1);
    u64 *o = (u64 *)out;
    for (; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 i = lo + t / (W + 1);
        const u32 w = (u32)(t % (W + 1));
        o[t] = w == 0 ? (i | ((u64)len[i] << 32)) : reads[i * (u64)S + (w - 1)];
    }
}
template <int W>
__global__ void read_items_place_kernel(const ReadItem<W> *__restrict__ items, u64 n_items, int S, u64 *__restrict__ reads, u16 *__restrict__ len)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = n_items * (u64)(W + 1);
    const u64 *in = (const u64 *)items;
    for (; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 x = t / (W + 1);
        const u32 w = (u32)(t % (W + 1));
        const u64 hdr = in[x * (W + 1)];
        const u64 id = hdr & 0xFFFFFFFFull;
        if (w == 0) len[id] = (u16)(hdr >> 32);
        else reads[id * (u64)S + (w - 1)] = in[t];
    }
}

/* bucket of a key in the grouping of ONE rank's reads: the rank's share of the hash range (disco_key_owner: [me, me + 1) * 2^32 / G)
 * stretched over all 32 bits again, so that the rank's groups spread over all its buckets as the groups of a whole job do */
__device__ __forceinline__ u32 order_bucket_local(u32 key, u32 G, u32 me, u32 shift)
{
    const u32 h = key * 0x9E3779B1u;
    const u64 lo = (((u64)me << 32) + G - 1) / G; /* smallest h with (h * G) >> 32 == me */
    const u64 x = ((u64)h - lo) * (u64)G;
    return (u32)(x > 0xFFFFFFFFull ? 0xFFFFFFFFull : x) >> shift;
}
/* owner of every read from its key (disco_key_owner) and the list of the own reads' ids: in id order inside a tile of OWN_TILE reads,
 * the tiles in the order their blocks arrive (one counting atomic per tile; the grouping that follows re-orders the list anyway) */
/* ocnt != null: the counting pass of the grouping rides along (order_count_list_kernel's: the own read's slot in its bucket, by its
 * position in the list) — the key is in a register here, and the list need not be walked once more */
#define OWN_TILE 4096
__global__ void __launch_bounds__(256) own_select_kernel(const u32 *__restrict__ okey, u64 n, u32 G, u32 me, u8 *__restrict__ otab, u32 *__restrict__ own_ids,
                                                         u64 *__restrict__ n_own, u32 *__restrict__ ocnt, u32 *__restrict__ oslot, u32 oshift)
{
    __shared__ u32 s_w[4];
    __shared__ u64 s_base;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    for (u64 t0 = (u64)blockIdx.x * OWN_TILE; t0 < n; t0 += (u64)gridDim.x * OWN_TILE) {
        u32 mask = 0, cnt = 0;
        u32 slot[OWN_TILE / 256];
#pragma unroll
        for (int q = 0; q < OWN_TILE / 256; q++) {
            const u64 i = t0 + (u64)q * 256u + tid;
            slot[q] = 0;
            if (i < n) {
                const u32 key = okey[i];
                const u32 o = disco_key_owner(key, G);
                otab[i] = (u8)o;
                if (o == me) {
                    mask |= 1u << q;
                    cnt++;
                    if (ocnt) slot[q] = atomicAdd(&ocnt[order_bucket_local(key, G, me, oshift)], 1u);
                }
            }
        }
        const u32 incl = wave_inclusive_add(cnt);
        __syncthreads();
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        u32 off = incl - cnt;
        for (u32 w = 0; w < wv; w++) off += s_w[w];
        const u32 total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (tid == 0 && total) s_base = atomicAdd((unsigned long long *)n_own, (unsigned long long)total);
        __syncthreads();
        u64 at = s_base + off;
        /* (thread t holds reads t, t + 256, ...: inside the tile the list is in id order only per thread; nobody relies on it) */
#pragma unroll
        for (int q = 0; q < OWN_TILE / 256; q++)
            if (mask & (1u << q)) {
                if (ocnt) oslot[at] = slot[q];
                own_ids[at++] = (u32)(t0 + (u64)q * 256u + tid);
            }
    }
}

/* the grouping (order_count_kernel / order_scatter_kernel) over a list of read ids */
__global__ void order_count_list_kernel(const u32 *__restrict__ okey, const u32 *__restrict__ ids, u64 nq, u32 G, u32 me, u32 shift, u32 *__restrict__ cnt, u32 *__restrict__ oslot)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; t < nq; t += (u64)gridDim.x * blockDim.x) oslot[t] = atomicAdd(&cnt[order_bucket_local(okey[ids[t]], G, me, shift)], 1u);
}
__global__ void order_scatter_list_kernel(const u32 *__restrict__ okey, const u32 *__restrict__ ids, const u32 *__restrict__ oslot, const u32 *__restrict__ start, u64 nq,
                                          u32 G, u32 me, u32 shift, const u16 *__restrict__ len, u64 *__restrict__ order)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; t < nq; t += (u64)gridDim.x * blockDim.x) {
        const u32 id = ids[t];
        order[(u64)start[order_bucket_local(okey[id], G, me, shift)] + oslot[t]] = ORDER_MAKE(id, len[id]);
    }
}
/* ... or, where no grouping is wanted (a handful of reads, DISCO_NO_ORDER), the list as it is */
__global__ void order_pack_list_kernel(const u32 *__restrict__ ids, u64 nq, const u16 *__restrict__ len, u64 *__restrict__ order)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; t < nq; t += (u64)gridDim.x * blockDim.x) order[t] = ORDER_MAKE(ids[t], len[ids[t]]);
}

/* a bitmap by read id <-> the list of its set bits (regime 2 with ranks that own loci: the reads that dropped a hit are scattered
 * over the id space, so their bitmap travels as lists; one counting atomic per 64 words that hold a bit) */
__global__ void bits_to_list_kernel(const u64 *__restrict__ bits, u64 n_words, u32 *__restrict__ list, u64 *__restrict__ n_list, u64 cap, u64 *ctr)
{
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u32 lane = threadIdx.x & 63u;
    for (u64 w0 = w - lane; w0 < n_words; w0 += (u64)gridDim.x * blockDim.x) {
        const u64 wi = w0 + lane;
        u64 x = wi < n_words ? bits[wi] : 0ull;
        const u32 c = (u32)__popcll(x);
        const u32 incl = wave_inclusive_add(c);
        const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        if (!tot) continue;
        u64 base = 0;
        if (lane == 0) base = atomicAdd((unsigned long long *)n_list, (unsigned long long)tot);
        base = readlane_u64(base, 0) + (incl - c);
        if (!list) continue; /* (counting pass) */
        while (x) {
            const u32 b = (u32)__ffsll((long long)x) - 1u;
            x &= x - 1;
            if (base < cap) list[base] = (u32)(wi * 64 + b);
            else atomicAdd((unsigned long long *)&ctr[CTR_OVERFLOW], 1ull);
            base++;
        }
    }
}
__global__ void list_to_bits_kernel(const u32 *__restrict__ list, u64 n, u64 *__restrict__ bits)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) atomicOr((unsigned long long *)&bits[list[i] >> 6], 1ull << (list[i] & 63u));
}

/* ---- containment across ranks: who IS contained is a bitmap's business, the key only matters to the row that is written at the end --- */
/* bit i: read i has a containment key on this rank (n a multiple of 64: the table's padded size; best = DISCO_NOKEY beyond the reads) */
__global__ void has_key_bits_kernel(const u64 *__restrict__ best, u64 n, u64 *__restrict__ bits)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 mk = __ballot(best[i] != DISCO_NOKEY);
        if ((threadIdx.x & 63) == 0) bits[i >> 6] = mk;
    }
}
/* OR of the ranks' bitmaps (rank p's at all + p * words) -> cbits of every read; byte flags and their count for the words [w0, w1)
 * (the home range: what disco_fetch_contained walks) */
__global__ void or_bits_kernel(const u64 *__restrict__ all, u32 G, u64 words, u64 *__restrict__ cbits, u64 w0, u64 w1, u8 *__restrict__ contained, u64 *ctr)
{
    u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 cnt = 0;
    for (; w < words; w += (u64)gridDim.x * blockDim.x) {
        u64 x = 0;
        for (u32 p = 0; p < G; p++) x |= all[(u64)p * words + w];
        cbits[w] = x;
        if (w >= w0 && w < w1 && x) {
            cnt += (u32)__popcll(x);
            for (u64 y = x; y; y &= y - 1) contained[w * 64 + (u64)(__ffsll((long long)y) - 1)] = 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd((unsigned long long *)&ctr[CTR_N_CONTAINED], (unsigned long long)cnt);
}

/* ---- hash-partitioned index build (the owner's side) ------------------------------------------------------------------ */
/* records received from all ranks: count per bucket of this rank's range; the atomic hands every record its slot */
__global__ void shard_count_kernel(ulonglong2 *__restrict__ rec, u64 n, u32 *__restrict__ bkt)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u64 b = rec[i].x >> 32;
        rec[i].x = (b << 32) | atomicAdd(&bkt[b], 1u);
    }
}

__global__ void add_u32_kernel(u32 *__restrict__ p, u64 n, u32 val)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) p[i] += val;
}

/* ---- neighbour rows on request (transitive reduction) -------------------------------------------------------------- */
/* Round 6: a fetched row is the WHOLE row of the other rank's node, kept as 8-byte entries behind the rank's own rows in the same array,
 * under the node's ONE reference word ref[u] (0 = not fetched; asked[]: one bit per node, set by the first request for it) — the marking
 * kernel reads it as it reads an own row (TrArgs). Rounds 2-5 fetched the entries of one class (u, cls) as 4-byte entries into a store of their own behind two
 * reference words per node: the marking variant that read both kinds of rows cost 5.5 ms more than the plain kernel at 50 M reads. */

/* append the lanes' requests (0xFFFFFFFF = none) to the list: one atomic per wavefront */
__device__ __forceinline__ void request_append(u32 rq, u32 *__restrict__ list, u64 *__restrict__ n_list, u64 cap, u64 *ctr)
{
    const bool has = rq != 0xFFFFFFFFu;
    const u64 mk = __ballot(has);
    if (!mk) return;
    u64 base = 0;
    const u32 leader = (u32)__ffsll((long long)mk) - 1u;
    if ((threadIdx.x & 63) == leader) base = atomicAdd(n_list, (u64)__popcll(mk));
    base = readlane_u64(base, leader);
    if (has) {
        const u64 pos = base + rank_below(mk);
        if (pos < cap) list[pos] = rq;
        else atomicAdd(&ctr[CTR_OVERFLOW], 1ull);
    }
}

/* the node whose row a sweep from this entry needs, if it is another rank's and nobody on this rank has asked for it yet: a test-and-set
 * on a bitmap by node (n / 8 bytes: it lives in the L2; rounds 2-4 marked an 8-byte word of a table of 16 n bytes, and round 5 asked without
 * looking). ROUND2: rows that round 1 brought are there (ref[u] != 0) */
template <bool ROUND2>
__device__ __forceinline__ u32 request_for(u64 e, const OwnSet &own, const u64 *__restrict__ ref, u32 *__restrict__ asked)
{
    const u64 u = ADJ_DST(e);
    if (own.mine(u)) return 0xFFFFFFFFu;
    if (ROUND2 && ref[u] != TR_UNAVAIL) return 0xFFFFFFFFu;
    const u32 bit = 1u << (u & 31u);
    if (atomicOr(&asked[u >> 5], bit) & bit) return 0xFFFFFFFFu;
    return (u32)u;
}

/* round 1: every register-resident node (degree <= 64) asks for the two rows its marking sweeps for certain — slot 0 and the
 * first slot on the other side of the node (exactly transitive_mark_kernel's speculative pair). One lane per node; the row's first eight
 * entries — a 64-byte line or two — are loaded at once (the first entry on the other side is among the first few; a loop that walks the row
 * entry by entry is a chain of dependent loads per lane: 2.3 ms per rank at 8 x 6.25 M nodes), the rare row without one among them is
 * walked */
__global__ void __launch_bounds__(64) tr_request_first_kernel(const u64 *__restrict__ adj, OwnSet own, const u64 *__restrict__ ref, u32 *__restrict__ asked,
                                                               u32 *__restrict__ list, u64 *__restrict__ n_list, u64 cap, u64 *ctr)
{
    const u64 nloc = own.count();
    const u32 lane = threadIdx.x;
    /* the requests are collected over many blocks of 64 nodes and appended 2048 at a time: the list's counter is ONE address, and an atomic
     * per block of 64 nodes (195 000 per rank) is 2.3 ms per rank */
    __shared__ u32 s_out[2048 + 128];
    u32 n_out = 0; /* wave uniform */
    auto flush = [&]() {
        if (n_out == 0) return;
        u64 base = 0;
        if (lane == 0) base = atomicAdd(n_list, (u64)n_out);
        base = readlane_u64(base, 0);
        for (u32 x = lane; x < n_out; x += 64) {
            if (base + x < cap) list[base + x] = s_out[x];
            else atomicAdd(&ctr[CTR_OVERFLOW], 1ull);
        }
        n_out = 0;
    };
    auto stash = [&](u32 rq) {
        const bool has = rq != 0xFFFFFFFFu;
        const u64 mk = __ballot(has);
        if (has) s_out[n_out + rank_below(mk)] = rq;
        n_out += (u32)__popcll(mk);
    };
    for (u64 blk = (u64)blockIdx.x * 64; blk < nloc; blk += (u64)gridDim.x * 64) { /* (wave uniform) */
        const u64 i = blk + lane;
        u32 rq0 = 0xFFFFFFFFu, rq2 = 0xFFFFFFFFu;
        const u64 rv = i < nloc ? ref[own.node(i)] : 0ull;
        const u32 d = REF_DEG(rv);
        if (d != 0 && d <= 64) {
            const u64 *row = adj + REF_POS(rv);
            u64 e[8];
#pragma unroll
            for (u32 s = 0; s < 8; s++) e[s] = row[s < d ? s : d - 1];
            const u32 side0 = ADJ_ORI(e[0]) >> 1;
            u64 e2 = 0;
            bool has2 = false;
#pragma unroll
            for (int s = 7; s >= 1; s--) /* (descending: the first one wins) */
                if ((u32)s < d && (ADJ_ORI(e[s]) >> 1) != side0) {
                    e2 = e[s];
                    has2 = true;
                }
            if (!has2)
                for (u32 s = 8; s < d; s++) {
                    const u64 x = row[s];
                    if ((ADJ_ORI(x) >> 1) != side0) {
                        e2 = x;
                        has2 = true;
                        break;
                    }
                }
            rq0 = request_for<false>(e[0], own, ref, asked);
            if (has2) rq2 = request_for<false>(e2, own, ref, asked);
        }
        stash(rq0);
        stash(rq2);
        if (n_out > 2048) {
            __syncthreads();
            flush();
            __syncthreads();
        }
    }
    __syncthreads();
    flush();
}

/* sum of the degrees of the listed nodes (upper bound of what the request-all round can ask for) */
__global__ void list_degree_sum_kernel(const u64 *__restrict__ list, u64 n_list, const u64 *__restrict__ ref, u64 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0;
    for (; i < n_list; i += (u64)gridDim.x * blockDim.x) s += REF_DEG(ref[list[i]]);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

__global__ void list_max_degree_kernel(const u64 *__restrict__ list, u64 n_list, const u64 *__restrict__ ref, u64 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 m = 0;
    for (; i < n_list; i += (u64)gridDim.x * blockDim.x) m = max(m, REF_DEG(ref[list[i]]));
    for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_down(m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, (u64)m);
}

/* sum of len[i] - k over the reads [lo, hi): the k-mer probes of the range (disco_counters.probes) */
__global__ void probes_sum_kernel(const u16 *__restrict__ len, u64 lo, u64 hi, u32 k, u64 *__restrict__ out)
{
    u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0;
    for (; i < hi; i += (u64)gridDim.x * blockDim.x) s += (u64)len[i] - k;
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

/* round 2: the listed nodes (deferred by round 1, or beyond the register path) ask for every row they do not have */
__global__ void __launch_bounds__(64) tr_request_all_kernel(const u64 *__restrict__ nodes, u64 n_nodes, const u64 *__restrict__ adj, OwnSet own, const u64 *__restrict__ ref,
                                                            u32 *__restrict__ asked, u32 *__restrict__ list, u64 *__restrict__ n_list, u64 cap, u64 *ctr)
{
    for (u64 it = blockIdx.x; it < n_nodes; it += gridDim.x) {
        const u64 rv = ref[nodes[it]];
        const u32 d = REF_DEG(rv);
        const u64 *row = adj + REF_POS(rv);
        for (u32 s0 = 0; s0 < d; s0 += 64) {
            const u32 s = s0 + threadIdx.x;
            u32 rq = 0xFFFFFFFFu;
            if (s < d) rq = request_for<true>(row[s], own, ref, asked);
            request_append(rq, list, n_list, cap, ctr);
        }
    }
}

/* the owner's side: the degree of every requested row ... */
__global__ void tr_respond_deg_kernel(const u32 *__restrict__ req, u64 n_req, const u64 *__restrict__ ref, u32 *__restrict__ deg)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_req; i += (u64)gridDim.x * blockDim.x) deg[i] = REF_DEG(ref[req[i]]);
}
/* ... and its entries as 4 bytes each (destination and orientation: all a sweep reads of a neighbour's row, BG/OverlapGraph.cpp:698-708)
 * at out + pos[i]; four requests per wavefront, sixteen lanes each */
__global__ void __launch_bounds__(64) tr_respond_kernel(const u32 *__restrict__ req, u64 n_req, const u64 *__restrict__ ref, const u64 *__restrict__ adj,
                                                        const u64 *__restrict__ pos, u32 *__restrict__ out)
{
    const u32 lane = threadIdx.x, sub = lane & 15u;
    for (u64 i0 = (u64)blockIdx.x * 4; i0 < n_req; i0 += (u64)gridDim.x * 4) {
        const u64 i = i0 + (lane >> 4);
        if (i >= n_req) continue;
        const u64 rv = ref[req[i]];
        const u32 d = REF_DEG(rv);
        const u64 *row = adj + REF_POS(rv);
        u32 *o = out + pos[i];
        for (u32 s = sub; s < d; s += 16) o[s] = NBR32_MAKE(row[s]);
    }
}

/* the requester's side: the rows that came back (pos = exclusive scan of the received degrees) go behind the own rows as 8-byte entries
 * (offset and length fields 0: never read of a neighbour's row), the nodes' reference words address them */
__global__ void __launch_bounds__(64) rows_place_kernel(const u32 *__restrict__ req, u64 n_req, const u32 *__restrict__ deg, const u64 *__restrict__ pos,
                                                        const u32 *__restrict__ in, u64 base, u64 *__restrict__ adj, u64 *__restrict__ ref)
{
    const u32 lane = threadIdx.x, sub = lane & 15u;
    for (u64 i0 = (u64)blockIdx.x * 4; i0 < n_req; i0 += (u64)gridDim.x * 4) {
        const u64 i = i0 + (lane >> 4);
        if (i >= n_req) continue;
        const u32 d = deg[i];
        const u64 p = pos[i];
        for (u32 s = sub; s < d; s += 16) adj[base + p + s] = NBR32_ENTRY(in[p + s]);
        if (sub == 0 && d) ref[req[i]] = REF_MAKE(base + p, d); /* (an empty row cannot be a neighbour's: the marking fails loudly) */
    }
}

/* ---- twin completion across ranks (some read dropped a verified hit: real data at their repeats) --------------------------- */
/* insertEdge puts the twin of every find into the other read's list (BG/OverlapGraph.cpp:614-626). A list can lack a twin only if
 * its read dropped a hit (twin_check in disco_hip.hip); the drop bitmap is all-gathered, every rank sends {w, twin} for its finds
 * u -> w into such reads w of OTHER ranks, the owner looks the twin up and appends what is missing to its extras. Wave per own
 * node; FILL = false counts the items. */
template <bool FILL>
__global__ void __launch_bounds__(256) twin_push_kernel(const u64 *__restrict__ ref, const u64 *__restrict__ adj, const u16 *__restrict__ len, OwnSet own,
                                                        const u64 *__restrict__ dropbits, ulonglong2 *__restrict__ list, u64 *__restrict__ n_list, u64 cap,
                                                        u64 *ctr)
{
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    u64 mine = 0;
    const u64 nloc = own.count();
    for (u64 ui = wave; ui < nloc; ui += nwaves) {
        const u64 u = own.node(ui);
        const u64 ru = ref[u];
        const u32 du = REF_DEG(ru), Lu = len[u];
        for (u32 p0 = 0; p0 < du; p0 += 64) {
            const u32 p = p0 + lane;
            const u64 e = p < du ? adj[REF_POS(ru) + p] & ~ADJ_FLAG : 0ull;
            const u64 w = ADJ_DST(e);
            const bool take = p < du && !own.mine(w) && ((dropbits[w >> 6] >> (w & 63)) & 1ull);
            const u64 mk = __ballot(take);
            if (!mk) continue;
            if (!FILL) {
                mine += __popcll(mk);
                continue;
            }
            const u32 leader = (u32)__ffsll((long long)mk) - 1u;
            u64 base = 0;
            if (lane == leader) base = atomicAdd(n_list, (u64)__popcll(mk));
            base = readlane_u64(base, leader);
            if (take) {
                const u64 q = base + rank_below(mk);
                if (q < cap) list[q] = make_ulonglong2(w, ADJ_MAKE(ADJ_DLEN(e) + ADJ_OFF(e) - Lu, u, disco_twin_orient(ADJ_ORI(e)), Lu));
                else atomicAdd(&ctr[CTR_OVERFLOW], 1ull);
            }
        }
    }
    if (!FILL && lane == 0 && mine) atomicAdd(n_list, mine);
}

/* the owner's side: item {w, twin}; the twin is missing from w's list <=> the pair was found from the other side only */
__global__ void twin_recv_kernel(const ulonglong2 *__restrict__ items, u64 n_items, const u64 *__restrict__ ref, const u64 *__restrict__ adj,
                                 u64 *__restrict__ extra_node, u64 *__restrict__ extra_key, u32 *__restrict__ extra_cnt, u32 *__restrict__ n_extra,
                                 u32 extra_cap, u64 *ctr)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_items; i += (u64)gridDim.x * blockDim.x) {
        const u64 w = items[i].x, twin = items[i].y;
        const u64 rw = ref[w];
        if (adj_find(adj + REF_POS(rw), REF_DEG(rw), twin) >= 0) continue;
        atomicAdd(&ctr[CTR_ASYM], 1ull);
        const u32 idx = atomicAdd(n_extra, 1u);
        if (idx < extra_cap) {
            extra_node[idx] = EXTRA_NODE_MAKE(w, atomicAdd(&extra_cnt[w], 1u));
            extra_key[idx] = twin;
        } else
            atomicAdd(&ctr[CTR_OVERFLOW], 1ull);
    }
}

/* ---- emission: surviving half-edges pushed to the owner of the smaller endpoint -------------------------------------- */
/* An edge (a, b), a < b, is emitted by the owner of a, and survives iff it is unflagged from both ends (BG/OverlapGraph.cpp:
 * 717-718). The owner of b pushes {a, twin} for each of b's unflagged entries (b -> a) with a on a lower rank; twin = the entry
 * (a -> b) as it stands in a's own list, so the receiver only has to look it up among a's survivors. Lane = own node b.
 * FILL = false counts the items. */
template <bool FILL>
__global__ void __launch_bounds__(64) emit_push_kernel(const u64 *__restrict__ ref, const u64 *__restrict__ adj, const u64 *__restrict__ half,
                                                        const u32 *__restrict__ hcnt, const u16 *__restrict__ len, OwnSet own,
                                                        ulonglong2 *__restrict__ list, u64 *__restrict__ n_list, u64 cap, u64 *ctr)
{
    const u64 nloc = own.count();
    const u64 n64 = (nloc + 63) & ~63ull;
    const u32 lane = threadIdx.x & 63u;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 mine = 0;
    /* the items of a wavefront's trips (up to HALF_CAP per node) are collected in LDS and appended 512 at a time (round 4: one atomic
     * per call was 390 000 atomics on ONE address per rank — 2.1 of this kernel's 2.3 ms). One wavefront per workgroup. */
    __shared__ ulonglong2 s_items[512 + 64 * HALF_CAP];
    u32 s_n = 0; /* wave uniform */
    auto flush_items = [&]() {
        if (!FILL || s_n == 0) return;
        u64 base = 0;
        if (lane == 0) base = atomicAdd(n_list, (u64)s_n);
        base = readlane_u64(base, 0);
        for (u32 x = lane; x < s_n; x += 64) {
            if (base + x < cap) list[base + x] = s_items[x];
            else atomicAdd(&ctr[CTR_OVERFLOW], 1ull);
        }
        s_n = 0;
    };
    auto item_of = [&](u64 e, u64 b, u32 Lb) {
        return make_ulonglong2(ADJ_DST(e), ADJ_MAKE(ADJ_DLEN(e) + ADJ_OFF(e) - Lb, b, disco_twin_orient(ADJ_ORI(e)), Lb));
    };
    auto push = [&](bool have, u64 e, u64 b, u32 Lb) { /* one candidate per lane and call (the rows of nodes with many survivors) */
        const bool take = have && ADJ_DST(e) < b && !own.mine(ADJ_DST(e)); /* else: a > b, or a on this rank (judged locally) */
        const u64 mk = __ballot(take);
        if (!mk) return;
        if (!FILL) {
            if (lane == 0) mine += __popcll(mk);
            return;
        }
        const u32 leader = (u32)__ffsll((long long)mk) - 1u;
        u64 base = 0;
        if (lane == leader) base = atomicAdd(n_list, (u64)__popcll(mk));
        base = readlane_u64(base, leader);
        if (take) {
            const u64 p = base + rank_below(mk);
            if (p < cap) list[p] = item_of(e, b, Lb);
            else atomicAdd(&ctr[CTR_OVERFLOW], 1ull);
        }
    };
    for (; i < n64; i += (u64)gridDim.x * blockDim.x) {
        const bool live = i < nloc;
        const u64 b = own.node_by_id(live ? i : 0);
        const u32 cnt = live ? hcnt[b] : 0u, Lb = live ? (u32)len[b] : 0u;
        const bool narrow = cnt <= HALF_CAP;
        u64 he[HALF_CAP];
        u32 takem = 0; /* bit r: survivor r of this lane's node is pushed */
#pragma unroll
        for (u32 r = 0; r < HALF_CAP; r++) {
            const bool have = narrow && r < cnt;
            he[r] = have ? half[b * HALF_CAP + r] : 0ull;
            if (have && ADJ_DST(he[r]) < b && !own.mine(ADJ_DST(he[r]))) takem |= 1u << r;
        }
        {
            const u32 mycnt = (u32)__popc(takem);
            const u32 incl = wave_inclusive_add(mycnt);
            const u32 tot = (u32)__builtin_amdgcn_readlane((int)incl, 63);
            if (tot) {
                if (!FILL) {
                    if (lane == 0) mine += tot;
                } else { /* into the wavefront's LDS buffer, flushed 512 items at a time (one atomic on the list's counter per flush) */
                    u32 p = s_n + (incl - mycnt);
#pragma unroll
                    for (u32 r = 0; r < HALF_CAP; r++)
                        if (takem & (1u << r)) s_items[p++] = item_of(he[r], b, Lb);
                    s_n += tot;
                    __syncthreads();
                    if (s_n > 512u) {
                        flush_items();
                        __syncthreads();
                    }
                }
            }
        }
        /* many survivors (rare): the row carries the flags; the wavefront walks the rows of its wide lanes one after the other */
        u64 wide = __ballot(!narrow);
        while (wide) {
            const u32 l = (u32)__ffsll((long long)wide) - 1u;
            wide &= wide - 1;
            const u64 bw = readlane_u64(b, l);
            const u32 Lw = (u32)__builtin_amdgcn_readlane((int)Lb, (int)l);
            const u64 rv = ref[bw];
            const u64 *row = adj + REF_POS(rv);
            const u32 d = REF_DEG(rv);
            for (u32 s0 = 0; s0 < d; s0 += 64) {
                const u32 sl = s0 + lane;
                const u64 e = sl < d ? row[sl] : ADJ_FLAG;
                push(!(e & ADJ_FLAG), e & ~ADJ_FLAG, bw, Lw);
            }
        }
    }
    __syncthreads();
    flush_items();
    if (!FILL && lane == 0 && mine) atomicAdd(n_list, mine);
}

/* the receiver: item {a, twin} -> the edge (a, twin) is emitted iff twin is among the survivors of a. Output through the same
 * wave-private chunks as emit_half_kernel. */
struct EmitRecvArgs {
    const ulonglong2 *items;
    u64 n_items;
    const u64 *ref;
    const u64 *adj;
    const u64 *half;
    const u32 *hcnt;
    u64 *out_src;
    u64 *out_ent;
    u64 out_cap;
    u64 *bump;
    u64 *wq;
};

__global__ void __launch_bounds__(64) emit_push_recv_kernel(EmitRecvArgs a)
{
    const u32 lane = threadIdx.x;
    u64 chunk_base = 0;
    u32 chunk_used = EMIT_CHUNK;
    bool have_chunk = false;
    auto close_chunk = [&]() {
        if (have_chunk)
            for (u32 i = chunk_used + lane; i < EMIT_CHUNK; i += 64)
                if (chunk_base + i < a.out_cap) a.out_src[chunk_base + i] = ~0ull;
    };
    u64 cbeg = 0, cend = 0;
    while (wq_grab(a.wq, (a.n_items + 63) / 64, cbeg, cend))
        for (u64 blk = cbeg; blk < cend; blk++) {
            const u64 i = blk * 64 + lane;
            bool keep = false;
            u64 v = 0, twin = 0;
            if (i < a.n_items) {
                const ulonglong2 it = a.items[i];
                v = it.x;
                twin = it.y;
                const u32 cw = a.hcnt[v];
                if (cw <= HALF_CAP) {
                    const u64 *hw = a.half + v * HALF_CAP;
                    for (u32 r = 0; r < cw; r++) keep |= (hw[r] == twin);
                } else {
                    const u64 rw = a.ref[v];
                    const u64 *roww = a.adj + REF_POS(rw);
                    const int ti = adj_find(roww, REF_DEG(rw), twin);
                    keep = (ti >= 0) && !(roww[ti] & ADJ_FLAG);
                }
            }
            const u64 mk = __ballot(keep);
            const u32 kc = __popcll(mk);
            if (kc) {
                if (chunk_used + kc > EMIT_CHUNK) {
                    close_chunk();
                    u64 base = 0;
                    if (lane == 0) base = atomicAdd(a.bump, (u64)EMIT_CHUNK);
                    chunk_base = __shfl(base, 0);
                    chunk_used = 0;
                    have_chunk = true;
                }
                if (keep) {
                    const u64 pos = chunk_base + chunk_used + rank_below(mk);
                    if (pos < a.out_cap) {
                        a.out_src[pos] = v;
                        a.out_ent[pos] = twin;
                    }
                }
                chunk_used += kc;
            }
        }
    close_chunk();
}

/* ==== the index that STAYS partitioned (disco_dist_run_graph with DISCO_DIST_KEEP_INDEX_PARTITIONED) ========================
 * replaces RMA/HashTable.cpp:644-705 (one MPI_Get per bucket out of the window over the range-split hashData) and :1066-1087
 * (needsProcessing by bucket owner) with the north star's all-to-all pair: every LOOKUP of a rank's reads travels to the rank that
 * owns the bucket (queries out), the owner walks its slice of the record array and the matching records travel back (hits back);
 * no rank ever holds more than its slice of the bucket table and of the records. A query is one minimizer RUN of a read
 * (index_runs_kernel): bucket, key fingerprint, the run's windows, occurrence and strand — the owner applies exactly
 * probe_runs_kernel's test (fingerprint, not the read itself, the one window the record's minimizer offset names lies inside the
 * run), so the set of (window, record) candidates is the replicated flow's.
 *   query : x = bucket << 32 | read index inside the requester's range ; y = PQ_Y(requester rank, fingerprint, strand, occ - first, end, first)
 *   hit   : x = requester rank << 32 | read index inside its range      ; y = HIT_MAKE(window, id, suffix, strand relation, length)
 * ======================================================================================================================= */
#define PQ_Y(rank, fp, rev, delta, wend, wstart) \
    (((u64)(rank) << 47) | ((u64)(fp) << 37) | ((u64)(rev) << 36) | ((u64)(delta) << 30) | ((u64)(wend) << 15) | (u64)(wstart))
#define PQ_WSTART(y) ((int)((y)&0x7FFFu))
#define PQ_WEND(y) ((int)(((y) >> 15) & 0x7FFFu))
#define PQ_DELTA(y) ((int)(((y) >> 30) & 63u))
#define PQ_REV(y) ((u32)(((y) >> 36) & 1u))
#define PQ_FP(y) ((u32)(((y) >> 37) & 0x3FFu))
#define PQ_RANK(y) ((u32)(((y) >> 47) & 0x3Fu))

struct RouteByHitRank { /* {rank << 32 | read index, hit} */
    __device__ __forceinline__ u32 operator()(const ulonglong2 &h) const { return (u32)(h.x >> 32); }
};

/* runs of the reads [lo, hi) -> queries; reads whose run list is unusable (ties, too many runs) -> slow list (pq_slow_kernel).
 * LPR u32 words of run entries per read (index_runs_kernel); one thread per word = two entries. out == nullptr: count only. */
template <int LPR>
__global__ void __launch_bounds__(256) pq_make_kernel(DiscoView v, const u32 *__restrict__ runs, u64 lo, u64 hi, u32 my_rank, ulonglong2 *__restrict__ out,
                                                      u64 *__restrict__ n_out, u32 *__restrict__ slow, u32 slow_cap, u32 *__restrict__ n_slow)
{
    const u64 nloc = hi - lo, total = nloc * (u64)LPR;
    const u64 t0 = (u64)blockIdx.x * 256u + threadIdx.x;
    const u32 lane = threadIdx.x & 63u;
    for (u64 base = t0 - lane; base < total; base += (u64)gridDim.x * 256u) { /* whole wavefronts walk together (ballots) */
        const u64 t = base + lane;
        const bool in = t < total;
        const u64 ri = in ? t / LPR : 0;
        const u32 e = (u32)(t % LPR);
        const u32 rw = in ? runs[t] : 0xFFFFFFFFu;
        const u32 e0 = rw & 0xFFFFu, e1 = rw >> 16;
        const u32 first = (u32)__shfl((int)e0, (int)(lane - e)); /* entry 0 of this read's list (LPR lanes of one read are neighbours) */
        const bool slowread = first == 0xFFFEu;
        const u32 nx = (u32)__shfl_down((int)e0, 1);
        const u64 A = lo + ri;
        const int L = in ? (int)v.len[A] : 0;
        const u32 npos = (u32)(L - v.k);
        const bool v0 = in && !slowread && e0 < 0xFFFEu, v1 = in && !slowread && e1 < 0xFFFEu;
        if (in && slowread && e == 0 && slow) { /* (first pass only) */
            const u32 si = atomicAdd(n_slow, 1u);
            if (si < slow_cap) slow[si] = (u32)ri;
        }
        const u32 wend0 = v1 ? RUN_W(e1) : npos;
        const u32 wend1 = (e + 1 < (u32)LPR && nx < 0xFFFEu) ? RUN_W(nx) : npos;
        const u64 m0 = __ballot(v0), m1 = __ballot(v1);
        const u32 cnt = (u32)__popcll(m0) + (u32)__popcll(m1);
        if (cnt == 0) continue;
        u64 wbase = 0;
        if (lane == 0) wbase = atomicAdd(n_out, (u64)cnt);
        wbase = ((u64)(u32)__shfl((int)(u32)(wbase >> 32), 0) << 32) | (u32)__shfl((int)(u32)wbase, 0);
        if (!out) continue;
        const u64 lt = lane_mask_lt();
        const u64 pos = wbase + (u32)__popcll(m0 & lt) + (u32)__popcll(m1 & lt);
        const u64 *row = v.reads + A * v.S;
        auto emit = [&](u64 where, u32 en, u32 wend) {
            const u32 wstart = RUN_W(en), delta = RUN_DELTA(en), rev = RUN_STRAND(en);
            const u64 key = mmer_key(row, v.S, (int)(wstart + delta), v.m);
            out[where] = make_ulonglong2(((key >> v.bshift) << 32) | (u64)(u32)ri, PQ_Y(my_rank, KEY_FP(key), rev, delta, wend, wstart));
        };
        if (v0) emit(pos, e0, wend0);
        if (v1) emit(pos + 1, e1, wend1);
    }
}

/* reads without a usable run list (ties, too many runs) — or, for shapes the index pass leaves no runs for (windows other than 17
 * m-mers, reads beyond 256 bases), ALL reads — the long way: window_minimizer's rule (disco_device.h) for every window from the order
 * words of its NF m-mers, one query per maximal range of consecutive windows with the same (occurrence, strand) — with ties an
 * occurrence can own several such ranges; their windows are disjoint, so the owner's test still yields every (window, record) pair
 * once. One thread per read. out == nullptr: queries per read -> cnt; else the queries at out + start[i]. list == nullptr: read i of
 * the range itself. */
template <bool LONGK>
__global__ void pq_slow_kernel(DiscoView v, const u32 *__restrict__ list, u64 n_list, u64 lo, u32 my_rank, u32 *__restrict__ cnt, const u64 *__restrict__ start,
                               ulonglong2 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_list; i += (u64)gridDim.x * blockDim.x) {
        const u64 ri = list ? (u64)list[i] : i, A = lo + ri;
        const u64 *row = v.reads + A * v.S;
        const int L = v.len[A], k = v.k, m = v.m, nf = k - m + 1, npos = L - k;
        ulonglong2 *o = out ? out + start[i] : nullptr;
        u32 n = 0;
        int run_w = 0, run_p = -1;
        u32 run_rev = 0;
        auto flush = [&](int wend) {
            if (run_p < 0) return;
            if (o) {
                const u64 key = mmer_key(row, v.S, run_p, m);
                o[n] = make_ulonglong2(((key >> v.bshift) << 32) | (u64)(u32)ri, PQ_Y(my_rank, KEY_FP(key), run_rev, run_p - run_w, wend, run_w));
            }
            n++;
        };
        for (int w = 0; w < npos; w++) {
            u32 k1 = 0xFFFFFFFFu, k2 = 0xFFFFFFFFu, st1 = 0;
            for (int f = 0; f < nf; f++) {
                const u32 ow = mmer_order(row, v.S, w + f, m);
                const u32 a1 = (ow & ~0x1FFu) | (u32)f, a2 = (ow & ~0x1FFu) | (u32)(511 - f);
                if (a1 < k1) {
                    k1 = a1;
                    st1 = ow & 1u;
                }
                k2 = a2 < k2 ? a2 : k2;
            }
            const int f1 = (int)(k1 & 511u), f2 = 511 - (int)(k2 & 511u);
            u32 rev;
            int p;
            if (f1 == f2) {
                rev = st1;
                p = w + f1;
            } else {
                rev = kmer_is_rev<false, LONGK>(row, v.S, w, k);
                p = w + (rev ? f2 : f1);
            }
            if (p != run_p || rev != run_rev) {
                flush(w);
                run_w = w;
                run_p = p;
                run_rev = rev;
            }
        }
        flush(npos);
        if (!out) cnt[i] = n;
    }
}

/* the owner's side: one thread per query walks its bucket (bkt / ent: this rank's slices, bucket indices relative to blo, record
 * positions relative to the slice). FILL = false: hits per query -> cnt; FILL = true: the hits at out + start[query]. */
template <bool FILL>
__global__ void __launch_bounds__(256) pq_answer_kernel(const ulonglong2 *__restrict__ q, u64 nq, const u32 *__restrict__ bkt, const u64 *__restrict__ ent, u64 blo,
                                                        u64 per, int nf, u32 *__restrict__ cnt, const u64 *__restrict__ start, ulonglong2 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nq; i += (u64)gridDim.x * blockDim.x) {
        const ulonglong2 qq = q[i];
        const u64 b = (qq.x >> 32) - blo;
        const u32 idx = (u32)qq.x, rank = PQ_RANK(qq.y), fp = PQ_FP(qq.y), rev = PQ_REV(qq.y);
        const int wstart = PQ_WSTART(qq.y), wend = PQ_WEND(qq.y), prel = wstart + PQ_DELTA(qq.y);
        const u64 self = (u64)rank * per + idx; /* BG/OverlapGraph.cpp:421,655: a read is no candidate of its own */
        const u32 s = bkt[b], e = bkt[b + 1];
        u32 n = 0;
        u64 at = FILL ? start[i] : 0;
        for (u32 r = s; r < e; r++) {
            const u64 pay = ent[r];
            const int t = (int)PAY_T(pay);
            const int w = rev ? prel - (nf - 1 - t) : prel - t;
            if (PAY_FP(pay) == fp && PAY_ID(pay) != self && w >= wstart && w < wend) {
                if (FILL) out[at + n] = make_ulonglong2(((u64)rank << 32) | idx, HIT_MAKE(w, PAY_ID(pay), PAY_SUFFIX(pay), PAY_REV(pay) ^ rev, PAY_LEN(pay)));
                n++;
            }
        }
        if (!FILL) cnt[i] = n;
    }
}

/* the requester's side: hits -> rows of the hit buffer. count: row_cnt[lo + idx]++ ; place: hits[row_start[idx] + slot] */
__global__ void pq_rows_count_kernel(const ulonglong2 *__restrict__ h, u64 n, u32 *__restrict__ cnt_own)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) atomicAdd(&cnt_own[(u32)h[i].x], 1u);
}
__global__ void pq_rows_place_kernel(const ulonglong2 *__restrict__ h, u64 n, const u64 *__restrict__ start_own, u32 *__restrict__ cursor_own, u64 *__restrict__ hits)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 idx = (u32)h[i].x;
        hits[start_own[idx] + atomicAdd(&cursor_own[idx], 1u)] = h[i].y;
    }
}
/* per-read headers: by read id for rows beyond the register paths, by position in the processing order for everybody (what
 * probe_kernel / probe_runs_kernel leave) */
__global__ void pq_rows_meta_kernel(const u64 *__restrict__ order, u64 lo, u64 nq, const u16 *__restrict__ len, const u64 *__restrict__ start_own,
                                    const u32 *__restrict__ cnt_own, u64 *__restrict__ row_start, u32 *__restrict__ row_cnt, ulonglong2 *__restrict__ meta_ord)
{
    u64 pos = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; pos < nq; pos += (u64)gridDim.x * blockDim.x) {
        const u64 A = order ? ORDER_ID(order[pos]) : lo + pos;
        const u32 c = cnt_own[A - lo];
        const u64 rs = start_own[A - lo];
        if (c > 64) {
            row_start[A] = rs;
            row_cnt[A] = c;
        }
        meta_ord[pos] = make_ulonglong2(rs, (u64)c | ((u64)len[A] << 32));
    }
}

/* u8 flags of the own range only: everything outside [lo, hi) is not this rank's business (fetch_contained scans all n) */
__global__ void sum_u32_kernel(const u32 *__restrict__ p, u64 n, u64 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 s = 0;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) s += p[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

#endif /* DISCO_DIST_H_ */
