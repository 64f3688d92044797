/*
 * disco_kernels.h — hand-written HIP kernels of the BuildGraph hot path for gfx950 (MI355X, wave64).
 *
 * Everything here is integer / compare work bounded by HBM traffic (no MFMA).  One wavefront (= one 64-thread
 * workgroup, so __syncthreads() is a wave-level fence) owns one read / graph node at a time; cross-lane
 * compaction uses ballot + popcount prefix, per-read staging lives in LDS.
 *
 * Reference semantics restated by each kernel are cited as BG/<file>:<lines>
 * (BG/ = /root/reference/src/BuildGraph/src/).
 */
#ifndef DISCO_KERNELS_H_
#define DISCO_KERNELS_H_

#include "disco_device.h"
#include "readgen.h"

/* ---- tunables -------------------------------------------------------------------------------------------------- */
#define PROBE_ROWCAP 512  /* room a wave makes sure of in its chunk before it starts a read (longer rows: BIG pass)    */
#define PROBE_SEGW 128    /* k-mer windows of a read handled per segment (reads up to 128+k bp: one segment)    */
#define PROBE_SEGP (PROBE_SEGW + 64) /* m-mer positions a segment covers: windows + (k - m) <= 63              */
#define PROBE_ACAP 32     /* words of the query read's own row staged in LDS (reads up to 1024 bp)              */
#define PROBE_CHUNK 4096  /* hit slots a wave reserves from the global bump pointer at a time                   */
#define ES_MID 1024       /* edge_select: rows of ES_CAP+1..ES_MID hits get LDS arrays of their own (edge_select_mid_kernel) */
#define ES_DUPBITS 11     /* edge_select: byte table of the duplicate-destination pre-check                         */
#define ES_DUPTAB (1 << ES_DUPBITS)
#ifndef ES_CAP
#define ES_CAP 256        /* edge_select: hits of one read sorted in LDS (longer rows: global-scratch variant)  */
#endif
#ifndef TR_CAP
#define TR_CAP 256        /* transitive_mark: neighbours of one node in LDS                                     */
#endif
/* the marking's time follows its resident waves (24 / 28 / 32 blocks per CU: 20.0 / 19.8 / 19.0 ms at 50 M reads): where no node, or
 * hardly any, has more than 128 neighbours — every read set of ordinary coverage — the variant with LDS arrays for 128 runs at eight
 * waves per SIMD (2.8 KB a block), nodes beyond go to the big-node pass; else the variant for TR_CAP at six */
#define TR_CAP_SMALL 128
#define VERIFY_SW 8 /* device row stride (words) of the staged variants: reads up to 256 bp, 64-byte rows */
#define ORDER_BUCKET(key, shift) (((key) * 0x9E3779B1u) >> (shift)) /* bucket of a read-level minimizer in the grouping ("processing order") */
#define SCAN_ITEMS 16     /* elements per thread in the scan kernels                                            */
#define SCAN_BLOCK 256
#define SCAN_TILE (SCAN_ITEMS * SCAN_BLOCK)

/* adjacency reference word of a node: position(40) | degree(24) << 40 ; transitive flag = bit 15 of an entry */
#define REF_MAKE(pos, deg) (((u64)(deg) << 40) | (u64)(pos))
#define REF_POS(r) ((r)&0xFFFFFFFFFFull)
#define REF_DEG(r) ((u32)((r) >> 40))
#define ADJ_FLAG (1ull << 15)
/* an entry of the extras list: the node (31 bits, like ADJ_DST) and the extra's place among that node's extras */
/* between edge selection and the twin search the flag bit says "the twin of this entry is hidden from the other read" */
#define ADJ_HIDDEN_OF(hit) ((u64)HIT_HIDDEN(hit) << 15)
#define EXTRA_NODE_MAKE(w, slot) ((u64)(w) | ((u64)(slot) << 32))
#define EXTRA_NODE(e) ((e)&0xFFFFFFFFull)
#define EXTRA_SLOT(e) ((u32)((e) >> 32))

/* global counters (u64 each) */
enum {
    CTR_KMER_HITS = 0,
    CTR_RAW_HITS,
    CTR_HITS_NEEDED,   /* high-water mark of the hit bump pointer                       */
    CTR_OVERFLOW,      /* != 0: a buffer was too small, results incomplete -> regrow      */
    CTR_CAP_SITES,
    CTR_ASYM,
    CTR_N_CONTAINED,
    CTR_BAD_LEN,
    CTR_BIG_ROWS,
    CTR_ES_BIG,
    CTR_TR_BIG,
    CTR_MAX_DEG,
    CTR_MAX_ROW,
    CTR_TW_UP,
    CTR_TW_DOWN,
    CTR_ES_SLOW, /* rows that took the sequential accept scan */
    CTR_ADJ_TOTAL, /* directed edges selected (sum of degrees of the query range) */
    CTR_MAX_LEN, /* longest read */
    CTR_DROPPED, /* verified hits to non-contained reads that edge selection did not turn into an edge */
    CTR_MIN_LEN, /* ~shortest read (stored complemented so that atomicMax finds the minimum from a zeroed counter) */
    CTR_DROP_ITEMS, /* dropped hits recorded in (or, beyond its capacity, lost to) the drop list of edge selection */
    CTR_SHORT_MAX,  /* two row classes: longest read of at most 256 bases */
    CTR_TR_MID,     /* nodes with more than TR_CAP_SMALL finds (edge selection counts them: which variant of the marking runs) */
    CTR_ES_MID,     /* rows with more than 64 verified hits (verify counts them: which variant of edge_select_flat_kernel runs) */
    CTR_COUNT
};

/* dynamic work distribution: a wave grabs WQ_CHUNK consecutive items at a time from a global counter, so the run time does
 * not depend on how many workgroups happen to be resident (a static blockIdx-strided loop ran 30-40 % slower whenever
 * the grid was not a multiple of the resident workgroups) */
#define WQ_CHUNK 64
/* entry of the processing order: read id | length << 32 — the length travels with the id, so the probe does not fetch len[id]
 * (a random 64-byte line per read) for reads it takes out of order */
#define ORDER_MAKE(id, l) ((u64)(id) | ((u64)(l) << 32))
#define ORDER_ID(o) ((o) & 0xFFFFFFFFull)
#define ORDER_LEN(o) ((int)((o) >> 32))
/* the first active lane's value as a SCALAR: a value broadcast by a cross-lane shuffle is divergent to the compiler, and with it every
 * loop bound, branch and address derived from it (counters in vector registers, exec-mask loops, saveexec around uniform branches) */
__device__ __forceinline__ u32 uniform_u32(u32 x) { return (u32)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ u64 uniform_u64(u64 x)
{
    return ((u64)uniform_u32((u32)(x >> 32)) << 32) | uniform_u32((u32)x);
}

template <u32 CHUNK = WQ_CHUNK>
__device__ __forceinline__ bool wq_grab(u64 *counter, u64 n, u64 &beg, u64 &end)
{
    u64 b = 0;
    if ((threadIdx.x & 63) == 0) b = atomicAdd(counter, (u64)CHUNK);
    /* lane 0's value as a SCALAR (every lane of the wave is here): a value that comes out of a cross-lane shuffle is divergent to the
     * compiler, and with it every loop over the chunk — counters in vector registers, exec-mask loops, wave-uniform branches through
     * saveexec. Round 5: read as a scalar the chunk loops of all kernels are scalar loops */
#ifdef WQ_EXP_SHFL /* timing experiment: the broadcast of rounds 1-4 (the chunk bounds divergent to the compiler) */
    b = __shfl(b, 0);
#else
    b = uniform_u64(b);
#endif
    beg = b;
    end = (b + CHUNK < n) ? b + CHUNK : n;
    return b < n;
}

/* the same chunks, MULT of them per atomic: the work queue is ONE address, and returning atomics on one address are served one after
 * the other — 12.3 ns apiece on MI355X (edge_select_mid_kernel: 80 681 one-row grabs took 0.99 ms whatever the grid). For kernels whose
 * items are small (a row per grab). The passes over the reads take 64 (verify: 32) reads per grab — 0.8 / 1.6·10^6 grabs, 10 / 19 ms of
 * that unit per kernel: busy half the time (verify: 87 %), yet four or eight chunks per atomic changed nothing but the tails (+0.2 .. 0.4 ms
 * per kernel): measured and not kept. gnext / gend: the wave's allocation (zero to begin with). Chunks stay aligned to CHUNK. */
template <u32 CHUNK = WQ_CHUNK, u32 MULT = 4>
__device__ __forceinline__ bool wq_grab_multi(u64 *counter, u64 n, u64 &beg, u64 &end, u64 &gnext, u64 &gend)
{
    if (gnext >= gend) {
        if (!wq_grab<CHUNK * MULT>(counter, n, gnext, gend)) return false;
    }
    beg = gnext;
    end = beg + CHUNK < gend ? beg + CHUNK : gend;
    gnext = end;
    return true;
}

/* round 6: the queue as WQ_NQ sub-queues, one counter each in a 128-byte slot of its own. The one address of wq_grab serves a returning
 * atomic per 12.3 ns: with 32 reads per grab (verify) that unit alone needs 19 of the kernel's 21.7 ms (an EMPTY verify — no candidate
 * touched — still took 18.9 ms: gpurun_out r6_run6; with four chunks per atomic 4.8), so every wave stands in line for it. Sub-queue q
 * covers the q-th of WQ_NQ contiguous ranges of the chunks; a workgroup starts at sub-queue blockIdx % WQ_NQ (workgroups go to the eight
 * XCDs in turn: an XCD's waves work through one contiguous eighth of the processing order and its L2 sees that eighth's rows) and moves
 * on to the next one when its own is exhausted — every chunk is handed out exactly once, the tail is shared by everybody. No register
 * lives across chunks but the sub-queue's number. Counters: wq[16 q]; the host clears all of them (WQ_WORDS). */
#define WQ_NQ 8u
#define WQ_WORDS (16u * WQ_NQ)
struct WqSplit {
    u32 qi = 0; /* sub-queues this wave has left behind (wave uniform) */
};
template <u32 CHUNK = WQ_CHUNK>
__device__ __forceinline__ bool wq_grab_split(u64 *counters, u64 n, u64 &beg, u64 &end, WqSplit &st)
{
    /* (32-bit chunk arithmetic: a launch has fewer than 2^32 chunks; one vector register for the atomic's result) */
    const u32 nchunks = (u32)((n + CHUNK - 1u) / CHUNK);
    const u32 per = (nchunks + WQ_NQ - 1u) / WQ_NQ; /* chunks of a sub-queue (the last ones may hold fewer) */
    while (st.qi < WQ_NQ) {
        const u32 q = ((u32)blockIdx.x + st.qi) % WQ_NQ;
        const u32 first = q * per;
        const u32 mine = first < nchunks ? (nchunks - first < per ? nchunks - first : per) : 0u;
        u32 b = mine; /* (an empty sub-queue is not asked) */
        if (mine) {
            u32 t = 0;
            if ((threadIdx.x & 63) == 0) t = atomicAdd((u32 *)&counters[16u * q], 1u);
            b = uniform_u32(t);
        }
        if (b < mine) {
            beg = (u64)(first + b) * CHUNK;
            end = beg + CHUNK < n ? beg + CHUNK : n;
            return true;
        }
        st.qi++;
    }
    return false;
}

struct DiscoView {
    const u64 *reads; /* [n][S] */
    const u16 *len;   /* [n]    */
    u64 n;
    int S;
    int k;
    int m; /* minimizer length, disco_minimizer_len(k) */
    /* index */
    const u32 *bkt;         /* [T+1] bucket b = entries [bkt[b], bkt[b+1]) */
    const u64 *ent;         /* [2n] 8-byte records (PAY_MAKE)               */
    int bshift;             /* bucket = key >> bshift                       */
    /* query shard */
    u64 q_lo, q_hi;
    u64 *ctr;
    u64 *wq; /* work-queue counter of the launch (zeroed by the host) */
    /* two row classes ("two classes of rows"; null / 0 otherwise): the table keeps its 64-byte rows although a few
     * reads are longer than 256 bases. Row i < n of a LONG read holds its first 256 bases (all an overlap with a short read at its
     * prefix end can touch); rows [n, n + n_long) hold the LAST tailb bases of the long reads (the suffix end: the suffix record of long
     * read long_ids[j] carries the id n + j, so that nothing between the index and the compare has to know); the whole read lives in
     * full[j][SL]. ovf[i] = number of long reads in front of read i (its j, if it is one). */
    const u64 *full;
    const u32 *ovf;
    const u32 *long_ids;
    u32 n_long;
    int SL;
    int tailb; /* 160 or 256: the bases the staged compare of the short class moves per row */
};

/* which graph nodes are this rank's. One GPU and the id-range form of the multi-GPU flow: the range [lo, hi). "Ranks own loci"
 * (DESIGN.md section 6): reads are dealt to the ranks by their read-level minimizer — otab[v] names the owner of read v, list holds the
 * own reads (ORDER_MAKE entries, in the rank's processing order) — and every "is it mine" / "whose is it" of the flow goes through
 * this instead of through id arithmetic (replaces needsProcessing, RMA/HashTable.cpp:1066-1087: work goes to the rank whose data it meets) */
struct OwnSet {
    const u8 *otab; /* [n] owner of every read, or null: the range */
    u32 me;
    u64 lo, hi;
    const u64 *list;
    u64 n_own;
    const u32 *ids; /* otab != null: the own reads once more, in (nearly) ascending id order — what a kernel that streams per-node tables walks */
    __device__ __forceinline__ bool mine(u64 v) const { return otab ? otab[v] == (u8)me : (v >= lo && v < hi); }
    __device__ __forceinline__ u64 count() const { return otab ? n_own : hi - lo; }
    __device__ __forceinline__ u64 node(u64 i) const { return otab ? ORDER_ID(list[i]) : lo + i; }       /* i-th own node in processing order */
    __device__ __forceinline__ u64 node_by_id(u64 i) const { return otab ? (u64)ids[i] : lo + i; }       /* ... in id order */
};
/* owner of a read-level minimizer key among G ranks: the key's grouping hash (ORDER_BUCKET's multiplicative hash) cut into G equal
 * ranges — reads that share their key (a group: the reads of one locus) always share their owner, and a rank's groups are a
 * contiguous range of the grouping's buckets */
__host__ __device__ __forceinline__ u32 disco_key_owner(u32 key, u32 G) { return (u32)(((u64)(key * 0x9E3779B1u) * (u64)G) >> 32); }

/* ================================================================================================================
 * synthetic reads straight into HBM (bench / tests) — twin of readgen.h / readgen.py
 * ============================================================================================================== */
/* rows [r_lo, r_hi) of the table (the whole table on one GPU, the rank's own range in the multi-GPU flow) */
__global__ void generate_reads_kernel(disco_genspec spec, u64 *__restrict__ reads, u16 *__restrict__ len, int S, u64 r_lo, u64 r_hi)
{
    u64 gid = r_lo * (u64)S + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 total = r_hi * (u64)S;
    for (; gid < total; gid += (u64)gridDim.x * blockDim.x) {
        u64 r = gid / S;
        int w = (int)(gid % S);
        disco_readloc loc = disco_read_location(&spec, r);
        u64 word = 0;
        int base0 = w * 32;
        for (int i = 0; i < 32; i++) {
            int p = base0 + i;
            if (p < (int)loc.len) word |= (u64)disco_read_base(&spec, &loc, (u32)p) << (62 - 2 * i);
        }
        reads[gid] = word;
        if (w == 0) len[r] = (u16)loc.len;
    }
}

/* substitution errors into resident reads (disco_substitute_bases) */
__global__ void substitute_bases_kernel(u64 seed, u32 rate_ppm, u64 *__restrict__ reads, const u16 *__restrict__ len, int S, u64 r_lo, u64 r_hi)
{
    u64 gid = r_lo * (u64)S + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = r_hi * (u64)S;
    for (; gid < total; gid += (u64)gridDim.x * blockDim.x) {
        const u64 r = gid / S;
        const int base0 = (int)(gid % S) * 32, L = (int)len[r];
        if (base0 >= L) continue;
        const u64 word = reads[gid];
        u64 out = 0;
        for (int i = 0; i < 32; i++) {
            u32 b = (u32)(word >> (62 - 2 * i)) & 3u;
            if (base0 + i < L) b = disco_substituted_base(seed, rate_ppm, r, (u32)(base0 + i), b);
            else b = 0;
            out |= (u64)b << (62 - 2 * i);
        }
        reads[gid] = out;
    }
}

/* reads must satisfy min_overlap < len <= 32767 (BG/Dataset.cpp:305, BG/HashTable.cpp:531) and fit the stride */
__global__ void validate_len_kernel(const u16 *__restrict__ len, u64 n, int S, int min_overlap, u64 *ctr)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 bad = 0, mx = 0, mn = 0xFFFFu;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 L = len[i];
        if (L <= (u32)min_overlap || L > 32767u || L > (u32)S * 32u) bad++;
        mx = L > mx ? L : mx;
        mn = L < mn ? L : mn;
    }
    if (bad) atomicAdd(&ctr[CTR_BAD_LEN], (u64)bad);
    for (int o = 32; o > 0; o >>= 1) {
        mx = max(mx, (u32)__shfl_down(mx, o));
        mn = min(mn, (u32)__shfl_down(mn, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMax(&ctr[CTR_MAX_LEN], (u64)mx);
        atomicMax(&ctr[CTR_MIN_LEN], (u64)(0xFFFFu - mn));
    }
}

/* ================================================================================================================
 * index build — replaces HashTable::insertDataset (BG/HashTable.cpp:46-114): count per bucket (populateReadLengths),
 * exclusive prefix sum (:58-67), fill (populateReadData / insertIntoTable :423-514).
 * The reference keys a record by a hash of the whole end k-mer and probes once per k-mer of every read (111 random
 * probes per 150 bp read). Here a record is keyed by the k-mer's MINIMIZER (smallest strand-symmetric m-mer hash, m = min(k, 23), odd)
 * and carries the minimizer's offset t inside the canonical k-mer: consecutive k-mer windows of a query read share their
 * minimizer, so the probe needs one bucket lookup per minimizer occurrence (about 13 per read) and recovers the window
 * from t by arithmetic. Which records a k-mer window matches is unchanged (exact k-mer equality, re-checked by the
 * verify compare), so the hit set is the reference's.
 * In-bucket order is arbitrary here: every consumer re-establishes the reference's bucket order (ascending read id,
 * prefix record before suffix record) from the ids carried in the hits (see HIT_MAKE / CKEY_MAKE).
 * bkt has T+1 slots (T <= 2^32); counts go to bkt[b] (the counting atomic hands each record its slot inside the bucket);
 * after the in-place exclusive scan of bkt[0..T] bucket b is [bkt[b], bkt[b+1]) and the fill is a plain scatter.
 * ============================================================================================================== */
/* One thread per read, ONE rolling pass over the read's m-mers (forward and reverse-complement m-mer updated per base,
 * m <= 23: 46 bits; every row word is loaded once). The first nf m-mers are the prefix k-mer's window, the last nf the suffix
 * k-mer's: their minimizers come out of the pass with window_minimizer's rule (two running minima per window; the rare tie
 * goes through kmer_is_rev as there). The same pass yields the read's grouping key for the probe / verify order:
 * okey[i] = smallest 32-bit order hash among ALL m-mers of the read (see "processing order" below).
 * (The previous version called the random-access mmer_order 34 times per read: 33 L2 requests per read, 10.2 ms at 50 M reads,
 * plus a separate 4.2 ms pass for the keys.) */
/* Reads [lo, hi) (the whole table on one GPU, the rank's own range in the multi-GPU flow); rec is indexed from lo. COUNT = false
 * (multi-GPU): no counting atomics — the records are routed to the rank that owns their bucket range first and counted there
 * (shard_count_kernel), which is what replaces the range-partitioned hashData of RMA/HashTable.cpp:95-116. */
/* LONGCLASS (two row classes, single GPU): [lo, hi) counts the long reads; read x is long_ids[x], its row full[x][SL]; its records, key
 * and slot go where the read's would (rec is indexed by read id); the suffix record carries the id of the read's tail row, n + x */
/* list (multi-GPU flow, ranks own loci: the own reads are no id range): position x of [lo, hi) stands for read ORDER_ID(list[x]);
 * records, slots and keys are indexed by position */
template <bool COUNT, bool LONGK = false, bool LONGCLASS = false>
__global__ void __launch_bounds__(256) index_count_kernel(DiscoView v, u32 *__restrict__ bkt, ulonglong2 *__restrict__ rec, u32 *__restrict__ okey, u64 lo, u64 hi,
                                                          u32 *__restrict__ ocnt, u32 *__restrict__ oslot, u32 oshift, const u64 *__restrict__ list = nullptr)
{
    /* rec[2i], rec[2i+1] = {bucket << 32 | slot inside the bucket, record} of the prefix / suffix k-mer of read i: the slot is
     * what the counting atomic returns, so the fill pass needs no second round of atomics */
    const u64 x = lo + (u64)blockIdx.x * 256u + threadIdx.x;
    if (x >= hi) return;
    const u64 i = LONGCLASS ? (u64)v.long_ids[x] : x; /* what the records, keys and slots are indexed by */
    const u64 lw = (!LONGCLASS && list) ? list[x] : 0ull;
    const u64 rid = (!LONGCLASS && list) ? ORDER_ID(lw) : i; /* the read */
    if (LONGCLASS) lo = 0;
    const int S = LONGCLASS ? v.SL : v.S;
    const u64 *__restrict__ p = LONGCLASS ? v.full + x * (u64)S : v.reads + rid * (u64)S;
    if (!LONGCLASS && v.full && v.len[rid] > DISCO_SHORT_MAX) return; /* (a long read's turn comes with the LONGCLASS launch) */
    const int L = (!LONGCLASS && list) ? ORDER_LEN(lw) : (int)v.len[rid], k = v.k, m = v.m, nf = k - m + 1;
    const int nmm = L - m + 1; /* m-mer positions (L >= k: validate_len_kernel) */
    const int sfx0 = nmm - nf; /* = L - k: first m-mer of the suffix k-mer */
    const u64 mask = (1ull << (2 * m)) - 1ull;
    const int rsh = 2 * (m - 1);
    u64 f = 0, r = 0, word = 0;
    int pos = 0; /* bases consumed */
    auto next = [&]() {
        if ((pos & 31) == 0) word = p[pos >> 5];
        const u32 b = (u32)(word >> 62);
        word <<= 2;
        ++pos;
        f = ((f << 2) | b) & mask;
        r = (r >> 2) | ((u64)(3u - b) << rsh);
    };
    u32 best = 0xFFFFFFFFu;
    auto order_word = [&]() { /* order word of the next m-mer (mmer_order's value) */
        next();
        const bool st = r < f;
        const u32 h = order_hash32(st ? r : f);
        best = min(best, h);
        return (h & ~0x1FFu) | (u32)st;
    };
    for (int q = 0; q < m - 1; ++q) next();
    u32 k1p = 0xFFFFFFFFu, k2p = 0xFFFFFFFFu, k1s = 0xFFFFFFFFu, k2s = 0xFFFFFFFFu;
    int q = 0;
    for (; q < nf; ++q) { /* prefix window (and the suffix window where the two overlap: L < k + nf) */
        const u32 o = order_word();
        k1p = min(k1p, o + ((u32)q << 1));
        k2p = min(k2p, o + ((u32)(63 - q) << 1));
        if (q >= sfx0) {
            k1s = min(k1s, o + ((u32)(q - sfx0) << 1));
            k2s = min(k2s, o + ((u32)(63 - (q - sfx0)) << 1));
        }
    }
    for (; q < sfx0; ++q) order_word();
    for (; q < nmm; ++q) { /* suffix window */
        const u32 o = order_word();
        k1s = min(k1s, o + ((u32)(q - sfx0) << 1));
        k2s = min(k2s, o + ((u32)(63 - (q - sfx0)) << 1));
    }
    if (okey) okey[i] = best;
    if (ocnt) oslot[i - lo] = atomicAdd(&ocnt[ORDER_BUCKET(best, oshift)], 1u); /* the grouping's counting pass (order_count_kernel), fused */
    auto resolve = [&](u32 k1, u32 k2, int j0, u32 &t, u32 &rev) {
        const int ffirst = (int)((k1 >> 1) & 63u), flast = 63 - (int)((k2 >> 1) & 63u);
        int fsel = ffirst;
        if (ffirst == flast)
            rev = k1 & 1u;
        else {
            rev = kmer_is_rev<false, LONGK>(p, S, j0, k);
            fsel = rev ? flast : ffirst;
        }
        t = rev ? (u32)(nf - 1 - fsel) : (u32)fsel;
        return mmer_key(p, S, j0 + fsel, m);
    };
    u32 tp, rp, ts, rs;
    const u64 kp = resolve(k1p, k2p, 0, tp, rp);
    const u64 ks = resolve(k1s, k2s, sfx0, ts, rs);
    const u64 bp = kp >> v.bshift, bs = ks >> v.bshift;
    const u32 sp = COUNT ? atomicAdd(&bkt[bp], 1u) : 0u;
    const u32 ss = COUNT ? atomicAdd(&bkt[bs], 1u) : 0u;
    rec[2 * (i - lo)] = make_ulonglong2((bp << 32) | sp, PAY_MAKE(kp, rid, tp, rp, 0, L));
    rec[2 * (i - lo) + 1] = make_ulonglong2((bs << 32) | ss, PAY_MAKE(ks, LONGCLASS ? v.n + x : rid, ts, rs, 1, L));
}

/* the rolling canonical m-mer of a 2-bit packed row on 32-bit halves, M a constant (odd, 2 M bits = one dword + 2 M - 32 bits): the
 * forward m-mer f and the reverse complement r of the M bases that end at the current position, every thread of the wavefront at the
 * SAME position (the dword switch is a scalar branch, the base leaves the current dword through a bit-field extract with a scalar offset) */
#define RUNS_M 23 /* minimizer length of the specialised instantiations (disco_minimizer_len gives 23 for every k from 23 to 86) */
/* M = 0: the length is a run-time value (17 .. 31: k above 86 takes minimizers of up to 31 bases) — the mask and the place of the entering
 * complement are then scalar operands of the same instructions, and the fold of order_hash32 takes its third product */
template <int M>
struct MmerRoll {
    static_assert(M == 0 || (M > 16 && M <= 24 && (M & 1)), "two dwords, the high one partly used; a 24-bit fold in order_hash32<false>");
    const u64 *__restrict__ p;
    int last_word;
    u32 himask, rsh; /* (M = 0) */
    int pos = 0; /* bases consumed (wave uniform) */
    u64 word = 0;
    u32 cw = 0;
    u32 flo = 0, fhi = 0, rlo = 0, rhi = 0;
    bool st = false;
    u32 clo = 0, chi = 0;
    __device__ __forceinline__ MmerRoll(const u64 *row, int S, int m = M) : p(row), last_word(S - 1), himask((1u << (2 * m - 32)) - 1u), rsh((u32)(2 * m - 34)) {}
    __device__ __forceinline__ void step()
    {
        if ((pos & 15) == 0) {
            if ((pos & 31) == 0) word = p[min(pos >> 5, last_word)];
            cw = (pos & 16) ? (u32)word : (u32)(word >> 32);
        }
        const u32 b = __builtin_amdgcn_ubfe(cw, (u32)(30 - 2 * (pos & 15)), 2u);
        ++pos;
        fhi = __builtin_amdgcn_alignbit(fhi, flo, 30u) & (M ? ((1u << (2 * (M ? M : 17) - 32)) - 1u) : himask);
        flo = (flo << 2) | b;
        rlo = __builtin_amdgcn_alignbit(rhi, rlo, 2u);
        rhi = (rhi >> 2) | ((b ^ 3u) << (M ? (u32)(2 * (M ? M : 17) - 34) : rsh));
        st = (((u64)rhi << 32) | rlo) < (((u64)fhi << 32) | flo); /* the reverse complement is the canonical one (M odd: never equal) */
        clo = st ? rlo : flo;
        chi = st ? rhi : fhi;
    }
    __device__ __forceinline__ u32 hash() const { return order_hash32<M == 0>(((u64)chi << 32) | clo); }
    __device__ __forceinline__ u32 strand() const { return st ? 1u : 0u; }
};
/* a read's run entry (16 bits): first window << 7 | (occurrence - first window) << 1 | strand — windows below 256, a window of at most 64
 * m-mers (round 6: six bits for the offset; five, for NF <= 32, before); 0xFFFF: unused, 0xFFFE in the first entry: no usable list */
#define RUN_W(e) ((u32)(e) >> 7)
#define RUN_DELTA(e) (((u32)(e) >> 1) & 63u)
#define RUN_STRAND(e) ((u32)(e)&1u)
/* ... and as probe_runs_kernel holds it in LDS: slot << 29 | strand << 23 | (occurrence - first) << 17 | end << 8 | first window */
#define OCC_MAKE(slot, e, wend) (((u32)(slot) << 29) | (RUN_STRAND(e) << 23) | (RUN_DELTA(e) << 17) | ((u32)(wend) << 8) | RUN_W(e))
#define OCC_SLOT(d) ((u32)(d) >> 29)
#define OCC_STRAND(d) (((u32)(d) >> 23) & 1u)
#define OCC_DELTA(d) (((u32)(d) >> 17) & 63u)
#define OCC_END(d) (((u32)(d) >> 8) & 0x1FFu)
#define OCC_FIRST(d) ((u32)(d)&0xFFu)

/* ----------------------------------------------------------------------------------------------------------------
 * index_runs_kernel — index_count_kernel's rolling pass, which additionally hands the probe every read's MINIMIZER RUNS, so
 * that probe_runs_kernel starts at the bucket lookups instead of re-deriving, wave per read, what this pass walks anyway
 * (hashing the m-mers and the window minima were half of probe_kernel's instructions: 200 of 421 vector instructions per read,
 * against 31 per m-mer and THREAD here).
 *
 * Window minima, branch free (van Herk / Gil-Werman blocks): the m-mer positions are cut into blocks of NF = k - m + 1, the
 * window length. The window at w = b NF + r is the tail [r, NF) of block b plus the head [0, r) of block b + 1, so its minimum is
 * min(suffix-min of block b at r, prefix-min of block b + 1 at r - 1): per position one step of a running prefix minimum, one
 * step of the backward suffix pass over the block's stored order words, and one combine — no rescan when the minimum leaves
 * the window (the serial sliding minimum needs one at about every ninth position per thread, i.e. at nearly every position for
 * SOME thread of a wavefront: index 11.4 -> 40.9 ms in round 1). Two keys per position as everywhere (window_minimizer's rule,
 * disco_device.h): order hash | position (leftmost of equal hashes) and order hash | 127 - position (rightmost); positions are
 * relative to the start of block b, the strand of the m-mer rides in bit 0. The two minima name the same position iff the
 * smallest hash is unique in the window: m1 ^ m2 == 0xFE exactly then.
 *
 * A RUN is a maximal range of consecutive windows with the same minimizer occurrence. Without ties the occurrence of a window is
 * the position of its unique smallest hash, which never moves left as the window slides: every occurrence has ONE run, runs are
 * disjoint and ordered, and all windows of a run have the strand of the occurrence's m-mer. The read's runs go out as 16-bit
 * entries  first window << 6 | (occurrence - first window) << 1 | strand  (windows < 256, NF <= 32), CAP = 32 NL of them per
 * read, unused ones 0xFFFF. A read with a tie in any window (the same m-mer twice within NF positions, or a collision of the
 * 23-bit order hash: about 2 reads in 10 000 on random sequence, many in low-complexity sequence) or with more than CAP runs
 * is marked 0xFFFE in its first entry: probe_runs_kernel hands those to probe_kernel, which resolves ties the long way.
 * The prefix / suffix k-mer records of the read are windows 0 and L - k of the same pass.
 * The entries are staged in LDS (each thread its own CAP slots) and leave as one coalesced copy per block. */
/* list (multi-GPU flow, ranks own loci): position i of [lo, hi) stands for read ORDER_ID(list[i]), its length rides in the entry;
 * records, runs, keys and slots by position — probe_runs_kernel then reads the run lists in the order it walks the reads */
/* NF = 0 (round 6): the window length is a run-time value nf <= NFMAX — every min-overlap the specialised instantiations (NF = 7, 12, 17,
 * 22, 27: min-overlap 30 .. 50) do not cover, k above 64 (LONGK: the three-word k-mer comparison of the rare tied end k-mer) and
 * minimizers of up to 31 bases included. The block's arrays hold NFMAX words, the loops over a block are unrolled NFMAX times with a
 * SCALAR test per position (every thread of the wavefront is at the same position of the same block), so the work follows nf, not NFMAX. */
template <bool COUNT, int NF, int NL, int NFMAX = NF, bool LONGK = false>
__global__ void __launch_bounds__(256) index_runs_kernel(DiscoView v, u32 *__restrict__ bkt, ulonglong2 *__restrict__ rec, u32 *__restrict__ okey, u64 lo, u64 hi,
                                                         u32 *__restrict__ runs, u32 *__restrict__ ocnt, u32 *__restrict__ oslot, u32 oshift,
                                                         const u64 *__restrict__ list = nullptr)
{
    static_assert(NF == 0 || NF == NFMAX, "a specialised instantiation holds exactly its window");
    constexpr int CAP = 32 * NL;
    __shared__ u16 s_runs[256 * CAP];
    __shared__ u32 s_x[3 * 256]; /* the suffix k-mer's window: its two minima and their base, per thread */
    const u32 tid = threadIdx.x;
    for (u32 x = tid; x < 256u * CAP / 2u; x += 256u) ((u32 *)s_runs)[x] = 0xFFFFFFFFu;
    __syncthreads();
    const u64 i = lo + (u64)blockIdx.x * 256u + tid;
    const u64 lw = (list && i < hi) ? list[i] : 0ull;
    const u64 rid = list ? ORDER_ID(lw) : i; /* the read behind position i */
    /* two row classes: a long read is index_count_kernel's (from its full row); here it only tells the probe to take it the long way */
    const bool other_class = i < hi && v.full && v.len[rid] > DISCO_SHORT_MAX;
    if (other_class) s_runs[tid * CAP] = 0xFFFEu;
    if (i < hi && !other_class) {
        const u64 *__restrict__ p = v.reads + rid * v.S;
        const int L = list ? ORDER_LEN(lw) : (int)v.len[rid], k = v.k;
        const int m = NF ? RUNS_M : v.m; /* (the host takes a specialised instantiation for m = RUNS_M only: runs_lpr_for) */
        const int nf = NF ? NF : k - m + 1; /* (wave uniform) */
        const int nmm = L - m + 1; /* m-mer positions */
        const int npos = nmm - nf; /* = L - k: the probe's windows are [0, npos), window npos is the suffix k-mer */
        /* round 6: the rolling pass on 32-bit halves with the minimizer length a constant (the masks and the place of the entering
         * complement were run-time values: 64-bit shifts by a register, two ANDs), the base taken out of the current dword by a SCALAR
         * offset (every thread of a wavefront is at the same base: the position never depends on the lane — reads shorter than the
         * longest only mask what they keep), and a mix of two 24-bit multiplies behind the fold (order_hash32): 27 -> 21 vector
         * instructions per m-mer and thread in this part of the kernel, which is bound by vector issue outright */
        MmerRoll<NF ? RUNS_M : 0> roll(p, v.S, m);
        u32 best = 0xFFFFFFFFu;
        for (int q = 0; q < m - 1; ++q) roll.step();
        u32 s1[NFMAX], s2[NFMAX]; /* suffix minima of the block before */
#pragma clang loop unroll(full)
        for (int t = 0; t < NFMAX; ++t) s1[t] = s2[t] = 0xFFFFFFFFu;
        u16 *my = s_runs + tid * CAP;
        u32 cnt = 0, lastm = 0xFFFFFFFFu, tie = 0;
        u32 P1 = 0, P2 = 0;
        u32 *sx = s_x + tid;
        /* window w (wave uniform: every thread walks the same positions) with minima m1 / m2 over positions relative to `base`; tw = w - base */
        auto window = [&](int w, u32 m1, u32 m2, int base, int tw) {
            if (w <= npos) {
                if (w == npos) { /* the suffix k-mer: its record; not one of the probe's windows. (Through LDS: as register values the three
                                    would be merged with their old selves behind EVERY window — six copies per window and thread) */
                    sx[0] = m1;
                    sx[256] = m2;
                    sx[512] = (u32)base;
                } else {
                    tie |= (m1 ^ m2) ^ 0xFEu;
                    /* a new occurrence: another hash or another position than the window before had (lastm holds that window's minimum with
                     * its position relative to THIS block: re-based at every block; an occurrence that has left the windows for good
                     * borrows from its hash bits there and matches nothing) */
                    if (((m1 ^ lastm) >> 1) != 0u) {
                        lastm = m1;
                        /* first window << 7 | (occurrence - first window) << 1 | strand: the low byte of the minimum is position << 1 | strand
                         * (bit 8 is never set), the rest is the same for every thread. A list that overflows is marked unusable below: its
                         * last slot may take whatever comes */
                        my[cnt < (u32)CAP ? cnt : (u32)CAP - 1u] = (u16)((m1 & 0xFFu) + (((u32)w << 7) - 2u * (u32)tw)); /* (RUN_W / RUN_DELTA / RUN_STRAND) */
                        ++cnt;
                    }
                }
            }
        };
        const int nblk = npos / nf + 2; /* window npos lies in block npos / nf and is complete once the next block has gone by */
        int q = 0;
        for (int b = 0; b < nblk; ++b) {
            const int base = (b - 1) * nf;
            lastm -= (u32)nf << 1;
            if (b >= 1) window(base, s1[0], s2[0], base, 0); /* the window that IS block b - 1 */
            if (b == 1) {
                P1 = s1[0]; /* window 0: the prefix k-mer's record */
                P2 = s2[0];
            }
            u32 p1 = 0xFFFFFFFFu, p2 = 0xFFFFFFFFu;
            u32 hc[NFMAX];
#pragma clang loop unroll(full)
            for (int t = 0; t < NFMAX; ++t) {
                if (NF != 0 || t < nf) { /* (scalar; a guard, not an exit: a loop with one exit is unrolled whatever its size, and the arrays stay registers) */
                    /* past the read the order word is whatever the row holds there: never part of a window the probe uses, and kept out of the read's key */
                    roll.step();
                    const u32 h = roll.hash();
                    best = min(best, q < nmm ? h : 0xFFFFFFFFu);
                    const u32 o = (h & ~0x1FFu) | roll.strand();
                    ++q;
                    hc[t] = o;
                    p1 = min(p1, o | ((u32)(nf + t) << 1));
                    p2 = min(p2, o | ((u32)(127 - (nf + t)) << 1));
                    if (t + 1 < nf && b >= 1) window(base + t + 1, min(s1[t + 1 < NFMAX ? t + 1 : 0], p1), min(s2[t + 1 < NFMAX ? t + 1 : 0], p2), base, t + 1);
                }
            }
            u32 a1 = 0xFFFFFFFFu, a2 = 0xFFFFFFFFu;
#pragma clang loop unroll(full)
            for (int t = NFMAX - 1; t >= 0; --t) {
                if (NF == 0 && t >= nf) continue; /* (scalar) */
                a1 = min(a1, hc[t] | ((u32)t << 1));
                a2 = min(a2, hc[t] | ((u32)(127 - t) << 1));
                s1[t] = a1;
                s2[t] = a2;
            }
        }
        if (okey) okey[i] = best;
#if !defined(INDEX_EXP_NOATOMIC)
        if (ocnt) oslot[i - lo] = atomicAdd(&ocnt[ORDER_BUCKET(best, oshift)], 1u); /* the grouping's counting pass (order_count_kernel), fused */
#endif
        if ((tie & ~1u) != 0 || cnt > (u32)CAP) my[0] = 0xFFFEu; /* (bit 0 of m1 ^ m2: the two strands of a tie may differ) */
        /* the two end k-mers' records: window_minimizer's rule on the minima of windows 0 and npos */
        auto resolve = [&](u32 k1, u32 k2, int wbase, int j0, u32 &t, u32 &rev) {
            const int ffirst = wbase + (int)((k1 >> 1) & 0x7Fu) - j0, flast = wbase + 127 - (int)((k2 >> 1) & 0x7Fu) - j0;
            int fsel = ffirst;
            if (ffirst == flast)
                rev = k1 & 1u;
            else {
                rev = kmer_is_rev<false, LONGK>(p, v.S, j0, k);
                fsel = rev ? flast : ffirst;
            }
            t = rev ? (u32)(nf - 1 - fsel) : (u32)fsel;
            return mmer_key(p, v.S, j0 + fsel, m);
        };
        u32 tp, rp, ts, rs;
        const u64 kp = resolve(P1, P2, 0, 0, tp, rp);
        const u64 ks = resolve(sx[0], sx[256], (int)sx[512], npos, ts, rs);
        const u64 bp = kp >> v.bshift, bs = ks >> v.bshift;
#if defined(INDEX_EXP_NOATOMIC) /* timing experiment (results are wrong): the pass without its three counting atomics */
        const u32 sp = 0u, ss = 0u;
#else
        const u32 sp = COUNT ? atomicAdd(&bkt[bp], 1u) : 0u;
        const u32 ss = COUNT ? atomicAdd(&bkt[bs], 1u) : 0u;
#endif
        rec[2 * (i - lo)] = make_ulonglong2((bp << 32) | sp, PAY_MAKE(kp, rid, tp, rp, 0, L));
        rec[2 * (i - lo) + 1] = make_ulonglong2((bs << 32) | ss, PAY_MAKE(ks, rid, ts, rs, 1, L));
    }
    __syncthreads();
#if defined(INDEX_EXP_NORUNS) /* timing experiment (results are wrong): the run lists stay in LDS */
    if (tid != 9999u) return;
#endif
    const u64 r0 = (u64)blockIdx.x * 256u;
    const u64 nr = (hi - lo) - r0 < 256u ? (hi - lo) - r0 : 256u;
    u32 *__restrict__ dst = runs + r0 * (CAP / 2);
    for (u32 x = tid; x < (u32)nr * (CAP / 2); x += 256u) dst[x] = ((const u32 *)s_runs)[x];
}

/* bkt = exclusive scan of the counts: record goes to bkt[bucket] + slot */
__global__ void index_fill_kernel(u64 n2, const ulonglong2 *__restrict__ rec, const u32 *__restrict__ bkt, u64 *__restrict__ ent)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n2; i += (u64)gridDim.x * blockDim.x) {
        const ulonglong2 r = rec[i];
        ent[(u64)bkt[r.x >> 32] + (u32)r.x] = r.y;
    }
}

/* the same fill with the reads taken in the PROCESSING ORDER (reads grouped by read-level minimizer: the reads of one locus back to back).
 * A bucket's records are the end k-mers of the reads that start (or end) within a few bases of each other — reads of ONE group, nine
 * times in ten — so in this order the eight-or-so writes into a bucket's 64 bytes and the reads of its start offset come from
 * neighbouring threads at the same time: one sector written once instead of eight partial writes a pass apart (in id order the fill is
 * 10^8 random 8-byte writes: 7 GB of traffic to place 1.6 GB). The price: the two records of a read are fetched by id (one 32-byte
 * sector) instead of streamed. */
__global__ void index_fill_ordered_kernel(u64 nq, const u64 *__restrict__ order, const ulonglong2 *__restrict__ rec, const u32 *__restrict__ bkt, u64 *__restrict__ ent)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nq; i += (u64)gridDim.x * blockDim.x) {
        const u64 id = ORDER_ID(order[i]);
        const ulonglong2 r0 = rec[2 * id], r1 = rec[2 * id + 1];
        ent[(u64)bkt[r0.x >> 32] + (u32)r0.x] = r0.y;
        ent[(u64)bkt[r1.x >> 32] + (u32)r1.x] = r1.y;
    }
}

/* ================================================================================================================
 * exclusive scan (hand-written, three launches): tile sums -> scan of tile sums -> per-tile scan + offset.
 * InT in {u8,u32}; OutT in {u32,u64}; out may alias in when sizeof(InT)==sizeof(OutT). out[n] = total if write_total.
 * ============================================================================================================== */
template <typename InT>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_tile_sums_kernel(const InT *__restrict__ in, u64 n, u64 *__restrict__ tile_sums)
{
    __shared__ u64 s_part[SCAN_BLOCK / DISCO_WAVE];
    u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 sum = 0;
    for (int i = 0; i < SCAN_ITEMS; i++) {
        u64 idx = base + (u64)i * SCAN_BLOCK + threadIdx.x;
        if (idx < n) sum += (u64)in[idx];
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 t = 0;
        for (int w = 0; w < SCAN_BLOCK / DISCO_WAVE; w++) t += s_part[w];
        tile_sums[blockIdx.x] = t;
    }
}

/* single block: exclusive scan of tile_sums[0..nt) in place; total -> *total */
__global__ void __launch_bounds__(1024) scan_sums_kernel(u64 *__restrict__ tile_sums, u64 nt, u64 *__restrict__ total)
{
    __shared__ u64 s_w[16];
    __shared__ u64 s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (u64 base = 0; base < nt; base += 1024) {
        u64 idx = base + threadIdx.x;
        u64 x = idx < nt ? tile_sums[idx] : 0;
        u64 incl = x;
        for (int o = 1; o < 64; o <<= 1) {
            u64 y = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += y;
        }
        if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
        __syncthreads();
        u64 woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); w++) woff += s_w[w];
        u64 carry = s_carry;
        if (idx < nt) tile_sums[idx] = carry + woff + incl - x;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = s_carry;
}

template <typename InT, typename OutT>
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(const InT *in, u64 n, const u64 *__restrict__ tile_sums, OutT *out)
{
    __shared__ u64 s_w[SCAN_BLOCK / DISCO_WAVE];
    /* thread t owns SCAN_ITEMS consecutive elements so that the tile is scanned in order */
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u64 vals[SCAN_ITEMS];
    u64 sum = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        u64 idx = base + i;
        vals[i] = idx < n ? (u64)in[idx] : 0;
        sum += vals[i];
    }
    u64 incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
        u64 y = __shfl_up(incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += y;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    u64 off = tile_sums[blockIdx.x] + incl - sum;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += s_w[w];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        u64 idx = base + i;
        if (idx < n) out[idx] = (OutT)off;
        off += vals[i];
    }
}

template <typename OutT>
__global__ void scan_write_total_kernel(const u64 *total, OutT *out_n) { *out_n = (OutT)*total; }

/* ================================================================================================================
 * probe — candidate generation (getListOfReads, BG/HashTable.cpp:521-571, for every k-mer window of every query read).
 * One wavefront per read. Per segment of PROBE_SEGW windows:
 *   1. lanes hash the m-mers the segment covers, 2. a range-minimum table (doubling in LDS) gives every window its minimizer
 *   occurrence and canonical strand, 3. the first window of each occurrence leads ONE bucket lookup, 4. the records of all
 *   led buckets are walked 64 at a time (lane = record): a record names the window(s) it can match through its minimizer
 *   offset t, and the window's own (occurrence, strand) must agree. Every surviving (window, record) pair is a candidate:
 *   the exact k-mer compare and the overlap / containment extension happen in verify_kernel. Candidates are written
 *   straight to a wave-private chunk of the global hit buffer (one atomic per PROBE_CHUNK slots).
 * The kernel is latency bound (two dependent random loads per read), so its footprint is kept at 8 waves per SIMD:
 * at most 64 VGPRs and 5 KB of LDS per wave.
 * ============================================================================================================== */
/* arguments the hot loop does not touch live in device memory and are fetched where they are used (chunk allocation, row
 * overflow): as kernel arguments they occupied 14 scalar registers for the whole kernel, which at 8 waves per SIMD
 * (96 SGPRs) made hipcc spill 61 scalars into VGPR lanes — 305 of the kernel's 868 vector instructions were reloads */
struct ProbeRare {
    u64 *bump;      /* bump pointer into hits                         */
    u64 hits_cap;
    u64 *big_list;  /* reads whose row did not fit their chunk        */
    u32 *big_cnt;
    u32 *n_big;
    u32 big_cap;
    u32 slow_cap;
    u64 *ctr;
    u64 *slow_list; /* probe_runs_kernel: reads without a usable run list (ties, too many runs), left to probe_kernel's list pass */
    u32 *n_slow;
};
struct ProbeArgs {
    DiscoView v;
    u64 *hits;      /* global hit buffer                              */
    u64 *row_start; /* [n]                                            */
    u32 *row_cnt;   /* [n]                                            */
    /* the same per-read header once more, indexed by POSITION IN THE PROCESSING ORDER: {row start, candidates | length << 32}.
     * verify_kernel walks the same order, so it loads the headers of a whole chunk with one coalesced load instead of three
     * random 64-byte fetches per read (row_cnt, row_start, len by read id: 13 % of its traffic). */
    ulonglong2 *meta_ord; /* [q_hi - q_lo] */
    const ProbeRare *rare;
    /* processing order of the query range (or null: ascending id): reads grouped by their read-level minimizer, see
     * "processing order" below. Reads of one group contain the same genome m-mer: they look up the same buckets here and fetch the
     * same candidate rows in verify_kernel, which walks the same order (so it also finds the wave's candidate lists one after
     * the other in the hit buffer). */
    const u64 *order;
};

#ifndef PROBE_WAVES_PER_SIMD
#define PROBE_WAVES_PER_SIMD 8
#endif
/* ROW17: k - m = 16, the window minimum comes from DPP row scans (below); that variant keeps more values in registers and
 * is built for one wave per SIMD fewer. Round 2: 8 / 7 waves per SIMD instead of 7 / 6 — the kernel as it stands now fits 73
 * registers without spilling (round 1's did not: 14 spilled registers, 48 instead of 40 ms) and the seventh wave hides more of the
 * bucket and record fetches: 38.9 -> 36.5 ms, A/B in one box (tools/ab_build.py); at 9 / 8 it spills 12 registers (40.4 ms) */
/* PMODE 0: the query range in processing order; 1 (BIG): the reads of big_list, whose rows did not fit their chunk, with rows of
 * exactly the size the first pass counted; 2 (LIST): the reads of slow_list (probe_runs_kernel could not use their run lists), rows in
 * chunks as in mode 0 */
/* LONGCLASS (lists only, two classes of rows): long reads of more than PROBE_ACAP words — walked in global memory (LDSROW = false), and so
 * are the short reads that share the list with them */
template <int PMODE, bool LDSROW, bool ROW17, bool LONGK = false, bool LONGCLASS = false>
__global__ void __launch_bounds__(64, (LONGK || LONGCLASS) ? PROBE_WAVES_PER_SIMD - 2 : (ROW17 ? PROBE_WAVES_PER_SIMD - 1 : PROBE_WAVES_PER_SIMD)) probe_kernel(ProbeArgs a)
{
    static_assert(!LONGCLASS || (PMODE != 0 && !LDSROW), "the long class is only ever met in the lists");
    constexpr bool BIG = PMODE == 1, LISTED = PMODE != 0;
    /* LDSROW: the query read's own row is staged in LDS (S <= PROBE_ACAP, decided by the host), so that every base
     * extract is a broadcast LDS read with a statically known address space instead of a global/flat load */
    __shared__ u32 s_k1[PROBE_SEGP + 32]; /* range-minimum tables over the order hashes of the segment's m-mers (+ slack  */
    __shared__ u32 s_k2[PROBE_SEGP + 32]; /*   for the doubling reads past the last position)                             */
    __shared__ u8 s_strand[PROBE_SEGP];   /* strand of every m-mer position                                              */
    __shared__ u32 s_first[PROBE_SEGP]; /* first window of the segment that chose the occurrence at this position     */
    __shared__ u16 s_wp[PROBE_SEGW];    /* per window: chosen occurrence (position in the segment) | strand << 15      */
    __shared__ u16 s_lead[PROBE_SEGW];  /* windows that lead a bucket lookup                                          */
    __shared__ u32 s_occ_fp[64];        /* current batch of led occurrences: key fingerprint, bucket start,            */
    __shared__ u32 s_occ_start[64];     /*   running record count, position                                            */
    __shared__ u32 s_occ_excl[64];
    __shared__ u16 s_occ_prel[64];
    __shared__ u8 s_mark[64];           /* starts of the non-empty buckets in the concatenated record list             */
    __shared__ u64 s_a[PROBE_ACAP + 2]; /* the query read's own packed row + zero padding for the branch-free extracts */
    const u32 lane = threadIdx.x;
    const int S = a.v.S, k = a.v.k;
    const int m = a.v.m, nf = k - m + 1;
    u64 chunk_base = 0;
    u32 chunk_used = PROBE_CHUNK;
    u32 my_maxrow = 0;
    const u64 n_items = BIG ? (u64)min(*a.rare->n_big, a.rare->big_cap) : (PMODE == 2 ? (u64)min(*a.rare->n_slow, a.rare->slow_cap) : (a.v.q_hi - a.v.q_lo));

    /* the next read's row and length are fetched while the current read is processed (LDSROW implies S <= 64 words) */
    u64 pre_w = 0;
    int pre_len = 0;
    u64 cbeg = 0, cend = 0;
    u64 ord_chunk = 0; /* lane i: the read the chunk's item i stands for (WQ_CHUNK == 64) */
    auto rid = [&](u64 it) { return readlane_u64(ord_chunk, (u32)(it - cbeg)); }; /* packed: ORDER_ID / ORDER_LEN */
    while (wq_grab(a.v.wq, n_items, cbeg, cend)) {
    if (!LISTED) {
        const u64 i = min(cbeg + lane, cend - 1);
        ord_chunk = a.order ? a.order[i] : ORDER_MAKE(a.v.q_lo + i, a.v.len[a.v.q_lo + i]);
    }
    if (!LISTED && LDSROW) {
        const u64 o0 = rid(cbeg);
        const u64 A0 = ORDER_ID(o0);
        pre_len = ORDER_LEN(o0);
        if ((int)lane < S) pre_w = a.v.reads[A0 * S + lane];
    }
    for (u64 it = cbeg; it < cend; it++) {
        const u64 bl = BIG ? a.rare->big_list[it] : (PMODE == 2 ? a.rare->slow_list[it] : 0ull); /* read id | position in the order << 32 */
        const u64 oe = LISTED ? 0ull : rid(it);
        const u64 A = LISTED ? (bl & 0xFFFFFFFFull) : ORDER_ID(oe);
        const u64 opos = LISTED ? (bl >> 32) : it;
        const u64 *ga = a.v.reads + A * S;
        int LA;
        /* two row classes: a long read (only ever met here: the lists) is walked in its full row; its suffix record carries the id of its tail row */
        int Sx = S;
        u64 A2 = A;
        if (LISTED && (LDSROW || LONGCLASS) && a.v.full && (int)a.v.len[A] > DISCO_SHORT_MAX) { /* (the host picks LDSROW when the long class has at most PROBE_ACAP words) */
            const u32 x = a.v.ovf[A];
            Sx = a.v.SL;
            ga = a.v.full + (u64)x * Sx;
            A2 = a.v.n + x;
        }
        __syncthreads();
        if (!LISTED && LDSROW) {
            LA = pre_len;
            if ((int)lane < PROBE_ACAP + 2) s_a[lane] = ((int)lane < S) ? pre_w : 0ull;
            const u64 itn = it + 1;
            if (itn < cend) {
                const u64 on = rid(itn);
                const u64 An = ORDER_ID(on);
                pre_len = ORDER_LEN(on);
                if ((int)lane < S) pre_w = a.v.reads[An * S + lane];
            }
        } else {
            LA = LISTED ? (int)a.v.len[A] : ORDER_LEN(oe);
            if (LDSROW && (int)lane < PROBE_ACAP + 2) s_a[lane] = ((int)lane < Sx) ? ga[lane] : 0ull;
        }
        const int npos = LA - k; /* windows j in [0, npos) : BG/OverlapGraph.cpp:401 (containment), :638 (edges, j >= 1) */
        const u64 *pa = LDSROW ? (const u64 *)s_a : ga;
        u32 nrow = 0;
        u64 *grow = nullptr;
        u32 want = 0; /* slots the row may use at grow */
        if (BIG) {
            u64 base = 0;
            want = a.rare->big_cnt[it];
            if (lane == 0) {
                base = atomicAdd(a.rare->bump, (u64)want);
                atomicMax(&a.rare->ctr[CTR_HITS_NEEDED], base + want);
                a.row_start[A] = base;
            }
            base = __shfl(base, 0);
            if (base + want <= a.rare->hits_cap) grow = a.hits + base;
            else {
                want = 0;
                if (lane == 0) atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
            }
        } else {
            /* candidates go straight to the wave's private chunk of the hit buffer (consecutive lanes, consecutive slots);
             * a row that outgrows what is left of the chunk (at least PROBE_ROWCAP slots) is redone by the BIG pass */
            if (PROBE_CHUNK - chunk_used < PROBE_ROWCAP) {
                u64 base = 0;
                if (lane == 0) {
                    base = atomicAdd(a.rare->bump, (u64)PROBE_CHUNK);
                    atomicMax(&a.rare->ctr[CTR_HITS_NEEDED], base + PROBE_CHUNK);
                    if (base + PROBE_CHUNK > a.rare->hits_cap) {
                        atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
                        base = ~0ull; /* no room: this wave writes nothing any more */
                    }
                }
                chunk_base = __shfl(base, 0);
                chunk_used = 0;
            }
            if (chunk_base != ~0ull) {
                grow = a.hits + chunk_base + chunk_used;
                want = PROBE_CHUNK - chunk_used;
            }
        }

        /* append the candidates flagged by `take` to the row */
        auto push = [&](bool take, u64 pay, u32 rev, int jj) {
            const u64 mm = __ballot(take);
            if (take) {
                const u32 pos = nrow + rank_below(mm);
                if (pos < want) grow[pos] = HIT_MAKE(jj, PAY_ID(pay), PAY_SUFFIX(pay), rev, PAY_LEN(pay));
            }
            nrow += __popcll(mm);
        };

        for (int w0 = 0; w0 < npos; w0 += PROBE_SEGW) { /* segments of PROBE_SEGW windows (one for reads up to 256+k bp) */
            const int nw = min(PROBE_SEGW, npos - w0);
            const int np = nw + nf - 1; /* m-mer positions the segment's windows cover */
            /* every window w: p1 / p2 = leftmost / rightmost position of the smallest order hash among its m-mers -> its
             * minimizer occurrence and canonical strand (window_minimizer's rule, disco_device.h) */
            auto choose = [&](int w, u32 m1, u32 m2) {
                const u32 p1 = m1 & 511u, p2 = 511u - (m2 & 511u);
                u32 rev_w, prel = p1;
                if (p1 == p2)
                    rev_w = s_strand[p1];
                else { /* the smallest hash occurs more than once in the window */
                    rev_w = kmer_is_rev<LDSROW, LONGK>(pa, Sx, w0 + w, k);
                    prel = rev_w ? p2 : p1;
                }
                s_wp[w] = (u16)(prel | (rev_w << 15));
                atomicMin(&s_first[prel], (u32)w);
            };
            __syncthreads();
            if (ROW17) {
                /* k - m = 16 (min-overlap 40): a window is 17 positions = the tail of one 16-lane row plus the head of the next
                 * up to the same lane, so its minimum is min(suffix-min of the row at w, prefix-min of the next row at w + 16):
                 * two 4-step DPP row scans per table and register instead of four doubling passes through LDS.
                 * key1 = hash | position (smallest hash, then LEFTMOST), key2 = hash | 511 - position (then RIGHTMOST). */
                u32 h[3];
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const int q = (int)lane + 64 * r;
                    h[r] = 0xFFFFFFFFu;
                    if (q < np) { /* np <= 144; the third pass is empty for reads up to 150 bp */
                        const u32 o = mmer_order<LDSROW>(pa, Sx, w0 + q, m);
                        s_first[q] = 0xFFFFFFFFu;
                        s_strand[q] = (u8)(o & 1u);
                        h[r] = o & ~0x1FFu;
                    }
                }
                u32 m1[2], m2[2];
                const u32 src = (lane + 16u) & 63u;
#pragma unroll
                for (int t = 0; t < 2; t++) { /* table 1, table 2 */
                    u32 pre[3], suf[2];
#pragma unroll
                    for (int r = 0; r < 3; r++) {
                        const u32 q = lane + 64u * r;
                        const u32 key = h[r] == 0xFFFFFFFFu ? 0xFFFFFFFFu : (h[r] | (t ? 511u - q : q));
                        pre[r] = row_prefix_min(key);
                        if (r < 2) suf[r] = row_suffix_min(key);
                    }
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const u32 ya = (u32)__shfl((int)pre[r], (int)src), yb = (u32)__shfl((int)pre[r + 1], (int)src);
                        const u32 y = lane < 48u ? ya : yb;
                        (t ? m2 : m1)[r] = suf[r] < y ? suf[r] : y;
                    }
                }
                __syncthreads(); /* s_first, s_strand */
#if defined(PROBE_STOP) && PROBE_STOP <= 1 /* timing attribution builds (tools/ab_build.py): hashing + window minima only */
                if (m1[0] + m1[1] + m2[0] + m2[1] == 12345u) nrow++;
                continue;
#endif
#pragma unroll
                for (int r = 0; r < 2; r++) {
                    const int w = (int)lane + 64 * r;
                    if (w < nw) choose(w, m1[r], m2[r]);
                }
            } else {
                /* 1. order hashes of the segment's m-mers; seeds of the two range-minimum tables: key1 = hash | position
                 *    (smallest hash, then LEFTMOST position), key2 = hash | 511 - position (then RIGHTMOST position) */
                for (int q = (int)lane; q < np; q += 64) {
                    const u32 o = mmer_order<LDSROW>(pa, Sx, w0 + q, m);
                    s_first[q] = 0xFFFFFFFFu;
                    s_strand[q] = (u8)(o & 1u);
                    s_k1[q] = (o & ~0x1FFu) | (u32)q;
                    s_k2[q] = (o & ~0x1FFu) | (511u - (u32)q);
                }
                __syncthreads();
                /* 2. minimum over [q, q + P) for every q by doubling, P = largest power of two <= nf; in place, ascending
                 *    passes (a pass reads only positions it has not written: d <= 32 < 64). Entries whose range leaves the
                 *    segment are never used by a window. Then every window w takes min(T[w], T[w + nf - P]). */
                int P = 1;
                for (; 2 * P <= nf; P <<= 1)
                    for (int q0 = 0; q0 < np; q0 += 64) {
                        const int q = min(q0 + (int)lane, np - 1);
                        const u32 a1 = s_k1[q], b1 = s_k1[q + P], a2 = s_k2[q], b2 = s_k2[q + P];
                        __syncthreads();
                        s_k1[q] = a1 < b1 ? a1 : b1;
                        s_k2[q] = a2 < b2 ? a2 : b2;
                    }
                __syncthreads();
                for (int ws = 0; ws < nw; ws += 64) {
                    const int w = ws + (int)lane;
                    if (w < nw) {
                        const int e = nf - P;
                        const u32 a1 = s_k1[w], b1 = s_k1[w + e], a2 = s_k2[w], b2 = s_k2[w + e];
                        choose(w, a1 < b1 ? a1 : b1, a2 < b2 ? a2 : b2);
                    }
                }
            }
            __syncthreads();
#if defined(PROBE_STOP) && PROBE_STOP <= 2 /* ... + choose */
            if (s_wp[lane] == 0xABCDu) nrow++;
            continue;
#endif
            /* 3. the first window of every occurrence leads one bucket lookup */
            u32 nlead = 0;
            for (int ws = 0; ws < nw; ws += 64) {
                const int w = ws + (int)lane;
                const bool isl = (w < nw) && s_first[s_wp[w] & 0x7FFFu] == (u32)w;
                const u64 lm = __ballot(isl);
                if (isl) s_lead[nlead + rank_below(lm)] = (u16)w;
                nlead += __popcll(lm);
            }
            __syncthreads();
#if defined(PROBE_STOP) && PROBE_STOP <= 3 /* ... + lead detection */
            if (s_lead[lane] == 0xABCDu) nrow++;
            continue;
#endif
            for (u32 lb = 0; lb < nlead; lb += 64) {
                const u32 li = lb + lane;
                u32 s = 0, cnt = 0;
                if (li < nlead) {
                    const u32 prel = s_wp[s_lead[li]] & 0x7FFFu;
                    const u64 key = mmer_key<LDSROW>(pa, Sx, w0 + (int)prel, m);
                    const u64 b = key >> a.v.bshift;
                    s = a.v.bkt[b];
                    cnt = a.v.bkt[b + 1] - s;
                    s_occ_fp[lane] = KEY_FP(key);
                    s_occ_prel[lane] = (u16)prel;
                }
                const u32 incl = wave_inclusive_add(cnt);
                const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
                /* only the buckets that hold records keep a slot (compacted): their start offsets in the concatenated record
                 * list are then strictly increasing, and a byte mark at every start turns "which bucket owns record i" into a
                 * ballot and a population count for the first 64 records (a binary search for the rest) */
                const u64 nz = __ballot(cnt > 0);
                const u32 nne = __popcll(nz);
                const u32 fp_o = s_occ_fp[lane];
                const u16 prel_o = s_occ_prel[lane];
                s_mark[lane] = 0;
                __syncthreads();
                if (cnt > 0) {
                    const u32 r = rank_below(nz);
                    s_occ_fp[r] = fp_o;
                    s_occ_prel[r] = prel_o;
                    s_occ_start[r] = s;
                    s_occ_excl[r] = incl - cnt;
                    if (incl - cnt < 64) s_mark[incl - cnt] = 1;
                }
                __syncthreads();
                const u64 startmask = __ballot(s_mark[lane] != 0);
#if defined(PROBE_STOP) && PROBE_STOP <= 4 /* ... + bucket lookups, no record walk */
                if (startmask == 0x123456789ull) nrow++;
                continue;
#endif
                /* 4. all records of all led buckets, lane = record: a record names the window(s) it can match through its
                 *    minimizer offset t; the window's own (occurrence, strand) must agree */
                for (u32 base = 0; base < total; base += 64) {
                    const u32 idx = base + lane;
                    bool match = false;
                    u64 pay = 0;
                    int oprel = 0;
                    if (idx < total) {
                        u32 lo;
                        if (base == 0)
                            lo = __popcll(startmask & (lane_mask_lt() | (1ull << lane))) - 1u;
                        else {
                            lo = 0;
                            u32 hi = nne; /* largest o with excl[o] <= idx */
                            while (hi - lo > 1) {
                                const u32 mid = (lo + hi) >> 1;
                                if (s_occ_excl[mid] <= idx) lo = mid;
                                else hi = mid;
                            }
                        }
                        pay = a.v.ent[s_occ_start[lo] + (idx - s_occ_excl[lo])];
                        oprel = (int)s_occ_prel[lo];
                        match = (PAY_FP(pay) == s_occ_fp[lo]) && (PAY_ID(pay) != A) && (PAY_ID(pay) != A2); /* self excluded: BG/OverlapGraph.cpp:421,655 */
                    }
                    /* a window in canonical-forward orientation starts t before the occurrence, a reversed one k-m-t before */
                    const int t = (int)PAY_T(pay);
                    const int w1 = oprel - t, w2 = oprel - (nf - 1 - t);
                    const bool take1 = match && w1 >= 0 && w1 < nw && s_wp[w1] == (u16)oprel;
                    const bool take2 = match && w2 >= 0 && w2 < nw && s_wp[w2] == (u16)(oprel | 0x8000);
                    /* strand relation query window vs record (0 = same strand) */
                    push(take1, pay, PAY_REV(pay), w0 + w1);
                    push(take2, pay, PAY_REV(pay) ^ 1u, w0 + w2);
                }
                __syncthreads();
            }
        }
        __syncthreads();

        if (nrow > my_maxrow) my_maxrow = nrow;
        if (BIG) {
            if (lane == 0) {
                a.row_cnt[A] = grow ? nrow : 0;
                a.meta_ord[opos] = make_ulonglong2(grow ? (u64)(grow - a.hits) : 0ull, (u64)(grow ? nrow : 0u) | ((u64)LA << 32));
            }
        } else if (grow && nrow > want) {
            if (lane == 0) {
                u32 idx = atomicAdd(a.rare->n_big, 1u);
                if (idx < a.rare->big_cap) {
                    a.rare->big_list[idx] = A | (opos << 32);
                    a.rare->big_cnt[idx] = nrow;
                } else
                    atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
                a.meta_ord[opos] = make_ulonglong2(0ull, (u64)LA << 32); /* the BIG pass fills it in */
            }
        } else {
            if (lane == 0) {
                /* the per-read arrays (a random 64-byte line each) only serve rows beyond the register paths of verify / edge
                 * selection (more than 64 entries); every other consumer reads the header by position in the order. row_cnt is
                 * zeroed by the host before the pass. */
                if (grow && nrow > 64) {
                    a.row_start[A] = chunk_base + chunk_used;
                    a.row_cnt[A] = nrow;
                }
                a.meta_ord[opos] = make_ulonglong2(grow ? chunk_base + chunk_used : 0ull, (u64)(grow ? nrow : 0u) | ((u64)LA << 32));
            }
            if (grow) chunk_used += nrow;
        }
    }
    }
    if (lane == 0) atomicMax(&a.rare->ctr[CTR_MAX_ROW], (u64)my_maxrow);
}

/* ================================================================================================================
 * probe_runs_kernel — candidate generation (the same getListOfReads hit set as probe_kernel) from the minimizer runs that
 * index_runs_kernel left for every read: the kernel starts at the bucket lookups. A run = (first window, end window, occurrence,
 * strand): one bucket lookup per run, and a record of the bucket matches AT MOST ONE window — t before the occurrence when the
 * run's windows are canonical-forward, k - m - t before it when they are reversed — which must lie inside the run.
 * Several reads per wavefront: a read has about 13 runs, so one read per wave would use a fifth of the lanes. A GROUP of
 * G = 64 / LPR reads is processed together (LPR lanes load one read's 2 LPR run entries): the runs of the group are compacted into
 * one occurrence list, looked up 64 at a time, and the records of all their buckets are walked as ONE concatenated list, lane =
 * record (about 400 records per group of 4: 6-7 full passes instead of 4 x 2 half-empty ones). Which occurrence owns record i:
 * a byte mark at the first record of every non-empty bucket, ballot + population count per pass. Occurrences are in (read, run)
 * order and records in occurrence order, so the candidates of one read leave the walk contiguously: every read of the group gets
 * its own row in the wave's private chunk of the hit buffer, in the order verify_kernel reads them.
 * Reads whose run list is marked unusable go to slow_list (probe_kernel<2>), rows that do not fit the chunk to big_list
 * (probe_kernel<1>), as before.
 * ============================================================================================================== */
#define PR_MARKCAP 1024 /* records of one batch of 64 lookups that can be owner-marked in LDS (more: binary search)        */
#define PR_CHUNK 8192   /* hit slots a wave reserves at a time                                                           */
#define PR_RESERVE 1024 /* room a wave makes sure of before it starts a group                                            */

__device__ __forceinline__ u64 shfl_u64(u64 x, u32 l)
{
    return ((u64)(u32)__shfl((int)(u32)(x >> 32), (int)l) << 32) | (u32)__shfl((int)(u32)x, (int)l);
}

#ifndef PR_WAVES_PER_SIMD
#define PR_WAVES_PER_SIMD 8 /* rounds 3-4: 72 vector registers, seven waves, and a cap of 7 or 8 spilled; with the chunk bounds read as scalars (wq_grab, round 5) it needs 65: eight waves without a spill, 16.8 -> 16.5 ms */
#endif
/* runs_lo == ~0: the run lists are stored by POSITION in the processing order (multi-GPU flow, ranks own loci: index_runs_kernel ran
 * over the order itself), not by read id */
template <int LPR>
__global__ void __launch_bounds__(64, PR_WAVES_PER_SIMD) probe_runs_kernel(ProbeArgs a, const u32 *__restrict__ runs, u64 runs_lo)
{
    constexpr int G = 64 / LPR;      /* reads per group                                            */
    constexpr int RS = VERIFY_SW + 1; /* words of a staged row (+ one readable word for the branch-free extract) */
    __shared__ u64 s_rows[G * RS];
    __shared__ u32 s_id[G];
    __shared__ u32 s_occ[128]; /* the group's runs, compacted: OCC_MAKE */
    __shared__ u32 s_o_fp[64], s_o_start[64], s_o_excl[64], s_o_desc[64]; /* current batch of lookups with records */
    __shared__ u8 s_mark[PR_MARKCAP];
#if defined(VERIFY_EXP_HALF)
    __shared__ u32 s_len_exp[G];
#endif
    const u32 lane = threadIdx.x;
    const int k = a.v.k, m = a.v.m, nf = k - m + 1;
    const u32 slot = lane / LPR, e = lane % LPR;
    const u64 le = lane_mask_lt() | (1ull << lane);
    const u64 nq = a.v.q_hi - a.v.q_lo;
    u64 chunk_base = 0;
    u32 chunk_used = PR_CHUNK;
    u32 my_maxrow = 0;
    for (u32 x = lane; x < PR_MARKCAP / 8; x += 64) ((u64 *)s_mark)[x] = 0ull;
    u64 cbeg = 0, cend = 0;
#if defined(WQ_SPLIT_ALL)
    WqSplit wqs;
    while (wq_grab_split(a.v.wq, nq, cbeg, cend, wqs)) {
#else
    while (wq_grab(a.v.wq, nq, cbeg, cend)) {
#endif
        const u64 ci = min(cbeg + lane, cend - 1);
        const u64 ord_chunk = a.order ? a.order[ci] : ORDER_MAKE(a.v.q_lo + ci, a.v.len[a.v.q_lo + ci]);
        if (a.v.full) { /* two classes of rows: the chunk's long reads go to the list pass — ONE counting atomic per chunk (one per read, on the one
                           counter, made this kernel twice as slow with 6 % long reads) */
            const bool lng = cbeg + lane < cend && ORDER_LEN(ord_chunk) > DISCO_SHORT_MAX;
            const u64 lm = __ballot(lng);
            if (lm) {
                u32 base = 0;
                if (lane == 0) base = atomicAdd(a.rare->n_slow, (u32)__popcll(lm));
                base = uniform_u32(base);
                if (lng) {
                    const u32 idx = base + rank_below(lm);
                    if (idx < a.rare->slow_cap) a.rare->slow_list[idx] = ORDER_ID(ord_chunk) | ((cbeg + lane) << 32);
                    else atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
                    a.meta_ord[cbeg + lane] = make_ulonglong2(0ull, (u64)ORDER_LEN(ord_chunk) << 32);
                }
            }
        }
        /* run words and rows of the group that starts at g0 (addresses clamped, loads unconditional: they are issued one group ahead) */
        auto fetch = [&](u64 g0, u32 &rw, u64 &roww) {
            const u64 it = min(g0 + slot, cend - 1);
            rw = runs[(runs_lo == ~0ull ? it : ORDER_ID(shfl_u64(ord_chunk, (u32)(it - cbeg))) - runs_lo) * LPR + e];
            const u32 rl = lane & (G * 8 - 1);
            const u64 itr = min(g0 + (rl >> 3), cend - 1);
            roww = a.v.reads[ORDER_ID(shfl_u64(ord_chunk, (u32)(itr - cbeg))) * VERIFY_SW + (rl & 7)];
        };
        u32 pre_rw;
        u64 pre_row;
        fetch(cbeg, pre_rw, pre_row);
        for (u64 g0 = cbeg; g0 < cend; g0 += G) {
            const u32 ng = (u32)min((u64)G, cend - g0);
            const bool sv = slot < ng; /* my slot holds a read */
            const u64 opos = g0 + (sv ? slot : 0u);
            const u64 oe = shfl_u64(ord_chunk, (u32)(opos - cbeg));
            const u32 A = (u32)ORDER_ID(oe);
            const int LA = ORDER_LEN(oe);
            const u32 npos = (u32)(LA - k);
            const u32 rw = pre_rw;
            __syncthreads();
            if (lane < (u32)(G * 8)) s_rows[(lane >> 3) * RS + (lane & 7)] = pre_row;
            if (e == 0) s_id[slot] = sv ? A : 0xFFFFFFFFu;
#if defined(VERIFY_EXP_HALF)
            if (e == 0) s_len_exp[slot] = (u32)LA;
#endif
            fetch(g0 + G, pre_rw, pre_row);
            if (PR_CHUNK - chunk_used < PR_RESERVE) {
                u64 base = 0;
                if (lane == 0) {
                    base = atomicAdd(a.rare->bump, (u64)PR_CHUNK);
                    atomicMax(&a.rare->ctr[CTR_HITS_NEEDED], base + PR_CHUNK);
                    if (base + PR_CHUNK > a.rare->hits_cap) {
                        atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
                        base = ~0ull; /* no room: this wave writes nothing any more */
                    }
                }
                chunk_base = uniform_u64(base);
                chunk_used = 0;
            }
            u64 *grow = nullptr;
            u32 want = 0;
            if (chunk_base != ~0ull) {
                grow = a.hits + chunk_base + chunk_used;
                want = PR_CHUNK - chunk_used;
            }
            /* 1. the group's runs -> one compacted occurrence list in (read, run) order */
            const u32 e0 = rw & 0xFFFFu, e1 = rw >> 16;
            const u64 slowm = __ballot(sv && e == 0 && e0 == 0xFFFEu);
            const bool myslow = (slowm >> (slot * LPR)) & 1ull;
            const bool v0 = sv && !myslow && e0 < 0xFFFEu, v1 = sv && !myslow && e1 < 0xFFFEu;
            const u32 nx = (u32)__shfl_down((int)e0, 1); /* the entry after e1 */
            const u32 wend0 = v1 ? RUN_W(e1) : npos;
            const u32 wend1 = (e + 1 < (u32)LPR && nx < 0xFFFEu) ? RUN_W(nx) : npos;
            const u64 m0 = __ballot(v0), m1 = __ballot(v1);
            const u32 below = rank_below(m0) + rank_below(m1);
            if (v0) s_occ[below] = OCC_MAKE(slot, e0, wend0);
            if (v1) s_occ[below + 1] = OCC_MAKE(slot, e1, wend1);
            const u32 n_occ = (u32)__popcll(m0) + (u32)__popcll(m1);
            if (sv && e == 0 && myslow && !(a.v.full && LA > DISCO_SHORT_MAX)) { /* ties / too many runs: probe_kernel<2> does this read (long reads: listed with their chunk, above) */
                const u32 idx = atomicAdd(a.rare->n_slow, 1u);
                if (idx < a.rare->slow_cap) a.rare->slow_list[idx] = (u64)A | (opos << 32);
                else atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
                a.meta_ord[opos] = make_ulonglong2(0ull, (u64)LA << 32);
            }
            u32 nrow = 0;
            u32 c_slot[G];
#pragma unroll
            for (int s = 0; s < G; s++) c_slot[s] = 0;
            __syncthreads();
            for (u32 ob = 0; ob < n_occ; ob += 64) {
                /* 2. one bucket lookup per occurrence */
                const u32 oi = ob + lane;
                u32 d = 0, st = 0, cnt = 0, fp = 0;
                if (oi < n_occ) {
                    d = s_occ[oi];
                    const int prel = (int)OCC_FIRST(d) + (int)OCC_DELTA(d);
                    const u64 key = mmer_key<true>(s_rows + OCC_SLOT(d) * RS, RS, prel, m);
                    const u64 b = key >> a.v.bshift;
                    uint2 se;
#if defined(PR_EXP) && PR_EXP == 3 /* timing experiment (results are wrong): the bucket bounds from one line */
                    __builtin_memcpy(&se, a.v.bkt + (b & 7u), sizeof se);
                    se.y = se.x + 3u;
#else
                    __builtin_memcpy(&se, a.v.bkt + b, sizeof se); /* bkt[b], bkt[b + 1] */
#endif
                    st = se.x;
                    cnt = se.y - se.x;
                    fp = KEY_FP(key);
                }
                const u32 incl = wave_inclusive_add(cnt);
                const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
                const u64 nz = __ballot(cnt > 0);
                const u32 nne = (u32)__popcll(nz);
                const u32 excl = incl - cnt;
                __syncthreads();
                if (cnt > 0) { /* only the buckets that hold records keep a slot: their starts in the concatenated list increase strictly */
                    const u32 r = rank_below(nz);
                    s_o_fp[r] = fp;
                    s_o_start[r] = st;
                    s_o_excl[r] = excl;
                    s_o_desc[r] = d;
                    if (excl < PR_MARKCAP) s_mark[excl] = 1;
                }
                __syncthreads();
                /* 3. all records of the batch's buckets, lane = record */
                const bool marks = total <= PR_MARKCAP;
                u32 nbefore = 0;
                for (u32 base = 0; base < total; base += 64) {
                    const u32 idx = base + lane;
                    const bool in = idx < total;
                    u32 lo;
                    if (marks) {
                        const u64 sm = __ballot(in && s_mark[in ? idx : 0u] != 0);
                        lo = nbefore + (u32)__popcll(sm & le) - 1u;
                        nbefore += (u32)__popcll(sm);
                    } else {
                        lo = 0;
                        u32 hi = nne; /* largest o with excl[o] <= idx */
                        while (hi - lo > 1) {
                            const u32 mid = (lo + hi) >> 1;
                            if (s_o_excl[mid] <= idx) lo = mid;
                            else hi = mid;
                        }
                    }
                    if (!in) lo = 0;
                    const u32 dd = s_o_desc[lo];
                    u64 pay = 0;
#if defined(PR_EXP) && PR_EXP == 2 /* timing experiment (results are wrong): the records from ONE line instead of the buckets' */
                    if (in) pay = a.v.ent[lane];
#else
                    if (in) pay = a.v.ent[s_o_start[lo] + (idx - s_o_excl[lo])];
#endif
                    const u32 dslot = OCC_SLOT(dd), rv = OCC_STRAND(dd);
                    const int wst = (int)OCC_FIRST(dd), wen = (int)OCC_END(dd), prel = wst + (int)OCC_DELTA(dd);
                    const int t = (int)PAY_T(pay);
                    /* a window in canonical-forward orientation starts t before the occurrence, a reversed one k - m - t before */
                    const int w = rv ? prel - (nf - 1 - t) : prel - t;
                    bool take = in && PAY_FP(pay) == s_o_fp[lo] && (u32)PAY_ID(pay) != s_id[dslot] /* self: BG/OverlapGraph.cpp:421,655 */
                                && w >= wst && w < wen;
#if defined(VERIFY_EXP_HALF) /* timing experiment (results are wrong): the reference compares a pair once and inserts the twin (BG/OverlapGraph.cpp:614-626,
                                652-653); here every overlap is compared from both sides. What half the overlap-type candidates would leave of probe
                                + verify: those into a read of smaller id are dropped (containment-type candidates stay: verify's MODE 1 rule) */
                    {
                        const int LAx = (int)s_len_exp[dslot], LBx = (int)PAY_LEN(pay);
                        const bool cont = (PAY_SUFFIX(pay) == (PAY_REV(pay) ^ rv)) ? (LAx - w >= LBx) : (w + k - LBx >= 0);
                        take = take && (cont || (u32)PAY_ID(pay) > s_id[dslot]);
                    }
#endif
                    const u64 mm = __ballot(take);
                    if (take) {
                        const u32 pos = nrow + rank_below(mm);
                        if (pos < want) grow[pos] = HIT_MAKE(w, PAY_ID(pay), PAY_SUFFIX(pay), PAY_REV(pay) ^ rv, PAY_LEN(pay));
                    }
                    nrow += (u32)__popcll(mm);
#pragma unroll
                    for (int s = 0; s < G; s++) c_slot[s] += (u32)__popcll(mm & __ballot(dslot == (u32)s));
                }
                if (cnt > 0 && excl < PR_MARKCAP) s_mark[excl] = 0;
            }
            /* rows: the candidates of slot s follow those of the slots before it */
            u32 mystart = 0, mycnt = 0, run = 0;
#pragma unroll
            for (int s = 0; s < G; s++) {
                if (slot == (u32)s) {
                    mystart = run;
                    mycnt = c_slot[s];
                }
                run += c_slot[s];
                my_maxrow = max(my_maxrow, c_slot[s]);
            }
            if (sv && e == 0 && !myslow) {
                if (grow && mystart + mycnt <= want) {
                    const u64 rs = chunk_base + chunk_used + mystart;
                    if (mycnt > 64) { /* rows beyond the register paths of verify / edge selection are found by read id (probe_kernel) */
                        a.row_start[A] = rs;
                        a.row_cnt[A] = mycnt;
                    }
                    a.meta_ord[opos] = make_ulonglong2(rs, (u64)mycnt | ((u64)LA << 32));
                } else {
                    if (grow) { /* does not fit what is left of the chunk: the BIG pass gives it a row of its own */
                        const u32 idx = atomicAdd(a.rare->n_big, 1u);
                        if (idx < a.rare->big_cap) {
                            a.rare->big_list[idx] = (u64)A | (opos << 32);
                            a.rare->big_cnt[idx] = mycnt;
                        } else
                            atomicAdd(&a.rare->ctr[CTR_OVERFLOW], 1ull);
                    }
                    a.meta_ord[opos] = make_ulonglong2(0ull, (u64)LA << 32);
                }
            }
            if (grow) chunk_used = nrow <= want ? chunk_used + nrow : PR_CHUNK;
        }
    }
    if (lane == 0) atomicMax(&a.rare->ctr[CTR_MAX_ROW], (u64)my_maxrow);
}

/* ================================================================================================================
 * verify — checkOverlapForContainedRead (BG/OverlapGraph.cpp:517-554) and checkOverlap (:567-595) on every candidate,
 * both as ONE shifted packed compare over the whole aligned region, preceded by the exact k-mer compare (the index
 * matches on minimizers only, so this is what makes a candidate a hit of getListOfReads).
 * One wavefront per query read, lane = candidate, no LDS: the read's own row is shared by all lanes, each lane fetches one
 * 64-byte candidate row. Containment is resolved with atomicMin on a packed key (closed form of the sequential loop,
 * SURVEY.md §8c-7): super(x) = smallest id A containing x with len(A) > len(x), or equal length and A < x
 * (BG/OverlapGraph.cpp:424,449); the recorded row is the first hit in (j, bucket order).
 * Candidates that are not overlap hits are overwritten with ~0 (dropped later by edge selection).
 * ============================================================================================================== */
/* A read B contained in A at one alignment is found twice in A's row: where B's prefix k-mer lies (the hit that is aligned by its prefix:
 * j = the alignment's offset d) and where its suffix k-mer lies (j = d + LB - k) — unless that window is A's own last one, which the
 * probe does not look up. Both compare the same bases; the containment key orders by (A, j, ...), so the second can never be the
 * minimum: only the first goes to the atomicMin (half the atomics on read sets in which most reads are contained). Exact overlaps
 * only: with substitutions allowed the prefix k-mer may be the one that carries one, and then only the second hit exists. */
#ifndef VERIFY_EXP_ALL_CONTAIN_KEYS
#define CONTAIN_KEY_HIT(prefix_align) (prefix_align)
#else
#define CONTAIN_KEY_HIT(prefix_align) true
#endif
struct VerifyArgs {
    DiscoView v;
    u64 *best;
    u64 *hits;
    const u64 *row_start;
    u32 *row_cnt; /* in: candidates, out: verified overlap hits (row compacted in place) */
    const u64 *order; /* [q_hi - q_lo] processing order (read ids), or null */
    ulonglong2 *meta_ord; /* [q_hi - q_lo] headers by position in the order (written by probe_kernel); out: the count becomes the
                             number of verified overlap hits — edge_select_kernel walks the same order and reads it here */
    const u64 *cbits;     /* MODE 2: contained bitmap (complete: the containment pass has run)                            */
    u32 max_subs;         /* INEXACT: substitutions an aligned region may carry (disco_params.max_substitutions)          */
};


/* 32 bases starting at base position pos >= -32 of a row staged in LDS with a zero word in front and zero words behind */
__device__ __forceinline__ u64 extract32_padded(const u64 *row, int pos)
{
    const int w = pos >> 5, sh = (pos & 31) * 2;
    const u64 a = row[w], b = row[w + 1];
    return (a << sh) | ((b >> 1) >> (63 - sh));
}

/* NW = 0: generic variant (any stride, rows read from global memory). NW = 5 / 8: staged variants for a row stride of
 * VERIFY_SW words whose reads have at most 32*NW bases (decided by the host): only the first NW words of a row are moved.
 * NW = 16 / 24 / 32: staged variants for a row stride of NW words (reads of 257..1024 bases: 2 x 300 bp runs, merged pairs,
 * long amplicons): the rows are too wide to be held in registers one read ahead, they are fetched when their read's turn comes
 * (headers and candidate lists stay pipelined). */
/* MODE 0: everything in one pass (exact kmer_hits counter). MODE 1 / 2: the two-pass form for read sets in which most reads are
 * contained (metagenomes: 3/4 of the reads). Whether a candidate is containment-type follows from (j, lengths, record kind) BEFORE
 * its row is fetched, and a containment-type candidate can never become an edge between two non-contained reads (if its compare
 * holds, one of the two reads is contained). MODE 1 fetches and compares the containment-type candidates only (best[]); the host
 * then fixes the contained flags; MODE 2 skips contained query reads altogether and fetches rows for the overlap-type candidates
 * of the others. Rows fetched: about half instead of all; kmer_hits then counts the compared candidates only (disco_params.flags). */
#ifndef VERIFY_WAVES_PER_SIMD
#define VERIFY_WAVES_PER_SIMD 1 /* no register cap beyond what the compiler chooses */
#endif
/* INEXACT (an extension: the reference writes 0 into the substitutions column, BG/OverlapGraph.cpp:815-816):
 * the aligned region may differ in up to a.max_subs bases — the XOR words are counted instead of OR-ed; the seed k-mer must
 * still match exactly (it is what made the pair a hit of getListOfReads), so its compare runs for every candidate. */
template <int NW, int MODE = 0, bool INEXACT = false>
__global__ void __launch_bounds__(64, VERIFY_WAVES_PER_SIMD) verify_kernel(VerifyArgs a)
{
    /* staged: the candidate rows (row i = candidate i = lane i's: NW words + zero words behind; odd stride; the last zero word of a
     * row is the zero word in front of the next one), the read's own row and its reverse complement live in LDS with a
     * statically known address space; the zero words make the shifted extracts branch free */
    constexpr bool staged = NW != 0;
    constexpr bool PREF = NW != 0 && NW <= VERIFY_SW; /* candidate rows prefetched into registers one read ahead */
    /* PREF: the candidate rows are fetched COOPERATIVELY — four lanes per row, 16 bytes each, 16 rows per load instruction, four
     * instructions for the 64 candidates of a batch: an instruction touches 16 cache lines instead of 64. With one lane per row
     * (16 + 16 + 8 bytes in three instructions) every instruction walked 64 different lines through the vector L1, and that walk, not
     * HBM bytes, instructions or occupancy, was what verify_kernel's time followed (round 3: 27 fewer / 130 fewer vector
     * instructions per read, 78 or 101 GB of traffic, 14 to 24 waves per CU — always 31 ms; every lane on ONE line: 19 ms). The
     * quarters go straight to their place in the LDS staging area, where lane = candidate again. */
    constexpr int NQ = PREF ? 4 : 1;                   /* row quarters carried in the pipeline registers */
    constexpr int SA = (NW > VERIFY_SW ? NW : VERIFY_SW) + 4;
    constexpr int BST = (NW + 3) | 1;
    __shared__ u64 s_b[staged ? 1 + 64 * BST : 1];
    __shared__ u64 s_a[SA];   /* [0] = 0, [1..NW] = the read's own row, zeros behind */
    __shared__ u64 s_arc[SA]; /* same layout: reverse complement of the read, left aligned */
    const u32 lane = threadIdx.x;
    const int S = !staged ? a.v.S : (NW > VERIFY_SW ? NW : VERIFY_SW), k = a.v.k;
    u64 my_khits = 0, my_raw = 0;
    u32 my_big = 0, my_mid = 0; /* rows of more than ES_CAP / 64 verified hits, this wavefront's */
    if (staged) {
        for (u32 i = lane; i < 1 + 64 * BST; i += 64) s_b[i] = 0;
        if (lane < (u32)SA) {
            s_a[lane] = 0;
            s_arc[lane] = 0;
        }
        __syncthreads();
    }

    /* 4-stage software pipeline over the reads of this wave: while read t is verified from registers, the candidate ROWS of
     * read t+1 (and its own row), the candidate list of read t+2 and the row header of read t+3 are in flight, so the random
     * row fetch — the one long latency of this kernel — overlaps the compare work of the previous read. */
    struct Meta {
        u32 c;
        int L;
        u64 rs;
    };
    struct Rows {
        ulonglong2 q[NQ]; /* q[p]: quarter (lane & 3) of the row of candidate 16 p + (lane >> 2) */
        u64 aw;           /* lane < NW: word `lane` of the read's own row */
    };
    u64 cbeg = 0, cend = 0;
    /* every pipelined load is UNCONDITIONAL (clamped address + select): a load under an exec-mask branch makes the number
     * of loads in flight unknown to the compiler, which then waits for (nearly) all of them at the next use and the
     * pipeline collapses into one exposed latency per stage */
    /* the reads of a chunk in processing order: position `it` of the query range holds read order[it] (a.order: reads that
     * share their read-level minimizer are neighbours there, so that the candidate rows one of them fetches are still in the
     * cache for the next ones), or simply read q_lo + it */
    u64 ord_chunk = 0;
    auto rid = [&](u64 it) { return ORDER_ID(readlane_u64(ord_chunk, (u32)((it < cend ? it : cend - 1) - cbeg))); };
    ulonglong2 meta_chunk = make_ulonglong2(0, 0); /* lane i: header of the chunk's read i (probe_kernel's meta_ord) */
    u32 skip_chunk = 0; /* MODE 2, lane i: read i of the chunk is contained (nothing to do for it) */
    auto load_meta = [&](u64 it) {
        Meta mt;
        const bool ok = it < cend;
        const u32 i = (u32)((ok ? it : cend - 1) - cbeg);
        const u64 w = readlane_u64(meta_chunk.y, i);
        mt.rs = readlane_u64(meta_chunk.x, i);
        mt.L = (int)(w >> 32);
        mt.c = ok ? (u32)w : 0u;
        if (MODE == 2 && __builtin_amdgcn_readlane((int)skip_chunk, (int)i)) mt.c = 0u;
        return mt;
    };
    /* is the candidate of this pass? containment-type <=> s2 lies inside A (BG/OverlapGraph.cpp:532,547); an overlap needs j >= 1 */
    auto in_pass = [&](u64 h, int LA) -> bool {
        if (MODE == 0) return true;
        const int j = (int)HIT_J(h), LB = (int)HIT_LEN(h);
        const bool contain = (HIT_SUFFIX(h) == HIT_REV(h)) ? (LA - j >= LB) : (j + k - LB >= 0);
        return MODE == 1 ? contain : (!contain && j >= 1);
    };
    /* idle lanes load something the wave touches anyway (never one fixed address: with every wave of the chip doing that,
     * the line's L2 channel becomes a hot spot) */
    auto load_cands = [&](const Meta &mt, u64 A) {
        const bool ok = lane < mt.c;
        const u64 h = a.hits[ok ? mt.rs + lane : (mt.c ? mt.rs : (A & 0xFFFFull))]; /* the hit buffer has more than 65536 slots */
        return ok ? h : 0ull;
    };
    auto load_rows = [&](const Meta &mt, u64 h, u64 A) {
        Rows r;
        r.q[0] = make_ulonglong2(0, 0);
        r.aw = 0;
        if (staged) {
            const u64 *own = a.v.reads + A * S;
            r.aw = own[lane < (u32)NW ? lane : 0u];
            if (PREF) {
                /* the row behind this lane's candidate (lanes without one: the read's own row, a line the wave holds anyway) */
#if defined(VERIFY_EXP_NOROWS) /* timing experiment (tools/ab_build.py; results are wrong): every lane fetches the read's own row */
                const u32 vid = (u32)A;
#else
                const u32 vid = (lane < mt.c && in_pass(h, mt.L)) ? (u32)HIT_ID(h) : (u32)A;
#endif
#pragma unroll
                for (int p = 0; p < NQ; p++) {
                    const u32 id = (u32)__shfl((int)vid, (int)(16 * p + (lane >> 2)));
                    r.q[p] = ((const ulonglong2 *)(a.v.reads + (u64)id * S))[lane & 3u]; /* (S = 8 words: the four quarters of a row) */
                }
            }
        }
        return r;
    };
    /* the quarters of a prefetched batch to their rows in the staging area (row r at s_b + 1 + r BST, NW words, zeros behind) */
    auto stage_rows = [&](const Rows &r) {
        const u32 qq = lane & 3u;
        if (2 * qq < (u32)NW) {
#pragma unroll
            for (int p = 0; p < NQ; p++) {
                u64 *dst = s_b + 1 + (16 * p + (lane >> 2)) * BST + 2 * qq;
                dst[0] = r.q[p].x;
                dst[1] = r.q[p].y;
            }
        }
        if (NW & 1) s_b[1 + lane * BST + NW] = 0; /* NW odd: the last quarter brought one word too many; the word behind a row stays zero */
    };
    /* wide rows (NW = 16 / 24 / 32 words, fetched when their read's turn comes): the same idea — NW / 2 lanes per row, 16 bytes each,
     * 8 / 5 / 4 rows per load instruction — straight into the staging area. (One lane per row meant NW / 2 instructions of 64 lines
     * each.) cnt: candidates of this batch, h: this lane's, A / LA: the read */
    auto fetch_wide_rows = [&](u32 cnt, u64 h, u64 A, int LA) {
        constexpr int LR = NW > VERIFY_SW ? NW / 2 : 1; /* lanes per row */
        constexpr int RPI = 64 / LR;                    /* rows per instruction */
        constexpr int NI = (64 + RPI - 1) / RPI;        /* instructions per batch */
        const u32 vid = (lane < cnt && in_pass(h, LA)) ? (u32)HIT_ID(h) : (u32)A;
        const u32 sub = lane % (u32)LR, rsel = lane / (u32)LR;
        ulonglong2 q[NI];
#pragma unroll
        for (int p = 0; p < NI; p++) {
            const u32 r = (u32)(p * RPI) + rsel;
            const u32 id = (u32)__shfl((int)vid, (int)(r < 64u ? r : 0u));
            q[p] = ((const ulonglong2 *)(a.v.reads + (u64)id * S))[sub];
        }
#pragma unroll
        for (int p = 0; p < NI; p++) {
            const u32 r = (u32)(p * RPI) + rsel;
            if (rsel < (u32)RPI && r < 64u) {
                u64 *dst = s_b + 1 + r * BST + 2 * sub;
                dst[0] = q[p].x;
                dst[1] = q[p].y;
            }
        }
    };
    while (wq_grab(a.v.wq, a.v.q_hi - a.v.q_lo, cbeg, cend)) {
    {
        const u64 i = cbeg + (lane < cend - cbeg ? lane : 0u); /* WQ_CHUNK <= 64: one lane per read of the chunk */
        ord_chunk = a.order ? a.order[i] : a.v.q_lo + i;
        meta_chunk = a.meta_ord[i];
        if (MODE == 2) {
            const u64 id = ORDER_ID(ord_chunk);
            skip_chunk = (((const u32 *)a.cbits)[id >> 5] >> (id & 31)) & 1u;
        }
    }
    u32 nk_chunk = 0; /* lane i: verified hits of the chunk's read i (0 for reads without candidates) */
    Meta m0 = load_meta(cbeg), m1 = load_meta(cbeg + 1), m2 = load_meta(cbeg + 2);
    u64 h0 = load_cands(m0, rid(cbeg)), h1 = load_cands(m1, rid(cbeg + 1));
    Rows R0 = load_rows(m0, h0, rid(cbeg));

    for (u64 it = cbeg; it < cend; it++) {
        const u64 A = rid(it);
        const Meta m3 = load_meta(it + 3);
        const u64 h2 = load_cands(m2, rid(it + 2));
        const Rows R1 = load_rows(m1, h1, rid(it + 1));
        const u32 c = uniform_u32(m0.c); /* wave uniform: scalar from here on (the loads themselves stay in flight as vectors) */
        if (c != 0) {
            u64 *row = a.hits + uniform_u64(m0.rs);
            const u64 *ga = a.v.reads + A * S;
            const int LA = (int)uniform_u32((u32)m0.L);
            __syncthreads();
            if (staged) {
                if (lane < (u32)NW) s_a[1 + lane] = R0.aw;
                __syncthreads();
                /* word i of revcomp(A) = reverse complement of A[LA - 32(i+1), LA - 32i); what lies beyond the read is masked by
                 * every consumer */
                if (lane < (u32)NW) {
                    const int pos = LA - 32 * ((int)lane + 1);
                    s_arc[1 + lane] = pos > -32 ? rev2_64(~extract32_padded(s_a + 1, pos)) : 0ull;
                }
            }
            u32 nkeep = 0;
            /* one batch of 64 candidates: lane = candidate h with its row words w (staged variants) */
            auto batch = [&](const bool act0, const u64 h) { /* (staged variants: the batch's rows are in the staging area) */
                const bool act = act0 && in_pass(h, LA);
                bool ov = false;
                const int j = (int)HIT_J(h);
                const u64 B = HIT_ID(h);
                const int LB = act ? (int)HIT_LEN(h) : k;
                const u32 suf = HIT_SUFFIX(h), rev = HIT_REV(h);
                const bool prefix_align = (suf == rev); /* types 0,2: prefix of s2 sits at j ; types 1,3: suffix of s2 ends at j+k */
                /* s2 = B or revcomp(B); s2[p] lies under A[p + d]; aligned region in A coordinates [x0, x1) */
                const int d = prefix_align ? j : j + k - LB;
                const int x0 = d > 0 ? d : 0, x1 = min(LA, d + LB);
                bool contain, overlap, hidden = false;
                const int hk = prefix_align ? x1 - k : x0; /* A's end k-mer inside a proper overlap: the seed of the twin find */
                if (prefix_align) {
                    contain = LA - j >= LB;       /* BG/OverlapGraph.cpp:532 */
                    overlap = !contain && j >= 1; /* :579 */
                } else {
                    contain = d >= 0;           /* :547 */
                    overlap = d <= 0 && j >= 1; /* :591 */
                }
                if (staged) {
                    __syncthreads();
                    /* ONE pass of XORs over the aligned region. A reversed candidate is compared as revcomp(A) against B itself
                     * (coordinates y = LA-1-x), so no candidate row is ever reverse-complemented:
                     * T[X] == B[X - dd] for X in [X0, X1), T = A or revcomp(A). */
                    const u64 *T = rev ? s_arc + 1 : s_a + 1;
                    const int X0 = rev ? LA - x1 : x0, X1 = rev ? LA - x0 : x1;
                    const int dd = rev ? LA - LB - d : d;
                    const int w0 = X0 >> 5, nl = act ? ((X1 - 1) >> 5) - w0 : -1;
                    const int p = 32 * w0 - dd; /* >= -31: at most one word in front of the row is touched */
                    const u64 *bp = s_b + 1 + lane * BST + (p >> 5);
                    const int sh = (p & 31) * 2;
                    const u64 firstmask = ~0ull >> (2 * (X0 & 31)), lastmask = ~0ull << (62 - 2 * ((X1 - 1) & 31));
                    auto xor_word = [&](int t, u64 blo, u64 bhi) { /* word t of the region: T word ^ shifted B window, masked */
                        u64 xt = T[w0 + t] ^ ((blo << sh) | ((bhi >> 1) >> (63 - sh)));
                        if (t == 0) xt &= firstmask;
                        if (t == nl) xt &= lastmask;
                        return xt;
                    };
                    u64 blo = 0;
                    u32 nsub = 0;
                    /* The seed k-mer [K0, K1) sits at one END of the aligned region (T coordinates): at its start for types 0 / 3
                     * (prefix_align != rev), at its end for 1 / 2. "The k-mer alone matches" — what makes a candidate a hit of
                     * getListOfReads, counted in kmer_hits — is therefore "the mismatch nearest that end is at least k bases in",
                     * and the first / last non-zero XOR word of the one pass below decides it. (Until round 3 the failing lanes
                     * re-walked their k-mer's words in a loop of their own: 130 vector instructions more whenever a lane of the
                     * batch failed — rare on clean reads, the rule on real ones — for a counter.) The other end of a proper overlap
                     * is this read's end k-mer inside it, the seed of the TWIN find: inexact overlaps read both ends
                     * (HIT_HIDDEN_BIT). The XOR words stay in registers; the scan for the first / last one runs when it is needed. */
                    const bool at_start = prefix_align != (rev != 0);
                    u64 xw[NW > 0 ? NW : 1]; /* exact: the region's XOR words t <= nl (the others are never read: no cost to keep them) */
                    u64 diff = 0;
                    u64 fx = 0, lx = 0; /* first / last non-zero XOR word ... */
                    int ft = 0, lt = 0; /* ... and which word it is */
                    auto track = [&](int t, u64 xt) {
                        const bool nz = xt != 0;
                        if (INEXACT) { /* both ends */
                            if (nz && fx == 0) {
                                fx = xt;
                                ft = t;
                            }
                            if (nz) {
                                lx = xt;
                                lt = t;
                            }
                        } else if (nz && (!at_start || fx == 0)) { /* exact: the seed's end */
                            fx = xt;
                            ft = t;
                        }
                    };
                    if (act) blo = bp[0];
#pragma unroll
                    for (int t = 0; t < NW; t++) {
                        if (!__any(t <= nl)) continue;
                        if (t <= nl) {
                            const u64 bhi = bp[t + 1];
                            const u64 xt = xor_word(t, blo, bhi);
                            if (INEXACT) {
                                nsub += base_mismatches(xt);
                                track(t, xt);
                            } else {
                                diff |= xt;
                                xw[t] = xt;
                            }
                            blo = bhi;
                        }
                    }
                    const bool region_ok = act && (INEXACT ? nsub <= a.max_subs : diff == 0);
                    /* exact overlaps: a failing lane is rare on clean reads (the words are scanned when there is one) and the rule on
                     * real ones; inexact overlaps tracked along the way */
                    if (!INEXACT && __any(act && !region_ok)) {
#pragma unroll
                        for (int t = 0; t < NW; t++)
                            if (t <= nl) track(t, xw[t]);
                    }
                    /* no differing base within k bases of the region's start / end (T coordinates)? */
                    const u64 ex = INEXACT ? lx : fx; /* the word the END side looks at */
                    const int et = INEXACT ? lt : ft;
                    const bool first_clean = fx == 0 || 32 * (w0 + ft) + (__clzll((long long)fx) >> 1) >= X0 + k;
                    const bool last_clean = ex == 0 || 32 * (w0 + et) + ((64 - __ffsll((long long)ex)) >> 1) < X1 - k;
                    const bool kmer_ok = act && (at_start ? first_clean : last_clean);
                    const bool full_ok = region_ok && kmer_ok;
                    if (kmer_ok) my_khits++;
                    if (full_ok) {
                        if (contain && (INEXACT || CONTAIN_KEY_HIT(prefix_align)) && (LA > LB || (LA == LB && A < B))) atomicMin(&a.best[B], CKEY_MAKE(A, j, suf, rev));
                        ov = overlap;
                    }
                    /* does B see this pair from its side? Its window there is A's end k-mer inside the overlap — the region's other
                     * end — and the seed must be exact: one substitution in it hides the pair */
                    if (INEXACT) hidden = ov && !(at_start ? last_clean : first_clean);
                } else if (act) {
                    const u64 *gb = a.v.reads + B * S;
                    if (seg_equal<false>(ga, gb, S, LB, j, prefix_align ? 0 : LB - k, k, rev)) {
                        my_khits++;
                        if (INEXACT ? seg_mismatches<false>(ga, gb, S, LB, x0, x0 - d, x1 - x0, rev) <= a.max_subs
                                    : seg_equal<false>(ga, gb, S, LB, x0, x0 - d, x1 - x0, rev)) {
                            if (contain && (INEXACT || CONTAIN_KEY_HIT(prefix_align)) && (LA > LB || (LA == LB && A < B))) atomicMin(&a.best[B], CKEY_MAKE(A, j, suf, rev));
                            ov = overlap;
                            if (INEXACT && ov) hidden = !seg_equal<false>(ga, gb, S, LB, hk, hk - d, k, rev);
                        }
                    }
                }
                /* compact the verified overlap hits to the front of the row (writes never pass the reads of this iteration) */
                if (MODE != 1) { /* the containment pass leaves the candidate list as it is */
                    const u64 mk = __ballot(ov);
                    if (ov) row[nkeep + rank_below(mk)] = INEXACT && hidden ? h | HIT_HIDDEN_BIT : h;
                    nkeep += __popcll(mk);
                }
                __syncthreads();
            };
            /* the first 64 candidates and their rows were prefetched; longer rows fetch the rest on the spot (kept out of the
             * first batch's code path: a load there would make the compiler drain the whole pipeline) */
            if (PREF) {
                stage_rows(R0);
                batch(lane < c, h0);
            } else { /* wide rows: fetched now (the generic variant compares from global memory) */
                if (staged) fetch_wide_rows(c, h0, A, LA);
                batch(lane < c, h0);
            }
            for (u32 i0 = 64; i0 < c; i0 += 64) {
                const bool act = i0 + lane < c;
                const u64 h = row[act ? i0 + lane : 0];
                if (PREF) { /* the same cooperative fetch, on the spot (100x coverage: two of a read's three batches) */
                    Meta mb;
                    mb.c = c - i0;
                    mb.L = LA;
                    mb.rs = 0;
                    stage_rows(load_rows(mb, h, A));
                } else if (staged)
                    fetch_wide_rows(c - i0, h, A, LA);
                batch(act, act ? h : 0ull);
            }
            if (MODE != 1) {
                if (lane == 0) {
                    if (c > 64) a.row_cnt[A] = nkeep; /* (rows of up to 64 candidates have no entry there: probe_kernel) */
                    my_big += nkeep > ES_CAP ? 1u : 0u; /* (sizes edge selection's big-row list: no counting pass, no host round trip in front of it) */
                    my_mid += nkeep > 64u ? 1u : 0u;    /* (counted per wavefront and added once: an atomic per row on the one address is 12 ns per row — the whole of verify at 100x coverage) */
                    my_raw += nkeep;
                }
                if (lane == (u32)(it - cbeg)) nk_chunk = nkeep;
            }
        }
        m0 = m1;
        m1 = m2;
        m2 = m3;
        h0 = h1;
        h1 = h2;
        R0 = R1;
    }
    /* the counts of the whole chunk in one coalesced store: {row start, verified hits | length << 32} by position in the order */
    if (MODE != 1 && lane < (u32)(cend - cbeg)) a.meta_ord[cbeg + lane].y = (u64)nk_chunk | (meta_chunk.y & 0xFFFFFFFF00000000ull);
    }
    for (int o = 32; o > 0; o >>= 1) my_khits += __shfl_down(my_khits, o);
    if (lane == 0) {
        if (my_khits) atomicAdd(&a.v.ctr[CTR_KMER_HITS], my_khits);
        if (my_raw) atomicAdd(&a.v.ctr[CTR_RAW_HITS], my_raw);
        if (my_big) atomicAdd(&a.v.ctr[CTR_ES_BIG], (u64)my_big);
        if (my_mid) atomicAdd(&a.v.ctr[CTR_ES_MID], (u64)my_mid);
    }
}

/* ================================================================================================================
 * verify_flat_kernel (round 4) — the same checkOverlapForContainedRead / checkOverlap compare (BG/OverlapGraph.cpp:517-595) as
 * verify_kernel, for the 64-byte rows (reads up to 256 bases, exact overlaps), with FULL wavefronts: verify_kernel runs lane =
 * candidate of ONE read, and a read has 44 candidates (150 bp, 30x): 69 % of the lanes of 276 vector instructions per read. Here the
 * candidates of the 64 reads of a work-queue chunk are ONE flat list (exclusive scan of the counts in the chunk's headers), cut into
 * batches of 64 regardless of read boundaries: lane = candidate f of the chunk, its read ("segment") found by a scalar walk over the
 * few boundaries inside the batch. What used to be scalar per read — own row, reverse complement, length, row start — is per lane:
 * the 64 reads' rows and reverse complements are staged in LDS once per chunk (lane = read), a lane picks its segment's by address.
 * Rows of any length are just longer segments (no path of their own). The compare runs on 32-bit words with v_alignbit_b32 (one
 * full-rate instruction per 16 bases instead of three 64-bit shifts and two ORs per 32): LDS rows hold the dwords in SEQUENCE order
 * (a packed u64 has its first 16 bases in the HIGH dword). Verified hits are compacted to the front of their own row as before
 * (segmented rank: ballot prefix minus the prefix at the segment's first lane, plus what the segment kept in earlier batches).
 * ============================================================================================================== */
/* bits [bitpos, bitpos + 32) of the MSB-first bit stream d[0], d[1], ... (bitpos >= -31; d[-1] and d[i + 1] readable) */
__device__ __forceinline__ u32 stream_bits32(const u32 *d, int bitpos)
{
    const int i = (bitpos - 1) >> 5;
    return __builtin_amdgcn_alignbit(d[i], d[i + 1], (u32)(32 * i + 32 - bitpos));
}
/* reverse the order of the 16 2-bit groups of x */
__device__ __forceinline__ u32 rev2_32(u32 x)
{
    const u32 y = __brev(x);
    return ((y >> 1) & 0x55555555u) | ((y & 0x55555555u) << 1);
}
/* a copy the compiler cannot see through (register moves at a place of the program's choosing) */
__device__ __forceinline__ u64 pinned_copy(u64 x)
{
    u32 lo, hi;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(lo), "=&v"(hi) : "v"((u32)x), "v"((u32)(x >> 32)));
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u32 pinned_copy32(u32 x)
{
    u32 y;
    asm volatile("v_mov_b32 %0, %1" : "=v"(y) : "v"(x));
    return y;
}
#ifndef VERIFY_FLAT_WAVES_PER_SIMD
#define VERIFY_FLAT_WAVES_PER_SIMD 4 /* round 5: the kernel's time follows its resident waves (8 / 10 / 12 blocks per CU: 29.7 / 25.3 / 22.1 ms) */
#endif
/* CACHE (round 4): two batches are decided TOGETHER — a candidate whose read is some other lane's candidate among the 128 takes that
 * lane's row instead of fetching it again (reads of one locus come back to back in the processing order and share four candidates in
 * five: 128 consecutive candidates are three reads' worth and name about 60 different reads). Every lane writes its slot number under
 * a hash of its read id and reads the winner back (the slot's id decides); leaders and displaced lanes are compacted into one fetch
 * list, four lanes per row as before, only as many load instructions as the list needs; the loads of pair p + 1 are issued while pair
 * p is compared. Measured bound: with two / four lanes sharing every row verify runs 17.1 / 14.3 ms instead of 23.2. */
template <int NW, int MODE = 0, bool CACHE = false>
__global__ void __launch_bounds__(64, NW == 5 ? VERIFY_FLAT_WAVES_PER_SIMD : 1) verify_flat_kernel(VerifyArgs a)
{
    constexpr int ND = 2 * NW;       /* dwords of a row */
    constexpr int BSTR = ND + 1;     /* candidate row r: dwords [1 + r BSTR, + ND); the dword in front belongs to the row before (never zero, never needed: masked) */
    constexpr int TSTR = 2 * ND + 1; /* segment s: [1 + s TSTR, + ND) the read, [+ ND, + 2 ND) its reverse complement */
    constexpr int S = VERIFY_SW;
#ifndef VF_CHUNK
#define VF_CHUNK 64u
#endif
    constexpr u32 CH = CACHE ? 32u : VF_CHUNK; /* reads of a work-queue chunk (CACHE: half, for the LDS its second staging buffer takes) */
    constexpr int NBUF = CACHE ? 2 : 1; /* staged rows: 64 per batch; CACHE: the 128 slots of a pair of batches */
    __shared__ u32 s_b[NBUF * 64 * BSTR + ND + 4];
    __shared__ u32 s_t[CH * TSTR + ND + 4];
    __shared__ ulonglong2 s_hdr[CH]; /* {row start, first flat index | candidates << 32} */
    __shared__ uint2 s_al[CH];       /* {read id, length} */
    __shared__ u32 s_nk[CH];         /* verified hits of the segment */
    __shared__ u32 s_hid[CACHE ? 128 : 1];  /* CACHE: read id whose row slot s holds; then (s_lid) the fetch list of the pair being decided: read ids */
    __shared__ u8 s_hash[CACHE ? 256 : 1];  /* ... hash of a read id -> the slot that registered it last */
    __shared__ u8 s_lslot[CACHE ? 128 : 1]; /* ... the slots the entries of the fetch list go to */
    /* (round 5: 10 240 bytes with NW = 5 — sixteen blocks per CU. The fetch list takes the place of the registrations it is made from:
     * a block is ONE wavefront, its LDS operations happen in program order, and every lane has read the registrations it needs before
     * the first entry of the list is written) */
    u32 *const s_lid = s_hid;
    const u32 lane = threadIdx.x;
    const int k = a.v.k;
    /* the wavefront's counters as SCALARS (round 6: six vector registers of per-lane accumulators in a kernel that lives at the edge of
     * its 128): k-mer hits by ballot + population count per batch; verified hits and the rows of more than ES_CAP / 64 of them from the
     * chunk's counts (s_nk), once per chunk */
    u64 w_khits = 0, w_raw = 0;
    u32 w_big = 0, w_mid = 0;
    u64 cbeg = 0, cend = 0;
    if (CACHE) {
        s_lid[lane] = 0u;
        s_lid[lane + 64] = 0u;
        __syncthreads();
    }
    /* is the candidate of this pass? (verify_kernel's in_pass) */
    auto in_pass = [&](u64 h, int LA) -> bool {
        if (MODE == 0) return true;
        const int j = (int)HIT_J(h), LB = (int)HIT_LEN(h);
        const bool contain = (HIT_SUFFIX(h) == HIT_REV(h)) ? (LA - j >= LB) : (j + k - LB >= 0);
        return MODE == 1 ? contain : (!contain && j >= 1);
    };
#if defined(VF_GRAB_MULT) /* experiment: VF_GRAB_MULT chunks per atomic of the work queue (its one address serves a returning atomic per 12.3 ns) */
    u64 gnext = 0, gend = 0;
    while (wq_grab_multi<CH, VF_GRAB_MULT>(a.v.wq, a.v.q_hi - a.v.q_lo, cbeg, cend, gnext, gend)) {
#elif defined(WQ_EXP_ONE_QUEUE)
    while (wq_grab<CH>(a.v.wq, a.v.q_hi - a.v.q_lo, cbeg, cend)) {
#else
    WqSplit wqs;
    while (wq_grab_split<CH>(a.v.wq, a.v.q_hi - a.v.q_lo, cbeg, cend, wqs)) {
#endif
        const u32 n = (u32)(cend - cbeg);
        const u64 ci = cbeg + (lane < n ? lane : 0u);
        const u64 ordw = a.order ? a.order[ci] : a.v.q_lo + ci;
        const u32 A = (u32)ORDER_ID(ordw);
        const ulonglong2 meta = a.meta_ord[ci];
        /* the read's own row (lane = read) */
        const ulonglong2 *own = (const ulonglong2 *)(a.v.reads + (u64)A * S);
        ulonglong2 ow[(NW + 1) / 2];
#pragma unroll
        for (int t = 0; t < (NW + 1) / 2; t++) ow[t] = own[t];
        u32 c = lane < n ? (u32)meta.y : 0u;
        const int L = (int)(meta.y >> 32);
        if (MODE == 2) { /* contained reads have nothing left to do */
            if ((((const u32 *)a.cbits)[A >> 5] >> (A & 31)) & 1u) c = 0u;
        }
        const u32 incl = wave_inclusive_add(c);
        const u32 P = incl - c;
        const u32 C = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        __syncthreads();
        u32 *tf = s_t + 1 + (lane < CH ? lane : 0u) * TSTR;
        if (lane < CH) {
            s_hdr[lane] = make_ulonglong2(meta.x, (u64)P | ((u64)c << 32));
            s_al[lane] = make_uint2(A, (u32)L);
            s_nk[lane] = 0u;
#pragma unroll
            for (int t = 0; t < NW; t++) {
                const u64 w = (t & 1) ? ow[t >> 1].y : ow[t >> 1].x;
                tf[2 * t] = (u32)(w >> 32);
                tf[2 * t + 1] = (u32)w;
            }
        }
        __syncthreads();
        /* dword g of revcomp(A) = reverse complement of A[L - 16 (g + 1), L - 16 g); what lies beyond the read is masked by every consumer */
        if (lane < CH) {
#pragma unroll
            for (int g = 0; g < ND; g++) {
                int pos = L - 16 * (g + 1);
                pos = pos > -15 ? pos : -15;
                tf[ND + g] = rev2_32(~stream_bits32(tf, 2 * pos));
            }
        }
        __syncthreads();

#if defined(VERIFY_EXP_NOBATCH) /* timing experiment (results are wrong): the chunk's fixed part alone — grab, headers, own rows, reverse complements */
        const u32 nb = 0;
#else
        const u32 nb = (C + 63u) >> 6;
#endif
        u32 sscan = 0; /* wave uniform (kept in a scalar register: readfirstlane): the last segment that starts at or before the first lane of the next batch to be located */
        /* flat index (clamped to the chunk's last candidate: idle lanes repeat it, a line the wave touches anyway) and segment of this
         * lane in batch b: a scalar walk over the segment starts inside the batch */
        auto locate = [&](u32 b, u32 &seg, u32 &f) {
            const u32 last = C - 1u;
            f = 64u * b + lane;
            f = f < last ? f : last;
            u32 fl = 64u * b + 63u;
            fl = fl < last ? fl : last;
            seg = sscan;
            u32 s = (u32)__builtin_amdgcn_readfirstlane((int)sscan) + 1u;
            while (s < CH) {
                const u32 Ps = (u32)__builtin_amdgcn_readlane((int)P, (int)s);
                if (Ps > fl) break;
                seg = f >= Ps ? s : seg;
                s++;
            }
            sscan = s - 1u;
        };
        auto load_cand = [&](u32 seg, u32 f) -> u64 {
            const ulonglong2 hd = s_hdr[seg];
            return a.hits[hd.x + (u64)(f - (u32)hd.y)];
        };
        /* q[p] of a batch: quarter (lane & 3) of the row of candidate 16 p + (lane >> 2) (CACHE: of entry 16 p + (lane >> 2) of the fetch list) */
        auto row_ptr = [&](u64 h, u32 seg, int p) -> const ulonglong2 * {
            u32 vid = (u32)HIT_ID(h);
#if defined(VERIFY_EXP_NOROWS) /* timing experiment (tools/ab_build.py; results are wrong): every lane fetches its read's own row */
            vid = s_al[seg].x;
#endif
#if defined(VERIFY_EXP_SHAREROWS) /* timing experiment (results are wrong): groups of VERIFY_EXP_SHAREROWS lanes fetch ONE row — what a row
                                     cache with that reuse would leave of the row traffic */
            vid = (u32)__shfl((int)vid, (int)(lane & ~(u32)(VERIFY_EXP_SHAREROWS - 1)));
#endif
            if (MODE != 0) { /* candidates of the other pass: the read's own row instead */
                const uint2 al = s_al[seg];
                if (!in_pass(h, (int)al.y)) vid = al.x;
            }
            const u32 id = (u32)__shfl((int)vid, (int)(16 * p + (lane >> 2)));
            return (const ulonglong2 *)(a.v.reads + (u64)id * S) + (lane & 3u);
        };
        /* one row's quarter into its place: row r of the staging area at 1 + r BSTR */
        auto put_quarter = [&](u32 r, const ulonglong2 q) {
            const u32 qq = lane & 3u;
            u32 *dst = s_b + 1 + r * BSTR + 4 * qq;
            if (2 * qq + 1 < (u32)NW) { /* both words of the quarter belong to the row */
                dst[0] = (u32)(q.x >> 32), dst[1] = (u32)q.x, dst[2] = (u32)(q.y >> 32), dst[3] = (u32)q.y;
            } else if (2 * qq < (u32)NW) { /* NW odd: the row's last word */
                dst[0] = (u32)(q.x >> 32), dst[1] = (u32)q.x;
            }
        };
        auto stage_rows = [&](const ulonglong2 q0, const ulonglong2 q1, const ulonglong2 q2, const ulonglong2 q3) {
            put_quarter(lane >> 2, q0);
            put_quarter(16 + (lane >> 2), q1);
            put_quarter(32 + (lane >> 2), q2);
            put_quarter(48 + (lane >> 2), q3);
        };
        /* CACHE: the rows of a PAIR of batches (candidate hA of segment sgA in batch 2p, hB / sgB in batch 2p + 1): which staged row does a
         * lane's candidate use — its own fetch (slot = lane, 64 + lane) or the fetch of another lane of the pair with the same read?
         * Every lane registers its slot under a hash of its read id; the last one to register a read fetches it, the others take its
         * row; a lane displaced by another read (same hash) fetches for itself. The fetches are listed in s_lid / s_lslot. */
        auto decide_pair = [&](u64 hA, u32 sgA, bool needA, u64 hB, u32 sgB, bool needB, u32 &slotA, u32 &slotB, u32 &nmiss) {
            if (MODE != 0) {
                needA = needA && in_pass(hA, (int)s_al[sgA].y);
                needB = needB && in_pass(hB, (int)s_al[sgB].y);
            }
            const u32 idA = (u32)HIT_ID(hA), idB = (u32)HIT_ID(hB);
            const u32 hsA = (idA * 0x9E3779B1u) >> 24, hsB = (idB * 0x9E3779B1u) >> 24;
            __syncthreads(); /* (s_hash needs no clearing: every entry a lane reads is written below, by itself if by nobody after it) */
            if (needA) {
                s_hid[lane] = idA;
                s_hash[hsA] = (u8)lane;
            }
            if (needB) {
                s_hid[64u + lane] = idB;
                s_hash[hsB] = (u8)(64u + lane);
            }
            __syncthreads();
            slotA = lane;
            slotB = 64u + lane;
            if (needA) {
                const u32 w = s_hash[hsA];
                if (s_hid[w & 127u] == idA) slotA = w & 127u;
            }
            if (needB) {
                const u32 w = s_hash[hsB];
                if (s_hid[w & 127u] == idB) slotB = w & 127u;
            }
            const bool fA = needA && slotA == lane, fB = needB && slotB == 64u + lane; /* this lane fetches */
            const u64 mA = __ballot(fA), mB = __ballot(fB);
            const u32 nA = (u32)__popcll(mA);
            if (fA) {
                const u32 r = (u32)rank_below(mA);
                s_lid[r] = idA;
                s_lslot[r] = (u8)lane;
            }
            if (fB) {
                const u32 r = nA + (u32)rank_below(mB);
                s_lid[r] = idB;
                s_lslot[r] = (u8)(64u + lane);
            }
            nmiss = nA + (u32)__popcll(mB);
            if (!needA) slotA = 0u;
            if (!needB) slotB = 0u;
            __syncthreads();
        };
        /* CACHE: entry 16 p + (lane >> 2) of the fetch list (clamped: lanes beyond it repeat its last row, a line on its way anyway) */
        auto list_ptr = [&](u32 nmiss, int p, u64 &lsl) -> const ulonglong2 * {
            u32 r = 16u * (u32)p + (lane >> 2);
            const u32 last = nmiss ? nmiss - 1u : 0u;
            r = r < last ? r : last;
            lsl |= (u64)s_lslot[r] << (8 * p);
            return (const ulonglong2 *)(a.v.reads + (u64)s_lid[r] * S) + (lane & 3u);
        };
        u32 carry = 0; /* wave uniform: hits the segment that is open at the batch's first lane has kept so far */
        /* batch b: lane's candidate h of segment seg against the row staged in slot myrow */
        auto compute = [&](const u32 b, const u64 h, const u32 seg, const u32 myrow) {
            const bool valid = 64u * b + lane < C;
            const ulonglong2 hd = s_hdr[seg];
            const uint2 al = s_al[seg];
            const u32 Pseg = (u32)hd.y, cseg = (u32)(hd.y >> 32);
            const u32 Aseg = al.x;
            const int LA = (int)al.y;
            const bool act = valid && in_pass(h, LA);
            const int j = (int)HIT_J(h);
            const u32 B = (u32)HIT_ID(h);
            const int LB = (int)HIT_LEN(h);
            const u32 suf = HIT_SUFFIX(h), rev = HIT_REV(h);
            const bool prefix_align = (suf == rev); /* types 0,2: prefix of s2 sits at j ; types 1,3: suffix of s2 ends at j+k */
            /* s2 = B or revcomp(B); s2[p] lies under A[p + d]; aligned region in A coordinates [x0, x1) */
            const int d = prefix_align ? j : j + k - LB;
            const int x0 = d > 0 ? d : 0, x1 = min(LA, d + LB);
            bool contain, overlap;
            if (prefix_align) {
                contain = LA - j >= LB;       /* BG/OverlapGraph.cpp:532 */
                overlap = !contain && j >= 1; /* :579 */
            } else {
                contain = d >= 0;           /* :547 */
                overlap = d <= 0 && j >= 1; /* :591 */
            }
            /* a reversed candidate is compared as revcomp(A) against B itself (coordinates y = LA-1-x):
             * T[X] == B[X - dd] for X in [X0, X1), T = A or revcomp(A) */
            const int X0 = rev ? LA - x1 : x0, X1 = rev ? LA - x0 : x1;
            int dd = rev ? LA - LB - d : d;
            /* two row classes: the suffix record of a long read names its tail row, B[LB - tailb, LB) (a.v.tailb = 0 otherwise) */
            if (a.v.tailb && LB > DISCO_SHORT_MAX && suf) dd += LB - a.v.tailb;
            const int W0 = X0 >> 4, nl = ((X1 - 1) >> 4) - W0; /* first dword of the region in T, index of its last one */
            const int bitpos = 2 * (16 * W0 - dd);             /* >= -30: B's bit under the first bit of T's dword W0 */
            const int i0 = (bitpos - 1) >> 5;
            const u32 sh = (u32)(32 * i0 + 32 - bitpos);
#if defined(VERIFY_EXP_NOCONFLICT) /* timing experiment (tools/ab_build.py; results are wrong): every LDS read of the compare at a lane-private
                                      address — rows of an odd stride, one per lane: no two lanes of a 32-lane group on one bank */
            const u32 *bp = s_b + 1 + lane * BSTR;
            const u32 *tp = s_t + 1 + (lane & (CH - 1u)) * TSTR;
            const int nlx = (int)pinned_copy32(0u); /* (opaque zero: the three reads at [nl] stay reads of their own) */
#else
            const u32 *bp = s_b + 1 + myrow * BSTR + i0;
            const u32 *tp = s_t + 1 + seg * TSTR + (rev ? ND : 0) + W0;
            const int nlx = nl;
#endif
            const u32 fm = ~0u >> (2 * (X0 & 15)), lm = ~0u << (30 - 2 * ((X1 - 1) & 15));
            u32 bd[ND + 1], td[ND];
#pragma unroll
            for (int t = 0; t <= ND; t++) bd[t] = bp[t];
#pragma unroll
            for (int t = 0; t < ND; t++) td[t] = tp[t];
            u32 diff = 0;
            /* dwords in front of the region's last one: whole (the first one under fm) */
#pragma unroll
            for (int t = 0; t < ND - 1; t++) {
                u32 x = __builtin_amdgcn_alignbit(bd[t], bd[t + 1], sh) ^ td[t];
                if (t == 0) x &= fm;
                diff |= t < nl ? x : 0u;
            }
            { /* the last one, wherever it is */
                const u32 x = __builtin_amdgcn_alignbit(bp[nlx], bp[nlx + 1], sh) ^ tp[nlx];
                diff |= x & (nl == 0 ? (lm & fm) : lm);
            }
#if defined(VERIFY_EXP_NOROWS) || defined(VERIFY_EXP_SHAREROWS) || defined(VERIFY_EXP_NOCONFLICT)
            diff = 0;
#endif
            const bool region_ok = act && diff == 0;
            /* "the k-mer alone matches" (kmer_hits, what makes a candidate a hit of getListOfReads): the seed k-mer sits at the
             * START of the region (T coordinates) for types 0 / 3, at its END for 1 / 2: the differing base nearest that end must
             * be at least k bases in. A failing region is rare on clean reads; its lanes walk their dwords again. */
            bool kmer_ok = region_ok;
            if (__any(act && diff != 0)) {
                if (act && diff != 0) {
                    const bool at_start = prefix_align != (rev != 0);
                    int fpos = -1, lpos = -1;
                    for (int t = 0; t <= nl; t++) {
                        u32 x = __builtin_amdgcn_alignbit(bp[t], bp[t + 1], sh) ^ tp[t];
                        if (t == 0) x &= fm;
                        if (t == nl) x &= lm;
                        if (x) {
                            if (fpos < 0) fpos = 16 * (W0 + t) + (__clz((int)x) >> 1);
                            lpos = 16 * (W0 + t) + ((32 - __ffs((int)x)) >> 1);
                        }
                    }
                    kmer_ok = at_start ? fpos >= X0 + k : lpos < X1 - k;
                }
            }
            w_khits += (u32)__popcll(__ballot(kmer_ok));
            bool ov = false;
            if (region_ok) {
                if (contain && CONTAIN_KEY_HIT(prefix_align) && (LA > LB || (LA == LB && Aseg < B))) atomicMin(&a.best[B], CKEY_MAKE(Aseg, j, suf, rev));
                ov = overlap;
            }
            /* compact the verified overlap hits to the front of their segment's row (writes never pass the candidates still to be read) */
            if (MODE != 1) {
                const u64 mk = __ballot(ov);
                const u32 below = __builtin_amdgcn_mbcnt_hi((u32)(mk >> 32), __builtin_amdgcn_mbcnt_lo((u32)mk, 0u));
                const int lo_lane = (int)Pseg - (int)(64u * b); /* the segment's first lane in this batch; < 0: it began earlier */
                const u32 below_seg = (u32)__shfl((int)below, lo_lane > 0 ? lo_lane : 0);
                const u32 rank = (lo_lane < 0 ? carry : 0u) + below - below_seg;
                if (ov) {
                    u64 hw = h;
                    if (a.v.tailb && B >= (u32)a.v.n) hw = (h & ~(0x7FFFFFFFull << 17)) | ((u64)a.v.long_ids[B - (u32)a.v.n] << 17); /* (a tail row: back to the read's id) */
                    a.hits[hd.x + rank] = hw;
                }
                const u32 kept = rank + (ov ? 1u : 0u);
                const bool last = valid && (64u * b + lane - Pseg == cseg - 1u); /* the segment ends on this lane */
                if (last) {
                    s_nk[seg] = kept;
                    if (cseg > 64u) a.row_cnt[Aseg] = kept; /* (rows of up to 64 candidates have no entry there: probe_kernel) */
                }
                carry = (u32)__builtin_amdgcn_readlane((int)(last ? 0u : kept), 63);
            }
        };
        if (nb && !CACHE) {
            /* software pipeline: while batch b is compared, the candidate rows of batch b + 1 and the candidates of batch b + 2 are in
             * flight. Every register that a load targets is free when the load is issued (the rows were staged, the candidates
             * copied on), so nothing that is still in flight is ever copied (a copy waits for its load) */
            u32 sg0, sg1, ftmp;
            locate(0, sg0, ftmp);
            u64 h0 = load_cand(sg0, ftmp);
            locate(1, sg1, ftmp);
            u64 h1 = load_cand(sg1, ftmp);
            ulonglong2 q0 = *row_ptr(h0, sg0, 0), q1 = *row_ptr(h0, sg0, 1), q2 = *row_ptr(h0, sg0, 2), q3 = *row_ptr(h0, sg0, 3);
            for (u32 b = 0; b < nb; b++) {
                __syncthreads();
                stage_rows(q0, q1, q2, q3);
                /* (the copies are opaque to the compiler: left alone it keeps h in h0's register, moves h1 -> h0 and the freshly loaded
                 * value -> h1 at the loop's end, and that last move waits for every load of the iteration) */
                const u64 h = pinned_copy(h0);
                const u32 seg = sg0;
                q0 = *row_ptr(h1, sg1, 0);
                q1 = *row_ptr(h1, sg1, 1);
                q2 = *row_ptr(h1, sg1, 2);
                q3 = *row_ptr(h1, sg1, 3);
                h0 = pinned_copy(h1);
                sg0 = sg1;
                locate(b + 2, sg1, ftmp);
                h1 = load_cand(sg1, ftmp);
                __syncthreads();
                compute(b, h, seg, lane);
            }
        }
        if (nb && CACHE) {
            /* pairs of batches: while pair p is compared, the rows of pair p + 1 (decided at the top of the iteration, right after pair
             * p's rows were staged and its registers became free) and the candidates of pair p + 2 are in flight */
            const u32 npairs = (nb + 1u) >> 1;
            u32 sg0A, sg0B, sg1A, sg1B, ftmp;
            locate(0, sg0A, ftmp);
            u64 h0A = load_cand(sg0A, ftmp);
            locate(1, sg0B, ftmp);
            u64 h0B = load_cand(sg0B, ftmp);
            locate(2, sg1A, ftmp);
            u64 h1A = load_cand(sg1A, ftmp);
            locate(3, sg1B, ftmp);
            u64 h1B = load_cand(sg1B, ftmp);
#ifndef VF_Q
#define VF_Q 4 /* (8 — the whole list of a pair in flight — needs 145 registers: three waves per SIMD; 4: 128, four) */
#endif
            constexpr int Q = NW == 5 ? VF_Q : 8; /* (rows of 256 bases: LDS holds them at ten blocks per CU whatever the registers do) row quarters a lane holds in flight: 16 Q rows of the pair's fetch list; what is beyond them (Q < 8) is fetched when the pair's turn comes */
            ulonglong2 q[Q];
#pragma unroll
            for (int p = 0; p < Q; p++) q[p] = make_ulonglong2(0, 0);
            u32 s0A = 0, s0B = 0, nm0 = 0;
            u64 lsl0 = 0;
            auto load_list = [&](u32 nmiss, u64 &lsl) { /* sixteen rows per load instruction, as many instructions as the list needs (wave uniform) */
                lsl = 0;
                q[0] = *list_ptr(nmiss, 0, lsl);
#pragma unroll
                for (int p = 1; p < Q; p++)
                    if (nmiss > 16u * (u32)p) q[p] = *list_ptr(nmiss, p, lsl);
            };
            decide_pair(h0A, sg0A, lane < C, h0B, sg0B, 64u + lane < C, s0A, s0B, nm0);
            load_list(nm0, lsl0);
            for (u32 pp = 0; pp < npairs; pp++) {
                const u32 b = 2u * pp;
                __syncthreads();
                { /* the fetched rows of this pair to their slots */
                    const u32 r = lane >> 2;
#pragma unroll
                    for (int p = 0; p < Q; p++)
                        if (16u * (u32)p + r < nm0) put_quarter((u32)(lsl0 >> (8 * p)) & 0xFFu, q[p]);
                    if (Q < 8 && nm0 > 16u * (u32)Q) { /* (wave uniform) the end of a long list: fetched now, in registers that are free again */
                        u64 lsl2 = 0;
#pragma unroll
                        for (int p = Q; p < 8; p++)
                            if (nm0 > 16u * (u32)p) q[p - Q] = *list_ptr(nm0, p, lsl2);
#pragma unroll
                        for (int p = Q; p < 8; p++)
                            if (16u * (u32)p + r < nm0) put_quarter((u32)(lsl2 >> (8 * p)) & 0xFFu, q[p - Q]);
                    }
                }
                const u64 hA = pinned_copy(h0A), hB = pinned_copy(h0B);
                const u32 segA = sg0A, segB = sg0B, rowA = s0A, rowB = s0B;
                h0A = pinned_copy(h1A);
                h0B = pinned_copy(h1B);
                sg0A = sg1A;
                sg0B = sg1B;
                __syncthreads(); /* (the staging stores are done before the next pair's slots are handed out: s_lslot / s_lid are rewritten) */
                decide_pair(h0A, sg0A, 64u * (b + 2u) + lane < C, h0B, sg0B, 64u * (b + 3u) + lane < C, s0A, s0B, nm0);
                load_list(nm0, lsl0);
                locate(b + 4, sg1A, ftmp);
                h1A = load_cand(sg1A, ftmp);
                locate(b + 5, sg1B, ftmp);
                h1B = load_cand(sg1B, ftmp);
                __syncthreads();
#if defined(VERIFY_EXP_NOCOMPUTE) /* timing experiment (results are wrong): candidates, fetch decisions, row fetches and staging without the compare */
                w_khits += (u32)__popcll(__ballot(((hA ^ hB ^ segA ^ segB ^ rowA ^ rowB) & 1ull) != 0));
#else
                compute(b, hA, segA, rowA);
                if (b + 1u < nb) compute(b + 1u, hB, segB, rowB);
#endif
            }
        }
        __syncthreads();
        /* the counts of the whole chunk in one coalesced store: {row start, verified hits | length << 32} by position in the order */
        if (MODE != 1) {
            const u32 nk = lane < n ? s_nk[lane] : 0u; /* (n <= CH) */
            if (lane < n) a.meta_ord[cbeg + lane].y = (u64)nk | (meta.y & 0xFFFFFFFF00000000ull);
            w_raw += (u32)__builtin_amdgcn_readlane((int)wave_inclusive_add(nk), 63);
            w_big += (u32)__popcll(__ballot(nk > ES_CAP)); /* (sizes edge selection's big-row list: no counting pass in front of it; added once per wavefront) */
            w_mid += (u32)__popcll(__ballot(nk > 64u));
        }
    }
    if (lane == 0) {
        if (w_khits) atomicAdd(&a.v.ctr[CTR_KMER_HITS], w_khits);
        if (w_raw) atomicAdd(&a.v.ctr[CTR_RAW_HITS], w_raw);
        if (w_big) atomicAdd(&a.v.ctr[CTR_ES_BIG], (u64)w_big);
        if (w_mid) atomicAdd(&a.v.ctr[CTR_ES_MID], (u64)w_mid);
    }
}

/* ================================================================================================================
 * two classes of rows (DiscoView: full / ovf / long_ids / tailb). The reference has no stride: a read is as long as it is
 * (BG/HashTable.cpp:456-477 packs every read at its own length). A table with one stride pays for its longest read in every row —
 * 0.1 % reads of 600 bases made every row 192 bytes and took the staged kernels away from the other 99.9 %. So: reads of more than
 * 256 bases ("long", at most one in five; otherwise the table keeps one stride) leave the 64-byte table
 * except for their two ENDS, which is all that a short read can overlap them with: the head stays in the read's own row, the tail
 * becomes row n + j, and the suffix record of the index names that row. The short class then runs the kernels of a pure short set
 * unchanged (a tail row is a row like any other until a verified hit is written down: its id becomes the read's again). The long
 * reads themselves — as query reads — take the wave-per-read paths that exist for rare rows anyway (the index's generic count pass, the
 * probe's list pass, verify_long_kernel below, edge selection's sequential rows), over their full rows.
 * ============================================================================================================== */
/* ovf[i] = 1 for a long read (the host scans it in place); the longest read of the short class */
__global__ void class_flag_kernel(const u16 *__restrict__ len, u64 n, u32 *__restrict__ ovf, u64 *ctr)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 mx = 0;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) {
        const u32 L = len[i];
        ovf[i] = L > (u32)DISCO_SHORT_MAX ? 1u : 0u;
        if (L <= (u32)DISCO_SHORT_MAX) mx = max(mx, L);
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (u32)__shfl_down(mx, o));
    if ((threadIdx.x & 63) == 0 && mx) atomicMax(&ctr[CTR_SHORT_MAX], (u64)mx);
}

/* ---- host reads into the table (disco_upload_reads / disco_upload_reads_ragged): a chunk of reads [lo, hi) lies in `src` either at one
 * stride (src_stride words per read) or back to back (read i at word woff[i] - wbase, ceil(len / 32) words: the reference's own form,
 * BG/HashTable.cpp:456-477) and goes to rows of S words; cap = bases a row takes (two classes of rows: 256, the head of a long read) */
__global__ void words_per_read_kernel(const u16 *__restrict__ len, u64 n, u32 *__restrict__ nw)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) nw[i] = ((u32)len[i] + 31u) >> 5;
}

__global__ void unpack_reads_kernel(const u64 *__restrict__ src, int src_stride, const u64 *__restrict__ woff, u64 wbase, const u16 *__restrict__ len, u64 lo, u64 hi,
                                   int S, u32 cap, u64 *__restrict__ rows)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = (hi - lo) * (u64)S;
    for (; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 i = lo + t / (u64)S;
        const u32 w = (u32)(t % (u64)S);
        const u32 L = min((u32)len[i], cap);
        const u64 *p = src + (src_stride ? (i - lo) * (u64)src_stride : woff[i] - wbase);
        u64 x = 0;
        if (32u * w < L) {
            x = p[w];
            if (L - 32u * w < 32u) x &= ~0ull << (2 * (32 - (L - 32u * w))); /* (bases behind the read, or behind the head of a long one) */
        }
        rows[i * (u64)S + w] = x;
    }
}

/* two classes of rows: the long reads among [lo, hi) — full rows full[j][SL] and tail rows rows8[n + j]; one thread per word */
__global__ void unpack_long_reads_kernel(const u64 *__restrict__ src, int src_stride, const u64 *__restrict__ woff, u64 wbase, const u16 *__restrict__ len, u64 lo, u64 hi,
                                        const u32 *__restrict__ long_ids, u64 n_long, u64 n, int SL, int tailb, u64 *__restrict__ full, u64 *__restrict__ rows8)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 per = (u64)SL + 8u, total = n_long * per;
    for (; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 j = t / per;
        const u32 w = (u32)(t % per);
        const u64 i = long_ids[j];
        if (i < lo || i >= hi) continue;
        const int L = len[i], nw = (L + 31) >> 5;
        const u64 *p = src + (src_stride ? (i - lo) * (u64)src_stride : woff[i] - wbase);
        if (w < (u32)SL) {
            u64 x = (int)w < nw ? p[w] : 0ull;
            if ((int)w == nw - 1 && (L & 31)) x &= ~0ull << (2 * (32 - (L & 31)));
            full[j * (u64)SL + w] = x;
        } else {
            const int tw = (int)w - SL;
            rows8[(n + j) * 8 + tw] = 32 * tw < tailb ? extract32<false>(p, nw, L - tailb + 32 * tw) : 0ull;
        }
    }
}

/* long_ids[j] = the j-th long read (the input stage: the rows are packed per class from the text) */
__global__ void class_ids_kernel(const u16 *__restrict__ len, const u32 *__restrict__ ovf, u64 n, u32 *__restrict__ long_ids)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (len[i] > DISCO_SHORT_MAX) long_ids[ovf[i]] = (u32)i;
}

/* one stride -> two classes: rows8 [n + n_long][8], full [n_long][S] (S = the stride of the table that is given up) */
__global__ void class_split_kernel(const u64 *__restrict__ old, int S, const u16 *__restrict__ len, const u32 *__restrict__ ovf, u64 n, int tailb,
                                   u64 *__restrict__ rows8, u64 *__restrict__ full, u32 *__restrict__ long_ids)
{
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; gid < n * 8; gid += (u64)gridDim.x * blockDim.x) {
        const u64 i = gid >> 3;
        const int w = (int)(gid & 7);
        const u64 *row = old + i * (u64)S;
        rows8[gid] = row[w];
        const int L = len[i];
        if (L > DISCO_SHORT_MAX) {
            const u64 j = ovf[i];
            rows8[(n + j) * 8 + w] = 32 * w < tailb ? extract32<false>(row, S, L - tailb + 32 * w) : 0ull;
            for (int x = w; x < S; x += 8) full[j * (u64)S + x] = row[x];
            if (w == 0) long_ids[j] = (u32)i;
        }
    }
}

/* and back (disco_download_reads): out [n][S] */
__global__ void class_join_kernel(const u64 *__restrict__ rows8, const u64 *__restrict__ full, int S, const u16 *__restrict__ len, const u32 *__restrict__ ovf, u64 n,
                                  u64 *__restrict__ out)
{
    u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; gid < n * (u64)S; gid += (u64)gridDim.x * blockDim.x) {
        const u64 i = gid / (u64)S;
        const int x = (int)(gid % (u64)S);
        out[gid] = len[i] > DISCO_SHORT_MAX ? full[(u64)ovf[i] * S + x] : (x < 8 ? rows8[i * 8 + x] : 0ull);
    }
}

/* the candidate rows of the long reads leave the flat pass: their headers {row start, candidates | length << 32} move to a list — with
 * their position in the order, their id and their number among the long reads, so that verify_long_kernel has nothing to chase — and
 * the flat pass finds rows of no candidates there. A block takes TAKE_SPAN consecutive positions and appends its finds with ONE
 * counting atomic: the list keeps the processing order in long stretches (long reads of one locus are neighbours there; in id order
 * verify_long_kernel ran 81 instead of 63 ms at 6 % long reads), and the counter is no hot spot (one atomic per wavefront trip: 6.5 ms). */
#define TAKE_SPAN 4096
__global__ void __launch_bounds__(256) class_take_rows_kernel(ulonglong2 *__restrict__ meta_ord, const u64 *__restrict__ order, u64 q_lo, u64 nq, const u32 *__restrict__ ovf,
                                                              u32 *__restrict__ lpos, ulonglong2 *__restrict__ lmeta, uint2 *__restrict__ linfo, u32 *n_list, u32 cap)
{
    __shared__ u32 s_w[4];
    __shared__ u32 s_base;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    for (u64 span = (u64)blockIdx.x * TAKE_SPAN; span < nq; span += (u64)gridDim.x * TAKE_SPAN) {
        const u64 end = span + TAKE_SPAN < nq ? span + TAKE_SPAN : nq;
        auto mine = [&](u64 ci, ulonglong2 &m) {
            m = ci < end ? meta_ord[ci] : make_ulonglong2(0, 0);
            return ci < end && (u32)(m.y >> 32) > (u32)DISCO_SHORT_MAX && (u32)m.y != 0u;
        };
        u32 cnt = 0; /* pass 1: the span's finds */
        for (u64 ci = span + tid; ci - tid < end; ci += 256) {
            ulonglong2 m;
            cnt += mine(ci, m) ? 1u : 0u;
        }
        cnt = wave_inclusive_add(cnt);
        __syncthreads();
        if (lane == 63) s_w[wv] = cnt;
        __syncthreads();
        const u32 total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (total == 0) continue; /* (block uniform) */
        if (tid == 0) s_base = atomicAdd(n_list, total);
        __syncthreads();
        u32 at = s_base; /* pass 2: in position order */
        for (u64 ci = span + tid; ci - tid < end; ci += 256) {
            ulonglong2 m;
            const bool take = mine(ci, m);
            const u64 tm = __ballot(take);
            __syncthreads();
            if (lane == 0) s_w[wv] = (u32)__popcll(tm);
            __syncthreads();
            u32 off = (u32)rank_below(tm);
            for (u32 w = 0; w < wv; w++) off += s_w[w];
            if (take) {
                const u32 x = at + off;
                const u64 A = ORDER_ID(order ? order[ci] : q_lo + ci);
                if (x < cap) { /* (cap = the number of long reads: never short) */
                    lpos[x] = (u32)ci;
                    lmeta[x] = m;
                    linfo[x] = make_uint2((u32)A, ovf[A]);
                }
                meta_ord[ci].y = m.y & 0xFFFFFFFF00000000ull;
            }
            at += s_w[0] + s_w[1] + s_w[2] + s_w[3];
        }
    }
}

/* verify for the rows class_take_rows_kernel set aside: one wavefront per long read, lane = candidate. The long read (up to VL_ACAP
 * words: 4096 bases) and its reverse complement are staged in LDS once per read; a candidate of the short class brings its 64-byte row
 * (four independent 16-byte loads per lane, into the lane's own staging row) and the aligned region — at most 256 bases, nine words —
 * is ONE pass of XORs over LDS words, verify_kernel's staged compare with the two reads at their own strides. A first version compared
 * every pair where it lay (seg_equal2 on global memory: a chain of up to fourteen dependent loads per candidate, 25 us per batch; with
 * 1 % long reads that was +14 ms of verify, with 6 % +75): kept for long candidates of long reads and for reads beyond VL_ACAP.
 * Candidate ids of tail rows become read ids first; the row is compacted in place as everywhere. */
#define VL_ACAP 128
__global__ void __launch_bounds__(64) verify_long_kernel(VerifyArgs a, const u32 *__restrict__ lpos, const ulonglong2 *__restrict__ lmeta, const uint2 *__restrict__ linfo,
                                                          const u32 *__restrict__ n_list)
{
    constexpr int BST = (VERIFY_SW + 3) | 1; /* a staged candidate row: 8 words + zero words behind (the last one is the zero word in front of the next row) */
    __shared__ u64 s_a[VL_ACAP + 12];   /* [0] = 0, [1 ..] = the long read, zeros behind */
    __shared__ u64 s_arc[VL_ACAP + 12]; /* same layout: its reverse complement, left aligned */
    __shared__ u64 s_b[1 + 64 * BST];
    __shared__ u64 s_def[64]; /* LONG candidates of the read, set aside: compared lane-dense, a batch of them at a time (below) */
    const u32 lane = threadIdx.x;
    const int k = a.v.k;
    u64 my_khits = 0, my_raw = 0;
    u32 my_big = 0, my_mid = 0; /* rows of more than ES_CAP / 64 verified hits, this wavefront's */
    const u32 nl = *n_list;
    for (u32 i = lane; i < 1 + 64 * BST; i += 64) s_b[i] = 0;
    /* the entries are taken from the work queue four at a time: rows differ in length, and a static deal left the last wavefronts
     * running alone (62 ms with 16 waves per CU, 51 with 256 per CU queued up behind each other: the same thing said with launches) */
    /* The headers of a grab's four entries arrive together (lane = entry), and while an entry is compared the NEXT one's read (two words
     * per lane) and first two batches of candidates are already on their way: an entry otherwise starts with three round trips in a row —
     * header, read, candidates — each a translation miss in a buffer of tens of GB (the kernel's time is those round trips: see DESIGN.md). */
    constexpr int VW = (VL_ACAP + 12 + 63) / 64; /* words of the staging area per lane */
    const int SA = a.v.SL;
    u64 qb = 0, qe = 0;
    while (wq_grab<4>(a.v.wq, (u64)nl, qb, qe)) {
    const u32 ne = (u32)(qe - qb);
    const u32 el = lane < ne ? lane : ne - 1u;
    const ulonglong2 meta_l = lmeta[qb + el];
    const u32 lpos_l = lpos[qb + el];
    const uint2 linfo_l = linfo[qb + el];
    u64 pre_aw[VW], pre_h0 = 0, pre_h1 = 0;
    auto prefetch = [&](u32 e) { /* (every address comes from the headers: nothing to chase; clamped, unconditional) */
        const u64 rs = readlane_u64(meta_l.x, e), my = readlane_u64(meta_l.y, e);
        const u32 cc = (u32)my;
        const int aw = ((int)(my >> 32) + 31) >> 5;
        const u64 *g = a.v.full + (u64)(u32)__builtin_amdgcn_readlane((int)linfo_l.y, (int)e) * SA;
#pragma unroll
        for (int t = 0; t < VW; t++) {
            const int w = (int)lane + 64 * t; /* staging word w holds word w - 1 of the read */
            pre_aw[t] = g[(w >= 1 && w <= aw) ? w - 1 : 0];
        }
        const u64 *r = a.hits + rs;
        pre_h0 = r[lane < cc ? lane : cc - 1u];
        pre_h1 = r[64u + lane < cc ? 64u + lane : cc - 1u];
    };
    prefetch(0);
    for (u32 e = 0; e < ne; e++) { /* entry qb + e of the list: header, position, id and number of the long read */
        const ulonglong2 meta = make_ulonglong2(readlane_u64(meta_l.x, e), readlane_u64(meta_l.y, e));
        const u32 c = (u32)meta.y;
        const u32 ci = (u32)__builtin_amdgcn_readlane((int)lpos_l, (int)e);
        const u64 A = (u32)__builtin_amdgcn_readlane((int)linfo_l.x, (int)e);
        const int LA = (int)(meta.y >> 32);
        const u64 *ga = a.v.full + (u64)(u32)__builtin_amdgcn_readlane((int)linfo_l.y, (int)e) * SA;
        const int AW = (LA + 31) >> 5;
        const bool a_lds = AW <= VL_ACAP;
        u64 aw_now[VW];
#pragma unroll
        for (int t = 0; t < VW; t++) aw_now[t] = pre_aw[t];
        u64 h0 = lane < c ? pre_h0 : 0ull, h1 = 64u + lane < c ? pre_h1 : 0ull;
        if (e + 1u < ne) prefetch(e + 1u);
        __syncthreads();
        if (a_lds) {
#pragma unroll
            for (int t = 0; t < VW; t++) {
                const int w = (int)lane + 64 * t;
                if (w < VL_ACAP + 12) s_a[w] = (w >= 1 && w <= AW) ? aw_now[t] : 0ull;
            }
            __syncthreads();
            /* word i of revcomp(A) = reverse complement of A[LA - 32(i+1), LA - 32i); what lies beyond the read is masked by every consumer */
            for (int w = (int)lane; w < VL_ACAP + 12; w += 64) {
                const int pos = LA - 32 * w; /* word w - 1 */
                s_arc[w] = (w >= 1 && w <= AW && pos > -32) ? rev2_64(~extract32_padded(s_a + 1, pos)) : 0ull;
            }
            __syncthreads();
        }
        u64 *row = a.hits + meta.x;
        u32 nkeep = 0;
        /* three stages, all loads unconditional (clamped): while batch b is compared, the rows of batch b + 1 and the candidates of batch
         * b + 2 are in flight (the compaction writes never reach a slot that has not been read: they stay below the batch's own slots) */
        auto load_cand = [&](u32 i0) -> u64 {
            const u32 i = i0 + lane;
            const u64 x = row[i < c ? i : c - 1u];
            return i < c ? x : 0ull;
        };
        struct Rows {
            ulonglong2 q0, q1, q2, q3;
        };
        auto load_rows = [&](u64 hx) { /* a short candidate's 64-byte row (others: the row of read 0, a line like any other) */
            const bool use = hx != 0ull && a_lds && HIT_LEN(hx) <= (u32)DISCO_SHORT_MAX;
            const ulonglong2 *q = (const ulonglong2 *)(a.v.reads + (use ? HIT_ID(hx) : 0ull) * VERIFY_SW);
            Rows r;
            r.q0 = q[0], r.q1 = q[1], r.q2 = q[2], r.q3 = q[3];
            return r;
        };
        /* A long candidate of a long read has no 64-byte row; its region can be as long as the shorter of the two. One such lane in a batch
         * used to send the whole wavefront through the word-by-word compare on global memory (seg_equal2: some forty dependent loads) —
         * with 6 % long reads 98 % of the batches had one, and that, not the short candidates, was the kernel's time. They are set aside
         * (s_def) and compared together when 64 have gathered or the row ends: T from the staged read, B's words straight from its full
         * row, four loads in flight per step, no early exit. */
        u32 ndef = 0;
        auto flush = [&]() {
            __syncthreads();
            const bool actd = lane < ndef;
            const u64 h = actd ? s_def[lane] : 0ull;
            bool ov = false;
            if (actd) {
                const int j = (int)HIT_J(h), LB = (int)HIT_LEN(h);
                const u64 B = HIT_ID(h);
                const u32 suf = HIT_SUFFIX(h), rev = HIT_REV(h);
                const bool prefix_align = (suf == rev);
                const int d = prefix_align ? j : j + k - LB;
                const int x0 = d > 0 ? d : 0, x1 = min(LA, d + LB);
                bool contain, overlap;
                if (prefix_align) {
                    contain = LA - j >= LB;       /* BG/OverlapGraph.cpp:532 */
                    overlap = !contain && j >= 1; /* :579 */
                } else {
                    contain = d >= 0;           /* :547 */
                    overlap = d <= 0 && j >= 1; /* :591 */
                }
                const u64 *T = rev ? s_arc + 1 : s_a + 1;
                const int X0 = rev ? LA - x1 : x0, X1 = rev ? LA - x0 : x1;
                const int dd = rev ? LA - LB - d : d;
                const int w0 = X0 >> 5, nlw = ((X1 - 1) >> 5) - w0;
                const int p = 32 * w0 - dd; /* >= -31 */
                const int bw = p >> 5, sh = (p & 31) * 2, BW = (LB + 31) >> 5;
                const u64 *gb = a.v.full + (u64)a.v.ovf[B] * SA;
                auto Bw = [&](int i) { /* word i of B, zero outside (unconditional load) */
                    const bool ok = i >= 0 && i < BW;
                    const u64 x = gb[ok ? i : 0];
                    return ok ? x : 0ull;
                };
                const u64 firstmask = ~0ull >> (2 * (X0 & 31)), lastmask = ~0ull << (62 - 2 * ((X1 - 1) & 31));
                u64 diff = 0, fx = 0, lx = 0, blo = Bw(bw);
                int ft = 0, lt = 0;
                for (int t0 = 0; t0 <= nlw; t0 += 4) {
                    u64 b[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) b[u] = Bw(bw + t0 + u + 1);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const int t = t0 + u;
                        if (t <= nlw) {
                            u64 xt = T[w0 + t] ^ ((blo << sh) | ((b[u] >> 1) >> (63 - sh)));
                            if (t == 0) xt &= firstmask;
                            if (t == nlw) xt &= lastmask;
                            diff |= xt;
                            if (xt && fx == 0) fx = xt, ft = t;
                            if (xt) lx = xt, lt = t;
                            blo = b[u];
                        }
                    }
                }
                const bool at_start = prefix_align != (rev != 0);
                const bool first_clean = fx == 0 || 32 * (w0 + ft) + (__clzll((long long)fx) >> 1) >= X0 + k;
                const bool last_clean = lx == 0 || 32 * (w0 + lt) + ((64 - __ffsll((long long)lx)) >> 1) < X1 - k;
                if (at_start ? first_clean : last_clean) my_khits++;
                if (diff == 0) {
                    if (contain && CONTAIN_KEY_HIT(prefix_align) && (LA > LB || (LA == LB && A < B))) atomicMin(&a.best[B], CKEY_MAKE(A, j, suf, rev));
                    ov = overlap;
                }
            }
            const u64 mk = __ballot(ov);
            if (ov) row[nkeep + rank_below(mk)] = h;
            nkeep += __popcll(mk);
            ndef = 0;
            __syncthreads();
        };
        Rows R0 = load_rows(h0);
        for (u32 i0 = 0; i0 < c; i0 += 64) {
            const u64 h2 = load_cand(i0 + 128);
            const Rows R1 = load_rows(h1);
            const bool act = i0 + lane < c;
            u64 h = h0;
            bool ov = false;
            u64 B = HIT_ID(h);
            const int LB = act ? (int)HIT_LEN(h) : k;
            const bool blong = LB > DISCO_SHORT_MAX;
            const bool fast = act && a_lds && !blong;
            { /* the rows into the lanes' staging rows */
                u64 *d = s_b + 1 + lane * BST;
                d[0] = R0.q0.x, d[1] = R0.q0.y, d[2] = R0.q1.x, d[3] = R0.q1.y, d[4] = R0.q2.x, d[5] = R0.q2.y, d[6] = R0.q3.x, d[7] = R0.q3.y;
            }
            if (act && B >= a.v.n) { /* (a tail row's id: a long candidate) */
                B = a.v.long_ids[B - a.v.n];
                h = (h & ~(0x7FFFFFFFull << 17)) | (B << 17);
            }
            const bool defer = act && blong && a_lds;
            {
                const u64 dm = __ballot(defer);
                if (dm) { /* (wave uniform) */
                    if (ndef + (u32)__popcll(dm) > 64u) flush();
                    if (defer) s_def[ndef + (u32)rank_below(dm)] = h;
                    ndef += (u32)__popcll(dm);
                }
            }
            if (act && !defer) {
                const int j = (int)HIT_J(h);
                const u32 suf = HIT_SUFFIX(h), rev = HIT_REV(h);
                const bool prefix_align = (suf == rev);
                const int d = prefix_align ? j : j + k - LB;
                const int x0 = d > 0 ? d : 0, x1 = min(LA, d + LB);
                bool contain, overlap;
                if (prefix_align) {
                    contain = LA - j >= LB;       /* BG/OverlapGraph.cpp:532 */
                    overlap = !contain && j >= 1; /* :579 */
                } else {
                    contain = d >= 0;           /* :547 */
                    overlap = d <= 0 && j >= 1; /* :591 */
                }
                bool kmer_ok, region_ok;
                if (fast) {
                    /* T[X] == B[X - dd] for X in [X0, X1), T = A or revcomp(A) (verify_kernel's staged compare) */
                    const u64 *T = rev ? s_arc + 1 : s_a + 1;
                    const int X0 = rev ? LA - x1 : x0, X1 = rev ? LA - x0 : x1;
                    const int dd = rev ? LA - LB - d : d;
                    const int w0 = X0 >> 5, nlw = ((X1 - 1) >> 5) - w0; /* <= 8: the region is at most LB <= 256 bases */
                    const int p = 32 * w0 - dd; /* >= -31: at most one word in front of the row is touched */
                    const u64 *bp = s_b + 1 + lane * BST + (p >> 5);
                    const int sh = (p & 31) * 2;
                    const u64 firstmask = ~0ull >> (2 * (X0 & 31)), lastmask = ~0ull << (62 - 2 * ((X1 - 1) & 31));
                    u64 diff = 0, fx = 0, lx = 0, blo = bp[0];
                    int ft = 0, lt = 0;
#pragma unroll
                    for (int t = 0; t <= VERIFY_SW; t++) {
                        if (t <= nlw) {
                            const u64 bhi = bp[t + 1];
                            u64 xt = T[w0 + t] ^ ((blo << sh) | ((bhi >> 1) >> (63 - sh)));
                            if (t == 0) xt &= firstmask;
                            if (t == nlw) xt &= lastmask;
                            diff |= xt;
                            if (xt && fx == 0) fx = xt, ft = t;
                            if (xt) lx = xt, lt = t;
                            blo = bhi;
                        }
                    }
                    /* the seed k-mer sits at the start of the region (T coordinates) for types 0 / 3, at its end for 1 / 2: it matches iff the
                     * differing base nearest that end is at least k bases in (verify_kernel) */
                    const bool at_start = prefix_align != (rev != 0);
                    const bool first_clean = fx == 0 || 32 * (w0 + ft) + (__clzll((long long)fx) >> 1) >= X0 + k;
                    const bool last_clean = lx == 0 || 32 * (w0 + lt) + ((64 - __ffsll((long long)lx)) >> 1) < X1 - k;
                    kmer_ok = at_start ? first_clean : last_clean;
                    region_ok = diff == 0;
                } else {
                    const int SB = blong ? a.v.SL : a.v.S;
                    const u64 *gb = blong ? a.v.full + (u64)a.v.ovf[B] * SB : a.v.reads + B * SB;
                    kmer_ok = seg_equal2(ga, SA, gb, SB, LB, j, prefix_align ? 0 : LB - k, k, rev);
                    region_ok = kmer_ok && seg_equal2(ga, SA, gb, SB, LB, x0, x0 - d, x1 - x0, rev);
                }
                if (kmer_ok) my_khits++;
                if (region_ok) {
                    if (contain && CONTAIN_KEY_HIT(prefix_align) && (LA > LB || (LA == LB && A < B))) atomicMin(&a.best[B], CKEY_MAKE(A, j, suf, rev));
                    ov = overlap;
                }
            }
            const u64 mk = __ballot(ov);
            if (ov) row[nkeep + rank_below(mk)] = h;
            nkeep += __popcll(mk);
            h0 = h1;
            h1 = h2;
            R0 = R1;
        }
        if (ndef) flush();
        if (lane == 0) {
            if (c > 64) a.row_cnt[A] = nkeep; /* (rows of up to 64 candidates have no entry there: probe_kernel) */
            my_big += nkeep > ES_CAP ? 1u : 0u;
            my_mid += nkeep > 64u ? 1u : 0u;
            my_raw += nkeep;
            a.meta_ord[ci].y = (u64)nkeep | ((u64)LA << 32);
        }
    }
    }
    for (int o = 32; o > 0; o >>= 1) my_khits += __shfl_down(my_khits, o);
    if (lane == 0) {
        if (my_khits) atomicAdd(&a.v.ctr[CTR_KMER_HITS], my_khits);
        if (my_raw) atomicAdd(&a.v.ctr[CTR_RAW_HITS], my_raw);
        if (my_big) atomicAdd(&a.v.ctr[CTR_ES_BIG], (u64)my_big);
        if (my_mid) atomicAdd(&a.v.ctr[CTR_ES_MID], (u64)my_mid);
    }
}

/* ================================================================================================================
 * containment finalisation — contained flag per read from the reduced keys (BG/OverlapGraph.cpp:495-503 count)
 * ============================================================================================================== */
__global__ void contain_flags_kernel(const u64 *__restrict__ best, u64 n, u8 *__restrict__ contained, u64 *__restrict__ cbits, u64 *ctr)
{
    /* cbits: one bit per read (n/8 bytes: 6 MB at 50 M reads, L2-resident for the gathers of edge selection) */
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    const u64 n64 = (n + 63) & ~63ull;
    for (; i < n64; i += (u64)gridDim.x * blockDim.x) {
        u8 f = (i < n) && best[i] != DISCO_NOKEY;
        if (i < n) contained[i] = f;
        const u64 mk = __ballot(f);
        if ((threadIdx.x & 63) == 0) cbits[i >> 6] = mk;
        c += f;
    }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&ctr[CTR_N_CONTAINED], (u64)c);
}

/* gather (read id, key) of the contained reads in ascending id order; pos = exclusive scan of the flags */
__global__ void contain_rows_kernel(const u64 *__restrict__ best, const u8 *__restrict__ contained, const u64 *__restrict__ pos,
                                    u64 n, u64 *__restrict__ out_id, u64 *__restrict__ out_key)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (contained[i]) {
            out_id[pos[i]] = i;
            out_key[pos[i]] = best[i];
        }
}

/* the same with 32-bit positions and ids (fewer than 2^31 reads per context): what travels to the host is 12 bytes per row */
__global__ void contain_rows32_kernel(const u64 *__restrict__ best, const u8 *__restrict__ contained, const u32 *__restrict__ pos,
                                      u64 n, u32 *__restrict__ out_id, u64 *__restrict__ out_key)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (contained[i]) {
            out_id[pos[i]] = (u32)i;
            out_key[pos[i]] = best[i];
        }
}

/* ---- the contained rows in the order the contained-read files are written in: grouped by containing read (rows of one containing
 * read adjacent: SG/DataSet.cpp:316-335), inside a group ascending (j, contained id) — a counting sort on the containing read
 * (count, scan, place) and an insertion sort inside the groups, which are a handful of rows (groups beyond CROW_GROUP_MAX are left
 * unsorted and counted: the caller then sorts on the host) ---------------------------------------------------------------------- */
#define CROW_GROUP_MAX 256
/* the lengths a contained row carries (len2 = the contained read's, len1 = the containing read's), next to its id and key: the host
 * decodes rows without gathering from its own copy of the length table (two random reads per row: 9 of the 9.1 ms of
 * disco_fetch_contained at 4.7 M rows) */
__global__ void crow_lens_kernel(const u32 *__restrict__ id, const u64 *__restrict__ key, u64 nc, const u16 *__restrict__ len, u32 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nc; i += (u64)gridDim.x * blockDim.x) out[i] = (u32)len[id[i]] | ((u32)len[CKEY_SUPER(key[i])] << 16);
}

__global__ void crow_count_kernel(const u64 *__restrict__ key, u64 nc, u32 *__restrict__ cnt)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nc; i += (u64)gridDim.x * blockDim.x) atomicAdd(&cnt[CKEY_SUPER(key[i])], 1u);
}
/* cursor = the exclusive scan of cnt (consumed: afterwards cursor[s] = end of group s) */
__global__ void crow_place_kernel(const u32 *__restrict__ id, const u64 *__restrict__ key, u64 nc, u32 *__restrict__ cursor, u32 *__restrict__ out_id, u64 *__restrict__ out_key)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nc; i += (u64)gridDim.x * blockDim.x) {
        const u32 at = atomicAdd(&cursor[CKEY_SUPER(key[i])], 1u);
        out_id[at] = id[i];
        out_key[at] = key[i];
    }
}
/* one thread per row that starts a group */
__global__ void crow_sort_groups_kernel(u32 *__restrict__ id, u64 *__restrict__ key, u64 nc, u64 *__restrict__ n_big_groups)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nc; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = CKEY_SUPER(key[i]);
        if (i > 0 && CKEY_SUPER(key[i - 1]) == s) continue;
        u64 e = i + 1;
        while (e < nc && CKEY_SUPER(key[e]) == s) e++;
        if (e - i < 2) continue;
        if (e - i > CROW_GROUP_MAX) {
            atomicAdd(n_big_groups, 1ull);
            continue;
        }
        for (u64 a = i + 1; a < e; a++) { /* (j, contained): the key's (super, j) bits, then the id */
            const u64 k = key[a], kj = k >> 2;
            const u32 d = id[a];
            u64 b = a;
            while (b > i && ((key[b - 1] >> 2) > kj || ((key[b - 1] >> 2) == kj && id[b - 1] > d))) {
                key[b] = key[b - 1];
                id[b] = id[b - 1];
                b--;
            }
            key[b] = k;
            id[b] = d;
        }
    }
}

/* ================================================================================================================
 * edge selection — insertAllEdgesOfRead (BG/OverlapGraph.cpp:631-678) for every non-contained query read, from the
 * verified hits of the probe: drop hits to contained reads (:657; getListOfReads skips them :533), order the rest as
 * the reference consumes them (j ascending, then bucket order), accept at most max_edges_per_kmer per j (:645), one
 * edge per destination (insertedEdgeList :656), then sort the finds by overlap offset (:675-676).
 * Finds overwrite the head of the read's hit row in place.
 * h / t are two work arrays of cap entries: LDS for ordinary rows, global scratch for the big-row variant.
 * ============================================================================================================== */
__device__ __forceinline__ bool is_contained(const u64 *__restrict__ cbits, u64 id)
{
    return (((const u32 *)cbits)[id >> 5] >> (id & 31)) & 1u;
}

struct EdgeSelArgs {
    DiscoView v;
    const u64 *contained; /* bitmap */
    u64 *hits;
    const u64 *row_start;
    const u32 *row_cnt;
    u64 *ref;      /* [n] out: row position | finds << 40          */
    u32 max_per_kmer;
    u64 *big_list; /* rows with more than ES_CAP hits (out/in)   */
    u32 *n_big;
    u32 big_cap;
    u64 *scratch;  /* BIG: gridDim.x * 2 * scratch_cap entries   */
    u64 scratch_cap;
    /* the reads are taken in the processing order of probe / verify (or ascending id when order is null), headers by position:
     * {row start, verified hits | length << 32}. Reads of one locus come back to back, so the bitmap words of their (shared)
     * destinations are still in the L2 for the next read — in file order 16 of a read's 44 bitmap gathers missed it */
    const u64 *order;
    const ulonglong2 *meta_ord;
    u64 *dropbits; /* out: one bit per read whose selection dropped a verified hit (only those lists can lack a twin: twin_check) */
    /* exact overlaps, or null: every dropped hit as {read, the entry it would have become}. A twin can be missing from a list only where
     * its owner dropped exactly that hit (twin_check in disco_hip.hip), so the twin search needs these few thousand items, not a
     * pass over every entry of every list (16 ms at 50 M reads with 0.1 % errors — real reads always drop something) */
    u64 *drop_node, *drop_key;
    u32 drop_cap;
    u32 hidden_flags; /* inexact overlaps: verified hits carry HIT_HIDDEN_BIT; it is kept out of the consumption order and handed on in
                         the entry's ADJ_FLAG bit (ADJ_HIDDEN_OF) */
};

/* stable-free rank sort of m distinct keys from src into dst (wave cooperative) */
__device__ __forceinline__ void wave_rank_sort(const u64 *src, u64 *dst, u32 m, u32 lane)
{
    for (u32 i = lane; i < m; i += 64) {
        u64 x = src[i];
        u32 r = 0;
        for (u32 t = 0; t < m; t++) r += (src[t] < x);
        dst[r] = x;
    }
}

/* ascending bitonic sort of h[0..P) in LDS, P a power of two >= 64 (wave cooperative, P/2 compare-exchanges per step) */
__device__ __forceinline__ void lds_bitonic_sort(u64 *h, u32 P, u32 lane)
{
    for (u32 k2 = 2; k2 <= P; k2 <<= 1)
        for (u32 j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            for (u32 t = lane; t < P / 2; t += 64) {
                const u32 i = 2 * t - (t & (j2 - 1)); /* t with a zero inserted at bit log2(j2) */
                const u64 x = h[i], y = h[i + j2];
                const bool up = (i & k2) == 0;
                if ((x > y) == up) {
                    h[i] = y;
                    h[i + j2] = x;
                }
            }
            __syncthreads();
        }
}

/* one lane: the dropped verified hit of read A goes to the drop list */
__device__ __forceinline__ void record_drop(const EdgeSelArgs &a, u64 A, u32 LA, u64 hit)
{
    if (!a.drop_node) return;
    /* a full list is a useless list (the host falls back to the bitmap search when the count differs from the number of drops): stop
     * counting then — a repeat-rich read set drops hits by the hundred million, all on this one address */
    if (__atomic_load_n(&a.v.ctr[CTR_DROP_ITEMS], __ATOMIC_RELAXED) >= a.drop_cap) return;
    const u64 idx = atomicAdd(&a.v.ctr[CTR_DROP_ITEMS], 1ull);
    if (idx < a.drop_cap) {
        u32 orient, off;
        disco_map_type(disco_hit_type(HIT_SUFFIX(hit), HIT_REV(hit)), LA, (u32)a.v.k, HIT_J(hit), &orient, &off);
        a.drop_node[idx] = A;
        a.drop_key[idx] = ADJ_MAKE(off, HIT_ID(hit), orient, HIT_LEN(hit));
    }
}

__device__ __forceinline__ void tr_wide_flush(const EdgeSelArgs &a, u64 tr_wide, u32 lane)
{
    if (lane == 0 && (u32)tr_wide) atomicAdd(&a.v.ctr[CTR_TR_BIG], (u64)(u32)tr_wide);
    if (lane == 0 && (tr_wide >> 32)) atomicAdd(&a.v.ctr[CTR_TR_MID], tr_wide >> 32);
}

/* WCAP > 0: h / t are LDS arrays of WCAP entries (a power of two) and s_jcnt a 128-slot LDS histogram: rows of up to WCAP hits
 * may take the accept-all shortcut */
template <int WCAP>
/* rs: start of the row in the hit buffer (the header by position for rows of the main pass; row_start[A] — kept for rows of more
 * than 64 entries only — for the listed ones) */
__device__ __forceinline__ void edge_select_row(const EdgeSelArgs &a, u64 A, u64 rs, u64 *h, u64 *t, u32 c, u32 lane, u32 *s_jcnt, u32 &cap_sites,
                                                u32 &dropped, u64 &n_edges, u64 &tr_wide)
{
    /* tr_wide: nodes with more than TR_CAP (low word) / TR_CAP_SMALL (high word) finds, this wavefront's — they size the marking's
     * big-node list and pick its variant; counted here and added once per wavefront (tr_wide_flush): an atomic per row on one address is
     * 12 ns per row, and at coverages where every row comes this way that was most of the pass */
#ifdef ES_EXP_STALE_ROWSTART /* tools/ab_build.py: the defect tests/test_gpu_parity.py::test_cap_binds_in_short_rows guards against */
    rs = a.row_start[A];
#endif
    u64 *row = a.hits + rs;
    const u32 LA = a.v.len[A];
    /* 1. drop hits to contained reads */
    u32 m = 0;
    for (u32 i0 = 0; i0 < c; i0 += 64) {
        u32 i = i0 + lane;
        u64 hit = 0;
        bool keep = false;
        if (i < c) {
            hit = row[i];
            keep = (hit != ~0ull) && !is_contained(a.contained, HIT_ID(hit));
        }
        u64 mk = __ballot(keep);
        if (keep) h[m + rank_below(mk)] = hit;
        m += __popcll(mk);
    }
    __syncthreads();
    if (WCAP > 0 && m <= (u32)WCAP) {
        /* 1b. rows of 65..WCAP hits (coverage above 45x): the same shortcut as edge_select_row_all. No destination twice —
         * decided exactly by an LDS hash set (linear probing, 2 WCAP slots in the space of t) — and no window over the cap
         * (LDS histogram; slots shared by windows 128 apart only make the test conservative): every hit becomes an edge and
         * only the sort by offset remains (bitonic in LDS). Otherwise the sequential scan below decides. */
        u32 *hs = (u32 *)t;
        constexpr u32 HS = 2u * (u32)(WCAP > 0 ? WCAP : 1);
        for (u32 i = lane; i < HS; i += 64) hs[i] = 0xFFFFFFFFu;
        s_jcnt[lane] = 0;
        s_jcnt[lane + 64] = 0;
        __syncthreads();
        bool bad = false;
        for (u32 i = lane; i < m; i += 64) {
            const u64 hit = h[i];
            const u32 id = (u32)HIT_ID(hit);
            u32 idx = ((id * 0x9E3779B1u) >> 8) & (HS - 1u);
            for (;;) {
                const u32 old = atomicCAS(&hs[idx], 0xFFFFFFFFu, id);
                if (old == 0xFFFFFFFFu) break;
                if (old == id) {
                    bad = true;
                    break;
                }
                idx = (idx + 1) & (HS - 1u);
            }
            atomicAdd(&s_jcnt[HIT_J(hit) & 127u], 1u);
        }
        __syncthreads();
        for (u32 i = lane; i < m; i += 64) bad |= s_jcnt[HIT_J(h[i]) & 127u] > a.max_per_kmer;
        if (!__any(bad)) {
            u32 P = 64;
            while (P < m) P <<= 1;
            for (u32 i = lane; i < P; i += 64) {
                u64 ent = ~0ull;
                if (i < m) {
                    const u64 hit = h[i];
                    u32 orient, off;
                    disco_map_type(disco_hit_type(HIT_SUFFIX(hit), HIT_REV(hit)), LA, (u32)a.v.k, HIT_J(hit), &orient, &off);
                    ent = ADJ_MAKE(off, HIT_ID(hit), orient, HIT_LEN(hit)) | ADJ_HIDDEN_OF(hit);
                }
                h[i] = ent;
            }
            __syncthreads();
            lds_bitonic_sort(h, P, lane);
            for (u32 i = lane; i < m; i += 64) row[i] = h[i];
            if (lane == 0) a.ref[A] = REF_MAKE(rs, m);
            tr_wide += (m > TR_CAP ? 1ull : 0ull) + (m > TR_CAP_SMALL ? 1ull << 32 : 0ull);
            n_edges += m;
            __syncthreads();
            return;
        }
        __syncthreads();
    }
    /* 2. consumption order (the hidden flag of a hit — inexact overlaps — kept out of it) */
    if (a.hidden_flags) {
        for (u32 i = lane; i < m; i += 64) h[i] = HIT_SORT_KEY(h[i]);
        __syncthreads();
    }
    wave_rank_sort(h, t, m, lane);
    __syncthreads();
    /* 3. sequential accept scan (wave-uniform control flow; lanes share the membership test) */
    u32 nacc = 0, ctr = 0, curj = 0xFFFFFFFFu;
    bool capflag = false;
    for (u32 i = 0; i < m; i++) {
        const u64 hit = a.hidden_flags ? HIT_FROM_SORT_KEY(t[i]) : t[i];
        const u32 j = HIT_J(hit);
        const u64 B = HIT_ID(hit);
        if (j != curj) {
            curj = j;
            ctr = 0;
            capflag = false;
        }
        if (ctr >= a.max_per_kmer && capflag) {
            if (lane == 0) record_drop(a, A, LA, hit);
            continue;
        }
        bool seen = false;
        for (u32 x = lane; x < nacc; x += 64) seen |= (ADJ_DST(h[x]) == B);
        if (__any(seen)) {
            if (lane == 0) record_drop(a, A, LA, hit);
            continue;
        }
        if (ctr < a.max_per_kmer) {
            u32 orient, off;
            disco_map_type(disco_hit_type(HIT_SUFFIX(hit), HIT_REV(hit)), LA, (u32)a.v.k, j, &orient, &off);
            if (lane == 0) h[nacc] = ADJ_MAKE(off, B, orient, HIT_LEN(hit)) | ADJ_HIDDEN_OF(hit);
            nacc++;
            ctr++;
            __syncthreads();
        } else {
            cap_sites++; /* the cap cut off a hit that would have been accepted */
            capflag = true;
            if (lane == 0) record_drop(a, A, LA, hit);
        }
    }
    __syncthreads();
    /* 4. list order of the reference: ascending overlap offset (total tie-break: dst, orient) */
    wave_rank_sort(h, t, nacc, lane);
    __syncthreads();
    for (u32 i = lane; i < nacc; i += 64) row[i] = t[i];
    if (lane == 0) a.ref[A] = REF_MAKE(rs, nacc);
    tr_wide += (nacc > TR_CAP ? 1ull : 0ull) + (nacc > TR_CAP_SMALL ? 1ull << 32 : 0ull);
    n_edges += nacc;
    dropped += m - nacc;
    if (lane == 0 && m != nacc) atomicOr(&a.dropbits[A >> 6], 1ull << (A & 63));
    __syncthreads();
}

/* rows of at most 64 hits entirely in registers: bitonic sort into consumption order, first occurrence of every
 * destination = accepted (valid while no k-mer group has more than max_per_kmer acceptable hits, i.e. the cap never
 * blocks anything — otherwise return false and let the sequential scan decide), bitonic sort by offset, write back. */
__device__ __forceinline__ bool edge_select_row_fast(const EdgeSelArgs &a, u64 A, u64 rs, u32 LA, u64 hit, u32 lane, u32 &dropped, u64 &n_edges)
{
    u64 *row = a.hits + rs;
    if (a.hidden_flags) hit = HIT_FROM_SORT_KEY(wave_bitonic_sort(HIT_SORT_KEY(hit), lane)); /* (~0 stays ~0) */
    else hit = wave_bitonic_sort(hit, lane);
    const bool valid = hit != ~0ull;
    const u32 m = __popcll(__ballot(valid));
    const u32 B = (u32)HIT_ID(hit);
    const u32 j = HIT_J(hit);
    bool dup = false;
    for (u32 t = 0; t + 1 < m; t++) {
        const u32 bt = (u32)__builtin_amdgcn_readlane((int)B, (int)t);
        dup |= (lane > t) && (B == bt);
    }
    const bool nondup = valid && !dup;
    const u32 jprev = __shfl_up(j, 1);
    const bool start = valid && (lane == 0 || j != jprev);
    const u64 sm = __ballot(start), nd = __ballot(nondup);
    const u64 lt = lane_mask_lt();
    const u64 le = lt | (1ull << lane);
    const u64 smle = sm & le;
    const u32 gs = smle ? 63u - (u32)__clzll((long long)smle) : 0u;
    const u64 range = lt & ~((1ull << gs) - 1ull);
    const bool viol = nondup && (u32)__popcll(nd & range) >= a.max_per_kmer;
    if (__any(viol)) return false;
    if (valid && dup) record_drop(a, A, LA, hit);
    u64 ent = ~0ull;
    if (nondup) {
        u32 orient, off;
        disco_map_type(disco_hit_type(HIT_SUFFIX(hit), HIT_REV(hit)), LA, (u32)a.v.k, j, &orient, &off);
        ent = ADJ_MAKE(off, B, orient, HIT_LEN(hit)) | ADJ_HIDDEN_OF(hit);
    }
    ent = wave_bitonic_sort(ent, lane);
    const u32 nacc = __popcll(nd);
    if (lane < nacc) row[lane] = ent;
    if (lane == 0) a.ref[A] = REF_MAKE(rs, nacc);
    n_edges += nacc;
    dropped += m - nacc; /* second and later hits to a destination already linked (BG/OverlapGraph.cpp:656) */
    if (lane == 0 && m != nacc) atomicOr(&a.dropbits[A >> 6], 1ull << (A & 63));
    return true;
}

/* ascending sort of the wave's adjacency entries (one per lane, ~0 = none) for reads of up to 256 bases, by COUNTING on the offset —
 * the entry's leading field, below 256 then: a histogram over 256 LDS bins whose counting atomic hands every entry its rank inside
 * its bin, an exclusive scan of the bins (four per lane + one wave scan), a scatter. Entries of EQUAL offset (common: a read has
 * about five such pairs, on its two sides) land in arbitrary order inside their bin; odd-even exchanges of neighbours put them right
 * (bins are runs of at most a few entries: one or two rounds, the last one finding nothing to do). About 75 vector instructions
 * where the 21-step bitonic network on 64-bit keys takes 126. bins: 256 u32, srt: 64 u64 (LDS). Returns lane i's entry of the
 * sorted order (~0 behind the last one). */
__device__ __forceinline__ u64 wave_sort_by_offset(u64 ent, bool valid, u32 lane, u32 *bins, u64 *srt)
{
    ((uint4 *)bins)[lane] = make_uint4(0u, 0u, 0u, 0u);
    srt[lane] = ~0ull;
    __syncthreads();
    const u32 off = valid ? ADJ_OFF(ent) : 0u;
    u32 rk = 0;
    if (valid) rk = atomicAdd(&bins[off], 1u);
    __syncthreads();
    const uint4 c = ((const uint4 *)bins)[lane]; /* lane owns bins 4 lane .. 4 lane + 3 */
    const u32 s1 = c.x + c.y, s2 = s1 + c.z, s3 = s2 + c.w;
    const u32 excl = wave_inclusive_add(s3) - s3;
    __syncthreads();
    ((uint4 *)bins)[lane] = make_uint4(excl, excl + c.x, excl + s1, excl + s2);
    __syncthreads();
    if (valid) srt[bins[off] + rk] = ent;
    __syncthreads();
    u64 x = srt[lane];
    /* entries of one bin in arbitrary order: neighbour exchanges until nothing is out of order (~0 sorts last) */
    for (;;) {
        const u64 nx = ((u64)(u32)__shfl_down((int)(u32)(x >> 32), 1) << 32) | (u32)__shfl_down((int)(u32)x, 1);
        if (!__any(lane < 63 && x > nx)) break;
        /* pairs (0,1) (2,3) ... */
        u64 y = lane_xor64(x, 1);
        bool lower = (lane & 1u) == 0;
        x = ((x < y) == lower) ? x : y;
        /* pairs (1,2) (3,4) ... : lanes 0 and 63 sit out */
        const u32 partner = (lane & 1u) ? lane + 1 : lane - 1;
        y = ((u64)(u32)__shfl((int)(u32)(x >> 32), (int)(partner & 63u)) << 32) | (u32)__shfl((int)(u32)x, (int)(partner & 63u));
        lower = (lane & 1u) == 1;
        if (lane != 0 && lane != 63) x = ((x < y) == lower) ? x : y;
    }
    return x;
}

/* the common case of the common case: no destination occurs twice and no k-mer has more than max_per_kmer hits, so every
 * verified hit to a non-contained read becomes an edge and the consumption order is irrelevant: one 32-bit sort of the
 * destinations proves the first condition, an LDS histogram of the windows the second, and only the sort by offset remains.
 * Anything else falls through to edge_select_row_fast (exact for every row of at most 64 hits). */
__device__ __forceinline__ bool edge_select_row_all(const EdgeSelArgs &a, u64 A, u64 rs, u32 LA, u64 hit, u32 lane, u32 *s_jcnt, u8 *s_dup,
                                                    u64 &n_edges, u32 *s_bins = nullptr, u64 *s_srt = nullptr)
{
    u64 *row = a.hits + rs;
    const bool valid = hit != ~0ull;
    const u32 j = HIT_J(hit);
    s_jcnt[lane] = 0;
    s_jcnt[lane + 64] = 0;
    /* "no destination twice": every hit stamps its lane on a hashed byte slot (ES_DUPTAB slots, all zero between reads); if
     * every lane reads its own stamp back no two destinations even share a slot. Otherwise (true duplicate or a slot
     * collision, about a third of the rows) the 32-bit sort of the destinations decides. */
    const u32 slot = ((u32)HIT_ID(hit) * 0x9E3779B1u) >> (32 - ES_DUPBITS);
    if (valid) s_dup[slot] = (u8)(lane + 1);
    __syncthreads();
    const bool lost = valid && s_dup[slot] != (u8)(lane + 1);
    __syncthreads();
    if (valid) s_dup[slot] = 0;
    bool dup = false;
    if (__any(lost)) {
        const u32 bkey = valid ? (u32)HIT_ID(hit) : (0x80000000u | lane);
        const u32 sb = wave_bitonic_sort32(bkey, lane);
        const u32 nb = __shfl_down(sb, 1);
        dup = (lane < 63) && (sb == nb);
    }
    __syncthreads();
    if (valid) atomicAdd(&s_jcnt[j & 127u], 1u);
    __syncthreads();
    const bool capped = valid && s_jcnt[j & 127u] > a.max_per_kmer;
    if (__any(dup || capped)) return false;
    u64 ent = ~0ull;
    if (valid) {
        u32 orient, off;
        disco_map_type(disco_hit_type(HIT_SUFFIX(hit), HIT_REV(hit)), LA, (u32)a.v.k, j, &orient, &off);
        ent = ADJ_MAKE(off, HIT_ID(hit), orient, HIT_LEN(hit)) | ADJ_HIDDEN_OF(hit);
    }
#ifndef ES_NO_COUNT_SORT
    if (s_bins && LA <= 256u) ent = wave_sort_by_offset(ent, valid, lane, s_bins, s_srt); /* (LA is wave uniform) */
    else
#endif
        ent = wave_bitonic_sort(ent, lane);
    const u32 nacc = __popcll(__ballot(valid));
    if (lane < nacc) row[lane] = ent;
    if (lane == 0) a.ref[A] = REF_MAKE(rs, nacc);
    n_edges += nacc;
    return true;
}

#ifndef SELECT_WAVES_PER_SIMD
#define SELECT_WAVES_PER_SIMD 1
#endif
template <bool BIG>
__global__ void __launch_bounds__(64, SELECT_WAVES_PER_SIMD) edge_select_kernel(EdgeSelArgs a)
{
    __shared__ __attribute__((aligned(16))) u64 s_h[BIG ? 2 : ES_CAP];
    __shared__ __attribute__((aligned(16))) u64 s_t[BIG ? 2 : ES_CAP];
    __shared__ u32 s_jcnt[128];
    __shared__ u8 s_dup[BIG ? 1 : ES_DUPTAB];
    const u32 lane = threadIdx.x;
    if (!BIG) {
        for (u32 i = lane; i < ES_DUPTAB; i += 64) s_dup[i] = 0;
        __syncthreads();
    }
    u32 cap_sites = 0, dropped = 0, n_slow = 0;
    u64 n_edges = 0, tr_wide = 0; /* wave-uniform */
    const u64 n_items = BIG ? (u64)min(*a.n_big, a.big_cap) : (a.v.q_hi - a.v.q_lo);
    u64 *h = BIG ? a.scratch + (u64)blockIdx.x * 2 * a.scratch_cap : s_h;
    u64 *t = BIG ? h + a.scratch_cap : s_t;
    u64 cbeg = 0, cend = 0;
    /* software pipeline over the reads of a chunk (ordinary variant): row metadata three reads ahead, the hit row two ahead,
     * the contained-bitmap gather of its destinations one ahead, so that the dependent chain meta -> row -> bitmap of one
     * read overlaps the sorting of the previous ones. c = 0 stands for "nothing to do" (empty row, contained read). */
    /* every pipelined load is unconditional (clamped address) and nothing is computed from a loaded value in the iteration
     * that issues the load (the compiler waits for a load at its first use), see verify_kernel: ld_* issue, fin_* consume */
    struct SelMeta {
        u32 c, LA, cw; /* raw: candidates, length, bitmap word holding the read's own contained bit */
        u64 rs;
    };
    u64 ord_chunk = 0;                             /* lane i: read of the chunk's position i (WQ_CHUNK == 64) */
    ulonglong2 meta_chunk = make_ulonglong2(0, 0); /* lane i: its header                                      */
    auto read_of = [&](u64 it) { return ORDER_ID(readlane_u64(ord_chunk, (u32)((it < cend ? it : cend - 1) - cbeg))); };
    auto ld_meta = [&](u64 it) {
        SelMeta m;
        const u32 i = (u32)((it < cend ? it : cend - 1) - cbeg);
        const u64 A = ORDER_ID(readlane_u64(ord_chunk, i));
        const u64 w = readlane_u64(meta_chunk.y, i);
        m.c = (u32)w;
        m.LA = (u32)(w >> 32);
        m.rs = readlane_u64(meta_chunk.x, i);
        m.cw = ((const u32 *)a.contained)[A >> 5];
        return m;
    };
    auto fin_meta = [&](const SelMeta &m, u64 it) -> u32 { /* BG/OverlapGraph.cpp:657 : both reads must be non-contained */
        return (it < cend && !((m.cw >> (read_of(it) & 31)) & 1u)) ? m.c : 0u;
    };
    auto ld_row = [&](u32 c, u64 rs, u64 it) -> u64 { /* the hit buffer has more than 65536 slots */
        return a.hits[(c <= 64 && lane < c) ? rs + lane : (c ? rs : (it & 0xFFFFull))];
    };
    auto fin_row = [&](u32 c, u64 h) -> u64 { return (c <= 64 && lane < c) ? h : ~0ull; };
    auto ld_gather = [&](u64 h, u64 it) -> u32 { return ((const u32 *)a.contained)[(h != ~0ull ? HIT_ID(h) : read_of(it)) >> 5]; };
    auto fin_gather = [&](u64 h, u32 w) -> u64 { return (h != ~0ull && !((w >> (HIT_ID(h) & 31)) & 1u)) ? h : ~0ull; };
    while (BIG ? wq_grab<1>(a.v.wq, n_items, cbeg, cend) : wq_grab(a.v.wq, n_items, cbeg, cend)) { /* big rows: one per grab */
        if (BIG) {
            for (u64 it = cbeg; it < cend; it++) {
                const u64 A = a.big_list[it];
                n_slow++;
                if (a.row_cnt[A] <= ES_MID) continue; /* done by edge_select_mid_kernel */
                edge_select_row<0>(a, A, a.row_start[A], h, t, a.row_cnt[A], lane, s_jcnt, cap_sites, dropped, n_edges, tr_wide);
            }
            continue;
        }
        {
            const u64 i = cbeg + (lane < cend - cbeg ? lane : 0u);
            ord_chunk = a.order ? a.order[i] : a.v.q_lo + i;
            meta_chunk = a.meta_ord[i];
        }
        /* stages: meta of read it+3 | hit row of it+2 | bitmap gather of it+1 | selection of it */
        SelMeta m0 = ld_meta(cbeg), m1 = ld_meta(cbeg + 1), m2 = ld_meta(cbeg + 2);
        u32 c0 = fin_meta(m0, cbeg), c1 = fin_meta(m1, cbeg + 1);
        u64 s0 = m0.rs, s1 = m1.rs;
        u32 L0 = m0.LA, L1 = m1.LA;
        u64 r0 = ld_row(c0, s0, cbeg), r1 = ld_row(c1, s1, cbeg + 1);
        u64 h0 = fin_row(c0, r0);
        u32 w0 = ld_gather(h0, cbeg);
        for (u64 it = cbeg; it < cend; it++) {
            const SelMeta m3 = ld_meta(it + 3);
            const u32 c2 = fin_meta(m2, it + 2);
            const u64 r2 = ld_row(c2, m2.rs, it + 2);
            const u64 h1 = fin_row(c1, r1);
            const u32 w1 = ld_gather(h1, it + 1);
            const u64 g0 = fin_gather(h0, w0);
            const u64 A = read_of(it);
            if (c0 == 0) {
                if (lane == 0) a.ref[A] = 0;
            } else if (c0 > ES_CAP) {
                if (lane == 0) {
                    u32 idx = atomicAdd(a.n_big, 1u);
                    if (idx < a.big_cap) a.big_list[idx] = A;
                    else atomicAdd(&a.v.ctr[CTR_OVERFLOW], 1ull);
                    a.ref[A] = 0;
                }
            } else if (c0 <= 64 && edge_select_row_all(a, A, s0, L0, g0, lane, s_jcnt, s_dup, n_edges, (u32 *)s_t, s_h)) { /* (s_t / s_h: free here) */
            } else if (c0 <= 64 && edge_select_row_fast(a, A, s0, L0, g0, lane, dropped, n_edges)) {
            } else {
                n_slow++;
                edge_select_row<ES_CAP>(a, A, s0, h, t, c0, lane, s_jcnt, cap_sites, dropped, n_edges, tr_wide);
            }
            c0 = c1; s0 = s1; L0 = L1; h0 = h1; w0 = w1;
            c1 = c2; s1 = m2.rs; L1 = m2.LA; r1 = r2;
            m2 = m3;
        }
    }
    if (lane == 0 && n_edges) atomicAdd(&a.v.ctr[CTR_ADJ_TOTAL], n_edges);
    if (lane == 0 && n_slow) atomicAdd(&a.v.ctr[CTR_ES_SLOW], (u64)n_slow);
    if (lane == 0 && cap_sites) atomicAdd(&a.v.ctr[CTR_CAP_SITES], (u64)cap_sites);
    tr_wide_flush(a, tr_wide, lane);
    if (lane == 0 && dropped) atomicAdd(&a.v.ctr[CTR_DROPPED], (u64)dropped);
}

/* ================================================================================================================
 * edge_select_flat_kernel (round 4) — the same selection (insertAllEdgesOfRead, BG/OverlapGraph.cpp:631-678) with FULL wavefronts for
 * reads of up to 256 bases, exact overlaps. edge_select_kernel runs lane = hit of ONE read: a row has about 40 hits, 62 % of the lanes of
 * 172 vector instructions per read. Here ROWS consecutive reads of the work-queue chunk form a sub-chunk whose verified hits are ONE
 * flat list (exclusive scan of the counts in the chunk's headers), processed 64 at a time regardless of row boundaries, lane = hit:
 *   1. hit, contained bit of its destination, entry (offset | dst | orient | len); per ROW and exact, by LDS atomics: "no destination
 *      twice" (a hash set of 128 slots, compare-and-swap) and "no k-mer above the cap" (byte counters by window mod 128) — a row that
 *      fails either is left untouched and done the old way at the end of the chunk (edge_select_row_fast / edge_select_row); the
 *      counting atomic of the row's 256 offset bins (byte counters, four per word) hands the entry its arrival rank inside its bin;
 *   2. per row one wave step: exclusive scan of the 256 bins -> their first positions;
 *   3. entries to their bin's places in LDS (arrival order inside a bin);
 *   4. entries of one offset (about five pairs per row) are ordered by the full key: an entry counts the bin-mates below it; the row
 *      goes to its place in the hit buffer, sorted by (offset, dst, orient) like wave_sort_by_offset's.
 * Every hit of the sub-chunk is in a register before the first row is written. Same rows, same counters as edge_select_kernel.
 * ============================================================================================================== */
#ifndef SELECT_FLAT_WAVES_PER_SIMD
#define SELECT_FLAT_WAVES_PER_SIMD 1
#endif
/* SMALL (round 5): the kernel's time follows its resident waves (12 / 14 / 16 blocks per CU: 23.8 / 21.4 / 19.8 ms at 50 M reads), and what
 * held it at 16 were the two work arrays of the sequential path — sized for rows of ES_CAP hits that a read set of ordinary coverage
 * does not have — and a dozen registers. Where rows of more than 64 verified hits are rare (verify counts them: CTR_ES_MID) they go to
 * the big-row list (edge_select_mid_kernel<ES_CAP>), the sequential path works in arrays of 64, the sub-chunk is 3 rows x 2 batches (4 x 3
 * would be faster still, 18.1 ms, and spills three registers with the sequential path compiled in) and the kernel holds five waves per
 * SIMD: 18.8 ms. */
#ifndef SELECT_SMALL_WAVES
#define SELECT_SMALL_WAVES 5
#endif
template <int ROWS, int NB, bool SMALL = false>
__global__ void __launch_bounds__(64, SMALL ? SELECT_SMALL_WAVES : SELECT_FLAT_WAVES_PER_SIMD) edge_select_flat_kernel(EdgeSelArgs a)
{
    constexpr u32 SEQ_CAP = SMALL ? 64u : (u32)ES_CAP; /* rows the sequential path takes; longer ones: the big-row list */
    /* a sub-chunk: consecutive reads of the chunk, as many as fit ROWS rows and NB batches of 64 hits (greedy) */
    constexpr u32 SETW = 128;
    static_assert(NB >= 1 && ROWS >= 1 && ROWS <= 8, "a row has at most 64 hits: one batch always holds a row; the jc array is cleared by one store per lane");
    __shared__ __attribute__((aligned(16))) u64 s_ent[ROWS * 64 > 2 * SEQ_CAP ? ROWS * 64 : 2 * SEQ_CAP]; /* row r: [64 r, 64 r + 64); old paths: their two work arrays */
    __shared__ __attribute__((aligned(16))) u32 s_bins[ROWS * 64];  /* row r, offset o: byte o & 3 of word 64 r + (o >> 2): entries */
    __shared__ __attribute__((aligned(16))) u32 s_start[ROWS * 64]; /* ... : first position of the bin */
    __shared__ __attribute__((aligned(16))) u32 s_jc[ROWS * 32 > 128 ? ROWS * 32 : 128]; /* row r, window j: byte j & 3 of word 32 r + ((j & 127) >> 2) */
    __shared__ __attribute__((aligned(16))) u32 s_set[ROWS * SETW + 64]; /* row r: destinations seen (0xFFFFFFFF = free); behind them a spare word per lane */
    __shared__ u32 s_flag[ROWS];                                    /* row r must be done the old way */
    __shared__ ulonglong2 s_hdr[64];                                /* read i of the chunk: {row start, first flat index | length << 32} */
    const u32 lane = threadIdx.x;
    const u32 k = (u32)a.v.k;
    u32 cap_sites = 0, dropped = 0, n_slow = 0;
    u64 n_edges = 0, tr_wide = 0;
    u64 cbeg = 0, cend = 0;
#if defined(WQ_SPLIT_ALL)
    WqSplit wqs;
    while (wq_grab_split(a.v.wq, a.v.q_hi - a.v.q_lo, cbeg, cend, wqs)) {
#else
    while (wq_grab(a.v.wq, a.v.q_hi - a.v.q_lo, cbeg, cend)) {
#endif
        const u32 n = (u32)(cend - cbeg);
        const u64 ci = cbeg + (lane < n ? lane : 0u);
        const u64 ordw = a.order ? a.order[ci] : a.v.q_lo + ci;
        const u32 A = (u32)ORDER_ID(ordw);
        const ulonglong2 meta = a.meta_ord[ci];
        const u32 cw = ((const u32 *)a.contained)[A >> 5];
        const bool mine = lane < n && !((cw >> (A & 31)) & 1u); /* BG/OverlapGraph.cpp:657 : both reads must be non-contained */
        const u32 craw = mine ? (u32)meta.y : 0u;
        const u32 LAme = (u32)(meta.y >> 32);
        const bool longrow = craw > 64u || (craw != 0u && LAme > (u32)DISCO_SHORT_MAX); /* (a read of the long class: offsets beyond the bins) */
        const u32 c = longrow ? 0u : craw;
        const u32 incl = wave_inclusive_add(c);
        const u32 P = incl - c;
        const u32 total = (u32)__builtin_amdgcn_readlane((int)incl, 63);
        u64 slowmask = __ballot(longrow); /* reads of the chunk for the old paths */
        u32 nacc_me = 0;                  /* lane = read: its finds */
        __syncthreads();
        s_hdr[lane] = make_ulonglong2(meta.x, (u64)P | ((u64)LAme << 32));
        __syncthreads();
        /* Three stages, one sub-chunk each: A issues the hit loads, B turns the arrived hits into entries and issues the gathers of their
         * destinations' contained words, C does the rows. All loads are unconditional (clamped addresses: the number in flight is
         * known to the compiler) and live in registers nobody copies while they are in flight. */
        struct Sub {
            u32 r0, r1, base, end; /* reads [r0, r1) of the chunk, flat indices [base, end) */
        };
        auto Pat = [&](u32 r) { return r < 64u ? (u32)__builtin_amdgcn_readlane((int)P, (int)r) : total; };
        auto sub_from = [&](u32 r0) { /* (wave uniform, scalar loop) */
            Sub x;
            x.r0 = r0 < 64u ? r0 : 64u;
            x.base = Pat(x.r0);
            x.r1 = x.r0;
            x.end = x.base;
            while (x.r1 < 64u && x.r1 - x.r0 < (u32)ROWS) {
                const u32 e = Pat(x.r1 + 1u);
                if (e - x.base > 64u * (u32)NB) break;
                x.r1++;
                x.end = e;
            }
            return x;
        };
        u64 H[NB];    /* hits in flight */
        u32 segH[NB]; /* their read of the chunk | valid << 6 */
        auto stageA = [&](const Sub &sb) {
            const u32 base = sb.base, end = sb.end, r0 = sb.r0;
#pragma unroll
            for (int i = 0; i < NB; i++) {
                u32 f = base + 64u * (u32)i + lane;
                const bool valid = f < end;
                u32 seg = 0;
                u64 addr = (cbeg & 0xFFFFull) + lane; /* nothing to load: some line of the hit buffer (it has more than 65536 slots) */
                if (base + 64u * (u32)i < end) {      /* (wave uniform) */
                    f = valid ? f : end - 1u;
                    /* the read of flat index f: the sub-chunk has at most ROWS of them — one compare per row start, no loop (round 6: the scalar
                     * walk over the boundaries was a loop of cross-lane reads and branches per batch) */
                    seg = r0;
#pragma unroll
                    for (u32 r = 1; r < (u32)ROWS; r++) {
                        const u32 Ps = Pat(r0 + r < sb.r1 ? r0 + r : 64u); /* (rows beyond the sub-chunk: the chunk's total — never reached by f) */
                        seg += (r0 + r < sb.r1 && f >= Ps) ? 1u : 0u;
                    }
                    const ulonglong2 hd = s_hdr[seg];
                    addr = hd.x + (u64)(f - (u32)hd.y);
                }
                H[i] = a.hits[addr];
                segH[i] = seg | (valid ? 64u : 0u);
            }
        };
        u64 ent1[NB];
        u32 seg1[NB], CW1[NB];
        auto stageB = [&]() {
#pragma unroll
            for (int i = 0; i < NB; i++) {
                const u64 h = H[i];
                const u32 seg = segH[i] & 63u;
                const bool valid = segH[i] & 64u;
                const u32 id = valid ? (u32)HIT_ID(h) : 0u;
                const u32 LA = (u32)(s_hdr[seg].y >> 32);
                u32 orient, off;
                disco_map_hit(HIT_SUFFIX(h), HIT_REV(h), LA, k, HIT_J(h), &orient, &off);
                /* (the window's low 7 bits — all the row's cap test needs — ride along in seg1) */
                ent1[i] = ADJ_MAKE(off, id, orient, HIT_LEN(h));
                seg1[i] = segH[i] | ((HIT_J(h) & 127u) << 8);
#if defined(SEL_EXP) && SEL_EXP == 4 /* timing experiment (results are wrong): without the gather of the destination's contained word */
                CW1[i] = 0u;
#elif defined(SEL_EXP) && SEL_EXP == 5 /* ... with the gathers coalesced (a word near the lane's own): what the pipeline costs when they all hit */
                CW1[i] = ((const u32 *)a.contained)[(lane + 64u * (u32)i) & 1023u];
#else
                CW1[i] = ((const u32 *)a.contained)[id >> 5];
#endif
            }
        };
        Sub sC = sub_from(0), sB = sub_from(sC.r1), sA = sub_from(sB.r1);
        stageA(sC);
        stageB();
        stageA(sB);
        for (; sC.r0 < 64u; sC = sB, sB = sA, sA = sub_from(sA.r1)) {
            /* this sub-chunk's entries and contained words (arrived: the copy is where the wait belongs) */
            u64 ent[NB];
            u32 segm[NB], cwd[NB];
#pragma unroll
            for (int i = 0; i < NB; i++) {
                ent[i] = ent1[i];
                segm[i] = seg1[i];
                cwd[i] = pinned_copy32(CW1[i]);
            }
            stageB();   /* the next sub-chunk: its hits have arrived */
            stageA(sA); /* the one after */
            const u32 r0 = sC.r0, base = sC.base, end = sC.end;
            if (end == base) continue;
            const u32 nbat = (end - base + 63u) >> 6;
#if defined(SEL_EXP) && SEL_EXP == 3 /* timing experiment (results are wrong): the load pipeline alone — hits, entries, contained words */
            {
                u32 acc = 0;
#pragma unroll
                for (int i = 0; i < NB; i++) acc ^= (u32)ent[i] ^ segm[i] ^ cwd[i];
                if (acc == 0x12345u) a.hits[s_hdr[0].x] = acc;
                continue;
            }
#endif
            __syncthreads();
            {
                const uint4 z = make_uint4(0u, 0u, 0u, 0u), f = make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
                for (u32 x = 0; x < (ROWS * 64 / 4 + 63) / 64; x++)
                    if (x * 64 + lane < ROWS * 64 / 4) ((uint4 *)s_bins)[x * 64 + lane] = z;
#pragma unroll
                for (u32 x = 0; x < (ROWS * SETW / 4 + 63) / 64; x++)
                    if (x * 64 + lane < ROWS * SETW / 4) ((uint4 *)s_set)[x * 64 + lane] = f;
                if (lane < ROWS * 32 / 4) ((uint4 *)s_jc)[lane] = z;
                if (lane < ROWS) s_flag[lane] = 0u;
            }
            __syncthreads();
            /* 1. the row's tests, bins. segm: read (6) | ok << 6 | window & 127 << 8 -> read | ok << 6 | arrival rank << 8 */
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if ((u32)i < nbat) {
                    const u32 seg = segm[i] & 63u, rl = seg - r0;
                    const u32 id = (u32)ADJ_DST(ent[i]);
                    const bool ok = (segm[i] & 64u) && !((cwd[i] >> (id & 31u)) & 1u);
                    const u32 jw = (segm[i] >> 8) & 127u;
                    const u32 off = ADJ_OFF(ent[i]);
                    u32 rank = 0;
                    bool dup = false;
#if !defined(SEL_EXP) || SEL_EXP != 1
                    { /* "no destination twice": a wave-uniform probe loop (one trip for nearly every entry). Round 6, second form: the votes
                       * are SCALAR masks (pending lanes, duplicates), the loop ends on the and-not's condition code, and a lane that is not
                       * pending swaps at a spare word of its own instead of being masked out (a compare-and-swap under the exec mask and the
                       * merging of three per-lane booleans behind it were 27 instructions a trip; this is 14) */
                        u32 idx = (id * 0x9E3779B1u) >> 25; /* SETW = 128 slots */
                        u64 pm = __ballot(ok), dupm = 0ull;
                        const u32 spare = (u32)(ROWS * SETW) + lane; /* s_set has 64 words behind the rows' sets */
                        do {
                            u32 slot;
                            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(slot) : "v"(spare), "v"(rl * SETW + idx), "s"(pm));
                            const u32 old = atomicCAS(&s_set[slot], 0xFFFFFFFFu, id);
                            const u64 hm = __ballot(old == id) & pm; /* a second hit to this destination (BG/OverlapGraph.cpp:656): the consumption order decides */
                            const u64 fm = __ballot(old == 0xFFFFFFFFu);
                            dupm |= hm;
                            pm &= ~(hm | fm);
                            idx = (idx + 1u) & (SETW - 1u);
                        } while (pm != 0ull);
                        u32 dupv;
                        asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(dupv) : "s"(dupm));
                        dup = dupv != 0u;
                    }
#endif
                    if (ok) {
                        bool bad = dup;
#if !defined(SEL_EXP) || SEL_EXP != 1 /* (1: timing experiment, results wrong where a row has a destination twice or a k-mer over the cap: without the two tests) */
                        const u32 jold = atomicAdd(&s_jc[rl * 32u + (jw >> 2)], 1u << (8u * (jw & 3u)));
                        bad |= ((jold >> (8u * (jw & 3u))) & 0xFFu) >= a.max_per_kmer; /* (windows 128 apart share a counter: conservative) */
#endif
                        if (bad) s_flag[rl] = 1u;
                        const u32 bold = atomicAdd(&s_bins[rl * 64u + (off >> 2)], 1u << (8u * (off & 3u)));
                        rank = (bold >> (8u * (off & 3u))) & 0xFFu;
                    }
                    segm[i] = seg | (ok ? 64u : 0u) | (rank << 8);
                }
            }
            __syncthreads();
            /* rows for the old paths */
            {
                const u32 fl = lane < ROWS ? s_flag[lane] : 0u;
                slowmask |= __ballot(fl != 0u) << r0;
            }
            /* 2. per row: bins -> first positions; the row's number of finds */
#pragma unroll
            for (u32 r = 0; r < (u32)ROWS; r++) {
                if (r0 + r >= sC.r1) break; /* (wave uniform) */
                const u32 x = s_bins[r * 64u + lane];
                const u32 sum = (x * 0x01010101u) >> 24;
                const u32 inc = wave_inclusive_add(sum);
                s_start[r * 64u + lane] = x * 0x01010100u + (inc - sum) * 0x01010101u; /* (no byte carries: a row has at most 64 entries) */
                const u32 tot = (u32)__builtin_amdgcn_readlane((int)inc, 63);
                if (lane == r0 + r) nacc_me = tot;
            }
            __syncthreads();
            /* 3. entries to their bins' places (arrival order inside a bin); segm: read | go << 6 | arrival rank << 8 | bin size << 16 | first position << 24 */
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if ((u32)i < nbat) {
                    const u32 rl = (segm[i] & 63u) - r0, off = ADJ_OFF(ent[i]);
                    const bool go = (segm[i] & 64u) && !s_flag[rl];
                    const u32 st = (s_start[rl * 64u + (off >> 2)] >> (8u * (off & 3u))) & 0xFFu;
                    const u32 cnt = go ? (s_bins[rl * 64u + (off >> 2)] >> (8u * (off & 3u))) & 0xFFu : 0u;
                    if (go) s_ent[rl * 64u + st + ((segm[i] >> 8) & 0xFFu)] = ent[i];
                    segm[i] = (segm[i] & 0xFF3Fu) | (go ? 64u : 0u) | (cnt << 16) | (st << 24);
                }
            }
            __syncthreads();
            /* 4. place among the entries of the same offset by the full key, and out. Bins of two (most ties): the other one decides */
#pragma unroll
            for (int i = 0; i < NB; i++) {
                if ((u32)i < nbat) {
                    const u32 seg = segm[i] & 63u, rl = seg - r0;
                    const u32 st = segm[i] >> 24, cnt = (segm[i] >> 16) & 0xFFu, rk = (segm[i] >> 8) & 0xFFu;
                    const u64 other = s_ent[rl * 64u + st + (cnt == 2u ? (rk ^ 1u) : rk)];
                    u32 below = other < ent[i] ? 1u : 0u;
                    if (__any(cnt > 2u)) {
                        if (cnt > 2u) {
                            below = 0;
                            for (u32 t = 0; t < cnt; t++) below += s_ent[rl * 64u + st + t] < ent[i] ? 1u : 0u;
                        }
                    }
                    if (segm[i] & 64u) a.hits[s_hdr[seg].x + st + below] = ent[i];
                }
            }
        }
        /* the reads' reference words; reads for the old paths write their own */
        if (lane < n && !((slowmask >> lane) & 1ull)) a.ref[A] = c ? REF_MAKE(meta.x, nacc_me) : 0ull;
        {
            u32 t = ((slowmask >> lane) & 1ull) ? 0u : nacc_me;
            t = wave_inclusive_add(t);
            n_edges += (u32)__builtin_amdgcn_readlane((int)t, 63);
        }
        /* rows with a destination twice, a k-mer over the cap, or more than 64 hits: as edge_select_kernel does them */
#ifdef SEL_EXP_NOSLOW /* experiment (rows of the old paths are lost): what the kernel needs without their code */
        slowmask = 0;
#endif
        while (slowmask) {
            const u32 i = (u32)__ffsll((long long)slowmask) - 1u;
            slowmask &= slowmask - 1ull;
            const u64 Ai = (u32)__builtin_amdgcn_readlane((int)A, (int)i);
            const u64 rs = readlane_u64(meta.x, i);
            const u32 c0 = (u32)__builtin_amdgcn_readlane((int)craw, (int)i);
            const u32 L0 = (u32)__builtin_amdgcn_readlane((int)LAme, (int)i);
            __syncthreads();
            if (c0 > SEQ_CAP) {
                if (lane == 0) {
                    const u32 idx = atomicAdd(a.n_big, 1u);
                    if (idx < a.big_cap) a.big_list[idx] = Ai;
                    else atomicAdd(&a.v.ctr[CTR_OVERFLOW], 1ull);
                    a.ref[Ai] = 0;
                }
                continue;
            }
            bool done = false;
            if (c0 <= 64u) {
                u64 g = ~0ull;
                if (lane < c0) {
                    g = a.hits[rs + lane];
                    if (is_contained(a.contained, HIT_ID(g))) g = ~0ull;
                }
                done = edge_select_row_fast(a, Ai, rs, L0, g, lane, dropped, n_edges);
            }
            if (!done) {
                n_slow++;
                edge_select_row<(int)SEQ_CAP>(a, Ai, rs, s_ent, s_ent + SEQ_CAP, c0, lane, s_jc, cap_sites, dropped, n_edges, tr_wide);
            }
        }
    }
    if (lane == 0 && n_edges) atomicAdd(&a.v.ctr[CTR_ADJ_TOTAL], n_edges);
    if (lane == 0 && n_slow) atomicAdd(&a.v.ctr[CTR_ES_SLOW], (u64)n_slow);
    if (lane == 0 && cap_sites) atomicAdd(&a.v.ctr[CTR_CAP_SITES], (u64)cap_sites);
    tr_wide_flush(a, tr_wide, lane);
    if (lane == 0 && dropped) atomicAdd(&a.v.ctr[CTR_DROPPED], (u64)dropped);
}

/* rows of ES_CAP+1 .. ES_MID hits (coverage of a few hundred): the big-row list again, with LDS arrays large enough for the
 * accept-all shortcut; longer rows are left to the global-scratch variant */
/* (CAP = ES_CAP, round 5: the rows of 65 .. ES_CAP hits that the five-wave variant of edge_select_flat_kernel lists — 4 KB of LDS
 * instead of 16: the rows of the list with more than `above` and at most CAP hits) */
template <int CAP>
__global__ void __launch_bounds__(64) edge_select_mid_kernel(EdgeSelArgs a, u32 above)
{
    __shared__ u64 s_h[CAP];
    __shared__ u64 s_t[CAP];
    __shared__ u32 s_jcnt[128];
    const u32 lane = threadIdx.x;
    u32 cap_sites = 0, dropped = 0, n_slow = 0;
    u64 n_edges = 0, tr_wide = 0;
    const u64 n_items = (u64)min(*a.n_big, a.big_cap);
    u64 cbeg = 0, cend = 0;
    u64 gnext = 0, gend = 0;
    while (wq_grab_multi<1, 8>(a.v.wq, n_items, cbeg, cend, gnext, gend)) { /* (eight rows per atomic: a row per atomic is 12 ns per row on the one address) */
        const u64 A = a.big_list[cbeg];
        const u32 c = a.row_cnt[A];
        if (c > (u32)CAP || c <= above) continue;
        n_slow++;
        edge_select_row<CAP>(a, A, a.row_start[A], s_h, s_t, c, lane, s_jcnt, cap_sites, dropped, n_edges, tr_wide);
    }
    if (lane == 0 && n_edges) atomicAdd(&a.v.ctr[CTR_ADJ_TOTAL], n_edges);
    if (lane == 0 && n_slow) atomicAdd(&a.v.ctr[CTR_ES_SLOW], (u64)n_slow);
    if (lane == 0 && cap_sites) atomicAdd(&a.v.ctr[CTR_CAP_SITES], (u64)cap_sites);
    tr_wide_flush(a, tr_wide, lane);
    if (lane == 0 && dropped) atomicAdd(&a.v.ctr[CTR_DROPPED], (u64)dropped);
}

/* ================================================================================================================
 * adjacency addressing. Every node has one reference word  ref[v] = position(40) | degree(24) << 40  into an entry
 * array `adj`. On a single GPU `adj` IS the hit buffer: edge selection leaves each read's finds at the head of its hit row
 * and nothing is copied. In the sharded flow the rows are compacted into node order for the exchange and the gathered
 * array is addressed the same way. The transitive flag lives in bit 15 of the entry itself (free between len and orient),
 * so the twin's flag arrives with the twin and no flag array is touched on a single GPU.
 * ============================================================================================================== */

/* degree of the nodes [lo,hi) -> out[v-lo] */
__global__ void deg_from_ref_kernel(const u64 *__restrict__ ref, u64 lo, u64 hi, u32 *__restrict__ out)
{
    u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < hi; i += (u64)gridDim.x * blockDim.x) out[i - lo] = REF_DEG(ref[i]);
}

/* ref[v] = start[v] | deg << 40 from a node-ordered CSR (start = exclusive scan of deg) */
__global__ void ref_from_start_kernel(const u64 *__restrict__ start, const u32 *__restrict__ deg, u64 n, u64 *__restrict__ ref)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) ref[i] = REF_MAKE(start[i], deg[i]);
}

/* rows of the nodes [lo,hi) copied into node order: dst[dst_start[v-lo] + i] (export for the all-gather; flags stripped) */
/* (strip: all flag bits of the copy cleared; 0 keeps the hidden flags of a selection with inexact overlaps for the twin search) */
__global__ void __launch_bounds__(64) rows_gather_kernel(const u64 *__restrict__ adj, const u64 *__restrict__ ref, u64 lo, u64 hi,
                                                         const u64 *__restrict__ dst_start, u64 *__restrict__ dst, u64 strip)
{
    for (u64 v = lo + blockIdx.x; v < hi; v += gridDim.x) {
        const u64 r = ref[v];
        const u32 d = REF_DEG(r);
        const u64 *src = adj + REF_POS(r);
        u64 *o = dst + dst_start[v - lo];
        for (u32 i = threadIdx.x; i < d; i += 64) o[i] = src[i] & ~strip;
    }
}

/* the same rows as 4-byte entries dst(30) | orient(2) << 30: all the transitive marking reads of a NEIGHBOUR's row
 * (BG/OverlapGraph.cpp:698-708 use the destination and the orientation only). Needs n < 2^30. */
#define NBR32_MAKE(e) ((u32)ADJ_DST(e) | (ADJ_ORI(e) << 30))
#define NBR32_ENTRY(x) ADJ_MAKE(0u, (u64)((x)&0x3FFFFFFFu), (x) >> 30, 0u)
__global__ void __launch_bounds__(64) rows_gather32_kernel(const u64 *__restrict__ adj, const u64 *__restrict__ ref, u64 lo, u64 hi,
                                                           const u64 *__restrict__ dst_start, u32 *__restrict__ dst)
{
    for (u64 v = lo + blockIdx.x; v < hi; v += gridDim.x) {
        const u64 r = ref[v];
        const u32 d = REF_DEG(r);
        const u64 *src = adj + REF_POS(r);
        u32 *o = dst + dst_start[v - lo];
        for (u32 i = threadIdx.x; i < d; i += 64) o[i] = NBR32_MAKE(src[i]);
    }
}

/* binary search of key in the sorted row r[0..d) (flag bit ignored); returns index or -1 */
__device__ __forceinline__ int adj_find(const u64 *__restrict__ r, u32 d, u64 key)
{
    u32 lo = 0, hi = d;
    while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        u64 x = r[mid] & ~ADJ_FLAG;
        if (x < key) lo = mid + 1;
        else hi = mid;
    }
    return (lo < d && (r[lo] & ~ADJ_FLAG) == key) ? (int)lo : -1;
}

/* ================================================================================================================
 * twin check — insertEdge puts the twin of every find into the other read's list (BG/OverlapGraph.cpp:614-626).
 * For every entry (u -> w) with w in [lo,hi): its twin (w -> u) must be in the list of w; if not, the pair was found from
 * one side only (asymmetric) and the twin is appended to the extras of w.
 * ============================================================================================================== */
struct TwinArgs {
    DiscoView v;
    const u64 *ref; /* [n] */
    const u64 *adj;
    u64 lo, hi;     /* nodes whose lists are completed by this launch */
    const u8 *otab; /* multi-GPU flow, ranks own loci: ... AND whose owner (otab[w]) is `me`; null: the range alone */
    u32 me;
    u32 *extra_cnt; /* [n]                                            */
    u64 *extra_node;
    u64 *extra_key;
    u32 *n_extra;
    u32 extra_cap;
    const u64 *dropbits; /* or null. Bit w set: the selection of w dropped a verified hit. With exact overlaps the twin of a find
                            u -> w can be missing from w's list only then (twin_check in disco_hip.hip has the argument), so every
                            other find needs no search */
    int up_only;    /* 1: search only finds with src < dst and count both kinds; equality of the two counts plus no
                       missing twin proves symmetry (the up-finds inject into the down-finds); no extras recorded */
    int hidden_flags; /* inexact overlaps, dropbits given: verify_kernel told which finds the other read cannot see (ADJ_FLAG of the
                         entry, ADJ_HIDDEN_OF). A find it CAN see is in its list unless that read's selection dropped something —
                         the argument of the exact case plus an exact seed — so only hidden finds and finds into reads of the bitmap
                         are searched: a quarter of the entries. clear_adj_flags_kernel / the rebuilding merge remove the flags afterwards. */
};

#ifndef TW_PEND
#define TW_PEND 256 /* missing twins a wavefront collects in LDS before it appends them to the extras with ONE atomic */
#endif
__global__ void __launch_bounds__(256) twin_check_kernel(TwinArgs a)
{
    /* With inexact overlaps a third of all finds lack their twin (a substitution inside the other read's end k-mer hides the pair
     * from that side): one append per trip was 50 M atomics on a single address — 0.42 s of a 0.52 s launch at 50 M reads. The
     * misses of several trips are parked in LDS and appended TW_PEND at a time: the list stays dense, the counter sees 1/50th. */
    __shared__ u64 s_node[4][TW_PEND];
    __shared__ u64 s_key[4][TW_PEND];
    const u32 lane = threadIdx.x & 63;
    const u32 wv = threadIdx.x >> 6;
    const u64 wave = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const u64 nwaves = ((u64)gridDim.x * blockDim.x) >> 6;
    u32 asym = 0, n_up = 0, n_down = 0;
    u32 pend = 0; /* wave-uniform */
    auto flush = [&]() {
        if (!pend) return;
        u32 base = 0;
        if (lane == 0) base = atomicAdd(a.n_extra, pend);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
        for (u32 i0 = 0; i0 < pend; i0 += 64) {
            const u32 i = i0 + lane;
            const bool have = i < pend;
            const bool put = have && base + i < a.extra_cap;
            if (put) {
                const u64 w = s_node[wv][i];
                a.extra_node[base + i] = EXTRA_NODE_MAKE(w, atomicAdd(&a.extra_cnt[w], 1u)); /* its place among w's extras rides along */
                a.extra_key[base + i] = s_key[wv][i];
            }
            const u64 over = __ballot(have && !put); /* the sizing pass overflows by design */
            if (over && lane == 0) atomicAdd(&a.v.ctr[CTR_OVERFLOW], (u64)__popcll(over));
        }
        pend = 0;
    };
    for (u64 u = wave; u < a.v.n; u += nwaves) {
        const u64 ru = a.ref[u];
        const u32 du = REF_DEG(ru);
        if (du == 0) continue;
        const u64 s = REF_POS(ru);
        const u32 Lu = a.v.len[u];
        const bool u_in = (u >= a.lo && u < a.hi);
        for (u32 p0 = 0; p0 < du; p0 += 64) { /* wave-uniform trips */
            const u32 p = p0 + lane;
            bool search = p < du;
            const u64 raw = search ? a.adj[s + p] : 0ull;
            const u64 ent = raw & ~ADJ_FLAG;
            const bool hid = (raw & ADJ_FLAG) != 0; /* (left in place: the launch is repeated when the extras overflow) */
            const u64 w = ADJ_DST(ent);
            if (search && a.up_only) {
                if (w < u) { /* a down-find, counted at its source */
                    n_down += u_in ? 1u : 0u;
                    search = false;
                } else if (w < a.lo || w >= a.hi)
                    search = false;
                else
                    n_up++;
            } else if (search && (w < a.lo || w >= a.hi || (a.otab && a.otab[w] != (u8)a.me)))
                search = false;
            if (search && a.dropbits && !(a.hidden_flags && hid) && !((a.dropbits[w >> 6] >> (w & 63)) & 1ull)) search = false;
            bool miss = false;
            u64 twin = 0;
            if (search) {
                const u32 Lw = ADJ_DLEN(ent);
                twin = ADJ_MAKE(Lw + ADJ_OFF(ent) - Lu, u, disco_twin_orient(ADJ_ORI(ent)), Lu); /* :617-619 */
                const u64 rw = a.ref[w];
                miss = adj_find(a.adj + REF_POS(rw), REF_DEG(rw), twin) < 0;
            }
            asym += miss ? 1u : 0u;
            const bool put = miss && !a.up_only;
            const u64 mk = __ballot(put);
            if (mk) {
                const u32 cnt = (u32)__popcll(mk);
                if (pend + cnt > TW_PEND) flush();
                if (put) {
                    const u32 pos = pend + (u32)rank_below(mk);
                    s_node[wv][pos] = w;
                    s_key[wv][pos] = twin;
                }
                pend += cnt;
            }
        }
    }
    flush();
    for (int o = 32; o > 0; o >>= 1) {
        asym += __shfl_down(asym, o);
        n_up += __shfl_down(n_up, o);
        n_down += __shfl_down(n_down, o);
    }
    if (lane == 0) {
        if (asym) atomicAdd(&a.v.ctr[CTR_ASYM], (u64)asym);
        if (n_up) atomicAdd(&a.v.ctr[CTR_TW_UP], (u64)n_up);
        if (n_down) atomicAdd(&a.v.ctr[CTR_TW_DOWN], (u64)n_down);
    }
}

/* the same from edge selection's drop list (exact overlaps, one GPU): item = {w, the entry w -> u that w did not keep}. Its twin u -> w
 * in u's list <=> the pair is one-sided and the entry is owed to w. Thread per item. */
__global__ void __launch_bounds__(256) twin_from_drops_kernel(TwinArgs a, const u64 *__restrict__ drop_node, const u64 *__restrict__ drop_key, u32 n_drop)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_drop; i += gridDim.x * blockDim.x) {
        const u64 w = drop_node[i], e = drop_key[i];
        const u64 u = ADJ_DST(e);
        const u32 Lw = a.v.len[w];
        const u64 twin = ADJ_MAKE(ADJ_DLEN(e) + ADJ_OFF(e) - Lw, w, disco_twin_orient(ADJ_ORI(e)), Lw); /* u -> w */
        const u64 ru = a.ref[u];
        if (adj_find(a.adj + REF_POS(ru), REF_DEG(ru), twin) < 0) continue;
        atomicAdd(&a.v.ctr[CTR_ASYM], 1ull);
        const u32 idx = atomicAdd(a.n_extra, 1u);
        if (idx < a.extra_cap) {
            a.extra_node[idx] = EXTRA_NODE_MAKE(w, atomicAdd(&a.extra_cnt[w], 1u));
            a.extra_key[idx] = e;
        } else
            atomicAdd(&a.v.ctr[CTR_OVERFLOW], 1ull);
    }
}

/* inexact overlaps: the hidden flags of edge selection out of the rows again (ADJ_FLAG belongs to the transitive marking from here
 * on); 16 lanes per row. Not needed when the merge rebuilds every row. */
__global__ void __launch_bounds__(256) clear_adj_flags_kernel(const u64 *__restrict__ ref, u64 *__restrict__ adj, u64 n)
{
    const u64 g = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ng = ((u64)gridDim.x * blockDim.x) >> 4;
    const u32 sub = threadIdx.x & 15u;
    for (u64 v = g; v < n; v += ng) {
        const u64 r = ref[v];
        const u64 s = REF_POS(r);
        const u32 d = REF_DEG(r);
        for (u32 i = sub; i < d; i += 16) {
            const u64 x = adj[s + i];
            if (x & ADJ_FLAG) adj[s + i] = x & ~ADJ_FLAG;
        }
    }
}

/* extras merge (only when asymmetric pairs exist): new_deg = deg + extra_cnt -> scan -> copy rows -> scatter extras
 * -> re-sort the rows that received extras -> node-ordered CSR */
__global__ void merge_deg_kernel(const u64 *__restrict__ ref, const u32 *__restrict__ extra_cnt, u64 n, u32 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) out[i] = REF_DEG(ref[i]) + extra_cnt[i];
}

__global__ void merge_scatter_kernel(const u64 *__restrict__ extra_node, const u64 *__restrict__ extra_key, u32 n_extra,
                                     const u64 *__restrict__ old_ref, const u64 *__restrict__ new_start, u64 *__restrict__ new_adj)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_extra; i += gridDim.x * blockDim.x) {
        const u64 en = extra_node[i];
        const u64 w = EXTRA_NODE(en);
        new_adj[new_start[w] + REF_DEG(old_ref[w]) + EXTRA_SLOT(en)] = extra_key[i]; /* the slot was drawn when the extra was recorded */
    }
}

/* few extras (the ordinary case on real reads: a few repeats bind the cap): only the rows that receive extras move — each gets
 * 2 x (deg + extras) fresh slots behind the used part of the entry buffer: merged unsorted in the upper half, rank-sorted into the
 * lower half, which becomes the row. merge_need_kernel sizes the space first; the old rows are simply abandoned. */
__global__ void merge_need_kernel(const u64 *__restrict__ ref, const u32 *__restrict__ extra_cnt, u64 n, u64 *__restrict__ need)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (extra_cnt[i]) atomicAdd(need, 2ull * (REF_DEG(ref[i]) + extra_cnt[i]));
}

__global__ void __launch_bounds__(64) merge_sparse_kernel(const u64 *__restrict__ extra_node, const u64 *__restrict__ extra_key, u32 n_extra,
                                                          const u32 *__restrict__ extra_cnt, u64 *__restrict__ ref, u64 *__restrict__ adj, u64 free_base,
                                                          u64 *__restrict__ bump)
{
    const u32 lane = threadIdx.x;
    for (u32 i = blockIdx.x; i < n_extra; i += gridDim.x) {
        const u64 v = EXTRA_NODE(extra_node[i]);
        /* the node is handled by the wave that holds its FIRST extra */
        bool earlier = false;
        for (u32 t = lane; t < i; t += 64) earlier |= EXTRA_NODE(extra_node[t]) == v;
        if (__any(earlier)) continue;
        const u64 r = ref[v];
        const u32 d = REF_DEG(r), nd = d + extra_cnt[v];
        u64 p = 0;
        if (lane == 0) p = free_base + atomicAdd(bump, 2ull * nd);
        p = __shfl(p, 0);
        u64 *lo = adj + p, *up = adj + p + nd;
        for (u32 t = lane; t < d; t += 64) up[t] = adj[REF_POS(r) + t] & ~ADJ_FLAG;
        u32 at = d;
        for (u32 t0 = i; t0 < n_extra; t0 += 64) {
            const u32 t = t0 + lane;
            const bool mine = t < n_extra && EXTRA_NODE(extra_node[t]) == v;
            const u64 mk = __ballot(mine);
            if (mine) up[at + rank_below(mk)] = extra_key[t];
            at += __popcll(mk);
        }
        __syncthreads();
        wave_rank_sort(up, lo, nd, lane);
        __syncthreads();
        if (lane == 0) ref[v] = REF_MAKE(p, nd);
    }
}

/* the merged row of every node, after the extras were scattered behind the place of the old row: old entries (sorted) + extras (not)
 * -> sorted new row. Up to 64 entries stay in registers: every extra is broadcast once, its rank is one ballot, every old entry
 * moves down by the extras before it (keys are unique) — 6 instructions per extra where a sorting network takes 250 per row; rows
 * with many extras take the network, longer rows a global scratch row. Rows without extras are copied. */
#define MERGE_RANK_MAX 24
__global__ void __launch_bounds__(64) merge_rows_kernel(const u64 *__restrict__ old_ref, const u64 *__restrict__ old_adj, const u32 *__restrict__ extra_cnt,
                                                        const u64 *__restrict__ new_start, u64 *__restrict__ new_adj, u64 n, u64 *__restrict__ scratch,
                                                        u64 scratch_cap)
{
    u64 *tmp = scratch + (u64)blockIdx.x * scratch_cap;
    const u32 lane = threadIdx.x;
    for (u64 v = blockIdx.x; v < n; v += gridDim.x) {
        const u64 r = old_ref[v];
        const u64 so = REF_POS(r);
        const u32 d0 = REF_DEG(r);
        const u32 x = extra_cnt[v];
        const u64 o = new_start[v];
        const u32 d = d0 + x;
        if (d <= 64) {
            u64 e = ~0ull; /* entries are < ~0: the padding sorts behind them */
            if (lane < d0) e = old_adj[so + lane] & ~ADJ_FLAG;
            else if (lane < d) e = new_adj[o + lane];
            u32 pos = lane;
            if (x > MERGE_RANK_MAX) e = wave_bitonic_sort(e, lane);
            else
                for (u32 j = 0; j < x; j++) { /* wave-uniform trips */
                    const u64 bj = readlane_u64(e, d0 + j);
                    const u32 rank = (u32)__popcll(__ballot(e < bj));
                    if (lane == d0 + j) pos = rank;
                    else if (lane < d0 && bj < e) pos++;
                }
            if (lane < d) new_adj[o + pos] = e;
            continue;
        }
        for (u32 i = lane; i < d0; i += 64) new_adj[o + i] = old_adj[so + i] & ~ADJ_FLAG;
        if (!x) continue;
        __syncthreads();
        wave_rank_sort(new_adj + o, tmp, d, lane);
        __syncthreads();
        for (u32 i = lane; i < d; i += 64) new_adj[o + i] = tmp[i];
        __syncthreads();
    }
}

/* ================================================================================================================
 * transitive marking — markTransitiveEdges (BG/OverlapGraph.cpp:687-723), Myers 2005, one wavefront per node v.
 * N(v) goes into an LDS hash keyed by read id (the reference's markedNodes map, :689-691); neighbours are visited in
 * list order (ascending offset); for every still-INPLAY neighbour u the lanes sweep the list of u and ELIMINATE the w in
 * N(v) reachable by a consistent walk v->u->w (:701-708). Edges v->ELIMINATED get ADJ_FLAG (:713-720); the twin's flag is
 * the other node's business and is combined at emission.
 * ============================================================================================================== */
#define HALF_CAP 4 /* edges of a node that survive its own marking, kept aside for the emission (almost always 2) */
struct TrArgs {
    DiscoView v;
    const u64 *ref;
    u64 *adj;      /* only the flag bit of the node's own row is written */
    u64 *half;     /* [n][HALF_CAP] entries not flagged from this node (first HALF_CAP in list order), or null */
    u32 *hcnt;     /* [n] how many there are (may exceed HALF_CAP: the emission then reads the row) */
    u64 *wide_list; /* nodes with more than HALF_CAP survivors */
    u32 *n_wide;
    u32 wide_cap;
    u64 *big_list;
    u32 *n_big;
    u32 big_cap;
    u64 *scratch;  /* BIG variant: per block hkey[hcap] | ent(u32)[hcap] | state(u8)[hcap] */
    u64 hcap;      /* power of two >= 2 * max degree */
    /* DEFER variants (multi-GPU flow): ref / adj hold the rank's own rows and, behind them in the same array, the rows of other ranks'
     * nodes that were fetched on request from their owners (tr_request_*_kernel; whole rows, expanded to 8-byte entries on arrival:
     * rows_place_kernel) — ONE reference word per node and one kind of entry, exactly what the single-GPU kernel reads (rounds 2-5 kept
     * the fetched rows as 4-byte entries behind two reference words per node, nref[2u + cls]: the variant cost 5.5 ms more than the
     * plain kernel for the same nodes with ONE rank, profiles/r06_experiments.txt F). A node of another rank whose row was not fetched
     * has degree 0 in ref — no node that appears in a row has an empty row of its own (the lists are symmetric when the marking runs) —
     * and a node whose sweep needs such a row goes to big_list and is redone after the request-all round. */
    /* 0: the transitive flag is written into the rows of nodes with more than HALF_CAP survivors only — everybody else's result
     * IS its survivor list, and rewriting 34 of 36 entries per node was a quarter of this kernel's memory requests (single GPU
     * with survivor lists); 1: every row gets its flags (sharded flows: the flag exchange may need all of them) */
    u32 all_flags;
    /* the nodes are taken in the processing order of probe / verify / edge selection (reads that share their read-level minimizer back
     * to back) where the caller has one for the query range, or in id order (null): nodes of one locus sweep the same few rows */
    const u64 *order;
};

#define TR_UNAVAIL 0ull /* reference word of another rank's node: row not fetched (degree 0) */

#ifndef TR_HASH_LOAD
#define TR_HASH_LOAD 4 /* slots per neighbour in the marking hash of the register path: fewer probe-loop trips (each trip of a
                          divergent loop is a dozen scalar exec-mask instructions, and this kernel is bound by its SCALAR unit) */
#endif
static_assert(TR_HASH_LOAD * 64 + 1 < 4 * TR_CAP_SMALL && TR_CAP_SMALL <= TR_CAP, "the register path's table (at most TR_HASH_LOAD x 64 four-byte slots in the space of s_hkey) needs two more slots behind it: one takes the stores of the lanes without a hit, one holds the multi-rank variant's verdict on the node");
#define TR_EMPTY 0xFFFFFFFFFFFFFFFFull
/* slot of a node id (< 2^31) in the marking hash: one 32-bit multiply (disco_hash64 costs two 64-bit multiplies — eight
 * quarter-rate 32-bit ones — and is evaluated for every entry of every swept row) */
#ifdef TR_EXP_XOR_HASH /* timing experiment: two full-rate instructions instead of a quarter-rate 32-bit multiply and a shift */
__device__ __forceinline__ u32 tr_hash(u64 id, u32 hmask) { return ((u32)id ^ ((u32)id >> 9)) & hmask; }
#else
__device__ __forceinline__ u32 tr_hash(u64 id, u32 hmask) { return (((u32)id * 0x9E3779B1u) >> 10) & hmask; }
#endif

template <bool DEFER>
__device__ __forceinline__ void tr_node(const TrArgs &a, u64 v, u32 d, u64 *hkey, u8 *hstate, u32 *sent, u32 hmask, u32 lane)
{
    const u64 vs = REF_POS(a.ref[v]);
    u64 *row = a.adj + vs;
    for (u32 i = lane; i <= hmask; i += 64) {
        hkey[i] = TR_EMPTY;
        hstate[i] = 0;
    }
    __syncthreads();
    for (u32 s = lane; s < d; s += 64) { /* markedNodes->insert(dst, INPLAY) */
        u64 id = ADJ_DST(row[s]);
        u32 idx = tr_hash(id, hmask);
        for (;;) {
            u64 old = atomicCAS(&hkey[idx], TR_EMPTY, id);
            if (old == TR_EMPTY || old == id) break;
            idx = (idx + 1) & hmask;
        }
        sent[s] = idx;
    }
    __syncthreads();
    /* :693 list order, :696 only neighbours that are still INPLAY when their turn comes. States only go INPLAY ->
     * ELIMINATED, so the next slot to sweep is the first INPLAY slot after the current one judged with the states as they are
     * now: found with one ballot per 64 slots instead of polling every slot */
    for (u32 g0 = 0; g0 < d; g0 += 64) {
        int cur = -1;
        for (;;) {
            const u32 s = g0 + lane;
            u64 mk = __ballot(s < d && !hstate[sent[s]]);
            if (cur >= 0) mk &= ~((2ull << cur) - 1ull);
            if (!mk) break;
            cur = (int)__ffsll((long long)mk) - 1;
            const u32 i = g0 + (u32)cur;
            const u64 e1 = row[i];
            const u64 u = ADJ_DST(e1);
            const u32 type1 = ADJ_ORI(e1);
            const bool in1 = (type1 == 0 || type1 == 2); /* v enters u reversed */
            const u64 ru = a.ref[u];
            const u64 us = REF_POS(ru);
            u32 du = REF_DEG(ru);
            if (DEFER && du == 0) { /* cannot happen after the request-all round: fail loudly */
                if (lane == 0) atomicAdd(&a.v.ctr[CTR_OVERFLOW], 1ull);
            }
            for (u32 t = lane; t < du; t += 64) {        /* :698 */
                const u64 e2 = a.adj[us + t];
                const u32 type2 = ADJ_ORI(e2);
                const bool ok = in1 ? (type2 == 0 || type2 == 1) : (type2 == 2 || type2 == 3); /* :705-708 */
                if (!ok) continue;
                const u64 w = ADJ_DST(e2);
                u32 idx = tr_hash(w, hmask);
                for (;;) {
                    u64 kk = hkey[idx];
                    if (kk == TR_EMPTY) break;
                    if (kk == w) {
                        hstate[idx] = 1; /* ELIMINATED */
                        break;
                    }
                    idx = (idx + 1) & hmask;
                }
            }
            __syncthreads();
        }
    }
    u32 nfree = 0;
    for (u32 s0 = 0; s0 < d; s0 += 64) {
        const u32 s = s0 + lane;
        const bool fl = (s < d) && hstate[sent[s]];
        const bool fr = (s < d) && !fl;
        if (fl) row[s] |= ADJ_FLAG;
        const u64 mk = __ballot(fr);
        if (a.half && fr) {
            const u32 r = nfree + rank_below(mk);
            if (r < HALF_CAP) a.half[v * HALF_CAP + r] = row[s] & ~ADJ_FLAG;
        }
        nfree += __popcll(mk);
    }
    if (a.hcnt && lane == 0) {
        a.hcnt[v] = nfree;
        if (nfree > HALF_CAP) {
            const u32 idx = atomicAdd(a.n_wide, 1u);
            if (idx < a.wide_cap) a.wide_list[idx] = v;
        }
    }
    __syncthreads();
}

/* nodes of at most 64 neighbours (virtually all of them): the node's list lives in registers (lane = slot), the rows of the
 * two neighbours that are INPLAY for certain or almost certainly — the first of the list and the first on the other side of
 * v — are fetched speculatively together with the hash build, and the next node's list is fetched while this one is
 * processed. The sequential INPLAY/ELIMINATED logic is unchanged; a speculative row is simply not used if its neighbour
 * turns out to be eliminated. */
struct TrNodeRegs {
    u32 v; /* (ids are below 2^31: ADJ_DST) */
    u64 vs;
    u32 d;      /* 0: nothing to do or not a register-resident node (no prefetch) */
    u32 dfull;  /* the node's degree */
    u64 e;      /* lane's entry (lane < d) */
    u32 s2;     /* first slot on the other side of v */
    u64 r0, r2; /* reference words of the neighbours in slot 0 and s2 */
    u64 p0, p2; /* lane's entry of their rows */
};

/* LISTS: the survivor lists are the result (half and hcnt are there, all_flags is 0: what every pass over a whole table does) — known when
 * the kernel is compiled instead of asked of three arguments per node */
template <bool DEFER, bool LISTS>
__device__ __forceinline__ void tr_node_small(const TrArgs &a, const TrNodeRegs &nd, u64 *hkey, u8 *hstate, u32 lane)
{
    const u32 d = nd.d;
    const u64 e = nd.e;
    /* the table has TR_HASH_LOAD x 64 slots whatever the degree (d <= 64): its size and mask are immediates and it is cleared by ONE
     * 16-byte store per lane — sizing it by the degree was a scalar loop and a masked clearing loop per node, twenty scalar instructions
     * of a kernel that runs out of scalar issue (0.08 ms each at 50 M nodes) */
    constexpr u32 hc = TR_HASH_LOAD * 64;
    constexpr u32 hmask = hc - 1;
    static_assert(hc == 256, "one uint4 per lane clears the table");
    /* round 4: a slot is ONE 32-bit word — the node id (below 2^31) with the ELIMINATED state in bit 31, 0xFFFFFFFF = free — in the
     * space of hkey: the table is cleared by one 16-byte store per lane (8-byte keys and a byte array of states took eight stores), a
     * probe reads and a mark writes one word */
    u32 *ht = (u32 *)hkey;
    (void)hstate;
    ((uint4 *)ht)[lane] = make_uint4(~0u, ~0u, ~0u, ~0u);
    /* speculative rows (fetched by the pipeline of the kernel): slot 0 and the first slot on the other side of v */
    const u32 s2 = nd.s2;
    const u64 st0 = REF_POS(nd.r0), st2 = REF_POS(nd.r2);
    const u32 d0 = REF_DEG(nd.r0), d2 = REF_DEG(nd.r2);
    const u64 p0 = lane < d0 ? nd.p0 : 0ull;
    const u64 p2 = lane < d2 ? nd.p2 : 0ull;
    /* DEFER: a row that is not there has degree 0 (another rank's, not fetched); a node that sweeps one is redone after the request-all
     * round. The two speculative rows are judged here, once per node and whether the second will be swept or not (round 1 asked for
     * both), a third sweep's where it happens. The verdict lives in the LDS word behind the spare slot and is read back as a VECTOR value
     * where the results are written: a scalar flag that lives across the sweep loop, with an exit in front of the output, made this
     * variant 1.2 ms slower than the plain kernel for the same nodes (profiles/r06_experiments.txt F) */
    if (DEFER) ht[hc + 1] = (d0 == 0 || d2 == 0) ? 1u : 0u;
    __syncthreads();
    u32 sent = 0;
#if defined(TR_EXP) && TR_EXP == 2 /* timing experiment (results are wrong): the pipeline and the output alone — no hash, no sweeps */
    sent = lane;
    if (p0 != 0x123456789ull || p2 != 0x123456789ull) ht[lane] = 0u;
#else
    { /* markedNodes->insert(dst, INPLAY). Every lane inserts: the lanes beyond the list hold entry 0 (stage_row's clamped load) and find it there */
        const u32 id = (u32)ADJ_DST(e);
        u32 idx = tr_hash(id, hmask);
        for (;;) {
            const u32 old = atomicCAS(&ht[idx], 0xFFFFFFFFu, id);
            if (old == 0xFFFFFFFFu || old == id) break;
            idx = (idx + 1) & hmask;
        }
        sent = idx;
    }
#endif
    __syncthreads();
    /* BG/OverlapGraph.cpp:693-696: walk the list in order, sweeping only neighbours that are still INPLAY when their turn
     * comes. States only ever go INPLAY -> ELIMINATED, so "the next INPLAY slot after the one just swept, judged with the
     * states as they are now" is exactly the sequential loop — found with one ballot instead of one LDS read per slot. */
    const u64 dmask = d >= 64u ? ~0ull : (1ull << d) - 1ull; /* the list's lanes, as a scalar mask: folding lane < d into every vote is a vector compare, a select and a scalar and per vote */
    /* The votes of the probe loop are kept as SCALAR masks (pending lanes, hits, free slots): the loop condition is the scalar
     * and-not's own condition code, and the lanes without a hit store to the spare slot through a select on the hit mask itself
     * (written as one v_cndmask: the compiler has no way from a scalar mask back to a per-lane condition but shifts and compares).
     * Round 5's form — per-lane booleans and a vote per trip — was 18 instructions a trip, this one is 14; addresses are kept as byte
     * offsets (slot << 2). Every lane calls: the loop is wave-uniform (one trip for nearly every entry at four slots per neighbour).
     * want: BG/OverlapGraph.cpp:705-708 — v enters u reversed (types 0 and 2) -> u's entries of types 0/1 count (type >> 1 == 0), else 2/3 */
    auto mark = [&](u32 want, bool act, u64 e2) {
        const u32 type2 = ADJ_ORI(e2);
        u64 pm = __ballot(act && ((type2 >> 1) == want));
        const u32 w = (u32)ADJ_DST(e2); /* ids are below 2^31: the low word of a slot identifies the node, 0xFFFFFFFF = empty */
        u32 off = tr_hash(w, hmask) << 2;
        const u32 spare = hc << 2; /* slot hc is never a table slot: it takes the stores of the lanes without a hit */
        do {
            const u32 kk = *(const u32 *)((const u8 *)ht + off);
            const u64 hm = __ballot((kk & 0x7FFFFFFFu) == w) & pm; /* (a free slot's low bits are no id) */
            const u64 fm = __ballot(kk == 0xFFFFFFFFu);
            u32 woff;
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(woff) : "v"(spare), "v"(off), "s"(hm));
            *(u32 *)((u8 *)ht + woff) = kk | 0x80000000u; /* ELIMINATED */
            pm &= ~(hm | fm);
            off = (off + 4u) & (hmask << 2);
        } while (pm != 0ull);
    };
    auto sweep = [&](u32 want, u64 us, u32 du, u64 pre) { /* :698 ; the first 64 entries of the row are in registers */
        mark(want, lane < du, pre);
        if (du > 64)
            for (u32 t0 = 64; t0 < du; t0 += 64) {
                const bool act = t0 + lane < du;
                mark(want, act, act ? a.adj[us + t0 + lane] : 0ull);
            }
    };
    u64 todo = dmask; /* the slots behind the one swept last */
#if !(defined(TR_EXP) && TR_EXP >= 1)
    /* slot 0 is INPLAY when the walk starts (nothing has been eliminated yet): its sweep needs no vote */
    sweep(ADJ_ORI((u64)(u32)__builtin_amdgcn_readlane((int)(u32)e, 0)) & 1u, st0, d0, p0);
    todo = dmask & ~1ull;
    __syncthreads();
#endif
    for (;;) {
#if defined(TR_EXP) && TR_EXP >= 1 /* timing experiment (results are wrong): pipeline, hash build and output without the sweeps (2: without the hash too) */
        if (p0 != 0x123456789ull || p2 != 0x123456789ull) break;
#endif
        /* (every lane reads a slot: a read under the exec mask of lane < d is three scalar instructions per trip, and every scalar
         * instruction of this per-node path is 0.08 ms at 50 M nodes) */
        const u32 hs = ht[sent];
        const u64 mk = __ballot((hs >> 31) == 0u) & todo;
        if (!mk) break;
        const u32 i = (u32)__ffsll((long long)mk) - 1u;
        todo = dmask & (~1ull << i);
        const u64 e1 = readlane_u64(e, i);
        const u32 want = ADJ_ORI(e1) & 1u;
        /* two copies on purpose: the row fetched on the spot must be consumed inside its own branch, or the wait for it
         * lands on the common path and drains the kernel's prefetch pipeline */
        /* DEFER: a row that is not there (degree 0: another rank's, not fetched — a third sweep's, mostly) sweeps nothing; the node is
         * redone after the request-all round. No early exit: the loop keeps the shape of the single-GPU kernel */
        if (i == s2) sweep(want, st2, d2, p2);
        else {
            const u64 ru = a.ref[ADJ_DST(e1)];
            const u64 us = REF_POS(ru);
            const u32 du = REF_DEG(ru);
            if (DEFER && du == 0) ht[hc + 1] = 1u; /* a third sweep whose row was not requested */
            sweep(want, us, du, (lane < du) ? a.adj[us + lane] : 0ull);
        }
        __syncthreads();
    }
    const bool keep = !DEFER || ht[hc + 1] == 0u; /* (the same for every lane) */
    if (DEFER && !keep && lane == 0) {
        const u32 idx = atomicAdd(a.n_big, 1u);
        if (idx < a.big_cap) a.big_list[idx] = nd.v;
        else atomicAdd(&a.v.ctr[CTR_OVERFLOW], 1ull);
    }
#if defined(TR_EXP) && TR_EXP >= 1 /* (the experiments leave the first two entries as survivors: the output side as on real data) */
    const bool fl = lane < d && lane >= 2u && (ht[sent & 63u] != 0x12345u);
#else
    const u32 hs_end = ht[sent];
    const bool fl = keep & (lane < d) & ((hs_end >> 31) != 0u);
#endif
    const bool fr = keep & (lane < d) & !fl;
    const u64 mk = __ballot(fr);
    if (fl && ((!LISTS && (a.all_flags || !a.half)) || __popcll(mk) > HALF_CAP)) a.adj[nd.vs + lane] = e | ADJ_FLAG;
    if ((LISTS || a.half) && fr) {
        const u32 r = rank_below(mk);
        if (r < HALF_CAP) a.half[(u64)nd.v * HALF_CAP + r] = e;
    }
    if ((LISTS || a.hcnt) && lane == 0 && keep) {
        const u32 nfree = __popcll(mk);
        a.hcnt[nd.v] = nfree;
        if (nfree > HALF_CAP) {
            const u32 idx = atomicAdd(a.n_wide, 1u);
            if (idx < a.wide_cap) a.wide_list[idx] = nd.v;
        }
    }
    __syncthreads();
}

#ifndef TR_WAVES_PER_SIMD
#define TR_WAVES_PER_SIMD 6
#endif
template <bool BIG, bool DEFER, int CAP = TR_CAP, bool LISTS = false>
__global__ void __launch_bounds__(64, (CAP == TR_CAP_SMALL && !BIG) ? 8 : TR_WAVES_PER_SIMD) transitive_mark_kernel(TrArgs a)
{
    __shared__ u64 s_hkey[BIG ? 1 : 2 * CAP];
    __shared__ u32 s_ent[BIG ? 1 : CAP];
    __shared__ u8 s_state[BIG ? 1 : 2 * CAP];
    __shared__ u64 s_mid[(BIG || DEFER) ? 1 : 64]; /* the chunk's nodes beyond the register path (a chunk has at most 64 nodes) */
    const u32 lane = threadIdx.x;
    const u64 n_items = BIG ? (u64)min(*a.n_big, a.big_cap) : (a.v.q_hi - a.v.q_lo);
    u64 *hkey = s_hkey;
    u32 *sent = s_ent;
    u8 *hstate = s_state;
    if (BIG) {
        u8 *base = (u8 *)a.scratch + (u64)blockIdx.x * (a.hcap * 8 + a.hcap * 4 + a.hcap);
        hkey = (u64 *)base;
        sent = (u32 *)(base + a.hcap * 8);
        hstate = base + a.hcap * 8 + a.hcap * 4;
    }
    /* Software pipeline over the nodes of a chunk, one dependent load per stage and iteration:
     *   chunk start: ref[v] of all nodes of the chunk (one coalesced load),
     *   node t+3: its row, node t+2: ref of its two speculative neighbours (slot 0 and the first slot on the other side of
     *   v; side = strand of v in the edge, bit 1), node t+1: their rows, node t: the sweep.
     * Every pipelined load is unconditional (clamped address): a load under an exec-mask branch makes the number of loads in
     * flight unknown to the compiler, which then drains the pipeline at the next use. */
    const u32 n_last = (u32)(a.v.n - 1);
    const u32 lane8 = lane << 3;
    u64 cbeg = 0, cend = 0;
    u64 rv_chunk = 0;
    u32 v_chunk = 0;
    u32 ccnt = 0; /* nodes of the chunk */
    auto stage_row = [&](u32 t) { /* node t of the chunk (32-bit: 64-bit compares and selects are vector instructions or scalar pairs); needs rv_chunk */
        TrNodeRegs r;
        const bool ok = t < ccnt;
        const u32 tc = ok ? t : ccnt - 1u;
        const u64 rv = readlane_u64(rv_chunk, tc);
        r.v = (u32)__builtin_amdgcn_readlane((int)v_chunk, (int)tc);
        r.vs = REF_POS(rv);
        r.dfull = ok ? REF_DEG(rv) : 0u;
        r.d = (r.dfull <= 64) ? r.dfull : 0u;
        r.e = *(const u64 *)((const u8 *)(a.adj + r.vs) + (lane < r.d ? lane8 : 0u)); /* (the lane's byte offset is a constant register: no shift per load) */
        r.s2 = 0;
        r.r0 = r.r2 = r.p0 = r.p2 = 0;
        return r;
    };
    auto stage_refs = [&](TrNodeRegs &r) { /* needs r.e */
        r.e &= ~ADJ_FLAG; /* (lanes beyond the list hold the row's first entry — stage_row's clamped load: a real entry, read by nobody) */
        const u32 side0 = ADJ_ORI(readlane_u64(r.e, 0)) >> 1;
        const u64 om = __ballot((ADJ_ORI(r.e) >> 1) != side0); /* (no lane < d: the lanes beyond the list hold entry 0) */
        r.s2 = om ? (u32)__ffsll((long long)om) - 1u : 0u;
        /* (a node without entries holds whatever lies at its row's position — an entry of another row, or nothing at all behind the
         * last row: one 32-bit minimum keeps the look-up inside the table; rounds 4-5 compared and selected 64-bit values, six scalar
         * instructions per look-up, and every scalar instruction of this per-node path is 0.08 ms at 50 M nodes) */
        const u64 e0 = readlane_u64(r.e, 0), e2 = readlane_u64(r.e, r.s2);
        r.r0 = a.ref[min((u32)ADJ_DST(e0), n_last)];
        r.r2 = a.ref[min((u32)ADJ_DST(e2), n_last)];
    };
    auto stage_rows = [&](TrNodeRegs &r) { /* needs r.r0, r.r2 (broadcast loads: scalar from here on) */
        r.r0 = uniform_u64(r.r0);
        r.r2 = uniform_u64(r.r2);
        const u32 d0 = REF_DEG(r.r0), d2 = REF_DEG(r.r2);
        r.p0 = *(const u64 *)((const u8 *)(a.adj + REF_POS(r.r0)) + (lane < d0 ? lane8 : 0u)); /* (d0 = 0: the first entry of whatever row the word names — a valid address, never used) */
        r.p2 = *(const u64 *)((const u8 *)(a.adj + REF_POS(r.r2)) + (lane < d2 ? lane8 : 0u));
    };
#if defined(WQ_SPLIT_ALL)
    WqSplit wqs;
    while (BIG ? wq_grab<1>(a.v.wq, n_items, cbeg, cend) : wq_grab_split(a.v.wq, n_items, cbeg, cend, wqs)) { /* big nodes: one per grab */
#else
    while (BIG ? wq_grab<1>(a.v.wq, n_items, cbeg, cend) : wq_grab(a.v.wq, n_items, cbeg, cend)) { /* big nodes: one per grab */
#endif
    if (BIG) {
        for (u64 it = cbeg; it < cend; it++) {
            const u64 v = a.big_list[it];
            const u32 d = REF_DEG(a.ref[v]);
            if (d == 0) continue;
            tr_node<DEFER>(a, v, d, hkey, hstate, sent, (u32)a.hcap - 1, lane);
        }
        continue;
    }
    {
        const u64 idx = cbeg + lane, ic = idx < cend ? idx : cend - 1;
        v_chunk = (u32)(a.order ? ORDER_ID(a.order[ic]) : a.v.q_lo + ic);
        rv_chunk = a.ref[v_chunk];
    }
    ccnt = (u32)(cend - cbeg);
    TrNodeRegs n0 = stage_row(0), n1 = stage_row(1), n2 = stage_row(2), n3;
    stage_refs(n0);
    stage_refs(n1);
    stage_rows(n0);
    /* one step: node A is processed while B's speculative rows, C's reference words and D's row (node it + 3) are fetched. The four
     * register sets take the four roles in turn — four copies of the step per trip of the loop — instead of being handed down the line
     * after every node (n0 = n1, n1 = n2, n2 = n3: some twenty-five moves per node, most of them scalar, in a kernel where a scalar
     * instruction per node is 0.08 ms) */
    u32 n_mid = 0; /* wave uniform */
    auto step = [&](TrNodeRegs &A, TrNodeRegs &B, TrNodeRegs &C, TrNodeRegs &D, u32 t) {
        stage_rows(B);
        stage_refs(C);
        D = stage_row(t + 3);
        if (A.d != 0)
            tr_node_small<DEFER, LISTS>(a, A, s_hkey, s_state, lane);
        else if (A.dfull != 0) {
            if (!DEFER && A.dfull <= (u32)CAP) { /* (multi-GPU: every node beyond the register path waits for the request-all round) */
                if (lane == 0) s_mid[n_mid] = A.v; /* the LDS arrays' path: behind the chunk's loop, ONE copy of it */
                n_mid++;
            } else if (lane == 0) {
                u32 idx = atomicAdd(a.n_big, 1u);
                if (idx < a.big_cap) a.big_list[idx] = A.v;
                else atomicAdd(&a.v.ctr[CTR_OVERFLOW], 1ull);
            }
        }
    };
    for (u32 t = 0; t < ccnt; t += 4) { /* (nodes beyond the chunk's end have degree 0: stage_row) */
        step(n0, n1, n2, n3, t);
        step(n1, n2, n3, n0, t + 1);
        step(n2, n3, n0, n1, t + 2);
        step(n3, n0, n1, n2, t + 3);
    }
    if (!DEFER && n_mid) { /* nodes of 65 .. CAP neighbours (wave uniform) */
        __syncthreads();
        for (u32 x = 0; x < n_mid; x++) {
            const u64 v = s_mid[x];
            const u32 dv = REF_DEG(a.ref[v]);
            u32 hc = 64;
            while (hc < 2 * dv) hc <<= 1;
            tr_node<DEFER>(a, v, dv, hkey, hstate, sent, hc - 1, lane);
        }
        __syncthreads();
    }
    }
}

/* ================================================================================================================
 * emission — removeTransitiveEdges (BG/OverlapGraph.cpp:731-761) + the canonical side of saveParGraphToFile (:808):
 * edge (v,w) with v < w survives iff it is flagged from neither end. One pass: survivors are appended to the output
 * through wave-private chunks of a global bump pointer (order is irrelevant: the canonical form is sorted, and the
 * reference's own file order depends on its BFS).
 * ============================================================================================================== */
#define EMIT_CHUNK 256
struct EmitArgs {
    DiscoView v;
    const u64 *ref;
    const u64 *adj;
    const u32 *hcnt; /* non-null: only nodes with more than HALF_CAP survivors (the others went through emit_half_kernel) */
    const u64 *half; /* with hcnt: survivor lists; a neighbour with at most HALF_CAP survivors is judged by its list (its row
                        carries no flags then, see TrArgs.all_flags) */
    const u64 *list; /* non-null: the nodes to emit (n_list of them) instead of the whole query range */
    u64 n_list;
    u64 *out_src;
    u64 *out_ent;
    u64 out_cap;
    u64 *bump;
    /* multi-GPU flow: only pairs whose larger endpoint is owned by this rank too are judged here (the survivors of a remote w
     * are not on this rank: its owner pushes them, emit_push_recv_kernel) */
    u32 local_only;
    OwnSet own; /* the nodes this launch emits from: the query range, or the rank's own nodes */
};

__global__ void __launch_bounds__(64) emit_kernel(EmitArgs a)
{
    const u32 lane = threadIdx.x;
    u64 chunk_base = 0;
    u32 chunk_used = EMIT_CHUNK; /* no chunk yet */
    bool have_chunk = false;
    /* slots of a chunk that stay unused are marked ~0 so that the compaction can drop them */
    auto close_chunk = [&]() {
        if (have_chunk)
            for (u32 i = chunk_used + lane; i < EMIT_CHUNK; i += 64)
                if (chunk_base + i < a.out_cap) a.out_src[chunk_base + i] = ~0ull;
    };
    u64 cbeg = 0, cend = 0;
    while (wq_grab(a.v.wq, a.list ? a.n_list : a.own.count(), cbeg, cend))
    for (u64 it = cbeg; it < cend; it++) {
        const u64 v = a.list ? a.list[it] : a.own.node(it);
        /* the list can name nodes of other ranks' ranges: the order-dependent regime of the multi-GPU flow marks ALL nodes on every
         * rank (and lists their wide ones) but emits its own range only */
        if (!a.own.mine(v)) continue;
        if (a.hcnt && a.hcnt[v] <= HALF_CAP) continue;
        const u64 rv = a.ref[v];
        const u32 d = REF_DEG(rv);
        if (d == 0) continue;
        const u64 vs = REF_POS(rv);
        const u32 Lv = a.v.len[v];
        for (u32 s0 = 0; s0 < d; s0 += 64) {
            const u32 s = s0 + lane;
            bool keep = false;
            u64 e = 0;
            if (s < d) {
                e = a.adj[vs + s];
                const u64 w = ADJ_DST(e);
                if (v < w && !(e & ADJ_FLAG) && (!a.local_only || a.own.mine(w))) {
                    const u32 Lw = ADJ_DLEN(e);
                    const u64 twin = ADJ_MAKE(Lw + ADJ_OFF(e) - Lv, v, disco_twin_orient(ADJ_ORI(e)), Lv);
                    const u32 cw = a.hcnt ? a.hcnt[w] : HALF_CAP + 1;
                    if (cw <= HALF_CAP) {
                        const u64 *hw = a.half + w * HALF_CAP;
                        for (u32 r = 0; r < cw; r++) keep |= (hw[r] == twin);
                    } else {
                        const u64 rw = a.ref[w];
                        const u64 *roww = a.adj + REF_POS(rw);
                        const int ti = adj_find(roww, REF_DEG(rw), twin);
                        keep = (ti >= 0) && !(roww[ti] & ADJ_FLAG);
                    }
                }
            }
            const u64 mk = __ballot(keep);
            const u32 cnt = __popcll(mk);
            if (cnt) {
                if (chunk_used + cnt > EMIT_CHUNK) {
                    close_chunk();
                    u64 base = 0;
                    if (lane == 0) base = atomicAdd(a.bump, (u64)EMIT_CHUNK);
                    chunk_base = uniform_u64(base);
                    chunk_used = 0;
                    have_chunk = true;
                }
                if (keep) {
                    const u64 pos = chunk_base + chunk_used + rank_below(mk);
                    if (pos < a.out_cap) {
                        a.out_src[pos] = v;
                        a.out_ent[pos] = e & ~ADJ_FLAG;
                    }
                }
                chunk_used += cnt;
            }
        }
    }
    close_chunk();
}

/* Emission from the half-edge lists (single GPU): lane = node. A node's few edges that survived its own marking sit in
 * half[v][0..hcnt) (written by transitive_mark_kernel); the edge (v,w), v < w, survives iff its twin is among the
 * survivors of w as well — one 32-byte gather instead of a binary search through the row of w. Nodes with more than
 * HALF_CAP survivors (and neighbours of such nodes) take the row path: `wide` nodes are left to emit_kernel. */
struct EmitHalfArgs {
    DiscoView v;
    const u64 *ref;
    const u64 *adj;
    const u64 *half;
    const u32 *hcnt;
    u64 *out_src;
    u64 *out_ent;
    u64 out_cap;
    u64 *bump;
    u32 local_only; /* see EmitArgs */
    OwnSet own;
    /* one GPU: the nodes in the processing order of the pass (or null: ascending id). A node's two or three surviving neighbours are
     * reads of its own locus: in this order their survivor lists — the 32-byte gathers of this kernel — are lists the neighbouring
     * lanes and the wavefronts next door touch too */
    const u64 *order;
};

__global__ void __launch_bounds__(64) emit_half_kernel(EmitHalfArgs a)
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
    const u64 nq = a.own.count();
    /* a work item = 64 consecutive nodes */
    u64 cbeg = 0, cend = 0;
    while (wq_grab(a.v.wq, (nq + 63) / 64, cbeg, cend))
    for (u64 blk = cbeg; blk < cend; blk++) {
        const bool live = blk * 64 + lane < nq;
        const u64 v = a.order ? ORDER_ID(a.order[live ? blk * 64 + lane : 0]) : a.own.node_by_id(live ? blk * 64 + lane : 0);
        const u32 cnt = live ? a.hcnt[v] : 0u;
        const u32 Lv = live ? (u32)a.v.len[v] : 0u;
#pragma unroll
        for (u32 r = 0; r < HALF_CAP; r++) {
            bool keep = false;
            u64 e = 0;
            if (cnt <= HALF_CAP && r < cnt) { /* wide nodes (cnt > HALF_CAP) are emitted by emit_kernel */
                e = a.half[v * HALF_CAP + r];
                const u64 w = ADJ_DST(e);
                /* (local_only, ranks own loci: no look at the owner table here — one more random fetch per edge — because hcnt is zero for
                 * every node of another rank (cleared before the marking, written for the own nodes only): no survivor, no edge; the
                 * owner of w pushes the pair instead. Id ranges: the range test costs nothing) */
                if (v < w && (!a.local_only || a.own.otab || w < a.own.hi)) {
                    const u64 twin = ADJ_MAKE(ADJ_DLEN(e) + ADJ_OFF(e) - Lv, v, disco_twin_orient(ADJ_ORI(e)), Lv);
                    const u32 cw = a.hcnt[w];
                    if (cw <= HALF_CAP) {
                        const ulonglong2 *hw = (const ulonglong2 *)(a.half + w * HALF_CAP);
                        const ulonglong2 h0 = hw[0], h1 = hw[1];
                        keep = (cw > 0 && h0.x == twin) || (cw > 1 && h0.y == twin) || (cw > 2 && h1.x == twin) || (cw > 3 && h1.y == twin);
                    } else { /* neighbour with many survivors: look the twin up in its row */
                        const u64 rw = a.ref[w];
                        const u64 *roww = a.adj + REF_POS(rw);
                        const int ti = adj_find(roww, REF_DEG(rw), twin);
                        keep = (ti >= 0) && !(roww[ti] & ADJ_FLAG);
                    }
                }
            }
            const u64 mk = __ballot(keep);
            const u32 kc = __popcll(mk);
            if (kc) {
                if (chunk_used + kc > EMIT_CHUNK) {
                    close_chunk();
                    u64 base = 0;
                    if (lane == 0) base = atomicAdd(a.bump, (u64)EMIT_CHUNK);
                    chunk_base = uniform_u64(base);
                    chunk_used = 0;
                    have_chunk = true;
                }
                if (keep) {
                    const u64 pos = chunk_base + chunk_used + rank_below(mk);
                    if (pos < a.out_cap) {
                        a.out_src[pos] = v;
                        a.out_ent[pos] = e;
                    }
                }
                chunk_used += kc;
            }
        }
    }
    close_chunk();
}

/* compact the chunked emission (drop the ~0 tails): count + gather are done on the host side of fetch via a scan */
__global__ void emit_valid_kernel(const u64 *__restrict__ out_src, u64 n, u8 *__restrict__ valid)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) valid[i] = out_src[i] != ~0ull;
}

/* the copy-out form: 12 bytes per edge (the source as 32 bits) */
__global__ void emit_compact32_kernel(const u64 *__restrict__ out_src, const u64 *__restrict__ out_ent, const u8 *__restrict__ valid,
                                      const u64 *__restrict__ pos, u64 n, u32 *__restrict__ dst_src, u64 *__restrict__ dst_ent)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (valid[i]) {
            dst_src[pos[i]] = (u32)out_src[i];
            dst_ent[pos[i]] = out_ent[i];
        }
}

__global__ void emit_compact_kernel(const u64 *__restrict__ out_src, const u64 *__restrict__ out_ent, const u8 *__restrict__ valid,
                                    const u64 *__restrict__ pos, u64 n, u64 *__restrict__ dst_src, u64 *__restrict__ dst_ent)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (valid[i]) {
            dst_src[pos[i]] = out_src[i];
            dst_ent[pos[i]] = out_ent[i];
        }
}

/* substitutions of the overlap every emitted edge stands for (the third column of an edge line; 0 by construction unless the
 * inexact mode is on), from the edge's geometry: orient 2,3 — string2 starts at `offset` of the source; orient 0,1 — string2 ends
 * where the first len_src - offset bases of the source end; string2 = the destination (0,3) or its reverse complement (1,2)
 * (BG/Edge.h:30-34, BG/OverlapGraph.cpp:614-626,660-666) */
__global__ void edge_subs_kernel(const u64 *__restrict__ out_src, const u64 *__restrict__ out_ent, const u8 *__restrict__ valid,
                                 const u64 *__restrict__ pos, u64 n, const u64 *__restrict__ reads, const u16 *__restrict__ len, int S,
                                 u16 *__restrict__ subs)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (valid[i]) {
            const u64 src = out_src[i], e = out_ent[i];
            const int l1 = (int)len[src], l2 = (int)ADJ_DLEN(e), off = (int)ADJ_OFF(e), ovl = l1 - off;
            const u32 o = ADJ_ORI(e);
            const u32 rev = (o == 1 || o == 2);
            subs[pos[i]] = (u16)seg_mismatches<false>(reads + src * S, reads + ADJ_DST(e) * S, S, l2, o >= 2 ? off : 0, o >= 2 ? 0 : l2 - ovl, ovl, rev);
        }
}

/* ================================================================================================================
 * file partition of the emitted edges by connected component (disco_partition_edges): the consumer pre-simplifies every
 * edge file on its own and may only touch nodes ALL of whose edges are in that file (SG/OverlapGraphSimple.cpp:344,
 * 636-644). The reference gets such files from the locality of its BFS batches; here the components of the reduced graph
 * (concurrent union-find: hook the larger root under the smaller with a CAS, path halving) are dealt out to the files, so
 * that every node has all its edges in one file.
 * ============================================================================================================== */
/* parent pointers are read and written by other workgroups DURING the kernel: plain loads may be served from this CU's vector
 * L1 or this XCD's L2 for ever (neither is refreshed by other CUs' stores), and a wave retrying its CAS on a stale "root"
 * would spin — every access goes to the coherence point (agent-scope atomics) */
__device__ __forceinline__ u32 uf_load(const u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ u32 uf_find(u32 *parent, u32 x)
{
    for (;;) {
        const u32 p = uf_load(&parent[x]);
        if (p == x) return x;
        const u32 g = uf_load(&parent[p]);
        if (g != p) __hip_atomic_store(&parent[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* path halving; parents only ever decrease, so a lost race is harmless */
        x = p;
    }
}

__global__ void uf_init_kernel(u32 *parent, u64 n)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) parent[i] = (u32)i;
}

__global__ void uf_hook_kernel(const u64 *__restrict__ out_src, const u64 *__restrict__ out_ent, const u8 *__restrict__ valid, u64 n_slots, u32 *parent)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_slots; i += (u64)gridDim.x * blockDim.x) {
        if (!valid[i]) continue;
        u32 a = (u32)out_src[i], b = (u32)ADJ_DST(out_ent[i]);
        for (;;) {
            a = uf_find(parent, a);
            b = uf_find(parent, b);
            if (a == b) break;
            if (a < b) {
                const u32 t = a;
                a = b;
                b = t;
            }
            if (atomicCAS(&parent[a], a, b) == a) break; /* a was still a root: now it hangs under the smaller root */
        }
    }
}

__global__ void uf_compress_kernel(u32 *parent, u64 n)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) {
        u32 r = (u32)i;
        while (parent[r] != r) r = parent[r];
        parent[i] = r;
    }
}

/* edges per component (at its root); parent is fully compressed */
/* round 6: through a table in LDS. A genome of a few dozen contigs reduces to a few dozen GIANT components, and an atomic per edge on a few
 * dozen addresses is the serialisation this code base keeps finding (12 ns per atomic on one address): 64 ms for the 45 M edges of config 3,
 * a fifth of everything the stage does behind the graph. A block counts its edges per root in a 1024-slot table (compare-and-swap on the
 * key, add on the count: LDS atomics) and adds each slot to the global count once; a root that finds no free slot within eight probes — a
 * block that meets more than a few hundred components: a metagenome's many small ones, which contend for nothing — goes to the global count
 * directly, as before. */
#define UF_SLOTS 1024u
__global__ void __launch_bounds__(256) uf_count_kernel(const u64 *__restrict__ out_src, const u8 *__restrict__ valid, u64 n_slots, const u32 *__restrict__ parent, u32 *cnt)
{
    __shared__ u32 s_key[UF_SLOTS], s_val[UF_SLOTS];
    for (u32 x = threadIdx.x; x < UF_SLOTS; x += blockDim.x) {
        s_key[x] = 0xFFFFFFFFu;
        s_val[x] = 0u;
    }
    __syncthreads();
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_slots; i += (u64)gridDim.x * blockDim.x)
        if (valid[i]) {
            const u32 root = parent[(u32)out_src[i]];
            u32 idx = (root * 0x9E3779B1u) >> 22; /* 10 bits */
            bool placed = false;
            for (int t = 0; t < 8 && !placed; t++) {
                const u32 old = atomicCAS(&s_key[idx], 0xFFFFFFFFu, root);
                if (old == 0xFFFFFFFFu || old == root) {
                    atomicAdd(&s_val[idx], 1u);
                    placed = true;
                } else
                    idx = (idx + 1u) & (UF_SLOTS - 1u);
            }
            if (!placed) atomicAdd(&cnt[root], 1u);
        }
    __syncthreads();
    for (u32 x = threadIdx.x; x < UF_SLOTS; x += blockDim.x)
        if (s_val[x]) atomicAdd(&cnt[s_key[x]], s_val[x]);
}

/* components of at least thr edges -> list (dealt out by size on the host); cfile[root] = 0xFFFF: "by hash" */
__global__ void uf_big_kernel(const u32 *__restrict__ cnt, u64 n, u32 thr, u64 *list, u32 *n_list, u32 cap)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x)
        if (cnt[i] >= thr) {
            const u32 k = atomicAdd(n_list, 1u);
            if (k < cap) list[k] = (i << 32) | cnt[i];
        }
}

__global__ void uf_assign_kernel(const u64 *__restrict__ pairs, u32 n_pairs, u16 *cfile)
{
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pairs) cfile[pairs[i] >> 32] = (u16)(pairs[i] & 0xFFFF);
}

__global__ void uf_edge_file_kernel(const u64 *__restrict__ out_src, const u8 *__restrict__ valid, const u64 *__restrict__ pos, u64 n_slots,
                                    const u32 *__restrict__ parent, const u16 *__restrict__ cfile, u32 n_files, u16 *edge_file)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_slots; i += (u64)gridDim.x * blockDim.x)
        if (valid[i]) {
            const u32 r = parent[(u32)out_src[i]];
            u16 f = cfile[r];
            if (f == 0xFFFFu) f = (u16)(((u64)(r * 0x9E3779B1u) * n_files) >> 32); /* small components: spread by hash */
            edge_file[pos[i]] = f;
        }
}

/* the query range grouped by read-level minimizer: count per hash value (the atomic hands every read its slot; the 32 MB of
 * counters stay in the L2 / Infinity Cache), exclusive scan, then order[start[hash] + slot] = read */
/* ================================================================================================================
 * processing order of the probe and verify passes. Reads are grouped by their READ-LEVEL MINIMIZER: the smallest order
 * hash among all m-mers of the read. Two reads with the same key contain the same genome m-mer (up to hash collisions), so
 * they overlap each other: a group is ~13 reads at 30x, laid over ~2 read lengths of genome, and its members look up the
 * same index buckets and gather the same candidate rows. Walking the groups one after the other turns most of those
 * random fetches into cache hits. The key keeps all 32 bits of the hash: the smallest of ~128 hashes lies in the lowest
 * 1/128 of the range, so the 23 bits the minimizer order keeps would leave ~10^5 distinct keys and lump unrelated loci together.
 * The keys come out of index_count_kernel's rolling pass over every read (okey[read]).
 * The order changes no result: every consumer is order independent per read (rows are keyed by read id).
 * ============================================================================================================== */
__global__ void order_count_kernel(const u32 *__restrict__ okey, u64 nq, u32 shift, u32 *__restrict__ cnt, u32 *__restrict__ oslot)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nq; i += (u64)gridDim.x * blockDim.x) oslot[i] = atomicAdd(&cnt[ORDER_BUCKET(okey[i], shift)], 1u);
}

__global__ void order_scatter_kernel(const u32 *__restrict__ okey, const u32 *__restrict__ oslot, const u32 *__restrict__ start, u32 shift, u64 lo, u64 nq,
                                     const u16 *__restrict__ len, u64 *__restrict__ order)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nq; i += (u64)gridDim.x * blockDim.x) order[(u64)start[ORDER_BUCKET(okey[i], shift)] + oslot[i]] = ORDER_MAKE(lo + i, len[lo + i]);
}

/* a caller's order (plain read ids) in the packed form the kernels walk */
__global__ void order_pack_kernel(const u64 *__restrict__ ids, u64 nq, const u16 *__restrict__ len, u64 *__restrict__ order)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < nq; i += (u64)gridDim.x * blockDim.x) order[i] = ORDER_MAKE(ids[i], len[ids[i]]);
}

/* how many items of [lo,hi) exceed a threshold: rows longer than ES_CAP (cnt = row_cnt) / nodes of degree above TR_CAP
 * (ref != null: degree field of the reference word) — sizes the big-item lists before the kernels that fill them */
__global__ void count_above_kernel(const u32 *__restrict__ cnt, const u64 *__restrict__ ref, u64 lo, u64 hi, u32 thr, u64 *out)
{
    u64 i = lo + (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 c = 0;
    for (; i < hi; i += (u64)gridDim.x * blockDim.x) c += (ref ? REF_DEG(ref[i]) : cnt[i]) > thr;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (u64)c);
}

/* streaming copy, 16 bytes per lane per iteration (bandwidth probe of disco_measure_hbm) */
__global__ void __launch_bounds__(256) stream_copy_kernel(const ulonglong2 *__restrict__ src, ulonglong2 *__restrict__ dst, u64 n)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 step = (u64)gridDim.x * blockDim.x;
    for (; i + 3 * step < n; i += 4 * step) {
        const ulonglong2 x0 = src[i], x1 = src[i + step], x2 = src[i + 2 * step], x3 = src[i + 3 * step];
        dst[i] = x0;
        dst[i + step] = x1;
        dst[i + 2 * step] = x2;
        dst[i + 3 * step] = x3;
    }
    for (; i < n; i += step) dst[i] = src[i];
}

/* random 64-byte row gather (bandwidth probe of disco_measure_gather): every lane fetches `per_lane` rows, four at a time */
__global__ void __launch_bounds__(256) gather_rows_kernel(const ulonglong2 *__restrict__ tab, u64 nrows, u32 per_lane, u64 *__restrict__ sink)
{
    const u64 tid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u64 acc = 0;
    for (u32 it = 0; it < per_lane; it += 4) {
        ulonglong2 v[4][4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const u64 r = disco_hash64(tid * per_lane + it + u) % nrows;
            const ulonglong2 *g = tab + r * 4;
            v[u][0] = g[0];
            v[u][1] = g[1];
            v[u][2] = g[2];
            v[u][3] = g[3];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= v[u][0].x ^ v[u][1].y ^ v[u][2].x ^ v[u][3].y;
    }
    if (acc == 0x1234567ull) sink[0] = acc; /* keeps the loads alive */
}

__global__ void max_u32_kernel(const u32 *__restrict__ p, u64 n, u64 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 m = 0;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) m = max(m, p[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_down(m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, (u64)m);
}

__global__ void iota_u64_kernel(u64 *p, u64 n)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = i;
}

__global__ void fill_u64_kernel(u64 *p, u64 n, u64 val)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) p[i] = val;
}

#endif /* DISCO_KERNELS_H_ */
