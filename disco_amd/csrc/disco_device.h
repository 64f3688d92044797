/*
 * disco_device.h — device-side primitives on 2-bit packed reads (gfx950, wave64).
 *
 * Packing: 32 bases per 64-bit word, base i of a read at bits 62-2(i mod 32) of word i/32, A0 C1 G2 T3
 * (reference layout: /root/reference/src/BuildGraph/src/HashTable.cpp:456-477, HashTable.h:16-24).
 * Unused bits of the last word and unused words of the fixed-stride row are zero.
 */
#ifndef DISCO_DEVICE_H_
#define DISCO_DEVICE_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;
typedef unsigned char u8;

#define DISCO_WAVE 64
/* 'not contained': largest value that is also positive as int64, so that a signed all-reduce(MIN) orders keys correctly */
#define DISCO_NOKEY 0x7FFFFFFFFFFFFFFFull

/* ---- index entry (8 bytes): key fingerprint | read id | minimizer offset t | record strand | isSuffix | length ---- */
/* t = offset of the end k-mer's minimizer inside the CANONICAL orientation of that k-mer (0 .. k-m <= 63).           */
/* The fingerprint (low 10 bits of the minimizer hash; the bucket uses the high bits) only prunes the records of other */
/* keys that share the bucket: every candidate is exactly re-checked by verify_kernel.                                 */
#define PAY_LEN(p) ((u32)((p)&0x7FFFu))
#define PAY_SUFFIX(p) ((u32)(((p) >> 15) & 1u))
#define PAY_REV(p) ((u32)(((p) >> 16) & 1u))
#define PAY_T(p) ((u32)(((p) >> 17) & 0x3Fu))
#define PAY_ID(p) (((p) >> 23) & 0x7FFFFFFFull)
#define PAY_FP(p) ((u32)((p) >> 54))
#define KEY_FP(key) ((u32)((key)&0x3FFu))
#define PAY_MAKE(key, id, t, rev, suf, len) \
    (((u64)KEY_FP(key) << 54) | ((u64)(id) << 23) | ((u64)(t) << 17) | ((u64)(rev) << 16) | ((u64)(suf) << 15) | (u64)(len))

/* ---- raw verified overlap hit: sorts numerically into the reference's consumption order (j, bucket order) ------- */
/* bucket order = ascending read id, prefix record before suffix record (BG/HashTable.cpp:451-454,486-489).          */
/* The length of the hit read rides along in the low bits (a function of id, so it never changes the order).         */
#define HIT_MAKE(j, id, suf, rev, len) (((u64)(j) << 48) | ((u64)(id) << 17) | ((u64)(suf) << 16) | ((u64)(rev) << 15) | (u64)(len))
#define HIT_J(h) ((u32)((h) >> 48) & 0x7FFFu)
#define HIT_ID(h) (((h) >> 17) & 0x7FFFFFFFull)
#define HIT_SUFFIX(h) ((u32)(((h) >> 16) & 1u))
#define HIT_REV(h) ((u32)(((h) >> 15) & 1u))
#define HIT_LEN(h) ((u32)((h)&0x7FFFu))
/* bit 63 of a VERIFIED hit, inexact overlaps only: the other read cannot find this pair from its side (a substitution inside this
 * read's end k-mer there). Edge selection carries it into the entry's ADJ_FLAG bit, the twin search reads and clears it. Hits are
 * consumed in ascending order of the other 63 bits: HIT_SORT_KEY moves the flag below them, HIT_FROM_SORT_KEY undoes it. */
#define HIT_HIDDEN_BIT (1ull << 63)
#define HIT_HIDDEN(h) ((u32)((h) >> 63))
#define HIT_SORT_KEY(h) (((h) << 1) | ((h) >> 63))
#define HIT_FROM_SORT_KEY(x) (((x) >> 1) | ((x) << 63))

/* ---- adjacency entry: sorts numerically by (offset, dst, orient) = list order of BG/OverlapGraph.cpp:675-676 ---- */
/* offset(15) | dst(31) | orient(2) | len(dst)(15): the destination's length rides along so that the twin's offset     */
/* (BG/OverlapGraph.cpp:617) needs no extra gather.                                                                    */
#define ADJ_MAKE(off, dst, o, dlen) (((u64)(off) << 49) | ((u64)(dst) << 18) | ((u64)(o) << 16) | (u64)(dlen))
#define ADJ_OFF(e) ((u32)((e) >> 49) & 0x7FFFu)
#define ADJ_DST(e) (((e) >> 18) & 0x7FFFFFFFull)
#define ADJ_ORI(e) ((u32)(((e) >> 16) & 3u))
#define ADJ_DLEN(e) ((u32)((e)&0x7FFFu))

/* ---- containment key for atomicMin: smallest super id wins, then smallest j, then prefix record first ---------- */
#define CKEY_MAKE(a, j, suf, rev) (((u64)(a) << 17) | ((u64)(j) << 2) | ((u64)(suf) << 1) | (u64)(rev))
#define CKEY_SUPER(c) ((c) >> 17)
#define CKEY_J(c) ((u32)((c) >> 2) & 0x7FFFu)
#define CKEY_SUFFIX(c) ((u32)(((c) >> 1) & 1u))
#define CKEY_REV(c) ((u32)((c)&1u))

/* hash-hit type of BG/HashTable.cpp:535-566 from (record kind, strand relation) */
__host__ __device__ __forceinline__ u32 disco_hit_type(u32 is_suffix, u32 rev)
{
    /* prefix record: same strand -> 0, opposite -> 3 ; suffix record: same strand -> 1, opposite -> 2 */
    return is_suffix ? (rev ? 2u : 1u) : (rev ? 3u : 0u);
}

/* orientation map of BG/OverlapGraph.cpp:428-434 / :660-666 ; offset = len1 - overlapLen */
__host__ __device__ __forceinline__ void disco_map_type(u32 type, u32 len1, u32 k, u32 j, u32 *orient, u32 *offset)
{
    switch (type) {
    case 0: *orient = 3; *offset = j; break;              /* ovl = len1 - j */
    case 1: *orient = 0; *offset = len1 - (k + j); break; /* ovl = k + j    */
    case 2: *orient = 2; *offset = j; break;
    default: *orient = 1; *offset = len1 - (k + j); break;
    }
}

/* the same map straight from a record's two bits, without a branch (the switch compiles to three nested exec-mask regions per hit: round 6).
 * (suffix, rev) -> type -> orient: (0,0) -> 0 -> 3 ; (1,0) -> 1 -> 0 ; (1,1) -> 2 -> 2 ; (0,1) -> 3 -> 1, i.e. bit 1 = (suffix == rev),
 * bit 0 = !suffix; the offset is j where suffix == rev (types 0, 2: ovl = len1 - j) and len1 - (k + j) otherwise */
__host__ __device__ __forceinline__ void disco_map_hit(u32 is_suffix, u32 rev, u32 len1, u32 k, u32 j, u32 *orient, u32 *offset)
{
    const u32 same = (is_suffix ^ rev ^ 1u) & 1u;
    *orient = (same << 1) | ((is_suffix ^ 1u) & 1u);
    *offset = same ? j : len1 - (k + j);
}

/* twinEdgeOrientation, BG/OverlapGraph.cpp:770-784 */
__host__ __device__ __forceinline__ u32 disco_twin_orient(u32 o) { return o == 0 ? 3u : (o == 3 ? 0u : o); }

__host__ __device__ __forceinline__ u64 disco_hash64(u64 x)
{
    x ^= x >> 33;
    x *= 0xFF51AFD7ED558CCDull;
    x ^= x >> 33;
    x *= 0xC4CEB9FE1A85EC53ull;
    x ^= x >> 33;
    return x;
}

#if defined(__HIPCC__)

__device__ __forceinline__ u64 lane_mask_lt()
{
    u32 l = __lane_id();
    return l ? (~0ull >> (64 - l)) : 0ull;
}
/* how many bits of a wave mask are set below this lane: v_mbcnt_lo + v_mbcnt_hi (the compiler does not find them in
 * __popcll(mask & lane_mask_lt()): it builds the mask with a 64-bit shift and two selects — nine vector instructions) */
__device__ __forceinline__ u32 rank_below(u64 mask) { return __builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u)); }

/* inclusive minimum scans inside the 16-lane rows of a wavefront (DPP row shifts; a lane without a source keeps ~0):
 * row_prefix_min: lane i gets the minimum over its row's lanes <= i; row_suffix_min: over its row's lanes >= i */
__device__ __forceinline__ u32 row_prefix_min(u32 x)
{
    u32 y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x111, 0xF, 0xF, false); /* row_shr:1 */
    x = x < y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x112, 0xF, 0xF, false);
    x = x < y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x114, 0xF, 0xF, false);
    x = x < y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x118, 0xF, 0xF, false);
    return x < y ? x : y;
}
__device__ __forceinline__ u32 row_suffix_min(u32 x)
{
    u32 y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x101, 0xF, 0xF, false); /* row_shl:1 */
    x = x < y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x102, 0xF, 0xF, false);
    x = x < y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x104, 0xF, 0xF, false);
    x = x < y ? x : y;
    y = (u32)__builtin_amdgcn_update_dpp(-1, (int)x, 0x108, 0xF, 0xF, false);
    return x < y ? x : y;
}

/* inclusive prefix sum across the wavefront, all in DPP adds: four row_shr steps inside the 16-lane rows, then row_bcast:15
 * into rows 1 and 3 and row_bcast:31 into rows 2 and 3 (six instructions; the shuffle form is six ds_bpermute round trips in a
 * dependency chain) */
__device__ __forceinline__ u32 wave_inclusive_add(u32 x)
{
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);
    x += (u32)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);
    return x;
}

/* x of lane (lane ^ j), j a compile-time power of two (the sorts below unroll completely): strides 1 and 2 are DPP quad
 * permutes, 8 is a rotation inside the 16-lane row — register-to-register moves with a few cycles of latency where
 * ds_bpermute goes through the LDS pipe; a bitonic network is one long dependency chain, so the latency is what counts.
 * (Strides 4, 16 and 32 can be had without LDS too — two bank-masked row rotations, v_permlane16/32_swap plus a row-masked
 * move; verified on the device, but edge_select_kernel ran 25.8 vs 25.7 ms with them, so they keep the plain shuffle.) */
__device__ __forceinline__ u32 lane_xor32(u32 x, int j)
{
    switch (j) {
    case 1: return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);  /* quad_perm [1,0,3,2] */
    case 2: return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);  /* quad_perm [2,3,0,1] */
    case 8: return (u32)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true); /* row_ror:8 */
    default: return (u32)__shfl_xor((int)x, j);
    }
}
__device__ __forceinline__ u64 lane_xor64(u64 x, int j) { return ((u64)lane_xor32((u32)(x >> 32), j) << 32) | lane_xor32((u32)x, j); }

/* ascending bitonic sort of one u64 per lane across the wavefront (21 compare-exchange steps) */
__device__ __forceinline__ u64 wave_bitonic_sort(u64 x, u32 lane)
{
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            const u64 y = lane_xor64(x, j2);
            const bool up = (lane & k2) == 0;
            const bool lower = (lane & j2) == 0;
            const bool take_min = (up == lower);
            x = ((x < y) == take_min) ? x : y; /* one compare; the lane masks are wave constants (scalar registers) */
        }
    }
    return x;
}

/* ascending bitonic sort of one u32 per lane across the wavefront */
__device__ __forceinline__ u32 wave_bitonic_sort32(u32 x, u32 lane)
{
#pragma unroll
    for (int k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
        for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
            const u32 y = lane_xor32(x, j2);
            const bool up = (lane & k2) == 0;
            const bool lower = (lane & j2) == 0;
            const bool take_min = (up == lower);
            x = ((x < y) == take_min) ? x : y; /* one compare; the lane masks are wave constants (scalar registers) */
        }
    }
    return x;
}

/* reverse the order of the 32 2-bit groups of x */
__device__ __forceinline__ u64 rev2_64(u64 x)
{
    u64 y = __brevll(x);
    return ((y >> 1) & 0x5555555555555555ull) | ((y & 0x5555555555555555ull) << 1);
}

/* 32 bases starting at base position pos (>= 0) of the row p[0..S); positions past the row read as A (0).
 * NB = true: branch-free form for rows staged in LDS with one readable word of padding after the row (callers mask what
 * lies beyond the read); the bounds checks of the generic form cost an exec-mask branch per word. */
template <bool NB = false>
__device__ __forceinline__ u64 extract32(const u64 *p, int S, int pos)
{
    const int w = pos >> 5, sh = (pos & 31) * 2;
    if (NB) {
        const u64 a = p[w], b = p[w + 1];
        return (a << sh) | ((b >> 1) >> (63 - sh)); /* (b >> 1) >> (63 - sh) == b >> (64 - sh), and 0 for sh == 0 */
    }
    u64 a = (w < S) ? p[w] : 0ull;
    if (sh == 0) return a;
    u64 b = (w + 1 < S) ? p[w + 1] : 0ull;
    return (a << sh) | (b >> (64 - sh));
}

/* k-mer (k <= 64) at base j as a right-aligned 2k-bit integer hi:lo */
template <bool NB = false>
__device__ __forceinline__ void kmer_at(const u64 *p, int S, int j, int k, u64 &hi, u64 &lo)
{
    if (k <= 32) {
        hi = 0;
        lo = extract32<NB>(p, S, j) >> (64 - 2 * k);
    } else {
        hi = extract32<NB>(p, S, j) >> (64 - 2 * (k - 32));
        lo = extract32<NB>(p, S, j + k - 32);
    }
}

/* reverse complement of a right-aligned 2k-bit k-mer */
__device__ __forceinline__ void kmer_revcomp(u64 hi, u64 lo, int k, u64 &rhi, u64 &rlo)
{
    u64 t_hi = rev2_64(~lo), t_lo = rev2_64(~hi); /* field now left-aligned in t_hi:t_lo */
    int sh = 128 - 2 * k;
    if (sh >= 64) {
        rhi = 0;
        rlo = (sh == 64) ? t_hi : (t_hi >> (sh - 64));
    } else if (sh == 0) {
        rhi = t_hi;
        rlo = t_lo;
    } else {
        rlo = (t_lo >> sh) | (t_hi << (64 - sh));
        rhi = t_hi >> sh;
    }
}

/* canonical orientation of the k-mer at j (the reference canonicalises through min(hash(fwd), hash(rc)),
 * BG/HashTable.cpp:383-391; any strand-symmetric choice gives the same buckets' contents up to order):
 * 1 when the reverse complement is the smaller 2k-bit integer. A palindrome (k even) has 0. */
/* LONGK (round 4: k up to 94; the kernels are instantiated for it separately — inside probe_kernel the registers of the loop are the
 * ones that spill, so the variants for k <= 64 do not carry it): 32 bases at a time from the left — word i of the reverse complement
 * is the reverse complement of the k-mer's bases [k - 32 i - n, k - 32 i), n = the bases of that word; the first differing word decides
 * (= the comparison of the two 2k-bit integers) */
template <bool NB = false, bool LONGK = false>
__device__ __forceinline__ u32 kmer_is_rev(const u64 *p, int S, int j, int k)
{
    if (!LONGK) {
        u64 hi, lo, rhi, rlo;
        kmer_at<NB>(p, S, j, k, hi, lo);
        kmer_revcomp(hi, lo, k, rhi, rlo);
        return ((rhi < hi) || (rhi == hi && rlo < lo)) ? 1u : 0u;
    }
#pragma clang loop unroll(disable)
    for (int i = 0; 32 * i < k; i++) {
        const int n = k - 32 * i < 32 ? k - 32 * i : 32;
        const u64 mask = n == 32 ? ~0ull : (~0ull << (64 - 2 * n));
        const u64 fw = extract32<NB>(p, S, j + 32 * i) & mask;
        u64 rc = rev2_64(~extract32<NB>(p, S, j + k - 32 * i - n));
        if (n < 32) rc <<= 2 * (32 - n);
        rc &= mask;
        if (rc != fw) return rc < fw ? 1u : 0u;
    }
    return 0u; /* its own reverse complement */
}

/* canonical m-mer (m <= 32) at base pos: value of min(m-mer, reverse complement); strand = 1 when the reverse complement
 * is the smaller one (m is odd, so the two never tie) */
template <bool NB = false>
__device__ __forceinline__ u64 mmer_canonical(const u64 *p, int S, int pos, int m, u32 &strand)
{
    const u64 v = extract32<NB>(p, S, pos) >> (64 - 2 * m);
    const u64 r = rev2_64(~v) >> (64 - 2 * m);
    strand = r < v ? 1u : 0u;
    return strand ? r : v;
}

/* order word of the m-mer at pos: 23-bit order hash of the canonical m-mer in bits 31..9, bits 8..1 zero (room for a
 * position), the m-mer's strand in bit 0. Minimizers are chosen by the order hash. m <= 23, so the canonical m-mer has at
 * most 46 bits: two full-rate 24-bit multiplies fold it into 32 bits and one 32-bit multiply mixes the result (the order
 * only has to be strand symmetric and unrelated to the base composition; bucket keys use the bijective disco_hash64). */
/* lane l's value of x (l wave-uniform) */
__device__ __forceinline__ u64 readlane_u64(u64 x, u32 l)
{
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x, (int)l);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(x >> 32), (int)l);
    return ((u64)hi << 32) | lo;
}

template <bool WIDE = true>
__device__ __forceinline__ u32 order_hash32(u64 c)
{
    /* (canonical m-mers of up to 23 bases have 46 bits: the third product is zero for them — the order of rounds 1-3; m up to 31,
     * which k above 86 needs, brings 62. WIDE = false: the caller knows that m <= 23 and saves the product) */
    u32 h = __umul24((u32)c & 0xFFFFFFu, 0x9E3779u) + __umul24((u32)(c >> 24) & 0xFFFFFFu, 0x85EBCBu) + (WIDE ? __umul24((u32)(c >> 48), 0xC2B2AEu) : 0u) + 0x7F4A7C15u;
#ifdef ORDER_HASH_MUL32 /* rounds 1-5: xor-shift and one 32-bit multiply (quarter rate: four issue slots) */
    h ^= h >> 15;
    return h * 0x2C1B3C6Du;
#else
    /* round 6: the mix as two more full-rate 24-bit multiplies — the low 24 bits and the high 24 bits of the fold, each spread upwards by
     * an odd constant (three issue slots instead of six; the order is internal: any strand-symmetric hash that is unrelated to the base
     * composition gives the same graph, and index, probe and the multi-GPU key pass all take it from here) */
    return __umul24(h & 0xFFFFFFu, 0xC1B3C7u) + __umul24(h >> 8, 0x9E3779u);
#endif
}
template <bool NB = false>
__device__ __forceinline__ u32 mmer_order(const u64 *p, int S, int pos, int m)
{
    u32 strand;
    const u64 c = mmer_canonical<NB>(p, S, pos, m, strand);
    return (order_hash32(c) & ~0x1FFu) | strand;
}

/* 64-bit bucket key of the m-mer at pos (bijective mix of the canonical m-mer: distinct m-mers never share a key) */
template <bool NB = false>
__device__ __forceinline__ u64 mmer_key(const u64 *p, int S, int pos, int m)
{
    u32 strand;
    return disco_hash64(mmer_canonical<NB>(p, S, pos, m, strand));
}

/* minimizer length for a given k: odd (no m-mer is its own reverse complement), at most 23 (a 150-bp read then has
 * 127 m-mer positions under its k-mer windows: two full 64-lane passes) */
__host__ __device__ __forceinline__ int disco_minimizer_len(int k)
{
    int m = k < 23 ? k : 23;
    if ((m & 1) == 0) m -= 1;
    /* a record stores the minimizer's offset inside its k-mer in six bits (PAY_T) and a window's m-mers are one 64-lane pass: k - m <= 63.
     * k above 86 takes longer minimizers for that: m = k - 63, odd, at most 31 (an m-mer is extracted as one 64-bit word): k <= 94 */
    if (k - m > 63) m = (k - 63) | 1;
    return m < 1 ? 1 : m;
}
#define DISCO_MAX_K 94
#define DISCO_SHORT_MAX 256 /* bases a 64-byte row holds (two classes of rows: longer reads are the long class) */

/* THE WINDOW-MINIMIZER RULE ("window_minimizer's rule" elsewhere). For the k-mer window at base j with the order words
 * h(0..nf-1) of its m-mers (forward offsets), the window's CANONICAL ORIENTATION and the chosen occurrence are defined
 * together, strand-symmetrically:
 *   - unique smallest order hash (23 bits): that m-mer is the minimizer; the window is "reversed" (rev = 1) iff the
 *     m-mer sits on its non-canonical strand. (A k-mer and its reverse complement contain the same physical m-mer on opposite strands.)
 *   - several positions tie (the same canonical m-mer twice, or an order-hash collision; rare): rev = 1 iff the reverse
 *     complement of the whole k-mer is the smaller integer (kmer_is_rev; a palindrome has rev = 0, like BG/HashTable.cpp:539-549
 *     tries the forward match first), and the LEFTMOST tied position in the canonical orientation is taken (leftmost forward
 *     offset when rev = 0, rightmost when rev = 1).
 * A record stores the offset t = rev ? nf-1-f : f of the chosen forward offset f inside the canonical orientation, so a k-mer
 * and its reverse complement always agree on minimizer and offset. Two evaluations exist, both through two running minima
 * over keys (hash | leftmost-first) and (hash | rightmost-first), whose agreement means "unique":
 *   index_count_kernel — serial, over the first / last nf m-mers of its rolling pass (the two end k-mers of a read);
 *   probe_kernel       — all windows of a read at once (DPP row scans or range-minimum tables in LDS). */

/* A[a0 .. a0+m) == s2[b0 .. b0+m) where s2 = B (rev = 0) or revcomp(B) (rev = 1); LB = length of B */
template <bool NB = false>
__device__ __forceinline__ bool seg_equal(const u64 *pa, const u64 *pb, int S, int LB, int a0, int b0, int m, u32 rev)
{
    for (int i = 0; i < m; i += 32) {
        int n = m - i;
        if (n > 32) n = 32;
        u64 wa = extract32<NB>(pa, S, a0 + i);
        u64 wb;
        if (!rev) {
            wb = extract32<NB>(pb, S, b0 + i);
        } else {
            /* s2[b0+i+t] = comp(B[LB-1-b0-i-t]) : take B[q .. q+n) and reverse-complement it */
            int q = LB - b0 - i - n;
            u64 x = extract32<NB>(pb, S, q);
            wb = rev2_64(~x);
            if (n < 32) wb <<= 2 * (32 - n);
        }
        u64 mask = (n == 32) ? ~0ull : (~0ull << (64 - 2 * n));
        if ((wa ^ wb) & mask) return false;
    }
    return true;
}

/* seg_equal for rows of different strides (two row classes: a long read against a short one) */
__device__ __forceinline__ bool seg_equal2(const u64 *pa, int SA, const u64 *pb, int SB, int LB, int a0, int b0, int m, u32 rev)
{
    for (int i = 0; i < m; i += 32) {
        int n = m - i;
        if (n > 32) n = 32;
        const u64 wa = extract32<false>(pa, SA, a0 + i);
        u64 wb;
        if (!rev) {
            wb = extract32<false>(pb, SB, b0 + i);
        } else {
            const int q = LB - b0 - i - n; /* s2[b0 + i + t] = comp(B[LB - 1 - b0 - i - t]) */
            wb = rev2_64(~extract32<false>(pb, SB, q));
            if (n < 32) wb <<= 2 * (32 - n);
        }
        const u64 mask = (n == 32) ? ~0ull : (~0ull << (64 - 2 * n));
        if ((wa ^ wb) & mask) return false;
    }
    return true;
}

/* bases that differ between two packed words (2 bits per base) */
__device__ __forceinline__ u32 base_mismatches(u64 x) { return (u32)__popcll((x | (x >> 1)) & 0x5555555555555555ull); }

/* number of positions at which A[a0 .. a0+m) and s2[b0 .. b0+m) differ (seg_equal's geometry) */
template <bool NB = false>
__device__ __forceinline__ u32 seg_mismatches(const u64 *pa, const u64 *pb, int S, int LB, int a0, int b0, int m, u32 rev)
{
    u32 c = 0;
    for (int i = 0; i < m; i += 32) {
        int n = m - i;
        if (n > 32) n = 32;
        u64 wa = extract32<NB>(pa, S, a0 + i);
        u64 wb;
        if (!rev) {
            wb = extract32<NB>(pb, S, b0 + i);
        } else {
            int q = LB - b0 - i - n;
            u64 x = extract32<NB>(pb, S, q);
            wb = rev2_64(~x);
            if (n < 32) wb <<= 2 * (32 - n);
        }
        u64 mask = (n == 32) ? ~0ull : (~0ull << (64 - 2 * n));
        c += base_mismatches((wa ^ wb) & mask);
    }
    return c;
}

#endif /* __HIPCC__ */
#endif /* DISCO_DEVICE_H_ */
