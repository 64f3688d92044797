/*
 * disco_ingest.h — the input stage on the GPU (SURVEY.md section 8 a-1 … a-3, f-2): FASTA text in HBM -> record starts -> clean +
 * Dataset::testRead per record -> ids of the good reads in file order -> 2-bit rows of the read table. Replaces, for the files it
 * accepts, the host pass of disco_amd/host/fastx.cpp (which stays for .gz, FASTQ and any FASTA it declines), i.e.
 * Dataset::readDataset / testRead (BG/Dataset.cpp:161-380,403-452) and the packing of HashTable::insertIntoTable
 * (BG/HashTable.cpp:456-477). Host side: disco_hip.hip "input stage on the GPU".
 *
 * Accepted forms (decided on the device, per file; anything else makes the caller fall back to the host stage, which follows the
 * reference's getline calls literally): FASTA — the file starts with '>' and every '>' is the first byte of a line; a record is its
 * header line and everything up to the next '>' with the newlines taken out (BG/Dataset.cpp:270-281), so sequences may be wrapped
 * (round 4). Records wrapped at ONE width (every line as long as the first, the last one at most that: what every FASTA writer
 * produces) are addressed by arithmetic; irregular ones by walking, which is why an irregular record of more than FX_WALK_MAX bases, or any
 * record of 2^21 bytes or more, still sends the file to the host stage. FASTQ — the file starts with '@': records of four lines
 * (found by counting lines, as the reference's four getline calls do). Lower case, N, CR and any other byte are handled as the
 * reference handles them (upper-cased; anything but ACGT rejects the read, BG/Dataset.cpp:411).
 *
 * All kernels are byte / integer work on text that is read once or twice: HBM-bound streaming (8.1 GB of text at 50 M reads).
 */
#ifndef DISCO_INGEST_H_
#define DISCO_INGEST_H_

#include "disco_device.h"

#define FX_TILE 4096 /* bytes of text per block of fx_starts_kernel */
#define FX_MAX_MOTIFS 16
#define FX_MAX_REPEATS 40
#define FX_WALK_MAX 4096u /* bases of an irregularly wrapped record the device stage still addresses by walking its bytes */
#define FX_WRAP_IRREGULAR 0xFFFFFFFFu

/* counters of one ingest (u64 each) */
enum { FX_CTR_BAD_GT = 0, FX_CTR_MULTILINE, FX_CTR_TOO_LONG, FX_CTR_MAX_LEN, FX_CTR_MIN_LEN_INV, FX_CTR_GOOD, FX_CTR_N_LONG, FX_CTR_SHORT_MAX, FX_CTR_COUNT };
/* (N_LONG / SHORT_MAX: good reads of more than DISCO_SHORT_MAX bases, and the longest of the others — the table may get two classes of rows) */

struct FxTables { /* Dataset::testRead's patterns (read_filter_tables.h), prepared by the host */
    u64 rep58[FX_MAX_REPEATS]; /* the 29-mers that may be neither prefix nor suffix of a read, 2 bits per base */
    u32 n_rep;
    u32 n_motif;
    u8 motif[FX_MAX_MOTIFS][8]; /* upper-case characters */
    u8 motif_len[FX_MAX_MOTIFS];
    u8 need[FX_MAX_MOTIFS][4];  /* bases of each kind in the motif */
};

/* byte p of the text through ALIGNED 8-byte loads (the buffer is padded to a multiple of 8 bytes, little endian) */
struct FxBytes {
    const u64 *w;
    u64 cur_idx;
    u64 cur;
    __device__ __forceinline__ explicit FxBytes(const u8 *text) : w((const u64 *)text), cur_idx(~0ull), cur(0) {}
    __device__ __forceinline__ u32 at(u64 p)
    {
        const u64 i = p >> 3;
        if (i != cur_idx) {
            cur = w[i];
            cur_idx = i;
        }
        return (u32)(cur >> (8 * (p & 7))) & 0xFFu;
    }
};

__device__ __forceinline__ u32 fx_upper(u32 c) { return (c >= 'a' && c <= 'z') ? c - 32u : c; }
/* A0 C1 G2 T3 (BG/HashTable.h:16-24), anything else 4 */
__device__ __forceinline__ u32 fx_code(u32 c) { return c == 'A' ? 0u : (c == 'C' ? 1u : (c == 'G' ? 2u : (c == 'T' ? 3u : 4u))); }

/* record starts: '>' at the first byte of a line. count != nullptr: starts per tile; pos != nullptr: their positions at
 * base[tile] + rank inside the tile. A '>' anywhere else raises FX_CTR_BAD_GT (the file is not of the accepted form). */
__global__ void __launch_bounds__(256) fx_starts_kernel(const u8 *__restrict__ text, u64 n, u32 *__restrict__ count, const u64 *__restrict__ base,
                                                        u64 *__restrict__ pos, u64 *__restrict__ ctr)
{
    __shared__ u32 s_w[4];
    const u64 tile = blockIdx.x;
    const u64 p0 = tile * FX_TILE + (u64)threadIdx.x * 16u;
    u32 mask = 0, bad = 0; /* bit i: byte p0 + i starts a record */
    if (p0 < n) {
        const uint4 q = *(const uint4 *)(text + p0); /* (padded buffer) */
        const u32 wds[4] = {q.x, q.y, q.z, q.w};
        u32 prev = p0 ? text[p0 - 1] : (u32)'\n';
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 c = (wds[i >> 2] >> (8 * (i & 3))) & 0xFFu;
            if (p0 + i < n && c == '>') {
                if (prev == '\n') mask |= 1u << i;
                else bad = 1;
            }
            prev = c;
        }
    }
    if (bad) atomicAdd(&ctr[FX_CTR_BAD_GT], 1ull);
    const u32 mine = (u32)__popc(mask);
    /* exclusive prefix of `mine` over the block */
    u32 incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 y = (u32)__shfl_up((int)incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += y;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 off = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += s_w[w];
    if (count) {
        if (threadIdx.x == 255) count[tile] = off + mine;
    } else {
        u64 at = base[tile] + off;
        u32 m = mask;
        while (m) {
            const int i = __ffs((int)m) - 1;
            m &= m - 1;
            pos[at++] = p0 + (u64)i;
        }
    }
}

/* FASTQ: a record is four lines (header, sequence, '+', qualities: the reference reads them with four getline calls,
 * BG/Dataset.cpp:255-293 — a quality line may well begin with '@', so only the line COUNT says where a record starts). Pass 1
 * (count != nullptr): newlines per tile. Pass 2: base[tile] = newlines before the tile; every byte that begins a line (byte 0, or the
 * byte behind a '\n') whose line index is a multiple of 4 is a record start: its position goes to pos[line / 4]. */
__global__ void __launch_bounds__(256) fx_lines_kernel(const u8 *__restrict__ text, u64 n, u32 *__restrict__ count, const u64 *__restrict__ base,
                                                       u64 *__restrict__ pos)
{
    __shared__ u32 s_w[4];
    const u64 tile = blockIdx.x;
    const u64 p0 = tile * FX_TILE + (u64)threadIdx.x * 16u;
    u32 nlmask = 0; /* bit i: byte p0 + i is a newline */
    if (p0 < n) {
        const uint4 q = *(const uint4 *)(text + p0);
        const u32 wds[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 16; i++)
            if (p0 + i < n && ((wds[i >> 2] >> (8 * (i & 3))) & 0xFFu) == '\n') nlmask |= 1u << i;
    }
    const u32 mine = (u32)__popc(nlmask);
    u32 incl = mine;
    for (int o = 1; o < 64; o <<= 1) {
        const u32 y = (u32)__shfl_up((int)incl, o);
        if ((int)(threadIdx.x & 63) >= o) incl += y;
    }
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    u32 off = incl - mine;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) off += s_w[w];
    if (count) {
        if (threadIdx.x == 255) count[tile] = off + mine;
        return;
    }
    if (p0 >= n) return;
    /* line index of the line that BEGINS at byte p0 + i = newlines before that byte */
    u64 before = base[tile] + off;
    const bool starts_line = p0 == 0 || text[p0 - 1] == '\n';
    if (starts_line && (before & 3ull) == 0) pos[before >> 2] = p0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (nlmask & (1u << i)) {
            before++;
            if (p0 + i + 1 < n && (before & 3ull) == 0) pos[before >> 2] = p0 + (u64)i + 1; /* the byte behind this newline begins a line */
        }
    }
}

struct FxFilterArgs {
    const u8 *text;
    u64 n;           /* bytes of the file */
    const u64 *start; /* [n_start] record starts */
    u64 n_start;
    u64 n_rec;       /* records (n_start, or one less when the last '>' is the last byte of the file) */
    u32 min_overlap;
    u32 fastq;       /* the record's sequence is its SECOND line only (four-line records) */
    u16 *glen;       /* out [n_rec]: 0 = rejected, else the read length */
    u64 *seq_begin;  /* out [n_rec]: first byte of the sequence */
    u32 *wrap;       /* out [n_rec]: bases per line of the (wrapped) sequence — base b is byte seq_begin + b + b / wrap; FX_WRAP_IRREGULAR: walk */
    u64 *ctr;
};

/* raw byte of base b of a sequence that begins at byte sb: by arithmetic for a sequence wrapped at `wrap` bases per line (a sequence on
 * one line: wrap = its length), by walking over the newlines otherwise */
__device__ __forceinline__ u64 fx_raw_of(FxBytes &tx, u64 sb, u32 wrap, u32 b)
{
    if (wrap != FX_WRAP_IRREGULAR) return sb + b + (wrap ? b / wrap : 0u);
    u64 p = sb;
    for (u32 seen = 0;; p++) {
        if (tx.at(p) == '\n') continue;
        if (seen == b) return p;
        seen++;
    }
}

/* one thread per record: clean (upper case) + count in one pass over the sequence line, then Dataset::testRead
 * (BG/Dataset.cpp:403-452) exactly as disco_amd/host/fastx.cpp:test_read_counted evaluates it */
__global__ void __launch_bounds__(256) fx_filter_kernel(FxFilterArgs a, FxTables tb)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    u32 my_max = 0, my_min = 0xFFFFu, my_good = 0, my_long = 0, my_smax = 0;
    for (; i < a.n_rec; i += (u64)gridDim.x * blockDim.x) {
        const u64 s = a.start[i], e = (i + 1 < a.n_start) ? a.start[i + 1] : a.n;
        FxBytes tx(a.text);
        u64 p = s;
        while (p < e && tx.at(p) != '\n') p++; /* header line */
        const u64 sb = p < e ? p + 1 : e;
        u64 se = e;
        if (a.fastq) { /* the sequence line ends at its newline (or with the file) */
            se = sb;
            while (se < e && tx.at(se) != '\n') se++;
        }
        a.seq_begin[i] = sb;
        /* counters packed into words (a dynamically indexed local array would live in scratch memory): A | C << 32, G | T << 32, the six
         * dimers of two different letters — their occurrences cannot overlap, so plain counts — AC | AG << 21 | AT << 42, CG | CT << 21 | GT << 42 */
        u64 acgt01 = 0, acgt23 = 0, dimA = 0, dimB = 0, other = 0;
        u32 aa = 0; /* "AA" adjacencies (round 6: the bound that keeps TAA / CAA / GAA off the greedy scan, below) */
        u64 head = 0, tail = 0;
        u32 prev = 4;
        /* FASTA: the newlines inside [sb, se) are not part of the sequence (BG/Dataset.cpp:270-281). W = bases on the first line; the
         * record is regular while every later line has W bases too, except that the lines may end with one shorter one */
        u64 L = 0;
        u32 W = 0, run = 0, n_nl = 0;
        bool short_seen = false, irregular = false;
        if (se - sb >= (1ull << 21)) { /* the 21-bit dimer counters; such a record is no read anyway: the host stage says so in its words */
            atomicAdd(&a.ctr[FX_CTR_MULTILINE], 1ull);
            a.glen[i] = 0;
            a.wrap[i] = 0;
            continue;
        }
        for (u64 q = sb; q < se; q++) {
            const u32 raw = tx.at(q);
            if (raw == '\n') { /* (FASTQ: se stops in front of the line's newline) */
                if (n_nl == 0) W = run;
                else {
                    if (run > W) irregular = true;
                    if (run < W) short_seen = true;
                }
                n_nl++;
                run = 0;
                continue;
            }
            if (short_seen) irregular = true; /* bases behind a line shorter than the first */
            run++;
            const u32 c = fx_code(fx_upper(raw));
            if (c < 2) acgt01 += 1ull << (32 * c);
            else if (c < 4) acgt23 += 1ull << (32 * (c - 2));
            else other++;
            if (c < 4 && c > prev) { /* (prev, c) in (0,1)(0,2)(0,3) | (1,2)(1,3)(2,3) */
                if (prev == 0) dimA += 1ull << (21 * (c - 1));
                else dimB += 1ull << (21 * (prev + c - 3));
            }
            aa += (c == 0 && prev == 0) ? 1u : 0u;
            tail = ((tail << 2) | (c & 3u)) & ((1ull << 58) - 1ull);
            if (L == 28) head = tail;
            prev = c;
            L++;
        }
        if (n_nl == 0) W = (u32)L;           /* one line without a newline behind it (the end of the file) */
        else if (run > W) irregular = true;  /* ... a last line without a newline, longer than the first */
        if (W == 0 && L != 0) irregular = true; /* an empty first line */
        const u32 wrap = irregular ? FX_WRAP_IRREGULAR : W;
        a.wrap[i] = wrap;
        const u32 cnt[5] = {(u32)acgt01, (u32)(acgt01 >> 32), (u32)acgt23, (u32)(acgt23 >> 32), (u32)other};
        const u32 dim[6] = {(u32)(dimA & 0x1FFFFFu), (u32)((dimA >> 21) & 0x1FFFFFu), (u32)((dimA >> 42) & 0x1FFFFFu),
                            (u32)(dimB & 0x1FFFFFu), (u32)((dimB >> 21) & 0x1FFFFFu), (u32)((dimB >> 42) & 0x1FFFFFu)};
        if (irregular && L > FX_WALK_MAX) { /* not a form this stage addresses: the whole file goes to the host stage */
            atomicAdd(&a.ctr[FX_CTR_MULTILINE], 1ull);
            a.glen[i] = 0;
            continue;
        }
        bool good = L > (u64)a.min_overlap && L >= 30 && cnt[4] == 0; /* BG/Dataset.cpp:305, MIN_READ_SIZE, :411 */
        if (good) {
            const u64 thr = (u64)((double)L * .7);
            for (int b = 0; b < 4; b++)
                if ((u64)cnt[b] >= thr) good = false;
        }
        if (good) {
            for (u32 r = 0; r < tb.n_rep; r++)
                if (head == tb.rep58[r] || tail == tb.rep58[r]) good = false;
        }
        if (good) {
            const u64 thr = (u64)((double)L * .5);
            for (u32 mi = 0; mi < tb.n_motif && good; mi++) {
                const u32 ml = tb.motif_len[mi];
                const u64 qq = (thr + ml - 1) / ml;
                bool skip = false; /* the motif cannot occur often enough for the base counts (exact upper bound: fastx.cpp) */
                for (int b = 0; b < 4; b++)
                    if (tb.need[mi][b] && (u64)cnt[b] < (u64)tb.need[mi][b] * qq) skip = true;
                if (skip) continue;
                u64 covered;
                const u32 x = fx_code(tb.motif[mi][0]), y = fx_code(tb.motif[mi][1]);
                /* round 6: a second exact bound before the greedy scan — every occurrence of the motif contains each of its adjacent letter
                 * pairs, and occurrences that do not overlap contain DIFFERENT adjacencies of the read: hits <= the read's count of any pair
                 * inside the motif. The pass above counts the six pairs of two different ascending letters and "AA": one of them lies inside
                 * every trimer of the table (AAT ATA AAC ACA AAG AGA: AT / AC / AG; TAA CAA GAA: AA). The base-count bound alone let one read
                 * in a hundred through per motif — a 0.9 % tail of the binomial — i.e. nearly every WAVEFRONT took the scan for some lane:
                 * 87 ms of the stage at 50 M reads, thirty times the pass that counts. */
                if (ml >= 3) {
                    u64 pair_bound = ~0ull;
                    for (u32 t = 0; t + 1 < ml; t++) {
                        const u32 p0 = fx_code(tb.motif[mi][t]), p1 = fx_code(tb.motif[mi][t + 1]);
                        if (p0 < p1 && p1 < 4) pair_bound = min(pair_bound, (u64)dim[p0 == 0 ? p1 - 1 : (p0 == 1 ? p1 + 1 : 5)]);
                        else if (p0 == 0 && p1 == 0) pair_bound = min(pair_bound, (u64)aa);
                    }
                    if (pair_bound != ~0ull && pair_bound * ml < thr) continue;
                }
                if (ml == 2 && x < y && y < 4) {
                    covered = 2ull * dim[x == 0 ? y - 1 : (x == 1 ? y + 1 : 5)];
                } else { /* left to right, non-overlapping: BG/Common.h:173-183 */
                    u64 hits = 0;
                    const bool one_line = wrap == (u32)L; /* (no newline inside the sequence: byte = sb + base, no division per byte) */
                    for (u64 q = 0; q + ml <= L;) {
                        bool eq = true;
                        for (u32 t = 0; t < ml && eq; t++) eq = fx_upper(tx.at(one_line ? sb + q + t : fx_raw_of(tx, sb, wrap, (u32)(q + t)))) == tb.motif[mi][t];
                        if (eq) {
                            hits++;
                            q += ml;
                        } else
                            q++;
                    }
                    covered = hits * ml;
                }
                if (covered >= thr) good = false;
            }
        }
        if (good && L > 32767) { /* the packed layout has the reference's 15-bit length field (BG/HashTable.cpp:531) */
            atomicAdd(&a.ctr[FX_CTR_TOO_LONG], 1ull);
            good = false;
        }
        a.glen[i] = good ? (u16)L : (u16)0;
        if (good) {
            my_good++;
            my_max = max(my_max, (u32)L);
            my_min = min(my_min, (u32)L);
            if (L > (u64)DISCO_SHORT_MAX) my_long++;
            else my_smax = max(my_smax, (u32)L);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        my_max = max(my_max, (u32)__shfl_down((int)my_max, o));
        my_min = min(my_min, (u32)__shfl_down((int)my_min, o));
        my_good += (u32)__shfl_down((int)my_good, o);
        my_long += (u32)__shfl_down((int)my_long, o);
        my_smax = max(my_smax, (u32)__shfl_down((int)my_smax, o));
    }
    if ((threadIdx.x & 63) == 0 && my_good) {
        atomicMax(&a.ctr[FX_CTR_MAX_LEN], (u64)my_max);
        atomicMax(&a.ctr[FX_CTR_MIN_LEN_INV], (u64)(0xFFFFu - my_min));
        atomicAdd(&a.ctr[FX_CTR_GOOD], (u64)my_good);
        if (my_long) atomicAdd(&a.ctr[FX_CTR_N_LONG], (u64)my_long);
        atomicMax(&a.ctr[FX_CTR_SHORT_MAX], (u64)my_smax);
    }
}

__global__ void fx_flags_kernel(const u16 *__restrict__ glen, u64 n, u8 *__restrict__ flag)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (u64)gridDim.x * blockDim.x) flag[i] = glen[i] != 0;
}

/* good record i -> read id_base + pos[i]: its record number (for the file index), its length */
__global__ void fx_ids_kernel(const u16 *__restrict__ glen, const u64 *__restrict__ pos, u64 n_rec, u64 id_base, u32 *__restrict__ rec_of_read, u16 *__restrict__ len)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_rec; i += (u64)gridDim.x * blockDim.x)
        if (glen[i]) {
            rec_of_read[id_base + pos[i]] = (u32)i;
            len[id_base + pos[i]] = glen[i];
        }
}

/* 32 bases (fewer at the read's end: zero behind) from base b0 of a sequence, 2 bits per base, MSB first (BG/HashTable.cpp:456-477) */
__device__ __forceinline__ u64 fx_pack_word(FxBytes &tx, u64 seq_begin, u32 wrap, u32 L, u32 b0)
{
    if (b0 >= L) return 0ull;
    const u32 nb = min(32u, L - b0);
    u64 acc = 0;
    u64 p = fx_raw_of(tx, seq_begin, wrap, b0); /* (the bytes from there on: newlines skipped as they come) */
    for (u32 x = 0; x < nb; x++, p++) {
        u32 ch = tx.at(p);
        while (ch == '\n') ch = tx.at(++p);
        acc = (acc << 2) | (fx_code(fx_upper(ch)) & 3u);
    }
    return acc << (2 * (32 - nb));
}

/* word w of the row of read id: bases [32 w, 32 w + 32) of its sequence line; words behind the read are zero. One thread per word of
 * the table rows [id_base, id_base + n_good). cap: bases of a read that a row takes (two classes of rows: the first 256 of a long read) */
__global__ void __launch_bounds__(256) fx_pack_kernel(const u8 *__restrict__ text, const u64 *__restrict__ seq_begin, const u32 *__restrict__ wrap,
                                                      const u32 *__restrict__ rec_of_read, const u16 *__restrict__ len, u64 id_base, u64 n_good, int S, u32 cap,
                                                      u64 *__restrict__ reads)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 total = n_good * (u64)S;
    for (; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 id = id_base + t / (u64)S;
        const u32 w = (u32)(t % (u64)S);
        const u32 L = min((u32)len[id], cap);
        u64 acc = 0;
        if (32u * w < L) {
            const u32 rec = rec_of_read[id];
            FxBytes tx(text);
            acc = fx_pack_word(tx, seq_begin[rec], wrap[rec], L, 32u * w);
        }
        reads[id * (u64)S + w] = acc;
    }
}

/* two classes of rows (disco_kernels.h): the long reads of the file whose reads are [id_base, id_base + n_good) — their full rows
 * full[j][SL] and their tail rows rows8[n + j] (the last tailb bases). One thread per word; long read j is long_ids[j]. */
__global__ void __launch_bounds__(256) fx_pack_long_kernel(const u8 *__restrict__ text, const u64 *__restrict__ seq_begin, const u32 *__restrict__ wrap,
                                                           const u32 *__restrict__ rec_of_read, const u16 *__restrict__ len, u64 id_base, u64 n_good,
                                                           const u32 *__restrict__ long_ids, u64 n_long, u64 n, int SL, int tailb, u64 *__restrict__ full,
                                                           u64 *__restrict__ rows8)
{
    u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const u64 per = (u64)SL + 8u, total = n_long * per;
    for (; t < total; t += (u64)gridDim.x * blockDim.x) {
        const u64 j = t / per;
        const u32 w = (u32)(t % per);
        const u64 id = long_ids[j];
        if (id < id_base || id >= id_base + n_good) continue;
        const u32 rec = rec_of_read[id], L = len[id];
        FxBytes tx(text);
        if (w < (u32)SL) full[j * (u64)SL + w] = fx_pack_word(tx, seq_begin[rec], wrap[rec], L, 32u * w);
        else {
            const u32 tw = w - (u32)SL;
            rows8[(n + j) * 8 + tw] = 32u * tw < (u32)tailb ? fx_pack_word(tx, seq_begin[rec], wrap[rec], L, L - (u32)tailb + 32u * tw) : 0ull;
        }
    }
}

#endif
