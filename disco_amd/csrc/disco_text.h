/*
 * disco_text.h — the edge lines of the stage's text files formatted on the GPU (saveParGraphToFile, BG/OverlapGraph.cpp:808-867):
 *     src \t dst \t orient,ovl,0,0,len1,start1,len1-1,len2,0,ovl-1,NA,2 \n          src < dst, 1-based file indices
 * 45 M edges are 2.5 GB of decimal numbers: 16 host threads need a second for them, the GPU a few milliseconds — one thread per
 * edge measures its line, a scan per file places it, the same thread writes it. The files come out byte for byte as
 * disco_amd/host/writer.cpp writes them (edges of a file in fetch order, every line with flag 2: the files are cut along connected
 * components), which is what tests/test_host.py compares.
 */
#ifndef DISCO_TEXT_H_
#define DISCO_TEXT_H_

#include "disco_kernels.h"

struct TextView {
    const u64 *src, *ent;
    const u8 *valid;
    const u64 *pos;
    const u16 *len;
    const u64 *file_index; /* [n] or null: id + 1 */
    u64 n_slots;
};

__device__ __forceinline__ u32 tx_digits(u64 v)
{
    u32 d = 1;
    while (v >= 10) {
        v /= 10;
        d++;
    }
    return d;
}
__device__ __forceinline__ char *tx_put(char *p, u64 v)
{
    const u32 d = tx_digits(v);
    for (u32 i = d; i-- > 0;) {
        p[i] = (char)('0' + (u32)(v % 10));
        v /= 10;
    }
    return p + d;
}
struct TextNumbers {
    u64 a, b;
    u32 orient, ovl, len1, off, len2;
};
__device__ __forceinline__ TextNumbers tx_numbers(const TextView &g, u64 s)
{
    TextNumbers t;
    const u64 src = g.src[s], e = g.ent[s], dst = ADJ_DST(e);
    t.a = g.file_index ? g.file_index[src] : src + 1;
    t.b = g.file_index ? g.file_index[dst] : dst + 1;
    t.orient = ADJ_ORI(e);
    t.len1 = g.len[src];
    t.off = ADJ_OFF(e);
    t.ovl = t.len1 - t.off; /* :814 */
    t.len2 = ADJ_DLEN(e);
    return t;
}

/* bytes of every edge's line, by the edge's rank in fetch order */
__global__ void text_measure_kernel(TextView g, u8 *__restrict__ bytes)
{
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s < g.n_slots; s += (u64)gridDim.x * blockDim.x) {
        if (!g.valid[s]) continue;
        const TextNumbers t = tx_numbers(g, s);
        /* a \t b \t o , ovl ,0,0, len1 , off , len1-1 , len2 ,0, ovl-1 ,NA,2 \n  : 20 fixed characters + the orientation digit */
        bytes[g.pos[s]] = (u8)(tx_digits(t.a) + tx_digits(t.b) + 1 + tx_digits(t.ovl) + tx_digits(t.len1) + tx_digits(t.off) + tx_digits(t.len1 - 1) + tx_digits(t.len2) +
                               tx_digits(t.ovl - 1) + 20);
    }
}

/* the lines of one file: their lengths, 0 for the edges of other files (scanned into offsets inside the file) */
__global__ void text_select_kernel(const u8 *__restrict__ bytes, const u16 *__restrict__ efile, u64 ne, u32 file, u8 *__restrict__ out)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < ne; i += (u64)gridDim.x * blockDim.x) out[i] = efile[i] == file ? bytes[i] : (u8)0;
}
__global__ void text_place_kernel(const u64 *__restrict__ within, const u16 *__restrict__ efile, u64 ne, u32 file, u64 base, u64 *__restrict__ place)
{
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < ne; i += (u64)gridDim.x * blockDim.x)
        if (efile[i] == file) place[i] = base + within[i];
}

__global__ void text_write_kernel(TextView g, const u64 *__restrict__ place, char *__restrict__ text)
{
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (; s < g.n_slots; s += (u64)gridDim.x * blockDim.x) {
        if (!g.valid[s]) continue;
        const TextNumbers t = tx_numbers(g, s);
        char *p = text + place[g.pos[s]];
        p = tx_put(p, t.a); *p++ = '\t';
        p = tx_put(p, t.b); *p++ = '\t';
        *p++ = (char)('0' + t.orient); *p++ = ',';
        p = tx_put(p, t.ovl); *p++ = ','; *p++ = '0'; *p++ = ','; *p++ = '0'; *p++ = ','; /* :815-816 substitutions, edits */
        p = tx_put(p, t.len1); *p++ = ',';
        p = tx_put(p, t.off); *p++ = ',';
        p = tx_put(p, t.len1 - 1); *p++ = ',';
        p = tx_put(p, t.len2); *p++ = ','; *p++ = '0'; *p++ = ',';
        p = tx_put(p, t.ovl - 1); *p++ = ','; *p++ = 'N'; *p++ = 'A'; *p++ = ','; *p++ = '2'; *p++ = '\n';
    }
}

#endif /* DISCO_TEXT_H_ */
