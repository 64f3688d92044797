/*
 * disco_hip.hip — C-ABI implementation (include/disco_hip.h) over the kernels in disco_kernels.h.
 * Built for gfx950 only:  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -o libdisco_hip.so disco_hip.hip
 *
 * Data layout in HBM (n reads, stride S words, T = bucket table size, Eraw = verified overlap hits, E = directed edges):
 *   reads      u64[n][S]      2-bit packed, fixed stride (coalesced row fetches, no offset indirection)
 *   len        u16[n]
 *   bkt        u32[T+1]       CSR bucket table of the end-k-mer index, T = pow2 >= 4n
 *   ent        u64[2n]        8-byte records: key fingerprint | id | minimizer offset | record strand | isSuffix | len
 *   best       u64[n]         containment keys (atomicMin), all-reduced(MIN) across ranks
 *   contained  u8[n]
 *   hits       u64[cap]       raw verified overlap hits, wave-private chunks, rows addressed by row_start/row_cnt
 *   adj_start  u64[n+1], adj u64[E]   adjacency CSR in node order (offset | dst | orient per entry)
 *   flag       u8[E]          bit0 = transitive from this end, bit1 = survives (emitted from this end)
 */
#include <hip/hip_runtime.h>
#include <rocprofiler-sdk-roctx/roctx.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <memory>
#include <thread>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/disco_hip.h"
#include "../../include/disco_hip_test.h"
#include "disco_kernels.h"
#include "disco_dist.h"

#define DROP_LIST_CAP (1u << 20) /* dropped hits edge selection writes down for the twin search (16 MB); more than that: the bitmap search */
#include "disco_chains.h"
#include "disco_text.h"
#include "disco_ingest.h"
#include "read_filter_tables.h"
#include "disco_comm.h"

static_assert(sizeof(disco_genspec) == sizeof(disco_genspec_abi), "genspec ABI mismatch");

static thread_local std::string g_create_error;

using HClock = std::chrono::steady_clock;
static float ms_since(HClock::time_point t0) { return std::chrono::duration<float, std::milli>(HClock::now() - t0).count(); }

/* phase timers: the public phases (disco_phase_ms) plus one internal slot — the index phase's second bracket (ph_collect adds it to the first) */
#define DISCO_PH_INDEX2 DISCO_PH_COUNT
#define DISCO_PH_SLOTS (DISCO_PH_COUNT + 1)

/* tracing hooks (the reference brackets its functions with CLOCKSTART / CLOCKSTOP, BG/Common.h:71-95): every
 * phase of the path is a named roctx range, so `rocprofv3 --marker-trace --kernel-trace` shows the kernels under the C-ABI call
 * and the phase that launched them; without a tool attached a range costs two calls into an empty library */
struct RoctxRange {
    explicit RoctxRange(const char *name) { (void)roctxRangePushA(name); }
    ~RoctxRange() { (void)roctxRangePop(); }
    RoctxRange(const RoctxRange &) = delete;
    RoctxRange &operator=(const RoctxRange &) = delete;
};
#define DISCO_TRACE(name) RoctxRange roctx_range_(name)

/* host-side loops over tens of millions of results (struct conversion, random reads of the length table) */
template <typename F>
static void parallel_for(u64 n, F f)
{
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("DISCO_HOST_THREADS")) nt = (unsigned)atoi(e);
    nt = std::max(1u, std::min(nt, 16u));
    if (n < (1u << 16) || nt == 1) {
        f((u64)0, n);
        return;
    }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; t++) th.emplace_back([=]() { f(n * t / nt, n * (t + 1) / nt); });
    for (auto &x : th) x.join();
}


/* What a pass did besides launching kernels, per host thread (= per rank: one thread drives one rank, in-process or not):
 * device allocations and frees that reached the HIP runtime, blocking waits on the device, operations on a communicator. A
 * multi-GPU pass reports them (disco_dist_info): allocations between two collectives of a pass are a hazard when several ranks
 * share a process, and every blocking wait / collective launch is host time on the critical path of an 8-GPU pass. */
struct PassCounters {
    u32 dev_allocs = 0, dev_frees = 0, host_syncs = 0, comm_ops = 0;
};
static thread_local PassCounters tl_pass;
static inline hipError_t counted_stream_sync(hipStream_t s)
{
    tl_pass.host_syncs++;
    return hipStreamSynchronize(s);
}
static inline hipError_t counted_event_sync(hipEvent_t e)
{
    tl_pass.host_syncs++;
    return hipEventSynchronize(e);
}
#define hipStreamSynchronize(s) counted_stream_sync(s)
#define hipEventSynchronize(e) counted_event_sync(e)

/* Arena of a multi-GPU context: ONE device allocation made before the first collective of the first pass; every buffer of the pass
 * is carved out of it (first fit, neighbours coalesced on free: a few hundred calls per pass), so that no hipMalloc / hipFree runs
 * between the collectives of a pass — with one host thread per rank in one process (buildG --gpus N) a device allocation while
 * another rank's RCCL kernels are in flight can serialise against them or deadlock. A request the arena cannot serve falls back to
 * the runtime and is counted (disco_dist_info.device_allocs). */
struct DevArena {
    char *base = nullptr;
    size_t size = 0, used = 0, peak = 0;
    std::map<size_t, size_t> free_at; /* offset -> bytes of every free block */
    bool owns(const void *p) const { return base && (const char *)p >= base && (const char *)p < base + size; }
    void *alloc(size_t bytes)
    {
        bytes = (bytes + 255) & ~(size_t)255;
        for (auto it = free_at.begin(); it != free_at.end(); ++it)
            if (it->second >= bytes) {
                const size_t off = it->first, len = it->second;
                free_at.erase(it);
                if (len > bytes) free_at[off + bytes] = len - bytes;
                live[off] = bytes;
                used += bytes;
                peak = std::max(peak, used);
                return base + off;
            }
        return nullptr;
    }
    void release(void *p)
    {
        const size_t off = (size_t)((char *)p - base);
        auto lv = live.find(off);
        if (lv == live.end()) return;
        size_t start = off, end = off + lv->second;
        used -= lv->second;
        live.erase(lv);
        auto nx = free_at.lower_bound(off); /* the first free block behind the freed one */
        if (nx != free_at.end() && nx->first == end) {
            end += nx->second;
            nx = free_at.erase(nx);
        }
        if (nx != free_at.begin()) {
            auto pv = std::prev(nx);
            if (pv->first + pv->second == start) {
                start = pv->first;
                free_at.erase(pv);
            }
        }
        free_at[start] = end - start;
    }
    std::map<size_t, size_t> live; /* offset -> bytes of every block handed out */
};

struct disco_ctx {
    int device = 0;
    disco_params prm{};
    DevArena arena;
    int k = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    int n_cu = 256;
    std::string err;
    size_t hbm_bytes = 0, hbm_peak = 0;

    /* reads */
    u64 n = 0;
    int S = 0;
    u64 *d_reads = nullptr;
    u16 *d_len = nullptr;
    bool reads_owned = false;
    std::vector<uint16_t> h_len; /* lazily mirrored for result decoding */
    bool h_len_ok = false;       /* h_len holds the lengths of the current reads */
    std::mutex h_len_mu;         /* disco_ingest_fetch may run on a host thread of its own next to a pass (the one call allowed to):
                                    it and ensure_host_len are the two writers that can meet */
    u64 q_lo = 0, q_hi = 0;
    /* two classes of rows (disco_kernels.h, "two classes of rows"): made by two_class_convert at the first index build over a table whose
     * stride a few long reads forced; c->S is 8 from then on and S_ext what the caller gave (disco_stride_words, disco_download_reads) */
    bool two_class = false;
    bool dist_two_class = false; /* set by a multi-GPU pass around its two_class_convert (round 6) */
    u64 job_n_long = 0;          /* ... reads of more than 256 bases in the whole job, and the longest of the others (dist_validate) */
    u32 job_short_max = 0;
    int S_ext = 0;
    u64 n_long = 0;
    u64 reads_rows = 0; /* rows d_reads was allocated with (two classes: n + n_long) */
    u64 *d_full = nullptr;
    u32 *d_ovf = nullptr, *d_long_ids = nullptr;
    u32 max_len_all = 0; /* longest read of the set (max_len: of the short class) */
    int tailb = 0;
    u32 *d_lpos = nullptr, *d_n_list = nullptr; /* the long reads' candidate rows of the pass (class_take_rows_kernel) */
    uint2 *d_linfo = nullptr;
    ulonglong2 *d_lmeta = nullptr;

    /* index */
    u64 T = 0;
    int bshift = 0;
    u32 *d_bkt = nullptr;
    u64 *d_ent = nullptr;
    ulonglong2 *d_rec = nullptr; /* {key, record}[2n] scratch of the index build */
    u64 rec_cap = 0;

    /* scan temporaries */
    u64 *d_tile = nullptr;
    size_t tile_cap = 0;
    u64 *d_total = nullptr;

    /* counters */
    u64 *d_ctr = nullptr;
    u64 *d_wq = nullptr; /* work-queue counter shared by the wave-per-item kernels (one launch at a time) */
    u64 h_ctr[CTR_COUNT] = {0};

    /* probe */
    u64 *d_best = nullptr;
    u64 *d_hits = nullptr;
    u64 hits_cap = 0;
    u64 hits_used = 0; /* high-water mark of the hit buffer after the probe: what lies behind it is free */
    u64 *d_bump = nullptr;
    u64 *d_row_start = nullptr;
    u32 *d_row_cnt = nullptr;
    u64 *d_big_list = nullptr;
    u32 *d_big_cnt = nullptr;
    u32 *d_n_big = nullptr;
    /* minimizer runs of the reads [runs_lo, runs_lo + runs_n) (index_runs_kernel -> probe_runs_kernel): runs_lpr u32 words per read
     * (16: up to 32 runs, reads of up to 128 windows; 32: up to 64 runs, 256 windows), 0 = the index pass left none */
    u32 *d_runs = nullptr;
    u64 runs_cap = 0, runs_lo = 0, runs_n = 0;
    int runs_lpr = 0;
    u64 *d_slow_list = nullptr; /* reads whose run list is unusable (ties, too many runs): probe_kernel<2> */
    u32 *d_n_slow = nullptr;
    u32 slow_cap = 0;
    u32 big_cap = 0;
    u64 big_rows = 0, slow_rows = 0;
    const u64 *d_order = nullptr; /* a caller's processing order of the query range (plain read ids), or null */
    const u64 *d_order_used = nullptr; /* what the last probe walked: packed entries (ORDER_MAKE) in d_order_own, or null = file order */
    bool order_external = false;
    u32 *d_ocnt = nullptr, *d_okey = nullptr, *d_oslot = nullptr;
    u64 *d_order_own = nullptr;
    u64 okey_cap = 0, oslot_cap = 0, order_cap = 0, ocnt_cap = 0;
    /* the grouping's counting pass ran inside the index pass, for the reads [lo, hi) with 2^bits buckets (d_ocnt holds the counts,
     * d_oslot the slots): the next disco_probe over exactly that range skips its own */
    bool order_counted = false;
    /* the big-item lists are sized from counts the producing kernels leave behind (verify: rows beyond ES_CAP; edge selection: nodes
     * beyond TR_CAP) — valid until something else changes the rows: then the counting pass (ensure_big_cap) runs as before */
    bool es_big_counted = false, tr_big_counted = false;
    bool contained_count_pending = false; /* disco_run_graph: n_contained is read with the next counters */
    bool order_ready = false; /* ... and disco_build_index went on to the order itself (d_order_own, for [order_counted_lo, order_counted_hi)): the fill walks it */
    /* contained rows on their way to the host while the pass goes on (disco_mark_contained -> disco_fetch_contained) */
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_crows = nullptr;
    u32 *d_cpos = nullptr, *d_crow_id = nullptr;
    u64 *d_crow_key = nullptr;
    u64 cpos_cap = 0, crow_cap = 0;
    u64 *d_tile2 = nullptr, *d_total2 = nullptr; /* scan temporaries of the side stream */
    size_t tile2_cap = 0;
    void *h_crows = nullptr; /* pinned, per order (by id, then grouped): keys u64[crows_hcap], ids u32[crows_hcap], lengths u32[crows_hcap] */
    u32 *d_crow_len = nullptr; /* [crow_cap] len2 | len1 << 16 of the rows about to leave */
    u64 crows_hcap = 0, crows_n = 0;
    bool crows_pending = false; /* the rows of the CURRENT flags are on their way / in h_crows */
    bool crows_in_ring = false; /* h_crows is the input stage's pinned ring (h_ring), not an allocation of its own */
    u32 *d_cgrp_cur = nullptr, *d_cgrp_id = nullptr; /* the same rows grouped by containing read (disco_fetch_contained_grouped) */
    u64 *d_cgrp_key = nullptr, *d_cgrp_big = nullptr;
    u64 cgrp_cur_cap = 0, cgrp_cap = 0;
    u64 h_cgrp_big = 0;
    bool cgrp_pending = false;
    /* disco_fetch_edges: the compacted edges before they travel (kept across passes) */
    u32 *d_fetch_src = nullptr;
    u64 *d_fetch_ent = nullptr;
    u64 fetch_cap = 0;
    /* input stage on the GPU (disco_ingest_fasta): what disco_ingest_fetch hands to the host afterwards */
    u32 *d_rec_of_read = nullptr;
    u64 rec_of_read_cap = 0;
    std::vector<u64> ingest_id_base, ingest_rec_base; /* per file: first read id, records before the file */
    u64 ingest_n = 0;
    void *h_ring = nullptr; /* pinned: two halves of the text staging ring */
    std::vector<u64> text_off; /* disco_format_edges: byte range of every file inside d_text */
    u8 *d_ingest = nullptr; /* the input stage's own arena (text, record arrays) when the hit buffer is allocated NEXT to it ... */
    u64 ingest_cap = 0;
    std::thread hits_prealloc; /* ... by this thread, while the files travel and the filter runs (settle_hits_prealloc) */
    u64 *prealloc_ptr = nullptr;
    u64 prealloc_cap = 0;
    size_t ring_half = 0;
    hipEvent_t ev_ring[2] = {nullptr, nullptr};
    hipStream_t copy_stream = nullptr; /* disco_upload_reads: the chunks of the host buffer travel here */
    hipEvent_t ev_copied[3] = {nullptr, nullptr, nullptr}, ev_unpacked[3] = {nullptr, nullptr, nullptr};
    bool index_counted = false; /* disco_upload_reads ran the index's count pass behind its copies: disco_build_index starts at the scan */
    u64 order_counted_lo = 0, order_counted_hi = 0;
    int order_counted_bits = 0;
    u32 *d_asked = nullptr; /* multi-GPU flow: one bit per node — its row has been asked for in this pass (tr_request_*_kernel) */
    u64 asked_cap = 0;
    u64 order_q_lo = 0, order_q_hi = 0;
    ulonglong2 *d_meta_ord = nullptr; /* per-read headers by position in the processing order (probe -> verify) */
    u64 meta_cap = 0;
    u64 dropped_local = 0;
    ProbeRare h_probe_rare;
    ProbeRare *d_probe_rare = nullptr;
    u32 max_len = 0, min_len = 0; /* longest / shortest read (validate_reads) */
    bool two_pass_last = false;    /* the last probe verified in two passes */
    bool contained_done = false;   /* multi-GPU pass: the containment exchange already ran between the two verify passes */

    /* containment */
    u8 *d_contained = nullptr;
    u64 *d_cbits = nullptr; /* one bit per read */
    u64 *d_dropbits = nullptr; /* one bit per read: its edge selection dropped a verified hit; valid for reads [drop_lo, drop_hi) */
    u64 *d_drop_node = nullptr, *d_drop_key = nullptr; /* exact overlaps: the dropped hits themselves (EdgeSelArgs.drop_node), DROP_LIST_CAP items */
    u64 n_drop_items = 0;                              /* how many the last selection recorded (or would have: > DROP_LIST_CAP = list useless) */
    u64 drop_lo = 0, drop_hi = 0;
    u64 n_contained = 0;

    /* edges */
    u64 *d_adj_ref = nullptr; /* [n] position | degree << 40 */
    u64 *d_adj = nullptr;     /* entries: the hit buffer itself (single GPU) or d_adj_own (imported / merged, node-ordered) */
    u64 *d_adj_own = nullptr; /* kept across passes: re-allocating GBs every pass costs more than the kernels */
    u64 *d_adj_spare = nullptr; /* the rebuilding merge writes here and swaps: its 9 GB at 50 M reads are not allocated and freed per pass */
    u64 adj_spare_cap = 0;
    u64 *d_start_tmp = nullptr; /* [n+1] scan scratch of export / import */
    u64 adj_total = 0; /* directed edges the context currently addresses */
    u64 adj_cap = 0, flag_cap = 0, out_cap = 0, valid_cap = 0, bkt_cap = 0, ent_cap = 0; /* buffers are kept across passes */
    bool flags_pending = false; /* sharded flow: gathered flag bytes wait to be OR-ed into the entries */
    bool adj_imported = false;
    u64 adj_span = 0;          /* size of the position space of d_adj in the sharded flow (compact: adj_total, padded: world*max) */
    bool half_complete = false; /* half/hcnt hold the survivor lists of ALL nodes */
    u64 start_cap = 0;
    u32 *d_extra_cnt = nullptr;
    u64 *d_extra_node = nullptr, *d_extra_key = nullptr;
    u32 *d_n_extra = nullptr;
    u32 extra_cap = 0;
    u32 n_extra = 0;
    u64 asym_local = 0;
    u64 dropped = 0; /* hits edge selection dropped in the local query range (0 => the selected edges are symmetric) */

    /* reduction */
    u8 *d_flag = nullptr;
    u64 *d_half = nullptr; /* [n][HALF_CAP] */
    u32 *d_hcnt = nullptr; /* [n] */
    bool use_half = false;
    u64 *d_wide = nullptr; /* nodes with more than HALF_CAP surviving edges */
    u32 *d_n_wide = nullptr;
    u32 wide_cap = 0, n_wide = 0;
    u8 *d_out_valid = nullptr;
    u64 *d_out_pos = nullptr;
    u64 *h_stage = nullptr; /* pinned: two halves of {sources, entries} for the chunked copy-out of disco_fetch_edges */
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    /* chain contraction (disco_contract_chains): composite edges, their links, absorbed flag per edge in fetch order */
    ChainEdgeOut *d_ch_comp = nullptr;
    ChainLinkOut *d_ch_links = nullptr;
    u8 *d_ch_dead = nullptr;
    u64 ch_comp_n = 0, ch_links_n = 0, ch_edges_n = 0, ch_comp_cap = 0, ch_links_cap = 0, ch_dead_cap = 0;
    bool ch_ready = false;
    char *d_text = nullptr; /* disco_format_edges: the edge lines of all files, file after file */
    u64 text_cap = 0, text_bytes = 0;
    u64 *d_out_src = nullptr, *d_out_ent = nullptr;
    u64 out_used = 0; /* chunk slots written by the emission (survivors + ~0 tails) */
    u64 n_out = 0;

    /* live kernel timing (HIP events on the stream the kernels are launched on) */
    hipEvent_t ev0[DISCO_PH_SLOTS] = {nullptr}, ev1[DISCO_PH_SLOTS] = {nullptr};
    bool ev_pending[DISCO_PH_SLOTS] = {false};
    float ph_ms[DISCO_PH_SLOTS] = {0};

    int phase = 0; /* 0 none, 1 reads, 2 index, 3 probe, 4 contained, 5 edges selected, 6 symmetrized, 7 marked, 8 emitted */

    /* multi-GPU flow (disco_comm_* / disco_dist_*): one context per rank */
    DiscoComm *comm = nullptr;
    DiscoComm *comm_bulk = nullptr;   /* a second communicator for the all-gather of the reads: it runs on bulk_stream while the */
    hipStream_t bulk_stream = nullptr; /* index is built and the own reads are probed (only verify needs the other ranks' rows)  */
    hipEvent_t ev_bulk = nullptr;
    bool wait_bulk_before_verify = false;
    u64 per = 0;        /* nodes per rank: ceil(n / world) rounded up to a multiple of 64 */
    u64 n_alloc = 0;    /* rows of the per-read tables (n on one GPU, world * per in the multi-GPU flow) */
    bool part_index = false;   /* the current multi-GPU pass keeps the index partitioned (DISCO_DIST_KEEP_INDEX_PARTITIONED) */
    u64 part_blo = 0, part_bhi = 0, part_nrec = 0; /* this rank's bucket range and records */
    u64 *d_pq_start = nullptr; /* scan scratch of the lookup exchange */
    u64 pq_start_cap = 0;
    bool dist_reads = false;   /* the read table was set through disco_dist_*: rows [q_lo, q_hi) are this rank's */
    bool dist_active = false;  /* the current pass is a multi-GPU pass in the regular regime (emission judges local pairs only) */
    u64 *d_route = nullptr;    /* [2 * DIST_MAX_WORLD] counters / cursors of the routing kernels */
    ulonglong2 *d_x16a = nullptr, *d_x16b = nullptr; /* 16-byte items: send (partitioned) / receive */
    u64 x16a_cap = 0, x16b_cap = 0;
    u32 *d_req_flat = nullptr, *d_req_s = nullptr, *d_req_r = nullptr; /* row requests: flat list, partitioned, received */
    u64 req_flat_cap = 0, req_s_cap = 0, req_r_cap = 0;
    u32 *d_rdeg_s = nullptr, *d_rdeg_r = nullptr, *d_rdata_s = nullptr; /* responses: degrees out / back, entries out */
    u64 rdeg_s_cap = 0, rdeg_r_cap = 0, rdata_s_cap = 0;
    u64 *d_rpos = nullptr;
    u64 rpos_cap = 0;
    u32 *d_nadj32_own = nullptr; /* rows fetched from other ranks as they arrive (4-byte entries), before rows_place_kernel puts them behind the own rows */
    u64 nadj_cap = 0, nadj_used = 0; /* nadj_used: entries of fetched rows behind the own rows so far */
    u32 *d_deg_tmp = nullptr;
    u64 deg_tmp_cap = 0;
    u64 *d_list_n = nullptr; /* length of the flat list being built */
    u64 *d_dense = nullptr; /* the job's reads at W words per row, for the all-gather (rows of the table are padded to 64 bytes) */
    u64 dense_cap = 0;
    ulonglong2 *d_push_r = nullptr; /* received half-edge pushes (alias of d_x16b while a pass is in flight) */
    u64 n_push_r = 0;
    /* ranks own loci: the reads — graph nodes — of a pass are dealt to the ranks by their read-level minimizer;
     * [home_lo, home_hi) is the id range the rank's reads arrived in (and the range whose containment flags it fixes). While such a pass
     * runs, q_lo / q_hi are POSITIONS in the rank's own list: [0, n_own) of d_order_own. */
    bool loci = false;
    bool runs_by_pos = false; /* d_runs is indexed by position in the processing order, not by read id */
    u8 *d_otab = nullptr;     /* [n_alloc] owner of every read */
    u32 *d_own_ids = nullptr; /* scratch: the own reads before they are grouped */
    u64 own_ids_cap = 0, n_own = 0;
    u64 home_lo = 0, home_hi = 0;
    u64 home_probes = 0; /* sum of len - k over the home range */
    /* the containment keys' reduce-scatter runs behind the pass on the second communicator (dist_mark_contained): whoever touches best[]
     * next waits for ev_keys */
    u64 *d_cb_all = nullptr; /* the ranks' "has a key" bitmaps, rank after rank */
    u64 cb_all_cap = 0;
    hipEvent_t ev_keys = nullptr, ev_keys_go = nullptr;
    bool keys_pending = false;
    disco_dist_info dinfo{};
    /* wait behind every all-to-all so that dinfo.ms shows the exchange and not its issue: DISCO_DIST_TIME_EXCHANGES=1, and — first contact
     *  — the FIRST pass of a context over RCCL with more than one rank (the pass a bench warms up with, the pass a watchdog
     * report is about: a stall then names the exchange it sits in); DISCO_DIST_NO_FIRST_CONTACT=1 takes that away */
    bool time_exchanges = false;
    unsigned dist_passes = 0;
};

/* ---------------------------------------------------------------------------------------------------------------- */
static int fail(disco_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                                          \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) return fail((c), e_ == hipErrorOutOfMemory ? DISCO_E_NOMEM : DISCO_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#define CHK(expr)                    \
    do {                             \
        int rc_ = (expr);            \
        if (rc_ != DISCO_OK) return rc_; \
    } while (0)

static void ph_begin(disco_ctx *c, int id)
{
    (void)hipEventRecord(c->ev0[id], c->stream);
}
static void ph_end(disco_ctx *c, int id)
{
    (void)hipEventRecord(c->ev1[id], c->stream);
    c->ev_pending[id] = true;
}
/* call after the stream has been synchronised */
static void ph_collect(disco_ctx *c)
{
    for (int i = 0; i < DISCO_PH_SLOTS; i++)
        if (c->ev_pending[i]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, c->ev0[i], c->ev1[i]) == hipSuccess) c->ph_ms[i] = ms;
            c->ev_pending[i] = false;
            if (i == DISCO_PH_INDEX) c->ph_ms[DISCO_PH_INDEX2] = 0; /* (a new index phase: its second bracket, if it has one, is collected below) */
        }
    /* the index phase in two brackets (the grouping sits between its count pass and its fill): reported as one */
    if (c->ph_ms[DISCO_PH_INDEX2] > 0) {
        c->ph_ms[DISCO_PH_INDEX] += c->ph_ms[DISCO_PH_INDEX2];
        c->ph_ms[DISCO_PH_INDEX2] = 0;
    }
}

template <typename T>
static int dev_alloc(disco_ctx *c, T **p, size_t count)
{
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (c->arena.base) {
        if (void *q = c->arena.alloc(bytes)) {
            *p = (T *)q;
            c->hbm_bytes += bytes;
            c->hbm_peak = std::max(c->hbm_peak, c->hbm_bytes);
            return DISCO_OK;
        }
    }
    tl_pass.dev_allocs++;
    HIPCHK(c, hipMalloc((void **)p, bytes));
    c->hbm_bytes += bytes;
    c->hbm_peak = std::max(c->hbm_peak, c->hbm_bytes);
    return DISCO_OK;
}

template <typename T>
static void dev_free(disco_ctx *c, T **p, size_t count)
{
    if (*p) {
        if (c->arena.owns(*p)) c->arena.release(*p);
        else {
            tl_pass.dev_frees++;
            (void)hipFree(*p);
        }
        size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        c->hbm_bytes = c->hbm_bytes >= bytes ? c->hbm_bytes - bytes : 0;
        *p = nullptr;
    }
}

/* grow-only buffer: reallocate when the need exceeds the capacity (steady-state passes allocate nothing) */
template <typename T>
static int ensure_cap(disco_ctx *c, T **p, u64 *cap, u64 need)
{
    if (*p && need <= *cap) return DISCO_OK;
    dev_free(c, p, *cap);
    *cap = 0;
    CHK(dev_alloc(c, p, need));
    *cap = std::max<u64>(need, 1);
    return DISCO_OK;
}

static DiscoView view(const disco_ctx *c)
{
    DiscoView v;
    v.reads = c->d_reads;
    v.len = c->d_len;
    v.n = c->n;
    v.S = c->S;
    v.k = c->k;
    v.m = disco_minimizer_len(c->k);
    if (const char *e = getenv("DISCO_MINIMIZER_LEN")) { /* tuning knob: odd, <= min(k, 23), k - m <= 63 */
        const int m = atoi(e);
        if (m >= 1 && (m & 1) && m <= v.m && c->k - m <= 63) v.m = m;
    }
    v.bkt = c->d_bkt;
    v.ent = c->d_ent;
    v.bshift = c->bshift;
    v.q_lo = c->q_lo;
    v.q_hi = c->q_hi;
    v.ctr = c->d_ctr;
    v.wq = c->d_wq;
    v.full = c->two_class ? c->d_full : nullptr;
    v.ovf = c->two_class ? c->d_ovf : nullptr;
    v.long_ids = c->two_class ? c->d_long_ids : nullptr;
    v.n_long = c->two_class ? (u32)c->n_long : 0u;
    v.SL = c->two_class ? c->S_ext : 0;
    v.tailb = c->two_class ? c->tailb : 0;
    return v;
}

/* the nodes the context works on as an OwnSet: the query range, or — inside a multi-GPU pass whose ranks own loci — the own list */
static OwnSet own_set(const disco_ctx *c)
{
    OwnSet o;
    o.otab = c->loci ? c->d_otab : nullptr;
    o.me = c->comm ? (u32)c->comm->rank : 0u;
    o.lo = c->q_lo;
    o.hi = c->q_hi;
    o.list = c->loci ? c->d_order_own : nullptr;
    o.n_own = c->loci ? c->n_own : 0;
    o.ids = c->loci ? c->d_own_ids : nullptr;
    return o;
}

/* best[] of the home range is final only when the reduce-scatter that runs behind the pass is through: the context's stream waits for it */
static int settle_keys(disco_ctx *c)
{
    if (c->keys_pending) {
        c->keys_pending = false;
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_keys, 0));
    }
    return DISCO_OK;
}

static int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int wave_grid(const disco_ctx *c, u64 items, int per_cu = 24)
{
    u64 g = (u64)c->n_cu * per_cu;
    if (items < g) g = items;
    return (int)std::max<u64>(g, 1);
}

/* grid for a work-queue kernel: every workgroup that can be resident (the queue balances the load), zeroes the queue */
template <typename K>
static int wq_grid(disco_ctx *c, K kernel, u64 items, const char *env, int cap = 32)
{
    (void)hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
    if (per_cu > cap) per_cu = cap;
    per_cu = env_int(env, per_cu);
    u64 g = (u64)c->n_cu * per_cu;
    const u64 chunks = (items + WQ_CHUNK - 1) / WQ_CHUNK;
    if (chunks < g) g = chunks;
    return (int)std::max<u64>(g, 1);
}

static int flat_grid(const disco_ctx *c, u64 items, int block = 256)
{
    u64 g = (items + block - 1) / block;
    u64 cap = (u64)c->n_cu * 16;
    return (int)std::max<u64>(std::min(g, cap), 1);
}

static int read_counters(disco_ctx *c)
{
    HIPCHK(c, hipMemcpyAsync(c->h_ctr, c->d_ctr, sizeof(c->h_ctr), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DISCO_OK;
}

static int zero_counter(disco_ctx *c, int idx)
{
    HIPCHK(c, hipMemsetAsync(c->d_ctr + idx, 0, sizeof(u64), c->stream));
    return DISCO_OK;
}

/* exclusive scan of in[0..n) into out[0..n) (+ out[n] = total when write_total); returns total through *total_host
 * when non-null (this synchronises the stream) */
template <typename InT, typename OutT>
static int scan_exclusive_on(disco_ctx *c, hipStream_t st, u64 **tile, size_t *tile_cap, u64 *total, const InT *in, u64 n, OutT *out, bool write_total, u64 *total_host)
{
    u64 nt = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (nt == 0) nt = 1;
    if (nt > *tile_cap) {
        dev_free(c, tile, *tile_cap);
        CHK(dev_alloc(c, tile, nt));
        *tile_cap = nt;
    }
    hipLaunchKernelGGL((scan_tile_sums_kernel<InT>), dim3((unsigned)nt), dim3(SCAN_BLOCK), 0, st, in, n, *tile);
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(1024), 0, st, *tile, nt, total);
    hipLaunchKernelGGL((scan_apply_kernel<InT, OutT>), dim3((unsigned)nt), dim3(SCAN_BLOCK), 0, st, in, n, *tile, out);
    if (write_total) hipLaunchKernelGGL((scan_write_total_kernel<OutT>), dim3(1), dim3(1), 0, st, total, out + n);
    HIPCHK(c, hipGetLastError());
    if (total_host) {
        HIPCHK(c, hipMemcpyAsync(total_host, total, sizeof(u64), hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
    }
    return DISCO_OK;
}

template <typename InT, typename OutT>
static int scan_exclusive(disco_ctx *c, const InT *in, u64 n, OutT *out, bool write_total, u64 *total_host)
{
    return scan_exclusive_on<InT, OutT>(c, c->stream, &c->d_tile, &c->tile_cap, c->d_total, in, n, out, write_total, total_host);
}

static void free_graph_state(disco_ctx *c)
{
    /* a key exchange that still runs behind a pass (second communicator, bulk_stream) reads and writes best[]: it is through before the
     * buffer goes back to the arena — a pass that failed after the exchange was issued, or new reads right after a pass, used to free
     * it under the collective */
    if (c->keys_pending) {
        c->keys_pending = false;
        if (c->ev_keys) (void)hipEventSynchronize(c->ev_keys);
    }
    dev_free(c, &c->d_bkt, c->bkt_cap);
    dev_free(c, &c->d_ent, c->ent_cap);
    dev_free(c, &c->d_rec, c->rec_cap);
    c->bkt_cap = c->ent_cap = c->rec_cap = 0;
    dev_free(c, &c->d_best, c->n_alloc);
    dev_free(c, &c->d_hits, c->hits_cap);
    c->hits_cap = 0;
    dev_free(c, &c->d_row_start, c->n);
    dev_free(c, &c->d_row_cnt, c->n);
    dev_free(c, &c->d_big_list, c->big_cap);
    dev_free(c, &c->d_big_cnt, c->big_cap);
    c->big_cap = 0;
    dev_free(c, &c->d_slow_list, c->slow_cap);
    c->slow_cap = 0;
    dev_free(c, &c->d_runs, c->runs_cap);
    c->runs_cap = c->runs_n = 0;
    c->runs_lpr = 0;
    dev_free(c, &c->d_contained, c->n_alloc);
    dev_free(c, &c->d_cbits, c->n_alloc / 64 + 1);
    dev_free(c, &c->d_dropbits, c->n_alloc / 64 + 1);
    dev_free(c, &c->d_drop_node, DROP_LIST_CAP);
    dev_free(c, &c->d_drop_key, DROP_LIST_CAP);
    dev_free(c, &c->d_adj_ref, c->n);
    dev_free(c, &c->d_adj_own, c->adj_cap);
    dev_free(c, &c->d_adj_spare, c->adj_spare_cap);
    dev_free(c, &c->d_start_tmp, c->start_cap);
    c->start_cap = 0;
    dev_free(c, &c->d_ocnt, c->ocnt_cap);
    dev_free(c, &c->d_okey, c->okey_cap);
    dev_free(c, &c->d_meta_ord, c->meta_cap);
    dev_free(c, &c->d_asked, c->asked_cap);
    c->meta_cap = 0;
    dev_free(c, &c->d_oslot, c->oslot_cap);
    dev_free(c, &c->d_order_own, c->order_cap);
    c->okey_cap = c->oslot_cap = c->order_cap = c->ocnt_cap = 0;
    c->d_order_used = nullptr;
    c->d_adj = nullptr;
    dev_free(c, &c->d_extra_cnt, c->n);
    dev_free(c, &c->d_extra_node, c->extra_cap);
    dev_free(c, &c->d_extra_key, c->extra_cap);
    c->extra_cap = 0;
    dev_free(c, &c->d_flag, c->flag_cap);
    dev_free(c, &c->d_half, c->n * HALF_CAP);
    dev_free(c, &c->d_hcnt, c->n);
    dev_free(c, &c->d_wide, c->wide_cap);
    dev_free(c, &c->d_n_wide, 1);
    c->wide_cap = 0;
    c->adj_total = c->adj_cap = c->adj_spare_cap = c->flag_cap = 0;
    dev_free(c, &c->d_out_valid, c->valid_cap);
    dev_free(c, &c->d_ch_comp, c->ch_comp_cap);
    dev_free(c, &c->d_ch_links, c->ch_links_cap);
    dev_free(c, &c->d_ch_dead, c->ch_dead_cap);
    dev_free(c, &c->d_text, c->text_cap);
    c->text_cap = c->text_bytes = 0;
    c->ch_comp_cap = c->ch_links_cap = c->ch_dead_cap = 0;
    c->ch_ready = false;
    dev_free(c, &c->d_out_pos, c->valid_cap + 1);
    dev_free(c, &c->d_out_src, c->out_cap);
    dev_free(c, &c->d_out_ent, c->out_cap);
    c->n_out = c->out_cap = c->valid_cap = c->out_used = 0;
    c->flags_pending = false;
    c->adj_imported = false;
    c->T = 0;
    dev_free(c, &c->d_x16a, c->x16a_cap);
    dev_free(c, &c->d_x16b, c->x16b_cap);
    dev_free(c, &c->d_req_flat, c->req_flat_cap);
    dev_free(c, &c->d_req_s, c->req_s_cap);
    dev_free(c, &c->d_req_r, c->req_r_cap);
    dev_free(c, &c->d_rdeg_s, c->rdeg_s_cap);
    dev_free(c, &c->d_rdeg_r, c->rdeg_r_cap);
    dev_free(c, &c->d_rdata_s, c->rdata_s_cap);
    dev_free(c, &c->d_rpos, c->rpos_cap);
    dev_free(c, &c->d_nadj32_own, c->nadj_cap);
    dev_free(c, &c->d_deg_tmp, c->deg_tmp_cap);
    dev_free(c, &c->d_dense, c->dense_cap);
    c->dense_cap = 0;
    dev_free(c, &c->d_pq_start, c->pq_start_cap);
    c->pq_start_cap = 0;
    c->x16a_cap = c->x16b_cap = c->req_flat_cap = c->req_s_cap = c->req_r_cap = c->rdeg_s_cap = c->rdeg_r_cap = c->rdata_s_cap = 0;
    c->rpos_cap = c->nadj_cap = c->nadj_used = c->deg_tmp_cap = 0;
    c->d_push_r = nullptr;
    c->n_push_r = 0;
    dev_free(c, &c->d_cb_all, c->cb_all_cap);
    c->cb_all_cap = 0;
    dev_free(c, &c->d_otab, c->n_alloc);
    dev_free(c, &c->d_own_ids, c->own_ids_cap);
    c->own_ids_cap = c->n_own = 0;
    c->loci = c->runs_by_pos = false;
}

/* the buffers of the long class (two classes of rows) — whatever of them exists: a failed two_class_alloc leaves some behind with
 * two_class still false, and the next table must not find a d_ovf sized for this one */
static void free_long_class(disco_ctx *c)
{
    dev_free(c, &c->d_full, c->n_long * (u64)c->S_ext);
    dev_free(c, &c->d_ovf, c->n_alloc);
    dev_free(c, &c->d_long_ids, c->n_long);
    dev_free(c, &c->d_lpos, c->n_long);
    dev_free(c, &c->d_lmeta, c->n_long);
    dev_free(c, &c->d_linfo, c->n_long);
    dev_free(c, &c->d_n_list, 1);
    c->two_class = false;
    c->n_long = c->reads_rows = 0;
    c->S_ext = 0;
    c->tailb = 0;
}

static void free_reads(disco_ctx *c)
{
    if (c->reads_owned) {
        dev_free(c, &c->d_reads, (c->reads_rows ? c->reads_rows : c->n_alloc) * (u64)c->S);
        dev_free(c, &c->d_len, c->n_alloc);
    }
    free_long_class(c);
    c->d_reads = nullptr;
    c->d_len = nullptr;
    c->reads_owned = false;
    c->h_len.clear();
    c->n = 0;
}

/* the count pass of the index build over the reads [lo, hi): with the minimizer runs for probe_runs_kernel where the shape of the job
 * allows them — a window of 17 m-mers (min-overlap 40, the default), 64-byte rows (reads up to 256 bases, so a read has at most 256
 * windows), and memory for 64 / 128 bytes per read (DISCO_RUNS_MAX_GB, default 16: beyond that — config 5's 2 x 10^8 reads on ONE
 * GPU — the buffer would crowd the hit buffer out of the 288 GB) — otherwise index_count_kernel and the wave-per-read probe_kernel. */
/* does disco_probe group a query range of nq reads itself, and into how many buckets (about one per read: a group has ~13 reads at
 * 30x, few groups share a bucket) */
static bool own_order_wanted(const disco_ctx *c, u64 nq, int *bits)
{
    const u64 order_min = getenv("DISCO_ORDER_MIN_READS") ? (u64)atoll(getenv("DISCO_ORDER_MIN_READS")) : 4096; /* tests: 1 */
    int b = 16;
    while (b < 27 && (1ull << b) < nq) ++b;
    *bits = b;
    return !c->order_external && !getenv("DISCO_NO_ORDER") && nq >= order_min && nq > 0;
}

/* the count pass in two steps so that it can run chunk by chunk behind an upload (disco_upload_reads): index_count_plan fixes, for the
 * whole range [lo, hi), whether the minimizer runs and the grouping's counting pass ride along and sizes their buffers;
 * index_count_chunk launches the kernel over a sub-range. */
struct IndexCountPlan {
    u64 lo = 0, hi = 0;
    int lpr = 0, nf = 0;
    u32 *ocnt = nullptr, *oslot = nullptr;
    u32 oshift = 0;
};

/* 32-bit words of minimizer runs per read for a table of 64-byte rows whose longest read has max_len bases, or 0: no runs */
static int runs_lpr_for(const disco_ctx *c, int nf, u32 max_len, u64 nloc)
{
    int lpr = 0;
    /* (index_runs_kernel keeps a block of NF order words in registers: one instantiation per window length — the reference's default
     * min-overlap 30 (NF 7), 35, BASELINE's 40 (NF 17), 45, 50) */
    /* round 6: every other window of 2 .. 64 m-mers — any min-overlap up to 95, k above 64 and minimizers of up to 31 bases included —
     * takes the instantiation with a run-time window length (DISCO_NO_GENERIC_RUNS=1: round 2's probe for those, as before) */
    const int m = view(c).m;
    const bool nf_built = (nf == 7 || nf == 12 || nf == 17 || nf == 22 || nf == 27) && m == RUNS_M;
    const bool nf_generic = nf >= 2 && nf <= 64 && m > 16 && m <= 31 && !getenv("DISCO_NO_GENERIC_RUNS");
    if ((nf_built || nf_generic) && max_len > (u32)c->k && !getenv("DISCO_NO_RUNS")) {
        const u32 maxwin = max_len - (u32)c->k;
        /* a read of W windows has about 2 W / (NF + 1) runs: 32 entries where that stays below 20 (room for the spread), else 64 */
        const u32 expect = 2u * maxwin / (u32)(nf + 1);
        lpr = maxwin <= 256 ? (maxwin <= 128 && expect <= 20 ? 16 : (expect <= 44 ? 32 : 0)) : 0;
        const double max_gb = getenv("DISCO_RUNS_MAX_GB") ? atof(getenv("DISCO_RUNS_MAX_GB")) : 16.0;
        if ((double)nloc * lpr * 4.0 > max_gb * 1e9) lpr = 0;
    }
    return lpr;
}

/* may a table of stride S (words) with n_long reads of more than 256 bases — the others at most short_max — go to two classes of rows? */
/* will_own: the caller is about to replace the table by one the context owns (an upload, the input stage) — the flag itself is only
 * set once the old table has been released and the new buffers exist (a table adopted from the caller must never be freed or kept) */
static bool two_class_ok(const disco_ctx *c, int S, u64 n, u64 n_long, u32 short_max, bool will_own = false)
{
    /* (a context with a communicator: only where its pass asks for it — dist_two_class, after the gather of the reads, over id ranges) */
    if (((c->comm || c->dist_reads) && !c->dist_two_class) || !(c->reads_owned || will_own) || c->prm.max_substitutions || getenv("DISCO_NO_TWO_CLASS")) return false;
    if (S <= VERIFY_SW || n_long == 0 || n_long * (u64)env_int("DISCO_TWO_CLASS_ONE_IN", 5) > n || n + n_long >= (1ull << 31)) return false;
    /* the short class takes the paths of a pure short set (minimizer runs, flat verify); the long one the lists those paths keep */
    return runs_lpr_for(c, c->k - view(c).m + 1, short_max, n) != 0;
}

static int two_class_alloc(disco_ctx *c, u64 n_long, int Sx, u32 short_max);

static int index_count_plan(disco_ctx *c, const DiscoView &v, u64 lo, u64 hi, IndexCountPlan *pl)
{
    const u64 nloc = hi - lo;
    *pl = IndexCountPlan();
    pl->lo = lo;
    pl->hi = hi;
    /* the counting pass of the grouping rides along when the indexed range is the query range (always, unless a caller narrows it) */
    c->order_counted = c->order_ready = false;
    int obits = 0;
    if (lo == c->q_lo && hi == c->q_hi && own_order_wanted(c, nloc, &obits) && !getenv("DISCO_NO_ORDER_FUSE")) {
        const u64 order_buckets = 1ull << obits;
        CHK(ensure_cap(c, &c->d_ocnt, &c->ocnt_cap, order_buckets + 1));
        CHK(ensure_cap(c, &c->d_oslot, &c->oslot_cap, nloc));
        HIPCHK(c, hipMemsetAsync(c->d_ocnt, 0, (order_buckets + 1) * sizeof(u32), c->stream));
        pl->ocnt = c->d_ocnt;
        pl->oslot = c->d_oslot;
        pl->oshift = 32u - (u32)obits;
        c->order_counted = true;
        c->order_counted_lo = lo;
        c->order_counted_hi = hi;
        c->order_counted_bits = obits;
    }
    c->runs_lpr = 0;
    c->runs_n = 0;
    c->runs_by_pos = false;
    const int nf = v.k - v.m + 1;
    const int lpr = c->S == VERIFY_SW ? runs_lpr_for(c, nf, c->max_len, nloc) : 0;
    if (lpr && nloc) {
        CHK(ensure_cap(c, &c->d_runs, &c->runs_cap, nloc * (u64)lpr));
        c->runs_lpr = lpr;
        c->runs_lo = lo;
        c->runs_n = nloc;
        pl->lpr = lpr;
        pl->nf = nf;
    }
    return DISCO_OK;
}

/* reads [a, b) of the planned range; rec_base = the records of read pl.lo */
template <bool COUNT>
static int index_count_chunk(disco_ctx *c, const DiscoView &v, const IndexCountPlan &pl, ulonglong2 *rec_base, u64 a, u64 b)
{
    if (b <= a) return DISCO_OK;
    const dim3 grid((unsigned)((b - a + 255) / 256));
    ulonglong2 *rec = rec_base + 2 * (a - pl.lo);
    u32 *oslot = pl.oslot ? pl.oslot + (a - pl.lo) : nullptr;
    if (pl.lpr) {
        u32 *runs = c->d_runs + (a - pl.lo) * (u64)pl.lpr;
#define DISCO_RUNS_LAUNCH(NF_)                                                                                                                              \
    do {                                                                                                                                                  \
        if (pl.lpr == 16) hipLaunchKernelGGL((index_runs_kernel<COUNT, NF_, 1>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, c->d_okey, a, b, runs, pl.ocnt, oslot, pl.oshift); \
        else hipLaunchKernelGGL((index_runs_kernel<COUNT, NF_, 2>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, c->d_okey, a, b, runs, pl.ocnt, oslot, pl.oshift);             \
    } while (0)
#define DISCO_RUNS_LAUNCH_RT(NFMAX_, LONGK_)                                                                                                                 \
    do {                                                                                                                                                  \
        if (pl.lpr == 16) hipLaunchKernelGGL((index_runs_kernel<COUNT, 0, 1, NFMAX_, LONGK_>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, c->d_okey, a, b, runs, pl.ocnt, oslot, pl.oshift); \
        else hipLaunchKernelGGL((index_runs_kernel<COUNT, 0, 2, NFMAX_, LONGK_>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, c->d_okey, a, b, runs, pl.ocnt, oslot, pl.oshift);             \
    } while (0)
        const bool built = v.m == RUNS_M && (pl.nf == 7 || pl.nf == 12 || pl.nf == 17 || pl.nf == 22 || pl.nf == 27);
        if (!built) { /* the window length as a run-time value: arrays for 32 or 64 m-mers */
            /* (the arrays are registers: an instantiation per size class keeps the waves per SIMD near the specialised kernels' — windows of 10
             * in arrays for 32 ran 9.8 ms at 20 M reads, in arrays for 12: 6.7) */
            if (c->k > 64) {
                if (pl.nf <= 48) DISCO_RUNS_LAUNCH_RT(48, true);
                else DISCO_RUNS_LAUNCH_RT(64, true);
            } else if (pl.nf <= 8) DISCO_RUNS_LAUNCH_RT(8, false);
            else if (pl.nf <= 12) DISCO_RUNS_LAUNCH_RT(12, false);
            else if (pl.nf <= 16) DISCO_RUNS_LAUNCH_RT(16, false);
            else if (pl.nf <= 24) DISCO_RUNS_LAUNCH_RT(24, false);
            else if (pl.nf <= 32) DISCO_RUNS_LAUNCH_RT(32, false);
            else if (pl.nf <= 48) DISCO_RUNS_LAUNCH_RT(48, false);
            else DISCO_RUNS_LAUNCH_RT(64, false);
        } else
            switch (pl.nf) {
            case 7: DISCO_RUNS_LAUNCH(7); break;
            case 12: DISCO_RUNS_LAUNCH(12); break;
            case 17: DISCO_RUNS_LAUNCH(17); break;
            case 22: DISCO_RUNS_LAUNCH(22); break;
            default: DISCO_RUNS_LAUNCH(27); break;
            }
#undef DISCO_RUNS_LAUNCH
#undef DISCO_RUNS_LAUNCH_RT
    } else
        if (c->k > 64) hipLaunchKernelGGL((index_count_kernel<COUNT, true>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, c->d_okey, a, b, pl.ocnt, oslot, pl.oshift);
        else hipLaunchKernelGGL(index_count_kernel<COUNT>, grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, c->d_okey, a, b, pl.ocnt, oslot, pl.oshift);
    HIPCHK(c, hipGetLastError());
    return DISCO_OK;
}

template <bool COUNT>
static int launch_index_count(disco_ctx *c, const DiscoView &v, ulonglong2 *rec, u64 lo, u64 hi)
{
    IndexCountPlan pl;
    CHK(index_count_plan(c, v, lo, hi, &pl));
    return index_count_chunk<COUNT>(c, v, pl, rec, lo, hi);
}

/* sizes of the index for the context's reads and the cleared bucket table (the start of disco_build_index, or of an upload that
 * counts while it copies) */
static int index_begin(disco_ctx *c)
{
    u64 T = 1024;
    int logT = 10;
    /* records are keyed by minimizer: about one distinct key per 3-4 records at 30x. T >= 2n buckets (measured at 50 M reads:
     * n / 2n / 4n / 8n buckets -> index 15.2 / 16.4 / 18.6 / 21.0 ms, probe 47.0 / 45.6 / 45.4 / 45.4 ms) */
    double tscale = 2.0;
    if (const char *e = getenv("DISCO_BUCKET_SCALE")) tscale = atof(e);
    while ((double)T < tscale * (double)c->n && logT < 32) {
        T <<= 1;
        logT++;
    }
    c->T = T;
    c->bshift = 64 - logT;
    CHK(ensure_cap(c, &c->d_bkt, &c->bkt_cap, T + 1));
    CHK(ensure_cap(c, &c->d_ent, &c->ent_cap, 2 * c->n));
    /* {key, record} of both end k-mers of every read, computed once by the count pass and re-read by the fill pass */
    CHK(ensure_cap(c, &c->d_rec, &c->rec_cap, 2 * c->n));
    CHK(ensure_cap(c, &c->d_okey, &c->okey_cap, c->n)); /* grouping keys of all reads (disco_probe orders its query range by them) */
    c->adj_imported = false;
    HIPCHK(c, hipMemsetAsync(c->d_bkt, 0, (T + 1) * sizeof(u32), c->stream));
    return DISCO_OK;
}

/* The contained rows for the host: (id, key) of the contained reads in ascending id (scan of the flags, gather: 12 bytes per row) into
 * buffers and pinned staging memory KEPT by the context — with three allocations of up to n + 1 words, the scan and two pageable
 * copies per call this was 32 ms of the host-to-host wall at 50 M reads; now 2 ms of device work and copy + the decode. On request the
 * same rows once more in the order of the contained-read files (a counting sort by containing read on the device: the host sort took
 * 0.23 s). Started by the fetch calls, on a side stream (measured: started from disco_mark_contained, to overlap the rest of the pass, the
 * extra kernels cost the pass 0.8 ms, and 3.5 ms with the grouping — more than they hid). Single-GPU passes with up to
 * DISCO_EAGER_ROWS_MAX (16 M) contained rows; otherwise disco_fetch_contained gathers on demand the old way. */
static int start_contained_rows(disco_ctx *c, bool grouped)
{
    const u64 nc = c->n_contained;
    if (c->crows_pending && c->crows_n == nc && (!grouped || c->cgrp_pending)) return DISCO_OK; /* already on their way */
    const bool have_rows = c->crows_pending && c->crows_n == nc;
    if (!have_rows) c->crows_pending = c->cgrp_pending = false;
    const u64 max_rows = getenv("DISCO_EAGER_ROWS_MAX") ? (u64)atoll(getenv("DISCO_EAGER_ROWS_MAX")) : (16ull << 20);
    if (c->comm || nc == 0 || nc > max_rows || c->n >= (1ull << 31)) return DISCO_OK;
    if (!c->aux_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_crows, hipEventDisableTiming));
        HIPCHK(c, hipMalloc((void **)&c->d_total2, sizeof(u64)));
    }
    CHK(ensure_cap(c, &c->d_cpos, &c->cpos_cap, c->n + 1));
    if (nc > c->crow_cap) {
        dev_free(c, &c->d_crow_id, c->crow_cap);
        dev_free(c, &c->d_crow_key, c->crow_cap);
        dev_free(c, &c->d_crow_len, c->crow_cap);
        c->crow_cap = 0;
        const u64 want = nc + nc / 4 + 1024;
        CHK(dev_alloc(c, &c->d_crow_id, want));
        CHK(dev_alloc(c, &c->d_crow_key, want));
        CHK(dev_alloc(c, &c->d_crow_len, want));
        c->crow_cap = want;
    }
    if (nc > c->crows_hcap && c->h_ring && !c->crows_in_ring && (u64)c->ring_half * 2 >= (nc + 1024) * 32) {
        /* the pinned ring of the input stage is idle between the input stage and the text output: the rows stage there (pinning
         * 140 MB for them took 0.05 s of the host thread that feeds the pass) */
        if (c->h_crows) (void)hipHostFree(c->h_crows);
        c->h_crows = c->h_ring;
        c->crows_hcap = (u64)c->ring_half * 2 / 32;
        c->crows_in_ring = true;
    }
    if (nc > c->crows_hcap) {
        if (c->h_crows && !c->crows_in_ring) (void)hipHostFree(c->h_crows);
        c->crows_in_ring = false;
        c->h_crows = nullptr;
        c->crows_hcap = 0;
        const u64 want = nc + nc / 4 + 1024;
        if (hipHostMalloc(&c->h_crows, want * 32) != hipSuccess) { /* (both orders) no pinned memory to be had: the on-demand path */
            c->h_crows = nullptr;
            (void)hipGetLastError();
            return DISCO_OK;
        }
        c->crows_hcap = want;
    }
    if (!have_rows) {
    /* (the flags are complete: disco_mark_contained synchronised the context's stream) */
    CHK((scan_exclusive_on<u8, u32>(c, c->aux_stream, &c->d_tile2, &c->tile2_cap, c->d_total2, c->d_contained, c->n, c->d_cpos, false, nullptr)));
    hipLaunchKernelGGL(contain_rows32_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->aux_stream, c->d_best, c->d_contained, c->d_cpos, c->n, c->d_crow_id, c->d_crow_key);
    HIPCHK(c, hipGetLastError());
    u64 *hkey = (u64 *)c->h_crows;
    u32 *hid = (u32 *)(hkey + c->crows_hcap), *hln = hid + c->crows_hcap;
    hipLaunchKernelGGL(crow_lens_kernel, dim3(flat_grid(c, nc)), dim3(256), 0, c->aux_stream, (const u32 *)c->d_crow_id, (const u64 *)c->d_crow_key, nc, (const u16 *)c->d_len, c->d_crow_len);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(hkey, c->d_crow_key, nc * 8, hipMemcpyDeviceToHost, c->aux_stream));
    HIPCHK(c, hipMemcpyAsync(hid, c->d_crow_id, nc * 4, hipMemcpyDeviceToHost, c->aux_stream));
    HIPCHK(c, hipMemcpyAsync(hln, c->d_crow_len, nc * 4, hipMemcpyDeviceToHost, c->aux_stream));
    HIPCHK(c, hipEventRecord(c->ev_crows, c->aux_stream));
    c->crows_n = nc;
    c->crows_pending = true;
    }
    /* ... and, on request, once more in the order of the contained-read files (crow_* kernels), behind the first copy on the same stream */
    if (grouped && !c->cgrp_pending) {
        CHK(ensure_cap(c, &c->d_cgrp_cur, &c->cgrp_cur_cap, c->n + 1));
        if (nc > c->cgrp_cap) {
            dev_free(c, &c->d_cgrp_id, c->cgrp_cap);
            dev_free(c, &c->d_cgrp_key, c->cgrp_cap);
            c->cgrp_cap = 0;
            const u64 want = nc + nc / 4 + 1024;
            CHK(dev_alloc(c, &c->d_cgrp_id, want));
            CHK(dev_alloc(c, &c->d_cgrp_key, want));
            c->cgrp_cap = want;
        }
        if (!c->d_cgrp_big) CHK(dev_alloc(c, &c->d_cgrp_big, 1));
        HIPCHK(c, hipMemsetAsync(c->d_cpos, 0, (c->n + 1) * sizeof(u32), c->aux_stream)); /* (the flag scan has served the gather) */
        HIPCHK(c, hipMemsetAsync(c->d_cgrp_big, 0, sizeof(u64), c->aux_stream));
        hipLaunchKernelGGL(crow_count_kernel, dim3(flat_grid(c, nc)), dim3(256), 0, c->aux_stream, (const u64 *)c->d_crow_key, nc, c->d_cpos);
        CHK((scan_exclusive_on<u32, u32>(c, c->aux_stream, &c->d_tile2, &c->tile2_cap, c->d_total2, c->d_cpos, c->n + 1, c->d_cgrp_cur, false, nullptr)));
        hipLaunchKernelGGL(crow_place_kernel, dim3(flat_grid(c, nc)), dim3(256), 0, c->aux_stream, (const u32 *)c->d_crow_id, (const u64 *)c->d_crow_key, nc, c->d_cgrp_cur, c->d_cgrp_id,
                           c->d_cgrp_key);
        hipLaunchKernelGGL(crow_sort_groups_kernel, dim3(flat_grid(c, nc)), dim3(256), 0, c->aux_stream, c->d_cgrp_id, c->d_cgrp_key, nc, c->d_cgrp_big);
        HIPCHK(c, hipGetLastError());
        u64 *gkey = (u64 *)((char *)c->h_crows + c->crows_hcap * 16);
        u32 *gid = (u32 *)(gkey + c->crows_hcap), *gln = gid + c->crows_hcap;
        /* (the lengths of the rows by id have left: their device array serves the grouped order) */
        hipLaunchKernelGGL(crow_lens_kernel, dim3(flat_grid(c, nc)), dim3(256), 0, c->aux_stream, (const u32 *)c->d_cgrp_id, (const u64 *)c->d_cgrp_key, nc, (const u16 *)c->d_len, c->d_crow_len);
        HIPCHK(c, hipMemcpyAsync(gkey, c->d_cgrp_key, nc * 8, hipMemcpyDeviceToHost, c->aux_stream));
        HIPCHK(c, hipMemcpyAsync(gid, c->d_cgrp_id, nc * 4, hipMemcpyDeviceToHost, c->aux_stream));
        HIPCHK(c, hipMemcpyAsync(gln, c->d_crow_len, nc * 4, hipMemcpyDeviceToHost, c->aux_stream));
        HIPCHK(c, hipMemcpyAsync(&c->h_cgrp_big, c->d_cgrp_big, sizeof(u64), hipMemcpyDeviceToHost, c->aux_stream));
        c->cgrp_pending = true;
    }
    return DISCO_OK;
}

/* the side stream reads best[] and the flags: nothing may rewrite them before it is done */
static int settle_contained_rows(disco_ctx *c)
{
    if (c->crows_pending) HIPCHK(c, hipStreamSynchronize(c->aux_stream));
    c->crows_pending = false;
    c->cgrp_pending = false;
    return DISCO_OK;
}

/* the pinned ring is wanted for text again (input stage, edge text): rows that stage in it and have not been fetched are given up
 * (the fetch gathers them again on demand) */
static int ring_back_from_rows(disco_ctx *c)
{
    if (!c->crows_in_ring) return DISCO_OK;
    if (c->crows_pending && c->aux_stream) HIPCHK(c, hipStreamSynchronize(c->aux_stream));
    c->crows_pending = c->cgrp_pending = false;
    c->h_crows = nullptr;
    c->crows_hcap = 0;
    c->crows_in_ring = false;
    return DISCO_OK;
}

/* the hit buffer the input stage asked for on a thread of its own: take it over (a device allocation costs 17-33 ms per GB on these
 * boxes — 0.5 to 0.9 s for the 28 GB of 50 M reads — and nothing needs the buffer before the probe) */
static void settle_hits_prealloc(disco_ctx *c)
{
    if (!c->hits_prealloc.joinable()) return;
    c->hits_prealloc.join();
    if (c->prealloc_ptr) {
        if (c->prealloc_cap > c->hits_cap) {
            dev_free(c, &c->d_hits, c->hits_cap);
            c->d_hits = c->prealloc_ptr;
            c->hits_cap = c->prealloc_cap;
            c->hbm_bytes += c->prealloc_cap * 8;
        } else
            (void)hipFree(c->prealloc_ptr);
        c->prealloc_ptr = nullptr;
        c->prealloc_cap = 0;
    }
}

static int dist_mark_contained(disco_ctx *c); /* multi-GPU flow, below */
static int dist_partitioned_probe(disco_ctx *c);

/* ================================================================================================================ */
extern "C" {

int disco_abi_version(void) { return DISCO_ABI_VERSION; }

int disco_create(int device, const disco_params *p, disco_ctx **out)
{
    if (!p || !out) return fail(nullptr, DISCO_E_ARG, "disco_create: null argument");
    if (p->min_overlap < 2 || p->min_overlap - 1 > DISCO_MAX_K)
        return fail(nullptr, DISCO_E_UNSUPPORTED, "disco_create: min_overlap %u unsupported (k = min_overlap-1 must be in [1,%d])", p->min_overlap, DISCO_MAX_K);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, DISCO_E_HIP, "disco_create: no HIP device (%s)", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, DISCO_E_ARG, "disco_create: device %d out of range (%d devices)", device, ndev);
    disco_ctx *c = new (std::nothrow) disco_ctx();
    if (!c) return fail(nullptr, DISCO_E_NOMEM, "disco_create: out of host memory");
    c->device = device;
    c->prm = *p;
    if (c->prm.max_edges_per_kmer == 0) c->prm.max_edges_per_kmer = 4;
    if (c->prm.max_substitutions > 32767) c->prm.max_substitutions = 32767; /* no read is longer (15-bit length field) */
    c->k = (int)p->min_overlap - 1;
#define CREATE_CHK(call)                                                                              \
    do {                                                                                              \
        hipError_t e2 = (call);                                                                       \
        if (e2 != hipSuccess) {                                                                       \
            int rc = fail(nullptr, DISCO_E_HIP, "disco_create: %s failed: %s", #call, hipGetErrorString(e2)); \
            delete c;                                                                                 \
            return rc;                                                                                \
        }                                                                                             \
    } while (0)
    CREATE_CHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, device));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    CREATE_CHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    CREATE_CHK(hipMalloc((void **)&c->d_ctr, sizeof(u64) * CTR_COUNT));
    CREATE_CHK(hipMemset(c->d_ctr, 0, sizeof(u64) * CTR_COUNT));
    CREATE_CHK(hipMalloc((void **)&c->d_total, sizeof(u64)));
    CREATE_CHK(hipMalloc((void **)&c->d_wq, sizeof(u64) * WQ_WORDS)); /* (WQ_NQ sub-queues, a 128-byte slot each: wq_grab_split) */
    CREATE_CHK(hipMalloc((void **)&c->d_bump, sizeof(u64)));
    CREATE_CHK(hipMalloc((void **)&c->d_n_big, sizeof(u32)));
    CREATE_CHK(hipMalloc((void **)&c->d_n_slow, sizeof(u32)));
    CREATE_CHK(hipMalloc((void **)&c->d_n_extra, sizeof(u32)));
    for (int i = 0; i < DISCO_PH_SLOTS; i++) {
        CREATE_CHK(hipEventCreate(&c->ev0[i]));
        CREATE_CHK(hipEventCreate(&c->ev1[i]));
    }
#undef CREATE_CHK
    *out = c;
    return DISCO_OK;
}

void disco_destroy(disco_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_graph_state(c);
    free_reads(c);
    dev_free(c, &c->d_tile, c->tile_cap);
    (void)hipFree(c->d_ctr);
    (void)hipFree(c->d_total);
    (void)hipFree(c->d_wq);
    (void)hipFree(c->d_bump);
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    for (int i = 0; i < 2; i++)
        if (c->ev_stage[i]) (void)hipEventDestroy(c->ev_stage[i]);
    (void)hipFree(c->d_n_big);
    (void)hipFree(c->d_n_slow);
    (void)hipFree(c->d_n_extra);
    dev_free(c, &c->d_probe_rare, 1);
    dev_free(c, &c->d_route, 2 * DIST_MAX_WORLD);
    dev_free(c, &c->d_list_n, 1);
    delete c->comm_bulk;
    delete c->comm;
    c->comm = c->comm_bulk = nullptr;
    if (c->bulk_stream) (void)hipStreamDestroy(c->bulk_stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->aux_stream) {
        (void)hipStreamSynchronize(c->aux_stream);
        (void)hipStreamDestroy(c->aux_stream);
        (void)hipEventDestroy(c->ev_crows);
        (void)hipFree(c->d_total2);
    }
    dev_free(c, &c->d_tile2, c->tile2_cap);
    dev_free(c, &c->d_cpos, c->cpos_cap);
    dev_free(c, &c->d_crow_id, c->crow_cap);
    dev_free(c, &c->d_crow_key, c->crow_cap);
    dev_free(c, &c->d_crow_len, c->crow_cap);
    dev_free(c, &c->d_cgrp_cur, c->cgrp_cur_cap);
    dev_free(c, &c->d_cgrp_id, c->cgrp_cap);
    dev_free(c, &c->d_cgrp_key, c->cgrp_cap);
    dev_free(c, &c->d_cgrp_big, 1);
    dev_free(c, &c->d_fetch_src, c->fetch_cap);
    dev_free(c, &c->d_fetch_ent, c->fetch_cap);
    if (c->h_crows && !c->crows_in_ring) (void)hipHostFree(c->h_crows);
    settle_hits_prealloc(c);
    dev_free(c, &c->d_ingest, c->ingest_cap);
    if (c->h_ring) (void)hipHostFree(c->h_ring);
    for (int i = 0; i < 2; i++)
        if (c->ev_ring[i]) (void)hipEventDestroy(c->ev_ring[i]);
    dev_free(c, &c->d_rec_of_read, c->rec_of_read_cap);
    for (int i = 0; i < 3; i++) {
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_unpacked[i]) (void)hipEventDestroy(c->ev_unpacked[i]);
    }
    if (c->ev_bulk) (void)hipEventDestroy(c->ev_bulk);
    if (c->ev_keys) (void)hipEventDestroy(c->ev_keys);
    if (c->ev_keys_go) (void)hipEventDestroy(c->ev_keys_go);
    for (int i = 0; i < DISCO_PH_SLOTS; i++) {
        if (c->ev0[i]) (void)hipEventDestroy(c->ev0[i]);
        if (c->ev1[i]) (void)hipEventDestroy(c->ev1[i]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->arena.base) (void)hipFree(c->arena.base); /* (every buffer carved out of it was released above or dies with it) */
    delete c;
}

const char *disco_last_error(const disco_ctx *c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int disco_set_stream(disco_ctx *c, void *hip_stream)
{
    if (!c) return DISCO_E_ARG;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return DISCO_OK;
}

int disco_synchronize(disco_ctx *c)
{
    if (!c) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DISCO_OK;
}

int disco_pack_ascii(const char *seq, uint32_t len, uint64_t *out_words)
{
    if (!seq || !out_words) return DISCO_E_ARG;
    uint32_t nw = (len + 31) / 32;
    for (uint32_t w = 0; w < nw; w++) out_words[w] = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint64_t b;
        switch (seq[i]) {
        case 'A': b = 0; break;
        case 'C': b = 1; break;
        case 'G': b = 2; break;
        case 'T': b = 3; break;
        default: return DISCO_E_ARG;
        }
        out_words[i >> 5] |= b << (62 - 2 * (i & 31));
    }
    return DISCO_OK;
}

void *disco_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

void disco_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

/* keep != null: the caller owns-and-fills the table itself (upload, generator). If the context already holds an owned table of exactly
 * this shape, NOTHING is freed — table, index, hit buffer, lists all keep their size and the next pass finds them as a repeated pass
 * does (*keep = true); dropping and re-allocating tens of GB per read set cost 60 ms at best and, on some boxes, 1.2–1.9 s in the
 * first allocation afterwards */
static int set_reads_common(disco_ctx *c, u64 n, uint32_t stride, bool *keep = nullptr)
{
    if (n >= (1ull << 31)) return fail(c, DISCO_E_UNSUPPORTED, "more than 2^31 reads per context are not supported");
    if (stride == 0 || stride > 1024) return fail(c, DISCO_E_ARG, "stride_words %u out of range", stride);
    c->index_counted = false;
    if (c->crows_pending && c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    c->crows_pending = false;
    c->cgrp_pending = false;
    if (keep) {
        *keep = c->reads_owned && !c->two_class && !c->dist_reads && !c->comm && c->d_reads && c->d_len && n > 0 && c->n == n && c->n_alloc == n && c->S == (int)stride &&
                !getenv("DISCO_NO_BUFFER_REUSE");
        if (*keep) {
            c->h_len_ok = false; /* (not cleared: the next upload overwrites it in place) */
            c->n_out = c->out_used = 0;
            c->adj_total = 0;
            c->flags_pending = false;
            c->adj_imported = false;
            c->ch_ready = false;
            c->d_adj = nullptr;
            c->d_order_used = nullptr;
            c->dist_active = false;
            c->q_lo = 0;
            c->q_hi = n;
            c->phase = 0;
            return DISCO_OK;
        }
    }
    free_graph_state(c);
    free_reads(c);
    c->n = n;
    c->n_alloc = n;
    c->dist_reads = false;
    c->dist_active = false;
    c->S = (int)stride;
    c->q_lo = 0;
    c->q_hi = n;
    c->phase = 0;
    return DISCO_OK;
}

/* the copy stream starts behind whatever the context's stream still holds (a pass the caller did not wait for may be reading the
 * table, or the memory the copies are about to fill) */
static int copy_stream_after_stream(disco_ctx *c)
{
    HIPCHK(c, hipEventRecord(c->ev_unpacked[0], c->stream));
    HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_unpacked[0], 0));
    return DISCO_OK;
}

static int validate_reads(disco_ctx *c)
{
    CHK(zero_counter(c, CTR_BAD_LEN));
    CHK(zero_counter(c, CTR_MAX_LEN));
    CHK(zero_counter(c, CTR_MIN_LEN));
    if (c->n) hipLaunchKernelGGL(validate_len_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_len, c->n, c->S, (int)c->prm.min_overlap, c->d_ctr);
    CHK(read_counters(c));
    c->min_len = 0xFFFFu - (u32)c->h_ctr[CTR_MIN_LEN];
    if (c->h_ctr[CTR_BAD_LEN])
        return fail(c, DISCO_E_ARG, "%llu reads have a length outside (min_overlap=%u, min(32767, 32*stride)]", (unsigned long long)c->h_ctr[CTR_BAD_LEN], c->prm.min_overlap);
    c->max_len = (u32)c->h_ctr[CTR_MAX_LEN];
    c->phase = 1;
    return DISCO_OK;
}

/* disco_upload_reads (stride_words > 0: one stride on the host) and disco_upload_reads_ragged (stride_words = 0: the reads back to back,
 * ceil(len / 32) words each — the form the reference itself keeps them in, BG/HashTable.cpp:456-477) */
static int upload_reads_impl(disco_ctx *c, const char *who, const uint64_t *packed, uint32_t stride_words, const uint16_t *len, uint64_t n)
{
    if (!c || (n && (!packed || !len))) return c ? fail(c, DISCO_E_ARG, "%s: null argument", who) : DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const bool ragged = stride_words == 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto lapms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    double t_scan = 0, t_alloc = 0, t_issued = 0, t_mirror = 0;
    u64 CH = getenv("DISCO_UPLOAD_CHUNK") ? (u64)atoll(getenv("DISCO_UPLOAD_CHUNK")) : (4ull << 20);
    if (!ragged) CH = std::min<u64>(CH, (512ull << 20) / ((u64)stride_words * 8)); /* (a chunk of wide rows: at most 512 MB per staging buffer) */
    CH = std::max<u64>((CH + 255) & ~255ull, 256);
    const u64 n_chunks = (n + CH - 1) / CH;
    /* the lengths are checked where they are (the host has them): min_overlap < len <= min(32767, 32 * stride) (BG/Dataset.cpp:305,
     * BG/HashTable.cpp:531); longest / shortest decide the kernel variants of the pass, the long reads the layout; ragged: the words of
     * every chunk of CH reads */
    std::atomic<u64> a_bad{0}, a_long{0};
    std::atomic<u32> a_max{0}, a_min{0xFFFFu}, a_smax{0};
    std::vector<u64> chunk_words(n_chunks + 1, 0);
    {
        const u32 mo = c->prm.min_overlap, cap = ragged ? 32767u : std::min<u32>(32767u, stride_words * 32u);
        std::mutex mu;
        parallel_for(n, [&](u64 b0, u64 e0) {
            u64 bad = 0, nlong = 0;
            u32 mx = 0, mn = 0xFFFFu, smx = 0;
            for (u64 k = b0 / CH; k * CH < e0; k++) { /* the piece of chunk k inside [b0, e0) */
                u64 words = 0;
                for (u64 i = std::max<u64>(b0, k * CH); i < std::min<u64>(e0, (k + 1) * CH); i++) {
                    const u32 L = len[i];
                    bad += (L <= mo || L > cap);
                    mx = std::max(mx, L);
                    mn = std::min(mn, L);
                    nlong += L > (u32)DISCO_SHORT_MAX;
                    if (L <= (u32)DISCO_SHORT_MAX) smx = std::max(smx, L);
                    words += (L + 31u) >> 5;
                }
                std::lock_guard<std::mutex> lk(mu);
                chunk_words[k + 1] += words;
            }
            a_bad += bad;
            a_long += nlong;
            u32 cur = a_max.load();
            while (mx > cur && !a_max.compare_exchange_weak(cur, mx)) {}
            cur = a_min.load();
            while (mn < cur && !a_min.compare_exchange_weak(cur, mn)) {}
            cur = a_smax.load();
            while (smx > cur && !a_smax.compare_exchange_weak(cur, smx)) {}
        });
        for (u64 k = 0; k < n_chunks; k++) chunk_words[k + 1] += chunk_words[k];
    }
    if (ragged) stride_words = std::max<u32>(1, (a_max.load() + 31) / 32);
    t_scan = lapms();
    /* rows are padded to a multiple of 8 words = 64 B so that a candidate row fetch touches whole, aligned HBM sectors */
    const uint32_t dstride = (stride_words + 7u) & ~7u;
    if (a_bad.load()) {
        /* the context is left with no reads: the stride of the empty table is any valid one (the bad length itself may ask for more
         * than set_reads_common accepts, and its refusal would leave the PREVIOUS reads and graph in place) */
        (void)set_reads_common(c, 0, VERIFY_SW, nullptr);
        return fail(c, DISCO_E_ARG, "%llu reads have a length outside (min_overlap=%u, min(32767, 32*stride)]", (unsigned long long)a_bad.load(), c->prm.min_overlap);
    }
    /* a few long reads among short ones: the chunks are unpacked per class (two classes of rows, disco_kernels.h) — the table of one
     * stride is never made on the device */
    const u64 n_long = a_long.load();
    const bool classes = n && two_class_ok(c, (int)dstride, n, n_long, a_smax.load(), true);
    bool kept = false;
    CHK(set_reads_common(c, n, classes ? (uint32_t)VERIFY_SW : dstride, classes ? nullptr : &kept));
    c->reads_owned = true; /* the old table is gone (or kept, and then it was the context's own): whatever is allocated from here on is released by free_reads */
    if (classes) {
        CHK(dev_alloc(c, &c->d_reads, (n + n_long) * 8));
        CHK(dev_alloc(c, &c->d_len, n));
        CHK(two_class_alloc(c, n_long, (int)dstride, a_smax.load()));
    } else if (!kept) {
        CHK(dev_alloc(c, &c->d_reads, n * (u64)dstride));
        CHK(dev_alloc(c, &c->d_len, n));
    }
    c->reads_owned = true;
    c->max_len = n ? (classes ? a_smax.load() : a_max.load()) : 0;
    c->max_len_all = n ? a_max.load() : 0;
    c->min_len = n ? a_min.load() : 0;
    if (n) {
        /* The copy runs in chunks on a stream of its own; behind it, chunk by chunk on the context's stream: rows spread to the 64-byte
         * stride of the table (host rows at the words they use: 5 of 8 at 150 bp — 2.0 instead of 3.2 GB over the link; a 2-D copy of 40-byte
         * rows ran at 8.7 GB/s) and the COUNT PASS OF THE INDEX over the chunk's reads (index_runs_kernel: 9 ms at 50 M reads, hidden behind
         * the 38 ms of the copy) — disco_build_index then starts at its scan. The call returns when the host buffer is free again. */
        if (!c->copy_stream) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            for (int i = 0; i < 3; i++) {
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
                HIPCHK(c, hipEventCreateWithFlags(&c->ev_unpacked[i], hipEventDisableTiming));
            }
        }
        CHK(copy_stream_after_stream(c));
        const bool direct = !ragged && !classes && dstride == stride_words; /* the host rows ARE the table's rows: copied where they belong */
        /* (two classes: the index is counted by disco_build_index, over the classes) */
        const bool eager = !c->comm && !classes && !getenv("DISCO_NO_EAGER_INDEX");
        HIPCHK(c, hipMemcpyAsync(c->d_len, len, n * 2, hipMemcpyHostToDevice, c->stream));
        u64 ring_words = 0; /* words of the largest chunk as it lies on the host */
        for (u64 k = 0; k < n_chunks; k++)
            ring_words = std::max<u64>(ring_words, ragged ? chunk_words[k + 1] - chunk_words[k] : (std::min<u64>(n, (k + 1) * CH) - k * CH) * (u64)stride_words);
        if (!direct) CHK(ensure_cap(c, &c->d_dense, &c->dense_cap, 3 * ring_words));
        u64 *woff = nullptr;
        u32 *nw = nullptr;
        if (ragged) { /* word offset of every read: scan of ceil(len / 32) */
            CHK(dev_alloc(c, &woff, n));
            CHK(dev_alloc(c, &nw, n));
            hipLaunchKernelGGL(words_per_read_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, (const u16 *)c->d_len, n, nw);
            CHK((scan_exclusive<u32, u64>(c, nw, n, woff, false, nullptr)));
        }
        if (classes) { /* the long reads' numbers */
            hipLaunchKernelGGL(class_flag_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, (const u16 *)c->d_len, n, c->d_ovf, c->d_ctr);
            CHK((scan_exclusive<u32, u32>(c, c->d_ovf, n, c->d_ovf, false, nullptr)));
            hipLaunchKernelGGL(class_ids_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, (const u16 *)c->d_len, (const u32 *)c->d_ovf, n, c->d_long_ids);
        }
        t_alloc = lapms();
        IndexCountPlan pl;
        DiscoView v;
        if (eager) {
            CHK(index_begin(c));
            v = view(c);
            CHK(index_count_plan(c, v, 0, n, &pl));
        }
        for (u64 k = 0; k < n_chunks; k++) {
            const u64 lo = k * CH, hi = std::min<u64>(n, lo + CH);
            const int b = (int)(k % 3);
            if (!direct) {
                u64 *ring = c->d_dense + (u64)b * ring_words;
                const u64 w0 = ragged ? chunk_words[k] : lo * (u64)stride_words, w1 = ragged ? chunk_words[k + 1] : hi * (u64)stride_words;
                if (k >= 3) HIPCHK(c, hipStreamWaitEvent(c->copy_stream, c->ev_unpacked[b], 0));
                HIPCHK(c, hipMemcpyAsync(ring, packed + w0, (w1 - w0) * 8, hipMemcpyHostToDevice, c->copy_stream));
                HIPCHK(c, hipEventRecord(c->ev_copied[b], c->copy_stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
                hipLaunchKernelGGL(unpack_reads_kernel, dim3(flat_grid(c, (hi - lo) * (u64)c->S)), dim3(256), 0, c->stream, (const u64 *)ring, ragged ? 0 : (int)stride_words, (const u64 *)woff, w0,
                                   (const u16 *)c->d_len, lo, hi, c->S, classes ? (u32)DISCO_SHORT_MAX : 0xFFFFu, c->d_reads);
                if (classes)
                    hipLaunchKernelGGL(unpack_long_reads_kernel, dim3(flat_grid(c, n_long * ((u64)dstride + 8))), dim3(256), 0, c->stream, (const u64 *)ring, ragged ? 0 : (int)stride_words,
                                       (const u64 *)woff, w0, (const u16 *)c->d_len, lo, hi, (const u32 *)c->d_long_ids, n_long, n, (int)dstride, c->tailb, c->d_full, c->d_reads);
                HIPCHK(c, hipEventRecord(c->ev_unpacked[b], c->stream));
            } else { /* (one 1-D copy per chunk: a 2-D copy whose width equals both pitches ran at a third of the rate) */
                HIPCHK(c, hipMemcpyAsync(c->d_reads + lo * dstride, packed + lo * dstride, (hi - lo) * (u64)dstride * 8, hipMemcpyHostToDevice, c->copy_stream));
                HIPCHK(c, hipEventRecord(c->ev_copied[b], c->copy_stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_copied[b], 0));
            }
            if (eager) CHK((index_count_chunk<true>(c, v, pl, c->d_rec, lo, hi)));
        }
        HIPCHK(c, hipGetLastError());
        t_issued = lapms();
        /* the host's copy of the lengths (result decoding) while the chunks travel */
        if (c->h_len.size() != n) c->h_len.resize(n);
        uint16_t *hl = c->h_len.data();
        parallel_for(n, [&, hl](u64 b0, u64 e0) { memcpy(hl + b0, len + b0, (e0 - b0) * 2); });
        c->h_len_ok = true;
        t_mirror = lapms();
        HIPCHK(c, hipStreamSynchronize(c->copy_stream));
        if (getenv("DISCO_VERBOSE"))
            fprintf(stderr, "[disco] %s: lengths checked %.1f ms, buffers %.1f, chunks issued %.1f, lengths mirrored %.1f, copies done %.1f (%s%s)\n", who, t_scan, t_alloc, t_issued,
                    t_mirror, lapms(), classes ? "two classes of rows" : "one stride", eager ? ", index counted behind the copies" : "");
        if (woff) { /* (read by the unpack kernels on the context's stream) */
            HIPCHK(c, hipStreamSynchronize(c->stream));
            dev_free(c, &woff, n);
            dev_free(c, &nw, n);
        }
        c->index_counted = eager;
    } else
        c->h_len.clear();
    c->phase = 1;
    return DISCO_OK;
}

int disco_upload_reads(disco_ctx *c, const uint64_t *packed, uint32_t stride_words, const uint16_t *len, uint64_t n)
{
    DISCO_TRACE("disco_upload_reads");
    if (c && stride_words == 0) return fail(c, DISCO_E_ARG, "stride_words 0 out of range");
    return upload_reads_impl(c, "disco_upload_reads", packed, stride_words, len, n);
}

int disco_upload_reads_ragged(disco_ctx *c, const uint64_t *words, const uint16_t *len, uint64_t n)
{
    DISCO_TRACE("disco_upload_reads_ragged");
    return upload_reads_impl(c, "disco_upload_reads_ragged", words, 0, len, n);
}

/* ================================================================================================================
 * input stage on the GPU (kernels: disco_ingest.h)
 * ============================================================================================================== */
namespace {
struct IngestFile {
    std::string path;
    int fd = -1;
    char last = 0;
    bool fastq = false;
    u64 n = 0;
    u8 *d_text = nullptr;
    u64 text_cap = 0;
    u64 *d_start = nullptr, *d_seq = nullptr;
    u32 *d_wrap = nullptr;
    u16 *d_glen = nullptr;
    u64 n_start = 0, n_rec = 0, good = 0;
};
/* the transient buffers of the input stage are carved out of ONE arena — the context's hit buffer, sized here as the probe will want it
 * (64 candidate slots per read: about three times the text): freeing 10 GB right before the pass made its first allocations take
 * 0.6 s on these boxes, and the text simply lives where the candidates will */
struct IngestArena {
    u8 *base = nullptr;
    u64 cap = 0, used = 0;
    void *take(u64 nbytes)
    {
        const u64 bytes = (nbytes + 255) & ~255ull;
        if (!base || used + bytes > cap) return nullptr;
        void *p = base + used;
        used += bytes;
        return p;
    }
};
} // namespace

static void ingest_tables(FxTables *tb)
{
    memset(tb, 0, sizeof *tb);
    auto code = [](char ch) { return ch == 'A' ? 0ull : (ch == 'C' ? 1ull : (ch == 'G' ? 2ull : 3ull)); };
    tb->n_rep = (u32)DISCO_N_END_REPEATS;
    for (u32 r = 0; r < tb->n_rep; r++) {
        u64 v = 0;
        for (int x = 0; x < 29; x++) v = (v << 2) | code(kEndRepeats[r][x]);
        tb->rep58[r] = v;
    }
    tb->n_motif = (u32)DISCO_N_MOTIFS;
    for (u32 m = 0; m < tb->n_motif; m++) {
        const size_t l = strlen(kMotifs[m]);
        tb->motif_len[m] = (u8)l;
        for (size_t x = 0; x < l; x++) {
            tb->motif[m][x] = (u8)kMotifs[m][x];
            tb->need[m][code(kMotifs[m][x])]++;
        }
    }
    static_assert(DISCO_N_END_REPEATS <= FX_MAX_REPEATS && DISCO_N_MOTIFS <= FX_MAX_MOTIFS, "filter tables");
}

/* the file's bytes into d_text: host threads pread slices of a chunk into one half of a pinned ring while the other half travels */
/* flags of the pinned staging ring (DISCO_RING_MODE: 0 the runtime's default, 1 non-coherent = cached on the host side) */
static unsigned ring_alloc_flags()
{
    const int mode = env_int("DISCO_RING_MODE", 0);
    return mode == 1 ? hipHostMallocNonCoherent : (mode == 2 ? hipHostMallocCoherent : hipHostMallocDefault);
}

/* the pinned ring, its events and the copy stream (first use). Measured, round 5: calls into the runtime from this thread — this
 * allocation, every hipMemcpyAsync of the reader — can wait behind the helper thread's hipMalloc of the 28 GB hit buffer; that is the
 * stage's "slow mode" (files into HBM 0.2 s or, in bursts, 0.6-1.8 s with the reader's own waits unchanged). Making the ring first moves
 * the overlap onto the filter kernels instead (0.12 -> 0.18 s): no gain, not kept. */
static int ingest_ring(disco_ctx *c)
{
    const size_t HALF = 128u << 20;
    CHK(ring_back_from_rows(c));
    if (!c->h_ring) {
        if (hipHostMalloc(&c->h_ring, 2 * HALF, ring_alloc_flags()) != hipSuccess) {
            c->h_ring = nullptr;
            (void)hipGetLastError();
            return fail(c, DISCO_E_NOMEM, "disco_ingest_fasta: no pinned staging memory");
        }
        c->ring_half = HALF;
        for (int i = 0; i < 2; i++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_ring[i], hipEventDisableTiming));
    }
    if (!c->copy_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 3; i++) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_unpacked[i], hipEventDisableTiming));
        }
    }
    return DISCO_OK;
}

static int ingest_read_file(disco_ctx *c, int fd, u64 n, u8 *d_text, unsigned threads)
{
    /* The file travels through a pinned ring of RING_SLOTS slots (256 MB in all, as the two halves of rounds 3-4 were): reader threads
     * that live as long as the file — no thread is started per slot — pread piece after piece, in file order, into the next slot that is
     * free; the main thread sends a slot to the device as soon as its pieces are in, two copies in flight, and hands a slot back when its
     * copy is through. With two halves the readers waited for the copy of the half before last before they could start (0.8 of every
     * 3.3 ms: `waiting for the ring` in the DISCO_VERBOSE line of round 4); with four slots reading runs ahead of the link. */
    constexpr unsigned RING_SLOTS = 4;
    const size_t HALF = 128u << 20, SLOT = 2 * HALF / RING_SLOTS;
    CHK(ingest_ring(c));
    /* (the ring's slots need an event each: ev_ring has two, the upload's ev_copied — idle here — the others) */
    hipEvent_t ev[RING_SLOTS] = {c->ev_ring[0], c->ev_ring[1], c->ev_copied[0], c->ev_copied[1]};
    /* SIX readers whatever -t says (round 6; DISCO_INGEST_THREADS: measurement): the link, not the page cache, is what the file waits for, and
     * every reader beyond what keeps the ring full only competes with the DMA engine for the host's memory — 8.1 GB into HBM in 155-160 ms
     * with 5-6 readers, 195-280 ms with 16, 165-300 with 3-4 (16-core host of this pool, four runs each: profiles/r06_experiments.txt H) */
    threads = (unsigned)env_int("DISCO_INGEST_THREADS", (int)std::min(threads, 6u));
    threads = std::max(1u, std::min(threads, 64u));
    CHK(copy_stream_after_stream(c));
    const u64 n_chunks = (n + SLOT - 1) / SLOT;
    const u64 n_pieces = n_chunks * threads;
    std::atomic<u64> next_piece{0};
    std::vector<std::atomic<int>> slot_free(n_chunks), pieces_in(n_chunks); /* chunk k: its slot may be written / pieces of it that are in */
    for (u64 k = 0; k < n_chunks; k++) {
        slot_free[k].store(k < RING_SLOTS ? 1 : 0);
        pieces_in[k].store(0);
    }
    std::atomic<bool> ok{true}, stop{false};
    auto nap = [] { std::this_thread::sleep_for(std::chrono::microseconds(30)); };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&]() {
            for (;;) {
                const u64 p = next_piece.fetch_add(1);
                if (p >= n_pieces) return;
                const u64 k = p / threads;
                const unsigned piece = (unsigned)(p % threads);
                while (!slot_free[k].load(std::memory_order_acquire)) {
                    if (stop.load()) return;
                    nap();
                }
                const u64 off = k * (u64)SLOT;
                const size_t len = (size_t)std::min<u64>(SLOT, n - off);
                char *slot = (char *)c->h_ring + (k % RING_SLOTS) * SLOT;
                size_t p0 = len * piece / threads, p1 = len * (piece + 1) / threads;
                while (p0 < p1 && ok.load()) {
                    const ssize_t got = pread(fd, slot + p0, p1 - p0, (off_t)(off + p0));
                    if (got <= 0) {
                        ok.store(false);
                        break;
                    }
                    p0 += (size_t)got;
                }
                pieces_in[k].fetch_add(1, std::memory_order_release);
            }
        });
    struct Join { /* whatever happens below, the readers are told and waited for */
        std::vector<std::thread> &th;
        std::atomic<bool> &stop;
        ~Join()
        {
            stop.store(true);
            for (auto &x : th)
                if (x.joinable()) x.join();
        }
    } join_readers{th, stop};
    double t_fill = 0, t_copy = 0;
    const auto t_all = HClock::now();
    for (u64 k = 0; k < n_chunks; k++) {
        auto tw = HClock::now();
        while (pieces_in[k].load(std::memory_order_acquire) < (int)threads && ok.load()) nap();
        t_fill += ms_since(tw);
        if (!ok.load()) return fail(c, DISCO_E_ARG, "disco_ingest_fasta: read error");
        const u64 off = k * (u64)SLOT;
        const size_t len = (size_t)std::min<u64>(SLOT, n - off);
        HIPCHK(c, hipMemcpyAsync(d_text + off, (char *)c->h_ring + (k % RING_SLOTS) * SLOT, len, hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipEventRecord(ev[k % RING_SLOTS], c->copy_stream));
        tw = HClock::now();
        if (k >= 1) { /* the copy before this one: through, its slot goes back to the readers (this copy is queued behind it already) */
            HIPCHK(c, hipEventSynchronize(ev[(k - 1) % RING_SLOTS]));
            if (k - 1 + RING_SLOTS < n_chunks) slot_free[k - 1 + RING_SLOTS].store(1, std::memory_order_release);
        }
        t_copy += ms_since(tw);
    }
    auto tw = HClock::now();
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    if (getenv("DISCO_VERBOSE"))
        fprintf(stderr, "[disco] file reader: %.1f MB in %llu slots of %zu MB, %u readers: waited %.1f ms for pieces, %.1f ms for copies, last copy %.1f ms, all %.1f ms\n", n / 1e6,
                (unsigned long long)n_chunks, SLOT >> 20, threads, t_fill, t_copy, ms_since(tw), ms_since(t_all));
    return DISCO_OK;
}

extern "C" int disco_ingest_fasta(disco_ctx *c, const char *const *paths, int n_files, uint32_t host_threads, disco_ingest_info *info, disco_ingest_file *files)
{
    DISCO_TRACE("disco_ingest_fasta");
    if (!c || !paths || n_files < 1 || !info || !files) return c ? fail(c, DISCO_E_ARG, "disco_ingest_fasta: null argument") : DISCO_E_ARG;
    if (c->comm) return fail(c, DISCO_E_UNSUPPORTED, "disco_ingest_fasta: single-GPU contexts only");
    HIPCHK(c, hipSetDevice(c->device));
    const auto t_begin = HClock::now();
    std::vector<IngestFile> F((size_t)n_files);
    struct Owned { /* pieces that did not fit the arena */
        void *p;
        size_t bytes;
    };
    std::vector<Owned> owned;
    IngestArena arena;
    auto cleanup = [&]() {
        for (auto &f : F)
            if (f.fd >= 0) close(f.fd);
        for (auto &o : owned) {
            (void)hipFree(o.p);
            c->hbm_bytes = c->hbm_bytes >= o.bytes ? c->hbm_bytes - o.bytes : 0;
        }
        owned.clear();
    };
    auto unsupported = [&](const char *why, const std::string &path) {
        cleanup();
        return fail(c, DISCO_E_UNSUPPORTED, "disco_ingest_fasta: %s (%s): the host input stage takes this job", why, path.c_str());
    };
    /* ---- the files: regular, not gzip, first byte '>' --------------------------------------------------------------------------- */
    u64 total_bytes = 0;
    for (int fi = 0; fi < n_files; fi++) {
        IngestFile &f = F[(size_t)fi];
        f.path = paths[fi] ? paths[fi] : "";
        if (f.path.size() >= 3 && f.path.compare(f.path.size() - 3, 3, ".gz") == 0) return unsupported("gzip input", f.path);
        f.fd = open(f.path.c_str(), O_RDONLY);
        if (f.fd < 0) return unsupported("unreadable file", f.path);
        struct stat st;
        char first = 0;
        if (fstat(f.fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 1 || pread(f.fd, &first, 1, 0) != 1 || pread(f.fd, &f.last, 1, st.st_size - 1) != 1)
            return unsupported("empty or unreadable file", f.path);
        if (first != '>' && first != '@') return unsupported("neither FASTA nor FASTQ", f.path);
        f.fastq = first == '@';
        f.n = (u64)st.st_size;
        f.text_cap = (f.n + FX_TILE + 63) / FX_TILE * FX_TILE + 64; /* whole tiles (16-byte loads) and aligned 8-byte words behind the end */
        total_bytes += f.text_cap;
    }
    /* ---- the arena. The transient buffers of this stage (text, record arrays) come out of ONE allocation. Where memory is plentiful it
     * is the stage's own (kept by the context: freeing 10 GB right before the pass made the pass's first allocations take 0.6 s), and the
     * hit buffer the probe will want — 64 candidate slots per read, 28 GB at 50 M reads, 0.5-0.9 s of hipMalloc — is allocated by a
     * thread of its own while the files travel to HBM and the filter runs. Where it is not (2 x 10^8 reads on one GPU), the arena IS
     * the hit buffer, as in round 3: the text lives where the candidates will. A previous pass's results are gone either way. */
    if (c->phase > 1) c->phase = 1;
    c->d_adj = nullptr;
    c->adj_total = 0;
    settle_hits_prealloc(c);
    {
        /* hit entries: 64 per read as the probe will ask, a chunk per resident wave; reads estimated at one per 150 bytes of text (shorter
         * records: the probe grows the buffer itself) */
        const u64 want = (total_bytes / 150) * 64 + (u64)c->n_cu * 32 * PR_CHUNK + (1u << 16);
        const u64 own_bytes = total_bytes + (total_bytes / 100) * 28 + (64ull << 20); /* text + record arrays (a record per 100 bytes at worst here; more: pieces of their own) */
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
            cleanup();
            return fail(c, DISCO_E_HIP, "disco_ingest_fasta: hipMemGetInfo failed");
        }
        const u64 rest = (total_bytes / 150) * 200 + (4ull << 30); /* the read table and the index next to it */
        if (c->arena.base && !c->comm && c->arena.used == 0 && !c->d_ingest) { /* the last call's arena, idle again: this call's text goes there */
            c->d_ingest = (u8 *)c->arena.base;
            c->ingest_cap = c->arena.size;
            c->arena = DevArena();
            c->hbm_bytes += c->ingest_cap; /* (counted as one buffer again: the handover below took it out of the sum) */
            c->hbm_peak = std::max(c->hbm_peak, c->hbm_bytes);
        }
        const bool split = !getenv("DISCO_INGEST_SHARED_ARENA") && !c->arena.base && (u64)fr + c->hits_cap * 8 + c->ingest_cap > std::max(want, c->hits_cap) * 8 + own_bytes + rest + (16ull << 30);
        if (split) {
            if (own_bytes > c->ingest_cap) {
                dev_free(c, &c->d_ingest, c->ingest_cap);
                c->ingest_cap = 0;
                if (dev_alloc(c, &c->d_ingest, own_bytes) == DISCO_OK) c->ingest_cap = own_bytes;
                else c->err.clear();
            }
            if (want > c->hits_cap && c->ingest_cap) {
                const int dev = c->device;
                disco_ctx *cc = c;
                c->hits_prealloc = std::thread([cc, dev, want]() {
                    void *p = nullptr;
                    if (hipSetDevice(dev) == hipSuccess && hipMalloc(&p, want * 8) == hipSuccess) {
                        cc->prealloc_ptr = (u64 *)p;
                        cc->prealloc_cap = want;
                    } else
                        (void)hipGetLastError();
                });
            }
        }
        if (split && c->ingest_cap) {
            arena.base = c->d_ingest;
            arena.cap = c->ingest_cap;
        } else {
            if (want > c->hits_cap && (u64)fr + c->hits_cap * 8 > want * 8 + rest) { /* room for it next to the table and the index */
                dev_free(c, &c->d_hits, c->hits_cap);
                c->hits_cap = 0;
                if (dev_alloc(c, &c->d_hits, want) == DISCO_OK) c->hits_cap = want;
                else c->err.clear();
            }
            arena.base = (u8 *)c->d_hits;
            arena.cap = c->hits_cap * 8;
        }
    }
    const float t_arena = ms_since(t_begin) * 1e-3f;
    int rc = DISCO_OK;
    auto get = [&](auto **pp, u64 count) -> int { /* from the arena, or an allocation of its own */
        using T = typename std::remove_pointer<typename std::remove_pointer<decltype(pp)>::type>::type;
        const size_t bytes = std::max<u64>(count, 1) * sizeof(T);
        *pp = (T *)arena.take(bytes);
        if (*pp) return DISCO_OK;
        if (hipMalloc((void **)pp, bytes) != hipSuccess) {
            (void)hipGetLastError();
            return fail(c, DISCO_E_NOMEM, "disco_ingest_fasta: out of device memory");
        }
        c->hbm_bytes += bytes;
        owned.push_back({(void *)*pp, bytes});
        return DISCO_OK;
    };
    FxTables tb;
    ingest_tables(&tb);
    u64 *d_ctr = nullptr;
    if ((rc = get(&d_ctr, (u64)FX_CTR_COUNT)) != DISCO_OK) {
        cleanup();
        return rc;
    }
    u64 total_records = 0, n_good = 0, too_long = 0;
    u32 longest = 0, shortest = 0xFFFFu, short_max = 0;
    u64 n_long_reads = 0;
    float read_s = 0;
    /* ---- pass A: every file into HBM, record starts, clean + filter ------------------------------------------------------------ */
    for (int fi = 0; fi < n_files; fi++) {
        IngestFile &f = F[(size_t)fi];
        const auto t_read = HClock::now();
        if ((rc = get(&f.d_text, f.text_cap)) == DISCO_OK && hipMemsetAsync(f.d_text + f.n, 0, f.text_cap - f.n, c->stream) != hipSuccess) rc = DISCO_E_HIP;
        if (rc == DISCO_OK) rc = ingest_read_file(c, f.fd, f.n, f.d_text, host_threads ? host_threads : 16u);
        read_s += ms_since(t_read) * 1e-3f;
        const u64 tiles = (f.n + FX_TILE - 1) / FX_TILE;
        u64 *d_tile_base = nullptr;
        u32 *d_tile_cnt = nullptr;
        auto body = [&]() -> int {
            CHK(get(&d_tile_base, tiles + 1));
            CHK(get(&d_tile_cnt, tiles));
            HIPCHK(c, hipMemsetAsync(d_ctr, 0, FX_CTR_COUNT * sizeof(u64), c->stream));
            if (f.fastq) {
                u64 n_newlines = 0;
                hipLaunchKernelGGL(fx_lines_kernel, dim3((unsigned)tiles), dim3(256), 0, c->stream, (const u8 *)f.d_text, f.n, d_tile_cnt, (const u64 *)nullptr, (u64 *)nullptr);
                CHK((scan_exclusive<u32, u64>(c, d_tile_cnt, tiles, d_tile_base, false, &n_newlines)));
                const u64 n_lines = n_newlines + (f.last == '\n' ? 0 : 1); /* a last line without a newline is a line */
                f.n_start = (n_lines + 3) / 4; /* the reference starts a record whenever bytes are left (BG/Dataset.cpp:255-293) */
                f.n_rec = f.n_start;
            } else {
                hipLaunchKernelGGL(fx_starts_kernel, dim3((unsigned)tiles), dim3(256), 0, c->stream, (const u8 *)f.d_text, f.n, d_tile_cnt, (const u64 *)nullptr, (u64 *)nullptr, d_ctr);
                CHK((scan_exclusive<u32, u64>(c, d_tile_cnt, tiles, d_tile_base, false, &f.n_start)));
                /* a '>' that is the very last byte starts nothing (the reference's next getline fails) unless it is the only one; it still
                 * ends the sequence of the record before it (disco_amd/host/fastx.cpp) */
                f.n_rec = (f.n_start > 1 && f.last == '>') ? f.n_start - 1 : f.n_start;
            }
            if (f.n_start == 0 || f.n_start >= (1ull << 32)) return DISCO_E_UNSUPPORTED;
            CHK(get(&f.d_start, f.n_start));
            CHK(get(&f.d_seq, f.n_rec));
            CHK(get(&f.d_glen, f.n_rec));
            CHK(get(&f.d_wrap, f.n_rec));
            if (f.fastq) hipLaunchKernelGGL(fx_lines_kernel, dim3((unsigned)tiles), dim3(256), 0, c->stream, (const u8 *)f.d_text, f.n, (u32 *)nullptr, (const u64 *)d_tile_base, f.d_start);
            else hipLaunchKernelGGL(fx_starts_kernel, dim3((unsigned)tiles), dim3(256), 0, c->stream, (const u8 *)f.d_text, f.n, (u32 *)nullptr, (const u64 *)d_tile_base, f.d_start, d_ctr);
            FxFilterArgs fa;
            fa.text = f.d_text;
            fa.n = f.n;
            fa.start = f.d_start;
            fa.n_start = f.n_start;
            fa.n_rec = f.n_rec;
            fa.min_overlap = c->prm.min_overlap;
            fa.fastq = f.fastq ? 1u : 0u;
            fa.glen = f.d_glen;
            fa.seq_begin = f.d_seq;
            fa.wrap = f.d_wrap;
            fa.ctr = d_ctr;
            hipLaunchKernelGGL(fx_filter_kernel, dim3(flat_grid(c, f.n_rec)), dim3(256), 0, c->stream, fa, tb);
            HIPCHK(c, hipGetLastError());
            u64 h[FX_CTR_COUNT];
            HIPCHK(c, hipMemcpyAsync(h, d_ctr, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (h[FX_CTR_BAD_GT] || h[FX_CTR_MULTILINE]) return DISCO_E_UNSUPPORTED;
            f.good = h[FX_CTR_GOOD];
            too_long += h[FX_CTR_TOO_LONG];
            if (f.good) {
                longest = std::max(longest, (u32)h[FX_CTR_MAX_LEN]);
                shortest = std::min(shortest, 0xFFFFu - (u32)h[FX_CTR_MIN_LEN_INV]);
                n_long_reads += h[FX_CTR_N_LONG];
                short_max = std::max(short_max, (u32)h[FX_CTR_SHORT_MAX]);
            }
            return DISCO_OK;
        };
        if (rc == DISCO_OK) rc = body();
        if (rc == DISCO_E_UNSUPPORTED) return unsupported("a '>' inside a line, an irregularly wrapped long record, or no record", f.path);
        if (rc != DISCO_OK) {
            cleanup();
            return rc;
        }
        files[fi].first_index = total_records + 1;
        files[fi].last_index = total_records + f.n_rec;
        files[fi].good = f.good;
        files[fi].bad = f.n_rec - f.good;
        total_records += f.n_rec;
        n_good += f.good;
    }
    if (n_good == 0 || n_good >= (1ull << 31)) return unsupported("no good read (or more than 2^31)", F[0].path);
    /* ---- pass B: ids in file order, rows of the read table -------------------------------------------------------------------- */
    const uint32_t stride_words = std::max<u32>(1, (longest + 31) / 32), dstride = (stride_words + 7u) & ~7u;
    auto pass_b = [&]() -> int {
        u64 *keep_hits = c->d_hits; /* set_reads_common may drop the graph state of a read set of another shape: the arena must survive it */
        const u64 keep_cap = c->hits_cap;
        c->d_hits = nullptr;
        c->hits_cap = 0;
        bool kept = false;
        /* a few long reads among short ones: the rows are packed per class straight from the text (two classes of rows, disco_kernels.h) —
         * the table of one stride, n rows as wide as the longest read, is never made */
        const bool classes = two_class_ok(c, (int)dstride, n_good, n_long_reads, short_max, true);
        const int src = set_reads_common(c, n_good, classes ? (uint32_t)VERIFY_SW : dstride, classes ? nullptr : &kept);
        c->d_hits = keep_hits;
        c->hits_cap = keep_cap;
        CHK(src);
        c->reads_owned = true; /* (as in upload_reads_impl: only now) */
        if (classes) {
            CHK(dev_alloc(c, &c->d_reads, (n_good + n_long_reads) * 8));
            CHK(dev_alloc(c, &c->d_len, n_good));
            CHK(two_class_alloc(c, n_long_reads, (int)dstride, short_max));
        } else if (!kept) {
            CHK(dev_alloc(c, &c->d_reads, n_good * (u64)dstride));
            CHK(dev_alloc(c, &c->d_len, n_good));
        }
        c->reads_owned = true;
        CHK(ensure_cap(c, &c->d_rec_of_read, &c->rec_of_read_cap, n_good));
        c->ingest_id_base.assign((size_t)n_files + 1, 0);
        c->ingest_rec_base.assign((size_t)n_files + 1, 0);
        u64 max_rec = 0;
        for (auto &f : F) max_rec = std::max(max_rec, f.n_rec);
        u64 *d_pos = nullptr;
        u8 *d_flag = nullptr;
        CHK(get(&d_pos, max_rec + 1));
        CHK(get(&d_flag, max_rec));
        u64 id_base = 0, rec_base = 0;
        for (int fi = 0; fi < n_files; fi++) {
            IngestFile &f = F[(size_t)fi];
            c->ingest_id_base[(size_t)fi] = id_base;
            c->ingest_rec_base[(size_t)fi] = rec_base;
            hipLaunchKernelGGL(fx_flags_kernel, dim3(flat_grid(c, f.n_rec)), dim3(256), 0, c->stream, (const u16 *)f.d_glen, f.n_rec, d_flag);
            u64 good = 0;
            CHK((scan_exclusive<u8, u64>(c, d_flag, f.n_rec, d_pos, false, &good)));
            if (good != f.good) return fail(c, DISCO_E_STATE, "disco_ingest_fasta: %llu good reads counted, %llu placed", (unsigned long long)f.good, (unsigned long long)good);
            hipLaunchKernelGGL(fx_ids_kernel, dim3(flat_grid(c, f.n_rec)), dim3(256), 0, c->stream, (const u16 *)f.d_glen, (const u64 *)d_pos, f.n_rec, id_base, c->d_rec_of_read, c->d_len);
            if (f.good)
                hipLaunchKernelGGL(fx_pack_kernel, dim3(flat_grid(c, f.good * (u64)c->S)), dim3(256), 0, c->stream, (const u8 *)f.d_text, (const u64 *)f.d_seq, (const u32 *)f.d_wrap, (const u32 *)c->d_rec_of_read,
                                   (const u16 *)c->d_len, id_base, f.good, c->S, classes ? (u32)DISCO_SHORT_MAX : 0xFFFFu, c->d_reads);
            HIPCHK(c, hipGetLastError());
            id_base += f.good;
            rec_base += f.n_rec;
        }
        c->ingest_id_base[(size_t)n_files] = id_base;
        c->ingest_rec_base[(size_t)n_files] = rec_base;
        if (classes) { /* every read has its id and length: number the long ones, then their full and tail rows, file by file */
            hipLaunchKernelGGL(class_flag_kernel, dim3(flat_grid(c, n_good)), dim3(256), 0, c->stream, (const u16 *)c->d_len, n_good, c->d_ovf, c->d_ctr);
            u64 counted = 0;
            CHK((scan_exclusive<u32, u32>(c, c->d_ovf, n_good, c->d_ovf, false, &counted)));
            if (counted != n_long_reads) return fail(c, DISCO_E_STATE, "disco_ingest_fasta: %llu long reads counted, %llu placed", (unsigned long long)n_long_reads, (unsigned long long)counted);
            hipLaunchKernelGGL(class_ids_kernel, dim3(flat_grid(c, n_good)), dim3(256), 0, c->stream, (const u16 *)c->d_len, (const u32 *)c->d_ovf, n_good, c->d_long_ids);
            for (int fi = 0; fi < n_files; fi++) {
                IngestFile &f = F[(size_t)fi];
                if (!f.good) continue;
                hipLaunchKernelGGL(fx_pack_long_kernel, dim3(flat_grid(c, n_long_reads * ((u64)dstride + 8))), dim3(256), 0, c->stream, (const u8 *)f.d_text, (const u64 *)f.d_seq, (const u32 *)f.d_wrap,
                                   (const u32 *)c->d_rec_of_read, (const u16 *)c->d_len, c->ingest_id_base[(size_t)fi], f.good, (const u32 *)c->d_long_ids, n_long_reads, n_good, (int)dstride, c->tailb,
                                   c->d_full, c->d_reads);
            }
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return DISCO_OK;
    };
    const float t_pass_a = ms_since(t_begin) * 1e-3f;
    rc = pass_b();
    cleanup();
    CHK(rc);
    if (getenv("DISCO_VERBOSE"))
        fprintf(stderr, "[disco] input stage: arena of %.1f GB %.3f s, files + records + filter %.3f s (files %.3f), table + ids + rows %.3f s\n", arena.cap / 1e9, t_arena,
                t_pass_a - t_arena, read_s, ms_since(t_begin) * 1e-3f - t_pass_a);
    /* the stage's own arena is free space from here on: the context's allocator serves the pass from it (its index, headers, adjacency
     * and result buffers: a dozen device allocations of 5-10 ms each that the first pass of a fresh context otherwise waits for) */
    if (c->d_ingest && arena.base == c->d_ingest && owned.empty() && !getenv("DISCO_NO_ARENA_HANDOVER")) {
        c->arena = DevArena();
        c->arena.base = (char *)c->d_ingest;
        c->arena.size = c->ingest_cap & ~(u64)255;
        c->arena.free_at[0] = c->arena.size;
        /* what the allocator serves out of it is counted buffer by buffer (dev_alloc): the arena itself leaves the sum, or every byte of
         * it would count twice in hbm_bytes / hbm_peak */
        c->hbm_bytes = c->hbm_bytes >= c->ingest_cap ? c->hbm_bytes - c->ingest_cap : 0;
        c->d_ingest = nullptr;
        c->ingest_cap = 0;
    }
    c->ingest_n = n_good;
    if (c->two_class) {
        c->max_len_all = longest;
        c->max_len = short_max;
    } else
        c->max_len = longest;
    c->min_len = shortest;
    c->h_len_ok = false;
    c->phase = 1;
    info->n_reads = n_good;
    info->total_records = total_records;
    info->too_long = too_long;
    info->stride_words = stride_words;
    info->shortest = shortest;
    info->longest = longest;
    info->read_s = read_s;
    info->device_s = ms_since(t_begin) * 1e-3f - read_s;
    return DISCO_OK;
}

extern "C" int disco_ingest_fetch(disco_ctx *c, uint16_t *len, uint64_t *file_index)
{
    if (!c || !len || !file_index) return c ? fail(c, DISCO_E_ARG, "disco_ingest_fetch: null argument") : DISCO_E_ARG;
    if (c->phase < 1 || c->ingest_n == 0 || c->ingest_n != c->n) return fail(c, DISCO_E_STATE, "disco_ingest_fetch: the reads of the context did not come from disco_ingest_fasta");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 n = c->n;
    std::unique_ptr<u32[]> rec(new u32[n]);
    /* on the copy stream: the call may run on a host thread of its own while the context's stream is busy with the pass (buildG does
     * that: the per-read arrays are only needed by the writers) */
    hipStream_t st = c->copy_stream ? c->copy_stream : c->stream;
    HIPCHK(c, hipMemcpyAsync(len, c->d_len, n * 2, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(rec.get(), c->d_rec_of_read, n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    const u32 *r = rec.get();
    for (size_t fi = 0; fi + 1 < c->ingest_id_base.size(); fi++) {
        const u64 lo = c->ingest_id_base[fi], hi = c->ingest_id_base[fi + 1], rb = c->ingest_rec_base[fi];
        parallel_for(hi - lo, [&, lo, rb](u64 b, u64 e_) {
            for (u64 i = lo + b; i < lo + e_; i++) file_index[i] = rb + (u64)r[i] + 1; /* BG/Dataset.cpp:294: every record counts */
        });
    }
    {
        std::lock_guard<std::mutex> lk(c->h_len_mu);
        if (!(c->h_len_ok && c->h_len.size() == n)) { /* (ensure_host_len may have mirrored them meanwhile) */
            if (c->h_len.size() != n) c->h_len.resize(n);
            uint16_t *hl = c->h_len.data();
            parallel_for(n, [&, hl](u64 b, u64 e_) { memcpy(hl + b, len + b, (e_ - b) * 2); });
            c->h_len_ok = true;
        }
    }
    return DISCO_OK;
}

int disco_adopt_reads(disco_ctx *c, const void *d_packed, uint32_t stride_words, const void *d_len, uint64_t n)
{
    if (!c || (n && (!d_packed || !d_len))) return c ? fail(c, DISCO_E_ARG, "disco_adopt_reads: null argument") : DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    CHK(set_reads_common(c, n, stride_words));
    c->d_reads = (u64 *)d_packed;
    c->d_len = (u16 *)d_len;
    c->reads_owned = false;
    return validate_reads(c);
}

int disco_generate_reads(disco_ctx *c, const disco_genspec_abi *s)
{
    DISCO_TRACE("disco_generate_reads");
    if (!c || !s) return c ? fail(c, DISCO_E_ARG, "disco_generate_reads: null argument") : DISCO_E_ARG;
    disco_genspec gs;
    memcpy(&gs, s, sizeof gs);
    const uint32_t longest = std::max<uint32_t>(s->len_max, DISCO_GEN_LONG_SHARE(&gs) ? DISCO_GEN_LONG_LEN(&gs) : 0u);
    if (s->len_min == 0 || s->len_max < s->len_min || longest > 32767 || s->n_contigs == 0 || s->contig_len < longest)
        return fail(c, DISCO_E_ARG, "disco_generate_reads: bad spec");
    HIPCHK(c, hipSetDevice(c->device));
    uint32_t stride = (((longest + 31) / 32) + 7u) & ~7u; /* 64-B aligned rows */
    bool kept = false;
    CHK(set_reads_common(c, s->n_reads, stride, &kept));
    if (!kept) {
        CHK(dev_alloc(c, &c->d_reads, c->n * (u64)stride));
        CHK(dev_alloc(c, &c->d_len, c->n));
    }
    c->reads_owned = true;
    disco_genspec g;
    memcpy(&g, s, sizeof g);
    if (c->n) hipLaunchKernelGGL(generate_reads_kernel, dim3(flat_grid(c, c->n * stride)), dim3(256), 0, c->stream, g, c->d_reads, c->d_len, (int)stride, (u64)0, c->n);
    HIPCHK(c, hipGetLastError());
    return validate_reads(c);
}

int disco_substitute_bases(disco_ctx *c, uint64_t seed, uint32_t rate_ppm)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase != 1) return fail(c, DISCO_E_STATE, "disco_substitute_bases: call after the reads are set and before disco_build_index");
    if (rate_ppm > 1000000) return fail(c, DISCO_E_ARG, "disco_substitute_bases: rate above 10^6 ppm");
    if (c->two_class) /* (uploaded / ingested with a tail of long reads: head, tail and full rows would have to change together) */
        return fail(c, DISCO_E_UNSUPPORTED, "disco_substitute_bases: the table has two classes of rows (reads of more than 256 bases next to short ones); substitute "
                                            "into generated reads, or set DISCO_NO_TWO_CLASS=1");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 lo = c->comm ? c->q_lo : 0, hi = c->comm ? c->q_hi : c->n; /* multi-GPU flow: the other ranks' rows arrive by all-gather */
    if (hi > lo && rate_ppm)
        hipLaunchKernelGGL(substitute_bases_kernel, dim3(flat_grid(c, (hi - lo) * (u64)c->S)), dim3(256), 0, c->stream, (u64)seed, rate_ppm, c->d_reads, c->d_len, c->S, lo, hi);
    c->index_counted = c->order_counted = c->order_ready = false; /* what an upload counted ahead was counted on the reads as they were */
    HIPCHK(c, hipGetLastError());
    return DISCO_OK;
}

int disco_download_reads(disco_ctx *c, uint64_t *packed, uint16_t *len)
{
    if (!c || c->phase < 1) return c ? fail(c, DISCO_E_STATE, "no reads") : DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    u64 *joined = nullptr;
    if (packed && c->n && c->two_class) { /* the table as the caller knows it: one stride */
        CHK(dev_alloc(c, &joined, c->n * (u64)c->S_ext));
        hipLaunchKernelGGL(class_join_kernel, dim3(flat_grid(c, c->n * (u64)c->S_ext)), dim3(256), 0, c->stream, (const u64 *)c->d_reads, (const u64 *)c->d_full, c->S_ext,
                           (const u16 *)c->d_len, (const u32 *)c->d_ovf, c->n, joined);
        (void)hipMemcpyAsync(packed, joined, c->n * (u64)c->S_ext * 8, hipMemcpyDeviceToHost, c->stream);
    } else if (packed && c->n)
        HIPCHK(c, hipMemcpyAsync(packed, c->d_reads, c->n * (u64)c->S * 8, hipMemcpyDeviceToHost, c->stream));
    if (len && c->n) (void)hipMemcpyAsync(len, c->d_len, c->n * 2, hipMemcpyDeviceToHost, c->stream);
    const hipError_t e = hipStreamSynchronize(c->stream);
    if (joined) dev_free(c, &joined, c->n * (u64)c->S_ext);
    if (e != hipSuccess) return fail(c, DISCO_E_HIP, "disco_download_reads: %s", hipGetErrorString(e));
    return DISCO_OK;
}

uint32_t disco_stride_words(const disco_ctx *c) { return c ? (uint32_t)(c->two_class ? c->S_ext : c->S) : 0; }
uint64_t disco_num_reads(const disco_ctx *c) { return c ? c->n : 0; }
uint64_t disco_long_rows(const disco_ctx *c) { return c && c->two_class ? c->n_long : 0; }

int disco_set_query_range(disco_ctx *c, uint64_t lo, uint64_t hi)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase < 1 || c->phase > 2) return fail(c, DISCO_E_STATE, "disco_set_query_range: call after the reads are set and before disco_probe");
    if (lo > hi || hi > c->n) return fail(c, DISCO_E_ARG, "query range [%llu,%llu) outside [0,%llu)", (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)c->n);
    c->q_lo = lo;
    c->q_hi = hi;
    return DISCO_OK;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* one stride -> two classes of rows (disco_kernels.h), at the first index build over the table: everything that edits a table (the
 * generator's substitutions) has been and gone by then. The decision is the device's own count of the long reads; the table that is
 * given up goes back to the allocator. */
/* the long class's buffers for a table of n_long long reads of stride Sx whose short class has reads of up to short_max bases; the rows
 * themselves (d_reads: [n + n_long][8]) are the caller's */
static int two_class_alloc(disco_ctx *c, u64 n_long, int Sx, u32 short_max)
{
    /* (the sizes first: free_long_class accounts with them, also for what a failure below leaves behind) */
    c->n_long = n_long;
    c->S_ext = Sx;
    int rc = DISCO_OK;
    if (!c->d_ovf) rc = dev_alloc(c, &c->d_ovf, c->n_alloc);
    if (rc == DISCO_OK) rc = dev_alloc(c, &c->d_full, n_long * (u64)Sx);
    if (rc == DISCO_OK) rc = dev_alloc(c, &c->d_long_ids, n_long);
    if (rc == DISCO_OK) rc = dev_alloc(c, &c->d_lpos, n_long);
    if (rc == DISCO_OK) rc = dev_alloc(c, &c->d_lmeta, n_long);
    if (rc == DISCO_OK) rc = dev_alloc(c, &c->d_linfo, n_long);
    if (rc == DISCO_OK) rc = dev_alloc(c, &c->d_n_list, 1);
    if (rc != DISCO_OK) {
        free_long_class(c);
        return rc;
    }
    c->reads_rows = c->n + n_long;
    c->S = VERIFY_SW;
    c->tailb = short_max <= 160 ? 160 : 256; /* what the staged compare of the short class moves per row (verify_flat_kernel<5 / 8>) */
    c->two_class = true;
    if (getenv("DISCO_VERBOSE"))
        fprintf(stderr, "[disco] two classes of rows: %llu of %llu reads are longer than 256 bases, the others up to %u: 64-byte rows + %d-word rows for those\n",
                (unsigned long long)n_long, (unsigned long long)c->n, short_max, Sx);
    return DISCO_OK;
}

static int two_class_convert(disco_ctx *c)
{
    if (c->two_class || !c->n || c->max_len <= (u32)DISCO_SHORT_MAX || !two_class_ok(c, c->S, c->n, 1, (u32)c->k + 1)) return DISCO_OK; /* (cheap part first) */
    const int Sx = c->S;
    u32 *ovf = nullptr;
    CHK(dev_alloc(c, &ovf, c->n_alloc));
    CHK(zero_counter(c, CTR_SHORT_MAX));
    hipLaunchKernelGGL(class_flag_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_len, c->n, ovf, c->d_ctr);
    u64 n_long = 0;
    CHK((scan_exclusive<u32, u32>(c, ovf, c->n, ovf, false, &n_long)));
    CHK(read_counters(c));
    const u32 short_max = (u32)c->h_ctr[CTR_SHORT_MAX];
    if (!two_class_ok(c, Sx, c->n, n_long, short_max)) {
        dev_free(c, &ovf, c->n_alloc);
        return DISCO_OK;
    }
    u64 *rows8 = nullptr, *old = c->d_reads;
    c->d_ovf = ovf; /* (the context's from here on: released with the long class, also when something below fails) */
    int rca = dev_alloc(c, &rows8, (c->n + n_long) * 8);
    if (rca == DISCO_OK) rca = two_class_alloc(c, n_long, Sx, short_max);
    if (rca != DISCO_OK) { /* the table keeps its one stride */
        dev_free(c, &rows8, (c->n + n_long) * 8);
        free_long_class(c);
        return rca;
    }
    hipLaunchKernelGGL(class_split_kernel, dim3(flat_grid(c, c->n * 8)), dim3(256), 0, c->stream, (const u64 *)old, Sx, (const u16 *)c->d_len, (const u32 *)ovf, c->n, c->tailb, rows8,
                       c->d_full, c->d_long_ids);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream)); /* (the old table goes back: nothing of it may still be in flight) */
    dev_free(c, &old, c->n_alloc * (u64)Sx);
    c->d_reads = rows8;
    c->max_len_all = c->max_len;
    c->max_len = short_max;
    return DISCO_OK;
}

int disco_build_index(disco_ctx *c)
{
    DISCO_TRACE("disco_build_index");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 1) return fail(c, DISCO_E_STATE, "disco_build_index: no reads");
    HIPCHK(c, hipSetDevice(c->device));
    ph_begin(c, DISCO_PH_INDEX);
    if (c->index_counted) { /* disco_upload_reads counted while it copied: the bucket counts, records, runs and keys are in place */
        c->index_counted = false;
    } else {
        CHK(two_class_convert(c));
        CHK(index_begin(c));
        const DiscoView v = view(c);
        IndexCountPlan pl;
        CHK(index_count_plan(c, v, 0, c->n, &pl));
        CHK(index_count_chunk<true>(c, v, pl, c->d_rec, 0, c->n));
        if (c->two_class) { /* the long reads' records, keys and slots, from their full rows */
            const dim3 g((unsigned)((c->n_long + 255) / 256));
            if (c->k > 64) hipLaunchKernelGGL((index_count_kernel<true, true, true>), g, dim3(256), 0, c->stream, v, c->d_bkt, c->d_rec, c->d_okey, (u64)0, c->n_long, pl.ocnt, pl.oslot, pl.oshift);
            else hipLaunchKernelGGL((index_count_kernel<true, false, true>), g, dim3(256), 0, c->stream, v, c->d_bkt, c->d_rec, c->d_okey, (u64)0, c->n_long, pl.ocnt, pl.oslot, pl.oshift);
            HIPCHK(c, hipGetLastError());
        }
    }
    CHK((scan_exclusive<u32, u32>(c, c->d_bkt, c->T + 1, c->d_bkt, false, nullptr)));
    /* the grouping of the reads (the processing order of probe / verify / selection / marking) was counted inside the index pass: it is
     * finished here, so that the fill can walk it (index_fill_ordered_kernel); disco_probe finds it ready */
    c->order_ready = false;
    if (c->order_counted && c->order_counted_lo == 0 && c->order_counted_hi == c->n && !c->two_class && !getenv("DISCO_NO_ORDERED_FILL")) {
        const u64 order_buckets = 1ull << c->order_counted_bits;
        CHK(ensure_cap(c, &c->d_order_own, &c->order_cap, c->n));
        ph_end(c, DISCO_PH_INDEX);
        ph_begin(c, DISCO_PH_ORDER);
        CHK((scan_exclusive<u32, u32>(c, c->d_ocnt, order_buckets + 1, c->d_ocnt, false, nullptr)));
        hipLaunchKernelGGL(order_scatter_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_okey, c->d_oslot, c->d_ocnt, 32u - (u32)c->order_counted_bits, (u64)0, c->n, c->d_len, c->d_order_own);
        ph_end(c, DISCO_PH_ORDER);
        ph_begin(c, DISCO_PH_INDEX2);
        hipLaunchKernelGGL(index_fill_ordered_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->n, (const u64 *)c->d_order_own, (const ulonglong2 *)c->d_rec, (const u32 *)c->d_bkt, c->d_ent);
        ph_end(c, DISCO_PH_INDEX2);
        HIPCHK(c, hipGetLastError());
        c->order_ready = true;
        c->phase = 2;
        return DISCO_OK;
    }
    if (c->n) hipLaunchKernelGGL(index_fill_kernel, dim3(flat_grid(c, 2 * c->n)), dim3(256), 0, c->stream, 2 * c->n, c->d_rec, c->d_bkt, c->d_ent);
    HIPCHK(c, hipGetLastError());
    ph_end(c, DISCO_PH_INDEX);
    c->phase = 2;
    return DISCO_OK;
}

/* ---------------------------------------------------------------------------------------------------------------- */
int disco_probe(disco_ctx *c)
{
    DISCO_TRACE("disco_probe");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 2) return fail(c, DISCO_E_STATE, "disco_probe: build the index first");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(settle_contained_rows(c)); /* rows of the previous pass still travelling read best[] */
    CHK(settle_keys(c));           /* ... and so does a key exchange that runs behind a multi-GPU pass */
    settle_hits_prealloc(c);
    const u64 nq = c->q_hi - c->q_lo;
    if (!c->d_best) {
        CHK(dev_alloc(c, &c->d_best, c->n_alloc));
        CHK(dev_alloc(c, &c->d_row_start, c->n));
        CHK(dev_alloc(c, &c->d_row_cnt, c->n));
    }
    if (c->n_alloc) hipLaunchKernelGGL(fill_u64_kernel, dim3(flat_grid(c, c->n_alloc)), dim3(256), 0, c->stream, c->d_best, c->n_alloc, DISCO_NOKEY);
    HIPCHK(c, hipMemsetAsync(c->d_row_cnt, 0, std::max<u64>(c->n, 1) * sizeof(u32), c->stream));
    const bool ldsrow = c->S <= PROBE_ACAP; /* (two classes of rows: c->S = 8; the lists' kernels take the long class's stride into account: launch_probe) */
    const bool row17 = c->k - view(c).m == 16 && !getenv("DISCO_NO_ROW17"); /* window = 17 m-mers: DPP row-scan variant */
    const bool longk = c->k > 64;                                            /* three-word k-mers: variants of their own (kmer_is_rev) */
    /* the index pass left the minimizer runs of the whole query range: probe_runs_kernel (several reads per wavefront, starts at the
     * bucket lookups); probe_kernel then only does the reads it is handed (unusable run lists, rows that outgrew their chunk) */
    const bool use_runs = c->runs_lpr != 0 && c->d_runs && c->q_lo >= c->runs_lo && c->q_hi <= c->runs_lo + c->runs_n;
    const int grid = use_runs ? (c->runs_lpr == 16 ? wq_grid(c, probe_runs_kernel<16>, nq, "DISCO_PROBE_WAVES") : wq_grid(c, probe_runs_kernel<32>, nq, "DISCO_PROBE_WAVES"))
                     : longk  ? (ldsrow ? wq_grid(c, probe_kernel<0, true, false, true>, nq, "DISCO_PROBE_WAVES") : wq_grid(c, probe_kernel<0, false, false, true>, nq, "DISCO_PROBE_WAVES"))
                     : row17  ? (ldsrow ? wq_grid(c, probe_kernel<0, true, true>, nq, "DISCO_PROBE_WAVES") : wq_grid(c, probe_kernel<0, false, true>, nq, "DISCO_PROBE_WAVES"))
                              : (ldsrow ? wq_grid(c, probe_kernel<0, true, false>, nq, "DISCO_PROBE_WAVES") : wq_grid(c, probe_kernel<0, false, false>, nq, "DISCO_PROBE_WAVES"));
    /* mode 0: the query range, 1: big_list (rows of known size), 2: slow_list */
    auto launch_probe = [&](const ProbeArgs &a, int mode, int g) {
#define DISCO_PROBE_LAUNCH(B, L, R) hipLaunchKernelGGL((probe_kernel<B, L, R>), dim3(g), dim3(64), 0, c->stream, a)
#define DISCO_PROBE_LAUNCH_LONG(B, L) hipLaunchKernelGGL((probe_kernel<B, L, false, true>), dim3(g), dim3(64), 0, c->stream, a)
#define DISCO_PROBE_MODE(B)                                                                      \
    do {                                                                                         \
        if (longk) { if (ldsrow) DISCO_PROBE_LAUNCH_LONG(B, true); else DISCO_PROBE_LAUNCH_LONG(B, false); }     \
        else if (ldsrow) { if (row17) DISCO_PROBE_LAUNCH(B, true, true); else DISCO_PROBE_LAUNCH(B, true, false); }   \
        else { if (row17) DISCO_PROBE_LAUNCH(B, false, true); else DISCO_PROBE_LAUNCH(B, false, false); }        \
    } while (0)
        if (mode != 0 && c->two_class && c->S_ext > PROBE_ACAP) { /* the long class does not fit the LDS row: the lists' reads from global memory */
#define DISCO_PROBE_LAUNCH_CLASS(B)                                                                                                            \
    do {                                                                                                                                        \
        if (longk) hipLaunchKernelGGL((probe_kernel<B, false, false, true, true>), dim3(g), dim3(64), 0, c->stream, a);                        \
        else if (row17) hipLaunchKernelGGL((probe_kernel<B, false, true, false, true>), dim3(g), dim3(64), 0, c->stream, a);                   \
        else hipLaunchKernelGGL((probe_kernel<B, false, false, false, true>), dim3(g), dim3(64), 0, c->stream, a);                             \
    } while (0)
            if (mode == 1) DISCO_PROBE_LAUNCH_CLASS(1);
            else DISCO_PROBE_LAUNCH_CLASS(2);
#undef DISCO_PROBE_LAUNCH_CLASS
        } else if (mode == 0 && use_runs) {
            if (c->runs_lpr == 16) hipLaunchKernelGGL(probe_runs_kernel<16>, dim3(g), dim3(64), 0, c->stream, a, (const u32 *)c->d_runs, c->runs_by_pos ? ~0ull : c->runs_lo);
            else hipLaunchKernelGGL(probe_runs_kernel<32>, dim3(g), dim3(64), 0, c->stream, a, (const u32 *)c->d_runs, c->runs_by_pos ? ~0ull : c->runs_lo);
        } else if (mode == 0) DISCO_PROBE_MODE(0);
        else if (mode == 1) DISCO_PROBE_MODE(1);
        else { /* slow_list only exists next to probe_runs_kernel: 64-byte rows */
            if (longk) DISCO_PROBE_LAUNCH_LONG(2, true);
            else if (row17) DISCO_PROBE_LAUNCH(2, true, true);
            else DISCO_PROBE_LAUNCH(2, true, false);
        }
#undef DISCO_PROBE_LAUNCH_LONG
#undef DISCO_PROBE_MODE
#undef DISCO_PROBE_LAUNCH
    };
    const u64 chunk_slots = use_runs ? PR_CHUNK : PROBE_CHUNK;
    u64 want_hits = nq * 64 + (u64)grid * chunk_slots + (1u << 16);
    u32 want_slow = use_runs ? (u32)std::min<u64>(nq, nq / 64 + 1024 + (c->two_class ? c->n_long : 0)) : 1u;
    u32 want_big = (u32)std::min<u64>(nq, nq / 64 + 1024);
    for (int attempt = 0; attempt < 8; attempt++) {
        if (want_hits > c->hits_cap) {
            dev_free(c, &c->d_hits, c->hits_cap);
            c->hits_cap = 0;
            size_t fr = 0, tot = 0;
            HIPCHK(c, hipMemGetInfo(&fr, &tot));
            if (want_hits * 8 > fr) {
                u64 can = fr / 8 / 10 * 9;
                if (can < (u64)grid * chunk_slots * 2) return fail(c, DISCO_E_NOMEM, "disco_probe: not enough HBM for the hit buffer (%llu entries wanted)", (unsigned long long)want_hits);
                want_hits = can;
            }
            CHK(dev_alloc(c, &c->d_hits, want_hits));
            c->hits_cap = want_hits;
        }
        if (want_slow > c->slow_cap) {
            dev_free(c, &c->d_slow_list, c->slow_cap);
            c->slow_cap = 0;
            CHK(dev_alloc(c, &c->d_slow_list, want_slow));
            c->slow_cap = want_slow;
        }
        if (want_big > c->big_cap) {
            dev_free(c, &c->d_big_list, c->big_cap);
            dev_free(c, &c->d_big_cnt, c->big_cap);
            c->big_cap = 0;
            CHK(dev_alloc(c, &c->d_big_list, want_big));
            CHK(dev_alloc(c, &c->d_big_cnt, want_big));
            c->big_cap = want_big;
        }
        HIPCHK(c, hipMemsetAsync(c->d_bump, 0, sizeof(u64), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_n_big, 0, sizeof(u32), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_n_slow, 0, sizeof(u32), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_ctr + CTR_KMER_HITS, 0, sizeof(u64) * 4, c->stream)); /* KMER_HITS, RAW_HITS, HITS_NEEDED, OVERFLOW */
        CHK(zero_counter(c, CTR_MAX_ROW));
        CHK(zero_counter(c, CTR_ES_BIG));
        CHK(zero_counter(c, CTR_ES_MID));
        c->es_big_counted = c->tr_big_counted = false;
        ProbeArgs a;
        a.v = view(c);
        a.hits = c->d_hits;
        a.row_start = c->d_row_start;
        a.row_cnt = c->d_row_cnt;
        c->h_probe_rare.bump = c->d_bump;
        c->h_probe_rare.hits_cap = c->hits_cap;
        c->h_probe_rare.big_list = c->d_big_list;
        c->h_probe_rare.big_cnt = c->d_big_cnt;
        c->h_probe_rare.n_big = c->d_n_big;
        c->h_probe_rare.big_cap = c->big_cap;
        c->h_probe_rare.slow_cap = c->slow_cap;
        c->h_probe_rare.slow_list = c->d_slow_list;
        c->h_probe_rare.n_slow = c->d_n_slow;
        c->h_probe_rare.ctr = c->d_ctr;
        if (!c->d_probe_rare) CHK(dev_alloc(c, &c->d_probe_rare, 1));
        HIPCHK(c, hipMemcpyAsync(c->d_probe_rare, &c->h_probe_rare, sizeof(ProbeRare), hipMemcpyHostToDevice, c->stream));
        a.rare = c->d_probe_rare;
        CHK(ensure_cap(c, &c->d_meta_ord, &c->meta_cap, std::max<u64>(nq, 1)));
        a.meta_ord = c->d_meta_ord;
        /* grouping of the query range by read-level minimizer for the probe and verify passes (DISCO_NO_ORDER=1: file order) */
        int order_bits = 16;
        const bool own_order = own_order_wanted(c, nq, &order_bits);
        a.order = nullptr;
        const u64 order_buckets = 1ull << order_bits;
        /* the index pass counted already (once: a retry of this loop finds the counters scanned and counts again) */
        const bool counted = c->order_counted && c->order_counted_lo == c->q_lo && c->order_counted_hi == c->q_hi && c->order_counted_bits == order_bits;
        c->order_counted = false;
        if (c->loci) { /* the own list IS the processing order (grouped when the reads were dealt: dist_build_index) */
            c->d_order_used = c->d_order_own;
        } else if (own_order && c->order_ready && counted) { /* disco_build_index finished the grouping (its fill walks the order) */
            c->d_order_used = c->d_order_own; /* (DISCO_PH_ORDER was timed there) */
        } else if (own_order) {
            CHK(ensure_cap(c, &c->d_ocnt, &c->ocnt_cap, order_buckets + 1));
            CHK(ensure_cap(c, &c->d_oslot, &c->oslot_cap, nq));
            CHK(ensure_cap(c, &c->d_order_own, &c->order_cap, nq));
            /* keys -> counts (+ slots) -> starts -> order */
            const u32 oshift = 32u - (u32)order_bits;
            ph_begin(c, DISCO_PH_ORDER);
            if (!counted) {
                HIPCHK(c, hipMemsetAsync(c->d_ocnt, 0, (order_buckets + 1) * sizeof(u32), c->stream));
                hipLaunchKernelGGL(order_count_kernel, dim3(flat_grid(c, nq)), dim3(256), 0, c->stream, c->d_okey + c->q_lo, nq, oshift, c->d_ocnt, c->d_oslot);
            }
            CHK((scan_exclusive<u32, u32>(c, c->d_ocnt, order_buckets + 1, c->d_ocnt, false, nullptr)));
            hipLaunchKernelGGL(order_scatter_kernel, dim3(flat_grid(c, nq)), dim3(256), 0, c->stream, c->d_okey + c->q_lo, c->d_oslot, c->d_ocnt, oshift, c->q_lo, nq, c->d_len, c->d_order_own);
            ph_end(c, DISCO_PH_ORDER);
            HIPCHK(c, hipGetLastError());
            c->d_order_used = c->d_order_own;
        } else if (c->order_external && nq) {
            CHK(ensure_cap(c, &c->d_order_own, &c->order_cap, nq));
            hipLaunchKernelGGL(order_pack_kernel, dim3(flat_grid(c, nq)), dim3(256), 0, c->stream, c->d_order, nq, c->d_len, c->d_order_own);
            HIPCHK(c, hipGetLastError());
            c->d_order_used = c->d_order_own;
        } else
            c->d_order_used = nullptr;
        a.order = c->d_order_used;
        c->order_q_lo = c->q_lo; /* (the range the order in d_order_used belongs to: later phases of the pass walk it too) */
        c->order_q_hi = c->q_hi;
        ph_begin(c, DISCO_PH_PROBE_KERNEL);
        HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
        const bool partitioned = c->dist_active && c->part_index; /* the lookups travel to the buckets' owners (collective) */
        if (partitioned) CHK(dist_partitioned_probe(c));
        else if (nq) {
            launch_probe(a, 0, grid);
        }
        ph_end(c, DISCO_PH_PROBE_KERNEL);
        HIPCHK(c, hipGetLastError());
        u32 n_big = 0, n_slow = 0;
        if (!partitioned) {
            HIPCHK(c, hipMemcpyAsync(&n_slow, c->d_n_slow, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&n_big, c->d_n_big, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        }
        CHK(read_counters(c));
        if (partitioned) c->h_ctr[CTR_HITS_NEEDED] = c->hits_used;
        if (!c->h_ctr[CTR_OVERFLOW] && n_slow) { /* reads without a usable run list, the long way (they may add big rows) */
            int g2 = wave_grid(c, (n_slow + WQ_CHUNK - 1) / WQ_CHUNK, 24);
            HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
            launch_probe(a, 2, g2);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(&n_big, c->d_n_big, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            CHK(read_counters(c));
        }
        if (!partitioned) c->slow_rows = n_slow;
        if (!c->h_ctr[CTR_OVERFLOW] && n_big) {
            int g2 = wave_grid(c, (n_big + WQ_CHUNK - 1) / WQ_CHUNK, 8);
            HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
            launch_probe(a, 1, g2);
            HIPCHK(c, hipGetLastError());
            CHK(read_counters(c));
        }
        if (!c->h_ctr[CTR_OVERFLOW]) {
            c->big_rows = n_big;
            c->hits_used = c->h_ctr[CTR_HITS_NEEDED];
            VerifyArgs va;
            va.v = view(c);
            va.best = c->d_best;
            va.hits = c->d_hits;
            va.row_start = c->d_row_start;
            va.row_cnt = c->d_row_cnt;
            va.order = c->d_order_used;
            va.meta_ord = c->d_meta_ord;
            if (c->wait_bulk_before_verify) { /* multi-GPU flow: the candidate rows of other ranks' reads arrive on bulk_stream */
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_bulk, 0));
                c->wait_bulk_before_verify = false;
            }
            /* two passes (containment-type candidates, flags, overlap-type candidates of non-contained reads) where most reads
             * can be contained: only on request (the kmer_hits counter then counts the compared candidates only), reads of mixed
             * length, rows of the 64-byte staged variants, and not inside a multi-GPU pass (its flags need an exchange in between) */
            /* (multi-GPU pass: every rank takes the same branch — min / max length are the job's, nq plays no part) */
            const bool inexact = c->prm.max_substitutions != 0; /* single pass: the extension is not tuned for metagenomes */
            va.max_subs = c->prm.max_substitutions;
            const bool two_pass = !inexact && !c->two_class && (c->prm.flags & DISCO_FLAG_TWO_PASS_VERIFY) && c->S == VERIFY_SW && (nq || c->dist_active) &&
                                  (u64)c->min_len * 10 < (u64)c->max_len * 9 && !getenv("DISCO_NO_TWO_PASS");
            c->two_pass_last = two_pass;
            /* 64-byte rows, exact overlaps: candidates of a 64-read chunk as one flat list, full wavefronts (verify_flat_kernel) */
            const bool flat = !inexact && c->S == VERIFY_SW && (c->two_class || !getenv("DISCO_NO_FLAT_VERIFY"));
            ph_begin(c, DISCO_PH_VERIFY);
            if (c->two_class && nq) { /* the long reads' candidate rows: out of the flat pass, to verify_long_kernel behind it */
                HIPCHK(c, hipMemsetAsync(c->d_n_list, 0, sizeof(u32), c->stream));
                hipLaunchKernelGGL(class_take_rows_kernel, dim3((unsigned)std::min<u64>((nq + TAKE_SPAN - 1) / TAKE_SPAN, (u64)c->n_cu * 8)), dim3(256), 0, c->stream, c->d_meta_ord,
                                   (const u64 *)c->d_order_used, c->q_lo, nq, (const u32 *)c->d_ovf, c->d_lpos, c->d_lmeta, c->d_linfo, c->d_n_list, (u32)c->n_long);
            }
            if (two_pass) {
                if (!c->d_contained) CHK(dev_alloc(c, &c->d_contained, c->n_alloc));
                if (!c->d_cbits) CHK(dev_alloc(c, &c->d_cbits, c->n_alloc / 64 + 1));
                va.cbits = c->d_cbits;
                const bool short_rows = c->max_len <= 160;
                if (nq && flat) {
                    if (short_rows) hipLaunchKernelGGL((verify_flat_kernel<5, 1>), dim3(wq_grid(c, verify_flat_kernel<5, 1>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                    else hipLaunchKernelGGL((verify_flat_kernel<8, 1>), dim3(wq_grid(c, verify_flat_kernel<8, 1>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                } else if (nq) {
                    if (short_rows) hipLaunchKernelGGL((verify_kernel<5, 1>), dim3(wq_grid(c, verify_kernel<5, 1>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                    else hipLaunchKernelGGL((verify_kernel<8, 1>), dim3(wq_grid(c, verify_kernel<8, 1>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                }
                if (c->dist_active) { /* the keys of other ranks' containing reads count too: exchange, then the flags (collective) */
                    CHK(dist_mark_contained(c));
                    c->contained_done = true;
                } else {
                    CHK(zero_counter(c, CTR_N_CONTAINED));
                    hipLaunchKernelGGL(contain_flags_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_best, c->n, c->d_contained, c->d_cbits, c->d_ctr);
                }
                if (nq && flat) {
                    if (short_rows) hipLaunchKernelGGL((verify_flat_kernel<5, 2>), dim3(wq_grid(c, verify_flat_kernel<5, 2>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                    else hipLaunchKernelGGL((verify_flat_kernel<8, 2>), dim3(wq_grid(c, verify_flat_kernel<8, 2>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                } else if (nq) {
                    if (short_rows) hipLaunchKernelGGL((verify_kernel<5, 2>), dim3(wq_grid(c, verify_kernel<5, 2>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                    else hipLaunchKernelGGL((verify_kernel<8, 2>), dim3(wq_grid(c, verify_kernel<8, 2>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                }
            } else if (nq && inexact) {
                va.cbits = nullptr;
#define VERIFY_INEXACT(NW) hipLaunchKernelGGL((verify_kernel<NW, 0, true>), dim3(wq_grid(c, verify_kernel<NW, 0, true>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va)
                if (c->S == VERIFY_SW && c->max_len <= 160) VERIFY_INEXACT(5);
                else if (c->S == VERIFY_SW) VERIFY_INEXACT(8);
                else if (c->S == 16) VERIFY_INEXACT(16);
                else if (c->S == 24) VERIFY_INEXACT(24);
                else if (c->S == 32) VERIFY_INEXACT(32);
                else VERIFY_INEXACT(0);
#undef VERIFY_INEXACT
            } else if (nq) {
                va.cbits = nullptr;
                /* pairs of batches share their row fetches (verify_flat_kernel<·, 0, true>): 21.5 against 22.6 ms and 18 GB less traffic at config 3
                 * (reads of up to 160 bases); 32.3 against 33.0–35.1 ms on 50 M reads of 100–250 bases, 72.9 against 76.3 with config 5's
                 * abundances (256-base rows, 15 KB of LDS per wavefront). DISCO_VERIFY_CACHE=0 forbids it */
                const char *vce = getenv("DISCO_VERIFY_CACHE");
                const bool vcache = vce ? atoi(vce) != 0 : true;
                /* (round 5: 128 registers and 10 KB of LDS hold sixteen blocks per CU; 12 / 14 / 16 resident: 22.8 / 21.5 / 22.0 ms — the rows of
                 * sixteen blocks' pairs no longer shared the L2 as well. Round 6: with the work queue split by XCD (wq_grab_split: an XCD's waves
                 * work through one contiguous eighth of the processing order) 12 / 14 / 16: 22.7 / 20.8 / 19.8 ms — all sixteen) */
                if (flat && vcache && c->max_len <= 160) hipLaunchKernelGGL((verify_flat_kernel<5, 0, true>), dim3(wq_grid(c, verify_flat_kernel<5, 0, true>, nq, "DISCO_VERIFY_WAVES", 16)), dim3(64), 0, c->stream, va);
                else if (flat && vcache) hipLaunchKernelGGL((verify_flat_kernel<8, 0, true>), dim3(wq_grid(c, verify_flat_kernel<8, 0, true>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (flat && c->max_len <= 160) hipLaunchKernelGGL(verify_flat_kernel<5>, dim3(wq_grid(c, verify_flat_kernel<5>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (flat) hipLaunchKernelGGL(verify_flat_kernel<8>, dim3(wq_grid(c, verify_flat_kernel<8>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (c->S == VERIFY_SW && c->max_len <= 160) hipLaunchKernelGGL(verify_kernel<5>, dim3(wq_grid(c, verify_kernel<5>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (c->S == VERIFY_SW) hipLaunchKernelGGL(verify_kernel<8>, dim3(wq_grid(c, verify_kernel<8>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (c->S == 16) hipLaunchKernelGGL(verify_kernel<16>, dim3(wq_grid(c, verify_kernel<16>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (c->S == 24) hipLaunchKernelGGL(verify_kernel<24>, dim3(wq_grid(c, verify_kernel<24>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else if (c->S == 32) hipLaunchKernelGGL(verify_kernel<32>, dim3(wq_grid(c, verify_kernel<32>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
                else hipLaunchKernelGGL(verify_kernel<0>, dim3(wq_grid(c, verify_kernel<0>, nq, "DISCO_VERIFY_WAVES")), dim3(64), 0, c->stream, va);
            }
            if (c->two_class && nq)
                hipLaunchKernelGGL(verify_long_kernel, dim3(wq_grid(c, verify_long_kernel, c->n_long * 16, "DISCO_VL_WAVES")), dim3(64), 0, c->stream, va, (const u32 *)c->d_lpos, (const ulonglong2 *)c->d_lmeta,
                                   (const uint2 *)c->d_linfo, (const u32 *)c->d_n_list);
            ph_end(c, DISCO_PH_VERIFY);
            HIPCHK(c, hipGetLastError());
            CHK(read_counters(c));
            ph_collect(c);
            c->es_big_counted = true; /* (h_ctr[CTR_ES_BIG]: rows with more than ES_CAP verified hits) */
            c->phase = 3;
            return DISCO_OK;
        }
        if (getenv("DISCO_VERBOSE"))
            fprintf(stderr, "[disco] probe attempt %d overflowed: hits_cap=%llu needed=%llu n_big=%u big_cap=%u overflow=%llu\n", attempt,
                    (unsigned long long)c->hits_cap, (unsigned long long)c->h_ctr[CTR_HITS_NEEDED], n_big, c->big_cap, (unsigned long long)c->h_ctr[CTR_OVERFLOW]);
        /* something was too small: grow and redo the pass (atomicMin on best is idempotent) */
        if (n_big > c->big_cap) want_big = (u32)std::min<u64>(nq, (u64)n_big + n_big / 4 + 1024);
        if (n_slow > c->slow_cap) want_slow = (u32)std::min<u64>(nq, (u64)n_slow + n_slow / 4 + 1024);
        u64 needed = c->h_ctr[CTR_HITS_NEEDED];
        want_hits = std::max<u64>(c->hits_cap * 2, needed + needed / 4 + (u64)grid * chunk_slots);
    }
    return fail(c, DISCO_E_CAPACITY, "disco_probe: hit buffer could not be sized");
}


int disco_mark_contained(disco_ctx *c, uint64_t *n_contained)
{
    DISCO_TRACE("disco_mark_contained");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 3) return fail(c, DISCO_E_STATE, "disco_mark_contained: run disco_probe first");
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->d_contained) CHK(dev_alloc(c, &c->d_contained, c->n_alloc));
    if (!c->d_cbits) CHK(dev_alloc(c, &c->d_cbits, c->n_alloc / 64 + 1));
    CHK(zero_counter(c, CTR_N_CONTAINED));
    ph_begin(c, DISCO_PH_CONTAIN);
    if (c->n) hipLaunchKernelGGL(contain_flags_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_best, c->n, c->d_contained, c->d_cbits, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    ph_end(c, DISCO_PH_CONTAIN);
    CHK(read_counters(c));
    ph_collect(c);
    c->n_contained = c->h_ctr[CTR_N_CONTAINED];
    if (n_contained) *n_contained = c->n_contained;
    c->phase = 4;
    c->crows_pending = c->cgrp_pending = false; /* rows of other flags */
    return DISCO_OK;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* the big-item lists (rows / nodes beyond the LDS capacities) must hold every such item: count them first */
static int ensure_big_cap(disco_ctx *c, const u32 *cnt, const u64 *ref, u32 thr)
{
    CHK(zero_counter(c, CTR_ES_BIG));
    /* (ranks own loci: q_lo / q_hi are positions; the rows and reference words of the other ranks' reads are zero) */
    const u64 ilo = c->loci ? 0 : c->q_lo, ihi = c->loci ? c->n : c->q_hi;
    if (ihi > ilo) hipLaunchKernelGGL(count_above_kernel, dim3(flat_grid(c, ihi - ilo)), dim3(256), 0, c->stream, cnt, ref, ilo, ihi, thr, c->d_ctr + CTR_ES_BIG);
    HIPCHK(c, hipGetLastError());
    CHK(read_counters(c));
    const u64 need = c->h_ctr[CTR_ES_BIG] + 1024;
    if (need > c->big_cap) {
        if (need > 0xFFFFFFFFull) return fail(c, DISCO_E_CAPACITY, "more than 2^32 big rows");
        dev_free(c, &c->d_big_list, c->big_cap);
        dev_free(c, &c->d_big_cnt, c->big_cap);
        c->big_cap = 0;
        CHK(dev_alloc(c, &c->d_big_list, need));
        CHK(dev_alloc(c, &c->d_big_cnt, need));
        c->big_cap = (u32)need;
    }
    return DISCO_OK;
}

static int select_edges(disco_ctx *c)
{
    DISCO_TRACE("select_edges");
    const u64 nq = c->q_hi - c->q_lo;
    if (!c->d_adj_ref) CHK(dev_alloc(c, &c->d_adj_ref, c->n));
    HIPCHK(c, hipMemsetAsync(c->d_adj_ref, 0, std::max<u64>(c->n, 1) * sizeof(u64), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_n_big, 0, sizeof(u32), c->stream));
    CHK(zero_counter(c, CTR_CAP_SITES));
    CHK(zero_counter(c, CTR_DROPPED));
    CHK(zero_counter(c, CTR_ES_SLOW));
    CHK(zero_counter(c, CTR_ADJ_TOTAL));
    CHK(zero_counter(c, CTR_OVERFLOW));
    CHK(zero_counter(c, CTR_TR_BIG));
    CHK(zero_counter(c, CTR_TR_MID));
    /* the kernel rewrites rows in place: it cannot be rerun after an overflow, so its big-row list must hold every such row — counted by
     * verify (no pass of its own, no wait in front of the selection), or here */
    /* the variant of edge_select_flat_kernel with five waves per SIMD where at most one row in a hundred has more than 64 verified hits (they
     * go to the big-row list); DISCO_SELECT_SMALL=0 / 1 forces either */
    const bool rows_counted = c->es_big_counted && !getenv("DISCO_COUNT_BIG_ROWS");
    const bool select_small = rows_counted && env_int("DISCO_SELECT_SMALL", c->h_ctr[CTR_ES_MID] * 100 <= nq ? 1 : 0) != 0;
    if (rows_counted) {
        const u64 need = c->h_ctr[select_small ? CTR_ES_MID : CTR_ES_BIG] + 1024;
        if (need > c->big_cap) {
            if (need > 0xFFFFFFFFull) return fail(c, DISCO_E_CAPACITY, "more than 2^32 big rows");
            dev_free(c, &c->d_big_list, c->big_cap);
            dev_free(c, &c->d_big_cnt, c->big_cap);
            c->big_cap = 0;
            CHK(dev_alloc(c, &c->d_big_list, need));
            CHK(dev_alloc(c, &c->d_big_cnt, need));
            c->big_cap = (u32)need;
        }
    } else
        CHK(ensure_big_cap(c, c->d_row_cnt, nullptr, ES_CAP));
    c->es_big_counted = false;
    EdgeSelArgs a;
    a.v = view(c);
    if (!c->d_dropbits) CHK(dev_alloc(c, &c->d_dropbits, c->n_alloc / 64 + 1));
    HIPCHK(c, hipMemsetAsync(c->d_dropbits, 0, (c->n_alloc / 64 + 1) * sizeof(u64), c->stream));
    c->drop_lo = c->loci ? 0 : c->q_lo;
    c->drop_hi = c->loci ? 0 : c->q_hi; /* (ranks own loci: the bitmap is complete once the lists have been exchanged: dist_complete_twins) */
    a.dropbits = c->d_dropbits;
    a.hidden_flags = c->prm.max_substitutions != 0;
    a.drop_node = a.drop_key = nullptr;
    a.drop_cap = 0;
    CHK(zero_counter(c, CTR_DROP_ITEMS));
    if (c->prm.max_substitutions == 0 && !getenv("DISCO_NO_DROP_LIST")) {
        if (!c->d_drop_node) CHK(dev_alloc(c, &c->d_drop_node, DROP_LIST_CAP));
        if (!c->d_drop_key) CHK(dev_alloc(c, &c->d_drop_key, DROP_LIST_CAP));
        a.drop_node = c->d_drop_node;
        a.drop_key = c->d_drop_key;
        a.drop_cap = (u32)std::min<u64>(DROP_LIST_CAP, (u64)env_int("DISCO_DROP_LIST_CAP", (int)DROP_LIST_CAP)); /* (tests: a list that overflows) */
    }
    a.contained = c->d_cbits;
    a.hits = c->d_hits;
    a.row_start = c->d_row_start;
    a.row_cnt = c->d_row_cnt;
    a.ref = c->d_adj_ref;
    a.max_per_kmer = c->prm.max_edges_per_kmer;
    a.big_list = c->d_big_list;
    a.n_big = c->d_n_big;
    a.big_cap = c->big_cap;
    a.scratch = nullptr;
    a.scratch_cap = 0;
    a.order = c->d_order_used; /* the headers in meta_ord are by position in THIS order */
    a.meta_ord = c->d_meta_ord;
    ph_begin(c, DISCO_PH_SELECT);
    /* reads of up to 256 bases, exact overlaps: the hits of several reads as one flat list, full wavefronts (edge_select_flat_kernel) */
    const bool flat_select = c->S == VERIFY_SW && c->max_len <= 256 && !a.hidden_flags && a.max_per_kmer < 255u && !getenv("DISCO_NO_FLAT_SELECT");
    /* sub-chunks of up to 4 rows / 4 batches: 9.7 KB of LDS, 16 waves per CU. Larger ones fill their last batch better and repeat the
     * per-sub-chunk work less often (8 rows: 140 instead of 175 vector instructions per read) but hold 11 waves per CU, and the kernel's
     * time follows the resident waves (LDS round trips between its phases): 8 x 4: 25.4 ms, 4 x 4: 19.2 ms at 50 M reads */
    /* (round 6: sub-chunks of 4 rows x 3 batches — 16.35 against 17.1 ms with 3 x 2; in round 5 that shape spilled three registers with the
     * sequential path compiled in, the branch-free load pipeline made room; 4 x 4, 5 x 3 and 5 x 4 still spill 12 / 3 / 17) */
#ifndef SEL_SMALL_ROWS
#define SEL_SMALL_ROWS 4
#define SEL_SMALL_NB 3
#endif
    if (nq && flat_select && select_small) hipLaunchKernelGGL((edge_select_flat_kernel<SEL_SMALL_ROWS, SEL_SMALL_NB, true>), dim3(wq_grid(c, edge_select_flat_kernel<SEL_SMALL_ROWS, SEL_SMALL_NB, true>, nq, "DISCO_SELECT_WAVES")), dim3(64), 0, c->stream, a);
    else if (nq && flat_select) hipLaunchKernelGGL((edge_select_flat_kernel<4, 4>), dim3(wq_grid(c, edge_select_flat_kernel<4, 4>, nq, "DISCO_SELECT_WAVES")), dim3(64), 0, c->stream, a);
    else if (nq) hipLaunchKernelGGL(edge_select_kernel<false>, dim3(wq_grid(c, edge_select_kernel<false>, nq, "DISCO_SELECT_WAVES")), dim3(64), 0, c->stream, a);
    ph_end(c, DISCO_PH_SELECT);
    HIPCHK(c, hipGetLastError());
    u32 n_big = 0;
    HIPCHK(c, hipMemcpyAsync(&n_big, c->d_n_big, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    CHK(read_counters(c));
    ph_collect(c);
    if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_CAPACITY, "edge selection: big-row list overflow (%u rows)", n_big);
    if (n_big) { /* rows of up to ES_MID hits: LDS arrays of their own (the five-wave variant lists rows of 65 .. ES_CAP hits too: a pass with small arrays first) */
        if (select_small) {
            HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
            hipLaunchKernelGGL(edge_select_mid_kernel<ES_CAP>, dim3((int)std::min<u64>(n_big, (u64)c->n_cu * 32)), dim3(64), 0, c->stream, a, 0u);
        }
        if (!select_small || c->h_ctr[CTR_ES_BIG]) {
            HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
            hipLaunchKernelGGL(edge_select_mid_kernel<ES_MID>, dim3((int)std::min<u64>(n_big, (u64)c->n_cu * 8)), dim3(64), 0, c->stream, a, select_small ? (u32)ES_CAP : 0u);
        }
        HIPCHK(c, hipGetLastError());
        CHK(read_counters(c));
    }
    if (n_big && c->h_ctr[CTR_MAX_ROW] > ES_MID) {
        int g2 = (int)std::min<u64>(n_big, (u64)c->n_cu * 8); /* work queue, one big row per grab */
        u64 cap = c->h_ctr[CTR_MAX_ROW] + 64;
        u64 *scratch = nullptr;
        CHK(dev_alloc(c, &scratch, (u64)g2 * 2 * cap));
        a.scratch = scratch;
        a.scratch_cap = cap;
        HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
        hipLaunchKernelGGL(edge_select_kernel<true>, dim3(g2), dim3(64), 0, c->stream, a);
        hipError_t e = hipGetLastError();
        int rc = read_counters(c);
        dev_free(c, &scratch, (u64)g2 * 2 * cap);
        if (e != hipSuccess) return fail(c, DISCO_E_HIP, "edge_select_kernel<true>: %s", hipGetErrorString(e));
        CHK(rc);
    }
    c->dropped = c->dropped_local = c->h_ctr[CTR_DROPPED];
    c->n_drop_items = a.drop_node ? c->h_ctr[CTR_DROP_ITEMS] : ~0ull;
    c->tr_big_counted = true; /* (h_ctr[CTR_TR_BIG]: nodes with more than TR_CAP finds) */
    if (c->contained_count_pending) { /* disco_run_graph: the flags' count came back with these counters */
        c->n_contained = c->h_ctr[CTR_N_CONTAINED];
        c->contained_count_pending = false;
    }
    if (getenv("DISCO_VERBOSE"))
        fprintf(stderr, "[disco] edge selection%s: %llu rows in the sequential path, %u in the big-row list (%llu rows of more than 64 hits), dropped %llu\n",
                select_small ? " (five waves per SIMD)" : "", (unsigned long long)c->h_ctr[CTR_ES_SLOW], n_big, (unsigned long long)c->h_ctr[CTR_ES_MID], (unsigned long long)c->dropped);
    /* the finds stay where they are: the hit buffer IS the adjacency array of the local rows */
    c->d_adj = c->d_hits;
    c->adj_total = c->h_ctr[CTR_ADJ_TOTAL];
    c->adj_imported = false;
    c->adj_span = 0;
    c->half_complete = true; /* one rank: the local lists are all the lists */
    c->ph_ms[DISCO_PH_CSR] = 0;
    c->phase = 5;
    return DISCO_OK;
}

/* the few-extras merge moves only the rows that grow; otherwise every row is rebuilt (merge_extras) */
static bool merge_is_sparse(const disco_ctx *c) { return c->d_adj == c->d_hits && c->n_extra <= 16384 && !getenv("DISCO_MERGE_REBUILD"); }

static int twin_check_search(disco_ctx *c, u64 lo, u64 hi);

/* twin check over targets [lo,hi); collects extras, does not merge */
static int twin_check(disco_ctx *c, u64 lo, u64 hi)
{
    CHK(twin_check_search(c, lo, hi));
    /* inexact overlaps: edge selection left "hidden from the other read" flags in the entries (ADJ_HIDDEN_OF); rows that the merge
     * does not rebuild are cleaned here */
    if (c->prm.max_substitutions != 0 && c->n && (c->n_extra == 0 || merge_is_sparse(c))) {
        hipLaunchKernelGGL(clear_adj_flags_kernel, dim3(flat_grid(c, c->n * 16)), dim3(256), 0, c->stream, c->d_adj_ref, c->d_adj, c->n);
        HIPCHK(c, hipGetLastError());
    }
    return DISCO_OK;
}

static int twin_check_search(disco_ctx *c, u64 lo, u64 hi)
{
    DISCO_TRACE("twin_check");
    if (!c->d_extra_cnt) CHK(dev_alloc(c, &c->d_extra_cnt, c->n));
    /* After the contained filter the verified-hit relation is symmetric: a hit A->B at window j >= 1 of overlap length
     * ovl >= k+1 shows up from B's side at window ovl-k >= 1 against A's other end record, and verifies over the same
     * region. A find can therefore miss its twin only if the twin's owner DROPPED a verified hit (second hit to the same
     * destination, BG/OverlapGraph.cpp:656, or the per-k-mer cap, :645). If no read of the target range dropped anything,
     * every list in the range already holds all twins and nothing needs to be searched. */
    /* (inexact mode: a substitution inside one read's end k-mer hides the pair from the other side — always search) */
    if (c->dropped == 0 && c->prm.max_substitutions == 0 && lo >= c->q_lo && hi <= c->q_hi && !getenv("DISCO_FORCE_TWIN_CHECK")) {
        c->n_extra = 0;
        c->asym_local = 0;
        c->ph_ms[DISCO_PH_TWIN] = 0;
        return DISCO_OK;
    }
    TwinArgs a;
    a.v = view(c);
    a.ref = c->d_adj_ref;
    a.adj = c->d_adj;
    a.lo = lo;
    a.hi = hi;
    a.otab = c->loci ? c->d_otab : nullptr; /* (ranks own loci: of the nodes [lo, hi) only the own ones) */
    a.me = c->comm ? (u32)c->comm->rank : 0u;
    a.extra_cnt = c->d_extra_cnt;
    a.n_extra = c->d_n_extra;
    /* Something was dropped — but only the lists of the reads that dropped something can lack a twin (same argument, per read):
     * with their bitmap at hand the search is limited to finds INTO those reads, a few thousand instead of every entry of
     * every list (real data always drop something at their repeats: 95 -> 5 ms at 50 M reads with 0.3 % errors) */
    /* One GPU, exact overlaps: the dropped hits themselves are on a list (every one of them, or the list is not used). A twin is missing
     * from w's list only where w dropped exactly that hit, so each item is looked up once in the OTHER read's list — no pass over the
     * entries at all (16 -> 0.1 ms at 50 M reads with 0.1 % errors) */
    const bool by_list = c->prm.max_substitutions == 0 && c->d_drop_node && lo == 0 && hi == c->n && c->q_lo == 0 && c->q_hi == c->n && !c->adj_imported &&
                         c->n_drop_items == c->dropped && c->n_drop_items <= (u64)env_int("DISCO_DROP_LIST_CAP", (int)DROP_LIST_CAP) && !getenv("DISCO_FORCE_TWIN_CHECK");
    const bool by_bitmap = !by_list && c->d_dropbits && lo >= c->drop_lo && hi <= c->drop_hi && !getenv("DISCO_FORCE_TWIN_CHECK");
    a.dropbits = by_bitmap ? c->d_dropbits : nullptr;
    /* inexact overlaps: the same, plus every find verify_kernel flagged as hidden from the other read (a substitution inside this
     * read's end k-mer there) — 0.13 -> 0.04 s of search at 50 M reads with 0.3 % errors */
    a.hidden_flags = by_bitmap && c->prm.max_substitutions != 0;
    /* (inexact overlaps: one-sided pairs are the rule — a substitution inside an end k-mer hides the pair from the other read —
     * so the proof of symmetry is not attempted) */
    if (!by_bitmap && !by_list && c->prm.max_substitutions == 0) {   /* one-sided pass: half the searches, no extras. Symmetric iff nothing is missing and #up == #down. */
        CHK(zero_counter(c, CTR_ASYM));
        CHK(zero_counter(c, CTR_TW_UP));
        CHK(zero_counter(c, CTR_TW_DOWN));
        a.extra_node = nullptr;
        a.extra_key = nullptr;
        a.extra_cap = 0;
        a.up_only = 1;
        ph_begin(c, DISCO_PH_TWIN);
        if (c->n) hipLaunchKernelGGL(twin_check_kernel, dim3(flat_grid(c, c->n * 64)), dim3(256), 0, c->stream, a);
        ph_end(c, DISCO_PH_TWIN);
        HIPCHK(c, hipGetLastError());
        CHK(read_counters(c));
        ph_collect(c);
        if (getenv("DISCO_VERBOSE"))
            fprintf(stderr, "[disco] twin check [%llu,%llu): one-sided pass: missing %llu, up %llu, down %llu (dropped %llu)\n", (unsigned long long)lo,
                    (unsigned long long)hi, (unsigned long long)c->h_ctr[CTR_ASYM], (unsigned long long)c->h_ctr[CTR_TW_UP],
                    (unsigned long long)c->h_ctr[CTR_TW_DOWN], (unsigned long long)c->dropped);
        if (c->h_ctr[CTR_ASYM] == 0 && c->h_ctr[CTR_TW_UP] == c->h_ctr[CTR_TW_DOWN]) {
            c->n_extra = 0;
            c->asym_local = 0;
            return DISCO_OK;
        }
    }
    u32 want = std::max<u32>(4096, c->extra_cap); /* the lists kept from the last pass are the best guess */
    if (by_list) want = std::max<u32>(want, (u32)c->n_drop_items); /* (an item yields at most one extra) */
    for (int attempt = 0; attempt < 6; attempt++) {
        if (want > c->extra_cap) {
            dev_free(c, &c->d_extra_node, c->extra_cap);
            dev_free(c, &c->d_extra_key, c->extra_cap);
            CHK(dev_alloc(c, &c->d_extra_node, want));
            CHK(dev_alloc(c, &c->d_extra_key, want));
            c->extra_cap = want;
        }
        HIPCHK(c, hipMemsetAsync(c->d_extra_cnt, 0, std::max<u64>(c->n, 1) * sizeof(u32), c->stream));
        HIPCHK(c, hipMemsetAsync(c->d_n_extra, 0, sizeof(u32), c->stream));
        CHK(zero_counter(c, CTR_ASYM));
        CHK(zero_counter(c, CTR_OVERFLOW));
        a.extra_node = c->d_extra_node;
        a.extra_key = c->d_extra_key;
        a.extra_cap = c->extra_cap;
        a.up_only = 0;
        ph_begin(c, DISCO_PH_TWIN);
        if (by_list) {
            if (c->n_drop_items)
                hipLaunchKernelGGL(twin_from_drops_kernel, dim3(flat_grid(c, c->n_drop_items)), dim3(256), 0, c->stream, a, (const u64 *)c->d_drop_node,
                                   (const u64 *)c->d_drop_key, (u32)c->n_drop_items);
        } else if (c->n)
            hipLaunchKernelGGL(twin_check_kernel, dim3(flat_grid(c, c->n * 64)), dim3(256), 0, c->stream, a);
        ph_end(c, DISCO_PH_TWIN);
        HIPCHK(c, hipGetLastError());
        u32 ne = 0;
        HIPCHK(c, hipMemcpyAsync(&ne, c->d_n_extra, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
        CHK(read_counters(c));
        ph_collect(c);
        if (!c->h_ctr[CTR_OVERFLOW]) {
            c->n_extra = ne;
            c->asym_local = c->h_ctr[CTR_ASYM];
            /* a partial range: the one-sided pass was unbalanced, i.e. a down-find FROM this range has no twin — but that twin is
             * missing from the list of a node of ANOTHER range, which this pass does not complete. Report "not symmetric" all the
             * same: the sharded caller sums these values over the ranks only to decide whether every rank must run the full pass
             * (found by tools/fuzz_sharded.py: one cap-bound site on rank 0, both ranks answered 0) */
            if (c->asym_local == 0 && !(lo == 0 && hi == c->n)) c->asym_local = 1;
            return DISCO_OK;
        }
        want = ne + ne / 4 + 4096;
    }
    return fail(c, DISCO_E_CAPACITY, "twin check: extras list could not be sized");
}

/* merge the collected extras into a node-ordered CSR (rare: only when pairs were found from one side only) */
static int merge_extras(disco_ctx *c)
{
    DISCO_TRACE("merge_extras");
    if (c->n_extra == 0) return DISCO_OK;
    c->tr_big_counted = false; /* rows grow */
    /* a handful of extras and the rows still where edge selection left them (one GPU): move only the rows that grow into the free
     * tail of the hit buffer — 84 ms of the 272 ms pass at 50 M reads with 0.3 % errors went into rebuilding all of it for 953 extras */
    if (merge_is_sparse(c)) {
        HIPCHK(c, hipMemsetAsync(c->d_bump, 0, sizeof(u64), c->stream));
        hipLaunchKernelGGL(merge_need_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_adj_ref, c->d_extra_cnt, c->n, c->d_bump);
        u64 need = 0;
        HIPCHK(c, hipMemcpyAsync(&need, c->d_bump, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (c->hits_used + need <= c->hits_cap) {
            HIPCHK(c, hipMemsetAsync(c->d_bump, 0, sizeof(u64), c->stream));
            hipLaunchKernelGGL(merge_sparse_kernel, dim3((unsigned)std::min<u64>(c->n_extra, (u64)c->n_cu * 16)), dim3(64), 0, c->stream, c->d_extra_node, c->d_extra_key,
                               c->n_extra, c->d_extra_cnt, c->d_adj_ref, c->d_adj, c->hits_used, c->d_bump);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(c->stream));
            c->hits_used += need;
            c->adj_total += c->n_extra;
            c->n_extra = 0;
            return DISCO_OK;
        }
    }
    u32 *new_deg = nullptr;
    u64 *new_start = nullptr, *new_adj = nullptr, *scratch = nullptr;
    u64 total = 0, scratch_n = 0, new_cap = 0;
    auto body = [&]() -> int {
        CHK(dev_alloc(c, &new_deg, c->n));
        CHK(dev_alloc(c, &new_start, c->n + 1));
        hipLaunchKernelGGL(merge_deg_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_adj_ref, c->d_extra_cnt, c->n, new_deg);
        CHK((scan_exclusive<u32, u64>(c, new_deg, c->n, new_start, true, &total)));
        if (c->adj_spare_cap >= total && c->d_adj_spare) { /* the buffer the previous pass's merge left behind */
            new_adj = c->d_adj_spare;
            new_cap = c->adj_spare_cap;
            c->d_adj_spare = nullptr;
            c->adj_spare_cap = 0;
        } else {
            CHK(dev_alloc(c, &new_adj, total + total / 16));
            new_cap = total + total / 16;
        }
        hipLaunchKernelGGL(merge_scatter_kernel, dim3(flat_grid(c, c->n_extra)), dim3(256), 0, c->stream, c->d_extra_node, c->d_extra_key, c->n_extra, c->d_adj_ref, new_start, new_adj);
        /* row scratch: the longest merged row (reduced on the device) */
        CHK(zero_counter(c, CTR_MAX_DEG));
        hipLaunchKernelGGL(max_u32_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, new_deg, c->n, c->d_ctr + CTR_MAX_DEG);
        HIPCHK(c, hipGetLastError());
        CHK(read_counters(c));
        const u64 maxdeg = c->h_ctr[CTR_MAX_DEG];
        const int g = (int)std::min<u64>(c->n, (u64)c->n_cu * 32); /* one wavefront per row at a time */
        scratch_n = (u64)g * (maxdeg + 1);
        CHK(dev_alloc(c, &scratch, scratch_n));
        hipLaunchKernelGGL(merge_rows_kernel, dim3(g), dim3(64), 0, c->stream, c->d_adj_ref, c->d_adj, c->d_extra_cnt, new_start, new_adj, c->n, scratch, maxdeg + 1);
        hipLaunchKernelGGL(ref_from_start_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, new_start, new_deg, c->n, c->d_adj_ref);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return DISCO_OK;
    };
    const int rc = body();
    dev_free(c, &scratch, scratch_n);
    dev_free(c, &new_deg, c->n);
    dev_free(c, &new_start, c->n + 1);
    if (rc != DISCO_OK) {
        dev_free(c, &new_adj, new_cap);
        return rc;
    }
    /* the rows move to new_adj; what held the imported / merged rows so far waits for the next pass's merge */
    dev_free(c, &c->d_adj_spare, c->adj_spare_cap);
    c->d_adj_spare = c->d_adj_own;
    c->adj_spare_cap = c->d_adj_own ? c->adj_cap : 0;
    c->d_adj_own = new_adj;
    c->d_adj = new_adj;
    c->adj_cap = new_cap;
    c->adj_total = total;
    c->adj_span = total; /* compact, node ordered */
    c->n_extra = 0;
    return DISCO_OK;
}

int disco_select_edges(disco_ctx *c)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase != 4) return fail(c, DISCO_E_STATE, "disco_select_edges: run disco_mark_contained first");
    HIPCHK(c, hipSetDevice(c->device));
    return select_edges(c);
}

/* full != 0: check (and complete) the lists of ALL nodes; else only those of the query range */
int disco_symmetrize(disco_ctx *c, int full, uint64_t *n_asym)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase < 5) return fail(c, DISCO_E_STATE, "disco_symmetrize: select edges first");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(twin_check(c, full ? 0 : c->q_lo, full ? c->n : c->q_hi));
    if (n_asym) *n_asym = c->asym_local;
    c->phase = 6;
    return DISCO_OK;
}

int disco_merge_extras(disco_ctx *c)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase < 6) return fail(c, DISCO_E_STATE, "disco_merge_extras: symmetrize first");
    HIPCHK(c, hipSetDevice(c->device));
    return merge_extras(c);
}

int disco_build_edges(disco_ctx *c, uint64_t *n_pre)
{
    if (!c) return DISCO_E_ARG;
    CHK(disco_select_edges(c));
    CHK(disco_symmetrize(c, 1, nullptr));
    CHK(merge_extras(c));
    if (n_pre) *n_pre = c->adj_total / 2;
    return DISCO_OK;
}

int disco_adjacency_size(disco_ctx *c, uint64_t *n_entries)
{
    if (!c || !n_entries) return DISCO_E_ARG;
    if (c->phase < 5) return fail(c, DISCO_E_STATE, "disco_adjacency_size: select edges first");
    *n_entries = c->adj_total;
    return DISCO_OK;
}

int disco_export_adjacency(disco_ctx *c, void *d_deg_u32, void *d_entries_u64)
{
    if (!c || !d_deg_u32) return DISCO_E_ARG;
    if (c->phase < 5 || c->adj_imported) return fail(c, DISCO_E_STATE, "disco_export_adjacency: needs the locally selected edges");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 nq = c->q_hi - c->q_lo;
    ph_begin(c, DISCO_PH_CSR);
    if (nq) hipLaunchKernelGGL(deg_from_ref_kernel, dim3(flat_grid(c, nq)), dim3(256), 0, c->stream, c->d_adj_ref, c->q_lo, c->q_hi, (u32 *)d_deg_u32);
    HIPCHK(c, hipGetLastError());
    if (nq && c->adj_total && d_entries_u64) { /* compact the local rows into node order */
        CHK(ensure_cap(c, &c->d_start_tmp, &c->start_cap, c->n + 1));
        u64 *start = c->d_start_tmp;
        u64 total = 0;
        int rc = scan_exclusive<u32, u64>(c, (const u32 *)d_deg_u32, nq, start, true, &total);
        if (rc == DISCO_OK && total != c->adj_total) rc = fail(c, DISCO_E_STATE, "disco_export_adjacency: degree sum %llu != %llu", (unsigned long long)total, (unsigned long long)c->adj_total);
        if (rc == DISCO_OK) hipLaunchKernelGGL(rows_gather_kernel, dim3(wave_grid(c, nq, 16)), dim3(64), 0, c->stream, c->d_adj, c->d_adj_ref, c->q_lo, c->q_hi, start, (u64 *)d_entries_u64,
                                              (c->phase == 5 && c->prm.max_substitutions != 0) ? 0ull : (u64)ADJ_FLAG); /* (hidden flags travel to the twin search) */
        hipError_t e = hipStreamSynchronize(c->stream);
        CHK(rc);
        if (e != hipSuccess) return fail(c, DISCO_E_HIP, "disco_export_adjacency: %s", hipGetErrorString(e));
    }
    ph_end(c, DISCO_PH_CSR);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ph_collect(c);
    return DISCO_OK;
}





int disco_import_adjacency(disco_ctx *c, const void *d_deg_u32_all, const void *d_entries_u64_all, uint64_t n_entries_all)
{
    if (!c || !d_deg_u32_all) return DISCO_E_ARG;
    if (c->phase < 5) return fail(c, DISCO_E_STATE, "disco_import_adjacency: select edges first");
    HIPCHK(c, hipSetDevice(c->device));
    CHK(ensure_cap(c, &c->d_start_tmp, &c->start_cap, c->n + 1));
    u64 *start = c->d_start_tmp;
    u64 total = 0;
    CHK((scan_exclusive<u32, u64>(c, (const u32 *)d_deg_u32_all, c->n, start, true, &total)));
    if (total != n_entries_all)
        return fail(c, DISCO_E_ARG, "disco_import_adjacency: degrees sum to %llu but %llu entries were passed", (unsigned long long)total, (unsigned long long)n_entries_all);
    CHK(ensure_cap(c, &c->d_adj_own, &c->adj_cap, total));
    c->d_adj = c->d_adj_own;
    c->adj_total = total;
    if (total) HIPCHK(c, hipMemcpyAsync(c->d_adj, d_entries_u64_all, total * 8, hipMemcpyDeviceToDevice, c->stream));
    if (c->n) hipLaunchKernelGGL(ref_from_start_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, start, (const u32 *)d_deg_u32_all, c->n, c->d_adj_ref);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->adj_imported = true;
    c->tr_big_counted = false;
    c->adj_span = total;
    c->half_complete = false;
    c->phase = 5;
    return DISCO_OK;
}




/* ---------------------------------------------------------------------------------------------------------------- */
int disco_transitive_mark(disco_ctx *c)
{
    DISCO_TRACE("disco_transitive_mark");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 6) return fail(c, DISCO_E_STATE, "disco_transitive_mark: build edges first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 nq = c->q_hi - c->q_lo;
    HIPCHK(c, hipMemsetAsync(c->d_n_big, 0, sizeof(u32), c->stream));
    CHK(zero_counter(c, CTR_OVERFLOW));
    /* the variant with LDS arrays for TR_CAP_SMALL neighbours (eight waves per SIMD) where at most one node in a hundred is beyond them
     * (they go to the big-node pass); DISCO_TR_SMALL=0 / 1 forces either */
    const bool counted_nodes = c->tr_big_counted && c->q_lo == 0 && c->q_hi == c->n && !getenv("DISCO_COUNT_BIG_ROWS");
    const bool tr_small = env_int("DISCO_TR_SMALL", counted_nodes && c->h_ctr[CTR_TR_MID] * 100 <= nq ? 1 : 0) != 0 && counted_nodes;
    if (counted_nodes) { /* edge selection counted the nodes beyond TR_CAP (and beyond TR_CAP_SMALL) */
        const u64 need = c->h_ctr[tr_small ? CTR_TR_MID : CTR_TR_BIG] + 1024;
        if (need > c->big_cap) {
            if (need > 0xFFFFFFFFull) return fail(c, DISCO_E_CAPACITY, "more than 2^32 big nodes");
            dev_free(c, &c->d_big_list, c->big_cap);
            dev_free(c, &c->d_big_cnt, c->big_cap);
            c->big_cap = 0;
            CHK(dev_alloc(c, &c->d_big_list, need));
            CHK(dev_alloc(c, &c->d_big_cnt, need));
            c->big_cap = (u32)need;
        }
    } else
        CHK(ensure_big_cap(c, nullptr, c->d_adj_ref, TR_CAP));
    TrArgs a;
    a.v = view(c);
    a.ref = c->d_adj_ref;
    a.adj = c->d_adj;
    a.big_list = c->d_big_list;
    a.n_big = c->d_n_big;
    a.big_cap = c->big_cap;
    a.scratch = nullptr;
    a.hcap = 0;
    a.half = nullptr;
    a.hcnt = nullptr;
    c->use_half = !getenv("DISCO_NO_HALF");
    if (c->use_half) {
        if (!c->d_half) CHK(dev_alloc(c, &c->d_half, c->n * HALF_CAP));
        if (!c->d_hcnt) CHK(dev_alloc(c, &c->d_hcnt, c->n));
        HIPCHK(c, hipMemsetAsync(c->d_hcnt, 0, std::max<u64>(c->n, 1) * sizeof(u32), c->stream));
        if (!c->d_wide) {
            c->wide_cap = (u32)std::min<u64>(c->n, c->n / 32 + 4096);
            CHK(dev_alloc(c, &c->d_wide, c->wide_cap));
            CHK(dev_alloc(c, &c->d_n_wide, 1));
        }
        HIPCHK(c, hipMemsetAsync(c->d_n_wide, 0, sizeof(u32), c->stream));
        a.half = c->d_half;
        a.hcnt = c->d_hcnt;
    }
    a.wide_list = c->d_wide;
    a.n_wide = c->d_n_wide;
    a.wide_cap = c->wide_cap;
    a.order = (c->d_order_used && c->order_q_lo == c->q_lo && c->order_q_hi == c->q_hi && !c->adj_imported && !getenv("DISCO_TR_NO_ORDER")) ? c->d_order_used : nullptr;
    /* every row needs its flags when the emission cannot rely on the survivor lists alone */
    a.all_flags = (c->adj_imported || !c->use_half || c->q_lo != 0 || c->q_hi != c->n) ? 1u : 0u;
    ph_begin(c, DISCO_PH_TRMARK);
    const bool lists = a.half && a.hcnt && !a.all_flags; /* the survivor lists are the result: the variant compiled for it */
#if defined(TR_EXP_DEFER_SINGLE) /* timing experiment: the multi-rank variant of the kernel on one GPU's nodes */
    if (nq && tr_small) hipLaunchKernelGGL((transitive_mark_kernel<false, true, TR_CAP_SMALL>), dim3(wq_grid(c, transitive_mark_kernel<false, true, TR_CAP_SMALL>, nq, "DISCO_TR_WAVES")), dim3(64), 0, c->stream, a);
#else
    if (nq && tr_small && lists) hipLaunchKernelGGL((transitive_mark_kernel<false, false, TR_CAP_SMALL, true>), dim3(wq_grid(c, transitive_mark_kernel<false, false, TR_CAP_SMALL, true>, nq, "DISCO_TR_WAVES")), dim3(64), 0, c->stream, a);
    else if (nq && tr_small) hipLaunchKernelGGL((transitive_mark_kernel<false, false, TR_CAP_SMALL>), dim3(wq_grid(c, transitive_mark_kernel<false, false, TR_CAP_SMALL>, nq, "DISCO_TR_WAVES")), dim3(64), 0, c->stream, a);
#endif
    else if (nq) hipLaunchKernelGGL((transitive_mark_kernel<false, false>), dim3(wq_grid(c, transitive_mark_kernel<false, false>, nq, "DISCO_TR_WAVES")), dim3(64), 0, c->stream, a);
    ph_end(c, DISCO_PH_TRMARK);
    HIPCHK(c, hipGetLastError());
    u32 n_big = 0;
    HIPCHK(c, hipMemcpyAsync(&n_big, c->d_n_big, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    c->n_wide = 0;
    if (c->use_half) HIPCHK(c, hipMemcpyAsync(&c->n_wide, c->d_n_wide, sizeof(u32), hipMemcpyDeviceToHost, c->stream)); /* (read again below if the big nodes add to it) */
    CHK(read_counters(c));
    ph_collect(c);
    if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_CAPACITY, "transitive marking: big-node list overflow (%u nodes)", n_big);
    if (n_big) {
        /* longest list among the big nodes bounds the hash size: one reduction on the device, one read-back */
        CHK(zero_counter(c, CTR_MAX_DEG));
        hipLaunchKernelGGL(list_max_degree_kernel, dim3(flat_grid(c, n_big)), dim3(256), 0, c->stream, c->d_big_list, (u64)n_big, c->d_adj_ref, c->d_ctr + CTR_MAX_DEG);
        HIPCHK(c, hipGetLastError());
        CHK(read_counters(c));
        const u64 maxd = c->h_ctr[CTR_MAX_DEG];
        u64 hcap = 64;
        while (hcap < 2 * maxd) hcap <<= 1;
        int g2 = (int)std::min<u64>(n_big, (u64)c->n_cu * 8);
        u64 per = hcap * 8 + hcap * 4 + hcap;
        u8 *scratch = nullptr;
        CHK(dev_alloc(c, &scratch, (u64)g2 * per));
        a.scratch = (u64 *)scratch;
        a.hcap = hcap;
        HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
        hipLaunchKernelGGL((transitive_mark_kernel<true, false>), dim3(g2), dim3(64), 0, c->stream, a);
        hipError_t e = hipGetLastError();
        const int rc2 = read_counters(c); /* synchronises; the big pass may have raised CTR_OVERFLOW */
        dev_free(c, &scratch, (u64)g2 * per);
        if (e != hipSuccess) return fail(c, DISCO_E_HIP, "transitive_mark_kernel (big nodes): %s", hipGetErrorString(e));
        CHK(rc2);
        if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_CAPACITY, "transitive marking (big nodes): list overflow");
    }
    if (n_big && c->use_half) HIPCHK(c, hipMemcpy(&c->n_wide, c->d_n_wide, sizeof(u32), hipMemcpyDeviceToHost));
    c->flags_pending = false;
    c->phase = 7;
    return DISCO_OK;
}


int disco_emit_edges(disco_ctx *c, uint64_t *n_out)
{
    DISCO_TRACE("disco_emit_edges");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 7) return fail(c, DISCO_E_STATE, "disco_emit_edges: run disco_transitive_mark first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 nq = c->q_hi - c->q_lo;
    ph_begin(c, DISCO_PH_EMIT);
    const int grid = wq_grid(c, emit_kernel, nq, "DISCO_EMIT_WAVES");
    const u64 n_push = c->dist_active ? c->n_push_r : 0; /* multi-GPU flow: survivors pushed by the owners of the larger endpoints */
    const int grid_push = (int)std::max<u64>(std::min<u64>((n_push + 63) / 64, (u64)c->n_cu * 16), 1);
    const u64 nwaves = (u64)grid * 2 + (n_push ? (u64)grid_push : 0); /* emit_half_kernel + emit_kernel (+ emit_push_recv_kernel) */
    u64 want = std::max<u64>(2 * nq, 1024) + n_push + nwaves * EMIT_CHUNK;
    for (int attempt = 0; attempt < 6; attempt++) {
        if (!c->d_out_src || want > c->out_cap) {
            dev_free(c, &c->d_out_src, c->out_cap);
            dev_free(c, &c->d_out_ent, c->out_cap);
            c->out_cap = 0;
            CHK(dev_alloc(c, &c->d_out_src, want));
            CHK(dev_alloc(c, &c->d_out_ent, want));
            c->out_cap = want;
        }
        HIPCHK(c, hipMemsetAsync(c->d_bump, 0, sizeof(u64), c->stream));
        const bool half_emit = c->use_half && c->half_complete; /* else: rows + flags of every node (sharded fallback) */
        if (half_emit && nq) {
            EmitHalfArgs h;
            h.v = view(c);
            h.ref = c->d_adj_ref;
            h.adj = c->d_adj;
            h.half = c->d_half;
            h.hcnt = c->d_hcnt;
            h.out_src = c->d_out_src;
            h.out_ent = c->d_out_ent;
            h.out_cap = c->out_cap;
            h.bump = c->d_bump;
            h.local_only = c->dist_active ? 1u : 0u;
            h.own = own_set(c);
            h.order = nullptr; /* (tried for one GPU, round 5: the nodes in processing order — 5.3 instead of 3.4 ms: the node's own entries become the random fetches) */
            const int gh = wq_grid(c, emit_half_kernel, (nq + 63) / 64, "DISCO_EMIT_WAVES");
            hipLaunchKernelGGL(emit_half_kernel, dim3(gh), dim3(64), 0, c->stream, h);
        }
        EmitArgs a;
        a.v = view(c);
        a.ref = c->d_adj_ref;
        a.adj = c->d_adj;
        a.hcnt = half_emit ? c->d_hcnt : nullptr;
        a.half = half_emit ? c->d_half : nullptr;
        const bool listed = half_emit && c->n_wide <= c->wide_cap; /* else the list overflowed: scan the whole range */
        a.list = listed ? c->d_wide : nullptr;
        a.n_list = listed ? c->n_wide : 0;
        a.out_src = c->d_out_src;
        a.out_ent = c->d_out_ent;
        a.out_cap = c->out_cap;
        a.bump = c->d_bump;
        a.local_only = c->dist_active ? 1u : 0u;
        a.own = own_set(c);
        HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
        if (nq && !(listed && c->n_wide == 0)) hipLaunchKernelGGL(emit_kernel, dim3(grid), dim3(64), 0, c->stream, a);
        if (n_push) {
            EmitRecvArgs pr;
            pr.items = c->d_push_r;
            pr.n_items = n_push;
            pr.ref = c->d_adj_ref;
            pr.adj = c->d_adj;
            pr.half = c->d_half;
            pr.hcnt = c->d_hcnt;
            pr.out_src = c->d_out_src;
            pr.out_ent = c->d_out_ent;
            pr.out_cap = c->out_cap;
            pr.bump = c->d_bump;
            pr.wq = c->d_wq;
            HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
            hipLaunchKernelGGL(emit_push_recv_kernel, dim3(grid_push), dim3(64), 0, c->stream, pr);
        }
        HIPCHK(c, hipGetLastError());
        u64 used = 0;
        HIPCHK(c, hipMemcpyAsync(&used, c->d_bump, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (used <= c->out_cap) {
            c->out_used = used;
            break;
        }
        want = used + used / 8 + nwaves * EMIT_CHUNK;
        if (attempt == 5) return fail(c, DISCO_E_CAPACITY, "disco_emit_edges: output buffer could not be sized");
    }
    /* survivors = chunk slots that are not the ~0 tail marker: count them (the compaction happens at fetch time) */
    u64 total = 0;
    if (c->out_used) {
        if (!c->d_out_valid || c->out_used > c->valid_cap) {
            dev_free(c, &c->d_out_valid, c->valid_cap);
            dev_free(c, &c->d_out_pos, c->valid_cap + 1);
            CHK(dev_alloc(c, &c->d_out_valid, c->out_cap));
            CHK(dev_alloc(c, &c->d_out_pos, c->out_cap + 1));
            c->valid_cap = c->out_cap;
        }
        hipLaunchKernelGGL(emit_valid_kernel, dim3(flat_grid(c, c->out_used)), dim3(256), 0, c->stream, c->d_out_src, c->out_used, c->d_out_valid);
        CHK((scan_exclusive<u8, u64>(c, c->d_out_valid, c->out_used, c->d_out_pos, true, nullptr)));
        HIPCHK(c, hipMemcpyAsync(&total, c->d_total, sizeof(u64), hipMemcpyDeviceToHost, c->stream)); /* (with the phase's one wait) */
    }
    ph_end(c, DISCO_PH_EMIT);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ph_collect(c);
    c->n_out = total;
    if (n_out) *n_out = total;
    c->phase = 8;
    return DISCO_OK;
}

int disco_transitive_reduce(disco_ctx *c, uint64_t *n_out)
{
    CHK(disco_transitive_mark(c));
    return disco_emit_edges(c, n_out);
}

int disco_run_graph(disco_ctx *c)
{
    DISCO_TRACE("disco_run_graph");
    const bool verbose = getenv("DISCO_VERBOSE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();
    CHK(disco_build_index(c));
    auto t1 = now();
    CHK(disco_probe(c));
    auto t2 = now();
    { /* disco_mark_contained without its wait: the count of the flags comes back with edge selection's counters */
        if (!c->d_contained) CHK(dev_alloc(c, &c->d_contained, c->n_alloc));
        if (!c->d_cbits) CHK(dev_alloc(c, &c->d_cbits, c->n_alloc / 64 + 1));
        CHK(zero_counter(c, CTR_N_CONTAINED));
        ph_begin(c, DISCO_PH_CONTAIN);
        if (c->n) hipLaunchKernelGGL(contain_flags_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_best, c->n, c->d_contained, c->d_cbits, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        ph_end(c, DISCO_PH_CONTAIN);
        c->contained_count_pending = true;
        c->phase = 4;
        c->crows_pending = c->cgrp_pending = false;
    }
    auto t3 = now();
    CHK(disco_build_edges(c, nullptr));
    auto t4 = now();
    int rc = disco_transitive_reduce(c, nullptr);
    auto t5 = now();
    if (verbose)
        fprintf(stderr, "[disco] host wall ms: index %.1f probe %.1f contain %.1f edges %.1f reduce %.1f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5));
    return rc;
}

/* ---------------------------------------------------------------------------------------------------------------- */
static int ensure_host_len(disco_ctx *c)
{
    std::lock_guard<std::mutex> lk(c->h_len_mu);
    if (c->h_len_ok && c->h_len.size() == c->n) return DISCO_OK;
    c->h_len.resize(c->n);
    if (c->n) HIPCHK(c, hipMemcpy(c->h_len.data(), c->d_len, c->n * 2, hipMemcpyDeviceToHost));
    c->h_len_ok = true;
    return DISCO_OK;
}

int64_t disco_fetch_contained(disco_ctx *c, disco_contained_row *out, uint64_t cap)
{
    DISCO_TRACE("disco_fetch_contained");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 4) return fail(c, DISCO_E_STATE, "disco_fetch_contained: run disco_mark_contained first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 nc = c->n_contained;
    if (!out) return (int64_t)nc;
    if (cap < nc) return fail(c, DISCO_E_ARG, "disco_fetch_contained: need room for %llu rows", (unsigned long long)nc);
    if (nc == 0) return 0;
    /* lens_h: len2 | len1 << 16 per row where the device sent them along; null: gathered from the host's copy of the length table */
    auto decode = [&](auto id_of, const u64 *keys_h, const u32 *lens_h) {
        const u16 *hlen = lens_h ? nullptr : c->h_len.data();
        const u32 kk = (u32)c->k;
        parallel_for(nc, [&, hlen, kk, lens_h](u64 b, u64 e_) {
            for (u64 i = b; i < e_; i++) {
                const u64 key = keys_h[i];
                disco_contained_row &r = out[i];
                r.contained = id_of(i);
                r.super = CKEY_SUPER(key);
                r.j = CKEY_J(key);
                r.type = disco_hit_type(CKEY_SUFFIX(key), CKEY_REV(key));
                r.len2 = lens_h ? (lens_h[i] & 0xFFFFu) : hlen[r.contained];
                r.len1 = lens_h ? (lens_h[i] >> 16) : hlen[r.super];
                u32 orient, off;
                disco_map_type(r.type, r.len1, kk, r.j, &orient, &off); /* BG/OverlapGraph.cpp:428-434 */
                r.orient = orient;
                r.start = off;
            }
        });
    };
    CHK(start_contained_rows(c, false));
    if (c->crows_pending && c->crows_n == nc) {
        HIPCHK(c, hipEventSynchronize(c->ev_crows));
        const u64 *hkey = (const u64 *)c->h_crows;
        const u32 *hid = (const u32 *)(hkey + c->crows_hcap);
        decode([hid](u64 i) { return (u64)hid[i]; }, hkey, hid + c->crows_hcap);
        return (int64_t)nc;
    }
    CHK(ensure_host_len(c));
    CHK(settle_keys(c));
    u64 *pos = nullptr, *ids = nullptr, *keys = nullptr;
    std::vector<u64> hid(nc), hkey(nc);
    auto gather = [&]() -> int {
        CHK(dev_alloc(c, &pos, c->n + 1));
        CHK(dev_alloc(c, &ids, nc));
        CHK(dev_alloc(c, &keys, nc));
        CHK((scan_exclusive<u8, u64>(c, c->d_contained, c->n, pos, false, nullptr)));
        hipLaunchKernelGGL(contain_rows_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_best, c->d_contained, pos, c->n, ids, keys);
        HIPCHK(c, hipMemcpyAsync(hid.data(), ids, nc * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(hkey.data(), keys, nc * 8, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return DISCO_OK;
    };
    const int grc = gather();
    dev_free(c, &pos, c->n + 1);
    dev_free(c, &ids, nc);
    dev_free(c, &keys, nc);
    CHK(grc);
    for (u64 i = 0; i < nc; i++) /* (a key that names no read must not index the length table: fail loudly) */
        if (CKEY_SUPER(hkey[i]) >= c->n || hid[i] >= c->n)
            return fail(c, DISCO_E_STATE, "disco_fetch_contained: row %llu of %llu: read %llu has the key %llx (no containing read)", (unsigned long long)i, (unsigned long long)nc,
                        (unsigned long long)hid[i], (unsigned long long)hkey[i]);
    decode([&hid](u64 i) { return hid[i]; }, hkey.data(), nullptr);
    return (int64_t)nc;
}

int64_t disco_fetch_contained_grouped(disco_ctx *c, disco_contained_row *out, uint64_t cap)
{
    DISCO_TRACE("disco_fetch_contained_grouped");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 4) return fail(c, DISCO_E_STATE, "disco_fetch_contained_grouped: run disco_mark_contained first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 nc = c->n_contained;
    if (!out) return (int64_t)nc;
    if (cap < nc) return fail(c, DISCO_E_ARG, "disco_fetch_contained_grouped: need room for %llu rows", (unsigned long long)nc);
    if (nc == 0) return 0;
    if (getenv("DISCO_NO_GROUPED_ROWS")) return fail(c, DISCO_E_UNSUPPORTED, "disco_fetch_contained_grouped: switched off (DISCO_NO_GROUPED_ROWS)");
    CHK(start_contained_rows(c, true));
    if (!(c->cgrp_pending && c->crows_pending && c->crows_n == nc))
        return fail(c, DISCO_E_UNSUPPORTED, "disco_fetch_contained_grouped: the rows were not grouped on the device (sort what disco_fetch_contained returns)");
    HIPCHK(c, hipStreamSynchronize(c->aux_stream));
    if (c->h_cgrp_big) return fail(c, DISCO_E_UNSUPPORTED, "disco_fetch_contained_grouped: a containing read has more than %d rows (sort what disco_fetch_contained returns)", CROW_GROUP_MAX);
    const u64 *hkey = (const u64 *)((const char *)c->h_crows + c->crows_hcap * 16);
    const u32 *hid = (const u32 *)(hkey + c->crows_hcap), *hln = hid + c->crows_hcap;
    const u32 kk = (u32)c->k;
    parallel_for(nc, [&, kk, hkey, hid, hln](u64 b, u64 e_) {
        for (u64 i = b; i < e_; i++) {
            const u64 key = hkey[i];
            disco_contained_row &r = out[i];
            r.contained = hid[i];
            r.super = CKEY_SUPER(key);
            r.j = CKEY_J(key);
            r.type = disco_hit_type(CKEY_SUFFIX(key), CKEY_REV(key));
            r.len2 = hln[i] & 0xFFFFu;
            r.len1 = hln[i] >> 16;
            u32 orient, off;
            disco_map_type(r.type, r.len1, kk, r.j, &orient, &off); /* BG/OverlapGraph.cpp:428-434 */
            r.orient = orient;
            r.start = off;
        }
    });
    return (int64_t)nc;
}

int64_t disco_fetch_edges(disco_ctx *c, disco_edge *out, uint64_t cap)
{
    DISCO_TRACE("disco_fetch_edges");
    if (!c) return DISCO_E_ARG;
    if (c->phase < 8) return fail(c, DISCO_E_STATE, "disco_fetch_edges: run disco_transitive_reduce first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 ne = c->n_out;
    if (!out) return (int64_t)ne;
    if (cap < ne) return fail(c, DISCO_E_ARG, "disco_fetch_edges: need room for %llu edges", (unsigned long long)ne);
    if (ne == 0) return 0;
    CHK(ensure_host_len(c));
    /* drop the unused chunk tails of the emission (into buffers the context keeps: two allocations of 0.4 GB per call cost more than the
     * copy), then copy out 12 bytes per edge — source as 32 bits, packed entry — chunk k into pinned staging memory (kept by the context)
     * while the host threads turn chunk k - 1 into disco_edge records; a copy into pageable memory ran at a third of the link's rate */
    if (ne > c->fetch_cap) {
        dev_free(c, &c->d_fetch_src, c->fetch_cap);
        dev_free(c, &c->d_fetch_ent, c->fetch_cap);
        c->fetch_cap = 0;
        const u64 want = ne + ne / 8 + 1024;
        CHK(dev_alloc(c, &c->d_fetch_src, want));
        CHK(dev_alloc(c, &c->d_fetch_ent, want));
        c->fetch_cap = want;
    }
    u32 *csrc = c->d_fetch_src;
    u64 *cent = c->d_fetch_ent;
    hipLaunchKernelGGL(emit_compact32_kernel, dim3(flat_grid(c, c->out_used)), dim3(256), 0, c->stream, c->d_out_src, c->d_out_ent, c->d_out_valid, c->d_out_pos, c->out_used, csrc, cent);
    const u64 CHUNK = 1ull << 22; /* 4 M edges: 48 MB per half */
    if (!c->h_stage) {
        if (hipHostMalloc((void **)&c->h_stage, 4 * CHUNK * sizeof(u64)) != hipSuccess) c->h_stage = nullptr;
        for (int i = 0; i < 2 && c->h_stage; i++)
            if (hipEventCreateWithFlags(&c->ev_stage[i], hipEventDisableTiming) != hipSuccess) {
                (void)hipHostFree(c->h_stage);
                c->h_stage = nullptr;
            }
    }
    const u16 *hlen = c->h_len.data();
    auto convert = [&](const u32 *hsp, const u64 *hep, u64 base, u64 cnt) {
        parallel_for(cnt, [&, hlen, hsp, hep, base](u64 b, u64 e_) {
            for (u64 i = b; i < e_; i++) {
                disco_edge &e = out[base + i];
                e.src = hsp[i];
                e.dst = ADJ_DST(hep[i]);
                e.orient = ADJ_ORI(hep[i]);
                e.offset = ADJ_OFF(hep[i]);
                e.len_src = hlen[e.src];
                e.len_dst = ADJ_DLEN(hep[i]);
            }
        });
    };
    bool ok = true;
    if (c->h_stage) {
        const u64 nch = (ne + CHUNK - 1) / CHUNK;
        auto half_ent = [&](u64 k) { return c->h_stage + (k & 1) * 2 * CHUNK; };          /* entries: CHUNK words   */
        auto half_src = [&](u64 k) { return (u32 *)(c->h_stage + (k & 1) * 2 * CHUNK + CHUNK); }; /* sources: CHUNK / 2 words */
        auto issue = [&](u64 k) {
            const u64 cnt = std::min(CHUNK, ne - k * CHUNK);
            ok = ok && hipMemcpyAsync(half_src(k), csrc + k * CHUNK, cnt * 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipMemcpyAsync(half_ent(k), cent + k * CHUNK, cnt * 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
                 hipEventRecord(c->ev_stage[k & 1], c->stream) == hipSuccess;
        };
        issue(0);
        for (u64 k = 0; k < nch && ok; k++) {
            ok = hipEventSynchronize(c->ev_stage[k & 1]) == hipSuccess;
            if (k + 1 < nch) issue(k + 1); /* into the other half, converted one iteration ago */
            if (ok) convert(half_src(k), half_ent(k), k * CHUNK, std::min(CHUNK, ne - k * CHUNK));
        }
        ok = (hipStreamSynchronize(c->stream) == hipSuccess) && ok;
    } else { /* no pinned memory to be had: one pageable copy */
        std::unique_ptr<u32[]> hs(new u32[ne]);
        std::unique_ptr<u64[]> he(new u64[ne]); /* not zero-filled: 0.5 GB at 45 M edges */
        ok = hipMemcpyAsync(hs.get(), csrc, ne * 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
             hipMemcpyAsync(he.get(), cent, ne * 8, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
        if (ok) convert(hs.get(), he.get(), 0, ne);
    }
    if (!ok) return fail(c, DISCO_E_HIP, "disco_fetch_edges: copy failed");
    return (int64_t)ne;
}

int64_t disco_fetch_edge_substitutions(disco_ctx *c, uint16_t *out, uint64_t cap)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase < 8) return fail(c, DISCO_E_STATE, "disco_fetch_edge_substitutions: run disco_transitive_reduce first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 ne = c->n_out;
    if (!out) return (int64_t)ne;
    if (cap < ne) return fail(c, DISCO_E_ARG, "disco_fetch_edge_substitutions: need room for %llu edges", (unsigned long long)ne);
    if (ne == 0) return 0;
    if (c->prm.max_substitutions == 0) { /* exact overlaps: nothing to count */
        memset(out, 0, ne * sizeof(uint16_t));
        return (int64_t)ne;
    }
    u16 *subs = nullptr;
    CHK(dev_alloc(c, &subs, ne));
    hipLaunchKernelGGL(edge_subs_kernel, dim3(flat_grid(c, c->out_used)), dim3(256), 0, c->stream, c->d_out_src, c->d_out_ent, c->d_out_valid, c->d_out_pos,
                       c->out_used, c->d_reads, c->d_len, c->S, subs);
    hipError_t e1 = hipMemcpyAsync(out, subs, ne * sizeof(u16), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    dev_free(c, &subs, ne);
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(c, DISCO_E_HIP, "disco_fetch_edge_substitutions: copy failed");
    return (int64_t)ne;
}

/* connected components of the edges in the slots [0, n_slots) (valid[i] != 0; pos[i] = rank of the edge among the valid ones)
 * dealt out to n_files files: out[pos[i]] = file of edge i */
static int64_t partition_edges(disco_ctx *c, const u64 *d_src, const u64 *d_ent, const u8 *d_valid, const u64 *d_pos, u64 n_slots, u64 ne, u64 n,
                               uint32_t n_files, uint16_t *out)
{
    DISCO_TRACE("partition_edges");
    u32 *parent = nullptr, *cnt = nullptr, *d_nlist = nullptr;
    u16 *cfile = nullptr, *efile = nullptr;
    u64 *list = nullptr;
    const u32 list_cap = n_files * 64 + 64;
    int rc = DISCO_OK;
    auto cleanup = [&]() {
        dev_free(c, &parent, n);
        dev_free(c, &cnt, n);
        dev_free(c, &cfile, n);
        dev_free(c, &efile, ne);
        dev_free(c, &list, list_cap);
        dev_free(c, &d_nlist, 1);
    };
#define PART_CHK(x)            \
    do {                       \
        rc = (x);              \
        if (rc != DISCO_OK) {  \
            cleanup();         \
            return rc;         \
        }                      \
    } while (0)
    PART_CHK(dev_alloc(c, &parent, n));
    PART_CHK(dev_alloc(c, &cnt, n));
    PART_CHK(dev_alloc(c, &cfile, n));
    PART_CHK(dev_alloc(c, &efile, ne));
    PART_CHK(dev_alloc(c, &list, list_cap));
    PART_CHK(dev_alloc(c, &d_nlist, 1));
    hipLaunchKernelGGL(uf_init_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, parent, n);
    hipLaunchKernelGGL(uf_hook_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, d_src, d_ent, d_valid, n_slots, parent);
    hipLaunchKernelGGL(uf_compress_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, parent, n);
    (void)hipMemsetAsync(cnt, 0, n * sizeof(u32), c->stream);
    (void)hipMemsetAsync(cfile, 0xFF, n * sizeof(u16), c->stream);
    (void)hipMemsetAsync(d_nlist, 0, sizeof(u32), c->stream);
    hipLaunchKernelGGL(uf_count_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, d_src, d_valid, n_slots, parent, cnt);
    /* components of at least 1/(64 files) of the edges are dealt out by size, largest first to the lightest file; the rest
     * (there can be millions of small ones) go by hash */
    const u32 thr = (u32)std::max<u64>(ne / ((u64)n_files * 64), 1);
    hipLaunchKernelGGL(uf_big_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, cnt, n, thr, list, d_nlist, list_cap);
    u32 n_list = 0;
    std::vector<u64> hl(list_cap);
    if (hipMemcpyAsync(&n_list, d_nlist, sizeof(u32), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipMemcpyAsync(hl.data(), list, list_cap * sizeof(u64), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) {
        cleanup();
        return fail(c, DISCO_E_HIP, "edge partition: %s", hipGetErrorString(hipGetLastError()));
    }
    n_list = std::min(n_list, list_cap);
    hl.resize(n_list);
    std::sort(hl.begin(), hl.end(), [](u64 a, u64 b) { return (u32)a != (u32)b ? (u32)a > (u32)b : a < b; }); /* by size, then root: deterministic */
    std::vector<u64> load(n_files, 0);
    for (u64 &e : hl) {
        const u32 f = (u32)(std::min_element(load.begin(), load.end()) - load.begin());
        load[f] += (u32)e;
        e = (e & 0xFFFFFFFF00000000ull) | f;
    }
    if (n_list) {
        if (hipMemcpyAsync(list, hl.data(), n_list * sizeof(u64), hipMemcpyHostToDevice, c->stream) != hipSuccess) {
            cleanup();
            return fail(c, DISCO_E_HIP, "edge partition: upload failed");
        }
        hipLaunchKernelGGL(uf_assign_kernel, dim3((n_list + 255) / 256), dim3(256), 0, c->stream, list, n_list, cfile);
    }
    hipLaunchKernelGGL(uf_edge_file_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, d_src, d_valid, d_pos, n_slots, parent, cfile, n_files, efile);
    hipError_t e1 = hipMemcpyAsync(out, efile, ne * sizeof(u16), hipMemcpyDeviceToHost, c->stream);
    hipError_t e2 = hipStreamSynchronize(c->stream);
    cleanup();
#undef PART_CHK
    if (e1 != hipSuccess || e2 != hipSuccess) return fail(c, DISCO_E_HIP, "edge partition: copy failed");
    return (int64_t)ne;
}

int64_t disco_fetch_edge_files(disco_ctx *c, uint32_t n_files, uint16_t *out, uint64_t cap)
{
    DISCO_TRACE("disco_fetch_edge_files");
    if (!c || !out || n_files == 0) return DISCO_E_ARG;
    if (n_files > 0xFFFEu) return fail(c, DISCO_E_ARG, "disco_fetch_edge_files: at most 65534 files (%u asked for)", n_files);
    if (c->phase < 8) return fail(c, DISCO_E_STATE, "disco_fetch_edge_files: run disco_transitive_reduce first");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 ne = c->n_out;
    if (cap < ne) return fail(c, DISCO_E_ARG, "disco_fetch_edge_files: need room for %llu edges", (unsigned long long)ne);
    if (ne == 0) return 0;
    return partition_edges(c, c->d_out_src, c->d_out_ent, c->d_out_valid, c->d_out_pos, c->out_used, ne, c->n, n_files, out);
}

/* the same partition for edges held by the HOST (buildG --gpus N: the edges of all ranks, concatenated): src / dst are read ids
 * below n_nodes */
int64_t disco_partition_edges(disco_ctx *c, const disco_edge *edges, uint64_t n_edges, uint64_t n_nodes, uint32_t n_files, uint16_t *out)
{
    if (!c || !out || n_files == 0 || (n_edges && !edges)) return DISCO_E_ARG;
    if (n_files > 0xFFFEu) return fail(c, DISCO_E_ARG, "disco_partition_edges: at most 65534 files (%u asked for)", n_files);
    if (n_nodes >= (1ull << 31)) return fail(c, DISCO_E_UNSUPPORTED, "disco_partition_edges: more than 2^31 nodes");
    HIPCHK(c, hipSetDevice(c->device));
    if (n_edges == 0) return 0;
    u64 *d_src = nullptr, *d_ent = nullptr, *d_pos = nullptr;
    u8 *d_valid = nullptr;
    std::unique_ptr<u64[]> hs(new u64[n_edges]), he(new u64[n_edges]);
    parallel_for(n_edges, [&](u64 b, u64 e_) {
        for (u64 i = b; i < e_; i++) {
            hs[i] = edges[i].src;
            he[i] = ADJ_MAKE(0u, edges[i].dst, 0u, 0u);
        }
    });
    int rc = DISCO_OK;
    int64_t res = 0;
    do {
        if ((rc = dev_alloc(c, &d_src, n_edges)) != DISCO_OK) break;
        if ((rc = dev_alloc(c, &d_ent, n_edges)) != DISCO_OK) break;
        if ((rc = dev_alloc(c, &d_pos, n_edges)) != DISCO_OK) break;
        if ((rc = dev_alloc(c, &d_valid, n_edges)) != DISCO_OK) break;
        if (hipMemcpyAsync(d_src, hs.get(), n_edges * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemcpyAsync(d_ent, he.get(), n_edges * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            hipMemsetAsync(d_valid, 1, n_edges, c->stream) != hipSuccess) {
            rc = fail(c, DISCO_E_HIP, "disco_partition_edges: upload failed");
            break;
        }
        hipLaunchKernelGGL(iota_u64_kernel, dim3(flat_grid(c, n_edges)), dim3(256), 0, c->stream, d_pos, n_edges);
        res = partition_edges(c, d_src, d_ent, d_valid, d_pos, n_edges, n_edges, n_nodes, n_files, out);
    } while (0);
    dev_free(c, &d_src, n_edges);
    dev_free(c, &d_ent, n_edges);
    dev_free(c, &d_pos, n_edges);
    dev_free(c, &d_valid, n_edges);
    return rc != DISCO_OK ? rc : res;
}

/* ---- the edge lines of the text files, formatted where the edges are (disco_text.h) ------------------------------------------ */
int64_t disco_format_edges(disco_ctx *c, uint32_t n_files, const uint16_t *edge_file, const uint64_t *file_index, uint64_t *file_offsets)
{
    DISCO_TRACE("disco_format_edges");
    if (!c || !file_offsets || n_files == 0) return DISCO_E_ARG;
    if (c->phase < 8) return fail(c, DISCO_E_STATE, "disco_format_edges: run disco_transitive_reduce first");
    if (n_files > 256) return fail(c, DISCO_E_UNSUPPORTED, "disco_format_edges: more than 256 files (one placement pass per file)");
    if (c->prm.max_substitutions) return fail(c, DISCO_E_UNSUPPORTED, "disco_format_edges: the substitutions column is written by the host writer");
    HIPCHK(c, hipSetDevice(c->device));
    const u64 ne = c->n_out;
    for (uint32_t f = 0; f <= n_files; f++) file_offsets[f] = 0;
    c->text_bytes = 0;
    if (ne == 0) return 0;
    if (!edge_file && n_files > 1) return fail(c, DISCO_E_ARG, "disco_format_edges: the file of every edge is needed for more than one file");
    TextView g;
    g.src = c->d_out_src;
    g.ent = c->d_out_ent;
    g.valid = c->d_out_valid;
    g.pos = c->d_out_pos;
    g.len = c->d_len;
    g.n_slots = c->out_used;
    u64 *d_findex = nullptr, *within = nullptr, *place = nullptr;
    u8 *bytes = nullptr, *sel = nullptr;
    u16 *efile = nullptr;
    auto body = [&]() -> int {
        if (file_index) {
            CHK(dev_alloc(c, &d_findex, c->n));
            HIPCHK(c, hipMemcpyAsync(d_findex, file_index, c->n * 8, hipMemcpyHostToDevice, c->stream));
        }
        g.file_index = d_findex;
        CHK(dev_alloc(c, &bytes, ne));
        CHK(dev_alloc(c, &sel, ne));
        CHK(dev_alloc(c, &efile, ne));
        CHK(dev_alloc(c, &within, ne + 1));
        CHK(dev_alloc(c, &place, ne));
        if (edge_file) HIPCHK(c, hipMemcpyAsync(efile, edge_file, ne * sizeof(u16), hipMemcpyHostToDevice, c->stream));
        else HIPCHK(c, hipMemsetAsync(efile, 0, ne * sizeof(u16), c->stream));
        hipLaunchKernelGGL(text_measure_kernel, dim3(flat_grid(c, g.n_slots)), dim3(256), 0, c->stream, g, bytes);
        u64 base = 0;
        for (uint32_t f = 0; f < n_files; f++) {
            u64 total = 0;
            hipLaunchKernelGGL(text_select_kernel, dim3(flat_grid(c, ne)), dim3(256), 0, c->stream, bytes, efile, ne, f, sel);
            CHK((scan_exclusive<u8, u64>(c, sel, ne, within, false, &total)));
            hipLaunchKernelGGL(text_place_kernel, dim3(flat_grid(c, ne)), dim3(256), 0, c->stream, within, efile, ne, f, base, place);
            file_offsets[f] = base;
            base += total;
        }
        file_offsets[n_files] = base;
        c->text_off.assign(file_offsets, file_offsets + n_files + 1);
        CHK(ensure_cap(c, &c->d_text, &c->text_cap, std::max<u64>(base, 1)));
        hipLaunchKernelGGL(text_write_kernel, dim3(flat_grid(c, g.n_slots)), dim3(256), 0, c->stream, g, place, c->d_text);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->text_bytes = base;
        return DISCO_OK;
    };
    const int rc = body();
    dev_free(c, &d_findex, c->n);
    dev_free(c, &bytes, ne);
    dev_free(c, &sel, ne);
    dev_free(c, &efile, ne);
    dev_free(c, &within, ne + 1);
    dev_free(c, &place, ne);
    return rc != DISCO_OK ? rc : (int64_t)c->text_bytes;
}

int disco_fetch_edge_text(disco_ctx *c, char *out, uint64_t cap)
{
    if (!c || (!out && c->text_bytes)) return DISCO_E_ARG;
    if (cap < c->text_bytes) return fail(c, DISCO_E_ARG, "disco_fetch_edge_text: need room for %llu bytes", (unsigned long long)c->text_bytes);
    HIPCHK(c, hipSetDevice(c->device));
    if (c->text_bytes) {
        HIPCHK(c, hipMemcpyAsync(out, c->d_text, c->text_bytes, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return DISCO_OK;
}

/* the formatted edge lines straight into the caller's open files (fds[f] receives the bytes of file f, from its offset 0): the text
 * leaves the device in 128 MB pieces through the context's pinned ring, and while a piece travels host threads pwrite the one before
 * it — the mirror image of disco_ingest_fasta's file reader. (disco_fetch_edge_text into 2.5 GB of fresh pageable memory, then the
 * writer: 0.19 s at config 3; this: the link's 0.05 s.) */
int disco_write_edge_text(disco_ctx *c, const int *fds, uint32_t n_files, uint32_t host_threads)
{
    DISCO_TRACE("disco_write_edge_text");
    if (!c || !fds) return DISCO_E_ARG;
    if (c->text_off.size() != (size_t)n_files + 1) return fail(c, DISCO_E_STATE, "disco_write_edge_text: run disco_format_edges for %u files first", n_files);
    HIPCHK(c, hipSetDevice(c->device));
    const u64 n = c->text_bytes;
    if (n == 0) return DISCO_OK;
    const size_t HALF = 128u << 20;
    CHK(ring_back_from_rows(c));
    if (!c->h_ring) {
        if (hipHostMalloc(&c->h_ring, 2 * HALF, ring_alloc_flags()) != hipSuccess) {
            c->h_ring = nullptr;
            (void)hipGetLastError();
            return fail(c, DISCO_E_NOMEM, "disco_write_edge_text: no pinned staging memory");
        }
        c->ring_half = HALF;
        for (int i = 0; i < 2; i++) HIPCHK(c, hipEventCreateWithFlags(&c->ev_ring[i], hipEventDisableTiming));
    }
    (void)host_threads; /* one writer per file and round: writes to ONE file serialise on its inode lock (16 threads on one file: 12 GB/s;
                           one thread on each of 16 files: the link's rate) */
    std::atomic<bool> ok{true};
    /* a round moves one slice of every file: slice r of file f = bytes [r S, (r + 1) S) of it, S = the ring half divided among the files */
    const size_t S = (HALF / n_files) & ~(size_t)4095;
    if (S == 0) return fail(c, DISCO_E_UNSUPPORTED, "disco_write_edge_text: too many files for the staging ring");
    u64 longest = 0;
    for (uint32_t f = 0; f < n_files; f++) longest = std::max(longest, c->text_off[f + 1] - c->text_off[f]);
    const u64 rounds = (longest + S - 1) / S;
    auto slice = [&](uint32_t f, u64 r, u64 &lo, size_t &len) { /* file-local byte range of the slice */
        const u64 fl = c->text_off[f + 1] - c->text_off[f];
        lo = std::min<u64>(r * S, fl);
        len = (size_t)(std::min<u64>((r + 1) * S, fl) - lo);
    };
    auto drain = [&](u64 r, const char *half) {
        std::vector<std::thread> th;
        for (uint32_t f = 0; f < n_files; f++) {
            u64 lo;
            size_t len;
            slice(f, r, lo, len);
            if (!len) continue;
            th.emplace_back([&, f, lo, len]() {
                size_t done = 0;
                while (done < len) {
                    const ssize_t w = pwrite(fds[f], half + (size_t)f * S + done, len - done, (off_t)(lo + done));
                    if (w <= 0) {
                        ok.store(false);
                        return;
                    }
                    done += (size_t)w;
                }
            });
        }
        for (auto &x : th) x.join();
    };
    for (u64 r = 0; r < rounds; r++) {
        char *half = (char *)c->h_ring + (r & 1) * HALF;
        for (uint32_t f = 0; f < n_files; f++) {
            u64 lo;
            size_t len;
            slice(f, r, lo, len);
            if (len) HIPCHK(c, hipMemcpyAsync(half + (size_t)f * S, c->d_text + c->text_off[f] + lo, len, hipMemcpyDeviceToHost, c->stream));
        }
        HIPCHK(c, hipEventRecord(c->ev_ring[r & 1], c->stream));
        if (r >= 1) drain(r - 1, (const char *)c->h_ring + ((r - 1) & 1) * HALF); /* (its copies were waited for below, one round ago) */
        HIPCHK(c, hipEventSynchronize(c->ev_ring[r & 1]));
    }
    if (rounds) drain(rounds - 1, (const char *)c->h_ring + ((rounds - 1) & 1) * HALF);
    if (!ok.load()) return fail(c, DISCO_E_ARG, "disco_write_edge_text: write error");
    return DISCO_OK;
}

/* the contained rows of the current flags start their way to the host now, on a side stream (grouped: also in the order of the
 * contained-read files), instead of when they are asked for: for callers that have work between disco_mark_contained and
 * disco_fetch_contained[_grouped] — buildG: edge selection and the reduction (they lose 1-3 ms to the side stream; the fetch then
 * only waits for what is left and decodes) */
int disco_start_contained_rows(disco_ctx *c, int grouped)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase < 4) return fail(c, DISCO_E_STATE, "disco_start_contained_rows: run disco_mark_contained first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_contained == 0) return DISCO_OK;
    return start_contained_rows(c, grouped != 0);
}

/* ---- chains of the reduced graph as composite edges (kernels and the argument: disco_chains.h) ------------- */
static int contract_chains(disco_ctx *c, const u64 *d_src, const u64 *d_ent, const u8 *d_valid, const u64 *d_pos, u64 n_slots, u64 ne, u64 n, u32 min_ovl)
{
    DISCO_TRACE("contract_chains");
    c->ch_ready = false;
    if (n_slots >= (1ull << 31)) return fail(c, DISCO_E_UNSUPPORTED, "chain contraction: more than 2^31 edge slots");
    ChainView g;
    g.src = d_src;
    g.ent = d_ent;
    g.valid = d_valid;
    g.pos = d_pos;
    g.len = c->d_len;
    g.n_slots = n_slots;
    g.min_ovl = min_ovl;
    u32 *deg = nullptr, *he = nullptr, *links_of = nullptr, *live = nullptr;
    u8 *internal = nullptr, *is_head = nullptr;
    ChRank *ra = nullptr, *rb = nullptr;
    u64 *comp_id = nullptr, *link_start = nullptr;
    const u64 n_half = 2 * n_slots;
    int rc = DISCO_OK;
    auto body = [&]() -> int {
        CHK(dev_alloc(c, &deg, n));
        CHK(dev_alloc(c, &he, 2 * n));
        CHK(dev_alloc(c, &internal, n));
        CHK(dev_alloc(c, &ra, n_half));
        CHK(dev_alloc(c, &rb, n_half));
        CHK(dev_alloc(c, &live, 1));
        HIPCHK(c, hipMemsetAsync(deg, 0, n * sizeof(u32), c->stream));
        hipLaunchKernelGGL(ch_degree_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, g, deg, he);
        hipLaunchKernelGGL(ch_internal_kernel, dim3(flat_grid(c, n)), dim3(256), 0, c->stream, g, deg, he, n, internal);
        hipLaunchKernelGGL(ch_init_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, g, internal, he, ra);
        HIPCHK(c, hipGetLastError());
        /* a chain of L half-edges is ranked after ceil(log2 L) rounds and no chain has more than n_half elements; a ring of
         * absorbable nodes never finishes (its elements are discarded by ch_heads_kernel and left to the host pass), so the
         * rounds stop at that bound instead of running while anything is live */
        int max_rounds = 1;
        while (max_rounds < 32 && (1ull << max_rounds) < n_half) ++max_rounds;
        for (int round = 0; round < max_rounds; round++) {
            u32 h_live = 0;
            HIPCHK(c, hipMemsetAsync(live, 0, sizeof(u32), c->stream));
            hipLaunchKernelGGL(ch_jump_kernel, dim3(flat_grid(c, n_half)), dim3(256), 0, c->stream, ra, rb, n_half, live);
            HIPCHK(c, hipMemcpyAsync(&h_live, live, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            std::swap(ra, rb);
            if (!h_live) break;
        }
        dev_free(c, &rb, n_half);
        dev_free(c, &deg, n);
        CHK(dev_alloc(c, &is_head, n_slots));
        CHK(dev_alloc(c, &links_of, n_slots));
        CHK(ensure_cap(c, &c->d_ch_dead, &c->ch_dead_cap, std::max<u64>(ne, 1)));
        HIPCHK(c, hipMemsetAsync(c->d_ch_dead, 0, std::max<u64>(ne, 1), c->stream));
        hipLaunchKernelGGL(ch_heads_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, g, internal, ra, is_head, links_of, c->d_ch_dead);
        CHK(dev_alloc(c, &comp_id, n_slots + 1));
        CHK(dev_alloc(c, &link_start, n_slots + 1));
        u64 n_comp = 0, n_links = 0;
        CHK((scan_exclusive<u8, u64>(c, is_head, n_slots, comp_id, false, &n_comp)));
        CHK((scan_exclusive<u32, u64>(c, links_of, n_slots, link_start, false, &n_links)));
        CHK(ensure_cap(c, &c->d_ch_comp, &c->ch_comp_cap, std::max<u64>(n_comp, 1)));
        CHK(ensure_cap(c, &c->d_ch_links, &c->ch_links_cap, std::max<u64>(n_links, 1)));
        hipLaunchKernelGGL(ch_emit_kernel, dim3(flat_grid(c, n_slots)), dim3(256), 0, c->stream, g, internal, ra, comp_id, link_start, c->d_ch_comp, c->d_ch_links);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->ch_comp_n = n_comp;
        c->ch_links_n = n_links;
        c->ch_edges_n = ne;
        c->ch_ready = true;
        return DISCO_OK;
    };
    rc = body();
    dev_free(c, &deg, n);
    dev_free(c, &he, 2 * n);
    dev_free(c, &internal, n);
    dev_free(c, &ra, n_half);
    dev_free(c, &rb, n_half);
    dev_free(c, &live, 1);
    dev_free(c, &is_head, n_slots);
    dev_free(c, &links_of, n_slots);
    dev_free(c, &comp_id, n_slots + 1);
    dev_free(c, &link_start, n_slots + 1);
    return rc;
}

int disco_contract_chains(disco_ctx *c, uint32_t min_overlap_simplify, uint64_t *n_composite, uint64_t *n_links)
{
    if (!c) return DISCO_E_ARG;
    if (c->phase < 8) return fail(c, DISCO_E_STATE, "disco_contract_chains: run disco_transitive_reduce first");
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_out == 0) {
        c->ch_comp_n = c->ch_links_n = c->ch_edges_n = 0;
        c->ch_ready = true;
    } else
        CHK(contract_chains(c, c->d_out_src, c->d_out_ent, c->d_out_valid, c->d_out_pos, c->out_used, c->n_out, c->n, min_overlap_simplify));
    if (n_composite) *n_composite = c->ch_comp_n;
    if (n_links) *n_links = c->ch_links_n;
    return DISCO_OK;
}

int disco_contract_chains_of(disco_ctx *c, const disco_edge *edges, uint64_t n_edges, uint32_t min_overlap_simplify, uint64_t *n_composite, uint64_t *n_links)
{
    if (!c || (n_edges && !edges)) return DISCO_E_ARG;
    if (c->phase < 1) return fail(c, DISCO_E_STATE, "disco_contract_chains_of: the context holds no reads (their lengths are needed)");
    HIPCHK(c, hipSetDevice(c->device));
    c->ch_ready = false;
    if (n_edges == 0) {
        c->ch_comp_n = c->ch_links_n = c->ch_edges_n = 0;
        c->ch_ready = true;
    } else {
        u64 *d_src = nullptr, *d_ent = nullptr, *d_pos = nullptr;
        u8 *d_valid = nullptr;
        std::unique_ptr<u64[]> hs(new u64[n_edges]), he(new u64[n_edges]);
        std::atomic<bool> bad{false};
        parallel_for(n_edges, [&](u64 b, u64 e_) {
            for (u64 i = b; i < e_; i++) {
                if (edges[i].src >= c->n || edges[i].dst >= c->n) bad.store(true, std::memory_order_relaxed);
                hs[i] = edges[i].src;
                he[i] = ADJ_MAKE(edges[i].offset, edges[i].dst, edges[i].orient, edges[i].len_dst);
            }
        });
        if (bad) return fail(c, DISCO_E_ARG, "disco_contract_chains_of: an edge names a read the context does not hold");
        int rc = DISCO_OK;
        do {
            if ((rc = dev_alloc(c, &d_src, n_edges)) != DISCO_OK) break;
            if ((rc = dev_alloc(c, &d_ent, n_edges)) != DISCO_OK) break;
            if ((rc = dev_alloc(c, &d_pos, n_edges)) != DISCO_OK) break;
            if ((rc = dev_alloc(c, &d_valid, n_edges)) != DISCO_OK) break;
            if (hipMemcpyAsync(d_src, hs.get(), n_edges * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(d_ent, he.get(), n_edges * 8, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemsetAsync(d_valid, 1, n_edges, c->stream) != hipSuccess) {
                rc = fail(c, DISCO_E_HIP, "disco_contract_chains_of: upload failed");
                break;
            }
            hipLaunchKernelGGL(iota_u64_kernel, dim3(flat_grid(c, n_edges)), dim3(256), 0, c->stream, d_pos, n_edges);
            rc = contract_chains(c, d_src, d_ent, d_valid, d_pos, n_edges, n_edges, c->n, min_overlap_simplify);
        } while (0);
        dev_free(c, &d_src, n_edges);
        dev_free(c, &d_ent, n_edges);
        dev_free(c, &d_pos, n_edges);
        dev_free(c, &d_valid, n_edges);
        if (rc != DISCO_OK) return rc;
    }
    if (n_composite) *n_composite = c->ch_comp_n;
    if (n_links) *n_links = c->ch_links_n;
    return DISCO_OK;
}

int disco_fetch_chains(disco_ctx *c, disco_chain_edge *comp, disco_chain_link *links, uint8_t *edge_absorbed)
{
    if (!c) return DISCO_E_ARG;
    if (!c->ch_ready) return fail(c, DISCO_E_STATE, "disco_fetch_chains: run disco_contract_chains first");
    HIPCHK(c, hipSetDevice(c->device));
    static_assert(sizeof(disco_chain_edge) == sizeof(ChainEdgeOut) && sizeof(disco_chain_link) == sizeof(ChainLinkOut), "chain records");
    if (comp && c->ch_comp_n) HIPCHK(c, hipMemcpyAsync(comp, c->d_ch_comp, c->ch_comp_n * sizeof(ChainEdgeOut), hipMemcpyDeviceToHost, c->stream));
    if (links && c->ch_links_n) HIPCHK(c, hipMemcpyAsync(links, c->d_ch_links, c->ch_links_n * sizeof(ChainLinkOut), hipMemcpyDeviceToHost, c->stream));
    if (edge_absorbed && c->ch_edges_n) HIPCHK(c, hipMemcpyAsync(edge_absorbed, c->d_ch_dead, c->ch_edges_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DISCO_OK;
}

int disco_set_query_order(disco_ctx *c, const void *d_order_u64)
{
    if (!c) return DISCO_E_ARG;
    c->d_order = (const u64 *)d_order_u64;
    c->order_external = d_order_u64 != nullptr;
    return DISCO_OK;
}

int disco_get_query_order(disco_ctx *c, const void **d_order_u64)
{
    if (!c || !d_order_u64) return DISCO_E_ARG;
    *d_order_u64 = c->d_order_used;
    return DISCO_OK;
}

int disco_phase_ms(disco_ctx *c, float *ms, int n)
{
    if (!c || !ms) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    ph_collect(c);
    for (int i = 0; i < n && i < DISCO_PH_COUNT; i++) ms[i] = c->ph_ms[i];
    return DISCO_OK;
}

int disco_memcpy_d2d(disco_ctx *c, void *dst, const void *src, uint64_t bytes)
{
    if (!c || (bytes && (!dst || !src))) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes) HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return DISCO_OK;
}

int disco_measure_hbm(disco_ctx *c, uint64_t bytes, int reps, double *gb_per_s)
{
    if (!c || !gb_per_s || bytes < 4096 || reps < 1) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const u64 n = bytes / sizeof(ulonglong2);
    ulonglong2 *src = nullptr, *dst = nullptr;
    if (hipMalloc(&src, n * sizeof(ulonglong2)) != hipSuccess || hipMalloc(&dst, n * sizeof(ulonglong2)) != hipSuccess) {
        if (src) (void)hipFree(src);
        return fail(c, DISCO_E_NOMEM, "disco_measure_hbm: cannot allocate 2 x %llu bytes", (unsigned long long)bytes);
    }
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipMemsetAsync(src, 0x5A, n * sizeof(ulonglong2), c->stream));
    const int grid = c->n_cu * 16;
    float best = 1e30f;
    for (int r = 0; r <= reps; r++) { /* launch 0 warms up */
        HIPCHK(c, hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, c->stream, src, dst, n);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(src);
    (void)hipFree(dst);
    *gb_per_s = 2.0 * (double)(n * sizeof(ulonglong2)) / ((double)best * 1e-3) / 1e9;
    return DISCO_OK;
}

int disco_measure_gather(disco_ctx *c, uint64_t bytes, int reps, double *gb_per_s)
{
    if (!c || !gb_per_s || bytes < 4096 || reps < 1) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const u64 nrows = bytes / 64;
    ulonglong2 *tab = nullptr;
    u64 *sink = nullptr;
    if (hipMalloc(&tab, nrows * 64) != hipSuccess) return fail(c, DISCO_E_NOMEM, "disco_measure_gather: cannot allocate %llu bytes", (unsigned long long)bytes);
    HIPCHK(c, hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    HIPCHK(c, hipMemsetAsync(tab, 0x5A, nrows * 64, c->stream));
    const int grid = c->n_cu * 64; /* 8 waves per SIMD */
    const u32 per_lane = 256;
    float best = 1e30f;
    for (int r = 0; r <= reps; r++) { /* launch 0 warms up */
        HIPCHK(c, hipEventRecord(e0, c->stream));
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid), dim3(256), 0, c->stream, tab, nrows, per_lane, sink);
        HIPCHK(c, hipEventRecord(e1, c->stream));
        HIPCHK(c, hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(tab);
    (void)hipFree(sink);
    *gb_per_s = (double)grid * 256.0 * per_lane * 64.0 / ((double)best * 1e-3) / 1e9;
    return DISCO_OK;
}

int disco_get_counters(disco_ctx *c, disco_counters *o)
{
    if (!c || !o) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    memset(o, 0, sizeof *o);
    o->n_reads = c->n;
    if (c->phase >= 1) {
        CHK(ensure_host_len(c));
        u64 q = 0;
        for (u64 i = c->q_lo; i < c->q_hi; i++) q += c->h_len[i] - c->k;
        o->probes = q;
    }
    o->kmer_hits = c->h_ctr[CTR_KMER_HITS];
    o->raw_hits = c->h_ctr[CTR_RAW_HITS];
    o->n_contained = c->n_contained;
    o->e_pre = c->phase >= 6 ? c->adj_total / 2 : 0;
    o->e_out = c->n_out;
    o->cap_bind_sites = c->h_ctr[CTR_CAP_SITES];
    o->asymmetric_pairs = c->asym_local;
    o->big_rows = c->big_rows;
    o->index_buckets = c->T;
    o->hbm_bytes = c->hbm_bytes;
    return DISCO_OK;
}

} /* extern "C" */

/* ================================================================================================================
 * multi-GPU flow (include/disco_hip.h "multi-GPU flow"; kernels: disco_dist.h; transport: disco_comm.h)
 *
 * One pass on rank r of G (own = [r*per, (r+1)*per) ∩ [0, n)):
 *   0. [flag] all-gather of the packed reads               every rank verifies candidates against any read
 *   1. index records of the OWN reads (no atomics)  →  all-to-all to the owner of their bucket range  →  the owner counts,
 *      scans and fills its shard straight into its slice of the global table  →  all-gather-v of the bkt / ent slices
 *   2. grouping + probe + verify of the own reads (unchanged kernels)
 *   3. reduce-scatter(MIN) of the containment keys, flags of the own range, all-gather of the bitmap
 *   4. edge selection of the own reads
 *   5. regular regime (no rank dropped a verified hit): rows of the neighbours a node's marking sweeps are requested from
 *      their owners, class-filtered 4-byte entries come back; marking; nodes that needed a row nobody had asked for are redone
 *      after a second, request-everything round
 *   6. surviving half-edges pushed to the owner of the smaller endpoint; emission
 *   order-dependent regime (some rank dropped a hit: per-k-mer cap, second hit to a destination): the whole adjacency is
 *   gathered and every rank completes, marks and judges all lists itself (exact, not scalable; never on BASELINE data)
 * ============================================================================================================== */

#define COMM_CHK(c, expr)                                                                        \
    do {                                                                                         \
        int rc_ = (expr);                                                                        \
        if (rc_ != DISCO_OK) return fail((c), rc_, "%s: %s", #expr, (c)->comm->err.c_str());    \
    } while (0)

/* grow-only buffer that keeps its first `used` elements */
template <typename T>
static int ensure_cap_keep(disco_ctx *c, T **p, u64 *cap, u64 need, u64 used)
{
    if (*p && need <= *cap) return DISCO_OK;
    T *q = nullptr;
    const u64 ncap = std::max<u64>(need + need / 8, 1);
    CHK(dev_alloc(c, &q, ncap));
    if (*p && used) HIPCHK(c, hipMemcpyAsync(q, *p, used * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    dev_free(c, p, *cap);
    *p = q;
    *cap = ncap;
    return DISCO_OK;
}

static void dist_range(const disco_ctx *c, u64 n, u64 *per, u64 *lo, u64 *hi)
{
    const u64 G = c->comm ? (u64)c->comm->world : 1, r = c->comm ? (u64)c->comm->rank : 0;
    u64 p = (n + G - 1) / G;
    p = (p + 63) & ~63ull;
    if (p == 0) p = 64;
    *per = p;
    *lo = std::min(r * p, n);
    *hi = std::min(*lo + p, n);
}

/* partition items[0..n) into one segment per destination rank: out = segments in rank order, cnt[g] = items for rank g */
template <typename T, typename F>
static int route_items(disco_ctx *c, const T *items, u64 n, F owner, T *out, std::vector<u64> &cnt)
{
    const u32 G = (u32)c->comm->world;
    cnt.assign(G, 0);
    if (!c->d_route) CHK(dev_alloc(c, &c->d_route, 2 * DIST_MAX_WORLD));
    HIPCHK(c, hipMemsetAsync(c->d_route, 0, 2 * DIST_MAX_WORLD * sizeof(u64), c->stream));
    if (n) hipLaunchKernelGGL((route_count_kernel<T, F>), dim3(flat_grid(c, n)), dim3(256), 0, c->stream, items, n, owner, G, c->d_route);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(cnt.data(), c->d_route, G * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    /* (the cursors are made on the device: the scattered segments are consumed by operations on this stream, nothing here waits for them) */
    hipLaunchKernelGGL(route_cursor_kernel, dim3(1), dim3(64), 0, c->stream, (const u64 *)c->d_route, G, c->d_route + DIST_MAX_WORLD);
    if (n) hipLaunchKernelGGL((route_scatter_kernel<T, F>), dim3(flat_grid(c, (n + ROUTE_ITEMS - 1) / ROUTE_ITEMS)), dim3(256), 0, c->stream, items, n, owner, G,
                              c->d_route + DIST_MAX_WORLD, out);
    HIPCHK(c, hipGetLastError());
    return DISCO_OK;
}

/* rcnt[p] = what rank p sends to this rank, from everybody's send counts */
static int exchange_counts(disco_ctx *c, const std::vector<u64> &scnt, std::vector<u64> &rcnt, std::vector<u64> *matrix = nullptr)
{
    const int G = c->comm->world;
    std::vector<u64> all((size_t)G * G);
    COMM_CHK(c, c->comm->host_all_gather((const unsigned long long *)scnt.data(), G, (unsigned long long *)all.data(), c->stream));
    rcnt.resize((size_t)G);
    for (int p = 0; p < G; p++) rcnt[(size_t)p] = all[(size_t)p * G + c->comm->rank];
    if (matrix) *matrix = std::move(all);
    return DISCO_OK;
}

/* all-to-all of item segments (elem bytes per item), both sides compact in rank order */
static int a2a_items(disco_ctx *c, int xid, const void *send, const std::vector<u64> &scnt, void *recv, const std::vector<u64> &rcnt, size_t elem)
{
    const int G = c->comm->world;
    std::vector<size_t> so((size_t)G), sc((size_t)G), ro((size_t)G), rc((size_t)G);
    size_t a = 0, b = 0;
    for (int p = 0; p < G; p++) {
        so[(size_t)p] = a;
        sc[(size_t)p] = scnt[(size_t)p] * elem;
        a += sc[(size_t)p];
        ro[(size_t)p] = b;
        rc[(size_t)p] = rcnt[(size_t)p] * elem;
        b += rc[(size_t)p];
        if (p != c->comm->rank) c->dinfo.bytes_sent[xid] += sc[(size_t)p];
    }
    const auto t0 = HClock::now();
    COMM_CHK(c, c->comm->all_to_all_v(send, so.data(), sc.data(), recv, ro.data(), rc.data(), c->stream));
    /* (stream ordered: whoever consumes the received blocks runs behind them on this stream. DISCO_DIST_TIME_EXCHANGES=1 waits here, so that
     * dinfo.ms shows the exchange itself and not the time to issue it) */
    if (c->time_exchanges) HIPCHK(c, hipStreamSynchronize(c->stream));
    c->dinfo.ms[xid] += ms_since(t0);
    return DISCO_OK;
}

static u64 vsum(const std::vector<u64> &v)
{
    u64 s = 0;
    for (u64 x : v) s += x;
    return s;
}

/* sum / max of a few host values over all ranks */
static int host_reduce(disco_ctx *c, u64 *vals, int n, bool take_max = false)
{
    const int G = c->comm->world;
    std::vector<u64> all((size_t)G * n);
    COMM_CHK(c, c->comm->host_all_gather((const unsigned long long *)vals, n, (unsigned long long *)all.data(), c->stream));
    for (int i = 0; i < n; i++) {
        u64 a = 0;
        for (int p = 0; p < G; p++) a = take_max ? std::max(a, all[(size_t)p * n + i]) : a + all[(size_t)p * n + i];
        vals[i] = a;
    }
    return DISCO_OK;
}

/* the rows and lengths of the own reads, from the ranks in whose home ranges they arrived (dist_deal_reads) */
template <int W>
static int dist_deal_rows(disco_ctx *c)
{
    typedef ReadItem<W> Item;
    const u64 nhome = c->home_hi - c->home_lo;
    const u64 per16 = (sizeof(Item) + 15) / 16; /* (d_x16a / d_x16b are arrays of 16-byte items) */
    CHK(ensure_cap(c, &c->d_x16a, &c->x16a_cap, std::max<u64>(2 * nhome * per16, 1)));
    Item *items = (Item *)c->d_x16a, *sorted = (Item *)(c->d_x16a + nhome * per16);
    if (nhome) hipLaunchKernelGGL(read_items_kernel<W>, dim3(flat_grid(c, nhome * (W + 1))), dim3(256), 0, c->stream, (const u64 *)c->d_reads, (const u16 *)c->d_len, c->S, c->home_lo, c->home_hi, items);
    HIPCHK(c, hipGetLastError());
    std::vector<u64> scnt, rcnt;
    RouteByReadItem<W> f{c->d_otab};
    CHK(route_items(c, items, nhome, f, sorted, scnt));
    CHK(exchange_counts(c, scnt, rcnt));
    const u64 nr = vsum(rcnt);
    if (nr != c->n_own) return fail(c, DISCO_E_STATE, "dealt reads: %llu rows arrive for %llu own reads", (unsigned long long)nr, (unsigned long long)c->n_own);
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(nr * per16, 1)));
    CHK(a2a_items(c, DISCO_X_READS_DEALT, sorted, scnt, c->d_x16b, rcnt, sizeof(Item)));
    if (nr) hipLaunchKernelGGL(read_items_place_kernel<W>, dim3(flat_grid(c, nr * (W + 1))), dim3(256), 0, c->stream, (const Item *)c->d_x16b, nr, c->S, c->d_reads, c->d_len);
    HIPCHK(c, hipGetLastError());
    return DISCO_OK;
}

/* ---- 0'. ranks own loci: the reads of the pass are dealt to the ranks by their read-level minimizer -------------------------- */
/* Replaces needsProcessing (RMA/HashTable.cpp:1066-1087: a read is the business of the rank that owns its bucket) and the id
 * ranges of rounds 1-4. A rank's reads used to be a random G-th of every locus: nobody's neighbours in the processing order shared
 * candidates, rows or buckets any more (verify, probe and marking ran 1.5 x the time per read: profiles/r04_dist8_kernels.json) and
 * nearly every neighbour row was another rank's. Now: keys of the home range (read_keys_kernel) -> all-gather of the 4-byte keys ->
 * owner of every read = its key's share of the hash range (disco_key_owner: whole groups, no exchange of boundaries needed — the
 * grouping hash spreads the groups evenly) -> the own reads, grouped (the rank's processing order, d_order_own). From here on q_lo /
 * q_hi are POSITIONS of that list. Collective. */
static int dist_deal_reads(disco_ctx *c)
{
    DISCO_TRACE("dist_deal_reads");
    const u32 G = (u32)c->comm->world, r = (u32)c->comm->rank;
    const u64 hlo = c->home_lo, hhi = c->home_hi;
    CHK(ensure_cap(c, &c->d_okey, &c->okey_cap, c->n_alloc));
    if (!c->d_otab) CHK(dev_alloc(c, &c->d_otab, c->n_alloc));
    DiscoView v = view(c);
    if (hhi > hlo) {
        if (v.m == RUNS_M) hipLaunchKernelGGL(read_keys_kernel<true>, dim3((unsigned)((hhi - hlo + 255) / 256)), dim3(256), 0, c->stream, v, hlo, hhi, c->d_okey);
        else hipLaunchKernelGGL(read_keys_kernel<false>, dim3((unsigned)((hhi - hlo + 255) / 256)), dim3(256), 0, c->stream, v, hlo, hhi, c->d_okey);
    }
    HIPCHK(c, hipGetLastError());
    {
        const auto t0 = HClock::now();
        COMM_CHK(c, c->comm->all_gather(c->d_okey + (u64)r * c->per, c->d_okey, c->per * sizeof(u32), c->stream));
        c->dinfo.bytes_sent[DISCO_X_KEYS] += (u64)(G - 1) * c->per * sizeof(u32);
        c->dinfo.ms[DISCO_X_KEYS] += ms_since(t0);
    }
    /* owners and the own reads (about n / G of them: sized for the whole job once — 4 bytes per read) */
    CHK(ensure_cap(c, &c->d_own_ids, &c->own_ids_cap, std::max<u64>(c->n, 1)));
    if (!c->d_list_n) CHK(dev_alloc(c, &c->d_list_n, 1));
    HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
    /* the grouping of the own reads (the rank's processing order) counts its buckets in the same pass: how many buckets follows from
     * the number of own reads, which this pass produces — it is taken from the job's share n / G instead (the keys spread the reads to
     * 0.3 %); the number of buckets shapes the order, never a result */
    int obits = 16;
    const bool order_own = own_order_wanted(c, std::max<u64>(c->n / G, 1), &obits) && !getenv("DISCO_DIST_ORDER_TWO_PASSES");
    const u32 oshift = 32u - (u32)obits;
    if (order_own) {
        CHK(ensure_cap(c, &c->d_ocnt, &c->ocnt_cap, (1ull << obits) + 1));
        CHK(ensure_cap(c, &c->d_oslot, &c->oslot_cap, std::max<u64>(c->n, 1))); /* (about n / G are used: sized like the list of own ids) */
        HIPCHK(c, hipMemsetAsync(c->d_ocnt, 0, ((1ull << obits) + 1) * sizeof(u32), c->stream));
    }
    if (c->n) hipLaunchKernelGGL(own_select_kernel, dim3((unsigned)std::min<u64>((c->n + OWN_TILE - 1) / OWN_TILE, (u64)c->n_cu * 16)), dim3(256), 0, c->stream, (const u32 *)c->d_okey, c->n, G, r,
                                 c->d_otab, c->d_own_ids, c->d_list_n, order_own ? c->d_ocnt : (u32 *)nullptr, order_own ? c->d_oslot : (u32 *)nullptr, oshift);
    HIPCHK(c, hipGetLastError());
    u64 n_own = 0;
    HIPCHK(c, hipMemcpyAsync(&n_own, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, counted_stream_sync(c->stream));
    c->n_own = n_own;
    c->loci = true;
    c->q_lo = 0;
    c->q_hi = n_own;
    /* the lengths and — for the index pass — the rows of the own reads: most of them arrived in other ranks' ranges. They come by an
     * all-to-all of their own, a seventh of what the all-gather of ALL reads moves (that one stays behind index build and probe: only
     * verify waits for it; it writes the same rows once more); shapes without a built variant wait for the all-gather here */
    if (c->wait_bulk_before_verify) {
        const int W = (int)((c->max_len + 31) / 32);
        if (W <= 5 && c->S >= 5 && !getenv("DISCO_DIST_NO_DEAL_ROWS")) CHK(dist_deal_rows<5>(c));
        else if (W <= 8 && c->S >= 8 && !getenv("DISCO_DIST_NO_DEAL_ROWS")) CHK(dist_deal_rows<8>(c));
        else {
            HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_bulk, 0));
            c->wait_bulk_before_verify = false;
        }
    }
    /* the processing order: the own reads grouped by key (disco_probe's grouping, over the list) */
    CHK(ensure_cap(c, &c->d_order_own, &c->order_cap, std::max<u64>(n_own, 1)));
    ph_begin(c, DISCO_PH_ORDER);
    int obits2 = 16;
    const bool order_two = !order_own && own_order_wanted(c, n_own, &obits2); /* (DISCO_DIST_ORDER_TWO_PASSES: the counting pass of rounds 5 as a pass of its own) */
    if ((order_own && n_own) || order_two) {
        const int ob = order_own ? obits : obits2;
        const u64 order_buckets = 1ull << ob;
        const u32 oshift = 32u - (u32)ob;
        if (order_two) {
            CHK(ensure_cap(c, &c->d_ocnt, &c->ocnt_cap, order_buckets + 1));
            CHK(ensure_cap(c, &c->d_oslot, &c->oslot_cap, n_own));
            HIPCHK(c, hipMemsetAsync(c->d_ocnt, 0, (order_buckets + 1) * sizeof(u32), c->stream));
            hipLaunchKernelGGL(order_count_list_kernel, dim3(flat_grid(c, n_own)), dim3(256), 0, c->stream, (const u32 *)c->d_okey, (const u32 *)c->d_own_ids, n_own, G, r, oshift, c->d_ocnt, c->d_oslot);
        }
        CHK((scan_exclusive<u32, u32>(c, c->d_ocnt, order_buckets + 1, c->d_ocnt, false, nullptr)));
        hipLaunchKernelGGL(order_scatter_list_kernel, dim3(flat_grid(c, n_own)), dim3(256), 0, c->stream, (const u32 *)c->d_okey, (const u32 *)c->d_own_ids, (const u32 *)c->d_oslot,
                           (const u32 *)c->d_ocnt, n_own, G, r, oshift, (const u16 *)c->d_len, c->d_order_own);
    } else if (n_own)
        hipLaunchKernelGGL(order_pack_list_kernel, dim3(flat_grid(c, n_own)), dim3(256), 0, c->stream, (const u32 *)c->d_own_ids, n_own, (const u16 *)c->d_len, c->d_order_own);
    ph_end(c, DISCO_PH_ORDER);
    HIPCHK(c, hipGetLastError());
    c->d_order_used = c->d_order_own;
    c->order_counted = false;
    c->dinfo.own_reads = n_own;
    return DISCO_OK;
}

/* the count pass of the index over the own list (ranks own loci): records and minimizer runs by POSITION in the processing order */
static int index_count_own_list(disco_ctx *c, const DiscoView &v, ulonglong2 *rec)
{
    const u64 n_own = c->n_own;
    const int nf = v.k - v.m + 1;
    c->runs_lpr = 0;
    c->runs_n = 0;
    c->runs_by_pos = false;
    const int lpr = c->S == VERIFY_SW ? runs_lpr_for(c, nf, c->max_len, n_own) : 0;
    if (!n_own) return DISCO_OK;
    const dim3 grid((unsigned)((n_own + 255) / 256));
    const u64 *list = c->d_order_own;
    if (lpr) {
        CHK(ensure_cap(c, &c->d_runs, &c->runs_cap, n_own * (u64)lpr));
        c->runs_lpr = lpr;
        c->runs_lo = 0;
        c->runs_n = n_own;
        c->runs_by_pos = true;
#define DISCO_RUNS_LAUNCH(NF_)                                                                                                                              \
    do {                                                                                                                                                  \
        if (lpr == 16) hipLaunchKernelGGL((index_runs_kernel<false, NF_, 1>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, (u32 *)nullptr, (u64)0, n_own, c->d_runs, (u32 *)nullptr, (u32 *)nullptr, 0u, list); \
        else hipLaunchKernelGGL((index_runs_kernel<false, NF_, 2>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, (u32 *)nullptr, (u64)0, n_own, c->d_runs, (u32 *)nullptr, (u32 *)nullptr, 0u, list);             \
    } while (0)
#define DISCO_RUNS_LAUNCH_RT(NFMAX_, LONGK_)                                                                                                                 \
    do {                                                                                                                                                  \
        if (lpr == 16) hipLaunchKernelGGL((index_runs_kernel<false, 0, 1, NFMAX_, LONGK_>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, (u32 *)nullptr, (u64)0, n_own, c->d_runs, (u32 *)nullptr, (u32 *)nullptr, 0u, list); \
        else hipLaunchKernelGGL((index_runs_kernel<false, 0, 2, NFMAX_, LONGK_>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, (u32 *)nullptr, (u64)0, n_own, c->d_runs, (u32 *)nullptr, (u32 *)nullptr, 0u, list);             \
    } while (0)
        const bool built = v.m == RUNS_M && (nf == 7 || nf == 12 || nf == 17 || nf == 22 || nf == 27);
        if (!built) {
            if (c->k > 64) {
                if (nf <= 48) DISCO_RUNS_LAUNCH_RT(48, true);
                else DISCO_RUNS_LAUNCH_RT(64, true);
            } else if (nf <= 8) DISCO_RUNS_LAUNCH_RT(8, false);
            else if (nf <= 12) DISCO_RUNS_LAUNCH_RT(12, false);
            else if (nf <= 16) DISCO_RUNS_LAUNCH_RT(16, false);
            else if (nf <= 24) DISCO_RUNS_LAUNCH_RT(24, false);
            else if (nf <= 32) DISCO_RUNS_LAUNCH_RT(32, false);
            else if (nf <= 48) DISCO_RUNS_LAUNCH_RT(48, false);
            else DISCO_RUNS_LAUNCH_RT(64, false);
        } else
            switch (nf) {
            case 7: DISCO_RUNS_LAUNCH(7); break;
            case 12: DISCO_RUNS_LAUNCH(12); break;
            case 17: DISCO_RUNS_LAUNCH(17); break;
            case 22: DISCO_RUNS_LAUNCH(22); break;
            default: DISCO_RUNS_LAUNCH(27); break;
            }
#undef DISCO_RUNS_LAUNCH
#undef DISCO_RUNS_LAUNCH_RT
    } else if (c->k > 64)
        hipLaunchKernelGGL((index_count_kernel<false, true>), grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, (u32 *)nullptr, (u64)0, n_own, (u32 *)nullptr, (u32 *)nullptr, 0u, list);
    else
        hipLaunchKernelGGL(index_count_kernel<false>, grid, dim3(256), 0, c->stream, v, c->d_bkt, rec, (u32 *)nullptr, (u64)0, n_own, (u32 *)nullptr, (u32 *)nullptr, 0u, list);
    HIPCHK(c, hipGetLastError());
    return DISCO_OK;
}

/* ---- 1. hash-partitioned index build ------------------------------------------------------------------------------- */
static int dist_build_index(disco_ctx *c)
{
    DISCO_TRACE("dist_build_index");
    const u32 G = (u32)c->comm->world, r = (u32)c->comm->rank;
    const u64 nloc = c->q_hi - c->q_lo;
    u64 T = 1024;
    int logT = 10;
    double tscale = 2.0;
    if (const char *e = getenv("DISCO_BUCKET_SCALE")) tscale = atof(e);
    while ((double)T < tscale * (double)c->n && logT < 32) {
        T <<= 1;
        logT++;
    }
    c->T = T;
    c->bshift = 64 - logT;
    auto blo_of = [&](u64 g) { return (g * T + G - 1) / G; };
    const u64 blo = blo_of(r), bhi = blo_of(r + 1);
    const bool part = c->part_index;
    if (!part) { /* (partitioned: the slices are sized below, once the records have arrived) */
        CHK(ensure_cap(c, &c->d_bkt, &c->bkt_cap, T + 1));
        CHK(ensure_cap(c, &c->d_ent, &c->ent_cap, 2 * c->n));
    }
    CHK(ensure_cap(c, &c->d_okey, &c->okey_cap, c->loci ? c->n_alloc : c->n));
    CHK(ensure_cap(c, &c->d_rec, &c->rec_cap, std::max<u64>(2 * nloc, 1)));
    c->adj_imported = false;
    ph_begin(c, DISCO_PH_INDEX);
    DiscoView v = view(c);
    if (c->loci) CHK(index_count_own_list(c, v, c->d_rec));
    else {
        IndexCountPlan pl;
        CHK(index_count_plan(c, v, c->q_lo, c->q_hi, &pl));
        CHK(index_count_chunk<false>(c, v, pl, c->d_rec, c->q_lo, c->q_hi));
        if (c->two_class && nloc) {
            /* the long reads of the rank's range — long_ids is ascending, ovf[i] counts the long reads in front of read i: they are the
             * x in [ovf[q_lo], ovf[q_hi]) — from their full rows; records by read id from q_lo (the kernel indexes rec by read id) */
            u32 xb[2] = {0, (u32)c->n_long};
            HIPCHK(c, hipMemcpyAsync(&xb[0], c->d_ovf + c->q_lo, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            if (c->q_hi < c->n) HIPCHK(c, hipMemcpyAsync(&xb[1], c->d_ovf + c->q_hi, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (xb[1] > xb[0]) {
                const dim3 g((unsigned)((xb[1] - xb[0] + 255) / 256));
                ulonglong2 *rec_by_id = c->d_rec - 2 * (ptrdiff_t)c->q_lo; /* rec_by_id[2 i] = the record of read i (never touched outside [q_lo, q_hi)) */
                u32 *oslot_by_id = pl.oslot ? pl.oslot - (ptrdiff_t)c->q_lo : nullptr; /* (the grouping's slots: by read id too) */
                if (c->k > 64) hipLaunchKernelGGL((index_count_kernel<false, true, true>), g, dim3(256), 0, c->stream, v, c->d_bkt, rec_by_id, c->d_okey, (u64)xb[0], (u64)xb[1], pl.ocnt, oslot_by_id, pl.oshift);
                else hipLaunchKernelGGL((index_count_kernel<false, false, true>), g, dim3(256), 0, c->stream, v, c->d_bkt, rec_by_id, c->d_okey, (u64)xb[0], (u64)xb[1], pl.ocnt, oslot_by_id, pl.oshift);
                HIPCHK(c, hipGetLastError());
            }
        }
    }
    /* records -> owner of their bucket range */
    CHK(ensure_cap(c, &c->d_x16a, &c->x16a_cap, std::max<u64>(2 * nloc, 1)));
    std::vector<u64> scnt, rcnt, matrix;
    RouteByBucket f{logT, G};
    CHK(route_items(c, c->d_rec, 2 * nloc, f, c->d_x16a, scnt));
    CHK(exchange_counts(c, scnt, rcnt, &matrix));
    const u64 nrec = vsum(rcnt);
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(nrec, 1)));
    CHK(a2a_items(c, DISCO_X_INDEX_RECORDS, c->d_x16a, scnt, c->d_x16b, rcnt, sizeof(ulonglong2)));
    /* my bucket range and the position of my records in the global record array (shards in rank order) */
    std::vector<u64> shard((size_t)G, 0);
    for (u32 p = 0; p < G; p++)
        for (u32 q = 0; q < G; q++) shard[q] += matrix[(size_t)p * G + q];
    u64 base = 0;
    for (u32 q = 0; q < r; q++) base += shard[q];
    if (shard[r] != nrec) return fail(c, DISCO_E_STATE, "index shard: %llu records received, %llu announced", (unsigned long long)nrec, (unsigned long long)shard[r]);
    if (vsum(shard) != 2 * c->n) return fail(c, DISCO_E_STATE, "index: %llu records over all ranks, expected %llu", (unsigned long long)vsum(shard), (unsigned long long)(2 * c->n));
    if (part) {
        /* the index STAYS partitioned: this rank's slice of the bucket table (bhi - blo + 1 entries, bucket b at [b - blo]) and its
         * records (positions inside the slice) are all it ever holds; the lookups come to it (dist_partitioned_probe) */
        CHK(ensure_cap(c, &c->d_bkt, &c->bkt_cap, bhi - blo + 1));
        CHK(ensure_cap(c, &c->d_ent, &c->ent_cap, std::max<u64>(nrec, 1)));
        u32 *bkt0 = c->d_bkt - blo; /* indexed by global bucket number */
        HIPCHK(c, hipMemsetAsync(c->d_bkt, 0, (bhi - blo + 1) * sizeof(u32), c->stream));
        if (nrec) hipLaunchKernelGGL(shard_count_kernel, dim3(flat_grid(c, nrec)), dim3(256), 0, c->stream, c->d_x16b, nrec, bkt0);
        CHK((scan_exclusive<u32, u32>(c, c->d_bkt, bhi - blo + 1, c->d_bkt, false, nullptr)));
        if (nrec) hipLaunchKernelGGL(index_fill_kernel, dim3(flat_grid(c, nrec)), dim3(256), 0, c->stream, nrec, c->d_x16b, bkt0, c->d_ent);
        HIPCHK(c, hipGetLastError());
        c->part_blo = blo;
        c->part_bhi = bhi;
        c->part_nrec = nrec;
        ph_end(c, DISCO_PH_INDEX);
        c->phase = 2;
        return DISCO_OK;
    }
    if (bhi > blo) HIPCHK(c, hipMemsetAsync(c->d_bkt + blo, 0, (bhi - blo) * sizeof(u32), c->stream));
    if (nrec) hipLaunchKernelGGL(shard_count_kernel, dim3(flat_grid(c, nrec)), dim3(256), 0, c->stream, c->d_x16b, nrec, c->d_bkt);
    if (bhi > blo) {
        CHK((scan_exclusive<u32, u32>(c, c->d_bkt + blo, bhi - blo, c->d_bkt + blo, false, nullptr)));
        if (base) hipLaunchKernelGGL(add_u32_kernel, dim3(flat_grid(c, bhi - blo)), dim3(256), 0, c->stream, c->d_bkt + blo, bhi - blo, (u32)base);
    }
    if (nrec) hipLaunchKernelGGL(index_fill_kernel, dim3(flat_grid(c, nrec)), dim3(256), 0, c->stream, nrec, c->d_x16b, c->d_bkt, c->d_ent);
    const u32 total = (u32)(2 * c->n);
    HIPCHK(c, hipMemcpyAsync(c->d_bkt + T, &total, sizeof(u32), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    /* shards -> everybody (in place: every rank wrote its slices where they belong) */
    {
        std::vector<size_t> off((size_t)G), cnt((size_t)G);
        const auto t0 = HClock::now();
        for (u32 p = 0; p < G; p++) {
            off[p] = blo_of(p) * sizeof(u32);
            cnt[p] = (blo_of(p + 1) - blo_of(p)) * sizeof(u32);
        }
        c->dinfo.bytes_sent[DISCO_X_INDEX_SHARDS] += (u64)(G - 1) * cnt[r]; /* (rounds 1-4 added cnt[r] inside the loop, before it was set for p < r: ranks above 0 under-reported) */
        COMM_CHK(c, c->comm->all_gather_v(c->d_bkt + blo, c->d_bkt, off.data(), cnt.data(), c->stream));
        size_t a = 0;
        for (u32 p = 0; p < G; p++) {
            off[p] = a;
            cnt[p] = shard[p] * sizeof(u64);
            a += cnt[p];
        }
        c->dinfo.bytes_sent[DISCO_X_INDEX_SHARDS] += (u64)(G - 1) * cnt[r];
        COMM_CHK(c, c->comm->all_gather_v(c->d_ent + base, c->d_ent, off.data(), cnt.data(), c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->dinfo.ms[DISCO_X_INDEX_SHARDS] += ms_since(t0);
    }
    ph_end(c, DISCO_PH_INDEX);
    c->phase = 2;
    return DISCO_OK;
}

/* ---- 2'. the probe against an index that stays partitioned: lookups to the owners, matching records back --------------------- */
/* (kernels and formats: disco_dist.h "the index that STAYS partitioned"). Fills the hit buffer and the per-read headers exactly as
 * probe_runs_kernel does for the replicated index: rows by read, headers by position in the processing order. Collective. */
static int dist_partitioned_probe(disco_ctx *c)
{
    DISCO_TRACE("dist_partitioned_probe");
    const u32 G = (u32)c->comm->world, r = (u32)c->comm->rank;
    const u64 lo = c->q_lo, nloc = c->q_hi - c->q_lo;
    DiscoView v = view(c);
    const int nf = v.k - v.m + 1;
    const bool have_runs = c->runs_lpr != 0 && c->runs_lo == lo && c->runs_n == nloc; /* (every rank decides alike: the shape of the JOB) */
    int logT = 0;
    while ((1ull << logT) < c->T) ++logT;
    /* 1. queries. With the minimizer runs of the index pass: one per run (count, then fill), reads with an unusable run list the long
     *    way; without them (windows other than 17 m-mers, reads beyond 256 bases) every read the long way */
    if (!c->d_list_n) CHK(dev_alloc(c, &c->d_list_n, 1));
    u32 *d_nslow = c->d_n_slow;
    auto make = [&](ulonglong2 *out, bool collect_slow) {
        const int grid = flat_grid(c, nloc * (u64)c->runs_lpr);
        u32 *slow = collect_slow ? (u32 *)c->d_slow_list : nullptr;
        const u32 cap = c->slow_cap * 2; /* (the list's u64 slots hold two read indices each) */
        if (c->runs_lpr == 16) hipLaunchKernelGGL(pq_make_kernel<16>, dim3(grid), dim3(256), 0, c->stream, v, (const u32 *)c->d_runs, lo, c->q_hi, r, out, c->d_list_n, slow, cap, d_nslow);
        else hipLaunchKernelGGL(pq_make_kernel<32>, dim3(grid), dim3(256), 0, c->stream, v, (const u32 *)c->d_runs, lo, c->q_hi, r, out, c->d_list_n, slow, cap, d_nslow);
    };
    u64 nq_fast = 0, n_slow = 0;
    const u32 *slow = nullptr; /* null: every read of the range */
    if (have_runs) {
        u32 ns = 0;
        for (int attempt = 0;; attempt++) { /* the list of reads without a usable run list: sized by a first try */
            const u32 want = attempt ? ns / 2 + 1024 : (u32)std::min<u64>(nloc, nloc / 128 + 1024);
            if (want > c->slow_cap) {
                dev_free(c, &c->d_slow_list, c->slow_cap);
                c->slow_cap = 0;
                CHK(dev_alloc(c, &c->d_slow_list, want));
                c->slow_cap = want;
            }
            HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
            HIPCHK(c, hipMemsetAsync(d_nslow, 0, sizeof(u32), c->stream));
            if (nloc) make(nullptr, true);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(&nq_fast, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipMemcpyAsync(&ns, d_nslow, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            if (ns <= c->slow_cap * 2 || attempt) break;
        }
        if (ns > c->slow_cap * 2) return fail(c, DISCO_E_CAPACITY, "dist_partitioned_probe: slow list could not be sized");
        n_slow = ns;
        slow = (const u32 *)c->d_slow_list;
    } else
        n_slow = nloc;
    /* the long way: queries per read, scan, fill behind the fast ones */
    u32 *slow_cnt = nullptr;
    u64 *slow_start = nullptr;
    u64 nq_slow = 0;
    int rc = DISCO_OK;
    auto slow_count = [&]() -> int {
        if (!n_slow) return DISCO_OK;
        CHK(dev_alloc(c, &slow_cnt, n_slow));
        CHK(dev_alloc(c, &slow_start, n_slow + 1));
        if (c->k > 64) hipLaunchKernelGGL(pq_slow_kernel<true>, dim3(flat_grid(c, n_slow, 64)), dim3(64), 0, c->stream, v, slow, n_slow, lo, r, slow_cnt, (const u64 *)nullptr, (ulonglong2 *)nullptr);
        else hipLaunchKernelGGL(pq_slow_kernel<false>, dim3(flat_grid(c, n_slow, 64)), dim3(64), 0, c->stream, v, slow, n_slow, lo, r, slow_cnt, (const u64 *)nullptr, (ulonglong2 *)nullptr);
        CHK((scan_exclusive<u32, u64>(c, slow_cnt, n_slow, slow_start, false, &nq_slow)));
        return DISCO_OK;
    };
    rc = slow_count();
    const u64 nqs = nq_fast + nq_slow;
    if (rc == DISCO_OK) rc = ensure_cap(c, &c->d_x16a, &c->x16a_cap, std::max<u64>(2 * nqs, 1)); /* flat list | partitioned by owner */
    if (rc == DISCO_OK) {
        HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
        if (have_runs && nloc) make(c->d_x16a, false);
        if (n_slow) {
            if (c->k > 64) hipLaunchKernelGGL(pq_slow_kernel<true>, dim3(flat_grid(c, n_slow, 64)), dim3(64), 0, c->stream, v, slow, n_slow, lo, r, slow_cnt, (const u64 *)slow_start, c->d_x16a + nq_fast);
            else hipLaunchKernelGGL(pq_slow_kernel<false>, dim3(flat_grid(c, n_slow, 64)), dim3(64), 0, c->stream, v, slow, n_slow, lo, r, slow_cnt, (const u64 *)slow_start, c->d_x16a + nq_fast);
        }
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(c, DISCO_E_HIP, "dist_partitioned_probe: query kernels failed");
    }
    dev_free(c, &slow_cnt, n_slow);
    dev_free(c, &slow_start, n_slow + 1);
    CHK(rc);
    c->slow_rows = have_runs ? n_slow : 0;
    /* 2. queries -> owners of their buckets */
    std::vector<u64> scnt, rcnt;
    RouteByBucket fb{logT, G};
    ulonglong2 *q_sorted = c->d_x16a + nqs;
    CHK(route_items(c, c->d_x16a, nqs, fb, q_sorted, scnt));
    CHK(exchange_counts(c, scnt, rcnt));
    const u64 nq_in = vsum(rcnt);
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(nq_in, 1)));
    CHK(a2a_items(c, DISCO_X_QUERIES, q_sorted, scnt, c->d_x16b, rcnt, sizeof(ulonglong2)));
    /* 3. the owner answers: hits per query, scan, fill */
    CHK(ensure_cap(c, &c->d_deg_tmp, &c->deg_tmp_cap, std::max<u64>(nq_in, 1)));
    CHK(ensure_cap(c, &c->d_pq_start, &c->pq_start_cap, nq_in + 1));
    const u32 *bkt = c->d_bkt;
    u64 n_hits_out = 0;
    if (nq_in) hipLaunchKernelGGL(pq_answer_kernel<false>, dim3(flat_grid(c, nq_in)), dim3(256), 0, c->stream, (const ulonglong2 *)c->d_x16b, nq_in, bkt, (const u64 *)c->d_ent, c->part_blo, c->per,
                                  nf, c->d_deg_tmp, (const u64 *)nullptr, (ulonglong2 *)nullptr);
    CHK((scan_exclusive<u32, u64>(c, c->d_deg_tmp, nq_in, c->d_pq_start, false, &n_hits_out)));
    /* hits: flat list | partitioned by requester — in the send buffer (the queries it held have been answered into counts) */
    CHK(ensure_cap_keep(c, &c->d_x16a, &c->x16a_cap, std::max<u64>(2 * n_hits_out, 1), 0));
    if (nq_in) hipLaunchKernelGGL(pq_answer_kernel<true>, dim3(flat_grid(c, nq_in)), dim3(256), 0, c->stream, (const ulonglong2 *)c->d_x16b, nq_in, bkt, (const u64 *)c->d_ent, c->part_blo, c->per,
                                  nf, c->d_deg_tmp, (const u64 *)c->d_pq_start, c->d_x16a);
    HIPCHK(c, hipGetLastError());
    /* 4. hits -> the reads' owners */
    ulonglong2 *h_sorted = c->d_x16a + n_hits_out;
    RouteByHitRank fh;
    CHK(route_items(c, c->d_x16a, n_hits_out, fh, h_sorted, scnt));
    CHK(exchange_counts(c, scnt, rcnt));
    const u64 n_hits = vsum(rcnt);
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(n_hits, 1)));
    CHK(a2a_items(c, DISCO_X_HITS, h_sorted, scnt, c->d_x16b, rcnt, sizeof(ulonglong2)));
    /* 5. rows of the hit buffer (exactly sized), headers by read and by position in the processing order */
    if (n_hits + 65536 > c->hits_cap) {
        dev_free(c, &c->d_hits, c->hits_cap);
        c->hits_cap = 0;
        const u64 want = n_hits + n_hits / 8 + 65536;
        CHK(dev_alloc(c, &c->d_hits, want));
        c->hits_cap = want;
    }
    CHK(ensure_cap(c, &c->d_deg_tmp, &c->deg_tmp_cap, std::max<u64>(2 * nloc, 1))); /* counts | cursors of the own reads */
    CHK(ensure_cap(c, &c->d_pq_start, &c->pq_start_cap, nloc + 1));
    HIPCHK(c, hipMemsetAsync(c->d_deg_tmp, 0, std::max<u64>(2 * nloc, 1) * sizeof(u32), c->stream));
    if (n_hits) hipLaunchKernelGGL(pq_rows_count_kernel, dim3(flat_grid(c, n_hits)), dim3(256), 0, c->stream, (const ulonglong2 *)c->d_x16b, n_hits, c->d_deg_tmp);
    u64 placed = 0;
    CHK((scan_exclusive<u32, u64>(c, c->d_deg_tmp, nloc, c->d_pq_start, false, &placed)));
    if (placed != n_hits) return fail(c, DISCO_E_STATE, "dist_partitioned_probe: %llu hits received, %llu placed", (unsigned long long)n_hits, (unsigned long long)placed);
    if (n_hits) hipLaunchKernelGGL(pq_rows_place_kernel, dim3(flat_grid(c, n_hits)), dim3(256), 0, c->stream, (const ulonglong2 *)c->d_x16b, n_hits, (const u64 *)c->d_pq_start, c->d_deg_tmp + nloc,
                                   c->d_hits);
    if (nloc) {
        hipLaunchKernelGGL(pq_rows_meta_kernel, dim3(flat_grid(c, nloc)), dim3(256), 0, c->stream, c->d_order_used, lo, nloc, (const u16 *)c->d_len, (const u64 *)c->d_pq_start,
                           (const u32 *)c->d_deg_tmp, c->d_row_start, c->d_row_cnt, c->d_meta_ord);
        CHK(zero_counter(c, CTR_MAX_ROW));
        hipLaunchKernelGGL(max_u32_kernel, dim3(flat_grid(c, nloc)), dim3(256), 0, c->stream, (const u32 *)c->d_deg_tmp, nloc, c->d_ctr + CTR_MAX_ROW);
    }
    HIPCHK(c, hipGetLastError());
    c->hits_used = n_hits;
    c->big_rows = 0;
    return DISCO_OK;
}

/* ---- 3. containment: smallest key wins across ranks, flags of the own range, bitmap to everybody -------------------- */
static int dist_mark_contained(disco_ctx *c)
{
    DISCO_TRACE("dist_mark_contained");
    const u32 G = (u32)c->comm->world;
    /* (the flags of an id range are fixed by the rank whose reads arrived in it, whoever processes them: best[] is a table by read id) */
    const u64 lo = c->home_lo, nloc = c->home_hi - c->home_lo, per = c->per;
    if (!c->d_contained) CHK(dev_alloc(c, &c->d_contained, c->n_alloc));
    if (!c->d_cbits) CHK(dev_alloc(c, &c->d_cbits, c->n_alloc / 64 + 1));
    const auto t0 = HClock::now();
    const u64 r = (u64)c->comm->rank;
    /* Who is contained decides everything that follows in the pass (edge selection, the second verify pass); by WHOM — the smallest key
     * over the ranks — only the row that is written at the end. So the pass exchanges bitmaps (every rank's "has a key" bits: n / 8
     * bytes per rank instead of 8 n) and the keys' reduce-scatter runs behind it, on the second communicator and stream: 1.0 ms of
     * link time at G = 8 / 50 M reads that nothing used to hide. DISCO_DIST_KEYS_ON_PATH=1 (every rank alike), one communicator: as before */
    if (c->comm_bulk && !getenv("DISCO_DIST_ONE_COMM") && !getenv("DISCO_DIST_KEYS_ON_PATH")) {
        const u64 words = c->n_alloc / 64;
        CHK(ensure_cap(c, &c->d_cb_all, &c->cb_all_cap, std::max<u64>(words * G, 1)));
        if (!c->ev_keys) {
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_keys, hipEventDisableTiming));
            HIPCHK(c, hipEventCreateWithFlags(&c->ev_keys_go, hipEventDisableTiming));
        }
        CHK(zero_counter(c, CTR_N_CONTAINED));
        ph_begin(c, DISCO_PH_CONTAIN);
        u64 *mine = c->d_cb_all + r * words;
        if (c->n_alloc) hipLaunchKernelGGL(has_key_bits_kernel, dim3(flat_grid(c, c->n_alloc)), dim3(256), 0, c->stream, (const u64 *)c->d_best, c->n_alloc, mine);
        HIPCHK(c, hipGetLastError());
        COMM_CHK(c, c->comm->all_gather(mine, c->d_cb_all, words * 8, c->stream));
        c->dinfo.bytes_sent[DISCO_X_CONTAIN] += (u64)(G - 1) * words * 8;
        HIPCHK(c, hipMemsetAsync(c->d_contained, 0, std::max<u64>(c->n_alloc, 1), c->stream));
        if (words) hipLaunchKernelGGL(or_bits_kernel, dim3(flat_grid(c, words)), dim3(256), 0, c->stream, (const u64 *)c->d_cb_all, G, words, c->d_cbits, r * per / 64, (r + 1) * per / 64, c->d_contained, c->d_ctr); /* (r per, not home_lo: a rank behind the last read has home_lo = n, in the middle of another rank's word) */
        HIPCHK(c, hipGetLastError());
        ph_end(c, DISCO_PH_CONTAIN);
        /* the keys, behind everything the stream has done so far (verify wrote them) */
        HIPCHK(c, hipEventRecord(c->ev_keys_go, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->bulk_stream, c->ev_keys_go, 0));
        if (c->comm_bulk->reduce_scatter_min_i64(c->d_best, per, c->bulk_stream) != DISCO_OK) return fail(c, DISCO_E_HIP, "containment keys: %s", c->comm_bulk->err.c_str());
        HIPCHK(c, hipEventRecord(c->ev_keys, c->bulk_stream));
        c->keys_pending = true;
        c->dinfo.bytes_sent[DISCO_X_CONTAIN_KEYS] += (u64)(G - 1) * per * 8;
        CHK(read_counters(c));
        ph_collect(c);
        c->dinfo.ms[DISCO_X_CONTAIN] += ms_since(t0);
        c->n_contained = c->h_ctr[CTR_N_CONTAINED]; /* home range: what disco_fetch_contained returns */
        c->phase = 4;
        return DISCO_OK;
    }
    COMM_CHK(c, c->comm->reduce_scatter_min_i64(c->d_best, per, c->stream));
    c->dinfo.bytes_sent[DISCO_X_CONTAIN] += (u64)(G - 1) * per * 8;
    CHK(zero_counter(c, CTR_N_CONTAINED));
    ph_begin(c, DISCO_PH_CONTAIN);
    HIPCHK(c, hipMemsetAsync(c->d_contained, 0, std::max<u64>(c->n_alloc, 1), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_cbits + r * per / 64, 0, per / 8, c->stream));
    if (nloc) hipLaunchKernelGGL(contain_flags_kernel, dim3(flat_grid(c, nloc)), dim3(256), 0, c->stream, c->d_best + lo, nloc, c->d_contained + lo, c->d_cbits + lo / 64, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    COMM_CHK(c, c->comm->all_gather(c->d_cbits + r * per / 64, c->d_cbits, per / 8, c->stream));
    c->dinfo.bytes_sent[DISCO_X_CONTAIN] += (u64)(G - 1) * per / 8;
    ph_end(c, DISCO_PH_CONTAIN);
    CHK(read_counters(c));
    ph_collect(c);
    c->dinfo.ms[DISCO_X_CONTAIN] += ms_since(t0);
    c->n_contained = c->h_ctr[CTR_N_CONTAINED]; /* own range: what disco_fetch_contained returns */
    c->phase = 4;
    return DISCO_OK;
}

/* ---- 5. neighbour rows on request ------------------------------------------------------------------------------------ */
/* room for `need` more entries of fetched rows behind the rank's own rows (and behind the nadj_used entries fetched so far) in the array
 * that holds them — the hit buffer (64 slots per read allotted, 36 used at 30 x: the tail is there) or the merged rows; *base = where they go */
static int adj_tail_reserve(disco_ctx *c, u64 need, u64 *base)
{
    const bool in_hits = c->d_adj == c->d_hits;
    const u64 used = (in_hits ? c->hits_used : c->adj_total) + c->nadj_used;
    /* test hook: the array counts as full — every call with something to place moves it (the path a probe that sizes the buffer exactly takes) */
    const bool tight = getenv("DISCO_TEST_TIGHT_TAIL") != nullptr && need != 0;
    auto move_to_larger = [&](u64 **p, u64 *cap) -> int { /* (ensure_cap_keep's steps, for an array that is large enough on paper) */
        u64 *q = nullptr;
        const u64 ncap = used + need + (used + need) / 8;
        CHK(dev_alloc(c, &q, ncap));
        if (used) HIPCHK(c, hipMemcpyAsync(q, *p, used * sizeof(u64), hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(c, p, *cap);
        *p = q;
        *cap = ncap;
        return DISCO_OK;
    };
    if (in_hits) {
        if (tight) CHK(move_to_larger(&c->d_hits, &c->hits_cap));
        else if (used + need > c->hits_cap) CHK(ensure_cap_keep(c, &c->d_hits, &c->hits_cap, used + need, used));
        c->d_adj = c->d_hits;
    } else {
        if (c->d_adj != c->d_adj_own) return fail(c, DISCO_E_STATE, "neighbour rows: the adjacency is in neither of the buffers that can grow");
        if (tight) CHK(move_to_larger(&c->d_adj_own, &c->adj_cap));
        else if (used + need > c->adj_cap) CHK(ensure_cap_keep(c, &c->d_adj_own, &c->adj_cap, used + need, used));
        c->d_adj = c->d_adj_own;
    }
    *base = used;
    return DISCO_OK;
}

/* ---- 5. neighbour rows on request (the exchange) ---------------------------------------------------------------------- */
/* one request round: the flat list of requested nodes in d_req_flat -> their rows behind the own rows (rows_place_kernel), their reference words set */
static int dist_fetch_rows(disco_ctx *c, u64 n_flat)
{
    DISCO_TRACE("dist_fetch_rows");
    const u32 G = (u32)c->comm->world;
    std::vector<u64> scnt, rcnt;
    CHK(ensure_cap(c, &c->d_req_s, &c->req_s_cap, std::max<u64>(n_flat, 1)));
    RouteByRowRequest f{c->per, c->loci ? c->d_otab : nullptr};
    CHK(route_items(c, c->d_req_flat, n_flat, f, c->d_req_s, scnt));
    CHK(exchange_counts(c, scnt, rcnt));
    const u64 nrq = vsum(rcnt);
    CHK(ensure_cap(c, &c->d_req_r, &c->req_r_cap, std::max<u64>(nrq, 1)));
    CHK(a2a_items(c, DISCO_X_ROW_REQUESTS, c->d_req_s, scnt, c->d_req_r, rcnt, sizeof(u32)));
    /* the owner's side: degree of every requested row, positions, entries */
    CHK(ensure_cap(c, &c->d_rdeg_s, &c->rdeg_s_cap, std::max<u64>(nrq, 1)));
    CHK(ensure_cap(c, &c->d_rpos, &c->rpos_cap, std::max<u64>(std::max(nrq, n_flat), 1) + 1));
    const int rgrid = (int)std::max<u64>(std::min<u64>((nrq + 3) / 4, (u64)c->n_cu * 32), 1);
    if (nrq) hipLaunchKernelGGL(tr_respond_deg_kernel, dim3(flat_grid(c, nrq)), dim3(256), 0, c->stream, c->d_req_r, nrq, c->d_adj_ref, c->d_rdeg_s);
    HIPCHK(c, hipGetLastError());
    u64 total_s = 0;
    CHK((scan_exclusive<u32, u64>(c, c->d_rdeg_s, nrq, c->d_rpos, true, &total_s)));
    std::vector<u64> ecnt_s((size_t)G, 0), ecnt_r;
    {   /* entries per requester = differences of the positions at the segment boundaries */
        std::vector<u64> bpos((size_t)G + 1, 0);
        u64 a = 0;
        for (u32 p = 0; p < G; p++) {
            if (nrq) HIPCHK(c, hipMemcpyAsync(&bpos[p], c->d_rpos + a, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            a += rcnt[p];
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bpos[G] = total_s;
        for (u32 p = 0; p < G; p++) ecnt_s[p] = bpos[p + 1] - bpos[p];
    }
    CHK(ensure_cap(c, &c->d_rdata_s, &c->rdata_s_cap, std::max<u64>(total_s, 1)));
    if (nrq) hipLaunchKernelGGL(tr_respond_kernel, dim3(rgrid), dim3(64), 0, c->stream, c->d_req_r, nrq, c->d_adj_ref, c->d_adj, c->d_rpos, c->d_rdata_s);
    HIPCHK(c, hipGetLastError());
    /* degrees back (same segmentation as the requests, reversed), then the entries. How many entries every owner sends follows from
     * the degrees themselves: the requester sums them per segment (positions at the segment boundaries) — no exchange of counts */
    CHK(ensure_cap(c, &c->d_rdeg_r, &c->rdeg_r_cap, std::max<u64>(n_flat, 1)));
    CHK(a2a_items(c, DISCO_X_ROW_REQUESTS, c->d_rdeg_s, rcnt, c->d_rdeg_r, scnt, sizeof(u32)));
    u64 total_r = 0;
    CHK((scan_exclusive<u32, u64>(c, c->d_rdeg_r, n_flat, c->d_rpos, true, &total_r))); /* (d_rpos: the owner's positions were consumed by tr_respond_kernel above) */
    ecnt_r.assign((size_t)G, 0);
    {
        std::vector<u64> bpos((size_t)G + 1, 0);
        u64 a = 0;
        for (u32 p = 0; p < G; p++) {
            if (n_flat) HIPCHK(c, hipMemcpyAsync(&bpos[p], c->d_rpos + a, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
            a += scnt[p];
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));
        bpos[G] = total_r;
        for (u32 p = 0; p < G; p++) ecnt_r[p] = bpos[p + 1] - bpos[p];
    }
    /* the rows that arrive (4-byte entries) ... */
    CHK(ensure_cap(c, &c->d_nadj32_own, &c->nadj_cap, std::max<u64>(total_r, 1)));
    CHK(a2a_items(c, DISCO_X_ROW_DATA, c->d_rdata_s, ecnt_s, c->d_nadj32_own, ecnt_r, sizeof(u32)));
    /* ... go behind the own rows, as 8-byte entries under the nodes' reference words */
    u64 base = 0;
    CHK(adj_tail_reserve(c, total_r, &base));
    if (n_flat) hipLaunchKernelGGL(rows_place_kernel, dim3((int)std::max<u64>(std::min<u64>((n_flat + 3) / 4, (u64)c->n_cu * 32), 1)), dim3(64), 0, c->stream, c->d_req_s, n_flat, c->d_rdeg_r, c->d_rpos,
                                   (const u32 *)c->d_nadj32_own, base, c->d_adj, c->d_adj_ref);
    HIPCHK(c, hipGetLastError());
    c->nadj_used += total_r;
    return DISCO_OK;
}

static int dist_transitive_mark(disco_ctx *c)
{
    DISCO_TRACE("dist_transitive_mark");
    const u64 nloc = c->q_hi - c->q_lo; /* own nodes (an id range, or the positions of the own list) */
    const OwnSet own = own_set(c);
    /* the reference words of other ranks' nodes are 0 = "not fetched" (edge selection cleared the table and wrote the own nodes' words; a
     * merge rebuilt it from degrees: 0 for every node of another rank); one bit per node: somebody on this rank has asked for its row */
    CHK(ensure_cap(c, &c->d_asked, &c->asked_cap, c->n / 32 + 2));
    HIPCHK(c, hipMemsetAsync(c->d_asked, 0, (c->n / 32 + 2) * sizeof(u32), c->stream));
    c->nadj_used = 0; /* fetched rows: behind the own rows (adj_tail_reserve) */
    /* round 1: slot 0 and the first slot on the other side of every register-resident node */
    if (!c->d_list_n) CHK(dev_alloc(c, &c->d_list_n, 1));
    CHK(ensure_cap(c, &c->d_req_flat, &c->req_flat_cap, 2 * nloc + 64));
    HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
    CHK(zero_counter(c, CTR_OVERFLOW));
    ph_begin(c, DISCO_PH_CSR);
    if (nloc) hipLaunchKernelGGL(tr_request_first_kernel, dim3((int)std::max<u64>(std::min<u64>((nloc + 63) / 64, (u64)c->n_cu * 32), 1)), dim3(64), 0, c->stream, (const u64 *)c->d_adj, own, (const u64 *)c->d_adj_ref, c->d_asked, c->d_req_flat, c->d_list_n, c->req_flat_cap, c->d_ctr);
    ph_end(c, DISCO_PH_CSR);
    HIPCHK(c, hipGetLastError());
    u64 n_flat = 0;
    HIPCHK(c, hipMemcpyAsync(&n_flat, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    CHK(read_counters(c));
    if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_CAPACITY, "row requests: list overflow");
    CHK(dist_fetch_rows(c, n_flat));
    c->dinfo.tr_rounds = 1;

    /* marking of the own nodes; nodes beyond the register path or short of a row land in big_list */
    const u64 need_big = nloc + 1024;
    if (need_big > c->big_cap) {
        dev_free(c, &c->d_big_list, c->big_cap);
        dev_free(c, &c->d_big_cnt, c->big_cap);
        c->big_cap = 0;
        CHK(dev_alloc(c, &c->d_big_list, need_big));
        CHK(dev_alloc(c, &c->d_big_cnt, need_big));
        c->big_cap = (u32)need_big;
    }
    HIPCHK(c, hipMemsetAsync(c->d_n_big, 0, sizeof(u32), c->stream));
    TrArgs a;
    a.v = view(c);
    a.order = c->loci ? c->d_order_own : nullptr; /* (ranks own loci: the own nodes are the list, in the processing order of probe / verify / selection) */
    a.ref = c->d_adj_ref;
    a.adj = c->d_adj;
    a.big_list = c->d_big_list;
    a.n_big = c->d_n_big;
    a.big_cap = c->big_cap;
    a.scratch = nullptr;
    a.hcap = 0;
    c->use_half = true;
    if (!c->d_half) CHK(dev_alloc(c, &c->d_half, c->n * HALF_CAP));
    if (!c->d_hcnt) CHK(dev_alloc(c, &c->d_hcnt, c->n));
    HIPCHK(c, hipMemsetAsync(c->d_hcnt, 0, std::max<u64>(c->n, 1) * sizeof(u32), c->stream));
    if (!c->d_wide) {
        c->wide_cap = (u32)std::min<u64>(c->n, c->n / 32 + 4096);
        CHK(dev_alloc(c, &c->d_wide, c->wide_cap));
        CHK(dev_alloc(c, &c->d_n_wide, 1));
    }
    HIPCHK(c, hipMemsetAsync(c->d_n_wide, 0, sizeof(u32), c->stream));
    a.half = c->d_half;
    a.hcnt = c->d_hcnt;
    a.wide_list = c->d_wide;
    a.n_wide = c->d_n_wide;
    a.wide_cap = c->wide_cap;
    /* the transitive flags go into the rows of nodes with more than HALF_CAP survivors only, as on one GPU: everybody who judges an edge
     * — the local emission, the survivor push and its receiver — reads a narrow node's survivor LIST and a wide node's row, never a narrow
     * node's row (rounds 1-4 wrote every flag here: a quarter of this kernel's memory requests, left over from the flag exchange the
     * push replaced). DISCO_DIST_ALL_FLAGS=1: as before */
    a.all_flags = getenv("DISCO_DIST_ALL_FLAGS") ? 1u : 0u;
    ph_begin(c, DISCO_PH_TRMARK);
    /* (every node beyond the register path waits for the request-all round here, whatever the LDS arrays could hold: the small variant) */
    if (nloc && !a.all_flags) hipLaunchKernelGGL((transitive_mark_kernel<false, true, TR_CAP_SMALL, true>), dim3(wq_grid(c, transitive_mark_kernel<false, true, TR_CAP_SMALL, true>, nloc, "DISCO_TR_WAVES")), dim3(64), 0, c->stream, a);
    else if (nloc) hipLaunchKernelGGL((transitive_mark_kernel<false, true, TR_CAP_SMALL>), dim3(wq_grid(c, transitive_mark_kernel<false, true, TR_CAP_SMALL>, nloc, "DISCO_TR_WAVES")), dim3(64), 0, c->stream, a);
    ph_end(c, DISCO_PH_TRMARK);
    HIPCHK(c, hipGetLastError());
    u32 n_big = 0;
    HIPCHK(c, hipMemcpyAsync(&n_big, c->d_n_big, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    CHK(read_counters(c));
    ph_collect(c);
    if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_CAPACITY, "transitive marking: deferred-node list overflow (%u nodes)", n_big);
    u64 any = n_big;
    CHK(host_reduce(c, &any, 1));
    c->dinfo.tr_deferred = any;
    if (any) { /* round 2 (collective): the listed nodes ask for every row they lack, then take the generic path */
        c->dinfo.tr_rounds = 2;
        u64 bound = 0, maxd = 0;
        HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
        if (n_big) hipLaunchKernelGGL(list_degree_sum_kernel, dim3(flat_grid(c, n_big)), dim3(256), 0, c->stream, c->d_big_list, (u64)n_big, c->d_adj_ref, c->d_list_n);
        HIPCHK(c, hipMemcpyAsync(&bound, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        CHK(ensure_cap(c, &c->d_req_flat, &c->req_flat_cap, bound + 64));
        HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
        if (n_big) hipLaunchKernelGGL(tr_request_all_kernel, dim3((int)std::min<u64>(n_big, (u64)c->n_cu * 32)), dim3(64), 0, c->stream, c->d_big_list, (u64)n_big, (const u64 *)c->d_adj, own, (const u64 *)c->d_adj_ref, c->d_asked, c->d_req_flat, c->d_list_n, c->req_flat_cap, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&n_flat, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        CHK(read_counters(c));
        if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_CAPACITY, "row requests (round 2): list overflow");
        CHK(dist_fetch_rows(c, n_flat));
        if (n_big) {
            CHK(zero_counter(c, CTR_MAX_DEG));
            hipLaunchKernelGGL(list_max_degree_kernel, dim3(flat_grid(c, n_big)), dim3(256), 0, c->stream, c->d_big_list, (u64)n_big, c->d_adj_ref, c->d_ctr + CTR_MAX_DEG);
            CHK(read_counters(c));
            maxd = c->h_ctr[CTR_MAX_DEG];
            u64 hcap = 64;
            while (hcap < 2 * maxd) hcap <<= 1;
            const int g2 = (int)std::min<u64>(n_big, (u64)c->n_cu * 8);
            const u64 perb = hcap * 8 + hcap * 4 + hcap;
            u8 *scratch = nullptr;
            CHK(dev_alloc(c, &scratch, (u64)g2 * perb));
            a.scratch = (u64 *)scratch;
            a.hcap = hcap;
            a.adj = c->d_adj; /* the array may have moved (adj_tail_reserve) */
            HIPCHK(c, hipMemsetAsync(c->d_wq, 0, sizeof(u64) * WQ_WORDS, c->stream));
            hipLaunchKernelGGL((transitive_mark_kernel<true, true>), dim3(g2), dim3(64), 0, c->stream, a);
            hipError_t e = hipGetLastError();
            int rc = read_counters(c);
            dev_free(c, &scratch, (u64)g2 * perb);
            if (e != hipSuccess) return fail(c, DISCO_E_HIP, "transitive_mark_kernel (second round): %s", hipGetErrorString(e));
            CHK(rc);
            if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_STATE, "transitive marking: a row was still missing after the request-all round");
        }
    }
    c->n_wide = 0;
    HIPCHK(c, hipMemcpy(&c->n_wide, c->d_n_wide, sizeof(u32), hipMemcpyDeviceToHost));
    c->flags_pending = false;
    c->phase = 7;
    return DISCO_OK;
}

/* ---- 6. surviving half-edges to the owner of the smaller endpoint ---------------------------------------------------- */
static int dist_push_survivors(disco_ctx *c)
{
    DISCO_TRACE("dist_push_survivors");
    const u64 nloc = c->q_hi - c->q_lo;
    const OwnSet own = own_set(c);
    u64 n_items = 0;
    /* the items (survivors whose smaller endpoint is another rank's: 400 000 per rank at 8 x 6.25 M nodes) are written in ONE pass into
     * the room the exchange buffer has anyway (the index records went through it: two per own read) — rounds 1-5 counted them first, a
     * second walk over every own node's survivors and a host round trip. A list that does not fit (the counter of lost items says so)
     * is counted and written again, as before */
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(nloc / 4, 1u << 16)));
    for (int attempt = 0; attempt < 2; attempt++) {
        u64 cap = attempt ? n_items : c->x16b_cap;
        if (!attempt && getenv("DISCO_TEST_TIGHT_PUSH")) cap = std::min<u64>(cap, 1); /* test hook: the first pass loses items, the second one runs */
        HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
        CHK(zero_counter(c, CTR_OVERFLOW));
        if (nloc) hipLaunchKernelGGL(emit_push_kernel<true>, dim3(flat_grid(c, nloc, 64)), dim3(64), 0, c->stream, c->d_adj_ref, c->d_adj, c->d_half, c->d_hcnt, c->d_len, own, c->d_x16b, c->d_list_n, cap, c->d_ctr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&n_items, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        CHK(read_counters(c)); /* (synchronises: n_items is there) */
        if (!c->h_ctr[CTR_OVERFLOW]) break;
        if (attempt) return fail(c, DISCO_E_STATE, "survivor push: the second pass produced more items than the first counted");
        CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(n_items, 1))); /* (the counter counted every item, lost or not) */
    }
    CHK(ensure_cap(c, &c->d_x16a, &c->x16a_cap, std::max<u64>(n_items, 1)));
    std::vector<u64> scnt, rcnt;
    RouteByNode f{c->per, c->loci ? c->d_otab : nullptr};
    CHK(route_items(c, c->d_x16b, n_items, f, c->d_x16a, scnt));
    CHK(exchange_counts(c, scnt, rcnt));
    const u64 nr = vsum(rcnt);
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(nr, 1)));
    CHK(a2a_items(c, DISCO_X_PUSH, c->d_x16a, scnt, c->d_x16b, rcnt, sizeof(ulonglong2)));
    c->d_push_r = c->d_x16b;
    c->n_push_r = nr;
    return DISCO_OK;
}

/* ---- some reads dropped a hit (real data: repeats): complete the lists across ranks, then carry on in the regular regime ------ */
/* Collective. *done = false: too many twins are missing somewhere (or a rank's rows cannot grow in place) — nothing was changed
 * on any rank and the caller gathers the adjacency instead. */
static int dist_complete_twins(disco_ctx *c, bool *done, u64 *asym_total)
{
    DISCO_TRACE("dist_complete_twins");
    const u32 G = (u32)c->comm->world, r = (u32)c->comm->rank;
    const u64 lo = c->q_lo, hi = c->q_hi, nloc = hi - lo, per = c->per;
    const OwnSet own = own_set(c);
    *done = false;
    const auto t0 = HClock::now();
    if (c->loci) {
        /* the reads that dropped something are scattered over the id space (a rank set the bits of ITS reads): their ids travel as
         * lists, every rank sets everybody's bits */
        const u64 n_words = (c->n + 63) / 64;
        u64 mine = 0;
        HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
        hipLaunchKernelGGL(bits_to_list_kernel, dim3(flat_grid(c, n_words)), dim3(256), 0, c->stream, (const u64 *)c->d_dropbits, n_words, (u32 *)nullptr, c->d_list_n, (u64)0, c->d_ctr);
        HIPCHK(c, hipMemcpyAsync(&mine, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        std::vector<u64> cnts((size_t)G);
        COMM_CHK(c, c->comm->host_all_gather((const unsigned long long *)&mine, 1, (unsigned long long *)cnts.data(), c->stream)); /* (synchronises: `mine` is there) */
        std::vector<size_t> off((size_t)G), cnt((size_t)G);
        size_t tot = 0;
        for (u32 p2 = 0; p2 < G; p2++) {
            off[p2] = tot * sizeof(u32);
            cnt[p2] = (size_t)cnts[p2] * sizeof(u32);
            tot += (size_t)cnts[p2];
        }
        CHK(ensure_cap(c, &c->d_req_flat, &c->req_flat_cap, std::max<u64>(tot, 1) + 64));
        u32 *my_list = c->d_req_flat + off[r] / sizeof(u32);
        HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
        CHK(zero_counter(c, CTR_OVERFLOW));
        hipLaunchKernelGGL(bits_to_list_kernel, dim3(flat_grid(c, n_words)), dim3(256), 0, c->stream, (const u64 *)c->d_dropbits, n_words, my_list, c->d_list_n, (u64)cnts[r], c->d_ctr);
        HIPCHK(c, hipGetLastError());
        COMM_CHK(c, c->comm->all_gather_v(my_list, c->d_req_flat, off.data(), cnt.data(), c->stream));
        if (tot) hipLaunchKernelGGL(list_to_bits_kernel, dim3(flat_grid(c, tot)), dim3(256), 0, c->stream, (const u32 *)c->d_req_flat, (u64)tot, c->d_dropbits);
        HIPCHK(c, hipGetLastError());
        c->dinfo.bytes_sent[DISCO_X_TWINS] += (u64)(G - 1) * cnt[r];
    } else {
        COMM_CHK(c, c->comm->all_gather(c->d_dropbits + (u64)r * per / 64, c->d_dropbits, per / 8, c->stream));
        c->dinfo.bytes_sent[DISCO_X_TWINS] += (u64)(G - 1) * (per / 8);
    }
    c->drop_lo = 0;
    c->drop_hi = c->n;
    /* pairs inside the rank: finds of the own nodes into own nodes that dropped something */
    CHK(twin_check(c, c->loci ? 0 : lo, c->loci ? c->n : hi));
    u64 asym = c->h_ctr[CTR_ASYM];
    /* pairs across ranks: {w, twin} to the owner of w */
    u64 n_items = 0;
    HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
    if (nloc) hipLaunchKernelGGL(twin_push_kernel<false>, dim3(flat_grid(c, nloc * 64)), dim3(256), 0, c->stream, c->d_adj_ref, c->d_adj, c->d_len, own, c->d_dropbits,
                                 (ulonglong2 *)nullptr, c->d_list_n, (u64)0, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&n_items, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(n_items, 1)));
    CHK(ensure_cap(c, &c->d_x16a, &c->x16a_cap, std::max<u64>(n_items, 1)));
    HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
    CHK(zero_counter(c, CTR_OVERFLOW));
    if (nloc) hipLaunchKernelGGL(twin_push_kernel<true>, dim3(flat_grid(c, nloc * 64)), dim3(256), 0, c->stream, c->d_adj_ref, c->d_adj, c->d_len, own, c->d_dropbits,
                                 c->d_x16b, c->d_list_n, n_items, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    CHK(read_counters(c));
    if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_STATE, "twin push: the fill pass produced more items than the count pass");
    std::vector<u64> scnt, rcnt;
    RouteByNode f{per, c->loci ? c->d_otab : nullptr};
    CHK(route_items(c, c->d_x16b, n_items, f, c->d_x16a, scnt));
    CHK(exchange_counts(c, scnt, rcnt));
    const u64 nr = vsum(rcnt);
    CHK(ensure_cap(c, &c->d_x16b, &c->x16b_cap, std::max<u64>(nr, 1)));
    CHK(a2a_items(c, DISCO_X_TWINS, c->d_x16a, scnt, c->d_x16b, rcnt, sizeof(ulonglong2)));
    /* room for every received item to turn out missing, behind the extras of the local check */
    const u64 want = (u64)c->n_extra + nr + 1;
    if (want > 0xFFFFFFFFull) return fail(c, DISCO_E_CAPACITY, "twin completion: more than 2^32 extras on one rank");
    if (want > c->extra_cap) {
        u64 cap_n = c->extra_cap, cap_k = c->extra_cap;
        CHK(ensure_cap_keep(c, &c->d_extra_node, &cap_n, want, c->n_extra));
        CHK(ensure_cap_keep(c, &c->d_extra_key, &cap_k, want, c->n_extra));
        if (cap_n != cap_k) return fail(c, DISCO_E_STATE, "twin completion: extras arrays out of step");
        c->extra_cap = (u32)std::min<u64>(cap_n, 0xFFFFFFFFull);
    }
    CHK(zero_counter(c, CTR_ASYM));
    CHK(zero_counter(c, CTR_OVERFLOW));
    if (nr) hipLaunchKernelGGL(twin_recv_kernel, dim3(flat_grid(c, nr)), dim3(256), 0, c->stream, (const ulonglong2 *)c->d_x16b, nr, c->d_adj_ref, c->d_adj, c->d_extra_node,
                               c->d_extra_key, c->d_extra_cnt, c->d_n_extra, c->extra_cap, c->d_ctr);
    HIPCHK(c, hipGetLastError());
    u32 ne = 0;
    HIPCHK(c, hipMemcpyAsync(&ne, c->d_n_extra, sizeof(u32), hipMemcpyDeviceToHost, c->stream));
    CHK(read_counters(c));
    if (c->h_ctr[CTR_OVERFLOW]) return fail(c, DISCO_E_STATE, "twin completion: extras list overflow");
    asym += c->h_ctr[CTR_ASYM];
    c->n_extra = ne;
    /* can every rank grow its few rows in place? (merge_extras' sparse path; otherwise nothing has been changed yet) */
    u64 need = 0;
    if (ne) {
        HIPCHK(c, hipMemsetAsync(c->d_bump, 0, sizeof(u64), c->stream));
        hipLaunchKernelGGL(merge_need_kernel, dim3(flat_grid(c, c->n)), dim3(256), 0, c->stream, c->d_adj_ref, c->d_extra_cnt, c->n, c->d_bump);
        HIPCHK(c, hipMemcpyAsync(&need, c->d_bump, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    const bool fits = c->d_adj == c->d_hits && ne <= 16384 && c->hits_used + need <= c->hits_cap && !getenv("DISCO_MERGE_REBUILD");
    u64 v[2] = {fits ? 0ull : 1ull, asym};
    CHK(host_reduce(c, v, 2));
    c->dinfo.ms[DISCO_X_TWINS] += ms_since(t0);
    *asym_total = v[1];
    if (v[0]) {
        c->n_extra = 0; /* the gathered adjacency is checked again as a whole */
        return DISCO_OK;
    }
    CHK(merge_extras(c));
    *done = true;
    return DISCO_OK;
}

/* ---- order-dependent regime: everybody gets the whole adjacency and finishes the pass on its own --------------------- */
static int dist_irregular(disco_ctx *c, const std::vector<u64> &adj_totals)
{
    DISCO_TRACE("dist_irregular");
    const u32 G = (u32)c->comm->world, r = (u32)c->comm->rank;
    const u64 lo = c->q_lo, hi = c->q_hi, nloc = hi - lo, per = c->per;
    u32 *deg_all = nullptr;
    u64 *rows_all = nullptr;
    const u64 total = vsum(adj_totals);
    CHK(dev_alloc(c, &deg_all, (u64)G * per));
    CHK(dev_alloc(c, &rows_all, std::max<u64>(total, 1)));
    u64 base = 0;
    for (u32 p = 0; p < r; p++) base += adj_totals[p];
    int rc = DISCO_OK;
    const auto t0 = HClock::now();
    do {
        if ((rc = hipMemsetAsync(deg_all + (u64)r * per, 0, per * sizeof(u32), c->stream) == hipSuccess ? DISCO_OK : DISCO_E_HIP) != DISCO_OK) break;
        if ((rc = disco_export_adjacency(c, deg_all + (u64)r * per, rows_all + base)) != DISCO_OK) break;
        if ((rc = c->comm->all_gather(deg_all + (u64)r * per, deg_all, per * sizeof(u32), c->stream)) != DISCO_OK) break;
        std::vector<size_t> off((size_t)G), cnt((size_t)G);
        size_t a = 0;
        for (u32 p = 0; p < G; p++) {
            off[p] = a;
            cnt[p] = adj_totals[p] * sizeof(u64);
            a += cnt[p];
            if (p != r) c->dinfo.bytes_sent[DISCO_X_ADJACENCY] += cnt[r] + per * sizeof(u32);
        }
        if ((rc = c->comm->all_gather_v(rows_all + base, rows_all, off.data(), cnt.data(), c->stream)) != DISCO_OK) break;
        if (hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = DISCO_E_HIP;
            break;
        }
        c->dinfo.ms[DISCO_X_ADJACENCY] += ms_since(t0);
        (void)nloc;
        if ((rc = disco_import_adjacency(c, deg_all, rows_all, total)) != DISCO_OK) break;
        /* which reads dropped a hit (ranges are multiples of 64 reads: whole bitmap words): the twin search below is limited to them */
        if ((rc = c->comm->all_gather(c->d_dropbits + (u64)r * per / 64, c->d_dropbits, per / 8, c->stream)) != DISCO_OK) break;
        c->dinfo.bytes_sent[DISCO_X_ADJACENCY] += (u64)(G - 1) * (per / 8);
        c->drop_lo = 0;
        c->drop_hi = c->n;
    } while (0);
    dev_free(c, &deg_all, (u64)G * per);
    dev_free(c, &rows_all, std::max<u64>(total, 1));
    if (rc != DISCO_OK) return c->err.empty() ? fail(c, rc, "adjacency exchange: %s", c->comm->err.c_str()) : rc;
    CHK(disco_symmetrize(c, 1, nullptr));
    CHK(merge_extras(c));
    c->q_lo = 0;
    c->q_hi = c->n;
    rc = disco_transitive_mark(c);
    c->q_lo = lo;
    c->q_hi = hi;
    CHK(rc);
    c->half_complete = true; /* every node was marked here */
    c->dist_active = false;  /* nothing is pushed: every pair is judged locally */
    c->n_push_r = 0;
    return DISCO_OK;
}

extern "C" {

int disco_comm_unique_id(void *out, size_t cap)
{
    if (!out || cap < DISCO_UNIQUE_ID_BYTES) return DISCO_E_ARG;
    static_assert(sizeof(ncclUniqueId) == DISCO_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return fail(nullptr, DISCO_E_HIP, "ncclGetUniqueId failed");
    memcpy(out, &id, sizeof id);
    return DISCO_OK;
}

static int comm_attach(disco_ctx *c, DiscoComm *cm, DiscoComm *bulk)
{
    delete c->comm;
    delete c->comm_bulk;
    c->comm = cm;
    c->comm_bulk = bulk;
    if (hipSetDevice(c->device) != hipSuccess) return fail(c, DISCO_E_HIP, "hipSetDevice failed");
    if (!c->bulk_stream) HIPCHK(c, hipStreamCreateWithFlags(&c->bulk_stream, hipStreamNonBlocking));
    if (!c->ev_bulk) HIPCHK(c, hipEventCreateWithFlags(&c->ev_bulk, hipEventDisableTiming));
    return DISCO_OK;
}

int disco_comm_init(disco_ctx *c, const void *unique_id, int nranks, int rank)
{
    if (!c || !unique_id || nranks < 1 || nranks > DIST_MAX_WORLD || rank < 0 || rank >= nranks) return c ? fail(c, DISCO_E_ARG, "disco_comm_init: bad argument") : DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const bool one_comm = getenv("DISCO_DIST_ONE_COMM") != nullptr; /* (every rank alike) */
    RcclComm *cm = new (std::nothrow) RcclComm();
    RcclComm *bulk = one_comm ? nullptr : new (std::nothrow) RcclComm();
    if (!cm || (!bulk && !one_comm)) {
        delete cm;
        delete bulk;
        return fail(c, DISCO_E_NOMEM, "disco_comm_init: out of host memory");
    }
    /* Order of the steps (round 6): everything that can fail on ONE rank alone — allocations — happens before that rank enters
     * a collective; every decision about the second communicator is taken from values ALL ranks hold (a broadcast, then an all-gather of
     * the ranks' own status), so the ranks keep or drop it together; no rank is ever left waiting inside a collective its peer skipped. */
    int rc = cm->prepare(nranks, rank);                         /* local */
    int bulk_ready = (bulk && bulk->prepare(nranks, rank) == DISCO_OK) ? 1 : 0; /* local; a failure only costs the second communicator */
    if (rc == DISCO_OK) rc = cm->join(unique_id);               /* collective (the launcher's rendezvous: MPI_Init's role) */
    if (rc == DISCO_OK && !one_comm) { /* the id of the second communicator travels over the first one */
        struct IdMsg {
            ncclUniqueId id;
            unsigned long long ok;
        } msg;
        static_assert(sizeof(IdMsg) <= RcclComm::SMALL_VALUES * 8, "fits the staging of host_all_gather");
        memset(&msg, 0, sizeof msg);
        if (rank == 0) msg.ok = ncclGetUniqueId(&msg.id) == ncclSuccess ? 1ull : 0ull;
        void *d_id = cm->d_small;
        memcpy(cm->h_small, &msg, sizeof msg);
        if (hipMemcpyAsync(d_id, cm->h_small, sizeof msg, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
            ncclBroadcast(d_id, d_id, sizeof msg, ncclInt8, 0, cm->comm, c->stream) != ncclSuccess ||
            hipMemcpyAsync(cm->h_small, d_id, sizeof msg, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = DISCO_E_HIP; /* the first communicator itself does not work: nothing to fall back to */
            cm->err = "broadcast of the second communicator's id failed";
        } else {
            memcpy(&msg, cm->h_small, sizeof msg);
            /* is everybody able to join it? (rank 0 has an id, every rank its staging) — one all-gather of one word */
            unsigned long long mine = (msg.ok && bulk_ready) ? 1ull : 0ull, all[DIST_MAX_WORLD];
            rc = cm->host_all_gather(&mine, 1, all, c->stream);
            bool everybody = rc == DISCO_OK;
            for (int p = 0; everybody && p < nranks; p++) everybody = all[p] != 0ull;
            if (rc == DISCO_OK && everybody) {
                const int rb = bulk->join(&msg.id); /* collective */
                /* ... and did everybody get in? Keep it or drop it TOGETHER */
                mine = rb == DISCO_OK ? 1ull : 0ull;
                rc = cm->host_all_gather(&mine, 1, all, c->stream);
                for (int p = 0; rc == DISCO_OK && everybody && p < nranks; p++) everybody = all[p] != 0ull;
            }
            if (rc == DISCO_OK && !everybody) {
                if (rank == 0) fprintf(stderr, "[disco] no second communicator on every rank: the reads are gathered on the first one\n");
                delete bulk;
                bulk = nullptr;
            }
        }
    }
    if (one_comm) {
        delete bulk;
        bulk = nullptr;
    }
    if (rc == DISCO_OK && bulk) { /* abort() of either takes both down */
        cm->sibling = bulk;
        bulk->sibling = cm;
    }
    if (rc != DISCO_OK) {
        fail(c, rc, "disco_comm_init: %s %s", cm->err.c_str(), bulk ? bulk->err.c_str() : "");
        delete cm;
        delete bulk;
        return rc;
    }
    return comm_attach(c, cm, bulk);
}

int disco_comm_init_local(disco_ctx *const *ctxs, int nranks)
{
    if (!ctxs || nranks < 1 || nranks > DIST_MAX_WORLD) return DISCO_E_ARG;
    for (int r = 0; r < nranks; r++)
        if (!ctxs[r]) return DISCO_E_ARG;
    auto grp = std::make_shared<LoopGroup>(nranks), grp2 = std::make_shared<LoopGroup>(nranks);
    for (int r = 0; r < nranks; r++) {
        const int rc = comm_attach(ctxs[r], new LoopComm(grp, r), new LoopComm(grp2, r));
        if (rc != DISCO_OK) return rc;
    }
    return DISCO_OK;
}

int disco_probe_run_words(const disco_ctx *c) { return c ? c->runs_lpr : 0; }
int disco_comm_rank(const disco_ctx *c) { return (c && c->comm) ? c->comm->rank : 0; }
int disco_comm_world(const disco_ctx *c) { return (c && c->comm) ? c->comm->world : 1; }
const char *disco_comm_kind(const disco_ctx *c) { return (c && c->comm) ? c->comm->kind() : "none"; }

int disco_dist_range(const disco_ctx *c, uint64_t n_total, uint64_t *lo, uint64_t *hi)
{
    if (!c || !lo || !hi) return DISCO_E_ARG;
    u64 per, l, h;
    dist_range(c, n_total, &per, &l, &h);
    *lo = l;
    *hi = h;
    return DISCO_OK;
}

/* table of world * per rows; the own range is filled by the caller / generator, the rest by the all-gather of the pass */
static int dist_set_reads(disco_ctx *c, u64 n_total, uint32_t dstride)
{
    if (!c->comm) return fail(c, DISCO_E_STATE, "no communicator: call disco_comm_init / disco_comm_init_local first");
    CHK(set_reads_common(c, n_total, dstride));
    u64 per, lo, hi;
    dist_range(c, n_total, &per, &lo, &hi);
    c->per = per;
    c->n_alloc = per * (u64)c->comm->world;
    c->q_lo = c->home_lo = lo;
    c->q_hi = c->home_hi = hi;
    CHK(dev_alloc(c, &c->d_reads, c->n_alloc * (u64)dstride));
    CHK(dev_alloc(c, &c->d_len, c->n_alloc));
    /* unused words of a row are zero (disco_device.h): the other ranks' rows arrive at their used words only */
    HIPCHK(c, hipMemsetAsync(c->d_reads, 0, c->n_alloc * (u64)dstride * 8, c->stream));
    c->reads_owned = true;
    c->dist_reads = true;
    return DISCO_OK;
}

static int dist_validate(disco_ctx *c)
{
    const u64 nloc = c->q_hi - c->q_lo;
    CHK(zero_counter(c, CTR_BAD_LEN));
    CHK(zero_counter(c, CTR_MAX_LEN));
    CHK(zero_counter(c, CTR_MIN_LEN));
    if (nloc) hipLaunchKernelGGL(validate_len_kernel, dim3(flat_grid(c, nloc)), dim3(256), 0, c->stream, c->d_len + c->q_lo, nloc, c->S, (int)c->prm.min_overlap, c->d_ctr);
    CHK(read_counters(c));
    /* the job's longest / shortest read (the shortest travels complemented, so both are a maximum over the ranks): every rank
     * then takes the same two-pass decision in disco_probe whatever its own reads or its context's history look like */
    u64 bad = c->h_ctr[CTR_BAD_LEN], ext[2] = {c->h_ctr[CTR_MAX_LEN], c->h_ctr[CTR_MIN_LEN]};
    CHK(host_reduce(c, &bad, 1));
    CHK(host_reduce(c, ext, 2, true));
    if (bad) return fail(c, DISCO_E_ARG, "%llu reads have a length outside (min_overlap=%u, min(32767, 32*stride)]", (unsigned long long)bad, c->prm.min_overlap);
    c->max_len = (u32)ext[0];
    c->min_len = 0xFFFFu - (u32)ext[1];
    /* the k-mer probes of the home range (disco_dist_info.probes): a property of the reads, summed once here instead of in every pass */
    if (!c->d_list_n) CHK(dev_alloc(c, &c->d_list_n, 1));
    HIPCHK(c, hipMemsetAsync(c->d_list_n, 0, sizeof(u64), c->stream));
    if (nloc) hipLaunchKernelGGL(probes_sum_kernel, dim3(flat_grid(c, nloc)), dim3(256), 0, c->stream, c->d_len, c->q_lo, c->q_hi, (u32)c->k, c->d_list_n);
    HIPCHK(c, hipMemcpyAsync(&c->home_probes, c->d_list_n, sizeof(u64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    /* round 6: the job's long reads and the longest of the others (two classes of rows under a communicator: every rank decides alike) */
    c->job_n_long = 0;
    c->job_short_max = c->max_len;
    if (c->max_len > (u32)DISCO_SHORT_MAX) {
        u64 *d_ls = nullptr, hls[2] = {0, 0};
        CHK(dev_alloc(c, &d_ls, 2));
        HIPCHK(c, hipMemsetAsync(d_ls, 0, 2 * sizeof(u64), c->stream));
        if (nloc) hipLaunchKernelGGL(long_stats_kernel, dim3(flat_grid(c, nloc)), dim3(256), 0, c->stream, (const u16 *)c->d_len, c->q_lo, c->q_hi, d_ls);
        HIPCHK(c, hipMemcpyAsync(hls, d_ls, sizeof hls, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        dev_free(c, &d_ls, 2);
        CHK(host_reduce(c, &hls[0], 1));
        CHK(host_reduce(c, &hls[1], 1, true));
        c->job_n_long = hls[0];
        c->job_short_max = (u32)hls[1];
    }
    c->phase = 1;
    return DISCO_OK;
}

int disco_dist_upload_reads(disco_ctx *c, const uint64_t *packed_own, uint32_t stride_words, const uint16_t *len_own, uint64_t n_total)
{
    DISCO_TRACE("disco_dist_upload_reads");
    if (!c) return DISCO_E_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t dstride = (stride_words + 7u) & ~7u;
    CHK(dist_set_reads(c, n_total, dstride));
    const u64 nloc = c->q_hi - c->q_lo;
    if (nloc && (!packed_own || !len_own)) return fail(c, DISCO_E_ARG, "disco_dist_upload_reads: null argument");
    if (nloc) {
        u64 *dst = c->d_reads + c->q_lo * (u64)dstride;
        if (dstride != stride_words) HIPCHK(c, hipMemsetAsync(dst, 0, nloc * (u64)dstride * 8, c->stream));
        HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)dstride * 8, packed_own, (size_t)stride_words * 8, (size_t)stride_words * 8, nloc, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(c->d_len + c->q_lo, len_own, nloc * 2, hipMemcpyHostToDevice, c->stream));
    }
    return dist_validate(c);
}

int disco_dist_generate_reads(disco_ctx *c, const disco_genspec_abi *s)
{
    DISCO_TRACE("disco_dist_generate_reads");
    if (!c || !s) return c ? fail(c, DISCO_E_ARG, "disco_dist_generate_reads: null argument") : DISCO_E_ARG;
    disco_genspec g;
    memcpy(&g, s, sizeof g);
    const uint32_t longest = std::max<uint32_t>(s->len_max, DISCO_GEN_LONG_SHARE(&g) ? DISCO_GEN_LONG_LEN(&g) : 0u);
    if (s->len_min == 0 || s->len_max < s->len_min || longest > 32767 || s->n_contigs == 0 || s->contig_len < longest)
        return fail(c, DISCO_E_ARG, "disco_dist_generate_reads: bad spec");
    HIPCHK(c, hipSetDevice(c->device));
    const uint32_t stride = (((longest + 31) / 32) + 7u) & ~7u;
    CHK(dist_set_reads(c, s->n_reads, stride));
    const u64 nloc = c->q_hi - c->q_lo;
    if (nloc) hipLaunchKernelGGL(generate_reads_kernel, dim3(flat_grid(c, nloc * stride)), dim3(256), 0, c->stream, g, c->d_reads, c->d_len, (int)stride, c->q_lo, c->q_hi);
    HIPCHK(c, hipGetLastError());
    return dist_validate(c);
}

static int dist_run_graph_pass(disco_ctx *c, uint32_t flags, bool allow_loci, bool *retry_with_id_ranges);

/* one pass; ranks own loci wherever the pass supports it — exact overlaps, fewer than 2^30 reads, the index replicated after its
 * partitioned build (DISCO_DIST_ID_RANGES=1, every rank alike: the id ranges of rounds 1-4). A pass that ends up in the
 * order-dependent regime (the whole adjacency to everybody: too many one-sided pairs) is redone over id ranges: the decision is taken
 * from all-gathered numbers, so every rank takes it alike. */
static int dist_run_graph_impl(disco_ctx *c, uint32_t flags)
{
    bool retry = false;
    int rc = dist_run_graph_pass(c, flags, true, &retry);
    if (rc == DISCO_OK && retry) rc = dist_run_graph_pass(c, flags & ~(uint32_t)DISCO_DIST_GATHER_READS, false, &retry); /* (the reads are everywhere by now) */
    return rc;
}

int disco_dist_run_graph(disco_ctx *c, uint32_t flags)
{
    if (c && c->comm)
        c->time_exchanges = getenv("DISCO_DIST_TIME_EXCHANGES") != nullptr ||
                            (c->dist_passes == 0 && c->comm->world > 1 && !strcmp(c->comm->kind(), "rccl") && !getenv("DISCO_DIST_NO_FIRST_CONTACT"));
    const int rc = dist_run_graph_impl(c, flags);
    if (c) c->dist_passes++;
    if (rc != DISCO_OK && c && c->comm && rc != DISCO_E_ARG && rc != DISCO_E_STATE) {
        /* this rank leaves the pass early: whoever waits for it inside a collective must not wait forever */
        c->comm->abort();
        if (c->comm_bulk) c->comm_bulk->abort();
    }
    return rc;
}

/* the arena of a multi-GPU context (DevArena): made once, before the first collective of the first pass. Sized for what a pass keeps
 * per read — measured with 8 ranks at 50 M x 150 bp (8.9 GB per rank): about 760 bytes per own read (hit buffer, adjacency, headers,
 * exchange buffers) + 46 per read of the job (index, containment keys, bitmaps) — with 60 % on top; DISCO_DIST_ARENA_MB overrides, DISCO_DIST_NO_ARENA=1
 * switches it off (every request then goes to the runtime, as in rounds 1-3) */
static int arena_reserve(disco_ctx *c)
{
    if (c->arena.base || getenv("DISCO_DIST_NO_ARENA")) return DISCO_OK;
    const u64 own = c->q_hi - c->q_lo;
    size_t want = (size_t)((double)(own * 760ull + c->n * 46ull) * 1.6) + (256ull << 20);
    if (const char *e = getenv("DISCO_DIST_ARENA_MB")) want = (size_t)atoll(e) << 20;
    size_t fr = 0, tot = 0;
    HIPCHK(c, hipMemGetInfo(&fr, &tot));
    want = std::min(want, fr / 10 * 9);
    void *p = nullptr;
    while (want >= (64ull << 20)) {
        if (hipMalloc(&p, want) == hipSuccess) break;
        (void)hipGetLastError();
        p = nullptr;
        want /= 2;
    }
    if (!p) return DISCO_OK; /* no arena: the runtime serves the pass (counted) */
    c->arena.base = (char *)p;
    c->arena.size = want & ~(size_t)255;
    c->arena.free_at[0] = c->arena.size;
    return DISCO_OK;
}

static int dist_run_graph_pass(disco_ctx *c, uint32_t flags, bool allow_loci, bool *retry_with_id_ranges)
{
    DISCO_TRACE("disco_dist_run_graph");
    *retry_with_id_ranges = false;
    if (!c) return DISCO_E_ARG;
    if (!c->comm) return fail(c, DISCO_E_STATE, "disco_dist_run_graph: no communicator");
    if (!c->dist_reads || c->phase < 1) return fail(c, DISCO_E_STATE, "disco_dist_run_graph: set the reads with disco_dist_upload_reads / disco_dist_generate_reads");
    HIPCHK(c, hipSetDevice(c->device));
    const u32 G = (u32)c->comm->world, r = (u32)c->comm->rank;
    /* the pass starts from the rank's home range; inside a pass whose ranks own loci q_lo / q_hi are positions of the own list, and
     * whichever way the pass ends they name the home range again */
    struct HomeRange {
        disco_ctx *c;
        ~HomeRange()
        {
            c->q_lo = c->home_lo;
            c->q_hi = c->home_hi;
        }
    } home_range{c};
    c->q_lo = c->home_lo;
    c->q_hi = c->home_hi;
    c->loci = false;
    CHK(arena_reserve(c));
    struct PassToken { /* (in-process transport, DISCO_LOOP_SERIALIZE: one rank's compute segment on the device at a time) */
        DiscoComm *cm;
        explicit PassToken(DiscoComm *x) : cm(x) { cm->begin_pass(); }
        ~PassToken() { cm->end_pass(); }
    } pass_token(c->comm);
    const auto t_pass = HClock::now();
    const PassCounters pc0 = tl_pass;
    const u32 ops0 = c->comm->n_ops + (c->comm_bulk ? c->comm_bulk->n_ops : 0u);
    const u32 hops0 = c->comm->n_host_ops + (c->comm_bulk ? c->comm_bulk->n_host_ops : 0u);
    c->hbm_peak = c->hbm_bytes;
    disco_dist_info &di = c->dinfo;
    memset(&di, 0, sizeof di);
    di.world = G;
    di.rank = r;
    di.n_reads = c->n;
    di.own_lo = c->home_lo;
    di.own_hi = c->home_hi;
    di.own_reads = c->home_hi - c->home_lo;
    c->dist_active = true;
    c->part_index = (flags & DISCO_DIST_KEEP_INDEX_PARTITIONED) != 0 || getenv("DISCO_DIST_PARTITIONED_INDEX") != nullptr;
    /* Round 6 — two classes of rows under a communicator (the reference packs every read at its own length,
     * BG/HashTable.cpp:456-477): a job whose table is wider than 64 bytes because of a FEW long reads (buildG --gpus N on a set with one
     * 600 bp read used to pay the generic probe and the wide-row verify for everybody: 1.6 x) gets 64-byte rows on every rank — each rank
     * converts its replica once the gather of the reads is through (two_class_convert: one pass over the table; the same decision on
     * every rank, from the same table), the pass runs over id ranges (the own reads' index pass needs the converted table, i.e. the whole
     * gather: nothing is dealt ahead), the long reads of a rank's range take the long class's kernels as on one GPU. Every rank decides
     * alike: the shape of the JOB (stride, longest read: dist_validate reduced them). DISCO_DIST_NO_TWO_CLASS=1: one stride, as before. */
    bool ragged_job = false;
    if (!c->part_index && c->prm.max_substitutions == 0 && !getenv("DISCO_DIST_NO_TWO_CLASS") && !getenv("DISCO_NO_TWO_CLASS")) {
        if (c->two_class) ragged_job = true;
        else if (c->S > VERIFY_SW && c->job_n_long) { /* the test of a single GPU (at most one long read in five, run lists for the others), on the JOB's figures */
            c->dist_two_class = true;
            ragged_job = two_class_ok(c, c->S, c->n, c->job_n_long, c->job_short_max);
            c->dist_two_class = false;
        }
    }
    const bool want_loci = allow_loci && !ragged_job && !c->part_index && c->prm.max_substitutions == 0 && c->n < (1ull << 30) && !getenv("DISCO_DIST_ID_RANGES") && !getenv("DISCO_DIST_FORCE_GATHER");
    c->n_push_r = 0;
    c->h_len.clear();
    /* 0. everybody gets every read — on the second communicator and stream: the index build and the probe of the own reads
     *    need nothing of it, verify waits for it (disco_probe).
     *    Two communicators in flight on one device are safe when every rank issues their operations in the same order (RCCL / NCCL's
     *    rule for concurrent communicators) and every kernel that shares the device with them terminates: all ranks run this very
     *    function — the all-gather of the reads is the only kind of operation comm_bulk ever carries and sits at the same place of
     *    the program order of every rank, in front of the pass's operations on comm — and the compute kernels next to it are finite passes over a work
     *    queue. DISCO_DIST_ONE_COMM=1 (every rank alike) takes the second communicator out of the picture: the all-gather then runs on
     *    comm and the context's stream, in front of the index exchanges, at the price of its 5 ms (G = 8, config 4) on the critical path. */
    const bool one_comm = c->comm_bulk == nullptr || getenv("DISCO_DIST_ONE_COMM") != nullptr;
    DiscoComm *const bulk = one_comm ? c->comm : c->comm_bulk;
    const hipStream_t bstream = one_comm ? c->stream : c->bulk_stream;
    if ((flags & DISCO_DIST_GATHER_READS) && !c->two_class) { /* (a table that has its two classes is complete: an earlier pass gathered it) */
        const auto t0 = HClock::now();
        const u64 row_bytes = (u64)c->S * 8;
        HIPCHK(c, hipStreamSynchronize(c->stream)); /* the own rows are in place */
#define BULK_CHK(expr)                                                                                        \
    do {                                                                                                      \
        int rc_ = (expr);                                                                                     \
        if (rc_ != DISCO_OK) return fail(c, rc_, "%s: %s", #expr, bulk->err.c_str());                 \
    } while (0)
        const int W = (int)((c->max_len + 31) / 32); /* words the longest read of the job uses (150 bp: 5 of the 8-word stride) */
        u64 sent_row_bytes = row_bytes;
        if (W < c->S && !getenv("DISCO_DIST_FULL_ROWS")) {
            /* pack the own rows, all-gather the dense array, spread the other ranks' rows back over the table: 37 % fewer bytes
             * on the links at 150 bp, paid with two streaming passes on the second stream while the index is being built */
            /* a rank's block = its rows and, behind them, its lengths (per is a multiple of 64: 2 per bytes are whole words): ONE operation */
            const u64 total_rows = c->per * (u64)G;
            const u64 rows_words = c->per * (u64)W, block_words = rows_words + c->per / 4;
            CHK(ensure_cap(c, &c->d_dense, &c->dense_cap, block_words * (u64)G));
            u64 *mine = c->d_dense + (u64)r * block_words;
            hipLaunchKernelGGL(pack_rows_kernel, dim3(flat_grid(c, c->per * W)), dim3(256), 0, bstream, c->d_reads, c->S, W, (u64)r * c->per, c->per, mine);
            HIPCHK(c, hipMemcpyAsync(mine + rows_words, c->d_len + (u64)r * c->per, c->per * 2, hipMemcpyDeviceToDevice, bstream));
            BULK_CHK(bulk->all_gather(mine, c->d_dense, block_words * 8, bstream));
            hipLaunchKernelGGL(unpack_rows_kernel, dim3(flat_grid(c, total_rows * W)), dim3(256), 0, bstream, c->d_dense, c->S, W, total_rows, (u64)r * c->per,
                               (u64)(r + 1) * c->per, c->d_reads, c->per, block_words);
            hipLaunchKernelGGL(unpack_lens_kernel, dim3(flat_grid(c, total_rows)), dim3(256), 0, bstream, (const u64 *)c->d_dense, c->per, block_words, rows_words, G, r, c->d_len);
            HIPCHK(c, hipGetLastError());
            sent_row_bytes = (u64)W * 8;
        } else {
            BULK_CHK(bulk->all_gather(c->d_reads + (u64)r * c->per * c->S, c->d_reads, c->per * row_bytes, bstream));
            BULK_CHK(bulk->all_gather(c->d_len + (u64)r * c->per, c->d_len, c->per * 2, bstream));
        }
#undef BULK_CHK
        HIPCHK(c, hipEventRecord(c->ev_bulk, bstream));
        c->wait_bulk_before_verify = true;
        di.bytes_sent[DISCO_X_READS] += (u64)(G - 1) * c->per * (sent_row_bytes + 2);
        di.ms[DISCO_X_READS] += ms_since(t0); /* time to ISSUE it (RCCL: asynchronous; in-process transport: the copies themselves) */
    }
    if (ragged_job && !c->two_class) {
        if (c->wait_bulk_before_verify) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_bulk, 0)); /* the whole table, then its two classes */
        c->dist_two_class = true;
        const int rc2 = two_class_convert(c);
        c->dist_two_class = false;
        CHK(rc2);
    }
    if (want_loci) CHK(dist_deal_reads(c));
    di.placement = c->loci ? 1u : 0u;
    CHK(dist_build_index(c));
    c->contained_done = false;
    CHK(disco_probe(c));
    if (!c->contained_done) CHK(dist_mark_contained(c));
    CHK(select_edges(c));
    /* whole-job figures and the regime decision */
    const u64 probes = c->home_probes; /* (dist_validate) */
    if (!c->d_list_n) CHK(dev_alloc(c, &c->d_list_n, 1));
    u64 g[8] = {c->adj_total, c->dropped_local, c->n_contained, c->h_ctr[CTR_CAP_SITES], c->h_ctr[CTR_KMER_HITS], probes, 0, 0};
    {
        constexpr int NV = 6;
        std::vector<u64> all((size_t)G * NV);
        COMM_CHK(c, c->comm->host_all_gather((const unsigned long long *)g, NV, (unsigned long long *)all.data(), c->stream));
        std::vector<u64> adj_totals((size_t)G);
        u64 sum[NV] = {0, 0, 0, 0, 0, 0};
        for (u32 p = 0; p < G; p++) {
            adj_totals[p] = all[(size_t)p * NV];
            for (int i = 0; i < NV; i++) sum[i] += all[(size_t)p * NV + i];
        }
        di.probes = sum[5];
        di.dropped_hits = sum[1];
        di.n_contained = sum[2];
        di.n_contained_local = c->n_contained;
        di.cap_bind_sites = sum[3];
        di.kmer_hits = sum[4];
        di.e_pre = sum[0] / 2;
        c->dropped = sum[1];
        bool irregular = c->n >= (1ull << 30) || c->prm.max_substitutions != 0 || getenv("DISCO_DIST_FORCE_GATHER");
        u64 asym_total = 0;
        if (!irregular && sum[1] != 0) {
            /* somebody dropped a hit (per-k-mer cap, second hit to a destination: real data do at their repeats). Only the lists of
             * the reads that dropped something can lack a twin: complete those across ranks, then the regular regime applies */
            bool done = false;
            if (!getenv("DISCO_DIST_NO_TWIN_PUSH")) CHK(dist_complete_twins(c, &done, &asym_total));
            irregular = !done;
            if (done) {
                u64 t[1] = {c->adj_total};
                CHK(host_reduce(c, t, 1));
                di.e_pre = t[0] / 2;
                di.regime = 2;
            }
        }
        if (irregular && c->loci) { /* the gather of the whole adjacency walks id ranges: the pass is redone over them (collective decision) */
            *retry_with_id_ranges = true;
            c->loci = false;
            return DISCO_OK;
        }
        if (irregular) {
            di.regime = 1;
            CHK(dist_irregular(c, adj_totals));
            di.e_pre = c->adj_total / 2;
            di.asymmetric_pairs = c->asym_local;
        } else {
            c->phase = 6; /* nobody dropped a hit, or the lists have just been completed: they are symmetric (twin_check's argument) */
            c->n_extra = 0;
            c->asym_local = asym_total;
            di.asymmetric_pairs = asym_total;
            if (di.regime == 0) c->ph_ms[DISCO_PH_TWIN] = 0;
            CHK(dist_transitive_mark(c));
            CHK(dist_push_survivors(c));
        }
    }
    uint64_t n_out = 0;
    CHK(disco_emit_edges(c, &n_out));
    CHK(settle_keys(c)); /* (whatever reads best[] after the pass — disco_fetch_contained — runs on this stream, behind the keys) */
    u64 tot[1] = {(u64)n_out};
    CHK(host_reduce(c, tot, 1));
    di.e_out = tot[0];
    di.e_out_local = n_out;
    di.ms_total = ms_since(t_pass);
    di.device_allocs = tl_pass.dev_allocs - pc0.dev_allocs;
    di.device_frees = tl_pass.dev_frees - pc0.dev_frees;
    di.host_syncs = tl_pass.host_syncs - pc0.host_syncs + (c->comm->n_host_ops + (c->comm_bulk ? c->comm_bulk->n_host_ops : 0u) - hops0);
    di.comm_ops = c->comm->n_ops + (c->comm_bulk ? c->comm_bulk->n_ops : 0u) - ops0;
    di.arena_bytes = c->arena.size;
    di.arena_peak = c->arena.peak;
    di.hbm_peak = c->hbm_peak;
    {
        float km = 0;
        for (int i = 0; i < DISCO_PH_COUNT; i++) km += c->ph_ms[i];
        di.kernel_ms = km;
    }
    return DISCO_OK;
}

int disco_dist_get_info(disco_ctx *c, disco_dist_info *out)
{
    if (!c || !out) return DISCO_E_ARG;
    *out = c->dinfo;
    return DISCO_OK;
}

} /* extern "C" */
