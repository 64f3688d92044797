/*
 * disco_comm.h — the exchange layer of the multi-GPU BuildGraph flow (host side, part of libdisco_hip.so).
 *
 * Replaces the communication of the reference's two multi-process binaries:
 *   buildG-MPI    : MPI_Isend / MPI_Recv gossip of marked and contained read ids (MPI/OverlapGraph.cpp:218-246,473-506)
 *   buildG-MPIRMA : one MPI_Get per hash bucket out of an MPI-3 RMA window over the sharded hashData
 *                   (RMA/HashTable.cpp:422-435 window, :644-653,694-705 gets)
 * with BULK collectives on device buffers: all-to-all-v (grouped ncclSend / ncclRecv: on the xGMI mesh every peer pair has its
 * own link, so all 7 links of a GPU carry a slice at once), all-gather(-v) and reduce-scatter(MIN).
 *
 * Two transports behind one interface:
 *   RcclComm : one rank per GPU over RCCL (the product path; processes under torchrun / mpirun-like launchers, or one host
 *              thread per GPU inside buildG --gpus N)
 *   LoopComm : ranks = host threads of ONE process whose contexts live on the same device (or on peers that can address each
 *              other): the collectives are device-to-device copies between the ranks' buffers, fenced by host barriers.
 *              RCCL refuses two ranks on one GPU; this is what lets the exact multi-rank code path run on the 1-GPU test
 *              boxes (tests/test_gpu_dist.py, buildG --gpus N --same-device).
 * All byte counts are size_t; every call is collective (all ranks, same order) and returns DISCO_OK or a negative code.
 */
#ifndef DISCO_COMM_H_
#define DISCO_COMM_H_

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/disco_hip.h"

struct DiscoComm {
    int rank = 0, world = 1;
    unsigned n_ops = 0;      /* operations issued on this communicator (counted by the implementations, reported per pass) */
    unsigned n_host_ops = 0; /* ... of which exchanges of host values: each one waits for the stream */
    std::string err;
    virtual ~DiscoComm() {}
    virtual const char *kind() const = 0;
    /* every rank contributes `bytes` at send; rank p's block lands at recv + p * bytes (in place when send == recv + rank * bytes) */
    virtual int all_gather(const void *send, void *recv, size_t bytes, hipStream_t s) = 0;
    /* blocks of different sizes: rank p's block (cnt[p] bytes) lands at recv + off[p]; send may alias recv + off[rank] */
    virtual int all_gather_v(const void *send, void *recv, const size_t *off, const size_t *cnt, hipStream_t s) = 0;
    /* block for rank p = send + soff[p] (scnt[p] bytes); the block from rank p lands at recv + roff[p] (rcnt[p] bytes) */
    virtual int all_to_all_v(const void *send, const size_t *soff, const size_t *scnt, void *recv, const size_t *roff, const size_t *rcnt,
                             hipStream_t s) = 0;
    /* buf = world blocks of `per` int64 values; afterwards block `rank` of this rank's buf holds the element-wise minimum of
     * that block over all ranks (the other blocks are unspecified) */
    virtual int reduce_scatter_min_i64(void *buf, size_t per, hipStream_t s) = 0;
    /* control plane: n host values per rank -> all[world * n], rank-major (synchronises the stream) */
    virtual int host_all_gather(const unsigned long long *mine, int n, unsigned long long *all, hipStream_t s) = 0;
    virtual int barrier(hipStream_t s) = 0;
    /* a rank that leaves a pass on an error of its own must not let the others wait for it inside a collective: in-process ranks are
     * released with an error; an RCCL communicator is aborted (its peers see the failure through their own communicator's async
     * error / launcher, buildG _exit()s — the process-level convention of the reference's MPI binaries) */
    virtual void abort() = 0;
    /* measurement aid of the in-process transport (DISCO_LOOP_SERIALIZE=1): the ranks of one process share a device, and with it the
     * compute segments of a pass — between two operations on the communicator — run one rank at a time, so that a rank's phase timers
     * show its own kernels and the sum over the ranks is the work of the whole job (tools/dist_profile.py: work inflation). A pass
     * holds the token from begin_pass to end_pass and gives it up inside every operation. No-ops on RCCL. */
    virtual void begin_pass() {}
    virtual void end_pass() {}
};

/* ---------------------------------------------------------------------------------------------------------------- */
#define DISCO_NCCL(call)                                                                     \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            err = std::string(#call) + ": " + ncclGetErrorString(r_);                        \
            return DISCO_E_HIP;                                                              \
        }                                                                                    \
    } while (0)
/* between ncclGroupStart and ncclGroupEnd: close the group before leaving on an error, or the communicator stays inside it */
#define DISCO_NCCL_G(call)                                                                   \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) {                                                             \
            err = std::string(#call) + ": " + ncclGetErrorString(r_);                        \
            (void)ncclGroupEnd();                                                            \
            return DISCO_E_HIP;                                                              \
        }                                                                                    \
    } while (0)
#define DISCO_COMM_HIP(call)                                                                 \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return DISCO_E_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define DISCO_COMM_ALIVE()                                                                   \
    do {                                                                                     \
        if (!comm) {                                                                         \
            err = "the communicator was aborted";                                            \
            return DISCO_E_STATE;                                                            \
        }                                                                                    \
    } while (0)

struct RcclComm final : DiscoComm {
    ncclComm_t comm = nullptr;
    unsigned long long *d_small = nullptr; /* staging of host_all_gather */
    size_t small_cap = 0;
    const char *kind() const override { return "rccl"; }
    /* Two steps, so that everything that can fail on ONE rank alone happens before that rank enters a collective (a rank whose
     * staging allocation failed after ncclCommInitRank left its peers inside the next collective for ever):
     *   prepare(): the staging of host_all_gather, device memory plus a pinned host mirror — no communication;
     *   join():    ncclCommInitRank — collective; a rank that returns from it with an error has told its peers through the bootstrap.
     * The staging exists before a pass starts: with one host thread per GPU in one process (buildG --gpus N) a device allocation while
     * peers sit in an RCCL kernel is the classic stall. SMALL_VALUES per rank cover every use of a pass (the largest: world x world counts
     * of an all-to-all-v); a larger request is an ERROR (round 6: it used to free and reallocate the staging inside the pass). */
    static constexpr size_t SMALL_VALUES = 4096;
    unsigned long long *h_small = nullptr; /* pinned: what the device-to-host copy of host_all_gather lands in */
    int prepare(int nranks, int rk)
    {
        rank = rk;
        world = nranks;
        small_cap = (size_t)(world + 1) * SMALL_VALUES;
        DISCO_COMM_HIP(hipMalloc((void **)&d_small, small_cap * 8));
        DISCO_COMM_HIP(hipHostMalloc((void **)&h_small, small_cap * 8, hipHostMallocDefault));
        return DISCO_OK;
    }
    int join(const void *unique_id)
    {
        ncclUniqueId id;
        memcpy(&id, unique_id, sizeof id);
        DISCO_NCCL(ncclCommInitRank(&comm, world, id, rank));
        return DISCO_OK;
    }
    int init(const void *unique_id, int nranks, int rk)
    {
        const int rc = prepare(nranks, rk);
        return rc != DISCO_OK ? rc : join(unique_id);
    }
    ~RcclComm() override
    {
        if (d_small) (void)hipFree(d_small);
        if (h_small) (void)hipHostFree(h_small);
        if (comm) (void)ncclCommDestroy(comm);
    }
    int all_gather(const void *send, void *recv, size_t bytes, hipStream_t s) override
    {
        n_ops++;
        DISCO_COMM_ALIVE();
        if (bytes == 0) return DISCO_OK;
        DISCO_NCCL(ncclAllGather(send, recv, bytes, ncclInt8, comm, s));
        return DISCO_OK;
    }
    int all_gather_v(const void *send, void *recv, const size_t *off, const size_t *cnt, hipStream_t s) override
    {
        n_ops++;
        DISCO_COMM_ALIVE();
        /* blocks of one size, laid out rank after rank (the bucket-table slices whenever the world divides the table): the library's own
         * all-gather instead of world - 1 send / receive pairs. (The record slices differ by a fraction of a per cent and stay a grouped
         * exchange: padding them to one pitch would put slots without a record inside the LAST bucket of every slice — a bucket ends
         * where the next one begins — and the probe walks every slot of a bucket) */
        bool regular = true;
        for (int p = 0; p < world; p++) regular = regular && cnt[p] == cnt[0] && off[p] == (size_t)p * cnt[0];
        if (regular) {
            if (cnt[0] == 0) return DISCO_OK;
            DISCO_NCCL(ncclAllGather(send, recv, cnt[0], ncclInt8, comm, s));
            return DISCO_OK;
        }
        DISCO_NCCL(ncclGroupStart());
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            if (cnt[rank]) DISCO_NCCL_G(ncclSend(send, cnt[rank], ncclInt8, p, comm, s));
            if (cnt[p]) DISCO_NCCL_G(ncclRecv((char *)recv + off[p], cnt[p], ncclInt8, p, comm, s));
        }
        DISCO_NCCL(ncclGroupEnd());
        if (cnt[rank] && send != (const char *)recv + off[rank])
            DISCO_COMM_HIP(hipMemcpyAsync((char *)recv + off[rank], send, cnt[rank], hipMemcpyDeviceToDevice, s));
        return DISCO_OK;
    }
    int all_to_all_v(const void *send, const size_t *soff, const size_t *scnt, void *recv, const size_t *roff, const size_t *rcnt,
                     hipStream_t s) override
    {
        n_ops++;
        DISCO_COMM_ALIVE();
        DISCO_NCCL(ncclGroupStart());
        for (int p = 0; p < world; p++) {
            if (p == rank) continue;
            if (scnt[p]) DISCO_NCCL_G(ncclSend((const char *)send + soff[p], scnt[p], ncclInt8, p, comm, s));
            if (rcnt[p]) DISCO_NCCL_G(ncclRecv((char *)recv + roff[p], rcnt[p], ncclInt8, p, comm, s));
        }
        DISCO_NCCL(ncclGroupEnd());
        if (scnt[rank]) /* the block a rank keeps never touches a link */
            DISCO_COMM_HIP(hipMemcpyAsync((char *)recv + roff[rank], (const char *)send + soff[rank], scnt[rank], hipMemcpyDeviceToDevice, s));
        return DISCO_OK;
    }
    int reduce_scatter_min_i64(void *buf, size_t per, hipStream_t s) override
    {
        n_ops++;
        DISCO_COMM_ALIVE();
        if (per == 0) return DISCO_OK;
        DISCO_NCCL(ncclReduceScatter(buf, (long long *)buf + (size_t)rank * per, per, ncclInt64, ncclMin, comm, s));
        return DISCO_OK;
    }
    int host_all_gather(const unsigned long long *mine, int n, unsigned long long *all, hipStream_t s) override
    {
        n_ops++;
        n_host_ops++;
        const size_t need = (size_t)(world + 1) * n;
        if (!comm || need > small_cap) { /* (never a reallocation inside a pass: peers may be inside an RCCL kernel) */
            err = !comm ? "host_all_gather: the communicator was aborted" : "host_all_gather: more values per rank than the staging made at init holds (RcclComm::SMALL_VALUES)";
            return DISCO_E_CAPACITY;
        }
        /* through the pinned mirror both ways: a copy out of / into pageable memory makes the runtime stage it and wait on its own */
        unsigned long long *d_mine = d_small + (size_t)world * n, *h_mine = h_small + (size_t)world * n;
        memcpy(h_mine, mine, (size_t)n * 8);
        DISCO_COMM_HIP(hipMemcpyAsync(d_mine, h_mine, (size_t)n * 8, hipMemcpyHostToDevice, s));
        DISCO_NCCL(ncclAllGather(d_mine, d_small, (size_t)n * 8, ncclInt8, comm, s));
        DISCO_COMM_HIP(hipMemcpyAsync(h_small, d_small, (size_t)world * n * 8, hipMemcpyDeviceToHost, s));
        DISCO_COMM_HIP(hipStreamSynchronize(s));
        memcpy(all, h_small, (size_t)world * n * 8);
        return DISCO_OK;
    }
    int barrier(hipStream_t s) override
    {
        n_ops++;
        n_host_ops++;
        unsigned long long x = 0;
        std::vector<unsigned long long> all((size_t)world);
        return host_all_gather(&x, 1, all.data(), s);
    }
    void abort() override
    {
        if (comm) (void)ncclCommAbort(comm);
        comm = nullptr;
        if (sibling && sibling->comm) sibling->abort(); /* the context's other communicator goes down with this one: a rank that has left a
                                                          pass must not keep its peers inside a collective of EITHER */
    }
    RcclComm *sibling = nullptr;
};

/* ---------------------------------------------------------------------------------------------------------------- */
__global__ void loop_min_i64_kernel(long long *__restrict__ dst, const long long *const *__restrict__ srcs, int world, size_t off, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) {
        long long m = srcs[0][off + i];
        for (int p = 1; p < world; p++) {
            const long long x = srcs[p][off + i];
            m = x < m ? x : m;
        }
        dst[off + i] = m;
    }
}

/* DISCO_LOOP_SERIALIZE=1: one token per process (the two communicators of a context share it) */
struct LoopDeviceToken {
    std::mutex m;
    const bool on = getenv("DISCO_LOOP_SERIALIZE") != nullptr;
    static LoopDeviceToken &get()
    {
        static LoopDeviceToken t;
        return t;
    }
    static bool &holding() /* this thread (= this rank) is inside a pass and has the token */
    {
        static thread_local bool h = false;
        return h;
    }
};
/* gives the token up for the duration of an operation on a communicator (the rank's stream has been synchronised: its segment is over) */
struct LoopTokenPause {
    const bool held;
    LoopTokenPause() : held(LoopDeviceToken::holding())
    {
        if (held) LoopDeviceToken::get().m.unlock();
    }
    ~LoopTokenPause()
    {
        if (held) LoopDeviceToken::get().m.lock();
    }
};

/* state shared by the ranks (threads) of one in-process group */
struct LoopGroup {
    int world = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long gen = 0;
    bool aborted = false;
    std::vector<const void *> ptr;                /* per rank: published buffer                */
    std::vector<std::vector<size_t>> off, cnt;    /* per rank: published per-peer offsets/counts */
    std::vector<unsigned long long> host;         /* host_all_gather scratch                   */
    /* DISCO_LOOP_ASYNC: per rank, "my send block is complete at this point of my stream" and "my pulls are through" (below) */
    std::vector<hipEvent_t> ev_ready, ev_done;
    explicit LoopGroup(int w) : world(w), ptr((size_t)w), off((size_t)w), cnt((size_t)w), ev_ready((size_t)w, nullptr), ev_done((size_t)w, nullptr) {}
    ~LoopGroup()
    {
        for (hipEvent_t e : ev_ready)
            if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_done)
            if (e) (void)hipEventDestroy(e);
    }
    bool wait()
    {
        std::unique_lock<std::mutex> lk(m);
        if (aborted) return false;
        const unsigned long g = gen;
        if (++arrived == world) {
            arrived = 0;
            ++gen;
            cv.notify_all();
            return true;
        }
        cv.wait(lk, [&] { return gen != g || aborted; });
        return !aborted;
    }
    void abort()
    {
        std::lock_guard<std::mutex> lk(m);
        aborted = true;
        cv.notify_all();
    }
};

/* DISCO_LOOP_ASYNC=1 (round 6): the device-side exchanges WITHOUT any wait of the host on the device — what RCCL does: an
 * operation is enqueued on the rank's stream, consumes what the stream produced before it, and completes in stream order; the host
 * returns at once. The in-process form: every rank records "my send block is complete" on its stream, the ranks meet on the HOST
 * (pointers and events change hands; nobody waits for a device), every rank makes its stream wait for its peers' events, enqueues its
 * pulls, records "my pulls are through", and after a second meeting makes its stream wait for every peer's pulls (nobody overwrites a
 * send block a peer still reads). A caller that reads a result without a stream dependency of its own — the assumption rounds 4-5 wrote
 * into the flow when they took the host waits out — now reads garbage in the tests instead of on the first multi-GPU node.
 * Not with DISCO_LOOP_SERIALIZE (that measurement hands the device from rank to rank at host waits). */
struct LoopComm final : DiscoComm {
    std::shared_ptr<LoopGroup> g;
    const long long **d_srcs = nullptr;
    const bool async = getenv("DISCO_LOOP_ASYNC") != nullptr && !LoopDeviceToken::get().on;
    const char *kind() const override { return "loop"; }
    /* first half: my block is complete here (stream order), meet, wait for everybody's blocks (on the stream) */
    int async_open(hipStream_t s)
    {
        hipEvent_t &er = g->ev_ready[(size_t)rank], &ed = g->ev_done[(size_t)rank];
        if (!er && (hipEventCreateWithFlags(&er, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ed, hipEventDisableTiming) != hipSuccess))
            return fail_abort(DISCO_E_HIP);
        if (hipEventRecord(er, s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        if (!g->wait()) {
            err = "loop communicator aborted by another rank";
            return DISCO_E_STATE;
        }
        for (int p = 0; p < world; p++)
            if (p != rank && hipStreamWaitEvent(s, g->ev_ready[(size_t)p], 0) != hipSuccess) return fail_abort(DISCO_E_HIP);
        return DISCO_OK;
    }
    /* second half: my pulls are enqueued; meet; nobody's stream goes on before every peer's pulls are through */
    int async_close(hipStream_t s)
    {
        if (hipEventRecord(g->ev_done[(size_t)rank], s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        if (!g->wait()) {
            err = "loop communicator aborted by another rank";
            return DISCO_E_STATE;
        }
        for (int p = 0; p < world; p++)
            if (p != rank && hipStreamWaitEvent(s, g->ev_done[(size_t)p], 0) != hipSuccess) return fail_abort(DISCO_E_HIP);
        /* (a third meeting: the events are re-recorded by the next operation only after every peer has enqueued its waits on them) */
        if (!g->wait()) {
            err = "loop communicator aborted by another rank";
            return DISCO_E_STATE;
        }
        return DISCO_OK;
    }
    LoopComm(std::shared_ptr<LoopGroup> grp, int rk) : g(std::move(grp))
    {
        rank = rk;
        world = g->world;
    }
    ~LoopComm() override
    {
        if (d_srcs) (void)hipFree(d_srcs);
    }
#define LOOP_WAIT()                                           \
    do {                                                      \
        if (!g->wait()) {                                     \
            err = "loop communicator aborted by another rank"; \
            return DISCO_E_STATE;                             \
        }                                                     \
    } while (0)
    int fail_abort(int rc)
    {
        g->abort();
        return rc;
    }
    int all_gather(const void *send, void *recv, size_t bytes, hipStream_t s) override
    {
        std::vector<size_t> off((size_t)world), cnt((size_t)world, bytes); /* (counted by all_gather_v) */
        for (int p = 0; p < world; p++) off[(size_t)p] = (size_t)p * bytes;
        return all_gather_v(send, recv, off.data(), cnt.data(), s);
    }
    int all_gather_v(const void *send, void *recv, const size_t *off, const size_t *cnt, hipStream_t s) override
    {
        n_ops++;
        if (async) {
            g->ptr[(size_t)rank] = send;
            const int rc = async_open(s);
            if (rc != DISCO_OK) return rc;
            for (int p = 0; p < world; p++) {
                char *dst = (char *)recv + off[p];
                if (cnt[p] && dst != g->ptr[(size_t)p])
                    if (hipMemcpyAsync(dst, g->ptr[(size_t)p], cnt[p], hipMemcpyDeviceToDevice, s) != hipSuccess) return fail_abort(DISCO_E_HIP);
            }
            return async_close(s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP); /* my block is complete */
        LoopTokenPause pause;
        g->ptr[(size_t)rank] = send;
        LOOP_WAIT();
        for (int p = 0; p < world; p++) {
            char *dst = (char *)recv + off[p];
            if (cnt[p] && dst != g->ptr[(size_t)p])
                if (hipMemcpyAsync(dst, g->ptr[(size_t)p], cnt[p], hipMemcpyDeviceToDevice, s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LOOP_WAIT(); /* nobody reuses its send block before every peer has pulled it */
        return DISCO_OK;
    }
    int all_to_all_v(const void *send, const size_t *soff, const size_t *scnt, void *recv, const size_t *roff, const size_t *rcnt,
                     hipStream_t s) override
    {
        n_ops++;
        if (async) {
            g->ptr[(size_t)rank] = send;
            g->off[(size_t)rank].assign(soff, soff + world);
            g->cnt[(size_t)rank].assign(scnt, scnt + world);
            const int rc = async_open(s);
            if (rc != DISCO_OK) return rc;
            for (int p = 0; p < world; p++) {
                const size_t n = g->cnt[(size_t)p][(size_t)rank];
                if (n != rcnt[p]) {
                    err = "all_to_all_v: receive count does not match the peer's send count";
                    return fail_abort(DISCO_E_ARG);
                }
                if (n && hipMemcpyAsync((char *)recv + roff[p], (const char *)g->ptr[(size_t)p] + g->off[(size_t)p][(size_t)rank], n, hipMemcpyDeviceToDevice, s) != hipSuccess)
                    return fail_abort(DISCO_E_HIP);
            }
            return async_close(s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LoopTokenPause pause;
        g->ptr[(size_t)rank] = send;
        g->off[(size_t)rank].assign(soff, soff + world);
        g->cnt[(size_t)rank].assign(scnt, scnt + world);
        LOOP_WAIT();
        for (int p = 0; p < world; p++) {
            const size_t n = g->cnt[(size_t)p][(size_t)rank];
            if (n != rcnt[p]) {
                err = "all_to_all_v: receive count does not match the peer's send count";
                return fail_abort(DISCO_E_ARG);
            }
            if (n && hipMemcpyAsync((char *)recv + roff[p], (const char *)g->ptr[(size_t)p] + g->off[(size_t)p][(size_t)rank], n, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return fail_abort(DISCO_E_HIP);
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LOOP_WAIT();
        return DISCO_OK;
    }
    int reduce_scatter_min_i64(void *buf, size_t per, hipStream_t s) override
    {
        n_ops++;
        if (async) {
            g->ptr[(size_t)rank] = buf;
            if (!d_srcs && hipMalloc((void **)&d_srcs, sizeof(void *) * (size_t)world) != hipSuccess) return fail_abort(DISCO_E_NOMEM);
            const int rc = async_open(s);
            if (rc != DISCO_OK) return rc;
            if (per) {
                /* (the table of the peers' pointers: a pageable copy is staged by the runtime before the call returns) */
                if (hipMemcpyAsync(d_srcs, g->ptr.data(), sizeof(void *) * (size_t)world, hipMemcpyHostToDevice, s) != hipSuccess) return fail_abort(DISCO_E_HIP);
                hipLaunchKernelGGL(loop_min_i64_kernel, dim3((unsigned)std::min<size_t>((per + 255) / 256, 4096)), dim3(256), 0, s, (long long *)buf, d_srcs, world,
                                   (size_t)rank * per, per);
            }
            return async_close(s);
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LoopTokenPause pause;
        g->ptr[(size_t)rank] = buf;
        LOOP_WAIT();
        if (per) {
            if (!d_srcs && hipMalloc((void **)&d_srcs, sizeof(void *) * (size_t)world) != hipSuccess) return fail_abort(DISCO_E_NOMEM);
            if (hipMemcpyAsync(d_srcs, g->ptr.data(), sizeof(void *) * (size_t)world, hipMemcpyHostToDevice, s) != hipSuccess) return fail_abort(DISCO_E_HIP);
            /* reads block `rank` of every rank's buffer, writes block `rank` of mine: nobody else touches those */
            hipLaunchKernelGGL(loop_min_i64_kernel, dim3((unsigned)std::min<size_t>((per + 255) / 256, 4096)), dim3(256), 0, s, (long long *)buf, d_srcs, world,
                               (size_t)rank * per, per);
        }
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LOOP_WAIT();
        return DISCO_OK;
    }
    int host_all_gather(const unsigned long long *mine, int n, unsigned long long *all, hipStream_t s) override
    {
        n_ops++;
        n_host_ops++;
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LoopTokenPause pause;
        {
            std::lock_guard<std::mutex> lk(g->m);
            if (g->host.size() < (size_t)world * n) g->host.resize((size_t)world * n);
        }
        LOOP_WAIT(); /* the scratch has its final size before anybody writes */
        for (int i = 0; i < n; i++) g->host[(size_t)rank * n + i] = mine[i];
        LOOP_WAIT();
        for (size_t i = 0; i < (size_t)world * n; i++) all[i] = g->host[i];
        LOOP_WAIT();
        return DISCO_OK;
    }
    int barrier(hipStream_t s) override
    {
        n_ops++;
        n_host_ops++;
        if (hipStreamSynchronize(s) != hipSuccess) return fail_abort(DISCO_E_HIP);
        LoopTokenPause pause;
        LOOP_WAIT();
        return DISCO_OK;
    }
    void abort() override { g->abort(); }
    void begin_pass() override
    {
        if (LoopDeviceToken::get().on && !LoopDeviceToken::holding()) {
            LoopDeviceToken::get().m.lock();
            LoopDeviceToken::holding() = true;
        }
    }
    void end_pass() override
    {
        if (LoopDeviceToken::holding()) {
            LoopDeviceToken::holding() = false;
            LoopDeviceToken::get().m.unlock();
        }
    }
#undef LOOP_WAIT
};

#endif /* DISCO_COMM_H_ */
