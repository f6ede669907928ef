// fwx_multi.hip -- the row-partitioned solve behind the C ABI: ONE process, several devices.
//
// The reference's loop has one call site (ProcessRequests.hs:82-84 -> floydWarshall,
// /root/reference/src/lib/Algorithms.hs:19-20); a host bound to that one call reaches the whole
// node through fwx_matrix_create_multi / fwx_solve_multi_*.  Step k of runAlgo (:42-61) needs, for
// a local row i, r[i][k] and next[i][k] (local) and pivot row k AS IT STANDS AT THE START OF STEP
// k -- nothing else -- so the only exchange is the snapshot panel of 64 pivot rows per pass, sent by
// the partition that owns them: ncclBroadcast on RCCL (distinct devices) or hipMemcpyPeerAsync
// (also when a device is listed more than once: logical partitions, which is how this schedule
// is tested on one GPU).  Same kernels, same operands, same order as the single-device fused
// engine: bit-identical results.
//
// Streams per partition: `main` runs colpanel + main kernels, `side` runs the owner's snapshot
// panel and the exchange, so that with look-ahead neither sits on the critical path.  One host
// thread enqueues everything; ordering is by events only (no host synchronisation inside a solve
// apart from the in-flight throttle).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl.so.1 is dlopen()ed on first use
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include "fwx.h"
#include "fwx_guard.h"
#include "fwx_internal.h"
#include "fwx_kernels.h"
#include "fwx_replay.h"

namespace fwxi {

// ---- RCCL, loaded lazily -------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;     // optional: used for communicators in an error state
    bool ok = false;
};

static RcclApi &rccl()
{
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // by SONAME: a process that already holds an RCCL (torch bundles one) gets that mapping
        api.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!api.lib) return;
        api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.lib, "ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.Broadcast = (decltype(api.Broadcast))dlsym(api.lib, "ncclBroadcast");
        api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
        api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
        api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");
        api.ok = api.CommInitAll && api.CommDestroy && api.Broadcast && api.GroupStart && api.GroupEnd &&
                 api.CommCount;
    });
    return api;
}

// (used where a MultiState `M` with a communicator set is in scope: a failure marks the set, see
// CommCache::release)
#define FWX_NCCL(call)                                                                             \
    do {                                                                                           \
        if ((call) != ncclSuccess) {                                                               \
            if (M.comms) M.comms->poisoned = true;                                                 \
            return FWX_ERR_RCCL;                                                                   \
        }                                                                                          \
    } while (0)

// ---- communicators, cached -------------------------------------------------------------------------
// ncclCommInitAll costs hundreds of milliseconds and a communicator is tied to nothing but its device
// list, so communicators outlive the handles that use them: a handle takes a set from the cache (or
// creates one) and puts it back when it is destroyed.  Like the per-call contexts the cache is never
// torn down (no RCCL / HIP calls from static destructors).
struct CommSet {
    bool poisoned = false;   // an RCCL call on it has failed: never handed to another handle
    int parts = 0;
    int devs[FWX_MAX_PARTS];
    ncclComm_t comm[FWX_MAX_PARTS];
};

class CommCache {
public:
    static int acquire(int parts, const int *devs, CommSet **out)
    {
        Cache &c = cache();
        {
            std::lock_guard<std::mutex> lk(c.mu);
            for (size_t i = 0; i < c.idle.size(); ++i)
                if (c.idle[i]->parts == parts && memcmp(c.idle[i]->devs, devs, sizeof(int) * parts) == 0) {
                    *out = c.idle[i];
                    c.idle.erase(c.idle.begin() + (long)i);
                    return FWX_OK;
                }
        }
        RcclApi &api = rccl();
        if (!api.ok) return FWX_ERR_RCCL;
        CommSet *cs = new (std::nothrow) CommSet();
        if (!cs) return FWX_ERR_OOM;
        cs->parts = parts;
        memcpy(cs->devs, devs, sizeof(int) * parts);
        if (api.CommInitAll(cs->comm, parts, cs->devs) != ncclSuccess) {
            delete cs;
            return FWX_ERR_RCCL;
        }
        *out = cs;
        return FWX_OK;
    }
    static void release(CommSet *cs)
    {
        if (!cs) return;
        if (!cs->poisoned) {
            Cache &c = cache();
            std::lock_guard<std::mutex> lk(c.mu);
            if (c.idle.size() < kMaxIdle) { c.idle.push_back(cs); return; }
        }
        // not cached: more idle sets than the cache keeps, or a communicator on which a call has failed --
        // it may be in an error or half-aborted state, and handing it to the next handle with the same
        // device list would make every later create_multi / solve_multi on those devices fail too
        RcclApi &api = rccl();
        for (int p = 0; p < cs->parts; ++p) {
            if (cs->poisoned && api.CommAbort) (void)api.CommAbort(cs->comm[p]);
            else (void)api.CommDestroy(cs->comm[p]);
        }
        delete cs;
    }

private:
    static constexpr size_t kMaxIdle = 4;
    struct Cache { std::mutex mu; std::vector<CommSet *> idle; };
    static Cache &cache() { static Cache *c = new Cache(); return *c; }   // leaked on purpose
};

// ---- partitions -----------------------------------------------------------------------------------
struct Part {
    int device = 0, row0 = 0, rows = 0;    // rows [row0, row0 + rows) of the nd x nd device matrix
    void *rate = nullptr;
    int32_t *next = nullptr, *hops = nullptr;
    fwx::PathLog plog;                      // slab-local trace matrices (rows x nd), or null
    int32_t *next0 = nullptr;
    void *rate0 = nullptr;                  // kept input (fwx_matrix_keep_input)
    int32_t *hops0 = nullptr;
    // FOUR panel sets, each contiguous over the sets (set s of W at w[0] + s * 64 * nd, of Ct at
    // ct + s * 64 * ct_ld, ...): the single-pass schedules ping-pong between sets 0 and 1, the double
    // pass keeps block q in set q & 3, so that a pair of blocks (2P, 2P + 1) is one 128-pivot panel
    void *w[4] = {nullptr, nullptr, nullptr, nullptr};        // snapshot panels, 64 x nd each
    int32_t *wh[4] = {nullptr, nullptr, nullptr, nullptr};    // their hops (iff hops)
    void *ct = nullptr;                     // pivot-column snapshots, 4 x 64 x ct_ld
    int32_t *cnt = nullptr, *cht = nullptr; // their next-hops / hops
    int ct_ld = 0;
    unsigned long long *upd = nullptr;
    int *flag = nullptr;
    // the panel a slot holds right now: w[slot] / wh[slot], or -- on a handle that records for resumed
    // solves -- the block's own rows of the all-pivot arrays below (bind_slot)
    void *wp[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t *whp[4] = {nullptr, nullptr, nullptr, nullptr};
    // Resumable solves (fwx_matrix_enable_resume on a partitioned handle; Resume in fwx_internal.h holds
    // the pivots and the validity marks): checkpoints = copies of this slab's arrays at the start of a
    // checkpoint pivot; all-pivot panels = what the passes produce anyway, kept for every pivot --
    // rw[k][j] = pivot row k at time k (EVERY partition keeps all of them: it receives them anyway, and the
    // replay of a changed entry (i, j) needs rw[k][j] for all k beside its own rct[k][i]), rct[k][i] = pivot
    // column k at time k for the local rows (NaN at i == k), rcnt / rwh / rcht likewise
    struct {
        std::vector<void *> rate;
        std::vector<int32_t *> next, hops, last, at_col, at_row;
        void *rw = nullptr, *rct = nullptr;
        int32_t *rcnt = nullptr, *rwh = nullptr, *rcht = nullptr;
        int64_t *idx = nullptr;
    } R;
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t rows_done = nullptr, main_done = nullptr, panel_done = nullptr;
    hipEvent_t w_ready[4] = {nullptr, nullptr, nullptr, nullptr}, main_free[4] = {nullptr, nullptr, nullptr, nullptr};
};

// Per-step event timings of a partitioned solve (fwx_matrix_set_timing / fwx_matrix_get_timing): what
// the N > 1 benchmark line reports so that a run on hardware this build never touched explains itself
// -- is a step bound by the slab sweep (BULK) or by what the next step waits for besides it (the
// look-ahead rows, the owner's panel kernel, the exchange: the CHAIN)?  Spans are pairs of timing
// events on the stream the work runs on, created on that stream's device and reused across solves.
struct MultiTimer {
    enum Kind { BULK, CHAIN, LOOKAHEAD, PANEL, XCHG, NKIND };
    struct Span { int kind, part, step; hipEvent_t e0, e1; };
    bool on = false;
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool[FWX_MAX_PARTS];
    size_t used[FWX_MAX_PARTS] = {0};
    fwx_multi_timing last;
    void reset() { spans.clear(); for (size_t &u : used) u = 0; memset(&last, 0, sizeof(last)); }
    hipEvent_t take(int part)
    {
        if (used[part] == pool[part].size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
            pool[part].push_back(e);
        }
        return pool[part][used[part]++];
    }
    // the device of `part` must be current
    int begin(int kind, int part, int step, hipStream_t st)
    {
        if (!on) return -1;
        Span sp{kind, part, step, take(part), take(part)};
        if (!sp.e0 || !sp.e1 || hipEventRecord(sp.e0, st) != hipSuccess) { (void)hipGetLastError(); return -1; }
        spans.push_back(sp);
        return (int)spans.size() - 1;
    }
    void end(int id, hipStream_t st)
    {
        if (id >= 0 && hipEventRecord(spans[(size_t)id].e1, st) != hipSuccess) (void)hipGetLastError();
    }
};

struct MultiState {
    MultiTimer timer;
    // self >= 0: ONE partition of `parts` lives in this process (fwx_matrix_create_part: one process per
    // GPU); only part[self] has arrays, streams and events -- the others are row bounds -- and the panel
    // exchange is the host's callback (a broadcast among the processes), called once per panel in the
    // same order on every rank.  The schedules below are the same code either way.
    int self = -1;
    fwx_exchange_fn xfn = nullptr;
    void *xctx = nullptr;
    bool here(int p) const { return self < 0 || p == self; }
    int first_here() const { return self < 0 ? 0 : self; }
    int parts = 0;
    int nd = 0;                 // device order: n rounded up to a multiple of 16 bytes of elements
    int exchange = FWX_XCHG_PEER;
    Part part[FWX_MAX_PARTS];
    CommSet *comms = nullptr;      // RCCL exchange: one communicator per partition (CommCache)
    bool peer_all = true;          // every pair of distinct devices has peer access (queries walk the
                                   // slabs from partition 0's device; the exchange does not need it)
    size_t slab_bytes = 0;         // rate slabs, all partitions: decides whether a one-shot call keeps the handle
    int32_t *qscratch = nullptr;   // query scratch on partition 0's device
    int32_t qcap = 0;
};

// Everything a query kernel needs to address entry (a, b) of a partitioned matrix.
struct SlabTab {
    int parts, n;                              // n = device pitch
    int row0[FWX_MAX_PARTS + 1];
    const int32_t *next[FWX_MAX_PARTS], *last[FWX_MAX_PARTS], *at_col[FWX_MAX_PARTS],
        *at_row[FWX_MAX_PARTS], *next0[FWX_MAX_PARTS];
    __device__ __forceinline__ size_t locate(int a, int b, int &p) const
    {
        p = 0;
        while (p + 1 < parts && a >= row0[p + 1]) ++p;
        return (size_t)(a - row0[p]) * n + b;
    }
};

static SlabTab make_tab(const MultiState &M)
{
    SlabTab t;
    memset(&t, 0, sizeof(t));
    t.parts = M.parts;
    t.n = M.nd;
    for (int p = 0; p < M.parts; ++p) {
        const Part &q = M.part[p];
        t.row0[p] = q.row0;
        t.next[p] = q.next;
        t.last[p] = q.plog.last;
        t.at_col[p] = q.plog.at_col;
        t.at_row[p] = q.plog.at_row;
        t.next0[p] = q.next0;
    }
    t.row0[M.parts] = M.nd;
    return t;
}

__global__ void multi_follow_path_kernel(SlabTab t, int n_real, int src, int dst, int32_t *out, int cap,
                                         int32_t *len_out)
{
    int len = 0, cur = src, p;
    {
        const size_t off = t.locate(src, dst, p);
        if (t.next[p][off] < 0) { *len_out = 0; return; }
    }
    while (cur != dst || len == 0) {
        const size_t off = t.locate(cur, dst, p);
        const int nx = t.next[p][off];
        if (nx < 0 || nx >= n_real || len >= n_real) { *len_out = FWX_ERR_CYCLE; return; }
        if (len >= cap) { *len_out = FWX_ERR_CAPACITY; return; }
        out[len++] = nx;
        cur = nx;
    }
    *len_out = len;
}

// Same walk as exact_path(s)_kernel in fwx_api.hip (see there), entries addressed through the table.
__global__ __launch_bounds__(64) void multi_exact_paths_kernel(SlabTab t, int n_real, int count,
                                                               const int32_t *src, const int32_t *dst,
                                                               int32_t *paths, int32_t *stacks, int cap,
                                                               int32_t *len_out)
{
    enum { FINAL = 0, AS_COLUMN = 1, AS_ROW = 2 };
    const int qi = blockIdx.x * 64 + threadIdx.x;
    if (qi >= count) return;
    const int s0 = src[qi], d0 = dst[qi];
    if (s0 < 0 || d0 < 0 || s0 >= n_real || d0 >= n_real) { len_out[qi] = FWX_ERR_INVALID; return; }
    int32_t *out = paths + (size_t)qi * cap;
    int32_t *stack = stacks + (size_t)qi * 3 * cap;
    int sp = 0, len = 0;
    stack[0] = s0; stack[1] = d0; stack[2] = FINAL; sp = 1;
    while (sp > 0) {
        --sp;
        const int a = stack[3 * sp], b = stack[3 * sp + 1], kind = stack[3 * sp + 2];
        int p;
        const size_t off = t.locate(a, b, p);
        const int q = kind == FINAL ? t.last[p][off] : kind == AS_COLUMN ? t.at_col[p][off] : t.at_row[p][off];
        if (q < 0) {
            if (t.next0[p][off] >= 0) {
                if (len >= cap) { len_out[qi] = FWX_ERR_CAPACITY; return; }
                out[len++] = b;
            }
        } else {
            if (sp + 2 > cap) { len_out[qi] = FWX_ERR_CAPACITY; return; }
            stack[3 * sp] = q; stack[3 * sp + 1] = b; stack[3 * sp + 2] = AS_ROW; ++sp;
            stack[3 * sp] = a; stack[3 * sp + 1] = q; stack[3 * sp + 2] = AS_COLUMN; ++sp;
        }
    }
    len_out[qi] = len;
}

static int set_dev(int d)
{
    FWX_HIP(hipSetDevice(d));
    return FWX_OK;
}

// Restores the caller's device when a multi call returns.
struct DevRestore {
    int prev = -1;
    DevRestore() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DevRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// (the partition's device must be current)
static void part_resume_free(Part &q)
{
    auto drop = [](void *p) { if (p) (void)hipFree(p); };
    for (void *p : q.R.rate) drop(p);
    for (auto *v : {&q.R.next, &q.R.hops, &q.R.last, &q.R.at_col, &q.R.at_row}) {
        for (int32_t *p : *v) drop(p);
        v->clear();
    }
    q.R.rate.clear();
    drop(q.R.rw); drop(q.R.rct); drop(q.R.rcnt); drop(q.R.rwh); drop(q.R.rcht); drop(q.R.idx);
    q.R.rw = q.R.rct = nullptr;
    q.R.rcnt = q.R.rwh = q.R.rcht = nullptr;
    q.R.idx = nullptr;
}

static void multi_free(MultiState *M)
{
    if (!M) return;
    DevRestore keep;
    // order: retire every command that used the partitions' arrays (drain_stream in fwx_internal.h),
    // give the communicators back, destroy the streams and events, and only then free the memory
    for (int p = 0; p < M->parts; ++p) {
        Part &q = M->part[p];
        if (hipSetDevice(q.device) != hipSuccess) continue;
        if (q.main) drain_stream(q.main);
        if (q.side) drain_stream(q.side);
    }
    CommCache::release(M->comms);
    M->comms = nullptr;
    for (int p = 0; p < M->parts; ++p) {
        Part &q = M->part[p];
        if (hipSetDevice(q.device) != hipSuccess) continue;
        if (q.main) (void)hipStreamDestroy(q.main);
        if (q.side) (void)hipStreamDestroy(q.side);
        for (hipEvent_t e : M->timer.pool[p]) (void)hipEventDestroy(e);
        hipEvent_t evs[] = {q.rows_done, q.main_done, q.panel_done, q.w_ready[0], q.w_ready[1], q.w_ready[2],
                            q.w_ready[3], q.main_free[0], q.main_free[1], q.main_free[2], q.main_free[3]};
        for (hipEvent_t e : evs)
            if (e) (void)hipEventDestroy(e);
        void *bufs[] = {q.rate, q.next, q.hops, q.plog.last, q.plog.at_col, q.plog.at_row, q.next0, q.rate0,
                        q.hops0, q.w[0], q.wh[0], q.ct, q.cnt, q.cht, q.upd, q.flag};
        for (void *b : bufs)
            if (b) (void)hipFree(b);
        part_resume_free(q);
        if (p == 0 && M->qscratch) (void)hipFree(M->qscratch);
    }
    delete M;
}

// First row of partition p of P: balanced row blocks, n * p / P -- rounded down to a multiple of 64 once every
// partition holds at least two blocks (n >= 128 P).  A pivot block never straddles two partitions, so with
// aligned partitions every block is a full, aligned 64 whatever n is: the pair schedule (128 pivots per main
// launch) and the checkpoints of resumable solves (block starts) then apply to ANY matrix order, not only to
// multiples of 64 P.  The imbalance is below one block per partition.  fwx_matrix_part_rows reports the bounds.
static int part_bound(int n, int parts, int p)
{
    if (p >= parts) return n;
    const int b = (int)((int64_t)n * p / parts);
    return n >= 2 * FWX_FUSED_BLOCK * parts ? b / FWX_FUSED_BLOCK * FWX_FUSED_BLOCK : b;
}

static int multi_alloc(fwx_matrix *m, int n_parts, const int32_t *devices, int exchange, int self = -1,
                       fwx_exchange_fn xfn = nullptr, void *xctx = nullptr)
{
    const size_t es = m->dtype == FWX_F64 ? 8 : 4;
    const int vw = (int)(16 / es);
    MultiState *M = new (std::nothrow) MultiState();
    if (!M) return FWX_ERR_OOM;
    m->multi = M;
    M->parts = n_parts;
    M->self = self;
    M->xfn = xfn;
    M->xctx = xctx;
    M->nd = (m->n + vw - 1) / vw * vw;
    bool distinct = true;
    for (int p = 0; p < n_parts; ++p)
        for (int q = 0; q < p; ++q) distinct = distinct && devices[p] != devices[q];
    if (exchange == FWX_XCHG_AUTO) exchange = (distinct && n_parts >= 2) ? FWX_XCHG_RCCL : FWX_XCHG_PEER;
    if (exchange == FWX_XCHG_RCCL && !distinct) return FWX_ERR_INVALID;
    if ((exchange == FWX_XCHG_CALLBACK) != (M->self >= 0)) return FWX_ERR_INVALID;
    M->exchange = exchange;
    const int nd = M->nd;
    for (int p = 0; p < n_parts; ++p) {
        Part &q = M->part[p];
        q.device = devices[p];
        q.row0 = part_bound(m->n, n_parts, p);
        const int r1 = p + 1 == n_parts ? nd : part_bound(m->n, n_parts, p + 1);   // padding rows: last
        q.rows = r1 - q.row0;
        q.ct_ld = (q.rows + 3) & ~3;
        if (!M->here(p)) continue;                 // (another process holds it)
        int rc = set_dev(q.device);
        if (rc) return rc;
        const size_t cells = (size_t)q.rows * nd;
        M->slab_bytes += cells * es;
        FWX_HIP(hipMalloc(&q.rate, cells * es ? cells * es : 16));
        if (m->next) FWX_HIP(hipMalloc((void **)&q.next, cells * 4 ? cells * 4 : 16));
        const size_t wset = (size_t)FWX_FUSED_BLOCK * nd, cset = (size_t)FWX_FUSED_BLOCK * (q.ct_ld ? q.ct_ld : 4);
        FWX_HIP(hipMalloc(&q.w[0], 4 * wset * es));
        for (int b = 1; b < 4; ++b) q.w[b] = (char *)q.w[0] + (size_t)b * wset * es;
        for (int b = 0; b < 4; ++b) q.wp[b] = q.w[b];
        FWX_HIP(hipMalloc(&q.ct, 4 * cset * es));
        if (m->next) FWX_HIP(hipMalloc((void **)&q.cnt, 4 * cset * 4));
        if (m->hops) {
            FWX_HIP(hipMalloc((void **)&q.hops, cells * 4 ? cells * 4 : 16));
            FWX_HIP(hipMalloc((void **)&q.wh[0], 4 * wset * 4));
            for (int b = 1; b < 4; ++b) q.wh[b] = q.wh[0] + (size_t)b * wset;
            for (int b = 0; b < 4; ++b) q.whp[b] = q.wh[b];
            FWX_HIP(hipMalloc((void **)&q.cht, 4 * cset * 4));
        }
        FWX_HIP(hipMalloc((void **)&q.upd, FWX_UPDATE_SHARDS * 8));
        FWX_HIP(hipMalloc((void **)&q.flag, 16));
        FWX_HIP(hipStreamCreateWithFlags(&q.main, hipStreamNonBlocking));
        FWX_HIP(hipStreamCreateWithFlags(&q.side, hipStreamNonBlocking));
        hipEvent_t *evs[] = {&q.rows_done, &q.main_done, &q.panel_done, &q.w_ready[0], &q.w_ready[1], &q.w_ready[2],
                             &q.w_ready[3], &q.main_free[0], &q.main_free[1], &q.main_free[2], &q.main_free[3]};
        for (hipEvent_t *e : evs) FWX_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    // Peer access is wanted, not required: the QUERIES walk every slab from partition 0's device and
    // need it (they say so when it is missing); the exchange does not -- RCCL has its own transports
    // and hipMemcpyPeerAsync stages through the host when two devices are not peers.
    for (int p = 0; p < n_parts && M->self < 0; ++p)
        for (int q = 0; q < n_parts; ++q) {
            if (devices[p] == devices[q]) continue;
            int rc = set_dev(devices[p]);
            if (rc) return rc;
            const hipError_t e = hipDeviceEnablePeerAccess(devices[q], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) M->peer_all = false;
            (void)hipGetLastError();
        }
    if (exchange == FWX_XCHG_RCCL) {
        int devs[FWX_MAX_PARTS];
        for (int p = 0; p < n_parts; ++p) devs[p] = devices[p];
        const int rc = CommCache::acquire(n_parts, devs, &M->comms);
        if (rc) return rc;
    }
    // placeholders: the single-device code paths test m->next for "carries next-hops"
    return FWX_OK;
}

// The arrays of partition p hold rows [row0, row0 + rows) at pitch nd; the caller's are n x n.
static int multi_copy(fwx_matrix *m, void *host_rate, int32_t *host_next, int32_t *host_hops,
                      bool to_device)
{
    MultiState &M = *m->multi;
    const size_t es = m->dtype == FWX_F64 ? 8 : 4;
    const int n = m->n, nd = M.nd;
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        int rc = set_dev(q.device);
        if (rc) return rc;
        const int real = (q.row0 + q.rows <= n ? q.rows : n - q.row0);   // rows that exist in the caller's arrays
        if (to_device && nd != n) {
            FWX_HIP(hipMemsetAsync(q.rate, 0, (size_t)q.rows * nd * es, q.main));              // +0.0
            if (q.next) FWX_HIP(hipMemsetAsync(q.next, 0xFF, (size_t)q.rows * nd * 4, q.main));   // -1
            if (q.hops) FWX_HIP(hipMemsetAsync(q.hops, 0, (size_t)q.rows * nd * 4, q.main));
        }
        if (real <= 0) continue;
        auto copy = [&](void *dev, char *host, size_t e) -> int {
            if (!host) return FWX_OK;
            if (M.self < 0) host += (size_t)q.row0 * n * e;      // (one partition per process: the caller's arrays ARE the slab)
            if (to_device)
                FWX_HIP(hipMemcpy2DAsync(dev, (size_t)nd * e, host, (size_t)n * e, (size_t)n * e, (size_t)real,
                                         hipMemcpyDefault, q.main));
            else
                FWX_HIP(hipMemcpy2DAsync(host, (size_t)n * e, dev, (size_t)nd * e, (size_t)n * e, (size_t)real,
                                         hipMemcpyDefault, q.main));
            return FWX_OK;
        };
        if ((rc = copy(q.rate, (char *)host_rate, es))) return rc;
        if (q.next && (rc = copy(q.next, (char *)host_next, 4))) return rc;
        if (q.hops && (rc = copy(q.hops, (char *)host_hops, 4))) return rc;
        if (to_device && q.next0)
            FWX_HIP(hipMemcpyAsync(q.next0, q.next, (size_t)q.rows * nd * 4, hipMemcpyDeviceToDevice, q.main));
        if (to_device && q.rate0) {
            FWX_HIP(hipMemcpyAsync(q.rate0, q.rate, (size_t)q.rows * nd * es, hipMemcpyDeviceToDevice, q.main));
            if (q.hops0)
                FWX_HIP(hipMemcpyAsync(q.hops0, q.hops, (size_t)q.rows * nd * 4, hipMemcpyDeviceToDevice, q.main));
        }
    }
    for (int p = 0; p < M.parts; ++p) {
        if (!M.here(p)) continue;
        int rc = set_dev(M.part[p].device);
        if (rc) return rc;
        FWX_HIP(hipStreamSynchronize(M.part[p].main));
    }
    return FWX_OK;
}

// One host thread per partition for the per-k engine's bulk launches.  A partitioned per-k solve issues
// n launches PER PARTITION; from a single thread (~3.5 us each) that is 8 x 16384 x 3.5 us = 460 ms
// for N = 16384 on 8 devices against 345 ms of kernel time per device: the host would bound the solve.
// The orchestration (events, panels, exchange) stays on the calling thread; per block it hands every
// partition's 64 relax_k launches to that partition's worker, waits until all of them are ENQUEUED
// (not executed), and carries on -- so the stream order each partition sees is unchanged.
class SweepWorkers {
public:
    explicit SweepWorkers(int n) : jobs_((size_t)n), rc_((size_t)n, FWX_OK)
    {
        threads_.reserve((size_t)n);
        for (int p = 0; p < n; ++p) threads_.emplace_back([this, p] { loop(p); });
    }
    ~SweepWorkers()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            quit_ = true;
            ++round_;
        }
        cv_.notify_all();
        for (std::thread &t : threads_) t.join();
    }
    void set(int p, std::function<int()> job) { jobs_[(size_t)p] = std::move(job); }
    // runs every job that was set since the last call, each on its own thread; first error wins
    int run_all()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            pending_ = (int)threads_.size();
            ++round_;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        int rc = FWX_OK;
        for (size_t p = 0; p < rc_.size(); ++p) {
            if (rc_[p] && !rc) rc = rc_[p];
            rc_[p] = FWX_OK;
            jobs_[p] = nullptr;
        }
        return rc;
    }

private:
    void loop(int p)
    {
        unsigned long long seen = 0;
        for (;;) {
            std::function<int()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return round_ != seen; });
                seen = round_;
                if (quit_) return;
                job = jobs_[(size_t)p];
            }
            int rc = FWX_OK;
            if (job) {
                try { rc = job(); } catch (...) { rc = FWX_ERR_INTERNAL; }
            }
            {
                std::lock_guard<std::mutex> lk(mu_);
                rc_[(size_t)p] = rc;
                if (--pending_ == 0) done_.notify_one();
            }
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::vector<std::thread> threads_;
    std::vector<std::function<int()>> jobs_;
    std::vector<int> rc_;
    unsigned long long round_ = 0;
    int pending_ = 0;
    bool quit_ = false;
};

struct Block { int k0, bt, owner; };

template <typename T> static fwx::FusedArgs<T> part_args(const MultiState &M, const Part &q, bool nonneg,
                                                         bool counting)
{
    fwx::FusedArgs<T> a;
    a.rate = (T *)q.rate; a.next = q.next; a.rows = q.rows; a.n = M.nd; a.row0 = q.row0;
    a.ct = (T *)q.ct; a.cnt = q.next ? q.cnt : nullptr; a.ct_ld = q.ct_ld;
    a.updates = counting ? q.upd : nullptr; a.nonneg = nonneg; a.plog = q.plog;
    a.hops = q.hops; a.cht = q.cht;
    return a;
}

// Snapshot panel of block b on its owner's side stream + its exchange into slot b & 1 of every
// partition.  Precondition: the owner's main stream has recorded rows_done after bringing the
// block's rows up to time k0.
template <typename T> static int issue_panel(MultiState &M, const Block &blk, int slot, int step = 0)
{
    Part &o = M.part[blk.owner];
    MultiTimer &tm = M.timer;
    const size_t bytes = (size_t)blk.bt * M.nd * sizeof(T);
    const size_t hbytes = (size_t)blk.bt * M.nd * sizeof(int32_t);
    int rc;
    if (M.here(blk.owner)) {
        if ((rc = set_dev(o.device))) return rc;
        FWX_HIP(hipStreamWaitEvent(o.side, o.rows_done, 0));
        FWX_HIP(hipStreamWaitEvent(o.side, o.main_free[slot], 0));       // own main kernels are done with this slot
        if (M.exchange == FWX_XCHG_PEER)
            for (int r = 0; r < M.parts; ++r)      // nobody is still copying the previous panel out of this slot
                if (r != blk.owner) FWX_HIP(hipStreamWaitEvent(o.side, M.part[r].w_ready[slot], 0));
        const size_t row_off = (size_t)(blk.k0 - o.row0) * M.nd;
        const int t_panel = tm.begin(MultiTimer::PANEL, blk.owner, step, o.side);
        FWX_HIP(fwx::launch_fused_panel<T>((const T *)o.rate + row_off, M.nd, blk.k0, blk.bt, (T *)o.wp[slot],
                                           o.side, plog_rows(o.plog, row_off),
                                           o.hops ? o.hops + row_off : nullptr, o.whp[slot]));
        tm.end(t_panel, o.side);
        FWX_HIP(hipEventRecord(o.w_ready[slot], o.side));
    }
    if (M.exchange == FWX_XCHG_CALLBACK) {
        // one partition per process: the host broadcasts the panel among the processes, ordered on this
        // partition's side stream (behind the panel kernel on the owner, behind the last reader of the
        // slot elsewhere); every rank gets here for every panel, in the same order
        Part &q = M.part[M.self];
        if ((rc = set_dev(q.device))) return rc;
        if (M.self != blk.owner) FWX_HIP(hipStreamWaitEvent(q.side, q.main_free[slot], 0));
        const int t_x = tm.begin(MultiTimer::XCHG, M.self, step, q.side);
        if (M.xfn(M.xctx, blk.k0, blk.bt, blk.owner, q.wp[slot], q.hops ? q.whp[slot] : nullptr,
                  (int64_t)blk.bt * M.nd, q.side) != 0)
            return FWX_ERR_RCCL;
        tm.end(t_x, q.side);
        FWX_HIP(hipEventRecord(q.w_ready[slot], q.side));
    } else if (M.exchange == FWX_XCHG_PEER) {
        for (int r = 0; r < M.parts; ++r) {
            if (r == blk.owner) continue;
            Part &q = M.part[r];
            if ((rc = set_dev(q.device))) return rc;
            FWX_HIP(hipStreamWaitEvent(q.side, q.main_free[slot], 0));
            FWX_HIP(hipStreamWaitEvent(q.side, o.w_ready[slot], 0));
            const int t_x = tm.begin(MultiTimer::XCHG, r, step, q.side);
            if (q.device == o.device) {
                FWX_HIP(hipMemcpyAsync(q.wp[slot], o.wp[slot], bytes, hipMemcpyDeviceToDevice, q.side));
                if (o.hops)
                    FWX_HIP(hipMemcpyAsync(q.whp[slot], o.whp[slot], hbytes, hipMemcpyDeviceToDevice, q.side));
            } else {
                FWX_HIP(hipMemcpyPeerAsync(q.wp[slot], q.device, o.wp[slot], o.device, bytes, q.side));
                if (o.hops)
                    FWX_HIP(hipMemcpyPeerAsync(q.whp[slot], q.device, o.whp[slot], o.device, hbytes, q.side));
            }
            tm.end(t_x, q.side);
            FWX_HIP(hipEventRecord(q.w_ready[slot], q.side));
        }
    } else {
        RcclApi &api = rccl();
        for (int r = 0; r < M.parts; ++r) {
            if (r == blk.owner) continue;
            if ((rc = set_dev(M.part[r].device))) return rc;
            FWX_HIP(hipStreamWaitEvent(M.part[r].side, M.part[r].main_free[slot], 0));
        }
        int t_x[FWX_MAX_PARTS];
        for (int r = 0; r < M.parts; ++r) {
            if ((rc = set_dev(M.part[r].device))) return rc;
            t_x[r] = tm.begin(MultiTimer::XCHG, r, step, M.part[r].side);
        }
        FWX_NCCL(api.GroupStart());
        for (int r = 0; r < M.parts; ++r) {
            Part &q = M.part[r];
            if ((rc = set_dev(q.device))) return rc;
            FWX_NCCL(api.Broadcast(q.wp[slot], q.wp[slot], (size_t)blk.bt * M.nd,
                                   sizeof(T) == 8 ? ncclFloat64 : ncclFloat32, blk.owner, M.comms->comm[r], q.side));
            if (q.hops)     // the hops of the pivot rows travel with their rates
                FWX_NCCL(api.Broadcast(q.whp[slot], q.whp[slot], (size_t)blk.bt * M.nd, ncclInt32, blk.owner,
                                       M.comms->comm[r], q.side));
        }
        FWX_NCCL(api.GroupEnd());
        for (int r = 0; r < M.parts; ++r) {
            Part &q = M.part[r];
            if ((rc = set_dev(q.device))) return rc;
            tm.end(t_x[r], q.side);
            FWX_HIP(hipEventRecord(q.w_ready[slot], q.side));
        }
    }
    return FWX_OK;
}

// Turns the spans of the solve that has just been synchronised into fwx_multi_timing.
static void summarize_timing(MultiState &M)
{
    MultiTimer &tm = M.timer;
    fwx_multi_timing &t = tm.last;
    t.struct_size = (uint32_t)sizeof(t);
    t.partitions = M.parts;
    if (!tm.on || tm.spans.empty()) return;
    // per kind: mean over steps of the MAX over partitions (what a step waits for), and the plain mean
    int steps = 0;
    for (const MultiTimer::Span &sp : tm.spans) steps = sp.step + 1 > steps ? sp.step + 1 : steps;
    std::vector<float> mx((size_t)steps * MultiTimer::NKIND, -1.0f);
    double sum[MultiTimer::NKIND] = {0};
    long cnt[MultiTimer::NKIND] = {0};
    for (const MultiTimer::Span &sp : tm.spans) {
        if (hipSetDevice(M.part[sp.part].device) != hipSuccess) continue;
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.e0, sp.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        float &slot = mx[(size_t)sp.step * MultiTimer::NKIND + sp.kind];
        slot = ms > slot ? ms : slot;
        sum[sp.kind] += ms;
        ++cnt[sp.kind];
    }
    auto mean_of_max = [&](int kind) -> float {
        double a = 0; long c = 0;
        for (int st = 0; st < steps; ++st) {
            const float v = mx[(size_t)st * MultiTimer::NKIND + kind];
            if (v >= 0) { a += v; ++c; }
        }
        return c ? (float)(1e3 * a / c) : 0.0f;
    };
    auto mean = [&](int kind) -> float { return cnt[kind] ? (float)(1e3 * sum[kind] / cnt[kind]) : 0.0f; };
    const int held = M.self >= 0 ? 1 : (M.parts > 0 ? M.parts : 1);      // partitions whose spans this process has
    t.steps = (int32_t)cnt[MultiTimer::BULK] / held;
    t.bulk_us = mean_of_max(MultiTimer::BULK);
    t.bulk_mean_us = mean(MultiTimer::BULK);
    t.lookahead_us = mean(MultiTimer::LOOKAHEAD);
    t.panel_us = mean(MultiTimer::PANEL);
    t.exchange_us = mean_of_max(MultiTimer::XCHG);
    // single pass: the serial chain of a step is look-ahead rows -> panel kernel -> exchange; the double
    // pass measures it whole, on the side stream (CHAIN), from the moment the previous main launch ends
    t.chain_us = cnt[MultiTimer::CHAIN] ? mean_of_max(MultiTimer::CHAIN)
                                        : t.lookahead_us + t.panel_us + t.exchange_us;
    t.chain_over_bulk = t.bulk_us > 0 ? t.chain_us / t.bulk_us : 0.0f;
}

static int finish_multi_solve(fwx_matrix *m, bool counting, const Opts &op)
{
    MultiState &M = *m->multi;
    int rc;
    uint64_t total = 0;
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        if ((rc = set_dev(q.device))) return rc;
        FWX_HIP(hipStreamSynchronize(q.side));
        FWX_HIP(hipStreamSynchronize(q.main));
        if (counting) {
            uint64_t u = 0;
            if ((rc = sum_updates(q.upd, &u, q.main))) return rc;
            total += u;
        }
    }
    summarize_timing(M);
    if (counting) *op.updates_out = total;
    m->last_u = total;
    return FWX_OK;
}

// crossover thresholds of the double pass: those of the single-device engine (fwx_api.hip fused_range),
// same environment overrides
static int env_threshold_multi(const char *name, int dflt)
{
    const char *e = getenv(name);
    if (e && *e) {
        char *end = nullptr;
        const long v = strtol(e, &end, 10);
        if (end != e && v >= 0 && v <= INT32_MAX) return (int)v;
    }
    return dflt;
}

// The double pass on partitions: fused_range's schedule (fwx_api.hip) with owners and an exchange.
// Blocks q = 0, 1, ... of 64 pivots; block q lives in panel set q & 3 on EVERY partition, so a pair
// (2P, 2P + 1) is one contiguous 128-pivot panel and every partition's main kernel applies a pair per
// launch.  Beside main(P) -- which leaves the rows and the columns of the NEXT pair alone -- every
// partition's side stream brings its part of that cross up to date with the pair being applied (its
// local rows of those columns; on the owners also the cross rows themselves), the owner of the first
// block of the next pair runs its snapshot panel and sends it, every partition forms its column panel
// from it, applies that one pass to the second block's columns (its owner: and rows), and the same
// again for the second block.  main(P + 1) waits for its own partition's side chain only.  Nothing
// orders two partitions but the panels: each exchange has a whole pair's sweep to arrive, twice the
// single pass's.  Same kernels, operands and order per entry as on one device: bit-identical.
// Buffer reuse: set s is overwritten every fourth block; a partition's main(P - 1) -- the last reader
// of the sets chain(P) writes -- precedes its chain(P) in stream order (main_done), and an owner waits
// for every peer's w_ready[s] before it overwrites a panel peers may still be copying (PEER exchange).
// done: the number of blocks applied (a multiple of 2 ... or all of them if the count is odd).
template <typename T>
static int multi_double_pass(fwx_matrix *m, const std::vector<Block> &blocks, Throttle &thr, int &done)
{
    MultiState &M = *m->multi;
    const int P = M.parts, nd = M.nd;
    constexpr int Bq = FWX_FUSED_BLOCK;
    const bool with_next = m->next != nullptr;
    done = 0;
    // full 64-aligned blocks from the start of the range; every partition non-empty and 64-aligned
    int nb = 0;
    while ((size_t)nb < blocks.size() && blocks[(size_t)nb].bt == Bq && blocks[(size_t)nb].k0 % Bq == 0) ++nb;
    for (int p = 0; p < P; ++p)
        if (M.part[p].rows <= 0 || M.part[p].row0 % Bq != 0) return FWX_OK;
    const int min_n = with_next ? env_threshold_multi("FWX_DOUBLE_PASS_NEXT_MIN_N", sizeof(T) == 4 ? 8192 : INT32_MAX)
                                : env_threshold_multi("FWX_DOUBLE_PASS_MIN_N", 6144);
    if (m->n < min_n || nb < 4) return FWX_OK;
    int rc;
    MultiTimer &tm = M.timer;
    tm.last.pivots_per_step = 2 * Bq;
    auto kq = [&](int q) { return blocks[(size_t)q].k0; };
    auto args = [&](Part &q, int blk, int nblocks) {
        fwx::FusedArgs<T> a = part_args<T>(M, q, true, false);
        const size_t set = (size_t)(blk & 3);
        a.k0 = kq(blk); a.bt = nblocks * Bq;
        a.w = (const T *)q.w[0] + set * Bq * nd;
        a.wh = q.wh[0] ? q.wh[0] + set * Bq * nd : nullptr;
        a.ct = (T *)q.ct + set * Bq * q.ct_ld;
        a.cnt = q.cnt ? q.cnt + set * Bq * q.ct_ld : nullptr;
        a.cht = q.cht ? q.cht + set * Bq * q.ct_ld : nullptr;
        return a;
    };
    auto local = [&](const Part &q, int g) { const int l = g - q.row0; return l < 0 ? 0 : l > q.rows ? q.rows : l; };
    // pivots of `nblocks` blocks from block blk onto the rows [lo, hi) this partition holds (all columns)
    // and onto the columns [lo, hi) of its other rows
    auto cross = [&](Part &q, int blk, int nblocks, int lo, int hi, hipStream_t st) -> int {
        fwx::FusedArgs<T> a = args(q, blk, nblocks);
        a.side = true;
        const int l_lo = local(q, lo), l_hi = local(q, hi);
        if (l_hi > l_lo) FWX_HIP(fwx::launch_fused_main<T>(a, l_lo, l_hi, st));
        FWX_HIP(fwx::launch_fused_main<T>(a, 0, q.rows, st, l_lo, l_hi, fwx::FusedCols::only(lo, hi)));
        return FWX_OK;
    };
    // snapshot panel of block blk on its owner + exchange, then every partition's column panel, all on
    // the side streams
    auto produce = [&](int blk, int step) -> int {
        int rc2 = issue_panel<T>(M, blocks[(size_t)blk], blk & 3, step);
        if (rc2) return rc2;
        for (int p = 0; p < P; ++p) {
            Part &q = M.part[p];
            if (!M.here(p)) continue;
            if ((rc2 = set_dev(q.device))) return rc2;
            FWX_HIP(hipStreamWaitEvent(q.side, q.w_ready[blk & 3], 0));
            FWX_HIP(fwx::launch_fused_colpanel<T>(args(q, blk, 1), q.side));
        }
        return FWX_OK;
    };
    auto each = [&](auto &&fn) -> int {
        for (int p = 0; p < P; ++p) {
            if (!M.here(p)) continue;
            int rc2 = set_dev(M.part[p].device);
            if (rc2 || (rc2 = fn(p, M.part[p]))) return rc2;
        }
        return FWX_OK;
    };
    // chain(0): panels of block 0, that pass onto block 1's rows and columns, panels of block 1 -- on the
    // side streams, behind everything the main streams hold so far
    if ((rc = each([&](int, Part &q) -> int {
            FWX_HIP(hipEventRecord(q.main_done, q.main));
            FWX_HIP(hipStreamWaitEvent(q.side, q.main_done, 0));
            // (issue_panel orders the owner's panel behind rows_done / main_free: both on the main stream)
            FWX_HIP(hipEventRecord(q.rows_done, q.main));
            return FWX_OK;
        })))
        return rc;
    if ((rc = produce(0, 0))) return rc;
    if ((rc = each([&](int, Part &q) -> int { return cross(q, 0, 1, kq(1), kq(1) + Bq, q.side); }))) return rc;
    if ((rc = produce(1, 0))) return rc;
    if ((rc = each([&](int, Part &q) -> int {
            FWX_HIP(hipEventRecord(q.panel_done, q.side));
            FWX_HIP(hipStreamWaitEvent(q.main, q.panel_done, 0));
            return FWX_OK;
        })))
        return rc;
    const int pairs = nb / 2;
    int t_chain[FWX_MAX_PARTS];
    for (int pr = 0; pr < pairs; ++pr) {
        const int q0 = 2 * pr;
        const int q_lo = q0 + 2, q_hi = q0 + 4 < nb ? q0 + 4 : nb;       // blocks of the next pair (or the odd last one)
        if (q_hi > q_lo) {
            const int x_lo = kq(q_lo), x_hi = kq(q_hi - 1) + Bq;
            if ((rc = each([&](int p, Part &q) -> int {
                    FWX_HIP(hipEventRecord(q.main_done, q.main));          // main(pr - 1) and chain(pr) precede
                    FWX_HIP(hipStreamWaitEvent(q.side, q.main_done, 0));
                    t_chain[p] = tm.begin(MultiTimer::CHAIN, p, pr, q.side);
                    return cross(q, q0, 2, x_lo, x_hi, q.side);            // the pair being applied onto the next cross
                })))
                return rc;
            if ((rc = produce(q_lo, pr))) return rc;
            if (q_hi - q_lo == 2) {
                if ((rc = each([&](int, Part &q) -> int { return cross(q, q_lo, 1, kq(q_lo + 1), x_hi, q.side); })))
                    return rc;
                if ((rc = produce(q_lo + 1, pr))) return rc;
            }
            if ((rc = each([&](int p, Part &q) -> int {
                    tm.end(t_chain[p], q.side);
                    FWX_HIP(hipEventRecord(q.panel_done, q.side));
                    const int t_bulk = tm.begin(MultiTimer::BULK, p, pr, q.main);
                    // (two halves where a retiring main workgroup leaves no room for a panel workgroup: fused_range)
                    const fwx::FusedArgs<T> am = args(q, q0, 2);
                    const int h = q.rows / 2 / 128 * 128;
                    if (h > 0 && fwx::fused_main_starves_panels<T>(am)) {
                        FWX_HIP(fwx::launch_fused_main<T>(am, 0, h, q.main, local(q, x_lo), local(q, x_hi),
                                                          fwx::FusedCols::except(x_lo, x_hi)));
                        FWX_HIP(fwx::launch_fused_main<T>(am, h, q.rows, q.main, local(q, x_lo), local(q, x_hi),
                                                          fwx::FusedCols::except(x_lo, x_hi)));
                    } else {
                        FWX_HIP(fwx::launch_fused_main<T>(am, 0, q.rows, q.main, local(q, x_lo), local(q, x_hi),
                                                          fwx::FusedCols::except(x_lo, x_hi)));
                    }
                    tm.end(t_bulk, q.main);
                    FWX_HIP(hipStreamWaitEvent(q.main, q.panel_done, 0));
                    return FWX_OK;
                })))
                return rc;
        } else {
            if ((rc = each([&](int p, Part &q) -> int {
                    const int t_bulk = tm.begin(MultiTimer::BULK, p, pr, q.main);
                    FWX_HIP(fwx::launch_fused_main<T>(args(q, q0, 2), 0, q.rows, q.main));
                    tm.end(t_bulk, q.main);
                    return FWX_OK;
                })))
                return rc;
        }
        if ((rc = set_dev(M.part[M.first_here()].device))) return rc;
        if ((rc = thr.tick(M.part[M.first_here()].main, 10))) return rc;
    }
    if (nb & 1) {                                       // the odd last block: its panels are ready
        if ((rc = each([&](int, Part &q) -> int {
                FWX_HIP(fwx::launch_fused_main<T>(args(q, nb - 1, 1), 0, q.rows, q.main));
                return FWX_OK;
            })))
            return rc;
    }
    // the single-pass loop (a ragged tail) and the final synchronisation follow in stream order: every
    // main stream has waited for its side chain; the sets 0 / 1 it may reuse are guarded by w_ready
    if ((rc = each([&](int, Part &q) -> int {
            for (int sl = 0; sl < 4; ++sl) FWX_HIP(hipEventRecord(q.main_free[sl], q.main));
            return FWX_OK;
        })))
        return rc;
    done = nb;
    return FWX_OK;
}

// resumed: the slabs AND their trace hold a restored checkpoint at time op.k_begin (multi_resolve)
template <typename T> static int multi_solve_typed(fwx_matrix *m, const Opts &op, bool resumed = false)
{
    MultiState &M = *m->multi;
    const int nd = M.nd, P = M.parts;
    const bool counting = op.updates_out != nullptr;
    const bool with_next = m->next != nullptr;
    const bool perk = op.engine == FWX_ENGINE_PERK;
    int rc;
    // domain (fwx.h "Domain"), every slab; the handle remembers the answer for what its arrays hold (the
    // domain is closed under the algorithm: only an upload or a patch outside it can change the answer)
    int bits = 3;
    if (m->dom_known) {
        bits = m->dom_bits;
    } else if (M.self >= 0) {
        return FWX_ERR_INVALID;     // one partition per process: the host combines the ranks' bits (fwx_matrix_set_domain)
    } else {
        for (int p = 0; p < P; ++p) {
            Part &q = M.part[p];
            if (q.rows == 0) continue;
            if ((rc = set_dev(q.device))) return rc;
            int b = 3;
            if ((rc = domain_bits<T>((const T *)q.rate, q.next, (size_t)q.rows * nd, q.flag, q.main, b))) return rc;
            bits &= b;
        }
        m->dom_bits = bits;
        m->dom_known = 1;
    }
    if (with_next && bits != 3) return FWX_ERR_UNSUPPORTED;   // see fwx.h: solved on one device
    const bool nonneg = !counting && (with_next ? bits == 3 : (bits & 1) != 0);   // max-form kernels
    // A resumable handle records the panels of every pass and the checkpoints it walks over -- if this
    // solve continues the kept input's own solve (the slabs are that input at time k_begin) on the fused
    // engine inside the domain; what it holds beyond k_begin belongs to an older solve until this one ends
    Resume *rec = nullptr;
    if (m->resume) {
        Resume &R = *m->resume;
        const bool chain = !perk && m->kept_valid && R.state_at == op.k_begin && op.k_begin % FWX_FUSED_BLOCK == 0 &&
                           op.k_begin <= R.valid_upto && (with_next ? bits == 3 : (bits & 1) != 0);
        R.valid_upto = chain ? op.k_begin : 0;
        R.state_at = -1;
        if (chain) rec = &R;
    }
    for (int p = 0; p < P; ++p) {
        Part &q = M.part[p];
        for (int sl = 0; sl < 4; ++sl) { q.wp[sl] = q.w[sl]; q.whp[sl] = q.wh[sl]; }
    }
    auto here = [&](int p) { return M.here(p); };
    // pass (k0, ...) reads / writes its own rows of the all-pivot arrays on a recording handle
    auto bind_slot = [&](const Block &blk, int slot) {
        if (!rec) return;
        for (int p = 0; p < P; ++p) {
            Part &q = M.part[p];
            if (!here(p)) continue;
            q.wp[slot] = (char *)q.R.rw + (size_t)blk.k0 * nd * sizeof(T);
            q.whp[slot] = q.R.rwh ? q.R.rwh + (size_t)blk.k0 * nd : nullptr;
        }
    };
    auto bind_cols = [&](fwx::FusedArgs<T> &a, const Part &q, const Block &blk) {
        if (!rec) return;
        a.ct = (T *)q.R.rct + (size_t)blk.k0 * q.ct_ld;
        a.cnt = q.R.rcnt ? q.R.rcnt + (size_t)blk.k0 * q.ct_ld : nullptr;
        a.cht = q.R.rcht ? q.R.rcht + (size_t)blk.k0 * q.ct_ld : nullptr;
    };
    for (int p = 0; p < P; ++p) {
        Part &q = M.part[p];
        if (!here(p)) continue;
        if ((rc = set_dev(q.device))) return rc;
        const size_t cells = (size_t)q.rows * nd;
        if (q.plog.last && !resumed) {
            FWX_HIP(hipMemsetAsync(q.plog.last, 0xFF, cells * 4, q.main));
            FWX_HIP(hipMemsetAsync(q.plog.at_col, 0xFF, cells * 4, q.main));
            FWX_HIP(hipMemsetAsync(q.plog.at_row, 0xFF, cells * 4, q.main));
        }
        if (counting) FWX_HIP(hipMemsetAsync(q.upd, 0, FWX_UPDATE_SHARDS * 8, q.main));
    }
    // pivot blocks of at most 64 that never straddle two owners; real pivots only (padding is inert)
    fail_point();
    std::vector<Block> blocks;
    for (int p = 0; p < P; ++p) {
        const Part &q = M.part[p];
        int k0 = q.row0 > op.k_begin ? q.row0 : op.k_begin;
        const int hi = q.row0 + q.rows < op.k_end ? q.row0 + q.rows : op.k_end;
        while (k0 < hi) {
            const int bt = hi - k0 < FWX_FUSED_BLOCK ? hi - k0 : FWX_FUSED_BLOCK;
            blocks.push_back({k0, bt, p});
            k0 += bt;
        }
    }
    if (blocks.empty()) return FWX_OK;
    M.timer.reset();
    M.timer.last.pivots_per_step = FWX_FUSED_BLOCK;
    Throttle thr;
    // double pass (the fused engine inside the domain, 64-aligned partitions, large matrices): the full
    // blocks in pairs, 128 pivots per main launch behind a two-deep look-ahead; what is left -- an odd
    // block's worth or a ragged tail -- goes through the single-pass loop below
    size_t first = 0;
    if (!perk && nonneg && !counting && !rec) {
        int done = 0;
        if ((rc = multi_double_pass<T>(m, blocks, thr, done))) return rc;
        first = (size_t)done;
        if (first == blocks.size()) return finish_multi_solve(m, counting, op);
    }
    // the state of every slab at the START of a checkpoint pivot (a block start), on a recording handle
    auto checkpoint = [&](int k0) -> int {
        if (!rec) return FWX_OK;
        for (size_t c = 0; c < rec->pivot.size(); ++c) {
            if (rec->pivot[c] != k0) continue;
            for (int p = 0; p < P; ++p) {
                Part &q = M.part[p];
                if (q.rows == 0 || !here(p)) continue;
                int rc2 = set_dev(q.device);
                if (rc2) return rc2;
                const size_t cells = (size_t)q.rows * nd;
                FWX_HIP(hipMemcpyAsync(q.R.rate[c], q.rate, cells * sizeof(T), hipMemcpyDeviceToDevice, q.main));
                if (q.next) FWX_HIP(hipMemcpyAsync(q.R.next[c], q.next, cells * 4, hipMemcpyDeviceToDevice, q.main));
                if (q.hops) FWX_HIP(hipMemcpyAsync(q.R.hops[c], q.hops, cells * 4, hipMemcpyDeviceToDevice, q.main));
                if (q.plog.last) {
                    FWX_HIP(hipMemcpyAsync(q.R.last[c], q.plog.last, cells * 4, hipMemcpyDeviceToDevice, q.main));
                    FWX_HIP(hipMemcpyAsync(q.R.at_col[c], q.plog.at_col, cells * 4, hipMemcpyDeviceToDevice, q.main));
                    FWX_HIP(hipMemcpyAsync(q.R.at_row[c], q.plog.at_row, cells * 4, hipMemcpyDeviceToDevice, q.main));
                }
            }
        }
        return FWX_OK;
    };
    // per-k engine on several partitions: one enqueueing thread per partition (see SweepWorkers)
    std::unique_ptr<SweepWorkers> workers;
    if (perk && P > 1 && M.self < 0) {
        fail_point();
        workers.reset(new SweepWorkers(P));
    }
    {   // the first panel: its rows are at time k0 already
        Part &o = M.part[blocks[first].owner];
        if (here(blocks[first].owner)) {
            if ((rc = set_dev(o.device))) return rc;
            FWX_HIP(hipEventRecord(o.rows_done, o.main));
        }
        bind_slot(blocks[first], 0);
        if ((rc = issue_panel<T>(M, blocks[first], 0, (int)first))) return rc;
    }
    MultiTimer &tm = M.timer;
    int t_bulk[FWX_MAX_PARTS];
    for (size_t b = first; b < blocks.size(); ++b) {
        const Block &blk = blocks[b];
        const int slot = (int)((b - first) & 1);
        const bool more = b + 1 < blocks.size();
        const int step = (int)b;
        // everything queued on the main streams so far precedes these copies: the previous block's sweep,
        // the look-ahead on this block's rows; this block's panel (side stream) only READS the slab
        if (b > first && (rc = checkpoint(blk.k0))) return rc;
        // fused engine: pivot-column snapshots on every partition; per-k engine: the pivot column is
        // read from the slab itself by every launch, the main stream just waits for the panel
        for (int p = 0; p < P; ++p) {
            Part &q = M.part[p];
            if (q.rows == 0 || !here(p)) continue;
            if ((rc = set_dev(q.device))) return rc;
            FWX_HIP(hipStreamWaitEvent(q.main, q.w_ready[slot], 0));
            t_bulk[p] = tm.begin(MultiTimer::BULK, p, step, q.main);
            if (perk) continue;
            fwx::FusedArgs<T> a = part_args<T>(M, q, nonneg, counting);
            a.k0 = blk.k0; a.bt = blk.bt; a.w = (const T *)q.wp[slot]; a.wh = q.whp[slot];
            bind_cols(a, q, blk);
            FWX_HIP(fwx::launch_fused_colpanel<T>(a, q.main));
        }
        int la_lo = 0, la_hi = 0, la_owner = -1;
        if (more) {
            // look-ahead: the rows of the next panel first, then their snapshot + exchange on the side
            const Block &nb = blocks[b + 1];
            Part &o = M.part[nb.owner];
            la_owner = nb.owner; la_lo = nb.k0 - o.row0; la_hi = la_lo + nb.bt;
            if (here(nb.owner)) {
            if ((rc = set_dev(o.device))) return rc;
            fwx::FusedArgs<T> a = part_args<T>(M, o, nonneg, counting);
            a.k0 = blk.k0; a.bt = blk.bt; a.w = (const T *)o.wp[slot]; a.wh = o.whp[slot];
            bind_cols(a, o, blk);
            const int t_la = tm.begin(MultiTimer::LOOKAHEAD, nb.owner, step, o.main);
            if (!perk) {
                a.side = true;                     // the look-ahead rows head the owner's chain
                FWX_HIP(fwx::launch_fused_main<T>(a, la_lo, la_hi, o.main));
            } else {
                // these few rows sit on the owner's critical path: one fused launch (column snapshots of
                // just these rows + the 64 pivots), bit-identical to 64 per-k launches
                const size_t off = (size_t)la_lo * nd;
                a.rate += off;
                if (a.next) a.next += off;
                if (a.hops) a.hops += off;
                a.rows = nb.bt; a.row0 = nb.k0;
                FWX_HIP(fwx::launch_fused_relax<T>(a, o.main));
            }
            tm.end(t_la, o.main);
            FWX_HIP(hipEventRecord(o.rows_done, o.main));
            }
            bind_slot(nb, slot ^ 1);
            if ((rc = issue_panel<T>(M, nb, slot ^ 1, step + 1))) return rc;
        }
        for (int p = 0; p < P; ++p) {
            Part &q = M.part[p];
            if (!here(p)) continue;
            if ((rc = set_dev(q.device))) return rc;
            if (perk) continue;                       // the per-k sweeps: below, all partitions at once
            if (q.rows > 0) {
                fwx::FusedArgs<T> a = part_args<T>(M, q, nonneg, counting);
                a.k0 = blk.k0; a.bt = blk.bt; a.w = (const T *)q.wp[slot]; a.wh = q.whp[slot];
                bind_cols(a, q, blk);
                if (p != la_owner) {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, q.rows, q.main));
                } else if (la_lo % 8 == 0 && la_hi % 8 == 0) {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, q.rows, q.main, la_lo, la_hi));
                } else {
                    FWX_HIP(fwx::launch_fused_main<T>(a, 0, la_lo, q.main));
                    FWX_HIP(fwx::launch_fused_main<T>(a, la_hi, q.rows, q.main));
                }
                tm.end(t_bulk[p], q.main);
            }
            FWX_HIP(hipEventRecord(q.main_free[slot], q.main));
        }
        if (perk) {
            // one launch per pivot over each slab, pivot rows from the snapshot panel (BASELINE config 4:
            // "row-partitioned, pivot-row broadcast per k" -- 64 rows per message)
            for (int p = 0; p < P; ++p) {
                Part *qp = &M.part[p];
                if (qp->rows == 0 || !here(p)) continue;
                const bool owner = p == la_owner;
                auto job = [=, &op]() -> int {
                    Part &q = *qp;
                    int rc2 = set_dev(q.device);      // the worker's own current device
                    if (rc2) return rc2;
                    auto sweep = [&](int lo, int hi, int skip_lo, int skip_hi) -> int {
                        if (hi <= lo) return FWX_OK;
                        const size_t off = (size_t)lo * nd;
                        return relax_range<T>((T *)q.rate + off, q.next ? q.next + off : nullptr,
                                              q.hops ? q.hops + off : nullptr, hi - lo, nd, q.row0 + lo,
                                              (const T *)q.wp[slot], q.whp[slot], nd, blk.k0, blk.k0 + blk.bt,
                                              op.serpentine, counting ? q.upd : nullptr, q.main, fwx::PathLog(),
                                              skip_lo, skip_hi);
                    };
                    if (!owner) return sweep(0, q.rows, 0, 0);
                    if (la_lo % 4 == 0 && la_hi % 4 == 0) return sweep(0, q.rows, la_lo, la_hi);
                    if ((rc2 = sweep(0, la_lo, 0, 0))) return rc2;
                    return sweep(la_hi, q.rows, 0, 0);
                };
                if (workers) workers->set(p, job);
                else if ((rc = job())) return rc;
            }
            if (workers && (rc = workers->run_all())) return rc;
            for (int p = 0; p < P; ++p) {
                Part &q = M.part[p];
                if (!here(p)) continue;
                if ((rc = set_dev(q.device))) return rc;
                if (q.rows > 0) tm.end(t_bulk[p], q.main);
                FWX_HIP(hipEventRecord(q.main_free[slot], q.main));
            }
        }
        if ((rc = set_dev(M.part[M.first_here()].device))) return rc;
        if ((rc = thr.tick(M.part[M.first_here()].main, perk ? blk.bt + 4 : 4))) return rc;
    }
    if ((rc = finish_multi_solve(m, counting, op))) return rc;
    if (rec) rec->valid_upto = rec->state_at = op.k_end;
    return FWX_OK;
}

// ---- entry points used by fwx_api.hip for handles with m->multi -----------------------------------
int multi_upload(fwx_matrix *m, const void *rate, const int32_t *next, const int32_t *hops)
{
    DevRestore keep;
    const int rc = multi_copy(m, const_cast<void *>(rate), const_cast<int32_t *>(next),
                              const_cast<int32_t *>(hops), true);
    if (rc) return rc;
    if (m->plog.last) m->rec_ready = 0;
    if (m->keep) m->kept_valid = 1;
    m->fresh = 1;
    m->dom_known = 0;                  // a new input: the domain check has to look at it
    if (m->resume) {                   // ... and nothing of the old solve can be resumed
        m->resume->valid_upto = 0;
        m->resume->state_at = m->keep ? 0 : -1;
    }
    return FWX_OK;
}

int multi_download(fwx_matrix *m, void *rate, int32_t *next, int32_t *hops)
{
    DevRestore keep;
    return multi_copy(m, rate, next, hops, false);
}

int multi_solve(fwx_matrix *m, const Opts &op, bool resumed)
{
    if (m->plog.last && ((!resumed && op.k_begin != 0) || op.k_end != m->n))
        return FWX_ERR_UNSUPPORTED;      // the trace covers whole solves (as on one device)
    if (op.engine == FWX_ENGINE_PERK && m->plog.last)
        return FWX_ERR_UNSUPPORTED;      // the per-k kernel keeps no path trace on slabs: AUTO / FUSED do
    if (m->plog.last && !m->fresh && !resumed) return FWX_ERR_INVALID;   // a traced solve starts from an upload
    DevRestore keep;
    m->fresh = 0;
    const int rc = m->dtype == FWX_F64 ? multi_solve_typed<double>(m, op, resumed)
                                       : multi_solve_typed<float>(m, op, resumed);
    if (rc) {
        if (m->resume) { m->resume->valid_upto = 0; m->resume->state_at = -1; }
        return rc;
    }
    if (m->plog.last) m->rec_ready = 1;
    return FWX_OK;
}

// ---- resumable solves on a partitioned handle (fwx_matrix_enable_resume / fwx_matrix_resolve) -------
void multi_resume_dims(const fwx_matrix *m, uint64_t *cells, uint64_t *col_cells, uint64_t *w_cells)
{
    const MultiState &M = *m->multi;
    *cells = *col_cells = *w_cells = 0;
    for (int p = 0; p < M.parts; ++p) {
        const Part &q = M.part[p];
        if (!M.here(p)) continue;
        *cells += (uint64_t)q.rows * M.nd;
        *col_cells += (uint64_t)M.nd * (q.ct_ld ? q.ct_ld : 4);
        *w_cells += (uint64_t)M.nd * M.nd;                   // every partition keeps all pivot rows
    }
}

int multi_enable_resume(fwx_matrix *m, int32_t checkpoints)
{
    MultiState &M = *m->multi;
    DevRestore keep;
    const int n = m->n, nd = M.nd;
    const size_t es = m->dtype == FWX_F64 ? 8 : 4;
    fail_point();
    struct Holder {
        Resume *r = new Resume();
        MultiState *M;
        ~Holder()
        {
            if (!r) return;
            for (int p = 0; p < M->parts; ++p)
                if (M->here(p) && hipSetDevice(M->part[p].device) == hipSuccess) part_resume_free(M->part[p]);
            delete r;
        }
    } hold;
    hold.M = &M;
    Resume *R = hold.r;
    R->ld = 0;
    // checkpoints at the multiples of 64 closest to q * n / (checkpoints + 1) that are block starts: a
    // block never straddles two partitions, so inside partition p blocks start at row0 + 64 t
    for (int q = 1; q <= checkpoints; ++q) {
        const int c = (int)(((int64_t)n * q / (checkpoints + 1) + 32) / 64 * 64);
        if (c <= 0 || c >= n || (!R->pivot.empty() && c <= R->pivot.back())) continue;
        int p = 0;
        while (p + 1 < M.parts && c >= M.part[p + 1].row0) ++p;
        if ((c - M.part[p].row0) % FWX_FUSED_BLOCK != 0) continue;
        R->pivot.push_back(c);
    }
    R->count = (int)R->pivot.size();
    auto alloc = [&](void **ptr, size_t bytes) -> int { FWX_HIP(hipMalloc(ptr, bytes ? bytes : 16)); return FWX_OK; };
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        int rc = set_dev(q.device);
        if (rc) return rc;
        const size_t cells = (size_t)q.rows * nd, pan = (size_t)nd * (q.ct_ld ? q.ct_ld : 4);
        for (int c = 0; c < R->count; ++c) {
            void *ptr = nullptr;
            if ((rc = alloc(&ptr, cells * es))) return rc;
            q.R.rate.push_back(ptr);
            if (q.next) { if ((rc = alloc(&ptr, cells * 4))) return rc; q.R.next.push_back((int32_t *)ptr); }
            if (q.hops) { if ((rc = alloc(&ptr, cells * 4))) return rc; q.R.hops.push_back((int32_t *)ptr); }
            if (q.plog.last)
                for (auto *v : {&q.R.last, &q.R.at_col, &q.R.at_row}) {
                    if ((rc = alloc(&ptr, cells * 4))) return rc;
                    v->push_back((int32_t *)ptr);
                }
        }
        if ((rc = alloc(&q.R.rw, (size_t)nd * nd * es)) || (rc = alloc(&q.R.rct, pan * es))) return rc;
        if (q.next && (rc = alloc((void **)&q.R.rcnt, pan * 4))) return rc;
        if (q.hops && ((rc = alloc((void **)&q.R.rwh, (size_t)nd * nd * 4)) || (rc = alloc((void **)&q.R.rcht, pan * 4))))
            return rc;
        if ((rc = alloc((void **)&q.R.idx, (size_t)FWX_MAX_PATCH * 8))) return rc;
    }
    R->state_at = (m->fresh && m->kept_valid) ? 0 : -1;
    m->resume = R;
    hold.r = nullptr;
    return R->count;
}

void multi_resume_free(fwx_matrix *m)
{
    if (!m->resume) return;
    delete m->resume;          // (the partitions' arrays go with the partitions: multi_free)
    m->resume = nullptr;
}

// The resumed path of fwx_matrix_resolve: patch the kept input, restore checkpoint c_idx on every
// partition, replay the changed entries through pivots [0, c) from the stored panels (each on the
// partition that owns its row: its column snapshots are local, the pivot rows are the exchanged copies
// every partition keeps), then run pivots [c, n).
template <typename T>
static int multi_resolve_typed(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                               const int32_t *next_vals, const int32_t *hops_vals, int c_idx)
{
    MultiState &M = *m->multi;
    Resume &R = *m->resume;
    const int nd = M.nd, c = R.pivot[(size_t)c_idx];
    int rc;
    std::vector<int64_t> local[FWX_MAX_PARTS];
    for (int32_t e = 0; e < count; ++e) {
        const int row = (int)(index[e] / m->n), col = (int)(index[e] % m->n);
        int p = 0;
        while (p + 1 < M.parts && row >= M.part[p + 1].row0) ++p;
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        if ((rc = set_dev(q.device))) return rc;
        const size_t off = (size_t)(row - q.row0) * nd + col;
        local[p].push_back((int64_t)off);
        FWX_HIP(hipMemcpyAsync((T *)q.rate0 + off, (const T *)rate_vals + e, sizeof(T), hipMemcpyHostToDevice, q.main));
        if (next_vals) FWX_HIP(hipMemcpyAsync(q.next0 + off, next_vals + e, 4, hipMemcpyHostToDevice, q.main));
        if (hops_vals) FWX_HIP(hipMemcpyAsync(q.hops0 + off, hops_vals + e, 4, hipMemcpyHostToDevice, q.main));
    }
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (q.rows == 0 || !M.here(p)) continue;
        if ((rc = set_dev(q.device))) return rc;
        const size_t cells = (size_t)q.rows * nd, ci = (size_t)c_idx;
        FWX_HIP(hipMemcpyAsync(q.rate, q.R.rate[ci], cells * sizeof(T), hipMemcpyDeviceToDevice, q.main));
        if (q.next) FWX_HIP(hipMemcpyAsync(q.next, q.R.next[ci], cells * 4, hipMemcpyDeviceToDevice, q.main));
        if (q.hops) FWX_HIP(hipMemcpyAsync(q.hops, q.R.hops[ci], cells * 4, hipMemcpyDeviceToDevice, q.main));
        if (q.plog.last) {
            FWX_HIP(hipMemcpyAsync(q.plog.last, q.R.last[ci], cells * 4, hipMemcpyDeviceToDevice, q.main));
            FWX_HIP(hipMemcpyAsync(q.plog.at_col, q.R.at_col[ci], cells * 4, hipMemcpyDeviceToDevice, q.main));
            FWX_HIP(hipMemcpyAsync(q.plog.at_row, q.R.at_row[ci], cells * 4, hipMemcpyDeviceToDevice, q.main));
        }
        if (local[p].empty()) continue;
        ReplayTargets tg;
        memset(&tg, 0, sizeof(tg));
        for (int t = 0; t <= c_idx; ++t) {
            const int k = tg.count++;
            tg.pivot[k] = R.pivot[(size_t)t];
            tg.rate[k] = q.R.rate[(size_t)t];
            tg.next[k] = q.next ? q.R.next[(size_t)t] : nullptr;
            tg.hops[k] = q.hops ? q.R.hops[(size_t)t] : nullptr;
            tg.last[k] = q.plog.last ? q.R.last[(size_t)t] : nullptr;
        }
        {
            const int k = tg.count++;
            tg.pivot[k] = c;
            tg.rate[k] = q.rate; tg.next[k] = q.next; tg.hops[k] = q.hops; tg.last[k] = q.plog.last;
        }
        FWX_HIP(hipMemcpyAsync(q.R.idx, local[p].data(), local[p].size() * 8, hipMemcpyHostToDevice, q.main));
        hipLaunchKernelGGL(replay_entries_kernel<T>, dim3((unsigned)local[p].size()), dim3(64), 0, q.main, q.R.idx, nd,
                           q.ct_ld, q.row0, c, (const T *)q.rate0, q.next ? q.next0 : nullptr,
                           q.hops ? q.hops0 : nullptr, (const T *)q.R.rw, (const T *)q.R.rct, q.next ? q.R.rcnt : nullptr,
                           q.R.rwh, q.R.rcht, tg);
        FWX_HIP(hipGetLastError());
        FWX_HIP(hipStreamSynchronize(q.main));      // (local[p] is read by the copy above)
    }
    return FWX_OK;
}

int multi_resolve(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                  const int32_t *next_vals, const int32_t *hops_vals, int c_idx, Opts op)
{
    DevRestore keep;
    Resume &R = *m->resume;
    const int c = R.pivot[(size_t)c_idx];
    R.valid_upto = 0;
    int rc = m->dtype == FWX_F64
                 ? multi_resolve_typed<double>(m, count, index, rate_vals, next_vals, hops_vals, c_idx)
                 : multi_resolve_typed<float>(m, count, index, rate_vals, next_vals, hops_vals, c_idx);
    if (rc) {
        R.state_at = -1;
        m->fresh = 0;
        m->rec_ready = 0;
        return rc;
    }
    m->fresh = 0;
    m->rec_ready = 0;
    R.valid_upto = c;
    R.state_at = c;
    op.k_begin = c;
    return multi_solve(m, op, true);
}

int multi_enable_path_log(fwx_matrix *m)
{
    MultiState &M = *m->multi;
    DevRestore keep;
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        int rc = set_dev(q.device);
        if (rc) return rc;
        const size_t bytes = (size_t)q.rows * M.nd * 4;
        FWX_HIP(hipMalloc((void **)&q.plog.at_col, bytes ? bytes : 16));
        FWX_HIP(hipMalloc((void **)&q.plog.at_row, bytes ? bytes : 16));
        if (!q.next0) FWX_HIP(hipMalloc((void **)&q.next0, bytes ? bytes : 16));
        FWX_HIP(hipMalloc((void **)&q.plog.last, bytes ? bytes : 16));
        if (m->fresh) {
            FWX_HIP(hipMemcpyAsync(q.next0, q.next, bytes, hipMemcpyDeviceToDevice, q.main));
            FWX_HIP(hipStreamSynchronize(q.main));
        }
    }
    m->plog.last = M.part[M.first_here()].plog.last;     // "enabled" marker for the shared handle logic
    m->rec_ready = 0;
    return FWX_OK;
}

static int read_rate(fwx_matrix *m, int src, int dst, double *rate_out)
{
    MultiState &M = *m->multi;
    int p = 0;
    while (p + 1 < M.parts && src >= M.part[p + 1].row0) ++p;
    Part &q = M.part[p];
    int rc = set_dev(q.device);
    if (rc) return rc;
    const size_t off = (size_t)(src - q.row0) * M.nd + dst;
    if (m->dtype == FWX_F64) {
        FWX_HIP(hipMemcpyAsync(rate_out, (double *)q.rate + off, 8, hipMemcpyDeviceToHost, q.main));
        FWX_HIP(hipStreamSynchronize(q.main));
    } else {
        float f = 0;
        FWX_HIP(hipMemcpyAsync(&f, (float *)q.rate + off, 4, hipMemcpyDeviceToHost, q.main));
        FWX_HIP(hipStreamSynchronize(q.main));
        *rate_out = (double)f;
    }
    return FWX_OK;
}

static int query_scratch(MultiState &M, int32_t ints)
{
    if (M.qscratch && M.qcap >= ints) return FWX_OK;
    if (M.qscratch) { drain_stream(M.part[0].main); (void)hipFree(M.qscratch); M.qscratch = nullptr; M.qcap = 0; }
    FWX_HIP(hipMalloc((void **)&M.qscratch, (size_t)ints * 4));
    M.qcap = ints;
    return FWX_OK;
}

// fwx_matrix_query where some pair of devices refused peer access: the same walk driven from the
// host, one 4-byte read per hop from the partition that owns the row (slow, correct, rarely needed).
static int host_walk(fwx_matrix *m, int32_t src, int32_t dst, int32_t *path_out, int32_t cap)
{
    MultiState &M = *m->multi;
    auto next_of = [&](int a, int32_t *out) -> int {
        int p = 0;
        while (p + 1 < M.parts && a >= M.part[p + 1].row0) ++p;
        Part &q = M.part[p];
        int rc = set_dev(q.device);
        if (rc) return rc;
        FWX_HIP(hipMemcpyAsync(out, q.next + (size_t)(a - q.row0) * M.nd + dst, 4, hipMemcpyDeviceToHost, q.main));
        FWX_HIP(hipStreamSynchronize(q.main));
        return FWX_OK;
    };
    int32_t nx = -1;
    int rc = next_of(src, &nx);
    if (rc) return rc;
    if (nx < 0) return 0;
    int32_t len = 0, cur = src;
    while (cur != dst || len == 0) {
        if ((rc = next_of(cur, &nx))) return rc;
        if (nx < 0 || nx >= m->n || len >= m->n) return FWX_ERR_CYCLE;
        if (len >= cap) return FWX_ERR_CAPACITY;
        path_out[len++] = nx;
        cur = nx;
    }
    return len;
}

int multi_query(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out, int32_t *path_out, int32_t cap)
{
    MultiState &M = *m->multi;
    if (M.self >= 0) return FWX_ERR_UNSUPPORTED;   // one partition per process: a walk crosses the ranks' slabs
    DevRestore keep;
    int rc;
    if (rate_out && (rc = read_rate(m, src, dst, rate_out))) return rc;
    if (!m->next) return FWX_ERR_INVALID;
    if (!M.peer_all) return host_walk(m, src, dst, path_out, cap);
    Part &z = M.part[0];
    if ((rc = set_dev(z.device))) return rc;
    const int dcap = cap < m->n ? cap : m->n;
    if ((rc = query_scratch(M, m->n + 2))) return rc;
    hipLaunchKernelGGL(multi_follow_path_kernel, dim3(1), dim3(1), 0, z.main, make_tab(M), m->n, src, dst,
                       M.qscratch + 1, dcap, M.qscratch);
    FWX_HIP(hipGetLastError());
    int32_t len = 0;
    FWX_HIP(hipMemcpyAsync(&len, M.qscratch, 4, hipMemcpyDeviceToHost, z.main));
    FWX_HIP(hipStreamSynchronize(z.main));
    if (len > 0) {
        FWX_HIP(hipMemcpyAsync(path_out, M.qscratch + 1, (size_t)len * 4, hipMemcpyDeviceToHost, z.main));
        FWX_HIP(hipStreamSynchronize(z.main));
    }
    return len;
}

int multi_query_exact_batch(fwx_matrix *m, int32_t count, const int32_t *src, const int32_t *dst,
                            int32_t *len_out, int32_t *path_out, int32_t cap)
{
    MultiState &M = *m->multi;
    if (!M.peer_all || M.self >= 0) return FWX_ERR_UNSUPPORTED;   // the walk reads every slab from one device
    DevRestore keep;
    Part &z = M.part[0];
    int rc = set_dev(z.device);
    if (rc) return rc;
    // device scratch from a pooled per-call context: no hipMalloc / hipFree per query (and no hipFree
    // right behind the kernel that used the memory: drain_stream in fwx_internal.h)
    CtxLease lease;
    if ((rc = lease.open())) return rc;
    lease.c->uses_stream(z.main);            // the kernel below runs on partition 0's stream
    const size_t c = (size_t)count;
    struct { void *p = nullptr; } d_src, d_dst, d_len, d_paths, d_stacks;
    void *ids = nullptr;
    if ((rc = lease.c->reserve(CallCtx::NEXT, c * 12, &ids)) ||
        (rc = lease.c->reserve(CallCtx::RATE, c * cap * 4, &d_paths.p)) ||
        (rc = lease.c->reserve(CallCtx::WS, c * cap * 12, &d_stacks.p)))
        return rc;
    d_src.p = ids;
    d_dst.p = (char *)ids + c * 4;
    d_len.p = (char *)ids + c * 8;
    FWX_HIP(hipMemcpyAsync(d_src.p, src, c * 4, hipMemcpyHostToDevice, z.main));
    FWX_HIP(hipMemcpyAsync(d_dst.p, dst, c * 4, hipMemcpyHostToDevice, z.main));
    hipLaunchKernelGGL(multi_exact_paths_kernel, dim3((unsigned)((c + 63) / 64)), dim3(64), 0, z.main,
                       make_tab(M), m->n, count, (const int32_t *)d_src.p, (const int32_t *)d_dst.p,
                       (int32_t *)d_paths.p, (int32_t *)d_stacks.p, cap, (int32_t *)d_len.p);
    FWX_HIP(hipGetLastError());
    FWX_HIP(hipMemcpyAsync(len_out, d_len.p, c * 4, hipMemcpyDeviceToHost, z.main));
    FWX_HIP(hipMemcpyAsync(path_out, d_paths.p, c * cap * 4, hipMemcpyDeviceToHost, z.main));
    FWX_HIP(hipStreamSynchronize(z.main));
    return FWX_OK;
}

int multi_query_exact(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out, int32_t *path_out,
                      int32_t cap)
{
    int rc;
    if (m->multi->self >= 0) return FWX_ERR_UNSUPPORTED;
    {
        DevRestore keep;
        if (rate_out && (rc = read_rate(m, src, dst, rate_out))) return rc;
    }
    int32_t len = 0;
    if ((rc = multi_query_exact_batch(m, 1, &src, &dst, &len, path_out, cap))) return rc;
    return len;
}

int multi_keep_input(fwx_matrix *m)
{
    MultiState &M = *m->multi;
    DevRestore keep;
    const size_t es = m->dtype == FWX_F64 ? 8 : 4;
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        int rc = set_dev(q.device);
        if (rc) return rc;
        const size_t cells = (size_t)q.rows * M.nd;
        FWX_HIP(hipMalloc(&q.rate0, cells * es ? cells * es : 16));
        if (q.next && !q.next0) FWX_HIP(hipMalloc((void **)&q.next0, cells * 4 ? cells * 4 : 16));
        if (q.hops) FWX_HIP(hipMalloc((void **)&q.hops0, cells * 4 ? cells * 4 : 16));
        if (m->fresh) {
            FWX_HIP(hipMemcpyAsync(q.rate0, q.rate, cells * es, hipMemcpyDeviceToDevice, q.main));
            if (q.next) FWX_HIP(hipMemcpyAsync(q.next0, q.next, cells * 4, hipMemcpyDeviceToDevice, q.main));
            if (q.hops) FWX_HIP(hipMemcpyAsync(q.hops0, q.hops, cells * 4, hipMemcpyDeviceToDevice, q.main));
            FWX_HIP(hipStreamSynchronize(q.main));
        }
    }
    m->keep = 1;
    m->kept_valid = m->fresh ? 1 : 0;
    return FWX_OK;
}

int multi_patch_input(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                      const int32_t *next_vals, const int32_t *hops_vals)
{
    MultiState &M = *m->multi;
    DevRestore keep;
    const size_t es = m->dtype == FWX_F64 ? 8 : 4;
    int rc;
    if (m->resume) m->resume->valid_upto = 0;      // the kept input changes without a replay
    if (m->dom_known && !patch_keeps_domain(m, count, rate_vals, next_vals)) m->dom_known = 0;
    for (int32_t e = 0; e < count; ++e) {
        const int row = (int)(index[e] / m->n), col = (int)(index[e] % m->n);
        int p = 0;
        while (p + 1 < M.parts && row >= M.part[p + 1].row0) ++p;
        Part &q = M.part[p];
        if (!M.here(p)) continue;                  // (an entry of a row another process holds)
        if ((rc = set_dev(q.device))) return rc;
        const size_t off = (size_t)(row - q.row0) * M.nd + col;
        FWX_HIP(hipMemcpyAsync((char *)q.rate0 + off * es, (const char *)rate_vals + (size_t)e * es, es,
                               hipMemcpyHostToDevice, q.main));
        if (next_vals) FWX_HIP(hipMemcpyAsync(q.next0 + off, next_vals + e, 4, hipMemcpyHostToDevice, q.main));
        if (hops_vals) FWX_HIP(hipMemcpyAsync(q.hops0 + off, hops_vals + e, 4, hipMemcpyHostToDevice, q.main));
    }
    for (int p = 0; p < M.parts; ++p) {
        Part &q = M.part[p];
        if (!M.here(p)) continue;
        if ((rc = set_dev(q.device))) return rc;
        const size_t cells = (size_t)q.rows * M.nd;
        FWX_HIP(hipMemcpyAsync(q.rate, q.rate0, cells * es, hipMemcpyDeviceToDevice, q.main));
        if (q.next) FWX_HIP(hipMemcpyAsync(q.next, q.next0, cells * 4, hipMemcpyDeviceToDevice, q.main));
        if (q.hops) FWX_HIP(hipMemcpyAsync(q.hops, q.hops0, cells * 4, hipMemcpyDeviceToDevice, q.main));
    }
    for (int p = 0; p < M.parts; ++p) {
        if (!M.here(p)) continue;
        if ((rc = set_dev(M.part[p].device))) return rc;
        FWX_HIP(hipStreamSynchronize(M.part[p].main));
    }
    m->fresh = 1;
    m->rec_ready = 0;
    if (m->resume) m->resume->state_at = 0;        // the patched kept input, unsolved
    return FWX_OK;
}

void multi_destroy(fwx_matrix *m)
{
    multi_free(m->multi);
    m->multi = nullptr;
    multi_resume_free(m);
}

}  // namespace fwxi

using namespace fwxi;

extern "C" {

int fwx_matrix_create_multi(fwx_matrix **out, int32_t n, int32_t dtype, int32_t with_next,
                            int32_t with_hops, int32_t n_parts, const int32_t *devices, int32_t exchange)
{
    return fwxi::guarded([&]() -> int {
        if (!out || n < 0 || (dtype != FWX_F32 && dtype != FWX_F64) || n_parts < 1 ||
            n_parts > FWX_MAX_PARTS || !devices ||
            (exchange != FWX_XCHG_AUTO && exchange != FWX_XCHG_PEER && exchange != FWX_XCHG_RCCL))
            return FWX_ERR_INVALID;
        *out = nullptr;
        if (with_hops && !with_next) return FWX_ERR_INVALID;
        const int cnt = device_count();
        if (cnt <= 0) return FWX_ERR_NO_DEVICE;
        int32_t resolved[FWX_MAX_PARTS];              // -1 = the caller's current device
        int cur = 0;
        FWX_HIP(hipGetDevice(&cur));
        for (int p = 0; p < n_parts; ++p) {
            resolved[p] = devices[p] == -1 ? cur : devices[p];
            if (resolved[p] < 0 || resolved[p] >= cnt) return FWX_ERR_INVALID;
        }
        devices = resolved;
        fwx_matrix *m = new (std::nothrow) fwx_matrix();
        if (!m) return FWX_ERR_OOM;
        memset(m, 0, sizeof(*m));
        m->n = m->nd = n; m->dtype = dtype; m->device = devices[0];   // (the slabs' pitch is MultiState::nd)
        m->next = with_next ? (int32_t *)(uintptr_t)16 : nullptr;   // markers only: the slabs own the arrays
        m->hops = with_hops ? (int32_t *)(uintptr_t)16 : nullptr;
        DevRestore keep;
        const int rc = multi_alloc(m, n_parts, devices, exchange);
        if (rc) {
            multi_destroy(m);
            delete m;
            return rc;
        }
        *out = m;
        return FWX_OK;
    });
}

int fwx_matrix_comm_ranks(const fwx_matrix *m)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (!m->multi || !m->multi->comms) return 0;
        int ranks = 0;
        if (rccl().CommCount(m->multi->comms->comm[0], &ranks) != ncclSuccess) return FWX_ERR_RCCL;
        return ranks;
    });
}

int fwx_matrix_create_part(fwx_matrix **out, int32_t n, int32_t dtype, int32_t with_next, int32_t with_hops,
                           int32_t rank, int32_t world, int32_t device, fwx_exchange_fn exchange, void *ctx)
{
    return fwxi::guarded([&]() -> int {
        if (!out || n < 0 || (dtype != FWX_F32 && dtype != FWX_F64) || world < 1 || world > FWX_MAX_PARTS ||
            rank < 0 || rank >= world || !exchange || (with_hops && !with_next))
            return FWX_ERR_INVALID;
        *out = nullptr;
        const int cnt = device_count();
        if (cnt <= 0) return FWX_ERR_NO_DEVICE;
        int cur = 0;
        FWX_HIP(hipGetDevice(&cur));
        const int dev = device == -1 ? cur : device;
        if (dev < 0 || dev >= cnt) return FWX_ERR_INVALID;
        int32_t devices[FWX_MAX_PARTS];
        for (int p = 0; p < world; ++p) devices[p] = dev;          // (only devices[rank] is ever used)
        fwx_matrix *m = new (std::nothrow) fwx_matrix();
        if (!m) return FWX_ERR_OOM;
        memset(m, 0, sizeof(*m));
        m->n = m->nd = n; m->dtype = dtype; m->device = dev;
        m->next = with_next ? (int32_t *)(uintptr_t)16 : nullptr;   // markers only: the slab owns the arrays
        m->hops = with_hops ? (int32_t *)(uintptr_t)16 : nullptr;
        DevRestore keep;
        const int rc = multi_alloc(m, world, devices, FWX_XCHG_CALLBACK, rank, exchange, ctx);
        if (rc) {
            multi_destroy(m);
            delete m;
            return rc;
        }
        *out = m;
        return FWX_OK;
    });
}

int fwx_matrix_part_rows(const fwx_matrix *m, int32_t part, int32_t *row0_out, int32_t *rows_out)
{
    return fwxi::guarded([&]() -> int {
        if (!m || part < 0) return FWX_ERR_INVALID;
        int row0 = 0, rows = m->n;
        if (m->multi) {
            if (part >= m->multi->parts) return FWX_ERR_INVALID;
            const Part &q = m->multi->part[part];
            row0 = q.row0;
            rows = q.row0 + q.rows <= m->n ? q.rows : m->n - q.row0;      // (the last slab also holds the padding rows)
            if (rows < 0) rows = 0;
        } else if (part != 0) {
            return FWX_ERR_INVALID;
        }
        if (row0_out) *row0_out = row0;
        if (rows_out) *rows_out = rows;
        return FWX_OK;
    });
}

int fwx_matrix_domain_bits(fwx_matrix *m, int32_t *bits_out)
{
    return fwxi::guarded([&]() -> int {
        if (!m || !bits_out || !m->multi || m->multi->self < 0) return FWX_ERR_INVALID;
        MultiState &M = *m->multi;
        Part &q = M.part[M.self];
        *bits_out = 3;
        if (q.rows == 0 || m->n == 0) return FWX_OK;
        DevRestore keep;
        int rc = set_dev(q.device), b = 3;
        if (rc) return rc;
        rc = m->dtype == FWX_F64
                 ? domain_bits<double>((const double *)q.rate, q.next, (size_t)q.rows * M.nd, q.flag, q.main, b)
                 : domain_bits<float>((const float *)q.rate, q.next, (size_t)q.rows * M.nd, q.flag, q.main, b);
        if (rc) return rc;
        *bits_out = b;
        return FWX_OK;
    });
}

int fwx_matrix_set_domain(fwx_matrix *m, int32_t bits)
{
    return fwxi::guarded([&]() -> int {
        if (!m || !m->multi || m->multi->self < 0 || bits < 0 || bits > 3) return FWX_ERR_INVALID;
        m->dom_bits = bits;
        m->dom_known = 1;
        return FWX_OK;
    });
}

int fwx_matrix_set_timing(fwx_matrix *m, int32_t on)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (!m->multi) return FWX_ERR_UNSUPPORTED;
        m->multi->timer.on = on != 0;
        return FWX_OK;
    });
}

int fwx_matrix_get_timing(const fwx_matrix *m, fwx_multi_timing *out)
{
    return fwxi::guarded([&]() -> int {
        if (!m || !out || out->struct_size < sizeof(fwx_multi_timing)) return FWX_ERR_INVALID;
        if (!m->multi) return FWX_ERR_UNSUPPORTED;
        *out = m->multi->timer.last;
        out->struct_size = (uint32_t)sizeof(fwx_multi_timing);
        return FWX_OK;
    });
}

int fwx_matrix_parts(const fwx_matrix *m, int32_t *exchange_out)
{
    return fwxi::guarded([&]() -> int {
        if (!m) return FWX_ERR_INVALID;
        if (exchange_out) *exchange_out = m->multi ? m->multi->exchange : FWX_XCHG_PEER;
        return m->multi ? m->multi->parts : 1;
    });
}

// ---- the one-shot entry points keep their handle ---------------------------------------------------
// fwx_solve_multi_* is stateless for the caller, like fwx_solve_*; what it needs -- the slabs, the
// panels, two streams and five events per partition, the RCCL communicator -- used to be created and
// destroyed per call (hipFree / hipStreamDestroy synchronise the device, ncclCommInitAll takes
// hundreds of milliseconds).  A finished call now parks its handle here, keyed by everything that
// shapes it; the next call with the same key takes it back and only uploads.  Handles whose rate
// slabs add up to more than kKeepBytes are destroyed as before (a one-off N = 16384 solve must not pin
// gigabytes),
// at most kMaxIdle are parked, and the pool is never torn down from a static destructor.
class MultiPool {
public:
    struct Key {
        int32_t n, dtype, with_next, with_hops, parts, exchange;
        int32_t devs[FWX_MAX_PARTS];
        bool operator==(const Key &o) const { return memcmp(this, &o, sizeof(Key)) == 0; }
    };
    static Key key(int32_t n, int dtype, bool with_next, bool with_hops, int32_t parts,
                   const int32_t *devices, int32_t exchange)
    {
        Key k;
        memset(&k, 0, sizeof(k));
        k.n = n; k.dtype = dtype; k.with_next = with_next; k.with_hops = with_hops; k.parts = parts;
        k.exchange = exchange;
        for (int p = 0; p < parts; ++p) k.devs[p] = devices[p];
        return k;
    }
    static fwx_matrix *take(const Key &k)
    {
        Pool &pl = pool();
        std::lock_guard<std::mutex> lk(pl.mu);
        for (size_t i = 0; i < pl.idle.size(); ++i)
            if (pl.idle[i].k == k) {
                fwx_matrix *m = pl.idle[i].m;
                pl.idle.erase(pl.idle.begin() + (long)i);
                return m;
            }
        return nullptr;
    }
    static void park(const Key &k, fwx_matrix *m)
    {
        fwx_matrix *evict = nullptr;
        if (m->multi->slab_bytes > kKeepBytes) {
            evict = m;
        } else {
            Pool &pl = pool();
            std::lock_guard<std::mutex> lk(pl.mu);
            if (pl.idle.size() >= kMaxIdle) {       // the oldest goes
                evict = pl.idle.front().m;
                pl.idle.erase(pl.idle.begin());
            }
            pl.idle.push_back({k, m});
        }
        if (evict) fwx_matrix_destroy(evict);
    }

private:
    static constexpr size_t kKeepBytes = (size_t)256 << 20, kMaxIdle = 2;
    struct Entry { Key k; fwx_matrix *m; };
    struct Pool { std::mutex mu; std::vector<Entry> idle; };
    static Pool &pool() { static Pool *p = new Pool(); return *p; }   // leaked on purpose
};

static int solve_multi_host(int32_t n, int dtype, void *rate, int32_t *next, int32_t *hops, int32_t n_parts,
                            const int32_t *devices, int32_t exchange, const fwx_opts *opts)
{
    if (n < 0) return FWX_ERR_INVALID;
    if (n == 0) return FWX_OK;
    if (!rate || (hops && !next)) return FWX_ERR_INVALID;
    if (n_parts < 1 || n_parts > FWX_MAX_PARTS || !devices) return FWX_ERR_INVALID;
    Opts op;
    int rc = read_opts(opts, n, op);
    if (rc) return rc;
    int32_t resolved[FWX_MAX_PARTS];                  // -1 = the caller's current device: the pool keys on ordinals
    {
        int cur = 0;
        if (device_count() <= 0) return FWX_ERR_NO_DEVICE;
        FWX_HIP(hipGetDevice(&cur));
        for (int p = 0; p < n_parts; ++p) resolved[p] = devices[p] == -1 ? cur : devices[p];
        devices = resolved;
    }
    const MultiPool::Key key = MultiPool::key(n, dtype, next != nullptr, hops != nullptr, n_parts, devices,
                                              exchange);
    // the handle is destroyed on every path that does not park it (an error or an exception leaves
    // it in an unknown state)
    struct Holder {
        fwx_matrix *m = nullptr;
        ~Holder() { if (m) fwx_matrix_destroy(m); }
    } h;
    h.m = MultiPool::take(key);
    if (!h.m && (rc = fwx_matrix_create_multi(&h.m, n, dtype, next != nullptr, hops != nullptr, n_parts,
                                              devices, exchange)))
        return rc;
    rc = multi_upload(h.m, rate, next, hops);
    if (!rc) rc = multi_solve(h.m, op);
    if (!rc) rc = multi_download(h.m, rate, next, hops);
    if (rc) return rc;
    fail_point();
    fwx_matrix *m = h.m;
    h.m = nullptr;
    MultiPool::park(key, m);
    return FWX_OK;
}

int fwx_solve_multi_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, int32_t n_parts,
                        const int32_t *devices, int32_t exchange, const fwx_opts *opts)
{
    return fwxi::guarded([&]() -> int {
        return solve_multi_host(n, FWX_F64, rate, next, hops, n_parts, devices, exchange, opts);
    });
}

int fwx_solve_multi_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, int32_t n_parts,
                        const int32_t *devices, int32_t exchange, const fwx_opts *opts)
{
    return fwxi::guarded([&]() -> int {
        return solve_multi_host(n, FWX_F32, rate, next, hops, n_parts, devices, exchange, opts);
    });
}

}  // extern "C"
