// fwx_internal.h -- pieces shared by the translation units behind the C ABI (fwx_api.hip: one
// device; fwx_multi.hip: the row-partitioned multi-device handle).  Not installed, not part of the ABI.
#ifndef FWX_INTERNAL_H
#define FWX_INTERNAL_H

#include <hip/hip_runtime.h>
#include <mutex>
#include <new>
#include <vector>
#include <stddef.h>
#include <stdlib.h>
#include <stdint.h>

#include "fwx.h"
#include "fwx_kernels.h"

static_assert(FWX_UPDATE_SHARDS == FWX_UPDATE_SHARDS_K, "shard count mismatch");

namespace fwxi {

inline thread_local int g_last_hip = 0;

#define FWX_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess) {                                                                   \
            g_last_hip = (int)e__;                                                                 \
            (void)hipGetLastError();                                                               \
            return e__ == hipErrorOutOfMemory ? FWX_ERR_OOM : FWX_ERR_HIP;                         \
        }                                                                                          \
    } while (0)

inline int device_count()
{
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return c;
}

// Sets the requested device for the scope of one ABI call and restores the caller's.
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    int enter(int device)
    {
        const int cnt = device_count();
        if (cnt <= 0) return FWX_ERR_NO_DEVICE;
        if (hipGetDevice(&prev) != hipSuccess) return FWX_ERR_HIP;
        if (device < 0) return FWX_OK;
        if (device >= cnt) return FWX_ERR_INVALID;
        if (device != prev) {
            FWX_HIP(hipSetDevice(device));
            changed = true;
        }
        return FWX_OK;
    }
    ~DeviceGuard()
    {
        if (changed) (void)hipSetDevice(prev);
    }
};

struct Opts {
    int device = -1, engine = FWX_ENGINE_AUTO, k_begin = 0, k_end = 0, block = 0, serpentine = 1;
    uint64_t *updates_out = nullptr;
    hipStream_t stream = nullptr;      // caller's stream (fwx_opts.stream), nullptr = library-owned
    bool has_stream = false;
};

// The stream one blocking ABI call runs on: the caller's (fwx_opts.stream) or a non-blocking stream
// of its own -- never the legacy null stream, which would serialise the call against every other
// blocking stream of the process (torch's included) and against solves on other host threads.
// Bounds the number of launches in flight on a stream: every EVERY launches an event is recorded
// and the host waits for the event recorded 2*EVERY launches earlier.  The GPU never idles (at
// least EVERY launches are queued behind the one being waited for), but a solve of N = 16384
// pivots no longer parks 16384 dispatches in the queue: rocprofv3's counter collection, which
// intercepts every AQL packet, crashed on exactly that.
// Its events come from a process-wide pool per device and go back there, recorded or not: an event is
// never destroyed while a command may still reference it (round 3 waited for armed events in the
// destructor instead, which made the asynchronous entry points -- fwx_dev_relax with 256 or more pivots
// -- block the host until their last recorded event had completed; fwx.h promises "nothing is
// synchronised"), and a call creates no event once the pool is warm.  Re-recording a pooled event whose
// earlier record is still pending is legal HIP; a Throttle only ever waits for records it made itself.
class ThrottleEvents {
public:
    static hipEvent_t take(int dev)
    {
        Pool &p = pool();
        {
            std::lock_guard<std::mutex> lk(p.mu);
            if (dev >= 0 && dev < kMaxDev && !p.idle[dev].empty()) {
                hipEvent_t e = p.idle[dev].back();
                p.idle[dev].pop_back();
                return e;
            }
        }
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            g_last_hip = (int)hipGetLastError();
            return nullptr;
        }
        return e;
    }
    static void give(int dev, hipEvent_t e)
    {
        if (!e) return;
        Pool &p = pool();
        std::lock_guard<std::mutex> lk(p.mu);
        if (dev >= 0 && dev < kMaxDev) p.idle[dev].push_back(e);     // (else: leaked, never destroyed pending)
    }

private:
    static constexpr int kMaxDev = 64;
    struct Pool { std::mutex mu; std::vector<hipEvent_t> idle[kMaxDev]; };
    static Pool &pool() { static Pool *p = new Pool(); return *p; }   // leaked on purpose
};

struct Throttle {
    static constexpr int EVERY = 256;
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool armed[2] = {false, false};
    int count = 0, slot = 0, dev = -1;
    ~Throttle()
    {
        for (int i = 0; i < 2; ++i) ThrottleEvents::give(dev, ev[i]);
    }
    // the stream's device must be current
    int tick(hipStream_t s, int launches = 1)
    {
        count += launches;
        if (count < EVERY) return FWX_OK;
        count = 0;
        if (!ev[slot]) {
            if (dev < 0) FWX_HIP(hipGetDevice(&dev));
            if (!(ev[slot] = ThrottleEvents::take(dev))) return FWX_ERR_HIP;
        }
        if (armed[slot]) FWX_HIP(hipEventSynchronize(ev[slot]));
        FWX_HIP(hipEventRecord(ev[slot], s));
        armed[slot] = true;
        slot ^= 1;
        return FWX_OK;
    }
};

inline int read_opts(const fwx_opts *o, int n, Opts &out)
{
    if (o) {
        // v1 callers pass the struct up to and including updates_out; later fields are optional
        if (o->struct_size < offsetof(fwx_opts, stream)) return FWX_ERR_INVALID;
        out.device = o->device;
        out.engine = o->engine;
        out.k_begin = o->k_begin;
        out.k_end = o->k_end;
        out.block = o->block;
        out.serpentine = o->serpentine == 0 ? 1 : 0;
        out.updates_out = o->updates_out;
        if (o->struct_size >= offsetof(fwx_opts, use_stream) + sizeof(int32_t) && o->use_stream) {
            out.stream = (hipStream_t)o->stream;
            out.has_stream = true;
        }
    }
    if (out.k_end <= 0) out.k_end = n;
    if (out.k_begin < 0 || out.k_begin > out.k_end || out.k_end > n) return FWX_ERR_INVALID;
    if (out.engine != FWX_ENGINE_AUTO && out.engine != FWX_ENGINE_PERK &&
        out.engine != FWX_ENGINE_FUSED)
        return FWX_ERR_INVALID;
    return FWX_OK;
}


inline int sum_updates(unsigned long long *d_updates, uint64_t *out, hipStream_t s)
{
    unsigned long long h[FWX_UPDATE_SHARDS];
    FWX_HIP(hipMemcpyAsync(h, d_updates, sizeof(h), hipMemcpyDeviceToHost, s));
    FWX_HIP(hipStreamSynchronize(s));
    uint64_t u = 0;
    for (int i = 0; i < FWX_UPDATE_SHARDS; ++i) u += h[i];
    *out = u;
    return FWX_OK;
}


// Retire barrier.  hipStreamSynchronize returns when the awaited command's status is set; the HIP
// runtime's signal-handler thread may still be RETIRING that command (releasing its references to the
// memory objects behind the kernel's pointer arguments).  The handler retires the commands of a queue
// strictly in order, so one more trivial command on the stream, waited for, proves that everything
// before it has been retired; its own retirement only touches a buffer that is never freed.  libfwx
// runs this before it releases anything a stream's commands used (buffers, events, the stream).
//
// Why it exists: tools/fuzz_domain.py died with a SIGSEGV in that handler thread (round 2, only under
// the HIP runtime bundled with the torch wheel, ROCm 7.0, not the /opt/rocm 7.2 one libfwx is built
// against): amd::KernelParameters::release -> amd::ReferenceCountedObject::release on the memory
// object of a kernel's pointer argument (DESIGN.md section 7 has the disassembly-level record).  This
// barrier made the crash rarer but did NOT remove it under that runtime, so the cause is not proven
// to be a hipFree racing the handler; it is kept as lifetime hygiene, not as the fix.
inline void drain_stream(hipStream_t s)
{
    static std::mutex mu;
    static void *pad[64] = {nullptr};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return;
    void *p = nullptr;
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!pad[dev] && hipMalloc(&pad[dev], 256) != hipSuccess) { (void)hipGetLastError(); pad[dev] = nullptr; }
        p = pad[dev];
    }
    if (s) (void)hipStreamSynchronize(s);
    if (p && s) {
        (void)hipMemsetAsync(p, 0, 4, s);
        (void)hipStreamSynchronize(s);
    }
}

struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t rows_done = nullptr, panel_done = nullptr, main_done = nullptr;
    ~SideStream()
    {
        if (main_done) (void)hipEventDestroy(main_done);
        if (rows_done) (void)hipEventDestroy(rows_done);
        if (panel_done) (void)hipEventDestroy(panel_done);
        if (s) (void)hipStreamDestroy(s);
    }
    void drain() { if (s) drain_stream(s); }
    int init()
    {
        FWX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        FWX_HIP(hipEventCreateWithFlags(&rows_done, hipEventDisableTiming));
        FWX_HIP(hipEventCreateWithFlags(&panel_done, hipEventDisableTiming));
        FWX_HIP(hipEventCreateWithFlags(&main_done, hipEventDisableTiming));
        return FWX_OK;
    }
};


// ------------------------------------------------------------------------------------------------
// Per-call context.  fwx_solve_f64 / _f32 / fwx_dev_solve are stateless for the caller, but what a
// call needs on the device -- a stream, the look-ahead stream and its events, device buffers for the
// caller's arrays, the fused engine's workspace -- is expensive to create and, above all, to
// release (hipFree and hipStreamDestroy synchronise the device): 5-12 ms per call whatever the
// matrix order, for a 4 x 4 solve that takes 20 us.  Contexts are therefore kept in a
// process-wide pool, one per concurrent call and device; a call takes one, grows its buffers if it
// must, and puts it back.  Buffers above kKeepBytes are released on the way back, so a one-off
// N = 16384 solve does not pin gigabytes; the pool itself is never torn down (no HIP calls from
// static destructors).
// ------------------------------------------------------------------------------------------------
struct CallCtx {
    enum { RATE, NEXT, HOPS, WS, SMALL, NBUF };            // SMALL: update shards + domain flag
    static constexpr size_t kKeepBytes = (size_t)256 << 20;
    int device = -1;
    hipStream_t s = nullptr;
    SideStream side;
    void *buf[NBUF] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap[NBUF] = {0, 0, 0, 0, 0};

    void *pin = nullptr;               // pinned host staging for small host <-> device transfers
    size_t pin_cap = 0;
    // solves whose arrays sum to more than this go straight from / to the caller's memory (measured,
    // fwx_solve_f64 + next + hops: staged wins up to n = 256 = 1 MiB, 0.92 against 1.04 ms, and loses
    // at n = 512 = 4 MiB, 2.07 against 1.91 ms; gpurun_out/r02_call_latency_staged.txt)
    static constexpr size_t kStageBytes = (size_t)3 << 19;
    int reserve_pinned(size_t bytes, void **out)
    {
        if (pin_cap < bytes) {
            if (pin) { drain(); (void)hipHostFree(pin); pin = nullptr; pin_cap = 0; }
            const size_t want = bytes < ((size_t)1 << 20) ? ((size_t)1 << 20) : bytes;
            FWX_HIP(hipHostMalloc(&pin, want, hipHostMallocDefault));
            pin_cap = want;
        }
        *out = pin;
        return FWX_OK;
    }
    // Streams that are not the context's own but ran commands on its buffers during the current
    // lease: the caller's stream (fwx_opts.stream) or a handle's stream (the batch queries borrow
    // scratch from a context).  They are drained with the context's streams before any buffer is
    // released, regrown or handed to the next lease.
    hipStream_t foreign[2] = {nullptr, nullptr};
    void uses_stream(hipStream_t st)
    {
        if (!st || st == s || st == side.s || st == foreign[0] || st == foreign[1]) return;
        if (!foreign[0]) foreign[0] = st;
        else if (!foreign[1]) foreign[1] = st;
        else { drain_stream(foreign[0]); foreign[0] = st; }
    }
    void drain()
    {
        drain_stream(s);
        side.drain();
        for (hipStream_t &f : foreign)
            if (f) drain_stream(f);
    }
    int reserve(int which, size_t bytes, void **out)
    {
        if (cap[which] < bytes) {
            if (buf[which]) { drain(); (void)hipFree(buf[which]); buf[which] = nullptr; cap[which] = 0; }
            const size_t want = bytes < 4096 ? 4096 : bytes + bytes / 8;      // a little head room
            if (hipMalloc(&buf[which], want) != hipSuccess) {
                (void)hipGetLastError();
                FWX_HIP(hipMalloc(&buf[which], bytes));                       // exactly, then
                cap[which] = bytes;
            } else {
                cap[which] = want;
            }
        }
        *out = buf[which];
        return FWX_OK;
    }
    void trim()
    {
        bool any = false;
        for (int i = 0; i < NBUF; ++i) any = any || cap[i] > kKeepBytes;
        if (any) drain();
        for (int i = 0; i < NBUF; ++i)
            if (cap[i] > kKeepBytes) { (void)hipFree(buf[i]); buf[i] = nullptr; cap[i] = 0; }
    }
    void destroy()
    {
        drain();
        if (pin) (void)hipHostFree(pin);
        for (int i = 0; i < NBUF; ++i)
            if (buf[i]) (void)hipFree(buf[i]);
        if (s) (void)hipStreamDestroy(s);
    }
};

class CtxPool {
public:
    // The current device must already be the call's device (DeviceGuard).
    static int acquire(CallCtx **out)
    {
        int dev = 0;
        FWX_HIP(hipGetDevice(&dev));
        Pool &p = pool();
        {
            std::lock_guard<std::mutex> lk(p.mu);
            for (size_t i = 0; i < p.free_.size(); ++i)
                if (p.free_[i]->device == dev) {
                    *out = p.free_[i];
                    p.free_.erase(p.free_.begin() + (long)i);
                    return FWX_OK;
                }
        }
        CallCtx *c = new (std::nothrow) CallCtx();
        if (!c) return FWX_ERR_OOM;
        c->device = dev;
        if (hipStreamCreateWithFlags(&c->s, hipStreamNonBlocking) != hipSuccess) {
            g_last_hip = (int)hipGetLastError();
            delete c;
            return FWX_ERR_HIP;
        }
        *out = c;
        return FWX_OK;
    }
    static void release(CallCtx *c)
    {
        if (!c) return;
        c->trim();
        Pool &p = pool();
        {
            std::lock_guard<std::mutex> lk(p.mu);
            if (p.free_.size() < kMaxFree) { p.free_.push_back(c); return; }
        }
        c->destroy();          // more concurrent callers than the pool keeps: this one goes
        delete c;
    }

private:
    static constexpr size_t kMaxFree = 8;
    struct Pool { std::mutex mu; std::vector<CallCtx *> free_; };
    static Pool &pool() { static Pool *p = new Pool(); return *p; }   // leaked on purpose
};

// A context for the duration of one call.
struct CtxLease {
    CallCtx *c = nullptr;
    int open() { return CtxPool::acquire(&c); }
    ~CtxLease()
    {
        if (c) {       // an error return may leave work queued: the context goes back idle
            (void)hipStreamSynchronize(c->s);
            if (c->side.s) (void)hipStreamSynchronize(c->side.s);
            for (hipStream_t &f : c->foreign)
                if (f) { drain_stream(f); f = nullptr; }   // the stream belongs to someone else: forget it
        }
        CtxPool::release(c);
    }
};

inline size_t fused_ws_bytes(int n, size_t es, bool with_hops)
{
    const size_t ld = ((size_t)n + 3) & ~(size_t)3;
    // FOUR panel sets -- W, Ct, CNt, and with hops WH, CHt -- of 64 pivots each: the double-pass
    // schedule (fused_range) applies two passes per main launch while the next two are produced; the
    // single-pass schedules use the first two sets
    size_t b = (size_t)4 * FWX_FUSED_B * ((size_t)n * es + ld * (es + 4)) + 256;
    if (with_hops) b += (size_t)4 * FWX_FUSED_B * ((size_t)n * 4 + ld * 4);
    return b;
}

// One read of the matrix (fwx.h "Domain"): bit 0 = every rate is >= +0.0 and not NaN; bit 1 = no
// entry has a non-zero rate and next < 0.  d_flag: a device int the caller owns.
template <typename T>
int domain_bits(const T *rate, const int32_t *next, size_t count, int *d_flag, hipStream_t s, int &bits)
{
    int h = 3;
    FWX_HIP(hipMemcpyAsync(d_flag, &h, sizeof(int), hipMemcpyHostToDevice, s));
    FWX_HIP(fwx::launch_nonneg_check(rate, next, count, d_flag, s));
    FWX_HIP(hipMemcpyAsync(&h, d_flag, sizeof(int), hipMemcpyDeviceToHost, s));
    FWX_HIP(hipStreamSynchronize(s));
    bits = h;
    return FWX_OK;
}


// One launch per pivot over a slab; pivot rows from `prow0 + (k-k_begin)*stride`.
template <typename T>
inline int relax_range(T *rate, int32_t *next, int32_t *hops, int rows, int n, int row0, const T *prow0,
                const int32_t *phops0, int64_t stride, int k_begin, int k_end, int serpentine,
                unsigned long long *d_updates, hipStream_t s, fwx::PathLog plog = fwx::PathLog(),
                int skip_lo = 0, int skip_hi = 0, const int32_t *pnext0 = nullptr)
{
    fwx::RelaxArgs<T> a;
    Throttle thr;
    a.rate = rate; a.next = next; a.hops = hops;
    a.rows = rows; a.n = n; a.row0 = row0; a.updates = d_updates; a.plog = plog;
    a.skip_lo = skip_lo; a.skip_hi = skip_hi;
    for (int k = k_begin; k < k_end; ++k) {
        a.prow = prow0 + (int64_t)(k - k_begin) * stride;
        a.phops = phops0 ? phops0 + (int64_t)(k - k_begin) * stride : nullptr;
        a.pnext = pnext0 ? pnext0 + (int64_t)(k - k_begin) * stride : nullptr;
        a.k = k;
        a.flip = serpentine ? (k & 1) : 0;
        const hipError_t e = fwx::launch_relax<T>(a, s);
        if (e == hipErrorInvalidValue) return FWX_ERR_INVALID;   // misaligned skip range
        FWX_HIP(e);
        const int rc = thr.tick(s);
        if (rc) return rc;
    }
    return FWX_OK;
}

// The path trace `off` elements further on (e.g. at pivot row k0: off = k0 * n); null stays null.
inline fwx::PathLog plog_rows(fwx::PathLog p, size_t off)
{
    if (p.last) { p.last += off; p.at_col += off; p.at_row += off; }
    return p;
}

struct MultiState;   // fwx_multi.hip: the partitions of a multi-device handle

// Resumable solves (fwx_matrix_enable_resume): what a handle keeps so
// that a solve of a PATCHED input can start at a stored state instead of at pivot 0.
//   panels      the time-k snapshots the fused engine produces anyway, for ALL pivots instead of two
//               ping-pong buffers: w[k][j] = row k at time k, ct[k][i] = column k at time k (NaN at
//               i == k), cnt / wh / cht likewise for next-hops and hops
//   checkpoint  a copy of the state (rate, next, hops, the three trace arrays) at the START of step
//               pivot[c], a multiple of 64
// An input entry (i,j) is an OPERAND only in steps i and j, so patched entries cannot influence any
// other entry before step m = min over their indices: the state at a checkpoint <= m is the stored
// one except for the patched entries themselves, and those are replayed through the pivots before
// the checkpoint from the stored panels (their operands (i,k), (k,j), k < m, are not patched).
struct Resume {
    int count = 0, ld = 0;
    std::vector<int> pivot;                        // ascending, each a multiple of 64 in (0, n)
    std::vector<void *> rate;
    std::vector<int32_t *> next, hops, last, at_col, at_row;
    void *w = nullptr, *ct = nullptr;              // n x n, n x ld elements of the handle's dtype
    int32_t *cnt = nullptr, *wh = nullptr, *cht = nullptr;
    int64_t *idx = nullptr;                        // FWX_MAX_PATCH entry indices of a resolve, on the device
    int valid_upto = 0;    // panels of pivots [0, valid_upto) and checkpoints with pivot <= valid_upto
                           // belong to the solve of the CURRENT kept input (0: nothing to resume from)
    int state_at = -1;     // the live arrays hold the kept input brought to the start of step state_at
                           // (0 right after an upload / patch; -1: unknown, e.g. solved twice over)
};

}  // namespace fwxi

struct fwx_matrix {
    int32_t n, dtype, device;
    int32_t nd;            // device order = pitch of every array below: n rounded up to a multiple of 16
                           // bytes of rate elements, so that the fused engine reads any n.  The padding
                           // (rate +0.0, next -1, hops 0, trace -1; written once, at create / enable) is
                           // inert: a padding index is never a pivot, and a +0.0 target never improves
                           // (0 < +-0 and 0 < NaN are false) -- as in fwx_solve_* and the partitioned handle
    void *rate;
    int32_t *next, *hops, *scratch;
    unsigned long long *upd;
    fwx::PathLog plog;     // path trace for exact `_path` lists (last == nullptr: disabled)
    int32_t *next0;        // the uploaded next-hop matrix: the path of an entry never improved
    void *rate0;           // kept input (fwx_matrix_keep_input): the uploaded rates ...
    int32_t *hops0;        // ... and hops (next0 serves both purposes)
    int32_t keep;          // the input is kept on the device
    int32_t kept_valid;    // ... and holds an upload
    int32_t *walk;         // scratch of the exact-path walk (stack + output)
    int32_t walk_cap;      // capacity (path entries) `walk` was sized for
    int32_t rec_ready;     // a traced solve of the current upload has completed
    int32_t fresh;         // the arrays hold an uploaded input that has not been solved yet
    unsigned long long last_u;   // U of the last traced solve
    hipStream_t stream;    // the handle's own non-blocking stream: every operation on the handle runs
                           // on it (never the legacy null stream), so handles on different host
                           // threads overlap and nothing synchronises with torch's streams
    void *ws;              // fused-engine workspace, allocated by the first fused solve and kept
    size_t ws_bytes;
    fwxi::SideStream *side;      // look-ahead stream + events of the fused engine, kept likewise
    int *flag;             // device int for the domain check
    int32_t dom_known;     // dom_bits is the domain check's answer for what the arrays hold now.  The
    int32_t dom_bits;      // domain (fwx.h) is closed under the algorithm -- products of non-negative
                           // rates are >= +0 or NaN (which never wins), and a relaxation only succeeds
                           // through a non-zero r[i][k], whose next-hop it copies -- so only an upload
                           // or a patch can change the answer: the upload forgets it, a patch whose
                           // values are themselves inside the domain keeps a "3".
    fwxi::MultiState *multi;   // non-null: a row-partitioned handle (fwx_matrix_create_multi); the
                           // single-device arrays above are then unused
    fwxi::Resume *resume;  // non-null: panels of all pivots + state checkpoints are kept (f3)
};



// fwx_multi.hip: what the handle entry points of fwx_api.hip call for a handle with m->multi
namespace fwxi {
int multi_upload(fwx_matrix *m, const void *rate, const int32_t *next, const int32_t *hops);
int multi_download(fwx_matrix *m, void *rate, int32_t *next, int32_t *hops);
int multi_solve(fwx_matrix *m, const Opts &op, bool resumed = false);
int multi_enable_resume(fwx_matrix *m, int32_t checkpoints);
void multi_resume_dims(const fwx_matrix *m, uint64_t *cells, uint64_t *col_cells, uint64_t *w_cells);
int multi_resolve(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                  const int32_t *next_vals, const int32_t *hops_vals, int c_idx, Opts op);
// Does a patch with these values keep a matrix inside the domain (fwx.h) inside it?  rate >= +0 and not
// NaN; a non-zero rate comes with a next-hop >= 0 on a handle that carries next-hops.
inline bool patch_keeps_domain(const fwx_matrix *m, int32_t count, const void *rate_vals, const int32_t *next_vals)
{
    for (int32_t q = 0; q < count; ++q) {
        const double r = m->dtype == FWX_F64 ? ((const double *)rate_vals)[q] : (double)((const float *)rate_vals)[q];
        if (r != r || r < 0.0 || (r == 0.0 && 1.0 / r < 0.0)) return false;       // NaN, negative, -0.0
        if (m->next && r != 0.0 && !(next_vals && next_vals[q] >= 0)) return false;
    }
    const int want = m->next ? 3 : 1;
    return (m->dom_bits & want) == want;
}
int multi_enable_path_log(fwx_matrix *m);
int multi_query(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out, int32_t *path_out, int32_t cap);
int multi_query_exact(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out, int32_t *path_out,
                      int32_t cap);
int multi_query_exact_batch(fwx_matrix *m, int32_t count, const int32_t *src, const int32_t *dst,
                            int32_t *len_out, int32_t *path_out, int32_t cap);
void multi_destroy(fwx_matrix *m);
int multi_keep_input(fwx_matrix *m);
int multi_patch_input(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                      const int32_t *next_vals, const int32_t *hops_vals);
}  // namespace fwxi

#endif
