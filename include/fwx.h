/*
 * fwx.h -- C ABI of libfwx, the MI355X (gfx950) max-product Floyd-Warshall engine.
 *
 * Drop-in boundary.  The reference has no FFI; its seam is the pure function
 *     floydWarshall :: Map (Vertex,Vertex) Double -> Matrix RateEntry
 *         /root/reference/src/lib/Algorithms.hs:19-20   (= runAlgo 0 . buildMatrix)
 * whose replaceable core is
 *     runAlgo :: Int -> Matrix RateEntry -> Matrix RateEntry
 *         /root/reference/src/lib/Algorithms.hs:42-61
 * The host keeps buildMatrix (:26-40) and optimum (:65-78); this library replaces runAlgo on the
 * dense structure-of-arrays form of `Matrix RateEntry` (Types.hs:24-29, :39):
 *
 *     rate[n*n]  row-major, T = double (the reference's precision, Types.hs:26) or float
 *     next[n*n]  int32: index of `head _path`, -1 for the empty path        (optional)
 *     hops[n*n]  int32: `length _path`                                      (optional)
 *
 * `_start` of row i is implicit (vertex i).  "No route" is rate 0.0 / next -1 / hops 0
 * (isolatedEntry, Utils.hs:13-14).
 *
 * Semantics preserved exactly: k ascending (:44); row k copied (:50);
 * entries j==i or j==k untouched (:54); operands from the state at the start of step k (:58-60);
 * one IEEE multiply (:61) and one strict ordered compare (:55) per relaxation -- no fast-math, no
 * FMA, denormals kept, NaN compares false.  Results are bit-identical to the reference loop.
 *
 * Errors: every entry point returns FWX_OK (0) or a negative fwx_status; no C++ exception or
 * abort crosses this boundary.  n == 0 is success and touches nothing (the reference returns the
 * empty matrix: src/test/AlgorithmsTest.hs:45-47, :62-64).  There is NO CPU fallback: without a
 * HIP device the solve entry points return FWX_ERR_NO_DEVICE.
 *
 * Domain.  The reference's parser only admits rates > 0 (Parsers.hs:40) and buildMatrix gives every
 * entry it fills a one-vertex path (Algorithms.hs:35-37), so in the reference every matrix obeys
 *     (D1) every rate is >= +0.0 and not NaN          (D2) rate != 0  =>  next >= 0  (path non-empty)
 * On that domain a winning product has two positive factors, hence a non-empty ikPath, and
 * head (ikPath ++ kjPath) = next[i][k].  The engine does NOT assume the domain: it checks it on the
 * device (one read of the matrix) before a fused solve, and a matrix outside it -- negative or NaN
 * rates, or a positive rate with next == -1 -- is solved by the per-k engine, which implements the
 * list rule in full (head = next[i][k] if that is >= 0, else next[k][j]).  Either way rates, next,
 * hops and U equal the reference's loop bit for bit.  Only the slab entry points (fwx_dev_relax*,
 * which see one slab and not the matrix) leave the check to the caller: see fwx_pivots.next and
 * fwx_dev_check_nonneg.
 *
 * Threading / streams: calls are blocking unless a stream is passed explicitly (fwx_dev_*, or
 * fwx_opts.stream), may come from any OS thread, keep no global mutable state besides a
 * mutex-protected pool of per-call contexts, and restore the
 * caller's current HIP device.  No entry point uses the legacy null stream: one-shot calls run on
 * a non-blocking stream of their own (taken, with the device buffers and workspace the call needs,
 * from a process-wide pool of per-call contexts: a call creates and frees nothing on the device
 * once the pool is warm), a handle on the handle's own non-blocking stream, so solves
 * on different host threads overlap on the device and nothing synchronises implicitly with the
 * streams of other libraries in the process (torch's included).
 */
#ifndef FWX_H
#define FWX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FWX_ABI_VERSION 3

typedef enum fwx_status {
    FWX_OK = 0,
    FWX_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, bad range, misaligned) */
    FWX_ERR_NO_DEVICE = -2,   /* no HIP device visible: the engine has no CPU fallback            */
    FWX_ERR_HIP = -3,         /* a HIP runtime call failed (fwx_last_hip_error gives the code)    */
    FWX_ERR_OOM = -4,         /* device or host allocation failed                                 */
    FWX_ERR_CYCLE = -5,       /* fwx_follow_path: next-hops do not reach dst within n hops        */
    FWX_ERR_CAPACITY = -6,    /* fwx_follow_path: output buffer too small                         */
    FWX_ERR_UNSUPPORTED = -7, /* option combination not implemented                               */
    FWX_ERR_RCCL = -8,        /* librccl.so.1 could not be loaded, or an RCCL call failed         */
    FWX_ERR_INTERNAL = -9     /* an unexpected C++ exception was stopped at the boundary          */
} fwx_status;

typedef enum fwx_dtype { FWX_F32 = 0, FWX_F64 = 1 } fwx_dtype;

/* Which relaxation engine runs the pivots.  All are bit-exact with the reference loop.
 * AUTO: n <= 64 -> the whole solve in one single-workgroup launch; above -> FUSED (two launches
 * per 64 pivots).  The fused kernels read rows in 16-byte vectors; buildMatrix (Algorithms.hs:29)
 * produces any n, so every entry point that OWNS its device memory -- fwx_solve_f32/f64, the handles
 * of fwx_matrix_create and fwx_matrix_create_multi -- keeps the matrix at a device order rounded up
 * to 4 (f32) / 2 (f64) with inert padding (rate +0.0, next -1: a padding index is never a pivot and
 * a +0.0 target never improves) and runs any engine on any n.  Where FUSED does not apply -- an
 * input outside the domain below, or caller-owned device memory (fwx_dev_*) whose rows are not a
 * multiple of 16 bytes -- AUTO takes the single-launch kernel up to n = 128 and PERK above; an
 * explicit FUSED on such caller-owned memory is refused (FWX_ERR_UNSUPPORTED).                     */
typedef enum fwx_engine {
    FWX_ENGINE_AUTO = 0,
    FWX_ENGINE_PERK = 1,  /* one N x N launch per pivot k (HBM-bound streaming kernel)            */
    FWX_ENGINE_FUSED = 2  /* B pivots per launch from time-k snapshots of the pivot panels         */
} fwx_engine;

/* Options; zero-initialise and set struct_size = sizeof(fwx_opts).  NULL means all defaults. */
typedef struct fwx_opts {
    uint32_t struct_size;
    int32_t device;        /* HIP device ordinal; -1 = the caller's current device                 */
    int32_t engine;        /* fwx_engine                                                           */
    int32_t k_begin;       /* first pivot (default 0): a solve over [k_begin,k_end) is resumable   */
    int32_t k_end;         /* one past the last pivot; <= 0 means n                                */
    int32_t block;         /* reserved, must be 0 (the fused engine works in FWX_FUSED_BLOCK pivots)*/
    int32_t serpentine;    /* 0 = default (on): alternate sweep direction per pivot so the tail of */
                           /* one launch is re-read from the Infinity Cache; 1 = off               */
    uint64_t *updates_out; /* host pointer, optional: receives U = number of successful updates    */
    void *stream;          /* hipStream_t to run on, honoured iff use_stream != 0 (fwx_matrix_solve,  */
    int32_t use_stream;    /* fwx_dev_solve, fwx_solve_*); the call still blocks until it is done.    */
    int32_t reserved0;     /* 0: a library-owned non-blocking stream (per call / per handle)          */
} fwx_opts;

int fwx_abi_version(void);
int fwx_device_count(void);              /* number of HIP devices, 0 if none (never an error)      */
const char *fwx_strerror(int status);
int fwx_last_hip_error(void);            /* hipError_t of the last FWX_ERR_HIP on this thread      */
/* HIP_VERSION libfwx was compiled against and the version of the HIP runtime it is bound to in this
 * process (hipRuntimeGetVersion; 0 if that fails).  Returns 1 if major.minor agree, else 0.  They
 * differ when another library mapped its own libamdhip64 first -- a torch wheel bundles one -- and
 * libfwx's DT_NEEDED resolved to that copy: it works, but it is not the pairing the library was
 * built and fuzzed on, so hosts should load libfwx before such a library or
 * at least log the mismatch (floydwarshall_amd/_lib.py does both).                                  */
int fwx_hip_versions(int32_t *built_against, int32_t *runtime);
/* TEST HOOK for the exception barrier: arms a countdown on the calling thread; the countdown-th
 * internal allocation point reached by later calls on this thread throws std::bad_alloc, which the
 * boundary must turn into FWX_ERR_OOM with nothing leaked.  0 disarms.  Not for production use.     */
int fwx_test_fail_after(int32_t countdown);

/* ---- one-shot host-buffer entry points: what the reference-side FFI binds --------------------
 * Replace runAlgo (Algorithms.hs:42-61) for a matrix produced by buildMatrix (:26-40).
 * rate/next/hops are n*n row-major HOST arrays, updated IN PLACE; next and hops may be NULL
 * (rates only).  The caller owns all buffers; no pointer is retained after return.            */
int fwx_solve_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, const fwx_opts *opts);
int fwx_solve_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, const fwx_opts *opts);

/* Index form of the `_path` list that optimum returns (Algorithms.hs:74-75): follow next-hops
 * from src until dst.  Returns the number of vertices written to out (dst included, src not), 0 if
 * next[src][dst] == -1 (empty path, "no exchange"), or a negative fwx_status.  Host arrays.      */
int fwx_follow_path(int32_t n, const int32_t *next, int32_t src, int32_t dst, int32_t *out,
                    int32_t cap);

/* ---- device-resident handle: keeps the solved matrix in HBM across queries --------------------
 * Host counterpart of `InSync exRates matrix` (Types.hs:35-37): upload once per rate change,
 * solve once, then answer any number of (src,dst) queries without re-solving.                   */
typedef struct fwx_matrix fwx_matrix;

int fwx_matrix_create(fwx_matrix **out, int32_t n, int32_t dtype, int32_t with_next,
                      int32_t with_hops, int32_t device);
/* upload: rate / next / hops may be HOST or DEVICE arrays (hipMemcpyDefault); download likewise. */
int fwx_matrix_upload(fwx_matrix *m, const void *rate, const int32_t *next, const int32_t *hops);
int fwx_matrix_solve(fwx_matrix *m, const fwx_opts *opts);
int fwx_matrix_download(fwx_matrix *m, void *rate, int32_t *next, int32_t *hops);
/* One entry + its path, read back from the device (8 + 4*len bytes instead of the matrix).      */
int fwx_matrix_query(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out, int32_t *path_out,
                     int32_t cap);
int fwx_matrix_destroy(fwx_matrix *m);

/* Incremental re-marshalling (SURVEY.md section 8f, row f3 -- the exact part of it).  The reference
 * rebuilds and re-solves the whole matrix after every accepted rate update
 * (ProcessRequests.hs:82-85, :99-102), although such an update changes TWO entries of
 * buildMatrix's output.  With the input KEPT on the device (one more copy of each array), the host
 * sends only the changed entries: fwx_matrix_patch_input replaces `count` entries of the kept input
 * (index[q] = i*n + j; rate_vals in the handle's dtype; next_vals / hops_vals may be NULL =
 * unchanged) and makes the patched input the handle's unsolved matrix again (device-to-device
 * restore), ready for fwx_matrix_solve -- a FULL solve, so every result stays bit-identical to the
 * reference; what is saved is buildMatrix over n^2 entries and the PCIe upload.
 * keep_input: call once after create (before or after an upload).  patch_input needs a kept
 * upload (FWX_ERR_INVALID otherwise); count <= FWX_MAX_PATCH.  Works on partitioned handles.     */
#define FWX_MAX_PATCH 4096
int fwx_matrix_keep_input(fwx_matrix *m);
int fwx_matrix_patch_input(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                           const int32_t *next_vals, const int32_t *hops_vals);

/* Resume instead of re-solve (row f3, exactly).  An input entry (i,j) is an OPERAND of runAlgo only in
 * steps i and j (Algorithms.hs:58-60: step k reads r[i][k] and r[k][j]), so changing entries cannot
 * influence any OTHER entry before step m = the smallest index among them: up to there the old solve
 * and the new one differ in the changed entries alone.  A resumable handle therefore keeps, from its
 * last solve, (a) the state (rate, next, hops, path trace) at a few CHECKPOINT pivots and (b) the
 * time-k snapshots of every pivot row and column -- the panels the fused engine produces anyway.
 * fwx_matrix_resolve(changed entries) restores the last checkpoint c <= m, replays just the changed
 * entries through pivots [0, c) from the stored panels (their operands there are unchanged entries),
 * and runs pivots [c, n) only.  Every operand, product and compare is the one a from-scratch solve
 * performs: results are bit-identical (rates, next, hops, exact `_path` lists); the saving is c / n of
 * the solve.  The checkpoints and panels are refreshed as the resumed solve passes them.
 *
 * enable_resume: after create, enable_path_log (if the trace is wanted: afterwards it is refused) and
 *   keep_input, before the upload whose solve is to be resumable; `checkpoints` in 1..FWX_MAX_CHECKPOINTS, spread evenly over the pivots on multiples
 *   of 64; returns the number placed (n < 128 leaves room for none: 0), or FWX_ERR_UNSUPPORTED where
 *   AUTO does not take the fused engine (n <= 64).  Any other n works (the handle pads its rows), on one
 *   device and on partitioned handles alike -- there every partition keeps the checkpoints of its slab, its
 *   own part of the pivot-column snapshots and ALL pivot rows (it receives them anyway), a changed entry is
 *   replayed on the partition that owns its row, and a checkpoint must be a block start: with partitions
 *   that do not start on a multiple of 64 fewer are placed.  Memory: one copy of every array per checkpoint
 *   + ~2.5 more for the panels (fwx_matrix_resume_bytes gives the figure).
 * resolve: fwx_matrix_patch_input + fwx_matrix_solve in one call, resuming where it can.  Falls back
 *   to exactly that pair (a full solve from the patched kept input, which records afresh) when there
 *   is nothing to resume from: no checkpoint at or below m, the previous solve did not record (other
 *   engine, input outside the reference's domain, a pivot range), a patch value outside the domain,
 *   or opts asking for U, a stream, a pivot range or the per-k engine.  *resumed_from (optional)
 *   receives the pivot the solve started at (0 = full solve).                                       */
#define FWX_MAX_CHECKPOINTS 16
int fwx_matrix_enable_resume(fwx_matrix *m, int32_t checkpoints);
/* What enable_resume(checkpoints) would allocate on the device(s) for this handle, in bytes (call it
 * after enable_path_log, like enable_resume itself), and the free / total memory of a device
 * (hipMemGetInfo; device -1 = the caller's current one): a host sizes its checkpoint count with these
 * -- the Session mirror keeps the resume memory below half of what is free and carries on WITHOUT
 * resuming when even one checkpoint does not fit or the allocation fails.                          */
int fwx_matrix_resume_bytes(const fwx_matrix *m, int32_t checkpoints, uint64_t *bytes_out);
int fwx_device_memory(int32_t device, uint64_t *free_bytes, uint64_t *total_bytes);
int fwx_matrix_resolve(fwx_matrix *m, int32_t count, const int64_t *index, const void *rate_vals,
                       const int32_t *next_vals, const int32_t *hops_vals, const fwx_opts *opts,
                       int32_t *resumed_from);

/* Exact `_path` lists.  Following next-hops (fwx_matrix_query) yields A best path; the reference
 * keeps, per entry, the list it concatenated when the entry was last improved
 * (`_path = ikPath ++ kjPath`, Algorithms.hs:55), and under exact ties (its built-in 1.0 edges
 * between the same currency on two exchanges make them common) that list can be a different,
 * longer route of equal rate.  With the PATH TRACE enabled the solve keeps three more n x n int32
 * matrices on the device -- for every entry the pivot of its newest successful relaxation, at the
 * end of the solve, at the start of the step named by its column index and at the start of the
 * step named by its row index -- from which fwx_matrix_query_exact rebuilds the reference's list
 * exactly: path(i,j) = path_q(i,q) ++ path_q(q,j) with q the newest pivot of (i,j), and a
 * sub-entry is only ever needed "as of the step named by one of its own indices".  No lists, no
 * update records, one pass; any n.  Enable after create; needs the next-hop matrix; every engine
 * keeps it (engine choice as for any matrix); whole pivot range only, and the solve starts from an
 * uploaded input (fwx_matrix_solve on an already solved traced matrix: FWX_ERR_INVALID).
 * query_exact before a completed traced solve of the current upload: FWX_ERR_INVALID.
 * fwx_matrix_path_log_count: U of that solve if it was asked to count (fwx_opts.updates_out),
 * else 0.  path_out receives the vertices after src up to dst; returns the length.              */
int fwx_matrix_enable_path_log(fwx_matrix *m);
int fwx_matrix_path_log_count(fwx_matrix *m, uint64_t *count_out);
int fwx_matrix_query_exact(fwx_matrix *m, int32_t src, int32_t dst, double *rate_out,
                           int32_t *path_out, int32_t cap);
/* Batch form: count (src, dst) pairs in one launch (one thread per pair).  len_out[q] = length of
 * list q or a negative fwx_status (FWX_ERR_CAPACITY: longer than cap); path_out + q*cap receives
 * it.  Host arrays in and out.                                                                  */
int fwx_matrix_query_exact_batch(fwx_matrix *m, int32_t count, const int32_t *src, const int32_t *dst,
                                 int32_t *len_out, int32_t *path_out, int32_t cap);

/* ---- row-partitioned solve behind the same call: ONE process, several devices ------------------
 * The reference has one call site, ProcessRequests.hs:82-84 -> floydWarshall (Algorithms.hs:19-20);
 * a host bound to that one call gets the whole node through these entry points.  The matrix is cut
 * into n_parts contiguous row blocks (partition p holds rows [n*p/P, n*(p+1)/P) of rate / next and
 * of the path trace -- the bounds rounded down to multiples of 64 for n >= 128 P, fwx_matrix_part_rows); step k needs, besides local values, only pivot row k AS IT STANDS AT THE
 * START OF STEP k, so pivots travel FWX_FUSED_BLOCK at a time as one snapshot panel per block,
 * produced by the partition that owns those rows and sent to all others, with look-ahead (the
 * owner of the next block relaxes those rows first and runs their panel + the exchange on a side
 * stream while every partition sweeps the rest of its slab).  No operand and no order changes:
 * results are bit-identical to the single-device solve and to the reference loop.
 *
 * devices[p] = HIP ordinal of partition p (-1 = the caller's current device).  A device may be listed MORE THAN ONCE (logical
 * partitions on one GPU: how the partitioned schedule is tested where only one GPU exists).
 * exchange:
 *   FWX_XCHG_RCCL  ncclBroadcast of each panel on RCCL (ncclCommInitAll over the devices, one
 *                  communicator per partition, calls grouped; librccl.so.1 is loaded on first use,
 *                  libfwx does not link it).  Needs pairwise distinct devices.  On this pool RCCL needs
 *                  HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment BEFORE the first HIP call of the
 *                  process (the host driver only supports dmabuf IPC; otherwise ncclCommInitAll fails
 *                  in hipIpcGetMemHandle and create_multi returns FWX_ERR_RCCL): a host exports it or
 *                  falls back to FWX_XCHG_PEER, which needs no IPC inside one process.
 *   FWX_XCHG_PEER  hipMemcpyPeerAsync from the owner's panel into every other partition's panel
 *                  buffer (plain device-to-device copies when the device is the same).
 *   FWX_XCHG_AUTO  RCCL when there are >= 2 partitions on pairwise distinct devices, else PEER.
 * Distinct devices get peer access enabled pairwise where the hardware allows; it is needed by the
 * QUERIES only (they walk the slabs from one device: without it fwx_matrix_query walks from the host
 * and query_exact[_batch] returns FWX_ERR_UNSUPPORTED), never by the exchange.
 *
 * A multi handle is an fwx_matrix: upload / solve / download / query / enable_path_log /
 * query_exact[_batch] / path_log_count / destroy work on it unchanged (pivot ranges as on one device:
 * any [k_begin, k_end) without the path trace, the whole range with it;
 * fwx_opts.engine AUTO / FUSED, or PERK = one launch per pivot and partition with the pivot rows
 * read from the exchanged panel -- rate / next / hops, no path trace; fwx_opts.device / stream are
 * ignored).  Any n: rows are padded on
 * the device to a multiple of 16 bytes.  rate, next, hops and the path trace are all carried
 * through the slabs (the hops of the pivot rows travel with their rates).  Not carried: matrices
 * outside the reference's domain WITH next-hops (see "Domain": the slab kernels take next[i][k];
 * such a matrix returns FWX_ERR_UNSUPPORTED from solve and is solved on one device by
 * fwx_solve_* / fwx_matrix_create).                                                               */
#define FWX_XCHG_AUTO 0
#define FWX_XCHG_PEER 1
#define FWX_XCHG_RCCL 2
#define FWX_XCHG_CALLBACK 3   /* fwx_matrix_create_part only */
#define FWX_MAX_PARTS 32
int fwx_matrix_create_multi(fwx_matrix **out, int32_t n, int32_t dtype, int32_t with_next,
                            int32_t with_hops, int32_t n_parts, const int32_t *devices,
                            int32_t exchange);
/* ---- the same partitioned solve, ONE PARTITION PER PROCESS (one process per GPU) -----------------
 * fwx_matrix_create_part builds the handle of partition `rank` of `world` row blocks: only that slab
 * lives in this process, and the one exchange of the algorithm -- the 64-pivot snapshot panel, from the
 * rank that owns those rows to every other -- is the host's: `exchange` is called on the solving thread
 * once per panel, in the same order on every rank, and must make `w` (count elements of the handle's
 * dtype) and, if not NULL, `wh` (count int32: the hops of the pivot rows) hold rank `owner`'s copy ON
 * EVERY RANK, ordered on `stream` (a hipStream_t: enqueue a broadcast on it -- RCCL, or MPI with the
 * stream synchronised around it; floydwarshall_amd/dist.py uses torch.distributed).  Return 0, or non-zero
 * to fail the solve (FWX_ERR_RCCL).  Everything else is the code the one-process handle runs -- same
 * schedules (single pass, the 128-pivot pair schedule, the per-k engine), same kernels, path trace, kept
 * input, resume -- and so the same bits.  What differs for the caller:
 *   upload / download  take THIS RANK'S row block (fwx_matrix_part_rows(m, rank, &row0, &rows)): rows x n
 *   the domain check   is per slab; the host combines the ranks' answers: fwx_matrix_domain_bits (local
 *                      bits: 1 = every rate >= +0 and not NaN, 2 = no positive rate without a path), an
 *                      all-reduce AND among the ranks, fwx_matrix_set_domain -- before the first solve of
 *                      every upload (a solve without it: FWX_ERR_INVALID)
 *   updates_out        receives this rank's share of U
 *   patch / resolve    take the caller's n x n entry indices on every rank; each rank applies its rows'
 *   queries            walk other ranks' slabs: FWX_ERR_UNSUPPORTED (download the slab instead)
 * Every rank must issue the same calls with the same options (the schedule is a function of n, world,
 * the options and the FWX_* environment), or the exchanges do not pair up.                           */
typedef int (*fwx_exchange_fn)(void *ctx, int32_t k0, int32_t bt, int32_t owner, void *w, int32_t *wh,
                               int64_t count, void *stream);
int fwx_matrix_create_part(fwx_matrix **out, int32_t n, int32_t dtype, int32_t with_next, int32_t with_hops,
                           int32_t rank, int32_t world, int32_t device, fwx_exchange_fn exchange, void *ctx);
int fwx_matrix_domain_bits(fwx_matrix *m, int32_t *bits_out);
int fwx_matrix_set_domain(fwx_matrix *m, int32_t bits);

/* Per-step timings of the last solve of a partitioned handle, from HIP events on the streams the work
 * ran on (a diagnostic: each span costs two event records; off by default).  A step is one block of 64
 * pivots, or a PAIR of blocks where the solve ran two passes per main launch (pivots_per_step = 128).
 *   bulk_us       what a step costs when nothing else limits it: the slab sweep (column panel + main
 *                 kernel / the 64 per-k launches) on a partition's main stream -- mean over steps of the
 *                 MAX over partitions; bulk_mean_us: the plain mean
 *   lookahead_us  the owner bringing the next block's rows up to date (single pass; main stream)
 *   panel_us      the owner's snapshot panel kernel (side stream)
 *   exchange_us   the panel's way to the other partitions: peer copy on each receiver / the RCCL
 *                 broadcast call on every partition's side stream -- mean over steps of the max
 *   chain_us      everything the NEXT step waits for besides the sweep.  Single pass: lookahead + panel
 *                 + exchange.  Double pass: measured whole on each partition's side stream, from the end
 *                 of the previous main launch to the last column panel of the next pair (cross launches,
 *                 both panels, both exchanges) -- mean over steps of the max over partitions
 *   chain_over_bulk > 1: the solve is bound by the panel chain, not by the sweep */
typedef struct fwx_multi_timing {
    uint32_t struct_size;
    int32_t steps, pivots_per_step, partitions;
    float bulk_us, bulk_mean_us, lookahead_us, panel_us, exchange_us, chain_us, chain_over_bulk;
} fwx_multi_timing;
int fwx_matrix_set_timing(fwx_matrix *m, int32_t on);      /* FWX_ERR_UNSUPPORTED on a single-device handle */
int fwx_matrix_get_timing(const fwx_matrix *m, fwx_multi_timing *out);   /* out->struct_size = sizeof(*out) */
/* Rows partition `part` holds: [*row0_out, *row0_out + *rows_out).  Balanced row blocks, n * p / P, rounded
 * down to a multiple of 64 once every partition holds two pivot blocks (n >= 128 P): blocks of 64 pivots never
 * straddle two partitions, so aligned partitions make every block full and aligned for ANY n, which is what
 * the 128-pivot pair schedule and the checkpoints of resumable solves need.  A host that feeds slabs
 * (fwx_matrix_create_part) asks here instead of computing the bounds itself.                        */
int fwx_matrix_part_rows(const fwx_matrix *m, int32_t part, int32_t *row0_out, int32_t *rows_out);
/* Partitions of a handle (1 for a single-device handle); exchange_out (optional) receives the
 * transport in use (FWX_XCHG_PEER / FWX_XCHG_RCCL).                                              */
int fwx_matrix_parts(const fwx_matrix *m, int32_t *exchange_out);
/* Ranks of the RCCL communicator the handle exchanges panels on (ncclCommCount), 0 when the
 * exchange is not RCCL (single device, FWX_XCHG_PEER).                                            */
int fwx_matrix_comm_ranks(const fwx_matrix *m);
/* One-shot host-buffer form, same contract as fwx_solve_f64 / _f32 (in place, caller owns the
 * arrays): create_multi + upload + solve + download.  Like the per-call contexts of fwx_solve_*,
 * the handle (slabs, streams, events, RCCL communicator) is parked in a process-wide pool keyed by
 * (n, dtype, fields, device list, exchange) and reused by the next call with the same key; handles
 * whose rate slabs exceed 256 MiB in total are destroyed on return.  opts: engine and updates_out are honoured.     */
int fwx_solve_multi_f64(int32_t n, double *rate, int32_t *next, int32_t *hops, int32_t n_parts,
                        const int32_t *devices, int32_t exchange, const fwx_opts *opts);
int fwx_solve_multi_f32(int32_t n, float *rate, int32_t *next, int32_t *hops, int32_t n_parts,
                        const int32_t *devices, int32_t exchange, const fwx_opts *opts);

/* ---- device-pointer step API (caller-owned DEVICE memory, caller's stream) --------------------
 * Used by the benchmark and by the row-partitioned multi-GPU driver, which own their buffers
 * through torch / torch.distributed.  All launches are asynchronous on `stream` (a hipStream_t,
 * NULL = default stream); nothing is synchronised.                                              */

/* `rows` consecutive rows [row0, row0+rows) of the global n x n matrix, rows*n elements each. */
typedef struct fwx_slab {
    int32_t n;       /* order of the global matrix = row length                                   */
    int32_t row0;    /* global index of the slab's first row                                      */
    int32_t rows;    /* rows held in this slab                                                    */
    int32_t dtype;   /* fwx_dtype                                                                 */
    void *rate;      /* device, rows*n                                                            */
    int32_t *next;   /* device, rows*n, or NULL                                                   */
    int32_t *hops;   /* device, rows*n, or NULL                                                   */
} fwx_slab;

/* Pivot rows for steps [k_begin, k_end): row k AS IT STANDS AT THE START OF STEP k is at
 * rate + (k - k_begin) * stride (elements).  For a single-GPU solve this is the matrix itself
 * (rate = matrix + k_begin*n, stride = n: row k is not modified by step k).  For a partitioned
 * solve it is the panel of time-k snapshots produced by fwx_dev_panel on the owner.             */
typedef struct fwx_pivots {
    int32_t k_begin;
    int32_t k_end;
    const void *rate;
    const int32_t *hops; /* same layout; required iff slab.hops != NULL                            */
    int64_t stride;
    const int32_t *next; /* same layout, optional: next-hop rows of the pivots at time k.  Given, an  */
                         /* update whose ikPath is empty (next[i][k] < 0) takes next[k][j], as the    */
                         /* reference's list concatenation does; NULL = the caller vouches for the    */
                         /* domain (see "Domain"), where that case cannot arise                        */
} fwx_pivots;

/* Apply pivots [k_begin,k_end) in order to every row of the slab (one launch per pivot).
 * d_updates: optional device array of FWX_UPDATE_SHARDS uint64 counters, incremented by U.      */
#define FWX_UPDATE_SHARDS 256
int fwx_dev_relax(const fwx_slab *slab, const fwx_pivots *piv, int32_t serpentine,
                  unsigned long long *d_updates, void *stream);
/* Same, leaving the slab rows [skip_lo, skip_hi) (relative to the slab, multiples of 4) alone: the
 * rows of the NEXT pivot panel, which a look-ahead step has already relaxed -- one launch per
 * pivot instead of one above and one below those rows.                                          */
int fwx_dev_relax_skip(const fwx_slab *slab, const fwx_pivots *piv, int32_t serpentine,
                       unsigned long long *d_updates, int32_t skip_lo, int32_t skip_hi, void *stream);

/* Owner-side panel phase for a partitioned solve.  `block` holds the pivot rows
 * [block->row0, block->row0 + block->rows) at time k = block->row0.  Evolves them in place through
 * those pivots and writes the time-k snapshot of each pivot row to w_rate (rows x n) (and its hops
 * row to w_hops if the slab carries hops).  Afterwards the block rows are at time row0+rows and
 * every other row of the matrix must be relaxed with fwx_dev_relax(pivots = w_rate).            */
int fwx_dev_panel(const fwx_slab *block, void *w_rate, int32_t *w_hops,
                  unsigned long long *d_updates, void *stream);

/* Whole solve on caller-owned DEVICE memory: `full` must hold the entire matrix (row0 = 0,
 * rows = n).  Same engines and options as fwx_matrix_solve (opts->device is ignored: the memory
 * decides); scratch is allocated and released internally; BLOCKING (returns after the solve has
 * completed on the device).  The fused engine overlaps each snapshot panel with the previous
 * pass on an internal side stream (look-ahead).                                                 */
int fwx_dev_solve(const fwx_slab *full, const fwx_opts *opts);

/* Batch form of `optimum`'s path reconstruction (Algorithms.hs:74-75) on the device: for each
 * pair q, walk next-hops from src[q] until dst[q].  All pointers are DEVICE pointers; `next` is
 * the full n x n matrix.  len_out[q] = path length (0 = no route, FWX_ERR_CYCLE if dst is not
 * reached within n hops).  Optional outputs (NULL to skip): prod_out[q] = product, accumulated in
 * double left to right, of edge_rate[u][v] along the path (edge_rate = the UNSOLVED input matrix,
 * dtype given) -- the solved rate must equal it up to rounding; path_out + q*cap receives up to
 * cap vertices (longer paths are truncated there but still walked and counted).                */
int fwx_dev_follow_paths(int32_t n, const int32_t *next, int32_t count, const int32_t *src,
                         const int32_t *dst, int32_t *len_out, const void *edge_rate,
                         int32_t dtype, double *prod_out, int32_t *path_out, int32_t cap,
                         void *stream);

/* ---- fused engine (FWX_FUSED_BLOCK pivots per pass; bit-identical to the per-k engine) --------
 * fwx_dev_panel_snap: snapshot panel of the pivot rows in `block` (rows [row0,row0+rows), rows <=
 *   FWX_FUSED_BLOCK, at time row0): writes the time-k snapshot of each pivot row to w_rate
 *   (rows x n) and, if the block carries hops, of its hops row to w_hops.  UNLIKE fwx_dev_panel THE
 *   MATRIX IS NOT MODIFIED: the pivot rows are then relaxed like any other row.  trace (optional):
 *   the path trace OF THE SAME ROWS (views of rows [row0, row0+rows) of the trace arrays); its
 *   at_row rows are written.
 * fwx_dev_relax_fused: applies the pivots [piv->k_begin, piv->k_end) (at most FWX_FUSED_BLOCK,
 *   piv->rate = snapshot panel, piv->hops = hops panel, stride n) to EVERY row of the slab in one
 *   pass.  scratch: device scratch of FWX_FUSED_BLOCK * ((slab->rows + 3) & ~3) elements per array
 *   (col_next only if the slab carries next, col_hops only if it carries hops).  trace
 *   (optional): the slab's path trace, rows x n like rate / next (LOCAL rows).  n must be a
 *   multiple of 16 bytes worth of elements.
 * The snapshot panel also feeds fwx_dev_relax (per-k engine), so fwx_dev_panel_snap +
 * fwx_dev_relax over all rows is a valid (slower) combination.
 *
 * flags: FWX_FLAG_NONNEG = the caller has verified (fwx_dev_check_nonneg on EVERY slab of the
 *   matrix, all ranks) that every entry is >= +0.0 and not NaN -- what the reference's parser
 *   guarantees (rates > 0, Parsers.hs:40; "no route" = +0.0) -- and, if next is carried, that no
 *   non-zero rate lacks a path (both bits).  On that domain the strict fold equals max() bit for
 *   bit, and f32 slabs take the kernels that fold two pivots per v_max3_f32 (rates only, or
 *   rates + next + trace + hops through the arg re-scan).  Without the flag nothing is assumed
 *   about the rates.
 *   A slab WITH next must be inside the domain in any case (D1 and D2 on every slab): the fused
 *   kernels take next[i][k] as the head of the concatenated path.  The whole-matrix entry points
 *   check this themselves; here it is the caller's duty.                                        */
#define FWX_FUSED_BLOCK 64
#define FWX_FLAG_NONNEG 1
typedef struct fwx_trace {      /* path trace of a slab (or of a block of pivot rows): rows x n each */
    int32_t *last, *at_col, *at_row;
} fwx_trace;
typedef struct fwx_fused_scratch {
    void *col_rate;             /* pivot-column snapshots of the slab's rows                       */
    int32_t *col_next;          /* ... their next-hops (iff slab->next)                            */
    int32_t *col_hops;          /* ... their hops      (iff slab->hops)                            */
} fwx_fused_scratch;
int fwx_dev_panel_snap(const fwx_slab *block, void *w_rate, int32_t *w_hops, const fwx_trace *trace,
                       void *stream);
int fwx_dev_relax_fused(const fwx_slab *slab, const fwx_pivots *piv, const fwx_fused_scratch *scratch,
                        const fwx_trace *trace, unsigned long long *d_updates, int32_t flags,
                        void *stream);
/* Same, leaving the slab rows [skip_lo, skip_hi) (multiples of 8) to an earlier look-ahead step. */
int fwx_dev_relax_fused_skip(const fwx_slab *slab, const fwx_pivots *piv,
                             const fwx_fused_scratch *scratch, const fwx_trace *trace,
                             unsigned long long *d_updates, int32_t flags, int32_t skip_lo,
                             int32_t skip_hi, void *stream);
/* Domain check of one slab (see "Domain").  *d_flag is a device int32 the caller has set to 3 (or
 * to 1 for a rates-only slab): bit 0 is cleared if any rate of the slab is negative, -0.0 or NaN,
 * bit 1 if the slab carries next and some entry has a non-zero rate with next < 0.  A partitioned
 * solve ANDs the flags of all slabs.                                                            */
int fwx_dev_check_nonneg(const fwx_slab *slab, int32_t *d_flag, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FWX_H */
