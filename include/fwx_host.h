/*
 * fwx_host.h -- C ABI of the host-side mirror of the reference's operator interface for the hot
 * path: buildMatrix / floydWarshall / optimum and the AppState InSync/OutSync re-run trigger.
 *
 * The reference is Haskell and its toolchain is absent from this image, so the host side above
 * the engine's C ABI (fwx.h) is written in C++ (floydwarshall_amd/csrc/host/) and exported here
 * with plain C types so that tests, the CLI and any FFI can drive it.  Each function names the
 * reference definition it mirrors; names, argument meaning and error TEXT follow the reference.
 *
 *   buildMatrix    /root/reference/src/lib/Algorithms.hs:26-40
 *   floydWarshall  /root/reference/src/lib/Algorithms.hs:19-20      (runAlgo runs on the GPU)
 *   optimum        /root/reference/src/lib/Algorithms.hs:65-78
 *   AppState       /root/reference/src/lib/Types.hs:31-37
 *   updateRates    /root/reference/src/lib/ProcessRequests.hs:89-102
 *   findBestRate   /root/reference/src/lib/ProcessRequests.hs:70-85
 *   serveReq       /root/reference/src/lib/ProcessRequests.hs:31-63
 *   userPrompt     /root/reference/src/app/Main.hs:18-37
 *   parseRates / parseExchPair   /root/reference/src/lib/Parsers.hs:25-72
 *
 * Return values: >= 0 success, < 0 an fwx_status (fwx.h) or FWXH_ERR_*.  Strings are UTF-8,
 * NUL-terminated; output buffers are caller-owned (cap bytes incl. NUL); a result that does not
 * fit returns FWX_ERR_CAPACITY.
 */
#ifndef FWX_HOST_H
#define FWX_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The request was understood but the reference answers it with an error message
 * (Left (AlgoOptimumError ..) / Left (ParseInputError ..)); the text is in the err buffer. */
#define FWXH_ERR_ALGO (-20)
#define FWXH_ERR_PARSE (-21)

#define FWXH_STATE_OUTSYNC 0 /* OutSync ExchRateTimes            (Types.hs:37) */
#define FWXH_STATE_INSYNC 1  /* InSync ExchRateTimes (Matrix ..) (Types.hs:36) */

typedef struct fwxh_session fwxh_session;

/* blankState = OutSync empty (Utils.hs:16-17).  device: HIP ordinal for the solves, -1 = current.
 * Creating a session needs no GPU; only a best-rate query that must solve does. */
int fwxh_session_create(fwxh_session **out, int32_t device);
int fwxh_session_destroy(fwxh_session *s);
/* Put the whole node behind the session's one floydWarshall call (ProcessRequests.hs:82-84): from
 * min_vertices vertices on, the solved matrix is a ROW-PARTITIONED handle over `devices`
 * (fwx_matrix_create_multi in fwx.h: one partition per entry, a device may repeat, panels
 * exchanged on RCCL or by peer copy); smaller matrices stay on devices[0].  n_parts == 0 returns
 * to the single device given at creation.  Answers are bit-identical either way.              */
int fwxh_session_set_devices(fwxh_session *s, int32_t n_parts, const int32_t *devices,
                             int32_t min_vertices);
int32_t fwxh_session_parts(const fwxh_session *s);   /* partitions of the resident matrix (0: none yet) */
int fwxh_session_state(const fwxh_session *s);        /* FWXH_STATE_* as the reference would hold */
int64_t fwxh_session_solves(const fwxh_session *s);   /* floydWarshall runs so far (GPU solves)   */
/* ... of which started from a PATCHED kept input instead of a full buildMatrix + upload: after an
 * update between known vertices only the two changed entries travel (fwx_matrix_patch_input); the
 * solve itself is always the full runAlgo, so answers are bit-identical either way.              */
int64_t fwxh_session_patched_solves(const fwxh_session *s);
/* ... of which did not start at pivot 0: the resident matrix keeps state checkpoints and the panels
 * of every pivot (fwx_matrix_enable_resume in fwx.h), and a solve after a price change between the
 * vertices u, v resumes at the last checkpoint <= min(u, v) -- the changed entries are operands of
 * steps u and v only (Algorithms.hs:58-60), so nothing else can differ before.  Bit-identical to
 * runAlgo 0; resumed_pivots = the pivots those solves skipped, in total.  set_checkpoints: how many
 * the next resident matrix keeps (default 7, 0 = never resume; drops the resident matrix).         */
int64_t fwxh_session_resumed_solves(const fwxh_session *s);
int64_t fwxh_session_resumed_pivots(const fwxh_session *s);
/* Checkpoints the resident matrix really keeps: the requested count cut to what fits into half of the
 * device's free memory; 0 = the handle cannot resume (too small, partitioned, nothing fits, or the
 * allocation failed) and every re-solve is a full solve of the patched input -- never an error.     */
int32_t fwxh_session_checkpoints_kept(const fwxh_session *s);
int fwxh_session_set_checkpoints(fwxh_session *s, int32_t checkpoints);
int32_t fwxh_session_rate_count(const fwxh_session *s);

/* updateRates on parsed fields (ProcessRequests.hs:89-102): stores (src->dst, fwd) and
 * (dst->src, bkd) at `posix_seconds` iff no rate is stored for (src,dst) or the stored time is
 * strictly older (:97-98); any accepted update makes the state OutSync (:101-102).
 * Vertices are (exch, ccy) as given (the parser upper-cases, Parsers.hs:35).
 * Returns 1 if the update was applied, 0 if it was ignored. */
int fwxh_update_rates(fwxh_session *s, int64_t posix_seconds, const char *exch, const char *src_ccy,
                      const char *dst_ccy, double fwd_rate, double bkd_rate);

/* buildMatrix of the session's current rates (Algorithms.hs:26-40), no solve, no GPU.
 * *n_out = vertex count.  Pass NULL buffers to query n only.  vertex_buf receives n lines
 * "EXCH CCY\n" in matrix order (ascending derived Ord, Types.hs:13-17). */
int fwxh_build_matrix(const fwxh_session *s, int32_t *n_out, double *rate, int32_t *next,
                      char *vertex_buf, size_t vertex_cap);

/* findBestRate (ProcessRequests.hs:70-85): sync the matrix if OutSync (floydWarshall on the GPU,
 * then InSync), then optimum src dest.  On success returns the path length L >= 1, *rate_out the
 * best rate, path_buf L+1 lines "(EXCH, CCY)\n": the start vertex then the path (Show Vertex,
 * Types.hs:19-20).  On FWXH_ERR_ALGO err_buf holds optimum's message verbatim and, as in the
 * reference (the State put is rolled back when optimum fails), the reported
 * state is unchanged -- but the solved matrix stays cached on the device, so nothing is re-solved. */
int fwxh_find_best_rate(fwxh_session *s, const char *src_exch, const char *src_ccy,
                        const char *dst_exch, const char *dst_ccy, double *rate_out,
                        char *path_buf, size_t path_cap, char *err_buf, size_t err_cap);

/* Download the session's solved matrix (solving first if OutSync): n*n rate, next, hops. */
int fwxh_solved_matrix(fwxh_session *s, int32_t *n_out, double *rate, int32_t *next, int32_t *hops);

/* optimum (Algorithms.hs:65-78) on caller-provided dense arrays: n_rows rows of n_cols entries
 * (n_cols == 0 with n_rows > 0 models the reference's "matrix with empty rows").  vertices: n_rows
 * pairs, exch[i] / ccy[i].  next == NULL means "path = [dest] wherever rate is reachable" is NOT
 * assumed: next is required when n_cols > 0.  Same outputs as fwxh_find_best_rate. */
int fwxh_optimum_dense(int32_t n_rows, int32_t n_cols, const char *const *exch,
                       const char *const *ccy, const double *rate, const int32_t *next,
                       const char *src_exch, const char *src_ccy, const char *dst_exch,
                       const char *dst_ccy, double *rate_out, char *path_buf, size_t path_cap,
                       char *err_buf, size_t err_cap);

/* Parsers (Parsers.hs:25-72).  On FWXH_ERR_PARSE err_buf holds attoparsec's message as the
 * reference prints it ("Failed reading: ...", "letter: Failed reading: satisfy").
 * Output vertex fields are upper-cased; each *_buf must hold the token + NUL (cap each). */
int fwxh_parse_rates(const char *line, int64_t *posix_seconds, char *exch, char *src_ccy,
                     char *dst_ccy, size_t cap, double *fwd, double *bkd, char *err_buf,
                     size_t err_cap);
int fwxh_parse_exch_pair(const char *line, char *src_exch, char *src_ccy, char *dst_exch,
                         char *dst_ccy, size_t cap, char *err_buf, size_t err_cap);

/* One turn of Main.userPrompt (Main.hs:18-37) = serveReq (ProcessRequests.hs:31-63) on one input
 * line: out receives exactly the lines the reference prints for it, each terminated by '\n'. */
int fwxh_serve_line(fwxh_session *s, const char *line, char *out, size_t out_cap);

/* show :: Double -> String as GHC prints it (1000.0, 9.0e-4, 1.0e7); used by every message. */
int fwxh_show_double(double x, char *out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* FWX_HOST_H */
