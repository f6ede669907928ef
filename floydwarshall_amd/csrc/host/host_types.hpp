// host_types.hpp -- data model of the host mirror (C++), after the reference's Types.hs.
//
//   Vertex         /root/reference/src/lib/Types.hs:13-20   (derived Ord: exch, then ccy; Show)
//   RateEntry      /root/reference/src/lib/Types.hs:24-29   (dense SoA form: rate / next / hops)
//   ExchRateTimes  /root/reference/src/lib/Types.hs:31
//   AppState       /root/reference/src/lib/Types.hs:35-37
#ifndef FWX_HOST_TYPES_HPP
#define FWX_HOST_TYPES_HPP

#include <stdint.h>

#include <map>
#include <string>
#include <utility>
#include <vector>

struct fwx_matrix;

namespace fwxh {

struct Vertex {
    std::string exch, ccy;
    bool operator<(const Vertex &o) const { return exch != o.exch ? exch < o.exch : ccy < o.ccy; }
    bool operator==(const Vertex &o) const { return exch == o.exch && ccy == o.ccy; }
    bool operator!=(const Vertex &o) const { return !(*this == o); }
    // instance Show Vertex (Types.hs:19-20)
    std::string show() const { return "(" + exch + ", " + ccy + ")"; }
};

using VertexPair = std::pair<Vertex, Vertex>;
// ExchRateTimes = Map (Vertex, Vertex) (Double, UTCTime); time as POSIX seconds
using ExchRateTimes = std::map<VertexPair, std::pair<double, int64_t>>;

// Dense form of `Matrix RateEntry` (Types.hs:39): entry (i,j) = (rate, vertices[i], path) with
// next = index of head path (-1 for []), hops = length path.
struct DenseMatrix {
    std::vector<Vertex> vertices;
    std::vector<double> rate;
    std::vector<int32_t> next;
    std::vector<int32_t> hops;
    int32_t n() const { return (int32_t)vertices.size(); }
};

// Result of optimum (Algorithms.hs:65-78): Right RateEntry or Left (AlgoOptimumError text)
struct OptimumResult {
    bool ok = false;
    std::string error;          // verbatim reference text when !ok
    double rate = 0.0;
    Vertex start;
    std::vector<Vertex> path;   // `_path`: vertices after start, dest last
    int status = 0;             // negative fwx_status if the engine itself failed
};

// Algorithms.hs:26-40
DenseMatrix build_matrix(const ExchRateTimes &rates);
void build_matrix_into(const ExchRateTimes &rates, DenseMatrix &out);   // reuses out's storage

// Algorithms.hs:65-78 on host arrays; n_cols == 0 with rows models "matrix with empty rows".
OptimumResult optimum_dense(const std::vector<Vertex> &vertices, int32_t n_cols, const double *rate,
                            const int32_t *next, const Vertex &src, const Vertex &dest);

// show :: Double -> String, show :: UTCTime -> String as GHC prints them
std::string show_double(double x);
std::string show_utctime(int64_t posix_seconds);
std::string show_string(const std::string &s);   // Haskell `show` of a String (quotes, escapes)

// Parsers.hs:25-72
struct ParsedRates {
    int64_t time = 0;
    Vertex src, dest;
    double fwd = 0, bkd = 0;
};
bool parse_rates(const std::string &line, ParsedRates &out, std::string &err);
bool parse_exch_pair(const std::string &line, Vertex &src, Vertex &dest, std::string &err);

// AppState + the request layer (ProcessRequests.hs, Main.hs)
class Session {
public:
    explicit Session(int device);
    ~Session();
    Session(const Session &) = delete;
    Session &operator=(const Session &) = delete;

    // Row-partitioned solves over `devices` from min_vertices vertices on (fwx_matrix_create_multi)
    void set_devices(const std::vector<int32_t> &devices, int32_t min_vertices);
    int parts() const;          // partitions of the resident matrix, 0 if there is none
    int state() const { return in_sync_ ? 1 : 0; }
    int64_t solves() const { return solves_; }
    int64_t patched_solves() const { return patched_solves_; }   // ... of which from a patched kept input
    int64_t resumed_solves() const { return resumed_solves_; }   // ... of which resumed at a checkpoint > 0
    int64_t resumed_pivots() const { return resumed_pivots_; }   // pivots those did NOT have to run, in total
    // checkpoints a resident matrix keeps for resumed re-solves (0 = off); takes effect with the next handle
    void set_checkpoints(int32_t c) { checkpoints_ = c; drop_device(); solved_version_ = ~0ull; }
    int32_t checkpoints_kept() const { return dev_ ? dev_checkpoints_ : 0; }   // of the resident handle
    const ExchRateTimes &rates() const { return rates_; }

    // updateRates (ProcessRequests.hs:89-102) on parsed fields; true if applied
    bool update_rates(int64_t time, const Vertex &src, const Vertex &dest, double fwd, double bkd);
    // findBestRate (ProcessRequests.hs:70-85)
    OptimumResult find_best_rate(const Vertex &src, const Vertex &dest);
    // serveReq + Main.run (ProcessRequests.hs:31-63, Main.hs:26-37): printed lines
    std::vector<std::string> serve_line(const std::string &line);
    // solved matrix on the host (solves if needed)
    int solved_matrix(DenseMatrix &out);

private:
    int ensure_solved();        // floydWarshall on the GPU if the cache is stale
    void drop_device();

    int device_;
    std::vector<int32_t> devices_;   // non-empty: partition the matrix over these from multi_from_ on
    int32_t multi_from_ = 0;
    bool dev_multi_ = false;         // dev_ is a partitioned handle
    int32_t dev_checkpoints_ = 0;    // checkpoints the resident handle really keeps (0: it cannot resume)
    ExchRateTimes rates_;
    bool in_sync_ = false;      // what the reference's AppState would be
    uint64_t version_ = 0;      // bumped by every accepted update that changes buildMatrix's output
                                // (a re-quote of the same two prices with a newer time does not)
    uint64_t solved_version_ = ~0ull;
    std::vector<Vertex> vertices_;   // of the cached solve
    fwx_matrix *dev_ = nullptr;      // solved matrix, resident in HBM
    static constexpr int32_t kFusedFrom = 65;    // fwx AUTO: fused engine from here; no resident hops
    int32_t dev_n_ = 0;              // order of the matrix dev_ was created for
    bool dev_hops_ = true;           // the device matrix carries `hops` (n < kFusedFrom)
    DenseMatrix initial_;            // buildMatrix output, storage kept across re-solves
    int64_t solves_ = 0;
    // Incremental re-marshalling (row f3, the exact part).  While the vertex set is unchanged an
    // accepted update changes exactly the entries (src,dest) and (dest,src) of buildMatrix's output:
    // they are patched into the kept host arrays at once and sent to the input kept on the device
    // (fwx_matrix_patch_input) at the next query, instead of rebuilding and uploading n^2 entries.
    struct Patch { int64_t idx; double rate; int32_t next, hops; };
    static constexpr size_t kMaxPatches = 64;    // beyond this a full marshal is as cheap
    std::map<Vertex, int32_t> vindex_;       // vertex -> row of initial_ (valid while !rebuild_)
    std::vector<Patch> patches_;             // entries changed since the kept device input was current
    bool rebuild_ = true;                    // the next solve marshals from scratch
    int64_t patched_solves_ = 0;
    int64_t resumed_solves_ = 0, resumed_pivots_ = 0;
    int32_t checkpoints_ = 7;                // state checkpoints of a resident matrix (fwx_matrix_enable_resume)
};

}  // namespace fwxh
#endif
