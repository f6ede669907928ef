// fwx_replay.h -- the replay kernel of resumable solves, shared by the single-device handle (fwx_api.hip)
// and the partitioned one (fwx_multi.hip).  Not installed, not part of the ABI.
#ifndef FWX_REPLAY_H
#define FWX_REPLAY_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace fwxi {

// Replay of patched input entries through the pivots [0, c) they were not part of (Resume in
// fwx_internal.h).  One wave per entry (i, j): lane l of chunk q forms the candidate of pivot
// k = 64 q + l from the stored panels, c[k] = ct[k][i] * w[k][j] -- the very operands step k used --
// and the wave folds the chunk at once: on the reference's domain the strict fold of Algorithms.hs:55
// ends at max(x, max_k c[k]) (a NaN candidate never wins), and its LAST update is the FIRST pivot that
// attains that maximum, which gives next = cnt[k*][i], hops = cht[k*][i] + wh[k*][j], last = k*.
// Checkpoints are multiples of 64, so the value at every checkpoint <= c falls on a chunk boundary
// and is written into that checkpoint; the value at time c goes to the live arrays.
struct ReplayTargets {
    enum { MAX = 20 };
    int count;
    int pivot[MAX];
    void *rate[MAX];
    int32_t *next[MAX], *hops[MAX], *last[MAX];
};

template <typename T>
__global__ __launch_bounds__(64) void replay_entries_kernel(const int64_t *index, int n, int ld, int row0, int c,
                                                            const T *rate0, const int32_t *next0,
                                                            const int32_t *hops0, const T *w, const T *ct,
                                                            const int32_t *cnt, const int32_t *wh,
                                                            const int32_t *cht, ReplayTargets tg)
{
    const int64_t idx = index[blockIdx.x];
    // idx: offset in the arrays of this slab (local row * n + column); row0: global index of its first row
    const int i = (int)(idx / n), j = (int)(idx % n), lane = threadIdx.x;
    T x = rate0[idx];
    int nx = next0 ? next0[idx] : -1, hp = hops0 ? hops0[idx] : 0, last = -1;
    int t = 0;
    for (int k0 = 0; k0 <= c; k0 += 64) {
        while (t < tg.count && tg.pivot[t] == k0) {
            if (lane == 0) {
                ((T *)tg.rate[t])[idx] = x;
                if (tg.next[t]) tg.next[t][idx] = nx;
                if (tg.hops[t]) tg.hops[t][idx] = hp;
                if (tg.last[t]) tg.last[t][idx] = last;
            }
            ++t;
        }
        if (k0 == c || i + row0 == j) continue;          // a diagonal entry is never a target (:54)
        const int k = k0 + lane;
        T v = ct[(size_t)k * ld + i] * w[(size_t)k * n + j];
        int arg = k;
        if (!(v == v)) v = -INFINITY;                    // NaN (inf * 0) never wins a strict compare
        for (int d = 1; d < 64; d <<= 1) {               // max, earliest pivot on ties
            const T ov = __shfl_xor(v, d);
            const int oa = __shfl_xor(arg, d);
            if (ov > v || (ov == v && oa < arg)) { v = ov; arg = oa; }
        }
        if (x < v) {
            x = v;
            last = arg;
            if (cnt) nx = cnt[(size_t)arg * ld + i];
            if (cht) hp = cht[(size_t)arg * ld + i] + wh[(size_t)arg * n + j];
        }
    }
}

}  // namespace fwxi

#endif
