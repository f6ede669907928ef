// fwx_kernels.h -- internal launch interface between the C ABI (fwx_api.hip) and the kernels.
#ifndef FWX_KERNELS_H
#define FWX_KERNELS_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// Must equal FWX_UPDATE_SHARDS of include/fwx.h (power of two).
#define FWX_UPDATE_SHARDS_K 256

namespace fwx {

template <typename T> struct RelaxArgs {
    T *rate;                       // slab: rows x n
    int32_t *next;                 // or nullptr
    int32_t *hops;                 // or nullptr
    const T *prow;                 // pivot row k at the start of step k (n elements)
    const int32_t *phops;          // its hops row (iff hops)
    int rows, n, row0, k, flip;
    unsigned long long *updates;   // FWX_UPDATE_SHARDS_K counters or nullptr
};

template <typename T> hipError_t launch_relax(const RelaxArgs<T> &a, hipStream_t s);

template <typename T>
hipError_t launch_snapshot_row(T *dst, const T *src, int32_t *hdst, const int32_t *hsrc, int n,
                               hipStream_t s);

}  // namespace fwx
#endif
