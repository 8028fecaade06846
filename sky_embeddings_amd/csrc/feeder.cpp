// Host side of the input feeder (reference: utils/dataloaders.py:285-328 H5Dataset.__getitem__, one python call and one
// file open per cutout).  Here a whole minibatch of rows is gathered from the memory-mapped contiguous HDF5 dataset into a
// pinned staging buffer by a few native threads (plain memcpy, no Python per item); the caller then issues ONE async
// host-to-device copy on its copy stream.  ctypes releases the GIL for the duration of the call.
#include <stdint.h>
#include <string.h>

#include <thread>
#include <vector>

#include "../../include/skyemb.h"

void skyemb_set_error(const char *fmt, ...);

extern "C" int skyemb_gather_rows_host(const void *src, int64_t row_bytes, const int64_t *idx, int64_t n, int64_t src_rows,
                                       void *dst, int nthreads) {
    if (!src || !dst || !idx || row_bytes <= 0 || n < 0 || src_rows <= 0) {
        skyemb_set_error("skyemb_gather_rows_host: bad arguments");
        return 1;
    }
    for (int64_t i = 0; i < n; ++i)
        if (idx[i] < 0 || idx[i] >= src_rows) {
            skyemb_set_error("skyemb_gather_rows_host: index %lld out of range [0, %lld)", (long long)idx[i], (long long)src_rows);
            return 1;
        }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if ((int64_t)nthreads > n) nthreads = n > 0 ? (int)n : 1;
    auto work = [=](int t) {
        const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
        for (int64_t i = lo; i < hi; ++i)
            memcpy((char *)dst + i * row_bytes, (const char *)src + idx[i] * row_bytes, (size_t)row_bytes);
    };
    if (nthreads == 1) {
        work(0);
        return 0;
    }
    std::vector<std::thread> th;
    th.reserve(nthreads - 1);
    for (int t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    return 0;
}
