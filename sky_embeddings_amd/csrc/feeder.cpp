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

// Chunked HDF5 datasets (what the reference's ETL writes: create_dataset(..., maxshape=(None, ...)) + resize, i.e. h5py's
// auto-chunked layout with chunks of a few rows x a slice of the bands / pixels: data_processing/2_create_h5_files.py:70-81)
// are scattered ONCE into a contiguous row-major image (the feeder's mmap fast path then serves them like a contiguous
// dataset).  file_base: the memory-mapped file; chunk i starts at byte chunk_addr[i] and covers element offsets
// chunk_off[i*rank .. +rank) with extent chunk_dims (edge chunks are stored whole: the part past dset_dims is dropped).
extern "C" int skyemb_h5_unchunk_host(const void *file_base, int64_t file_bytes, const int64_t *chunk_addr, const int64_t *chunk_off,
                                      int64_t nchunks, int rank, const int64_t *chunk_dims, const int64_t *dset_dims, int elem_size,
                                      void *dst, int nthreads) {
    if (!file_base || !chunk_addr || !chunk_off || !chunk_dims || !dset_dims || !dst || rank < 1 || rank > 8 || elem_size < 1 ||
        nchunks < 0) {
        skyemb_set_error("skyemb_h5_unchunk_host: bad arguments");
        return 1;
    }
    int64_t chunk_elems = 1, dstride[8], cstride[8];
    for (int d = rank - 1; d >= 0; --d) {
        cstride[d] = chunk_elems;
        chunk_elems *= chunk_dims[d];
    }
    int64_t acc = 1;
    for (int d = rank - 1; d >= 0; --d) {
        dstride[d] = acc;
        acc *= dset_dims[d];
    }
    for (int64_t i = 0; i < nchunks; ++i) {
        if (chunk_addr[i] < 0 || chunk_addr[i] + chunk_elems * elem_size > file_bytes) {
            skyemb_set_error("skyemb_h5_unchunk_host: chunk %lld lies outside the file", (long long)i);
            return 1;
        }
        for (int d = 0; d < rank; ++d)
            if (chunk_off[i * rank + d] < 0 || chunk_off[i * rank + d] >= dset_dims[d]) {
                skyemb_set_error("skyemb_h5_unchunk_host: chunk %lld starts outside the dataset", (long long)i);
                return 1;
            }
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if ((int64_t)nthreads > nchunks) nthreads = nchunks > 0 ? (int)nchunks : 1;
    auto work = [&](int t) {
        const int64_t lo = nchunks * t / nthreads, hi = nchunks * (t + 1) / nthreads;
        for (int64_t i = lo; i < hi; ++i) {
            const char *src = (const char *)file_base + chunk_addr[i];
            const int64_t *off = chunk_off + i * rank;
            int64_t ext[8];                       // valid extent of the chunk inside the dataset
            for (int d = 0; d < rank; ++d) {
                const int64_t left = dset_dims[d] - off[d];
                ext[d] = left < chunk_dims[d] ? left : chunk_dims[d];
            }
            const int64_t run = ext[rank - 1] * elem_size;
            int64_t rows = 1;
            for (int d = 0; d + 1 < rank; ++d) rows *= ext[d];
            int64_t c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int64_t r = 0; r < rows; ++r) {
                int64_t so = 0, dofs = off[rank - 1];
                for (int d = 0; d + 1 < rank; ++d) {
                    so += c[d] * cstride[d];
                    dofs += (off[d] + c[d]) * dstride[d];
                }
                memcpy((char *)dst + dofs * elem_size, src + so * elem_size, (size_t)run);
                for (int d = rank - 2; d >= 0; --d) {
                    if (++c[d] < ext[d]) break;
                    c[d] = 0;
                }
            }
        }
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
