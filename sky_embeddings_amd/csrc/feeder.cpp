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

// Rice-compressed tiles of FITS tile-compressed images (FITS standard 4.0, section 10.4.1; the codec fpack and astropy's
// CompImageHDU use by default -- utils/dataloaders.py:418 reads such survey tiles through astropy).  Restated from the standard's
// description of the bit stream: the first pixel verbatim (big-endian, `bytepix` bytes), then blocks of `blocksize` pixel
// DIFFERENCES, each block opened by an fsbits-wide code fs + 1: 0 = every difference of the block is zero; fsmax + 1 = the
// differences stand verbatim in bbits bits each; otherwise every difference is a unary count of zero bits (its top part), a one
// bit, and fs low bits.  Differences are sign-folded (even = +v/2, odd = -(v+1)/2) and wrap in the pixel width.
namespace {
struct RiceBits {
    const uint8_t *p, *end;
    uint64_t acc = 0;       // the low `n` bits are the unread bits
    int n = 0;
    bool refill() {
        while (n <= 56 && p < end) {
            acc = (acc << 8) | *p++;
            n += 8;
        }
        return n > 0;
    }
    // v = the next `k` bits (k <= 32); false at the end of the stream
    bool take(int k, uint32_t &v) {
        if (k == 0) {
            v = 0;
            return true;
        }
        if (n < k) {
            refill();
            if (n < k) return false;
        }
        v = (uint32_t)((acc >> (n - k)) & ((k == 32) ? 0xffffffffull : ((1ull << k) - 1)));
        n -= k;
        return true;
    }
    // the number of zero bits before the next one bit (which is consumed too)
    bool unary(uint32_t &zeros) {
        zeros = 0;
        for (;;) {
            if (n == 0 && !refill()) return false;
            const uint64_t window = n == 64 ? acc : (acc & ((1ull << n) - 1));
            if (window == 0) {
                zeros += (uint32_t)n;
                n = 0;
                continue;
            }
            const int top = 63 - __builtin_clzll(window);      // position of the first one bit among the n unread bits
            zeros += (uint32_t)(n - 1 - top);
            n = top;
            return true;
        }
    }
};

template <typename U>
int rice_decode_tile(const uint8_t *src, int64_t nbytes, int blocksize, int64_t npix, U *out) {
    constexpr int bbits = 8 * (int)sizeof(U);
    constexpr int fsbits = sizeof(U) == 4 ? 5 : sizeof(U) == 2 ? 4 : 3;
    constexpr int fsmax = sizeof(U) == 4 ? 25 : sizeof(U) == 2 ? 14 : 6;
    if (npix == 0) return 0;
    if (nbytes < (int64_t)sizeof(U)) return 1;
    U last = 0;
    for (size_t b = 0; b < sizeof(U); ++b) last = (U)((last << 8) | src[b]);
    RiceBits bits{src + sizeof(U), src + nbytes};
    for (int64_t i = 0; i < npix;) {
        uint32_t code;
        if (!bits.take(fsbits, code)) return 1;
        const int fs = (int)code - 1;
        const int64_t stop = i + blocksize < npix ? i + blocksize : npix;
        if (fs < 0) {
            for (; i < stop; ++i) out[i] = last;
        } else if (fs == fsmax) {
            for (; i < stop; ++i) {
                uint32_t d;
                if (!bits.take(bbits, d)) return 1;
                const U diff = (U)d;
                last = (U)(((diff & 1) ? (U)~(diff >> 1) : (U)(diff >> 1)) + last);
                out[i] = last;
            }
        } else {
            for (; i < stop; ++i) {
                uint32_t zeros, low;
                if (!bits.unary(zeros)) return 1;
                if (!bits.take(fs, low)) return 1;
                const U diff = (U)(((uint64_t)zeros << fs) | low);
                last = (U)(((diff & 1) ? (U)~(diff >> 1) : (U)(diff >> 1)) + last);
                out[i] = last;
            }
        }
    }
    return 0;
}
}  // namespace

// ntiles Rice streams -> pixels in the host's byte order.  Tile t: bytes [off[t], off[t] + len[t]) of `base` hold npix[t] pixels of
// `bytepix` (1, 2, 4) bytes, written to dst + dst_off[t] * bytepix.  HOST function (threads over tiles).
extern "C" int skyemb_fits_rice_tiles_host(const void *base, int64_t base_bytes, const int64_t *off, const int64_t *len, const int64_t *npix,
                                           const int64_t *dst_off, int64_t ntiles, int bytepix, int blocksize, void *dst, int64_t dst_pixels,
                                           int nthreads) {
    if (!base || !off || !len || !npix || !dst_off || !dst || ntiles < 0 || blocksize < 1 || (bytepix != 1 && bytepix != 2 && bytepix != 4)) {
        skyemb_set_error("skyemb_fits_rice_tiles_host: bad arguments (bytepix %d, blocksize %d)", bytepix, blocksize);
        return 1;
    }
    for (int64_t t = 0; t < ntiles; ++t)
        if (off[t] < 0 || len[t] < 0 || off[t] + len[t] > base_bytes || npix[t] < 0 || dst_off[t] < 0 || dst_off[t] + npix[t] > dst_pixels) {
            skyemb_set_error("skyemb_fits_rice_tiles_host: tile %lld lies outside its buffers", (long long)t);
            return 1;
        }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if ((int64_t)nthreads > ntiles) nthreads = ntiles > 0 ? (int)ntiles : 1;
    std::vector<int64_t> bad(nthreads, -1);
    auto work = [&](int w) {
        const int64_t lo = ntiles * w / nthreads, hi = ntiles * (w + 1) / nthreads;
        for (int64_t t = lo; t < hi; ++t) {
            const uint8_t *s = (const uint8_t *)base + off[t];
            int rc;
            if (bytepix == 4) rc = rice_decode_tile<uint32_t>(s, len[t], blocksize, npix[t], (uint32_t *)dst + dst_off[t]);
            else if (bytepix == 2) rc = rice_decode_tile<uint16_t>(s, len[t], blocksize, npix[t], (uint16_t *)dst + dst_off[t]);
            else rc = rice_decode_tile<uint8_t>(s, len[t], blocksize, npix[t], (uint8_t *)dst + dst_off[t]);
            if (rc != 0 && bad[w] < 0) bad[w] = t;
        }
    };
    if (nthreads == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        th.reserve(nthreads - 1);
        for (int w = 1; w < nthreads; ++w) th.emplace_back(work, w);
        work(0);
        for (auto &x : th) x.join();
    }
    for (int w = 0; w < nthreads; ++w)
        if (bad[w] >= 0) {
            skyemb_set_error("skyemb_fits_rice_tiles_host: tile %lld: the Rice stream ends before its %lld pixels", (long long)bad[w],
                             (long long)npix[bad[w]]);
            return 1;
        }
    return 0;
}

// Quantised floating-point tiles -> float32 pixels placed in the image (FITS 4.0 section 10.2: value = q * ZSCALE + ZZERO, or with
// subtractive dithering (q - r + 0.5) * ZSCALE + ZZERO where r walks the convention's table of 10 000 random numbers: tile row n
// (1-based) of the table starts at index int(rand[(n + ZDITHER0 - 2) % 10000] * 500) and, at the table's end, restarts from the next
// seed entry).  `rand`: the table (the caller builds it: fits_lite.dither_sequence); method 0 = no dither, 1 / 2 = SUBTRACTIVE_DITHER_1 / 2
// (2: q == -2147483646 is exactly 0).  has_blank[t] != 0: q == blank[t] -> NaN.  Tile t: q + q_off[t], h[t] x w[t] pixels, row-major, to
// out[(y0[t] + y) * W + x0[t] + x].  big_endian_out: the floats are stored byte-swapped (the layout of an uncompressed FITS image, which
// skyemb_tile_cutouts decodes on the device).  HOST function, threads over tiles.
extern "C" int skyemb_fits_dequantise_tiles_host(const int32_t *q, const int64_t *q_off, const int64_t *y0, const int64_t *x0, const int64_t *h,
                                                 const int64_t *w, const int64_t *table_row, int64_t ntiles, const double *zscale,
                                                 const double *zzero, const int32_t *blank, const uint8_t *has_blank, const float *rand,
                                                 int method, int zdither0, float *out, int64_t H, int64_t W, int big_endian_out, int nthreads) {
    constexpr int N_RANDOM = 10000;
    if (!q || !q_off || !y0 || !x0 || !h || !w || !table_row || !zscale || !zzero || !out || ntiles < 0 || method < 0 || method > 2 ||
        (method != 0 && !rand) || H < 0 || W < 0) {
        skyemb_set_error("skyemb_fits_dequantise_tiles_host: bad arguments");
        return 1;
    }
    for (int64_t t = 0; t < ntiles; ++t)
        if (y0[t] < 0 || x0[t] < 0 || h[t] < 0 || w[t] < 0 || y0[t] + h[t] > H || x0[t] + w[t] > W || q_off[t] < 0 || table_row[t] < 0) {
            skyemb_set_error("skyemb_fits_dequantise_tiles_host: tile %lld lies outside the image", (long long)t);
            return 1;
        }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 64) nthreads = 64;
    if ((int64_t)nthreads > ntiles) nthreads = ntiles > 0 ? (int)ntiles : 1;
    auto work = [&](int wk) {
        const int64_t lo = ntiles * wk / nthreads, hi = ntiles * (wk + 1) / nthreads;
        for (int64_t t = lo; t < hi; ++t) {
            const int32_t *src = q + q_off[t];
            const double scale = zscale[t], zero = zzero[t];
            const bool nulls = has_blank && has_blank[t];
            const int32_t null_q = blank ? blank[t] : 0;
            int iseed = 0, next = 0;
            if (method != 0) {
                iseed = (int)((table_row[t] + zdither0 - 1) % N_RANDOM);
                next = (int)(rand[iseed] * 500);
            }
            for (int64_t y = 0; y < h[t]; ++y) {
                float *dst = out + (y0[t] + y) * W + x0[t];
                for (int64_t x = 0; x < w[t]; ++x) {
                    const int32_t v = *src++;
                    float f;
                    if (nulls && v == null_q) f = __builtin_nanf("");
                    else if (method == 2 && v == -2147483646) f = 0.0f;
                    else if (method != 0) f = (float)(((double)v - rand[next] + 0.5) * scale + zero);
                    else f = (float)((double)v * scale + zero);
                    if (method != 0 && ++next == N_RANDOM) {
                        if (++iseed == N_RANDOM) iseed = 0;
                        next = (int)(rand[iseed] * 500);
                    }
                    if (big_endian_out) {
                        uint32_t u;
                        memcpy(&u, &f, 4);
                        u = __builtin_bswap32(u);
                        memcpy(dst + x, &u, 4);
                    } else {
                        dst[x] = f;
                    }
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
    for (int wk = 1; wk < nthreads; ++wk) th.emplace_back(work, wk);
    work(0);
    for (auto &x : th) x.join();
    return 0;
}
