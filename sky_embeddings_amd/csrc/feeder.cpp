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


// ---- PLIO_1 (FITS 4.0 section 10.4.3: the IRAF pixel-list line code, for integer masks).  The tile is a list of big-endian 16-bit
// words: a 7-word header {0, 7, -100, length mod 32768, length / 32768, 0, 0} (or the old 3-word form {.., .., length}), then
// instructions opcode = word >> 12, data = word & 4095 acting on a running "high value" pv (initially 1) and a pixel cursor:
//   0 ZN: data zeros            4 HN: data pixels of pv          5 PN: data - 1 zeros, then one pixel of pv
//   1 SH: pv = (next word << 12) + data (two words)              2 IH / 3 DH: pv += / -= data
//   6 IS / 7 DS: pv += / -= data, then ONE pixel of pv
// Pixels not reached by the list are zero.
int plio_decode_tile(const uint8_t *src, int64_t nwords, int64_t npix, int32_t *out) {
    auto word = [&](int64_t i) { return (int)(int16_t)((src[2 * i] << 8) | src[2 * i + 1]); };       // i: 0-based word index
    for (int64_t i = 0; i < npix; ++i) out[i] = 0;
    if (nwords < 3) return nwords == 0 ? 0 : 1;
    int64_t len, first;
    if (word(2) > 0) {
        len = word(2);
        first = 3;
    } else {
        if (nwords < 7) return 1;
        len = (int64_t)word(4) * 32768 + word(3);
        first = word(1);
    }
    if (len < 0 || len > nwords || first < 0) return 1;
    int64_t op = 0, x = 0;      // next output pixel; list cursor (the whole line is decoded: op == x)
    int64_t pv = 1;
    for (int64_t ip = first; ip < len; ++ip) {
        const int w = word(ip), opcode = (w >> 12) & 7, data = w & 4095;
        switch (opcode) {
            case 0: case 4: case 5: {
                const int64_t stop = x + data < npix ? x + data : npix;
                if (opcode == 4)
                    for (int64_t i = x; i < stop; ++i) out[i] = (int32_t)pv;
                else if (opcode == 5 && data > 0 && x + data <= npix)
                    out[x + data - 1] = (int32_t)pv;
                x += data;
                op = stop;
                break;
            }
            case 1:
                if (ip + 1 >= len) return 1;
                pv = (int64_t)word(ip + 1) * 4096 + data;
                ++ip;
                break;
            case 2: pv += data; break;
            case 3: pv -= data; break;
            case 6: case 7:
                pv += opcode == 6 ? data : -data;
                if (x < npix) out[x] = (int32_t)pv;
                ++x;
                break;
        }
        if (x >= npix) break;
    }
    (void)op;
    return 0;
}

// ---- HCOMPRESS_1 (FITS 4.0 section 10.4.4; White 1992): the tile's H-transform (a 2 x 2 Haar-like pyramid: sum h0 and the
// differences hx, hy, hc of every 2 x 2 block, the sums transformed again ...), its coefficients divided by `scale` (0 / 1 = lossless),
// coded bit plane by bit plane: per quadrant of the coefficient array a quadtree of 4-bit codes (Huffman-coded) or, where that does not
// pay, the plane's 2 x 2 blocks verbatim; then the signs of the nonzero coefficients.  Stream: magic DD 99, nx, ny, scale (big-endian
// int32; ny is the FAST axis of the array), the sum of all pixels (int64), the bit-plane counts of the three quadrant classes.
struct HBits {
    const uint8_t *p, *end;
    uint32_t buffer = 0;      // (only the low `togo` bits are unread; older bits fall off the top)
    int togo = 0;
    bool bad = false;
    void start() { togo = 0; }
    int byte() {
        if (p >= end) {
            bad = true;
            return 0;
        }
        return *p++;
    }
    int bit() {
        if (togo == 0) {
            buffer = (uint32_t)byte();
            togo = 8;
        }
        --togo;
        return (int)((buffer >> togo) & 1u);
    }
    int nbits(int n) {
        if (togo < n) {
            buffer = (buffer << 8) | (uint32_t)byte();
            togo += 8;
        }
        togo -= n;
        return (int)((buffer >> togo) & ((1u << n) - 1));
    }
    int nybble() { return nbits(4); }
    // the fixed code of the 16 quadtree values: 3 bits for 1, 2, 4, 8; 4 bits for 3, 5, 10, 12, 15; 5 for 6, 7, 9, 11, 13; 6 for 0, 14
    int huffman() {
        int c = nbits(3);
        if (c < 4) return 1 << c;
        c = bit() | (c << 1);
        if (c < 13) {
            switch (c) {
                case 8: return 3;
                case 9: return 5;
                case 10: return 10;
                case 11: return 12;
                case 12: return 15;
            }
        }
        c = bit() | (c << 1);
        if (c < 31) {
            switch (c) {
                case 26: return 6;
                case 27: return 7;
                case 28: return 9;
                case 29: return 11;
                case 30: return 13;
            }
        }
        c = bit() | (c << 1);
        return c == 62 ? 0 : 14;
    }
};

// 4-bit codes a[(nx+1)/2][(ny+1)/2] -> one flag per element b[nx][ny] (row length n), in place: code bit 3 = (0,0), 2 = (0,1), 1 = (1,0), 0 = (1,1)
void qtree_copy(uint8_t *a, int nx, int ny, uint8_t *b, int n) {
    const int nx2 = (nx + 1) / 2, ny2 = (ny + 1) / 2;
    int k = ny2 * (nx2 - 1) + ny2 - 1;
    for (int i = nx2 - 1; i >= 0; --i) {
        int s00 = 2 * (n * i + ny2 - 1);
        for (int j = ny2 - 1; j >= 0; --j) {
            b[s00] = a[k];
            --k;
            s00 -= 2;
        }
    }
    int i;
    for (i = 0; i < nx - 1; i += 2) {
        int s00 = n * i, s10 = s00 + n, j;
        for (j = 0; j < ny - 1; j += 2) {
            const uint8_t v = b[s00];
            b[s10 + 1] = v & 1;
            b[s10] = (v >> 1) & 1;
            b[s00 + 1] = (v >> 2) & 1;
            b[s00] = (v >> 3) & 1;
            s00 += 2;
            s10 += 2;
        }
        if (j < ny) {
            const uint8_t v = b[s00];
            b[s10] = (v >> 1) & 1;
            b[s00] = (v >> 3) & 1;
        }
    }
    if (i < nx) {
        int s00 = n * i, j;
        for (j = 0; j < ny - 1; j += 2) {
            const uint8_t v = b[s00];
            b[s00 + 1] = (v >> 2) & 1;
            b[s00] = (v >> 3) & 1;
            s00 += 2;
        }
        if (j < ny) b[s00] = (b[s00] >> 3) & 1;
    }
}
void qtree_expand(HBits &in, uint8_t *a, int nx, int ny, uint8_t *b) {
    qtree_copy(a, nx, ny, b, ny);
    for (int i = nx * ny - 1; i >= 0; --i)
        if (b[i]) b[i] = (uint8_t)in.huffman();
}
// the 4-bit codes of a[(nx+1)/2][(ny+1)/2] set bit `bit` of the coefficients b[nx][ny] (row length n)
void qtree_bitins(const uint8_t *a, int nx, int ny, int64_t *b, int n, int bit) {
    const int64_t plane = (int64_t)1 << bit;
    int k = 0, i;
    for (i = 0; i < nx - 1; i += 2) {
        int s00 = n * i, s10 = s00 + n, j;
        for (j = 0; j < ny - 1; j += 2) {
            const uint8_t v = a[k++];
            if (v & 1) b[s10 + 1] |= plane;
            if (v & 2) b[s10] |= plane;
            if (v & 4) b[s00 + 1] |= plane;
            if (v & 8) b[s00] |= plane;
            s00 += 2;
            s10 += 2;
        }
        if (j < ny) {
            const uint8_t v = a[k++];
            if (v & 2) b[s10] |= plane;
            if (v & 8) b[s00] |= plane;
        }
    }
    if (i < nx) {
        int s00 = n * i, j;
        for (j = 0; j < ny - 1; j += 2) {
            const uint8_t v = a[k++];
            if (v & 4) b[s00 + 1] |= plane;
            if (v & 8) b[s00] |= plane;
            s00 += 2;
        }
        if (j < ny) {
            if (a[k++] & 8) b[s00] |= plane;
        }
    }
}
int qtree_decode(HBits &in, int64_t *a, int n, int nqx, int nqy, int nbitplanes, std::vector<uint8_t> &scratch) {
    const int nqmax = nqx > nqy ? nqx : nqy;
    int log2n = 0;
    while ((1 << log2n) < nqmax) ++log2n;
    const int nqx2 = (nqx + 1) / 2, nqy2 = (nqy + 1) / 2;
    // (the expansions run in place on an [nqx][nqy] grid)
    scratch.assign((size_t)(nqx > 0 ? nqx : 1) * (nqy > 0 ? nqy : 1) + 4, 0);
    for (int bit = nbitplanes - 1; bit >= 0; --bit) {
        const int b = in.nybble();
        if (in.bad) return 1;
        if (b == 0) {
            for (int i = 0; i < nqx2 * nqy2; ++i) scratch[i] = (uint8_t)in.nybble();
        } else if (b != 0xf) {
            return 1;
        } else {
            scratch[0] = (uint8_t)in.huffman();
            int nx = 1, ny = 1, nfx = nqx, nfy = nqy, c = 1 << log2n;
            for (int k = 1; k < log2n; ++k) {
                c >>= 1;
                nx <<= 1;
                ny <<= 1;
                if (nfx <= c) nx -= 1; else nfx -= c;
                if (nfy <= c) ny -= 1; else nfy -= c;
                qtree_expand(in, scratch.data(), nx, ny, scratch.data());
            }
        }
        if (in.bad) return 1;
        qtree_bitins(scratch.data(), nqx, nqy, a, n, bit);
    }
    return 0;
}
// a[0 .. n) with stride n2: first half -> even positions, second half -> odd positions
void unshuffle(int64_t *a, int n, int n2, int64_t *tmp) {
    const int nhalf = (n + 1) >> 1;
    for (int i = nhalf; i < n; ++i) tmp[i - nhalf] = a[(int64_t)n2 * i];
    for (int i = nhalf - 1; i >= 0; --i) a[(int64_t)n2 * 2 * i] = a[(int64_t)n2 * i];
    for (int i = 1, t = 0; i < n; i += 2, ++t) a[(int64_t)n2 * i] = tmp[t];
}
// The optional smoothing of a lossy image on decompression (ZNAME 'SMOOTH' = 1): between the levels of the inverse transform the
// differences of a 2 x 2 block are moved towards the slopes and the curvature that the sums of the NEIGHBOURING blocks imply -- by at
// most scale / 2 (what the division by `scale` may have rounded away) and never past a monotone interpolation.
void hsmooth(int64_t *a, int nxtop, int nytop, int ny, int scale) {
    const int64_t smax = scale >> 1;
    if (smax <= 0) return;
    auto mn = [](int64_t x, int64_t y) { return x < y ? x : y; };
    auto mx = [](int64_t x, int64_t y) { return x > y ? x : y; };
    const int64_t ny2 = (int64_t)ny << 1;
    for (int i = 2; i < nxtop - 2; i += 2) {              // the x difference hx
        int64_t s00 = (int64_t)ny * i, s10 = s00 + ny;
        for (int j = 0; j < nytop; j += 2) {
            const int64_t hm = a[s00 - ny2], h0 = a[s00], hp = a[s00 + ny2];
            int64_t diff = hp - hm;
            const int64_t dmax = mx(mn(hp - h0, h0 - hm), 0) * 4, dmin = mn(mx(hp - h0, h0 - hm), 0) * 4;
            if (dmin < dmax) {
                diff = mx(mn(diff, dmax), dmin);
                int64_t s_ = diff - a[s10] * 8;
                s_ = s_ >= 0 ? (s_ >> 3) : ((s_ + 7) >> 3);
                s_ = mx(mn(s_, smax), -smax);
                a[s10] += s_;
            }
            s00 += 2;
            s10 += 2;
        }
    }
    for (int i = 0; i < nxtop; i += 2) {                  // the y difference hy
        int64_t s00 = (int64_t)ny * i + 2;
        for (int j = 2; j < nytop - 2; j += 2) {
            const int64_t hm = a[s00 - 2], h0 = a[s00], hp = a[s00 + 2];
            int64_t diff = hp - hm;
            const int64_t dmax = mx(mn(hp - h0, h0 - hm), 0) * 4, dmin = mn(mx(hp - h0, h0 - hm), 0) * 4;
            if (dmin < dmax) {
                diff = mx(mn(diff, dmax), dmin);
                int64_t s_ = diff - a[s00 + 1] * 8;
                s_ = s_ >= 0 ? (s_ >> 3) : ((s_ + 7) >> 3);
                s_ = mx(mn(s_, smax), -smax);
                a[s00 + 1] += s_;
            }
            s00 += 2;
        }
    }
    for (int i = 2; i < nxtop - 2; i += 2) {              // the curvature difference hc
        int64_t s00 = (int64_t)ny * i + 2, s10 = s00 + ny;
        for (int j = 2; j < nytop - 2; j += 2) {
            const int64_t hmm = a[s00 - ny2 - 2], hpm = a[s00 + ny2 - 2], hmp = a[s00 - ny2 + 2], hpp = a[s00 + ny2 + 2], h0 = a[s00];
            int64_t diff = hpp + hmm - hmp - hpm;
            const int64_t hx2 = a[s10] * 2, hy2 = a[s00 + 1] * 2;
            int64_t m1 = mn(mx(hpp - h0, 0) - hx2 - hy2, mx(h0 - hpm, 0) + hx2 - hy2);
            int64_t m2 = mn(mx(h0 - hmp, 0) - hx2 + hy2, mx(hmm - h0, 0) + hx2 + hy2);
            const int64_t dmax = mn(m1, m2) * 16;
            m1 = mx(mn(hpp - h0, 0) - hx2 - hy2, mn(h0 - hpm, 0) + hx2 - hy2);
            m2 = mx(mn(h0 - hmp, 0) - hx2 + hy2, mn(hmm - h0, 0) + hx2 + hy2);
            const int64_t dmin = mx(m1, m2) * 16;
            if (dmin < dmax) {
                diff = mx(mn(diff, dmax), dmin);
                int64_t s_ = diff - a[s10 + 1] * 64;
                s_ = s_ >= 0 ? (s_ >> 6) : ((s_ + 63) >> 6);
                s_ = mx(mn(s_, smax), -smax);
                a[s10 + 1] += s_;
            }
            s00 += 2;
            s10 += 2;
        }
    }
}
void hinv(int64_t *a, int nx, int ny, int smooth, int scale) {
    const int nmax = nx > ny ? nx : ny;
    int log2n = 0;
    while ((1 << log2n) < nmax) ++log2n;
    if (log2n == 0) return;                               // a single pixel: the sum is the pixel
    std::vector<int64_t> tmp((nmax + 1) / 2 + 1);
    int shift = 1;
    int64_t bit0 = (int64_t)1 << (log2n - 1), bit1 = bit0 << 1, bit2 = bit0 << 2;
    int64_t mask0 = -bit0, mask1 = mask0 * 2, mask2 = mask0 * 4;
    int64_t prnd0 = bit0 >> 1, prnd1 = bit1 >> 1, prnd2 = bit2 >> 1;
    int64_t nrnd0 = prnd0 - 1, nrnd1 = prnd1 - 1, nrnd2 = prnd2 - 1;
    a[0] = (a[0] + (a[0] >= 0 ? prnd2 : nrnd2)) & mask2;
    int nxtop = 1, nytop = 1, nxf = nx, nyf = ny, c = 1 << log2n;
    for (int k = log2n - 1; k >= 0; --k) {
        c >>= 1;
        nxtop <<= 1;
        nytop <<= 1;
        if (nxf <= c) nxtop -= 1; else nxf -= c;
        if (nyf <= c) nytop -= 1; else nyf -= c;
        if (k == 0) {
            nrnd0 = 0;
            shift = 2;
        }
        for (int i = 0; i < nxtop; ++i) unshuffle(a + (int64_t)ny * i, nytop, 1, tmp.data());
        for (int j = 0; j < nytop; ++j) unshuffle(a + j, nxtop, ny, tmp.data());
        if (smooth) hsmooth(a, nxtop, nytop, ny, scale);
        const int oddx = nxtop % 2, oddy = nytop % 2;
        int i;
        for (i = 0; i < nxtop - oddx; i += 2) {
            int64_t s00 = (int64_t)ny * i, s10 = s00 + ny;
            int j;
            for (j = 0; j < nytop - oddy; j += 2) {
                int64_t h0 = a[s00], hx = a[s10], hy = a[s00 + 1], hc = a[s10 + 1];
                hx = (hx + (hx >= 0 ? prnd1 : nrnd1)) & mask1;
                hy = (hy + (hy >= 0 ? prnd1 : nrnd1)) & mask1;
                hc = (hc + (hc >= 0 ? prnd0 : nrnd0)) & mask0;
                const int64_t lowbit0 = hc & bit0;
                hx = hx >= 0 ? hx - lowbit0 : hx + lowbit0;
                hy = hy >= 0 ? hy - lowbit0 : hy + lowbit0;
                const int64_t lowbit1 = (hc ^ hx ^ hy) & bit1;
                h0 = h0 >= 0 ? h0 + lowbit0 - lowbit1 : h0 + (lowbit0 == 0 ? lowbit1 : lowbit0 - lowbit1);
                a[s10 + 1] = (h0 + hx + hy + hc) >> shift;
                a[s10] = (h0 + hx - hy - hc) >> shift;
                a[s00 + 1] = (h0 - hx + hy - hc) >> shift;
                a[s00] = (h0 - hx - hy + hc) >> shift;
                s00 += 2;
                s10 += 2;
            }
            if (oddy) {
                int64_t h0 = a[s00], hx = a[s10];
                hx = (hx >= 0 ? hx + prnd1 : hx + nrnd1) & mask1;
                const int64_t lowbit1 = hx & bit1;
                h0 = h0 >= 0 ? h0 - lowbit1 : h0 + lowbit1;
                a[s10] = (h0 + hx) >> shift;
                a[s00] = (h0 - hx) >> shift;
            }
        }
        if (oddx) {
            int64_t s00 = (int64_t)ny * i;
            int j;
            for (j = 0; j < nytop - oddy; j += 2) {
                int64_t h0 = a[s00], hy = a[s00 + 1];
                hy = (hy >= 0 ? hy + prnd1 : hy + nrnd1) & mask1;
                const int64_t lowbit1 = hy & bit1;
                h0 = h0 >= 0 ? h0 - lowbit1 : h0 + lowbit1;
                a[s00 + 1] = (h0 + hy) >> shift;
                a[s00] = (h0 - hy) >> shift;
                s00 += 2;
            }
            if (oddy) a[s00] = a[s00] >> shift;
        }
        bit2 = bit1;
        bit1 = bit0;
        bit0 >>= 1;
        mask1 = mask0;
        mask0 >>= 1;
        prnd1 = prnd0;
        prnd0 >>= 1;
        nrnd1 = nrnd0;
        nrnd0 = prnd0 - 1;
    }
}
int hcompress_decode_tile(const uint8_t *src, int64_t nbytes, int64_t npix, int32_t *out, int smooth) {
    if (nbytes < 2 + 12 + 8 + 3 || src[0] != 0xDD || src[1] != 0x99) return 1;
    auto be32 = [&](const uint8_t *p) { return (int32_t)(((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]); };
    const int nx = be32(src + 2), ny = be32(src + 6), scale = be32(src + 10);
    if (nx <= 0 || ny <= 0 || (int64_t)nx * ny != npix) return 1;
    int64_t sumall = 0;
    for (int b = 0; b < 8; ++b) sumall = (int64_t)(((uint64_t)sumall << 8) | src[14 + b]);
    const int nbp[3] = {src[22], src[23], src[24]};
    // (bounds far beyond what 32-bit pixels produce -- coefficients below 2^36, tile sums below 2^56 -- so that a damaged stream
    // cannot overflow the 64-bit arithmetic below)
    const int64_t sum_cap = ((int64_t)1 << 56) / (scale > 1 ? scale : 1);      // (x 64 in the smoothing stays below 2^63)
    if (nbp[0] > 36 || nbp[1] > 36 || nbp[2] > 36 || scale < 0 || scale > (1 << 20) || sumall > sum_cap || sumall < -sum_cap) return 1;
    std::vector<int64_t> a((size_t)npix, 0);
    std::vector<uint8_t> scratch;
    HBits in{src + 25, src + nbytes};
    const int nx2 = (nx + 1) / 2, ny2 = (ny + 1) / 2;
    in.start();
    if (qtree_decode(in, a.data(), ny, nx2, ny2, nbp[0], scratch)) return 1;
    if (qtree_decode(in, a.data() + ny2, ny, nx2, ny / 2, nbp[1], scratch)) return 1;
    if (qtree_decode(in, a.data() + (int64_t)ny * nx2, ny, nx / 2, ny2, nbp[1], scratch)) return 1;
    if (qtree_decode(in, a.data() + (int64_t)ny * nx2 + ny2, ny, nx / 2, ny / 2, nbp[2], scratch)) return 1;
    if (in.nybble() != 0 || in.bad) return 1;       // the end-of-planes symbol
    in.start();
    for (int64_t i = 0; i < npix; ++i)
        if (a[i] != 0 && in.bit()) a[i] = -a[i];
    if (in.bad) return 1;
    a[0] = sumall;
    if (scale > 1)
        for (int64_t i = 0; i < npix; ++i) a[i] *= scale;
    hinv(a.data(), nx, ny, smooth, scale);
    for (int64_t i = 0; i < npix; ++i) out[i] = (int32_t)a[i];
    return 0;
}
}  // namespace

// ntiles compressed tile streams -> integer pixels in the host's byte order.  codec 1 = RICE_1 (pixels of `bytepix` = 1, 2 or 4 bytes,
// blocks of `blocksize` differences), 2 = PLIO_1, 3 = HCOMPRESS_1 (both: int32 pixels, bytepix must be 4; the stream of an
// HCOMPRESS tile carries its own dimensions, whose product must be npix[t]; for HCOMPRESS `blocksize` is the SMOOTH flag, 0 or 1).  Tile t: bytes [off[t], off[t] + len[t]) of `base`
// hold npix[t] pixels, written to dst + dst_off[t] * bytepix.  HOST function (threads over tiles).
extern "C" int skyemb_fits_decode_tiles_host(int codec, const void *base, int64_t base_bytes, const int64_t *off, const int64_t *len,
                                             const int64_t *npix, const int64_t *dst_off, int64_t ntiles, int bytepix, int blocksize, void *dst,
                                             int64_t dst_pixels, int nthreads) {
    if (!base || !off || !len || !npix || !dst_off || !dst || ntiles < 0 || codec < 1 || codec > 3 || (codec == 1 && blocksize < 1) || (codec == 3 && (blocksize < 0 || blocksize > 1)) ||
        (bytepix != 1 && bytepix != 2 && bytepix != 4) || (codec != 1 && bytepix != 4)) {
        skyemb_set_error("skyemb_fits_decode_tiles_host: bad arguments (codec %d, bytepix %d, blocksize %d)", codec, bytepix, blocksize);
        return 1;
    }
    for (int64_t t = 0; t < ntiles; ++t)
        if (off[t] < 0 || len[t] < 0 || off[t] + len[t] > base_bytes || npix[t] < 0 || dst_off[t] < 0 || dst_off[t] + npix[t] > dst_pixels) {
            skyemb_set_error("skyemb_fits_decode_tiles_host: tile %lld lies outside its buffers", (long long)t);
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
            if (codec == 2) rc = (len[t] & 1) ? 1 : plio_decode_tile(s, len[t] / 2, npix[t], (int32_t *)dst + dst_off[t]);
            else if (codec == 3) rc = hcompress_decode_tile(s, len[t], npix[t], (int32_t *)dst + dst_off[t], blocksize);
            else if (bytepix == 4) rc = rice_decode_tile<uint32_t>(s, len[t], blocksize, npix[t], (uint32_t *)dst + dst_off[t]);
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
            skyemb_set_error("skyemb_fits_decode_tiles_host: tile %lld: the %s stream is damaged or ends before its %lld pixels", (long long)bad[w],
                             codec == 1 ? "Rice" : codec == 2 ? "PLIO" : "HCOMPRESS", (long long)npix[bad[w]]);
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
