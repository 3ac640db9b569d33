// Mask tail (gfx950): detector_postprocess + paste_masks_in_image + the reference's mask geometry,
// without materialising N x 2160 x 3840 masks.
//
//   det_postprocess : Boxes.scale / clip / nonempty + the integer paste window
//                     (detectron2 detector_postprocess, reached from dcnn/networks/track_rcnn.py:57)
//   paste_masks     : sigmoid -> bilinear grid_sample(align_corners=False, zero pad) -> >= 0.5, written as
//                     bit-packed rows of the *frame* (one 64-bit word = 64 pixels), only inside the window;
//                     integer centroid sums accumulated on the fly
//                     (dcnn/utils/mask_utils.py:27-38 get_mask_centroid, 1-based coordinates)
//   closest_points  : first row-major mask pixel minimising the f32 squared distance to a target point
//                     (dcnn/utils/mask_utils.py:6-23 compute_closest_point); targets = every detection's
//                     centroid in the same image, so the host pick (which needs track ids) happens later.
// f32 arithmetic is compiled with -ffp-contract=off (products and sums rounded like the CPU path).
#include "apse_common.h"

struct PasteParams {
    const float* boxes;      // packed [n][4], resized-image coordinates
    const int* cls;          // packed [n]
    const int* total;        // device count of packed detections
    const float* logits;     // [n][M][M][ldc] mask head output (NHWC), class channel = cls[n]
    int M, ldc;
    float sx, sy;            // output/resized scale factors (f32 of the Python doubles)
    int out_h, out_w;
    int words_per_row;       // ceil(out_w / 64)
    float thresh;
    float* boxes_out;        // [n][4] scaled + clipped boxes
    int* valid;              // [n] nonempty after scaling
    int* rect;               // [n][4] x0, y0, x1, y1 paste window
    uint64_t* bits;          // [n][out_h][words_per_row]
    unsigned long long* sums;   // [n][3] mass, sum(x+1), sum(y+1): zero when the launch starts (pack_detections clears them)
};

#define MT_TARGETS 100            // detections per image (TEST.DETECTIONS_PER_IMAGE <= 100, apse_create)
#define MT_MAXDET 1024            // packed detections per forward (apse_create enforces max_batch * dets_per_image <= this)
#define PASTE_BLOCKS 1024
// Rows of a window per work item.  The band is picked per launch from the windows' total row count so that there are enough
// items to fill the card: a single frame's 8 windows (~4 200 rows) in 16-row bands are 275 items = one 4-wave block on each CU,
// which runs at the latency of its own chain (round 3, rocprofv3, f32 batch 1 / fp16 batch 8, us): paste 16 rows 47 / 125,
// 8 rows 29 / 114, 4 rows 24 / 132 (2 rows 26, 1 row 34 at batch 1); closest points 32 rows 28 / 38, 16 rows 20 / 36, 8 rows 19 / 54.
// Results do not depend on it.
#define PASTE_BAND_MAX 16
#ifndef PASTE_BAND_MIN
#define PASTE_BAND_MIN 4
#endif
#ifndef PASTE_ITEMS_MIN
#define PASTE_ITEMS_MIN 3000
#endif
// largest band in {bmax, bmax/2, .., bmin} that still gives at least `want` items (rows / band is a lower bound of the item count)
__device__ __forceinline__ int mt_pick_band(int total_rows, int bmax, int bmin, int want) {
    int b = bmax;
    while (b > bmin && total_rows / b < want) b >>= 1;
    return b;
}

struct DetPost { float x0, y0, x1, y1; int ok, rx0, ry0, rx1, ry1; };
// Boxes.scale / clip / nonempty + the integer paste window of detection i (detectron2 detector_postprocess)
__device__ __forceinline__ DetPost det_post(const PasteParams& p, int i) {
    DetPost d;
    float x0 = p.boxes[i * 4 + 0] * p.sx, y0 = p.boxes[i * 4 + 1] * p.sy;
    float x1 = p.boxes[i * 4 + 2] * p.sx, y1 = p.boxes[i * 4 + 3] * p.sy;
    const float w = (float)p.out_w, h = (float)p.out_h;
    x0 = fminf(fmaxf(x0, 0.f), w); y0 = fminf(fmaxf(y0, 0.f), h);
    x1 = fminf(fmaxf(x1, 0.f), w); y1 = fminf(fmaxf(y1, 0.f), h);
    d.x0 = x0; d.y0 = y0; d.x1 = x1; d.y1 = y1;
    d.ok = ((x1 - x0) > 0.f) && ((y1 - y0) > 0.f);
    d.rx0 = (int)fmaxf(floorf(x0) - 1.f, 0.f); d.ry0 = (int)fmaxf(floorf(y0) - 1.f, 0.f);
    d.rx1 = (int)fminf(ceilf(x1) + 1.f, w); d.ry1 = (int)fminf(ceilf(y1) + 1.f, h);
    if (!d.ok) { d.rx1 = d.rx0; d.ry1 = d.ry0; }
    return d;
}

// Exclusive prefix of cnt[0..nd) (nd <= MT_MAXDET) in place, cnt[nd] = total; 256 threads, 4 entries each.
__device__ __forceinline__ void mt_scan(int* cnt, int nd, int* wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int v[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = tid * 4 + k; v[k] = i < nd ? cnt[i] : 0; s += v[k]; }
    int incl = s;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    int base = incl - s;
    for (int w = 0; w < wave; ++w) base += wtot[w];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = tid * 4 + k; if (i < nd) cnt[i] = base; base += v[k]; }
    if (tid == 255) cnt[nd] = base;        // the last thread's running sum covers every entry (entries past nd count 0)
    __syncthreads();
}
__device__ __forceinline__ int mt_find(const int* band0, int nd, int item) {   // largest i with band0[i] <= item (bands of i non-empty)
    int lo = 0, hi = nd;                    // invariant: band0[lo] <= item < band0[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (band0[mid] <= item) lo = mid; else hi = mid; }
    return lo;
}

// Work item = (detection, band of `pb` window rows); a 1-D grid strides over the items of the LIVE
// detections only, so one frame-sized window is spread over ~135 blocks instead of 16.  256 threads = 4
// waves; a wave covers 64 pixels per step; ballot -> one 64-bit word of the bit plane.
// Every block derives the paste windows of all detections itself (one thread per detection; block 0 also writes them to the
// record) and scans the band counts in parallel: no separate post-processing launch, no serial prefix.  The closest-point
// keys of the NEXT launch are reset here (they live in the record's `closest` field: 8 bytes per entry either way).
__global__ __launch_bounds__(256) void paste_masks(const PasteParams p, int n_max, unsigned long long* __restrict__ keys, int kd) {
    __shared__ float prob[32 * 32];
    __shared__ unsigned long long red[4][3];
    __shared__ int band0[MT_MAXDET + 1];       // first item of each detection (prefix over live detections)
    __shared__ int wtot[4];
    __shared__ int yt_i[PASTE_BAND_MAX];
    __shared__ float yt_w0[PASTE_BAND_MAX], yt_w1[PASTE_BAND_MAX];
    __shared__ int rowsv[MT_MAXDET];
    const int total = *p.total < n_max ? *p.total : n_max;
    const int nd = total < MT_MAXDET ? total : MT_MAXDET;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        const DetPost d = det_post(p, i);
        band0[i] = rowsv[i] = d.ry1 - d.ry0;
        if (blockIdx.x == 0) {
            p.boxes_out[i * 4 + 0] = d.x0; p.boxes_out[i * 4 + 1] = d.y0; p.boxes_out[i * 4 + 2] = d.x1; p.boxes_out[i * 4 + 3] = d.y1;
            p.valid[i] = d.ok;
            p.rect[i * 4 + 0] = d.rx0; p.rect[i * 4 + 1] = d.ry0; p.rect[i * 4 + 2] = d.rx1; p.rect[i * 4 + 3] = d.ry1;
        }
    }
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total * kd; t += gridDim.x * blockDim.x) keys[t] = ~0ull;
    __syncthreads();
    mt_scan(band0, nd, wtot);                                   // band0[nd] = rows of all windows
    const int pb = mt_pick_band(band0[nd], PASTE_BAND_MAX, PASTE_BAND_MIN, PASTE_ITEMS_MIN);
    __syncthreads();
    for (int i = threadIdx.x; i < nd; i += blockDim.x) band0[i] = (rowsv[i] + pb - 1) / pb;
    __syncthreads();
    mt_scan(band0, nd, wtot);
    const int nitems = band0[nd];
    int cur = -1;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        const int i = mt_find(band0, nd, item);
        const int band = item - band0[i];
        const int M = p.M;
        if (i != cur) {
            const float* lg = p.logits + (size_t)i * M * M * p.ldc + p.cls[i];
            __syncthreads();                    // previous item's prob readers are done
            for (int t = threadIdx.x; t < M * M; t += blockDim.x) prob[t] = 1.0f / (1.0f + expf(-lg[(size_t)t * p.ldc]));
            cur = i;
        }
        __syncthreads();
        const DetPost d = det_post(p, i);
        const int rx0 = d.rx0, ry0 = d.ry0, rx1 = d.rx1, ry1 = d.ry1;
        const float bx0 = d.x0, by0 = d.y0;
        const float bw = d.x1 - d.x0, bh = d.y1 - d.y0;
        const float Mf = (float)M;
        uint64_t* bits = p.bits + (size_t)i * p.out_h * p.words_per_row;
        unsigned long long mass = 0, sx = 0, sy = 0;
        const int w0 = rx0 >> 6, w1 = (rx1 + 63) >> 6;
        const int yb = ry0 + band * pb;
        const int ye = (yb + pb) < ry1 ? (yb + pb) : ry1;
        // The y terms of the band's rows are wave-uniform: threads 0..15 put them in LDS once.  A wave then takes whole
        // 64-pixel columns: the x terms (one IEEE division per pixel) are computed once per column and reused for every row
        // of the band -- the arithmetic of a pixel is unchanged (same expressions, same order).
        const int nrows = ye - yb;
        if (threadIdx.x < PASTE_BAND_MAX) {
            const int y = yb + threadIdx.x;
            // normalised y, grid_sample unnormalise (align_corners=False)
            const float gy = ((float)y + 0.5f - by0) / bh * 2.0f - 1.0f;
            const float iy = ((gy + 1.0f) * Mf - 1.0f) / 2.0f;
            const float fy = floorf(iy);
            const int iy0 = (int)fy;
            const float wy1 = iy - fy;                     // torch CPU grid_sample: n = y - floor(y), s = 1 - n
            yt_i[threadIdx.x] = iy0; yt_w1[threadIdx.x] = wy1; yt_w0[threadIdx.x] = 1.0f - wy1;
        }
        __syncthreads();
        for (int w = w0 + wave; w < w1; w += 4) {
            const int x = (w << 6) + lane;
            const bool xin = x >= rx0 && x < rx1;
            const float gx = ((float)x + 0.5f - bx0) / bw * 2.0f - 1.0f;
            const float ix = ((gx + 1.0f) * Mf - 1.0f) / 2.0f;
            const float fx = floorf(ix);
            const int ix0 = (int)fx, ix1 = ix0 + 1;
            const float wx1 = ix - fx, wx0 = 1.0f - wx1;
            const bool x0in = xin && (unsigned)ix0 < (unsigned)M, x1in = xin && (unsigned)ix1 < (unsigned)M;
            const int cx0 = x0in ? ix0 : 0, cx1 = x1in ? ix1 : 0;
            for (int r = 0; r < nrows; ++r) {
                const int y = yb + r;
                const int iy0 = yt_i[r], iy1 = iy0 + 1;
                const float wy0 = yt_w0[r], wy1 = yt_w1[r];
                const bool y0in = (unsigned)iy0 < (unsigned)M, y1in = (unsigned)iy1 < (unsigned)M;
                const int r0 = (y0in ? iy0 : 0) * M, r1 = (y1in ? iy1 : 0) * M;
                float v = 0.f;
                if (y0in && x0in) v += prob[r0 + cx0] * (wy0 * wx0);
                if (y0in && x1in) v += prob[r0 + cx1] * (wy0 * wx1);
                if (y1in && x0in) v += prob[r1 + cx0] * (wy1 * wx0);
                if (y1in && x1in) v += prob[r1 + cx1] * (wy1 * wx1);
                const bool on = xin && v >= p.thresh;
                const uint64_t word = __ballot(on);
                if (lane == 0) bits[(size_t)y * p.words_per_row + w] = word;
                if (on) { mass += 1; sx += (unsigned long long)(x + 1); sy += (unsigned long long)(y + 1); }
            }
        }
        for (int o = 32; o > 0; o >>= 1) {
            mass += __shfl_xor(mass, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o);
        }
        __syncthreads();                        // red[] of the previous item has been consumed
        if (lane == 0) { red[wave][0] = mass; red[wave][1] = sx; red[wave][2] = sy; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const unsigned long long t = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
            if (t) atomicAdd(p.sums + (size_t)i * 3 + threadIdx.x, t);
        }
    }
}

// Closest mask pixel of ONE 64-pixel word (row y, columns 64 w .. 64 w + 63) to an INTEGER target, in O(1) (round 4).
// compute_closest_point (dcnn/utils/mask_utils.py:6-23) takes the argmin of the f32 value fl(fl(dx^2) + fl(dy^2)) in row-major
// order.  Within a row dy is fixed, and for integer targets in a frame of at most 4096 rows fl(d) is STRICTLY increasing in |dx|:
// dx, dy and their squares (< 2^24) are exact; below 2^24 d is an exact integer; at d >= 2^24 (rounded to a multiple of 2)
// dy^2 < 2^24 forces |dx| >= 3464, so neighbouring |dx| differ by >= 6929 in d.  The row's minimiser is therefore the set bit with
// the smallest |dx| -- the nearest set bit at or left of the target column, or the nearest one right of it; at equal |dx| both have
// the same d and the left one (smaller row-major index) wins, which the (distance bits, index) key order gives for free.  Two
// candidates per (word, target) instead of one per set bit: the cost no longer depends on how many pixels the mask has (the
// synthetic weights' masks are speckle: restricting the scan to boundary pixels, tried first, only went from 491 to 162 us per
// 4 frames at ~38 detections per frame because almost every pixel of such a mask IS a boundary pixel).
__device__ __forceinline__ unsigned long long mt_word_nearest(uint64_t word, int w, int y, int out_w, float px, float py) {
    const int r = (int)px - 1 - (w << 6);                 // the target's column relative to the word (may be outside 0..63)
    const float dy = (float)(y + 1) - py;
    const float dy2 = dy * dy;
    const unsigned rowlin = (unsigned)(y * out_w + (w << 6));
    unsigned long long best = ~0ull;
    if (r >= 0) {
        const uint64_t m = r >= 63 ? word : (word & ((2ull << r) - 1ull));          // bits 0 .. r
        if (m) {
            const int bpos = 63 - __clzll((long long)m);
            const float dx = (float)((w << 6) + bpos + 1) - px;
            best = ((unsigned long long)__float_as_uint(dx * dx + dy2) << 32) | (rowlin + (unsigned)bpos);
        }
    }
    if (r < 63) {
        const uint64_t m = r < 0 ? word : ((word >> (r + 1)) << (r + 1));            // bits r + 1 .. 63
        if (m) {
            const int bpos = __ffsll((long long)m) - 1;
            const float dx = (float)((w << 6) + bpos + 1) - px;
            const unsigned long long key = ((unsigned long long)__float_as_uint(dx * dx + dy2) << 32) | (rowlin + (unsigned)bpos);
            best = key < best ? key : best;
        }
    }
    return best;
}

// closest[i][jl] for every detection i and every target jl of the same image (jl = index of the target among the image's
// detections).  Round 4 decomposition: a block = (mask i, one of up to CP_PARTS row ranges of its window).
//   * Per ROW, not per word: a row's minimiser for a target at column c is its nearest set bit on either side of c (the argument of
//     mt_word_nearest applies to the whole row).  The block first records every row's first and last set bit (one pass over the
//     words, shared by all targets); then for a (row, target) pair: c at or left of the first bit -> the first bit; at or right of
//     the last -> the last; inside the span -> the word under c gives the nearest bit on each side, and only when that word has
//     none on a side are neighbouring words walked (dense masks: never; a solid blob with a hole under c: a few words).  The work
//     is rows x targets instead of words x targets (8-30 x fewer pairs on 500-2000 px wide windows): the word form ran 124-129 us
//     per 4 frames at ~38 detections per frame whatever its decomposition (item list with block reductions, or wave tasks) --
//     28 M VALU wave-instructions, compute-bound.
//   * A wave takes target groups of eight (images with fewer than three groups: the waves split the range's rows instead), one row
//     per lane, reduces its eight minima by lane exchange and merges them with one global atomicMin per target on the 64-bit
//     (f32 distance bits, row-major index) keys IN the record's `closest` field; the host turns a key into 1-based (x, y) when it has
//     the record (apse_read_results_end).  No scans, no item search, no block barrier in the loop.
// Block (i, 0) writes detection i's centroid -- (floor(sum_x / mass), floor(sum_y / mass)) or (-1, -1) for an empty mask -- and
// mass to the record.
#ifndef CP_PARTS
#define CP_PARTS 16               // row ranges per mask (grid x)
#endif
#ifndef CP_ROWS_MIN
#define CP_ROWS_MIN 16            // rows per range before a mask is cut further (sweep, us per 4 frames at ~38 detections per frame:
#endif                            // 16 parts / 16 rows 102, 32 / 8 164, 64 / 4 310: the per-block prologue is what more blocks multiply)
#define CP_LDS_WORDS 6144         // words of a row range kept in LDS (48 KB): the walks along sparse rows then never leave the CU
#define CP_ROWS_MAX 256           // rows of a range (frames up to 4096 rows: 4096 / CP_PARTS fits with room)
__device__ __forceinline__ unsigned long long mt_key(int x, int y, int out_w, float px, float py) {
    const float dx = (float)(x + 1) - px, dy = (float)(y + 1) - py;
    return ((unsigned long long)__float_as_uint(dx * dx + dy * dy) << 32) | (unsigned)(y * out_w + x);
}
template <bool WORDWISE>      // false: frames wider or taller than 4096 pixels (the exactness argument needs both bounds): every pixel
__global__ __launch_bounds__(256) void closest_points(const uint64_t* __restrict__ bits_all, const int* __restrict__ rect,
                                                      const int* __restrict__ valid, const unsigned long long* __restrict__ sums,
                                                      const int* __restrict__ img, const int* __restrict__ offset,
                                                      const int* __restrict__ total, int n_max, int kd, int out_h, int out_w,
                                                      int words_per_row, unsigned long long* __restrict__ keys,
                                                      int* __restrict__ cent_out, int* __restrict__ mass_out) {
    __shared__ int cent[MT_TARGETS][2];
    __shared__ int rfirst[CP_ROWS_MAX], rlast[CP_ROWS_MAX];        // first / last set bit (frame column) of the range's rows; empty row: last = -1
    __shared__ uint64_t lw[WORDWISE ? CP_LDS_WORDS : 1];           // the range's words, row-major [rows][nw]
    const int n = *total < n_max ? *total : n_max;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = blockIdx.y; i < n; i += gridDim.y) {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            const unsigned long long m = sums[i * 3], sx = sums[i * 3 + 1], sy = sums[i * 3 + 2];
            mass_out[i] = (int)m;
            cent_out[i * 2] = m ? (int)(sx / m) : -1;
            cent_out[i * 2 + 1] = m ? (int)(sy / m) : -1;
        }
        const int rx0 = rect[i * 4 + 0], ry0 = rect[i * 4 + 1], rx1 = rect[i * 4 + 2], ry1 = rect[i * 4 + 3];
        const int rows = valid[i] ? ry1 - ry0 : 0;
        int P = (rows + CP_ROWS_MIN - 1) / CP_ROWS_MIN;
        P = P < 1 ? 1 : (P > (int)gridDim.x ? (int)gridDim.x : P);
        if (rows <= 0 || (int)blockIdx.x >= P) continue;            // block-uniform
        const int j0 = offset[img[i]];
        int j1 = offset[img[i] + 1];
        j1 = j1 < n ? j1 : n;
        const int nt = j1 - j0 < MT_TARGETS ? j1 - j0 : MT_TARGETS;   // kd <= 100 targets per image (apse_create)
        const int rpp = (rows + P - 1) / P;
        const int pb = ry0 + (int)blockIdx.x * rpp;
        const int pe = pb + rpp < ry1 ? pb + rpp : ry1;
        const int w0 = rx0 >> 6, w1 = (rx1 + 63) >> 6;
        const uint64_t* bits = bits_all + (size_t)i * out_h * words_per_row;
        __syncthreads();                                            // the previous mask's tables are no longer read
        for (int t = threadIdx.x; t < nt; t += blockDim.x) {
            const unsigned long long m = sums[(j0 + t) * 3], sx = sums[(j0 + t) * 3 + 1], sy = sums[(j0 + t) * 3 + 2];
            cent[t][0] = m ? (int)(sx / m) : -1;
            cent[t][1] = m ? (int)(sy / m) : -1;
        }
        const int nw = w1 - w0;
        const bool in_lds = WORDWISE && (pe - pb) * nw <= CP_LDS_WORDS;
        if (WORDWISE) {
            for (int r = threadIdx.x; r < pe - pb && r < CP_ROWS_MAX; r += blockDim.x) { rfirst[r] = 0x7fffffff; rlast[r] = -1; }
            __syncthreads();
            // one coalesced pass over the range's words: every word goes to LDS (when the range fits) and bids for its row's first /
            // last set bit with LDS atomics -- all loads independent (a thread walking its row word by word made this pass a chain of
            // 20-60 dependent round trips per block)
            int ry = (int)threadIdx.x / nw, wx = (int)threadIdx.x - ry * nw;
            const int dry = 256 / nw, dwx = 256 - dry * nw;
            for (int t = threadIdx.x; t < (pe - pb) * nw; t += 256) {
                const uint64_t word = bits[(size_t)(pb + ry) * words_per_row + w0 + wx];
                if (in_lds) lw[t] = word;
                if (word && ry < CP_ROWS_MAX) {
                    atomicMin(&rfirst[ry], ((w0 + wx) << 6) + __ffsll((long long)word) - 1);
                    atomicMax(&rlast[ry], ((w0 + wx) << 6) + 63 - __clzll((long long)word));
                }
                ry += dry; wx += dwx;
                if (wx >= nw) { wx -= nw; ++ry; }
            }
        }
        __syncthreads();
        const int ngroups = (nt + 7) >> 3;
        // few target groups: the four waves split the rows of the range instead of the groups
        const int wsplit = ngroups >= 3 ? 1 : (ngroups == 2 ? 2 : 4);
        const int sub = wave % wsplit;
        const int srows = (pe - pb + wsplit - 1) / wsplit;
        const int yb = pb + sub * srows;
        const int nrows = (yb + srows < pe ? yb + srows : pe) - yb;
        if (nrows <= 0) continue;                                   // wave-uniform; every wave still reaches the barriers at the loop top
        for (int jg = (wave / wsplit) * 8; jg < nt; jg += (4 / wsplit) * 8) {
            float px[8], py[8];
            unsigned long long b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bool ok = jg + k < nt && cent[jg + k < nt ? jg + k : 0][0] >= 0;
                px[k] = ok ? (float)cent[jg + k][0] : 0.f;
                py[k] = ok ? (float)cent[jg + k][1] : 0.f;
                b[k] = ~0ull;
            }
            if (WORDWISE) {
                for (int ry = lane; ry < nrows; ry += 64) {
                    const int y = yb + ry;
                    const int f = rfirst[y - pb], l = rlast[y - pb];
                    if (l < 0) continue;
                    const uint64_t* grow = bits + (size_t)y * words_per_row;       // the row in the bit plane ...
                    const int lbase = (y - pb) * nw - w0;                          // ... and in LDS: lw[lbase + frame word]
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int c = (int)px[k] - 1;                  // the target's column (0-based)
                        unsigned long long key;
                        if (c <= f) key = mt_key(f, y, out_w, px[k], py[k]);
                        else if (c >= l) key = mt_key(l, y, out_w, px[k], py[k]);
                        else {
                            // f < c < l: a set bit exists on both sides.  Nearest at or left of c, nearest right of c.
                            const int w = c >> 6;
                            const int r = c & 63;
                            const uint64_t word = in_lds ? lw[lbase + w] : grow[w];
                            uint64_t m = r >= 63 ? word : (word & ((2ull << r) - 1ull));
                            int ww = w;
                            while (!m) { --ww; m = in_lds ? lw[lbase + ww] : grow[ww]; }      // ends at the first bit's word at the latest
                            const int xl = (ww << 6) + 63 - __clzll((long long)m);
                            m = r >= 63 ? 0ull : ((word >> (r + 1)) << (r + 1));
                            ww = w;
                            while (!m) { ++ww; m = in_lds ? lw[lbase + ww] : grow[ww]; }      // ends at the last bit's word at the latest
                            const int xr = (ww << 6) + __ffsll((long long)m) - 1;
                            const unsigned long long kl = mt_key(xl, y, out_w, px[k], py[k]), kr = mt_key(xr, y, out_w, px[k], py[k]);
                            key = kl < kr ? kl : kr;
                        }
                        b[k] = key < b[k] ? key : b[k];
                    }
                }
            } else {
                for (int t = lane; t < nw * nrows; t += 64) {
                    const int ry = t / nw, w = w0 + (t - ry * nw);
                    const int y = yb + ry;
                    uint64_t word = bits[(size_t)y * words_per_row + w];
                    while (word) {
                        const int bit = __ffsll((long long)word) - 1;
                        word &= word - 1;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const unsigned long long key = mt_key((w << 6) + bit, y, out_w, px[k], py[k]);
                            b[k] = key < b[k] ? key : b[k];
                        }
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                unsigned long long v = b[k];
                for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(v, o); v = other < v ? other : v; }
                if (lane == 0 && v != ~0ull && jg + k < nt && cent[jg + k][0] >= 0) atomicMin(keys + (size_t)i * kd + jg + k, v);
            }
        }
    }
}

// Mask windows of up to MW_MAX detections out of the bit planes in one launch (TrackRCNN.instances_from: the reference's
// pred_masks of a frame; round 3 issued one 2-D copy per detection).  Window k: rows[k] rows of nw[k] words from src word offset
// src[k] (row pitch = words_per_row) to dst word offset dst[k] (dense).  blockIdx.y = window.
#define MW_MAX 100
struct MaskWindows { long long src[MW_MAX], dst[MW_MAX]; int nw[MW_MAX], rows[MW_MAX]; };
__global__ __launch_bounds__(256) void copy_mask_windows(const uint64_t* __restrict__ bits, uint64_t* __restrict__ out,
                                                         const MaskWindows mw, int words_per_row) {
    const int k = blockIdx.y;
    const int nw = mw.nw[k], total = nw * mw.rows[k];
    const uint64_t* s = bits + mw.src[k];
    uint64_t* d = out + mw.dst[k];
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int r = t / nw, w = t - r * nw;
        d[t] = s[(size_t)r * words_per_row + w];
    }
}

// Expand one detection's packed bits to a dense bool (uint8) frame: the reference's pred_masks view.
__global__ __launch_bounds__(256) void bits_to_dense(const uint64_t* __restrict__ bits, const int* __restrict__ rect4,
                                                     int out_h, int out_w, int words_per_row, uint8_t* __restrict__ dense) {
    const size_t total = (size_t)out_h * out_w;
    const int rx0 = rect4[0], ry0 = rect4[1], rx1 = rect4[2], ry1 = rect4[3];
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(e / out_w), x = (int)(e - (size_t)y * out_w);
        uint8_t v = 0;
        if (x >= rx0 && x < rx1 && y >= ry0 && y < ry1) v = (bits[(size_t)y * words_per_row + (x >> 6)] >> (x & 63)) & 1ull;
        dense[e] = v;
    }
}

// Dense bool frame (any mask, e.g. user supplied) -> packed bits + rect = full frame; sums accumulated.
__global__ __launch_bounds__(256) void dense_to_bits(const uint8_t* __restrict__ dense, int out_h, int out_w,
                                                     int words_per_row, uint64_t* __restrict__ bits,
                                                     unsigned long long* __restrict__ sums) {
    const int lane = threadIdx.x & 63;
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwords = out_h * words_per_row;
    unsigned long long mass = 0, sx = 0, sy = 0;
    for (int wi = gw; wi < nwords; wi += (gridDim.x * blockDim.x) >> 6) {
        const int y = wi / words_per_row, w = wi - y * words_per_row;
        const int x = (w << 6) + lane;
        const bool on = x < out_w && dense[(size_t)y * out_w + x] != 0;
        const uint64_t word = __ballot(on);
        if (lane == 0) bits[wi] = word;
        if (on) { mass += 1; sx += (unsigned long long)(x + 1); sy += (unsigned long long)(y + 1); }
    }
    for (int o = 32; o > 0; o >>= 1) { mass += __shfl_xor(mass, o); sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); }
    if (lane == 0 && mass) { atomicAdd(sums, mass); atomicAdd(sums + 1, sx); atomicAdd(sums + 2, sy); }
}

// Closest point of ONE packed mask (full-frame rect) to an explicit target (stateless op for mask_utils).  WORDWISE: the target is
// an integer pixel position and the frame has at most 4096 rows -> two candidates per word (mt_word_nearest, the context's table
// kernel); any other target: every pixel.
template <bool WORDWISE>
__global__ __launch_bounds__(256) void closest_point_single(const uint64_t* __restrict__ bits, int out_h, int out_w,
                                                            int words_per_row, float px, float py,
                                                            unsigned long long* __restrict__ best_out) {
    unsigned long long b = ~0ull;
    const int nwords = out_h * words_per_row;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nwords; t += gridDim.x * blockDim.x) {
        const int y = t / words_per_row, w = t - y * words_per_row;
        uint64_t word = bits[t];
        if (!word) continue;
        if (WORDWISE) {
            const unsigned long long key = mt_word_nearest(word, w, y, out_w, px, py);
            b = key < b ? key : b;
            continue;
        }
        const float dy = (float)(y + 1) - py;
        const float dy2 = dy * dy;
        while (word) {
            const int bit = __ffsll((long long)word) - 1;
            word &= word - 1;
            const int x = (w << 6) + bit;
            const float dx = (float)(x + 1) - px;
            const float d = dx * dx + dy2;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)(y * out_w + x);
            b = key < b ? key : b;
        }
    }
    for (int k = 32; k > 0; k >>= 1) { const unsigned long long other = __shfl_xor(b, k); b = other < b ? other : b; }
    if ((threadIdx.x & 63) == 0 && b != ~0ull) atomicMin(best_out, b);
}

extern "C" {
int apse_k_closest_single(const uint64_t* bits, int out_h, int out_w, int words_per_row, float px, float py,
                          unsigned long long* best_out, hipStream_t s) {
    hipMemsetAsync(best_out, 0xff, sizeof(unsigned long long), s);
    const bool integral = px == floorf(px) && py == floorf(py) && fabsf(px) < 1048576.f && fabsf(py) < 1048576.f && out_h <= 4096 && out_w <= 4096;
    if (integral) hipLaunchKernelGGL(closest_point_single<true>, dim3(512), dim3(256), 0, s, bits, out_h, out_w, words_per_row, px, py, best_out);
    else hipLaunchKernelGGL(closest_point_single<false>, dim3(512), dim3(256), 0, s, bits, out_h, out_w, words_per_row, px, py, best_out);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_mask_paste(const PasteParams* p, int n_max, unsigned long long* keys, int kd, hipStream_t s) {
    if (n_max <= 0) return APSE_OK;
    if (p->M > 32 || n_max > MT_MAXDET) return APSE_E_INVALID;
    hipLaunchKernelGGL(paste_masks, dim3(PASTE_BLOCKS), dim3(256), 0, s, *p, n_max, keys, kd);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_closest_points(const uint64_t* bits, const int* rect, const int* valid, const unsigned long long* sums, const int* img,
                          const int* offset, const int* total, int n_max, int kd, int out_h, int out_w, int words_per_row,
                          int* cent, int* mass, unsigned long long* keys, int hint, hipStream_t s) {
    if (n_max <= 0) return APSE_OK;
    // grid: CP_PARTS row ranges x masks; the masks dimension is sized from the caller's hint of the live count (any count is handled:
    // the blocks stride over the masks)
    int gy = hint > 0 ? 2 * hint : 16;
    gy = gy < 16 ? 16 : (gy > n_max ? n_max : gy);
    if (out_h <= 4096 && out_w <= 4096)
        hipLaunchKernelGGL(closest_points<true>, dim3(CP_PARTS, gy), dim3(256), 0, s, bits, rect, valid, sums, img, offset, total, n_max, kd, out_h,
                           out_w, words_per_row, keys, cent, mass);
    else
        hipLaunchKernelGGL(closest_points<false>, dim3(CP_PARTS, gy), dim3(256), 0, s, bits, rect, valid, sums, img, offset, total, n_max, kd, out_h,
                           out_w, words_per_row, keys, cent, mass);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_copy_mask_windows(const uint64_t* bits, uint64_t* out, int n, const long long* src, const long long* dst, const int* nw,
                             const int* rows, int words_per_row, hipStream_t s) {
    if (n <= 0) return APSE_OK;
    if (n > MW_MAX) return APSE_E_INVALID;
    MaskWindows mw;
    int most = 1;
    for (int k = 0; k < n; ++k) {
        mw.src[k] = src[k]; mw.dst[k] = dst[k]; mw.nw[k] = nw[k]; mw.rows[k] = rows[k];
        most = nw[k] * rows[k] > most ? nw[k] * rows[k] : most;
    }
    int gx = (most + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(copy_mask_windows, dim3(gx, n), dim3(256), 0, s, bits, out, mw, words_per_row);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_bits_to_dense(const uint64_t* bits, const int* rect4, int out_h, int out_w, int words_per_row, uint8_t* dense,
                         hipStream_t s) {
    hipLaunchKernelGGL(bits_to_dense, dim3(2048), dim3(256), 0, s, bits, rect4, out_h, out_w, words_per_row, dense);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_dense_to_bits(const uint8_t* dense, int out_h, int out_w, int words_per_row, uint64_t* bits,
                         unsigned long long* sums, hipStream_t s) {
    hipMemsetAsync(sums, 0, 3 * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(dense_to_bits, dim3(1024), dim3(256), 0, s, dense, out_h, out_w, words_per_row, bits, sums);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
}
