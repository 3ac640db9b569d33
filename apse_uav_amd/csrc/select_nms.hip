// Decision kernels of the dcnn hot path (gfx950): top-k, box decoding, NMS, final ranking.
// Everything here is f32 with -ffp-contract=off so that, given identical inputs, the index
// results are identical to the CPU oracle's (each product and sum rounded separately, IEEE
// division).  Ordering rule everywhere: score descending, ties by ascending input index
// (DESIGN.md "Tie-breaks").
//
// Restates (from detectron2 0.1.2 / torchvision 0.6, reached from
// /root/reference/dcnn/networks/track_rcnn.py:46,51):
//   RPNOutputs.predict_proposals + find_top_rpn_proposals  -> rpn_topk_stage, rpn_decode,
//                                                             nms_percat, rank_final
//   FastRCNNOutputs.inference / fast_rcnn_inference_single_image -> box_candidates,
//                                                             nms_percat, rank_final
#include "apse_common.h"

#define TK_N 4096          // elements sorted per block in the top-k tournament
#define NMS_MAX 1024       // boxes per category (<= 1000 by construction)

__device__ __forceinline__ uint32_t mono_key(float f) {   // order-preserving float -> uint
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float mono_inv(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}
// ascending sort of this composite = score descending, index ascending
__device__ __forceinline__ uint64_t comp_key(float score, uint32_t idx) {
    return ((uint64_t)(~mono_key(score)) << 32) | idx;
}
__device__ __forceinline__ float comp_score(uint64_t k) { return mono_inv(~(uint32_t)(k >> 32)); }

__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {   // l must be wave-uniform
    const uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return ((uint64_t)hi << 32) | lo;
}

// Bitonic sort of N = 1024 * EPT 64-bit keys in LDS, ascending, by a 1024-thread block.  Thread t owns the EPT consecutive
// keys a[EPT t ..] in registers: a compare-exchange distance j < EPT stays inside the thread, j < 64 EPT inside the wave
// (64-bit lane exchange), and only j >= 64 EPT goes through LDS with a block barrier -- 10 of the 78 steps for N = 4096
// (EPT = 4), 10 of 55 for N = 1024 (EPT = 1).  The all-LDS form paid a barrier per step and ran at ~0.55 us a step with one
// block per CU (rpn_topk_stage 43 us per 4096 keys).  Same network, same result (the keys are distinct).
template <int EPT>
__device__ __forceinline__ void block_bitonic_sort(uint64_t* a, int tid) {
    constexpr int N = 1024 * EPT;
    uint64_t e[EPT];
#pragma unroll
    for (int s = 0; s < EPT; ++s) e[s] = a[tid * EPT + s];
    for (int k = 2; k <= N; k <<= 1) {
        int j = k >> 1;
        if (j >= 64 * EPT) {
            __syncthreads();                               // every thread has taken its keys out of LDS
#pragma unroll
            for (int s = 0; s < EPT; ++s) a[tid * EPT + s] = e[s];
            __syncthreads();
            for (; j >= 64 * EPT; j >>= 1) {
                for (int pidx = tid; pidx < N / 2; pidx += 1024) {
                    const int i = ((pidx & ~(j - 1)) << 1) | (pidx & (j - 1));
                    const int l = i | j;
                    const bool asc = (i & k) == 0;
                    const uint64_t x = a[i], y = a[l];
                    if ((x > y) == asc) { a[i] = y; a[l] = x; }
                }
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < EPT; ++s) e[s] = a[tid * EPT + s];
        }
        for (; j >= EPT; j >>= 1) {                        // partner key lives in lane ^ (j / EPT), same register slot
            const int tj = j / EPT;
            const bool lower = (tid & tj) == 0;
#pragma unroll
            for (int s = 0; s < EPT; ++s) {
                const bool asc = ((tid * EPT + s) & k) == 0;
                const uint64_t o = __shfl_xor(e[s], tj);
                const bool take_min = lower == asc;
                e[s] = ((o < e[s]) == take_min) ? o : e[s];
            }
        }
#pragma unroll
        for (int jj = EPT / 2; jj >= 1; jj >>= 1) {        // both keys in this thread
            if (jj > (k >> 1)) continue;
#pragma unroll
            for (int s = 0; s < EPT; ++s) {
                if (s & jj) continue;
                const bool asc = ((tid * EPT + s) & k) == 0;
                const uint64_t x = e[s], y = e[s | jj];
                if ((x > y) == asc) { e[s] = y; e[s | jj] = x; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < EPT; ++s) a[tid * EPT + s] = e[s];
    __syncthreads();
}

// ---------------------------------------------------------------- RPN top-k tournament
struct RpnLevel {
    const float* head;     // [B][H*W][head_ld] : channels 0..2 objectness, 3..14 deltas (a*4+coord)
    int H, W, stride;
    int n;                 // H*W*3
    int k;                 // min(pre_topk, n)
    float base[3][4];      // cell anchors (x0,y0,x1,y1)
};
struct RpnLevels {
    RpnLevel lv[5];
    int head_ld;
    int pre_topk;          // 1000
};
struct TopkJob {
    int kind;              // 0: raw logits chunk, 1: merge of lists
    int level;
    int begin, count;      // kind 0: element range within the level
    int nsrc;
    int src[4];            // kind 1: source list slots
    int src_count[4];
    int dst;               // destination list slot
    int dst_count;         // min(pre_topk, total)
};

// lists: [B][nslots][1024] u64.  kind 0: sort one 4096-element chunk of raw logits (bitonic, LDS).
// kind 1: merge up to four sorted lists by rank (position + binary-search counts in the other lists):
// no sort, no barriers after the load.  `zero_word` (stage 0 only): image b's max-coordinate cell is
// cleared here so the decode kernel's atomicMax needs no separate memset.
__global__ __launch_bounds__(1024) void rpn_topk_stage(const RpnLevels* __restrict__ Lp, const TopkJob* __restrict__ jobs,
                                                       uint64_t* __restrict__ lists, int nslots,
                                                       uint32_t* __restrict__ zero_word) {
    __shared__ uint64_t a[TK_N];
    const TopkJob* job = jobs + blockIdx.x;
    const int b = blockIdx.y, tid = threadIdx.x;
    uint64_t* pool = lists + (size_t)b * nslots * 1024;
    if (zero_word && blockIdx.x == 0 && tid == 0) zero_word[b] = 0u;
    if (job->kind == 0) {
        const RpnLevel* lv = &Lp->lv[job->level];
        const int head_ld = Lp->head_ld;
        const float* head = lv->head + (size_t)b * lv->H * lv->W * head_ld;
        const int jb = job->begin, jc = job->count;
        for (int i = tid; i < TK_N; i += 1024) {
            uint64_t k = ~0ull;
            if (i < jc) {
                const int e = jb + i;
                const int pix = e / 3, an = e - pix * 3;
                k = comp_key(head[(size_t)pix * head_ld + an], (uint32_t)e);
            }
            a[i] = k;
        }
        __syncthreads();
        block_bitonic_sort<TK_N / 1024>(a, tid);
        if (tid < job->dst_count) pool[(size_t)job->dst * 1024 + tid] = a[tid];
    } else {
        const int ns = job->nsrc;
        int cnt[4];
        for (int s = 0; s < 4; ++s) cnt[s] = s < ns ? job->src_count[s] : 0;
        for (int s = 0; s < ns; ++s)
            if (tid < cnt[s]) a[s * 1024 + tid] = pool[(size_t)job->src[s] * 1024 + tid];
        __syncthreads();
        const int K = job->dst_count;
        uint64_t* dst = pool + (size_t)job->dst * 1024;
        for (int s = 0; s < ns; ++s) {
            if (tid >= cnt[s]) continue;
            const uint64_t key = a[s * 1024 + tid];
            int rank = tid;
            for (int o = 0; o < ns; ++o) {
                if (o == s) continue;
                int lo = 0, hi = cnt[o];
                const uint64_t* ko = a + o * 1024;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (ko[mid] < key) lo = mid + 1; else hi = mid; }
                rank += lo;
            }
            if (rank < K) dst[rank] = key;
        }
    }
}

// Decode the selected anchors of every level: Box2BoxTransform.apply_deltas (weights 1,1,1,1),
// clip to the image, nonempty flag, and the running max coordinate of the kept boxes.
// out arrays are [B][5*pre_topk].
__global__ __launch_bounds__(256) void rpn_decode(const RpnLevels* __restrict__ Lp, const uint64_t* __restrict__ lists, int nslots,
                                                  const int* __restrict__ final_slot, float img_h, float img_w,
                                                  float scale_clamp, float* __restrict__ boxes, float* __restrict__ scores,
                                                  int* __restrict__ valid, uint32_t* __restrict__ maxc, int level_mask) {
    const int b = blockIdx.z, l = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int pre_topk = Lp->pre_topk, head_ld = Lp->head_ld;
    if (i >= pre_topk) return;
    const RpnLevel lv = Lp->lv[l];
    const size_t o = (size_t)b * 5 * pre_topk + (size_t)l * pre_topk + i;
    if (i >= lv.k || !((level_mask >> l) & 1)) { valid[o] = 0; scores[o] = 0.f; return; }
    const uint64_t key = lists[((size_t)b * nslots + final_slot[l]) * 1024 + i];
    const uint32_t e = (uint32_t)key;
    const float sc = comp_score(key);
    const int pix = e / 3, an = e - pix * 3;
    const int y = pix / lv.W, x = pix - y * lv.W;
    const float sx = (float)(x * lv.stride), sy = (float)(y * lv.stride);
    const float ax0 = sx + lv.base[an][0], ay0 = sy + lv.base[an][1];
    const float ax1 = sx + lv.base[an][2], ay1 = sy + lv.base[an][3];
    const float* d = lv.head + ((size_t)b * lv.H * lv.W + pix) * head_ld + 3 + an * 4;
    const float w = ax1 - ax0, h = ay1 - ay0;
    const float cx = ax0 + 0.5f * w, cy = ay0 + 0.5f * h;
    float dx = d[0] / 1.0f, dy = d[1] / 1.0f, dw = d[2] / 1.0f, dh = d[3] / 1.0f;
    dw = dw > scale_clamp ? scale_clamp : dw;
    dh = dh > scale_clamp ? scale_clamp : dh;
    const float pcx = dx * w + cx, pcy = dy * h + cy;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float x0 = pcx - 0.5f * pw, y0 = pcy - 0.5f * ph, x1 = pcx + 0.5f * pw, y1 = pcy + 0.5f * ph;
    x0 = fminf(fmaxf(x0, 0.f), img_w);
    y0 = fminf(fmaxf(y0, 0.f), img_h);
    x1 = fminf(fmaxf(x1, 0.f), img_w);
    y1 = fminf(fmaxf(y1, 0.f), img_h);
    const int ok = ((x1 - x0) > 0.f) && ((y1 - y0) > 0.f);
    boxes[o * 4 + 0] = x0; boxes[o * 4 + 1] = y0; boxes[o * 4 + 2] = x1; boxes[o * 4 + 3] = y1;
    scores[o] = sc;
    valid[o] = ok;
    if (ok) {
        const float m = fmaxf(fmaxf(x0, x1), fmaxf(y0, y1));   // all >= 0 after the clip
        atomicMax(maxc + b, __float_as_uint(m));
    }
}

// ---------------------------------------------------------------- per-category NMS
// Entries: boxes/scores/valid are [B][n_total]; the category of entry e is
// cat_div ? e / cat_div : e % cat_mod.  Within a category, entries are ordered by (score desc,
// entry index asc); IoU is evaluated on boxes shifted by cat * (max_coord + 1) in f32 (torchvision
// batched_nms); `iou > thr` suppresses.  Three launches:
//   nms_prepare : one block per (category, image): ordered compaction + sort -> sorted shifted boxes
//   nms_matrix  : 64x64 tiles of the upper-triangular suppression bit matrix, one wave each
//   nms_scan    : one block per (category, image): greedy scan, 64 rows at a time
// scratch per (image, category): entry[1024] int, box[4][1024] f32, area[1024] f32, mask[1024][16] u64, n.
struct NmsScratch {
    int* entry; float* box; float* area; uint64_t* mask; int* n;
};

__global__ __launch_bounds__(1024) void nms_prepare(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                    const int* __restrict__ valid, int n_total, int cat_div, int cat_mod,
                                                    const uint32_t* __restrict__ maxc, NmsScratch S, int ncat, int cat_shift,
                                                    int presorted) {
    __shared__ uint64_t keys[NMS_MAX];
    __shared__ int wsum[17];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    boxes += (size_t)b * n_total * 4;
    scores += (size_t)b * n_total;
    valid += (size_t)b * n_total;
    const size_t slot = (size_t)b * ncat + c;
    if (tid == 0) wsum[16] = 0;
    keys[tid] = ~0ull;
    __syncthreads();
    for (int base = 0; base < n_total; base += 1024) {
        const int e = base + tid;
        bool f = false;
        if (e < n_total) {
            const int cat = cat_div ? e / cat_div : e % cat_mod;
            f = (cat == c) && valid[e];
        }
        const uint64_t bal = __ballot(f);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int off = wsum[16];
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (f && pos < NMS_MAX) keys[pos] = comp_key(scores[e], (uint32_t)e);
        __syncthreads();
        if (tid == 0) { int t = wsum[16]; for (int w = 0; w < 16; ++w) t += wsum[w]; wsum[16] = t; }
        __syncthreads();
    }
    int n = wsum[16];
    n = n < NMS_MAX ? n : NMS_MAX;
    // presorted: the entries of a category already come in (score desc, entry asc) order -- the RPN stage, whose entry
    // l * pre_topk + i is rank i of level l's top-k list (sorted by the same key; ties by anchor index = by i) -- so the ordered
    // compaction above IS the sorted list and the 55 barrier steps of the bitonic network are skipped
    if (!presorted) block_bitonic_sort<NMS_MAX / 1024>(keys, tid);
    // batched_nms numbers the categories that are present in the call: cat_shift = index of the first one
    const float off = (float)(c - cat_shift) * (__uint_as_float(maxc[b]) + 1.0f);
    float* bx = S.box + slot * 4 * NMS_MAX;
    if (tid < n) {
        const uint32_t e = (uint32_t)keys[tid];
        const float x0 = boxes[e * 4 + 0] + off, y0 = boxes[e * 4 + 1] + off;
        const float x1 = boxes[e * 4 + 2] + off, y1 = boxes[e * 4 + 3] + off;
        bx[tid] = x0; bx[NMS_MAX + tid] = y0; bx[2 * NMS_MAX + tid] = x1; bx[3 * NMS_MAX + tid] = y1;
        S.area[slot * NMS_MAX + tid] = (x1 - x0) * (y1 - y0);
        S.entry[slot * NMS_MAX + tid] = (int)e;
    }
    if (tid == 0) S.n[slot] = n;
}

// grid (16 column chunks, 16 row chunks, ncat*B), 64 threads: thread t owns row 64*ic + t.
__global__ __launch_bounds__(64) void nms_matrix(NmsScratch S, float thr) {
    __shared__ float cb[5][64];
    const int jc = blockIdx.x, ic = blockIdx.y;
    const size_t slot = blockIdx.z;
    const int n = S.n[slot];
    if (jc < ic || 64 * ic >= n || 64 * jc >= n) return;
    const int t = threadIdx.x;
    const float* bx = S.box + slot * 4 * NMS_MAX;
    const float* ar = S.area + slot * NMS_MAX;
    const int j = 64 * jc + t;
    if (j < n) {
        cb[0][t] = bx[j]; cb[1][t] = bx[NMS_MAX + j]; cb[2][t] = bx[2 * NMS_MAX + j]; cb[3][t] = bx[3 * NMS_MAX + j];
        cb[4][t] = ar[j];
    }
    __syncthreads();
    const int i = 64 * ic + t;
    uint64_t bits = 0;
    if (i < n) {
        const float ix0 = bx[i], iy0 = bx[NMS_MAX + i], ix1 = bx[2 * NMS_MAX + i], iy1 = bx[3 * NMS_MAX + i];
        const float ia = ar[i];
        const int jend = (n - 64 * jc) < 64 ? (n - 64 * jc) : 64;
        for (int jj = 0; jj < jend; ++jj) {
            if (64 * jc + jj <= i) continue;
            const float xx0 = fmaxf(ix0, cb[0][jj]), yy0 = fmaxf(iy0, cb[1][jj]);
            const float xx1 = fminf(ix1, cb[2][jj]), yy1 = fminf(iy1, cb[3][jj]);
            const float ww = fmaxf(0.f, xx1 - xx0), hh = fmaxf(0.f, yy1 - yy0);
            const float inter = ww * hh;
            const float ovr = inter / (ia + cb[4][jj] - inter);
            if (ovr > thr) bits |= (1ull << jj);
        }
    }
    S.mask[(slot * NMS_MAX + i) * 16 + jc] = bits;
}

__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    return v;
}

// Greedy scan, one block per (category, image).  Wave 0 resolves the 64 rows of a chunk on the
// diagonal word (scalar readlanes), then wave w ORs the kept rows' word w into the removed bitmap.
__global__ __launch_bounds__(1024) void nms_scan(NmsScratch S, int* __restrict__ keep_idx, int* __restrict__ keep_cnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* mask = reinterpret_cast<uint64_t*>(smem);          // [n][16]
    __shared__ uint64_t removed[16];
    __shared__ uint64_t keepbits_s;
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = S.n[slot];
    const int nw = (n + 63) >> 6;
    const uint64_t* gm = S.mask + slot * NMS_MAX * 16;
    for (int i = tid; i < n * 16; i += 1024) {
        const int r = i >> 4, w = i & 15;
        mask[i] = (w >= (r >> 6) && w < nw) ? gm[i] : 0ull;       // lower-triangle words were never written
    }
    if (tid < 16) removed[tid] = 0;
    __syncthreads();
    int kept = 0;
    int* out = keep_idx + slot * NMS_MAX;
    const int* entry = S.entry + slot * NMS_MAX;
    for (int ch = 0; ch < nw; ++ch) {
        const int rows_here = (n - 64 * ch) < 64 ? (n - 64 * ch) : 64;
        if (wave == 0) {
            const int row = 64 * ch + lane;
            const uint64_t diag = row < n ? mask[(size_t)row * 16 + ch] : 0ull;
            uint64_t rem = removed[ch];
            uint64_t keepbits = 0;
            // visit only rows that are still alive: the next kept row is the lowest clear bit of `rem` above the
            // last one (a removed row never suppresses anything), so the loop runs once per KEPT row, not per row
            uint64_t alive = ~rem & (rows_here == 64 ? ~0ull : ((1ull << rows_here) - 1ull));
            while (alive) {
                const int r = __builtin_ctzll(alive);
                keepbits |= (1ull << r);
                rem |= readlane64(diag, r) | (1ull << r);
                alive &= ~rem;
            }
            if ((keepbits >> lane) & 1ull) out[kept + __popcll(keepbits & ((1ull << lane) - 1ull))] = entry[row];
            kept += __popcll(keepbits);
            if (lane == 0) keepbits_s = keepbits;
        }
        __syncthreads();
        if (wave > ch && wave < nw) {
            const uint64_t kb = keepbits_s;
            uint64_t v = 0;
            if (lane < rows_here && ((kb >> lane) & 1ull)) v = mask[(size_t)(64 * ch + lane) * 16 + wave];
            v = wave_or64(v);
            if (lane == 0) removed[wave] |= v;
        }
        __syncthreads();
    }
    if (tid == 0) keep_cnt[slot] = kept;
}

// Final ranking by merging: each category's kept list is already sorted by (score desc, entry asc),
// so the global rank of an element is its own position plus, for every other list, the number of
// keys that precede it (binary search).  No barriers after the load; first K ranks are written.
// Grid (image, category): every block loads all lists of its image (the searches need them) and ranks the elements of ITS
// list -- the ncat x 4 dependent binary searches of one block were 8 of this launch's 21 us at batch 1.
__global__ __launch_bounds__(1024) void rank_merge(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                   int n_total, const int* __restrict__ keep_idx,
                                                   const int* __restrict__ keep_cnt, int ncat, int K,
                                                   float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                   int* __restrict__ out_entry, int* __restrict__ out_count,
                                                   uint32_t* __restrict__ zero_word) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);      // [ncat][NMS_MAX]
    const int b = blockIdx.x, c = blockIdx.y, tid = threadIdx.x;
    if (zero_word && c == 0 && tid == 0) zero_word[b] = 0u;      // next stage's max-coordinate cell
    __shared__ int cnt[8];
    boxes += (size_t)b * n_total * 4;
    scores += (size_t)b * n_total;
    if (tid < ncat) cnt[tid] = keep_cnt[b * ncat + tid];
    __syncthreads();
    int total = 0;
    for (int o = 0; o < ncat; ++o) {
        const int* src = keep_idx + ((size_t)b * ncat + o) * NMS_MAX;
        if (tid < cnt[o]) { const int e = src[tid]; keys[o * NMS_MAX + tid] = comp_key(scores[e], (uint32_t)e); }
        total += cnt[o];
    }
    __syncthreads();
    const int nout = total < K ? total : K;
    if (tid < cnt[c]) {
        const uint64_t key = keys[c * NMS_MAX + tid];
        int rank = tid;
        for (int o = 0; o < ncat; ++o) {
            if (o == c) continue;
            int lo = 0, hi = cnt[o];
            const uint64_t* ko = keys + o * NMS_MAX;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (ko[mid] < key) lo = mid + 1; else hi = mid; }
            rank += lo;
        }
        if (rank < K) {
            const uint32_t e = (uint32_t)key;
            float* ob = out_boxes + ((size_t)b * K + rank) * 4;
            ob[0] = boxes[e * 4 + 0]; ob[1] = boxes[e * 4 + 1]; ob[2] = boxes[e * 4 + 2]; ob[3] = boxes[e * 4 + 3];
            out_scores[(size_t)b * K + rank] = scores[e];
            out_entry[(size_t)b * K + rank] = (int)e;
        }
    }
    if (c != 0) return;
    for (int i = nout + tid; i < K; i += 1024) {
        float* ob = out_boxes + ((size_t)b * K + i) * 4;
        ob[0] = ob[1] = ob[2] = ob[3] = 0.f;
        out_scores[(size_t)b * K + i] = 0.f;
        out_entry[(size_t)b * K + i] = -1;
    }
    if (tid == 0) out_count[b] = nout;
}

// ---------------------------------------------------------------- box head post-processing
// One thread per ROI: softmax over K+1 logits, per-class box decoding (weights wx,wy,ww,wh), clip,
// score filter.  pred: [B*P][ld] with logits at [0..K] (background last) and deltas at [K+1 ..].
// Candidate slot = roi*K + class (== torch nonzero() order).
__global__ __launch_bounds__(256) void box_candidates(const float* __restrict__ pred, int ld, int K,
                                                      const float* __restrict__ props, const int* __restrict__ prop_cnt,
                                                      int P, float img_h, float img_w, float thresh, float wx, float wy,
                                                      float ww, float wh, float scale_clamp, float* __restrict__ cboxes,
                                                      float* __restrict__ cscores, int* __restrict__ cvalid,
                                                      uint32_t* __restrict__ maxc, float* __restrict__ probs_out) {
    const int b = blockIdx.y;
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= P) return;
    const size_t roi = (size_t)b * P + r;
    const bool live = r < prop_cnt[b];
    const float* lg = pred + roi * ld;
    float mx = lg[0];
    for (int k = 1; k <= K; ++k) mx = fmaxf(mx, lg[k]);
    float ex[8], sum = 0.f;
    for (int k = 0; k <= K; ++k) { ex[k] = expf(lg[k] - mx); sum += ex[k]; }
    const float* pb = props + roi * 4;
    const float w = pb[2] - pb[0], h = pb[3] - pb[1];
    const float cx = pb[0] + 0.5f * w, cy = pb[1] + 0.5f * h;
    for (int k = 0; k < K; ++k) {
        const size_t slot = roi * K + k;
        const float p = ex[k] / sum;
        if (probs_out) probs_out[roi * (K + 1) + k] = p;
        const float* d = lg + (K + 1) + 4 * k;
        float dx = d[0] / wx, dy = d[1] / wy, dw = d[2] / ww, dh = d[3] / wh;
        dw = dw > scale_clamp ? scale_clamp : dw;
        dh = dh > scale_clamp ? scale_clamp : dh;
        const float pcx = dx * w + cx, pcy = dy * h + cy;
        const float pw = expf(dw) * w, ph = expf(dh) * h;
        float x0 = pcx - 0.5f * pw, y0 = pcy - 0.5f * ph, x1 = pcx + 0.5f * pw, y1 = pcy + 0.5f * ph;
        x0 = fminf(fmaxf(x0, 0.f), img_w);
        y0 = fminf(fmaxf(y0, 0.f), img_h);
        x1 = fminf(fmaxf(x1, 0.f), img_w);
        y1 = fminf(fmaxf(y1, 0.f), img_h);
        const int ok = live && (p > thresh);
        cboxes[slot * 4 + 0] = x0; cboxes[slot * 4 + 1] = y0; cboxes[slot * 4 + 2] = x1; cboxes[slot * 4 + 3] = y1;
        cscores[slot] = p;
        cvalid[slot] = ok;
        if (ok) atomicMax(maxc + b, __float_as_uint(fmaxf(fmaxf(x0, x1), fmaxf(y0, y1))));
    }
    if (probs_out) probs_out[roi * (K + 1) + K] = ex[K] / sum;
}

// Pack per-image detections into one dense list (image id kept per entry) so the mask tail's
// GEMMs see a contiguous M.  det_* are [B][Kd]; packed_* are [B*Kd].
__global__ void pack_detections(const float* __restrict__ det_boxes, const float* __restrict__ det_scores,
                                const int* __restrict__ det_entry, const int* __restrict__ det_cnt, int B, int Kd,
                                int ncls, float* __restrict__ pk_boxes, float* __restrict__ pk_scores,
                                int* __restrict__ pk_cls, int* __restrict__ pk_img, int* __restrict__ pk_roi,
                                int* __restrict__ pk_total, int* __restrict__ pk_offset,
                                unsigned long long* __restrict__ zero_sums) {
    __shared__ int offs[65];
    // the mask tail's per-detection integer sums (paste_masks adds into them)
    if (zero_sums) for (int i = threadIdx.x; i < B * Kd * 3; i += blockDim.x) zero_sums[i] = 0ull;
    if (threadIdx.x == 0) {
        int t = 0;
        for (int b = 0; b < B; ++b) { offs[b] = t; pk_offset[b] = t; t += det_cnt[b]; }
        offs[B] = t;
        pk_offset[B] = t;
        *pk_total = t;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * Kd; i += blockDim.x) {
        const int b = i / Kd, k = i - b * Kd;
        if (k < det_cnt[b]) {
            const int o = offs[b] + k;
            const float* s = det_boxes + (size_t)i * 4;
            pk_boxes[o * 4 + 0] = s[0]; pk_boxes[o * 4 + 1] = s[1]; pk_boxes[o * 4 + 2] = s[2]; pk_boxes[o * 4 + 3] = s[3];
            pk_scores[o] = det_scores[i];
            const int e = det_entry[i];
            pk_cls[o] = e % ncls;
            pk_roi[o] = e / ncls;
            pk_img[o] = b;
        }
    }
}

extern "C" {
int apse_k_rpn_topk_stage(const RpnLevels* L_dev, const TopkJob* jobs_dev, int njobs, uint64_t* lists, int nslots, int B,
                          uint32_t* zero_word, hipStream_t s) {
    if (njobs <= 0) return APSE_OK;
    hipLaunchKernelGGL(rpn_topk_stage, dim3(njobs, B), dim3(1024), 0, s, L_dev, jobs_dev, lists, nslots, zero_word);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_rpn_decode(const RpnLevels* L_dev, int pre_topk, const uint64_t* lists, int nslots, const int* final_slot_dev,
                      float img_h, float img_w, float scale_clamp, float* boxes, float* scores, int* valid, uint32_t* maxc,
                      int level_mask, int B, hipStream_t s) {
    hipLaunchKernelGGL(rpn_decode, dim3((pre_topk + 255) / 256, 5, B), dim3(256), 0, s, L_dev, lists, nslots, final_slot_dev,
                       img_h, img_w, scale_clamp, boxes, scores, valid, maxc, level_mask);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
size_t apse_nms_scratch_bytes(int slots) {
    return (size_t)slots * (NMS_MAX * 4 + 4 * NMS_MAX * 4 + NMS_MAX * 4 + (size_t)NMS_MAX * 16 * 8 + 16);
}
static NmsScratch carve(void* scratch, int slots) {
    NmsScratch S;
    char* p = reinterpret_cast<char*>(scratch);
    S.mask = reinterpret_cast<uint64_t*>(p); p += (size_t)slots * NMS_MAX * 16 * 8;
    S.box = reinterpret_cast<float*>(p); p += (size_t)slots * 4 * NMS_MAX * 4;
    S.area = reinterpret_cast<float*>(p); p += (size_t)slots * NMS_MAX * 4;
    S.entry = reinterpret_cast<int*>(p); p += (size_t)slots * NMS_MAX * 4;
    S.n = reinterpret_cast<int*>(p);
    return S;
}
int apse_k_nms_percat(const float* boxes, const float* scores, const int* valid, int n_total, int cat_div, int cat_mod,
                      const uint32_t* maxc, float thr, int* keep_idx, int* keep_cnt, int ncat, void* scratch, int cat_shift,
                      int B, int presorted, hipStream_t s) {
    static bool done = false;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&nms_scan), hipFuncAttributeMaxDynamicSharedMemorySize,
                            NMS_MAX * 16 * 8);
        done = true;
    }
    NmsScratch S = carve(scratch, ncat * B);
    hipLaunchKernelGGL(nms_prepare, dim3(ncat, B), dim3(1024), 0, s, boxes, scores, valid, n_total, cat_div, cat_mod, maxc, S,
                       ncat, cat_shift, presorted);
    hipLaunchKernelGGL(nms_matrix, dim3(16, 16, ncat * B), dim3(64), 0, s, S, thr);
    hipLaunchKernelGGL(nms_scan, dim3(ncat, B), dim3(1024), NMS_MAX * 16 * 8, s, S, keep_idx, keep_cnt);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_rank_final(const float* boxes, const float* scores, int n_total, const int* keep_idx, const int* keep_cnt,
                      int ncat, int K, float* out_boxes, float* out_scores, int* out_entry, int* out_count,
                      uint32_t* zero_word, int B, hipStream_t s) {
    static bool done = false;
    if (!done) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_merge), hipFuncAttributeMaxDynamicSharedMemorySize,
                            8 * NMS_MAX * 8);
        done = true;
    }
    if (ncat > 8) return APSE_E_INVALID;
    hipLaunchKernelGGL(rank_merge, dim3(B, ncat), dim3(1024), (size_t)ncat * NMS_MAX * 8, s, boxes, scores, n_total, keep_idx, keep_cnt,
                       ncat, K, out_boxes, out_scores, out_entry, out_count, zero_word);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_box_candidates(const float* pred, int ld, int K, const float* props, const int* prop_cnt, int P, float img_h,
                          float img_w, float thresh, const float* wts, float scale_clamp, float* cboxes, float* cscores,
                          int* cvalid, uint32_t* maxc, float* probs_out, int B, hipStream_t s) {
    if (K > 7) return APSE_E_INVALID;
    hipLaunchKernelGGL(box_candidates, dim3((P + 255) / 256, B), dim3(256), 0, s, pred, ld, K, props, prop_cnt, P, img_h,
                       img_w, thresh, wts[0], wts[1], wts[2], wts[3], scale_clamp, cboxes, cscores, cvalid, maxc, probs_out);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
int apse_k_pack_detections(const float* det_boxes, const float* det_scores, const int* det_entry, const int* det_cnt, int B,
                           int Kd, int ncls, float* pk_boxes, float* pk_scores, int* pk_cls, int* pk_img, int* pk_roi,
                           int* pk_total, int* pk_offset, unsigned long long* zero_sums, hipStream_t s) {
    if (B > 64) return APSE_E_INVALID;
    hipLaunchKernelGGL(pack_detections, dim3(1), dim3(256), 0, s, det_boxes, det_scores, det_entry, det_cnt, B, Kd, ncls,
                       pk_boxes, pk_scores, pk_cls, pk_img, pk_roi, pk_total, pk_offset, zero_sums);
    return hipGetLastError() == hipSuccess ? APSE_OK : APSE_E_HIP;
}
}
