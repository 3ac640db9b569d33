// libapse_hip.so context: weights, buffers, launch plan and the C ABI of include/apse_hip.h.
// Host-side C++ only orchestrates; all arithmetic is in the HIP kernels of this directory.
// One context per device/process rank; the caller's stream carries every launch (no hidden syncs
// except apse_read_results).
#include "apse_common.h"
#include "../../include/apse_hip.h"
#include "preproc_pixel.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

// ---- kernels' extern "C" launchers (defined in the other .hip files)
struct RpnLevel { const float* head; int H, W, stride; int n; int k; float base[3][4]; };
struct RpnLevels { RpnLevel lv[5]; int head_ld; int pre_topk; };
struct TopkJob { int kind; int level; int begin, count; int nsrc; int src[4]; int src_count[4]; int dst; int dst_count; };
struct FpnMaps { const void* p[4]; int H[4], W[4]; float scale[4]; int st; };
struct PasteParams {
    const float* boxes; const int* cls; const int* total; const float* logits; int M, ldc; float sx, sy; int out_h, out_w;
    int words_per_row; float thresh; float* boxes_out; int* valid; int* rect; uint64_t* bits; unsigned long long* sums;
};
extern "C" {
int apse_k_pil_resize(const uint8_t*, uint8_t*, void*, int, uint8_t*, const int*, const int*, int, const int*, const int*, int, int,
                      int, int, int, int, int, int, const float*, const UndistortParams*, const LabTables*, const void*, const int*, int, hipStream_t);
int apse_k_undistort_build_map_compact(const UndistortParams*, void*, int*, hipStream_t);
int apse_k_chw_norm(const float*, void*, int, int, int, int, int, int, const float*, hipStream_t);
int apse_k_maxpool3x3s2(const void*, void*, int, int, int, int, int, hipStream_t);
bool apse_assoc_fc_ok(int K, int N);
int apse_k_assoc_fc(const float*, const float*, const float*, float*, const int*, int, int, int, float*, float*, hipStream_t);
int apse_k_stem_pool16(const void*, const uint16_t*, const float*, void*, int, int, int, int, hipStream_t, hipEvent_t, hipEvent_t);
int apse_k_subsample2(const void*, void*, int, int, int, int, int, hipStream_t);
int apse_k_bottleneck64_fused16(const void*, const void*, void*, const uint16_t*, const float*, const uint16_t*, const float*,
                                const uint16_t*, const float*, int, int, int, int, int, hipStream_t, hipEvent_t, hipEvent_t);
int apse_k_nhwc_to_nchw(const void*, float*, int, int, int, int, hipStream_t);
int apse_k_rpn_topk_stage(const RpnLevels*, const TopkJob*, int, uint64_t*, int, int, uint32_t*, hipStream_t);
int apse_k_rpn_decode(const RpnLevels*, int, const uint64_t*, int, const int*, float, float, float, float*, float*, int*,
                      uint32_t*, int, int, hipStream_t);
int apse_k_nms_percat(const float*, const float*, const int*, int, int, int, const uint32_t*, float, int*, int*, int,
                      void*, int, int, int, hipStream_t);
size_t apse_nms_scratch_bytes(int slots);
int apse_k_rank_final(const float*, const float*, int, const int*, const int*, int, int, float*, float*, int*, int*, uint32_t*,
                      int, hipStream_t);
int apse_k_box_candidates(const float*, int, int, const float*, const int*, int, float, float, float, const float*, float,
                          float*, float*, int*, uint32_t*, float*, int, hipStream_t);
int apse_k_pack_detections(const float*, const float*, const int*, const int*, int, int, int, float*, float*, int*, int*,
                           int*, int*, int*, unsigned long long*, hipStream_t);
int apse_k_roi_align(const FpnMaps*, const float*, const int*, const int*, const int*, int, int, int, void*, int, hipStream_t);
int apse_k_roi_pool(const void*, int, int, int, const float*, const int*, const int*, int, int, float, float*, int, int, hipStream_t);
int apse_k_mask_resize(const uint8_t*, int, int, int, int, int, float*, hipStream_t);
int apse_k_round16(const float*, uint16_t*, size_t, int, hipStream_t);
int apse_k_roi_align_masked(const void*, int, int, int, int, const float*, const float*, int, int, int, float, float*, hipStream_t);
int apse_k_l2_normalize(const float*, float*, int, const int*, int, hipStream_t);
int apse_k_sqdist(const float*, const float*, int, int, int, float*, hipStream_t);
int apse_k_mask_paste(const PasteParams*, int, unsigned long long*, int, hipStream_t);
int apse_k_closest_points(const uint64_t*, const int*, const int*, const unsigned long long*, const int*, const int*, const int*, int,
                          int, int, int, int, int*, int*, unsigned long long*, int, hipStream_t);
int apse_k_closest_single(const uint64_t*, int, int, int, float, float, unsigned long long*, hipStream_t);
int apse_k_undistort_gamma(const UndistortParams*, const uint8_t*, uint8_t*, const LabTables*, int, hipStream_t);
int apse_k_bits_to_dense(const uint64_t*, const int*, int, int, int, uint8_t*, hipStream_t);
int apse_k_copy_mask_windows(const uint64_t*, uint64_t*, int, const long long*, const long long*, const int*, const int*, int, hipStream_t);
int apse_k_dense_to_bits(const uint8_t*, int, int, int, uint64_t*, unsigned long long*, hipStream_t);
}

static std::string g_create_error;
#define NMS_SLOT 1024
#define APSE_EV_HALF 1024    // HIP events per half of the profiling pool (one pair per timed launch)
#define APSE_EXPECTED_DETS 8      // list length the packed-list GEMMs are shaped for (static: see add_conv)

struct HostW { std::vector<float> v; std::vector<int64_t> shape; };
struct Tens { float* p = nullptr; int H = 0, W = 0, C = 0; int st = 0; };   // per-item NHWC dims; st: 0 f32, 1 bf16, 2 f16 storage

struct ConvStep {
    ConvParams p;          // B/M filled at launch
    int b_mult = 1;        // items per image (1, post_topk, dets_per_image)
    int fixed_items = 0;   // > 0: the launch always covers this many items (a tensor laid out for max_batch: merged RPN head)
    int cfg = 0;
    double flops_per_item = 0;   // algorithmic 2*MACs per item (one image / one roi / one detection)
    int count_kind = 0;    // 0 none, 1 prop_cnt[0] (batch 1 only), 2 packed total
    void* pool_y = nullptr;   // != nullptr: the stem of the 16-bit modes, run as stem_s2d_pool16 (conv + ReLU + 3x3/2 max-pool) into this map
    std::string name;
};
enum StepKind { S_CONV, S_MAXPOOL, S_SUBSAMPLE, S_BNECK };
// S_BNECK: a whole 64-channel bottleneck as one launch (bottleneck16.hip); c = its conv1 (name, flops of all three), c2 / c3 the others
struct Step { StepKind kind; ConvStep c; const float* x; float* y; int H, W, C; int st = 0; ConvParams p2, p3; };

struct apse_ctx {
    apse_config cfg;
    std::string err;
    std::map<std::string, HostW> hw;
    bool finalized = false;
    int PH = 0, PW = 0;
    std::vector<void*> allocs;
    std::map<std::string, Tens> t;
    std::vector<Step> backbone, rpnhead, boxhead, maskhead, embedfc;
    float* ws = nullptr; size_t ws_floats = 0; int* tile_cnt = nullptr;
    // resize tables
    int *hb = nullptr, *hc = nullptr, *vb = nullptr, *vc = nullptr; int hk = 0, vk = 0; uint8_t* rs_tmp = nullptr;
    int rs_pitch = 0;
    int* hcT = nullptr;                                  // horizontal taps tap-major [8][image_w], zero past a pixel's count (hk <= 8)
    // rpn
    RpnLevels rl_host; RpnLevels* rl_dev = nullptr;
    std::vector<std::vector<TopkJob>> stages; std::vector<TopkJob*> stage_dev; int nslots = 0; uint64_t* lists = nullptr;
    int final_slot_host[5]; int* final_slot_dev = nullptr;
    float *dec_boxes = nullptr, *dec_scores = nullptr; int* dec_valid = nullptr; uint32_t* maxc = nullptr;   // maxc[2*B]: rpn, box
    int *keep_idx = nullptr, *keep_cnt = nullptr; void* nms_scratch = nullptr;
    float *props = nullptr, *prop_scores = nullptr; int *prop_entry = nullptr;
    // box head
    FpnMaps fm;
    float *cand_boxes = nullptr, *cand_scores = nullptr, *probs = nullptr; int* cand_valid = nullptr;
    float *det_boxes = nullptr, *det_scores = nullptr; int *det_entry = nullptr, *det_cnt = nullptr;
    // results block (device) and layout
    apse_results_layout lay; uint8_t* res = nullptr;
    // mask tail
    uint64_t* bits2[2] = {nullptr, nullptr}; int bits_cur = 0, bits_read = 0;   // mask bit planes, alternating per forward (see apse_mask_tail)
    unsigned long long* sums = nullptr; int wpr = 0; bool sums_dirty = false;
    float* emb_raw = nullptr;
    float* ws_assoc = nullptr;      // [K / 128][max detections][embed_dim]: K slices of the association FC (apse_k_assoc_fc), or nullptr
    float* rf_mask = nullptr; size_t rf_mask_floats = 0;      // apse_roi_features: masks at p2 resolution (grown on demand)
    bool box_maxc_clean = false;
    UndistortParams cam; bool cam_on = false; LabTables* cam_lut = nullptr; void* cam_map = nullptr; bool cam_map_ok = false;     // apse_set_camera: fused undistort + gamma in apse_preprocess_frames
    int hint_total = 8;      // detections seen in the previous forward: sizes the GRID of the packed-list GEMMs, nothing else
    hipEvent_t read_ev = nullptr; void* read_pending = nullptr;      // apse_read_results_begin / _end
    // apse_set_detections: two pinned staging blocks, each guarded by the event behind its H2D copies, so the call only enqueues
    // (a sequence driver puts the next given-boxes forward behind apse_read_results_begin like any other forward)
    uint8_t* given_host[2] = {nullptr, nullptr}; hipEvent_t given_ev[2] = {nullptr, nullptr}; int given_k = 0;
    // per-kernel profiling with HIP events on the caller's stream (bench.py roofline)
    bool prof_on = false; std::vector<hipEvent_t> ev_pool; int ev_used = 0;
    struct Pending { int cfg; double flops_per_item; int count_kind; int b_mult; int batch; int e0, e1; };
    std::vector<Pending> pending; double prof[APSE_NCFG][3] = {{0}};
    // the event pool has two halves: apse_read_results_begin hands the half (and the pending list) of the forward it reads to _end
    // and switches recording to the other half, so a forward enqueued between the two halves of a read keeps its own events
    int ev_base = 0, cal_read = -1; std::vector<Pending> pending_read;
    // stateless-op scratch
    uint64_t* op_bits = nullptr; unsigned long long* op_sums = nullptr; size_t op_bits_words = 0;
};

static int fail(apse_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
#define HIPCHK(c, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) return fail(c, APSE_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <typename T>
static T* dalloc(apse_ctx* c, size_t n, bool zero = true) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) > 0 ? n * sizeof(T) : 16) != hipSuccess) return nullptr;
    if (zero) hipMemset(p, 0, n * sizeof(T));
    c->allocs.push_back(p);
    return reinterpret_cast<T*>(p);
}
template <typename T>
static T* dupload(apse_ctx* c, const std::vector<T>& v) {
    T* p = dalloc<T>(c, v.size(), false);
    if (p && !v.empty()) hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    return p;
}

// ------------------------------------------------------------------------------------------------
// weight packing: OIHW (+ per-channel scale) -> [Cout_p][KH][KWCp], run = (kw, cin_p)
static void pack_oihw(const float* w, int Cout, int Cin, int KH, int KW, int cin_p, const float* scale, float* out,
                      int KWCp) {
    for (int o = 0; o < Cout; ++o) {
        const float sc = scale ? scale[o] : 1.0f;
        for (int r = 0; r < KH; ++r)
            for (int s = 0; s < KW; ++s)
                for (int ci = 0; ci < Cin; ++ci)
                    out[((size_t)o * KH + r) * KWCp + s * cin_p + ci] = w[(((size_t)o * Cin + ci) * KH + r) * KW + s] * sc;
    }
}
static int pow2_at_least(int v) { int p = 4; while (p < v) p <<= 1; return p; }

static const HostW* getw(apse_ctx* c, const std::string& n) {
    auto it = c->hw.find(n);
    return it == c->hw.end() ? nullptr : &it->second;
}

static Tens make_t(apse_ctx* c, const std::string& name, int items, int H, int W, int C, int st = 0) {
    Tens t;
    t.H = H; t.W = W; t.C = C; t.st = st;
    const size_t elems = (size_t)items * H * W * C;
    t.p = dalloc<float>(c, st ? (elems + 1) / 2 : elems);          // 16-bit storage: half the bytes
    c->t[name] = t;
    return t;
}

// Build one convolution step from reference weights `wname` (OIHW) with optional FrozenBN `wname.norm.*`.
// extra rows (fused heads) can be appended through `more`.
struct ConvSpec {
    std::string name; std::vector<std::string> wnames;   // one or more OIHW weights concatenated along Cout
    int KH, KW, stride, pad, relu;
    int fc_h = 0, fc_w = 0;   // >0: weight is [Cout][C*fc_h*fc_w] flattened (c,h,w): treat as fc_h x fc_w valid conv
    int deconv = 0;
    int s2d = 0;              // stem on the space-to-depth(2) input: the 7x7 / stride-2 filter is re-indexed as 4x4 / stride-1 over 12 channels
};

// 16-bit storage mode: every activation the bulk GEMMs produce lives in HBM in the operand type; the narrow
// decision heads (Cout <= 32) and the association FC keep f32 outputs.
// f32 -> bf16 (dtype 1) / f16 (dtype 2) bits, round-to-nearest-even like the in-kernel converts; a NaN stays a NaN
static uint16_t round16(float v, int dtype) {
    if (dtype == 2) {
        const _Float16 hval = (_Float16)v;
        uint16_t r;
        memcpy(&r, &hval, 2);
        return r;
    }
    uint32_t b;
    memcpy(&b, &v, 4);
    if ((b & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((b >> 16) | 0x40);
    return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}
static int storage_type(const apse_ctx* c) { return (c->cfg.compute_dtype >= 1 && c->cfg.storage16) ? c->cfg.compute_dtype : 0; }

static int add_conv(apse_ctx* c, std::vector<Step>& plan, const ConvSpec& sp, const Tens& in, int in_items_mult, Tens* out,
                    const std::string& out_name, const Tens* res, int res_mode, int y_ld_override, int count_kind,
                    float* out_ptr_override = nullptr, const Tens* out_view = nullptr) {
    // gather rows
    std::vector<float> rows;    // OIHW concatenated
    std::vector<float> bias;
    int Cout = 0, Cin = 0;
    const int KH = sp.KH, KW = sp.KW;
    for (const auto& wn : sp.wnames) {
        const HostW* w = getw(c, wn + ".weight");
        if (!w) return fail(c, APSE_E_MISSING, "missing weight " + wn + ".weight");
        int co, ci;
        std::vector<float> oihw;
        if (sp.fc_h > 0) {
            co = (int)w->shape[0];
            const int flat = (int)w->shape[1];
            ci = flat / (sp.fc_h * sp.fc_w);
            oihw = w->v;                      // [co][c][h][w] already (c,h,w) flattened == OIHW
        } else if (sp.deconv) {
            // ConvTranspose2d weight [Cin][Cout][2][2] -> rows n = (dy*2+dx)*Cout + co, K = ci (1x1)
            ci = (int)w->shape[0];
            const int cc = (int)w->shape[1];
            co = 4 * cc;
            oihw.assign((size_t)co * ci, 0.f);
            for (int i = 0; i < ci; ++i)
                for (int o = 0; o < cc; ++o)
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx)
                            oihw[((size_t)((dy * 2 + dx) * cc + o)) * ci + i] = w->v[(((size_t)i * cc + o) * 2 + dy) * 2 + dx];
        } else if (w->shape.size() == 2) {
            co = (int)w->shape[0]; ci = (int)w->shape[1]; oihw = w->v;
        } else if (sp.s2d) {
            // out(oy, ox) = sum w7[ky][kx][c] x[2 oy - 3 + ky][2 ox - 3 + kx][c]; with input row 2 Y + dy, Y = oy - 2 + r (r = 0..3):
            // ky = 2 r + dy - 1, kx = 2 s + dx - 1 (taps outside 0..6 do not exist: zero), channel (2 dy + dx) 3 + c
            if (w->shape.size() != 4 || w->shape[2] != 7 || w->shape[3] != 7 || w->shape[1] != 3 || KH != 4 || KW != 4)
                return fail(c, APSE_E_INVALID, "space-to-depth stem needs a 7x7 filter over 3 channels: " + wn);
            co = (int)w->shape[0]; ci = 12;
            oihw.assign((size_t)co * 12 * 16, 0.f);
            for (int o = 0; o < co; ++o)
                for (int ch = 0; ch < 3; ++ch)
                    for (int r = 0; r < 4; ++r)
                        for (int sx = 0; sx < 4; ++sx)
                            for (int dy = 0; dy < 2; ++dy)
                                for (int dx = 0; dx < 2; ++dx) {
                                    const int ky = 2 * r + dy - 1, kx = 2 * sx + dx - 1;
                                    if (ky < 0 || ky > 6 || kx < 0 || kx > 6) continue;
                                    oihw[(((size_t)o * 12 + (dy * 2 + dx) * 3 + ch) * 4 + r) * 4 + sx] = w->v[(((size_t)o * 3 + ch) * 7 + ky) * 7 + kx];
                                }
        } else {
            co = (int)w->shape[0]; ci = (int)w->shape[1]; oihw = w->v;
            if ((int)w->shape[2] != KH || (int)w->shape[3] != KW) return fail(c, APSE_E_INVALID, "kernel size mismatch " + wn);
        }
        if (Cin && ci != Cin) return fail(c, APSE_E_INVALID, "Cin mismatch in fused conv " + sp.name);
        Cin = ci;
        // FrozenBN fold (detectron2 FrozenBatchNorm2d, eps 1e-5): scale = g * rsqrt(var + eps), bias = b - mean*scale
        const HostW* g = getw(c, wn + ".norm.weight");
        std::vector<float> scale;
        const int nb = sp.deconv ? co / 4 : co;
        std::vector<float> b(nb, 0.f);
        if (g) {
            const HostW* be = getw(c, wn + ".norm.bias");
            const HostW* mu = getw(c, wn + ".norm.running_mean");
            const HostW* var = getw(c, wn + ".norm.running_var");
            if (!be || !mu || !var) return fail(c, APSE_E_MISSING, "incomplete norm for " + wn);
            scale.resize(co);
            for (int o = 0; o < co; ++o) {
                scale[o] = g->v[o] * (1.0f / sqrtf(var->v[o] + 1e-5f));
                b[o] = be->v[o] - mu->v[o] * scale[o];
            }
            const size_t per = (size_t)ci * KH * KW;
            for (int o = 0; o < co; ++o)
                for (size_t k = 0; k < per; ++k) oihw[(size_t)o * per + k] *= scale[o];
        } else {
            const HostW* bw = getw(c, wn + ".bias");
            if (bw) for (int o = 0; o < nb; ++o) b[o] = bw->v[o];
        }
        rows.insert(rows.end(), oihw.begin(), oihw.end());
        bias.insert(bias.end(), b.begin(), b.end());
        Cout += co;
    }
    const int cin_p = in.C;
    if (pow2_at_least(Cin) != cin_p && Cin != cin_p) return fail(c, APSE_E_INVALID, "input channels mismatch at " + sp.name);
    if (sp.s2d && (in.C != 16 || sp.stride != 1 || sp.pad != 2)) return fail(c, APSE_E_INVALID, "space-to-depth stem geometry");
    const int KWC = KW * cin_p, KWCp = apse_roundup(KWC, 32);
    const int Cout_p = apse_roundup(Cout, 128);
    std::vector<float> packed((size_t)Cout_p * KH * KWCp, 0.f);
    pack_oihw(rows.data(), Cout, Cin, KH, KW, cin_p, nullptr, packed.data(), KWCp);
    std::vector<float> bias_p(Cout_p, 0.f);
    for (size_t i = 0; i < bias.size(); ++i) bias_p[i] = bias[i];
    const bool use_bf16 = (c->cfg.compute_dtype >= 1 && Cout > 32 && sp.name != "assoc_fc");      // bf16 or f16 operands
    float* wd = nullptr;
    uint16_t* wd16 = nullptr;
    if (use_bf16) {
        // filters pre-rounded to the 16-bit operand type (round-to-nearest-even, as the in-kernel converts do)
        std::vector<uint16_t> p16(packed.size());
        for (size_t i = 0; i < packed.size(); ++i) p16[i] = round16(packed[i], c->cfg.compute_dtype);
        wd16 = dupload(c, p16);
    } else {
        wd = dupload(c, packed);
    }
    float* bd = dupload(c, bias_p);
    if ((!wd && !wd16) || !bd) return fail(c, APSE_E_NOMEM, "weight upload failed at " + sp.name);

    Step st;
    st.kind = S_CONV;
    ConvStep& cs = st.c;
    memset(&cs.p, 0, sizeof(cs.p));
    cs.name = sp.name;
    cs.b_mult = in_items_mult;
    cs.count_kind = count_kind;
    ConvParams& p = cs.p;
    p.x = in.p; p.w = wd; p.w16 = wd16; p.bias = bd; p.res = res ? res->p : nullptr; p.res_mode = res_mode;
    p.x_st = in.st; p.res_st = res ? res->st : 0;
    p.H = in.H; p.W = in.W; p.cin_log2 = apse_ilog2(cin_p);
    p.KH = KH; p.KW = KW; p.stride = sp.stride; p.pad = sp.pad; p.KWCp = KWCp;
    p.OH = (in.H + 2 * sp.pad - KH) / sp.stride + 1;
    p.OW = (in.W + 2 * sp.pad - KW) / sp.stride + 1;
    if (sp.s2d) { p.OH = in.H; p.OW = in.W; }       // pad 2 above / left, 1 below / right: taps past the map read zeros (range check)
    p.Cout = Cout; p.relu = sp.relu;
    p.steps_total = KH * (KWCp / 32);
    p.splitk = 1;
    p.out_mode = sp.deconv ? 1 : 0;
    // bf16 matrix cores for the bulk GEMMs; decision layers (narrow heads) and the association FC stay exact f32
    p.prec = use_bf16 ? c->cfg.compute_dtype : 0;
    p.cdec = sp.deconv ? Cout / 4 : 0;
    const int out_c = sp.deconv ? Cout / 4 : (y_ld_override > 0 ? y_ld_override : Cout);
    const int oh = sp.deconv ? 2 * p.OH : p.OH, ow = sp.deconv ? 2 * p.OW : p.OW;
    const int out_st = out_view ? out_view->st : ((use_bf16 && !out_ptr_override) ? storage_type(c) : 0);
    if (out_st && (out_c & 7)) return fail(c, APSE_E_INVALID, "16-bit tensors need C % 8 == 0 at " + sp.name);
    Tens o;
    if (out_view) { o = *out_view; o.H = oh; o.W = ow; o.C = out_c; c->t[out_name] = o; }      // a slice of a larger allocation
    else if (out_ptr_override) { o.p = out_ptr_override; o.H = oh; o.W = ow; o.C = out_c; }
    else o = make_t(c, out_name, c->cfg.max_batch * in_items_mult, oh, ow, out_c, out_st);
    p.y_st = o.st;
    if (!o.p) return fail(c, APSE_E_NOMEM, "activation alloc failed at " + sp.name);
    p.y = o.p; p.y_ld = out_c; p.y_coff = 0;
    cs.flops_per_item = sp.s2d ? 2.0 * p.OH * p.OW * (double)Cout * 7 * 7 * 3        // algorithmic: the reference's 7x7x3 taps
                               : 2.0 * p.OH * p.OW * (double)Cout * KH * KW * Cin;
    // tile config / split-K chosen for the full batch; workspace sized for the worst case over 1..max_batch
    const int Mfull = c->cfg.max_batch * in_items_mult * p.OH * p.OW;
    int sk = 1;
    cs.cfg = apse_conv_pick_cfg(Mfull, Cout, p.steps_total, &sk);
    if (count_kind == 2) {
        // GEMMs over the packed detection list: the tile shape and the K split fix the f32 summation order, so they are
        // chosen HERE, once, from plan constants only (a typical list of APSE_EXPECTED_DETS detections per image of the
        // context's max_batch -- like every other layer's shape; never from the batch of a forward or an earlier frame's
        // count).  Within a context a frame's masks / embeddings are then the same bits whatever ran before it, in whatever
        // batch; contexts with equal configuration (shards, pipeline slots) agree with each other.  The live count only
        // sizes the grid (m_hint).
        const int kd = c->cfg.dets_per_image < APSE_EXPECTED_DETS ? c->cfg.dets_per_image : APSE_EXPECTED_DETS;
        const int rows = kd * c->cfg.max_batch * p.OH * p.OW;
        sk = 1;
        cs.cfg = apse_conv_pick_cfg(rows < Mfull ? rows : Mfull, Cout, p.steps_total, &sk);
    }
    p.splitk = sk;
    if (sk > 1) {
        const size_t need = (size_t)sk * Mfull * Cout;
        if (need > c->ws_floats) c->ws_floats = need;
    }
    if (out) *out = o;
    plan.push_back(st);
    return APSE_OK;
}

static int run_plan(apse_ctx* c, std::vector<Step>& plan, int batch, hipStream_t s) {
    int* total_dev = reinterpret_cast<int*>(c->res + c->lay.total);
    int* propcnt_dev = reinterpret_cast<int*>(c->res + c->lay.prop_count);
    for (size_t si = 0; si < plan.size(); ++si) {
        Step& st = plan[si];
        int rc = APSE_OK;
        if (st.kind == S_CONV) {
            ConvParams p = st.c.p;
            // next convolution of this plan: its filters are prefetched by this launch
            for (size_t sj = si + 1; sj < plan.size() && sj <= si + 2; ++sj)
                if (plan[sj].kind == S_CONV) {
                    const ConvParams& q = plan[sj].c.p;
                    const size_t elems = (size_t)apse_roundup(q.Cout, 128) * q.KH * q.KWCp;
                    p.next_w = q.w16 ? (const void*)q.w16 : (const void*)q.w;
                    const size_t bytes = elems * (q.w16 ? 2 : 4);
                    p.next_w_bytes = bytes > (64u << 20) ? (64u << 20) : (unsigned)bytes;
                    break;
                }
            p.B = st.c.fixed_items > 0 ? st.c.fixed_items : batch * st.c.b_mult;
            p.M = p.B * p.OH * p.OW;
            p.ws = c->ws;
            // In-launch split-K reduction (last arriver) measured SLOWER here than the separate reduce kernel
            // (f32 132 -> 111 FPS): 64-512 KB of slabs per tile and an agent-scope release (L2 write-back) per
            // block; it stays available through apse_conv_desc.fuse_reduce for small slabs.
            p.tile_cnt = nullptr;
            p.m_count = nullptr; p.m_per_item = p.OH * p.OW; p.m_hint = 0;
            int cfg = st.c.cfg;
            if (st.c.count_kind == 2) {
                p.m_count = total_dev;
                // the previous forward's count sizes the GRID only (blocks are persistent over the live tiles, so any
                // grid computes every tile the same way); tile shape and K split are plan constants (add_conv)
                const int mh = (c->hint_total > 0 ? c->hint_total : 1) * p.m_per_item;
                p.m_hint = mh < p.M ? mh : p.M;
                // 16-bit modes, unsplit layers (round 4): the plan shapes these GEMMs for 8 detections per image; a frame with 40
                // has five times the rows, and 64x64 tiles then run at half the rate of 128x128 ones.  The single-k-group 4-wave
                // tiles (128x128, 64x64, 128x64) add every accumulator's products in the same ascending-k order -- the same bits
                // (tests/test_gpu_ops.py::test_conv2d_16bit_tiles_are_bit_identical) -- so the tile may follow the hinted row count
                // without a frame's results depending on what ran before it.  f32 (two-k-group shapes, split K) keeps the plan's.
                if (p.prec != 0 && p.splitk == 1 && (cfg == 0 || cfg == 1 || cfg == 3)) {
                    int sk_dyn = 1;
                    const int cfg_dyn = apse_conv_pick_cfg(p.m_hint, p.Cout, p.steps_total, &sk_dyn);
                    if (sk_dyn == 1 && (cfg_dyn == 0 || cfg_dyn == 1 || cfg_dyn == 3)) cfg = cfg_dyn;
                }
            }
            else if (st.c.count_kind == 1 && batch == 1) p.m_count = propcnt_dev;
            int e0 = -1;
            if (c->prof_on && c->ev_used == 0) {
                // calibration pair: two back-to-back records; their elapsed time (the marker overhead a timed
                // kernel also pays) is subtracted from every measurement of this forward
                hipEventRecord(c->ev_pool[c->ev_base], s);
                hipEventRecord(c->ev_pool[c->ev_base + 1], s);
                c->ev_used = 2;
            }
            if (c->prof_on && c->ev_used + 2 <= APSE_EV_HALF) { e0 = c->ev_base + c->ev_used; c->ev_used += 2; }
            if (st.c.pool_y) {
                rc = apse_k_stem_pool16(p.x, p.w16, p.bias, st.c.pool_y, batch, p.H, p.W, p.prec, s, e0 >= 0 ? c->ev_pool[e0] : nullptr,
                                        e0 >= 0 ? c->ev_pool[e0 + 1] : nullptr);
                cfg = APSE_CFG_STEMPOOL;
            } else {
                rc = apse_launch_conv(p, cfg, s, e0 >= 0 ? c->ev_pool[e0] : nullptr, e0 >= 0 ? c->ev_pool[e0 + 1] : nullptr);
                cfg = apse_conv_effective_cfg(p, cfg);                        // profile label of the kernel that actually ran
            }
            if (e0 >= 0) c->pending.push_back({cfg, st.c.flops_per_item, st.c.count_kind, st.c.b_mult, batch, e0, e0 + 1});
        } else if (st.kind == S_BNECK) {
            int e0 = -1;
            if (c->prof_on && c->ev_used == 0) {
                hipEventRecord(c->ev_pool[c->ev_base], s);
                hipEventRecord(c->ev_pool[c->ev_base + 1], s);
                c->ev_used = 2;
            }
            if (c->prof_on && c->ev_used + 2 <= APSE_EV_HALF) { e0 = c->ev_base + c->ev_used; c->ev_used += 2; }
            const ConvParams& q1 = st.c.p;
            rc = apse_k_bottleneck64_fused16(st.x, st.p3.res, st.y, q1.w16, q1.bias, st.p2.w16, st.p2.bias, st.p3.w16, st.p3.bias, batch,
                                             st.H, st.W, st.C, st.st, s, e0 >= 0 ? c->ev_pool[e0] : nullptr, e0 >= 0 ? c->ev_pool[e0 + 1] : nullptr);
            if (e0 >= 0) c->pending.push_back({APSE_CFG_BNECK, st.c.flops_per_item, 0, 1, batch, e0, e0 + 1});
        } else if (st.kind == S_MAXPOOL) {
            rc = apse_k_maxpool3x3s2(st.x, st.y, batch, st.H, st.W, st.C, st.st, s);
        } else {
            rc = apse_k_subsample2(st.x, st.y, batch, st.H, st.W, st.C, st.st, s);
        }
        if (rc != APSE_OK) return fail(c, rc, "launch failed at step " + st.c.name);
    }
    return APSE_OK;
}

// ------------------------------------------------------------------------------------------------
static void layout_results(apse_ctx* c) {
    apse_results_layout& L = c->lay;
    memset(&L, 0, sizeof(L));
    const int B = c->cfg.max_batch, kd = c->cfg.dets_per_image, n = B * kd, E = c->cfg.embed_dim;
    L.n_max = n; L.dets_per_image = kd; L.embed_dim = E; L.max_batch = B;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t r = o; o += (bytes + 15) / 16 * 16; return r; };
    L.total = take(4);
    L.offset = take(4 * (B + 1));
    L.prop_count = take(4 * B);
    L.img = take(4 * n); L.cls = take(4 * n); L.roi = take(4 * n); L.score = take(4 * n);
    L.box_resized = take(16 * n); L.box = take(16 * n); L.valid = take(4 * n); L.rect = take(16 * n);
    L.mass = take(4 * n); L.centroid = take(8 * n);
    L.closest = take((size_t)8 * n * kd);
    L.embedding = take((size_t)4 * n * E);
    L.bytes = o;
}

static int build_plan(apse_ctx* c) {
    const apse_config& g = c->cfg;
    const int B = g.max_batch;
    c->PH = apse_roundup(g.image_h, 32);
    c->PW = apse_roundup(g.image_w, 32);
    layout_results(c);
    c->res = dalloc<uint8_t>(c, c->lay.bytes);
    if (!c->res) return fail(c, APSE_E_NOMEM, "results alloc");
    int rc;
    // ---- backbone
    Tens cur;
    bool fused_stem = false;
    if (storage_type(c)) {
        // 16-bit storage modes: space-to-depth(2) input (elementwise.hip, input_store) and the stem as a 4x4 / stride-1 convolution
        // over 16 channels: one 64-element k-step per filter row on the scheduled 16-bit kernel (K = 256 instead of the 448 a
        // 7-pixel x 8-channel run would pad to; the f32-input stem ran on the legacy conditional-load kernel at 428 us per batch 8)
        Tens x0 = make_t(c, "input", B, c->PH / 2, c->PW / 2, 16, storage_type(c));
        ConvSpec sp{"stem.conv1", {"backbone.bottom_up.stem.conv1"}, 4, 4, 1, 2, 1};
        sp.s2d = 1;
        // ... and the max-pool behind it in the same kernel (stem_pool16.hip): the stem output never goes to HBM.
        // APSE_NO_STEM_FUSE (read when the context is built): the two-kernel form, for the equality test and A/B runs.
        if (!getenv("APSE_NO_STEM_FUSE")) {
            Tens pooled = make_t(c, "stem", B, (x0.H + 2 - 3) / 2 + 1, (x0.W + 2 - 3) / 2 + 1, 64, storage_type(c));
            if (!pooled.p) return fail(c, APSE_E_NOMEM, "stem alloc");
            Tens unused;
            rc = add_conv(c, c->backbone, sp, x0, 1, &unused, "stem.conv1", nullptr, 0, 0, 0, nullptr, &pooled);
            if (rc) return rc;
            c->t.erase("stem.conv1");                       // no such tensor in this form
            c->backbone.back().c.pool_y = pooled.p;
            cur = pooled;
            fused_stem = true;
        } else {
            rc = add_conv(c, c->backbone, sp, x0, 1, &cur, "stem.conv1", nullptr, 0, 0, 0);
        }
    } else {
        Tens x0 = make_t(c, "input", B, c->PH, c->PW, 4);
        rc = add_conv(c, c->backbone, ConvSpec{"stem.conv1", {"backbone.bottom_up.stem.conv1"}, 7, 7, 2, 3, 1}, x0, 1, &cur,
                      "stem.conv1", nullptr, 0, 0, 0);
    }
    if (rc) return rc;
    if (!fused_stem) {
        Step st; st.kind = S_MAXPOOL; st.x = cur.p; st.H = cur.H; st.W = cur.W; st.C = cur.C;
        Tens o = make_t(c, "stem", B, (cur.H + 2 - 3) / 2 + 1, (cur.W + 2 - 3) / 2 + 1, cur.C, cur.st);
        st.y = o.p; st.c.name = "stem.pool"; st.st = cur.st;
        c->backbone.push_back(st);
        cur = o;
    }
    for (int si = 0; si < 4; ++si) {
        char stage[16];
        snprintf(stage, sizeof stage, "res%d", si + 2);
        for (int bi = 0; bi < g.blocks[si]; ++bi) {
            char pre[96];
            snprintf(pre, sizeof pre, "backbone.bottom_up.%s.%d", stage, bi);
            const std::string P = pre;
            const int stride = (bi == 0 && si > 0) ? 2 : 1;
            Tens a, b2, sc, out;
            const Tens* resp = &cur;
            if (getw(c, P + ".shortcut.weight")) {
                rc = add_conv(c, c->backbone, ConvSpec{P + ".shortcut", {P + ".shortcut"}, 1, 1, stride, 0, 0}, cur, 1, &sc,
                              P + ".shortcut", nullptr, 0, 0, 0);
                if (rc) return rc;
                resp = &sc;
            }
            rc = add_conv(c, c->backbone, ConvSpec{P + ".conv1", {P + ".conv1"}, 1, 1, stride, 0, 1}, cur, 1, &a, P + ".conv1",
                          nullptr, 0, 0, 0);
            if (rc) return rc;
            rc = add_conv(c, c->backbone, ConvSpec{P + ".conv2", {P + ".conv2"}, 3, 3, 1, 1, 1}, a, 1, &b2, P + ".conv2",
                          nullptr, 0, 0, 0);
            if (rc) return rc;
            const bool last = (bi == g.blocks[si] - 1);
            rc = add_conv(c, c->backbone, ConvSpec{P + ".conv3", {P + ".conv3"}, 1, 1, 1, 0, 1}, b2, 1, &out,
                          last ? std::string(stage) : P + ".out", resp, 1, 0, 0);
            if (rc) return rc;
            // 16-bit storage modes, 64 mid channels (res2): conv1 -> conv2 -> conv3 + residual as ONE launch with the two
            // 64-channel intermediates in LDS (bottleneck16.hip; same bits as the three launches).  APSE_NO_BNECK_FUSE (read when
            // the context is built) keeps the three-kernel form, for the equality test and A/B runs.
            {
                const size_t n = c->backbone.size();
                const ConvParams &q1 = c->backbone[n - 3].c.p, &q2 = c->backbone[n - 2].c.p, &q3 = c->backbone[n - 1].c.p;
                const int st16 = storage_type(c);
                if (st16 && stride == 1 && a.C == 64 && out.C == 256 && (cur.C == 64 || cur.C == 256) && q1.w16 && q2.w16 && q3.w16 &&
                    q1.x_st == st16 && q1.y_st == st16 && q2.y_st == st16 && q3.y_st == st16 && q3.res_st == st16 && q2.KWCp == 192 &&
                    q1.KWCp == cur.C && q3.KWCp == 64 && (size_t)B * cur.H * cur.W * cur.C * 2 < 0xfffffff0ull &&
                    !getenv("APSE_NO_BNECK_FUSE")) {
                    Step fs;
                    fs.kind = S_BNECK;
                    fs.c = c->backbone[n - 3].c;
                    fs.c.name = P + ".fused";
                    fs.c.flops_per_item = c->backbone[n - 3].c.flops_per_item + c->backbone[n - 2].c.flops_per_item + c->backbone[n - 1].c.flops_per_item;
                    fs.p2 = q2; fs.p3 = q3;
                    fs.x = cur.p; fs.y = out.p; fs.H = cur.H; fs.W = cur.W; fs.C = cur.C; fs.st = st16;
                    c->backbone.resize(n - 3);
                    c->backbone.push_back(fs);
                }
            }
            cur = out;
        }
    }
    // ---- FPN (top-down): inner5 = lateral5(res5); p5 = output5(inner5); inner_l = lateral_l(res_l) + up(inner_{l+1})
    Tens inner, pl[5];
    for (int lvl = 5; lvl >= 2; --lvl) {
        char ln[64], on[64], rn[16], in_name[16], pn[8];
        snprintf(ln, sizeof ln, "backbone.fpn_lateral%d", lvl);
        snprintf(on, sizeof on, "backbone.fpn_output%d", lvl);
        snprintf(rn, sizeof rn, "res%d", lvl);
        snprintf(in_name, sizeof in_name, "inner%d", lvl);
        snprintf(pn, sizeof pn, "p%d", lvl);
        Tens ninner;
        rc = add_conv(c, c->backbone, ConvSpec{ln, {ln}, 1, 1, 1, 0, 0}, c->t[rn], 1, &ninner, in_name,
                      lvl == 5 ? nullptr : &inner, lvl == 5 ? 0 : 2, 0, 0);
        if (rc) return rc;
        inner = ninner;
        rc = add_conv(c, c->backbone, ConvSpec{on, {on}, 3, 3, 1, 1, 0}, inner, 1, &pl[lvl - 2], pn, nullptr, 0, 0, 0);
        if (rc) return rc;
    }
    {
        Step st; st.kind = S_SUBSAMPLE; st.x = pl[3].p; st.H = pl[3].H; st.W = pl[3].W; st.C = 256;
        pl[4] = make_t(c, "p6", B, (pl[3].H - 1) / 2 + 1, (pl[3].W - 1) / 2 + 1, 256, pl[3].st);
        st.y = pl[4].p; st.c.name = "p6"; st.st = pl[3].st;
        c->backbone.push_back(st);
    }
    // ---- RPN head per level: conv3x3+relu, fused 1x1 (3 objectness + 12 deltas) -> ld 16
    static const int sizes[5] = {32, 64, 128, 256, 512};
    static const int strides[5] = {4, 8, 16, 32, 64};
    memset(&c->rl_host, 0, sizeof(c->rl_host));
    c->rl_host.head_ld = 16;
    c->rl_host.pre_topk = g.rpn_pre_topk;
    // The 3x3 convolution runs per level; its outputs are slices of ONE buffer ([level][max_batch][H][W][256]) so that the
    // fused 1x1 head (objectness + deltas, shared weights) is a single launch over all rows of all levels
    // (five launches of 10-20 us, four of them with a handful of blocks, become one).
    size_t rows_total = 0, row_off[6] = {0};
    for (int l = 0; l < 5; ++l) { row_off[l] = rows_total; rows_total += (size_t)B * pl[l].H * pl[l].W; }
    row_off[5] = rows_total;
    const int rpn_st = storage_type(c);
    Tens t_all = make_t(c, "rpn_t_all", 1, 1, (int)rows_total, 256, rpn_st);
    if (!t_all.p) return fail(c, APSE_E_NOMEM, "rpn feature buffer");
    for (int l = 0; l < 5; ++l) {
        char tn[32];
        snprintf(tn, sizeof tn, "rpn_t%d", l + 2);
        Tens view = t_all;
        view.p = reinterpret_cast<float*>(reinterpret_cast<char*>(t_all.p) + row_off[l] * 256 * (rpn_st ? 2 : 4));
        Tens tt;
        rc = add_conv(c, c->rpnhead, ConvSpec{tn, {"proposal_generator.rpn_head.conv"}, 3, 3, 1, 1, 1}, pl[l], 1, &tt, tn, nullptr,
                      0, 0, 0, nullptr, &view);
        if (rc) return rc;
    }
    Tens h_all;
    {
        Tens in_all = t_all;                       // [1][rows_total][256] as one 1 x rows image
        float* hbuf = dalloc<float>(c, rows_total * 16);
        if (!hbuf) return fail(c, APSE_E_NOMEM, "rpn head buffer");
        rc = add_conv(c, c->rpnhead,
                      ConvSpec{"rpn_head_all", {"proposal_generator.rpn_head.objectness_logits", "proposal_generator.rpn_head.anchor_deltas"},
                               1, 1, 1, 0, 0},
                      in_all, 1, &h_all, "rpn_head_all", nullptr, 0, 16, 0, hbuf);
        if (rc) return rc;
        ConvStep& hs = c->rpnhead.back().c;
        hs.fixed_items = 1;                        // all rows of all levels, whatever the batch of this forward
        hs.flops_per_item /= (double)B;            // profile accounting is per image
    }
    for (int l = 0; l < 5; ++l) {
        char hn[32];
        snprintf(hn, sizeof hn, "rpn_head%d", l + 2);
        Tens hh = h_all;
        hh.p = h_all.p + row_off[l] * 16;
        hh.H = pl[l].H; hh.W = pl[l].W; hh.C = 16;
        c->t[hn] = hh;
        RpnLevel& L = c->rl_host.lv[l];
        L.head = hh.p; L.H = hh.H; L.W = hh.W; L.stride = strides[l];
        L.n = hh.H * hh.W * 3;
        L.k = L.n < g.rpn_pre_topk ? L.n : g.rpn_pre_topk;
        static const double ratios[3] = {0.5, 1.0, 2.0};
        for (int a = 0; a < 3; ++a) {
            const double area = (double)sizes[l] * sizes[l];
            const double w = sqrt(area / ratios[a]), h = ratios[a] * w;
            L.base[a][0] = (float)(-w / 2.0); L.base[a][1] = (float)(-h / 2.0);
            L.base[a][2] = (float)(w / 2.0); L.base[a][3] = (float)(h / 2.0);
        }
    }
    c->rl_dev = dalloc<RpnLevels>(c, 1);
    hipMemcpy(c->rl_dev, &c->rl_host, sizeof(RpnLevels), hipMemcpyHostToDevice);
    // top-k tournament plan
    {
        int slot = 0;
        std::vector<std::vector<int>> cur_slots(5), cur_counts(5);
        std::vector<TopkJob> st0;
        for (int l = 0; l < 5; ++l) {
            const int n = c->rl_host.lv[l].n;
            for (int beg = 0; beg < n; beg += 4096) {
                TopkJob j; memset(&j, 0, sizeof j);
                j.kind = 0; j.level = l; j.begin = beg; j.count = (n - beg) < 4096 ? (n - beg) : 4096;
                j.dst = slot++; j.dst_count = j.count < g.rpn_pre_topk ? j.count : g.rpn_pre_topk;
                cur_slots[l].push_back(j.dst); cur_counts[l].push_back(j.dst_count);
                st0.push_back(j);
            }
        }
        c->stages.push_back(st0);
        for (;;) {
            std::vector<TopkJob> stn;
            bool any = false;
            for (int l = 0; l < 5; ++l) {
                if (cur_slots[l].size() <= 1) continue;
                any = true;
                std::vector<int> ns, nc;
                for (size_t i = 0; i < cur_slots[l].size(); i += 4) {
                    TopkJob j; memset(&j, 0, sizeof j);
                    j.kind = 1; j.level = l; int tot = 0;
                    for (size_t k = i; k < i + 4 && k < cur_slots[l].size(); ++k) {
                        j.src[j.nsrc] = cur_slots[l][k]; j.src_count[j.nsrc] = cur_counts[l][k]; tot += cur_counts[l][k]; ++j.nsrc;
                    }
                    j.dst = slot++; j.dst_count = tot < g.rpn_pre_topk ? tot : g.rpn_pre_topk;
                    ns.push_back(j.dst); nc.push_back(j.dst_count);
                    stn.push_back(j);
                }
                cur_slots[l] = ns; cur_counts[l] = nc;
            }
            if (!any) break;
            c->stages.push_back(stn);
        }
        c->nslots = slot;
        for (int l = 0; l < 5; ++l) c->final_slot_host[l] = cur_slots[l][0];
        for (auto& sv : c->stages) c->stage_dev.push_back(dupload(c, sv));
        std::vector<int> fs(c->final_slot_host, c->final_slot_host + 5);
        c->final_slot_dev = dupload(c, fs);
        c->lists = dalloc<uint64_t>(c, (size_t)B * slot * 1024);
    }
    const int PRE = g.rpn_pre_topk, POST = g.rpn_post_topk, K = g.num_classes, KD = g.dets_per_image;
    c->dec_boxes = dalloc<float>(c, (size_t)B * 5 * PRE * 4);
    c->dec_scores = dalloc<float>(c, (size_t)B * 5 * PRE);
    c->dec_valid = dalloc<int>(c, (size_t)B * 5 * PRE);
    c->maxc = dalloc<uint32_t>(c, (size_t)2 * B);
    c->keep_idx = dalloc<int>(c, (size_t)B * 8 * NMS_SLOT);
    c->keep_cnt = dalloc<int>(c, (size_t)B * 8);
    c->nms_scratch = dalloc<uint8_t>(c, apse_nms_scratch_bytes(8 * B));
    c->props = dalloc<float>(c, (size_t)B * POST * 4);
    c->prop_scores = dalloc<float>(c, (size_t)B * POST);
    c->prop_entry = dalloc<int>(c, (size_t)B * POST);
    // ---- box head: ROIAlign 7x7 -> fc1 (7x7 valid conv) -> fc2 -> fused predictor (K+1 logits, 4K deltas), ld 32
    for (int l = 0; l < 4; ++l) { c->fm.p[l] = pl[l].p; c->fm.H[l] = pl[l].H; c->fm.W[l] = pl[l].W; c->fm.scale[l] = 1.0f / (float)strides[l]; }
    c->fm.st = pl[0].st;
    Tens pooled = make_t(c, "box_pooled", B * POST, 7, 7, 256, storage_type(c));
    Tens f1, f2, pr;
    rc = add_conv(c, c->boxhead, ConvSpec{"box_fc1", {"roi_heads.box_head.fc1"}, 7, 7, 1, 0, 1, 7, 7}, pooled, POST, &f1, "box_fc1",
                  nullptr, 0, 0, 1);
    if (rc) return rc;
    rc = add_conv(c, c->boxhead, ConvSpec{"box_fc2", {"roi_heads.box_head.fc2"}, 1, 1, 1, 0, 1}, f1, POST, &f2, "box_fc2", nullptr, 0,
                  0, 1);
    if (rc) return rc;
    rc = add_conv(c, c->boxhead,
                  ConvSpec{"box_pred", {"roi_heads.box_predictor.cls_score", "roi_heads.box_predictor.bbox_pred"}, 1, 1, 1, 0, 0}, f2,
                  POST, &pr, "box_pred", nullptr, 0, 32, 1);
    if (rc) return rc;
    if (pr.C != 32 || 5 * K + 1 > 32) return fail(c, APSE_E_INVALID, "num_classes too large for the fused predictor");
    c->cand_boxes = dalloc<float>(c, (size_t)B * POST * K * 4);
    c->cand_scores = dalloc<float>(c, (size_t)B * POST * K);
    c->cand_valid = dalloc<int>(c, (size_t)B * POST * K);
    c->probs = dalloc<float>(c, (size_t)B * POST * (K + 1));
    c->det_boxes = dalloc<float>(c, (size_t)B * KD * 4);
    c->det_scores = dalloc<float>(c, (size_t)B * KD);
    c->det_entry = dalloc<int>(c, (size_t)B * KD);
    c->det_cnt = dalloc<int>(c, (size_t)B);
    // ---- mask head on the packed detection list
    const int NM = B * KD;
    Tens mp = make_t(c, "mask_pooled", NM, 14, 14, 256, storage_type(c));
    Tens m = mp, md, ml;
    for (int i = 1; i <= 4; ++i) {
        char nm[48], wn[64];
        snprintf(nm, sizeof nm, "mask_fcn%d", i);
        snprintf(wn, sizeof wn, "roi_heads.mask_head.mask_fcn%d", i);
        Tens o;
        rc = add_conv(c, c->maskhead, ConvSpec{nm, {wn}, 3, 3, 1, 1, 1}, m, KD, &o, nm, nullptr, 0, 0, 2);
        if (rc) return rc;
        m = o;
    }
    {
        ConvSpec sp{"mask_deconv", {"roi_heads.mask_head.deconv"}, 1, 1, 1, 0, 1};
        sp.deconv = 1;
        rc = add_conv(c, c->maskhead, sp, m, KD, &md, "mask_deconv", nullptr, 0, 0, 2);
        if (rc) return rc;
    }
    rc = add_conv(c, c->maskhead, ConvSpec{"mask_logits", {"roi_heads.mask_head.predictor"}, 1, 1, 1, 0, 0}, md, KD, &ml,
                  "mask_logits", nullptr, 0, 0, 2);
    if (rc) return rc;
    c->wpr = (g.frame_w + 63) / 64;
    for (int k = 0; k < 2; ++k) c->bits2[k] = dalloc<uint64_t>(c, (size_t)NM * g.frame_h * c->wpr, false);
    c->sums = dalloc<unsigned long long>(c, (size_t)NM * 3);      // cleared by pack_detections in front of every mask tail
    if (!c->bits2[0] || !c->bits2[1]) return fail(c, APSE_E_NOMEM, "mask bit planes alloc");
    // ---- association head: roi_pool(p2) -> FC (RxR valid conv) -> L2 normalise
    const int R = g.assoc_roi;
    Tens ap = make_t(c, "assoc_pooled", NM, R, R, 256);
    Tens er;
    {
        ConvSpec sp{"assoc_fc", {"association.fc"}, R, R, 1, 0, 0, R, R};
        c->emb_raw = dalloc<float>(c, (size_t)NM * g.embed_dim);
        rc = add_conv(c, c->embedfc, sp, ap, KD, &er, "assoc_fc", nullptr, 0, 0, 2, c->emb_raw);
        if (rc) return rc;
        const ConvParams& fp = c->embedfc[0].c.p;
        // the same filters through the K-sliced form when the shape allows (K = 25600, N = 128 in the reference); APSE_NO_ASSOC_FC
        // (read when the context is built) keeps the split-K convolution + normalise kernels
        if (fp.w && fp.KWCp == R * 256 && apse_assoc_fc_ok(fp.KH * fp.KWCp, g.embed_dim) && !getenv("APSE_NO_ASSOC_FC")) {
            c->ws_assoc = dalloc<float>(c, (size_t)(fp.KH * fp.KWCp / 128) * NM * g.embed_dim, false);
            if (!c->ws_assoc) return fail(c, APSE_E_NOMEM, "association FC workspace");
        }
    }
    if (c->ws_floats) {
        c->ws = dalloc<float>(c, c->ws_floats, false);
        if (!c->ws) return fail(c, APSE_E_NOMEM, "split-K workspace alloc");
    }
    c->tile_cnt = dalloc<int>(c, 65536);        // zero-initialised; every launch leaves it zero
    c->rs_pitch = (g.image_w * 3 + 15) & ~15;            // row pitch of the intermediate image: dword loads in the vertical pass
    c->rs_tmp = dalloc<uint8_t>(c, (size_t)B * g.frame_h * c->rs_pitch, false);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(c, APSE_E_HIP, std::string("plan build: ") + hipGetErrorString(e));
    return APSE_OK;
}

// ================================================================================================ C ABI
extern "C" {

#ifndef APSE_SRC_HASH
#define APSE_SRC_HASH "unknown"
#endif
const char* apse_version(void) { return "apse_hip 0.6 (gfx950, f32 / bf16 / f16 MFMA) src " APSE_SRC_HASH; }

int apse_create(const apse_config* cfg, apse_ctx** out) {
    if (!cfg || !out) return fail(nullptr, APSE_E_INVALID, "null argument");
    if (cfg->struct_size != (int)sizeof(apse_config)) return fail(nullptr, APSE_E_INVALID, "apse_config size mismatch");
    if (cfg->max_batch < 1 || cfg->max_batch > 64 || cfg->rpn_pre_topk > 1000 || cfg->rpn_post_topk > 1000 ||
        cfg->dets_per_image > 100 || cfg->num_classes < 1 || cfg->num_classes > 6 || cfg->embed_dim > 256 ||
        (cfg->frame_w & 3) != 0 || cfg->max_batch * cfg->dets_per_image > 1024)
        return fail(nullptr, APSE_E_INVALID, "config out of supported range");
    if (cfg->compute_dtype < 0 || cfg->compute_dtype > 2 || (cfg->compute_dtype && !cfg->storage16))
        return fail(nullptr, APSE_E_INVALID, "compute_dtype 1 / 2 (bf16 / f16 matrix cores) needs storage16 = 1: the f32-storage variant "
                                             "of the 16-bit modes was removed in round 4 (no BASELINE configuration uses it)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, APSE_E_HIP, "no HIP device visible: the apse_uav hot path has no CPU fallback");
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, APSE_E_HIP, "hipSetDevice failed");
    apse_ctx* c = new apse_ctx();
    c->cfg = *cfg;
    *out = c;
    return APSE_OK;
}

void apse_destroy(apse_ctx* c) {
    if (!c) return;
    hipSetDevice(c->cfg.device);
    for (void* p : c->allocs) hipFree(p);
    if (c->rf_mask) hipFree(c->rf_mask);
    if (c->read_ev) hipEventDestroy(c->read_ev);
    for (int k = 0; k < 2; ++k) {
        if (c->given_ev[k]) hipEventDestroy(c->given_ev[k]);
        if (c->given_host[k]) hipHostFree(c->given_host[k]);
    }
    delete c;
}

const char* apse_last_error(apse_ctx* c) { return c ? c->err.c_str() : g_create_error.c_str(); }

int apse_set_weight(apse_ctx* c, const char* name, const float* host, const int64_t* shape, int ndim) {
    if (!c || !name || !host || !shape || ndim < 1 || ndim > 4) return fail(c, APSE_E_INVALID, "bad weight argument");
    if (c->finalized) return fail(c, APSE_E_STATE, "weights already finalized");
    HostW w;
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) { w.shape.push_back(shape[i]); n *= (size_t)shape[i]; }
    w.v.assign(host, host + n);
    c->hw[name] = std::move(w);
    return APSE_OK;
}

int apse_finalize_weights(apse_ctx* c) {
    if (!c) return APSE_E_INVALID;
    if (c->finalized) return fail(c, APSE_E_STATE, "already finalized");
    hipSetDevice(c->cfg.device);
    int rc = build_plan(c);
    if (rc) return rc;
    c->hw.clear();
    c->finalized = true;
    return APSE_OK;
}

int apse_set_resize_tables(apse_ctx* c, const int* hb, const int* hc, int hk, const int* vb, const int* vc, int vk) {
    if (!c || !hb || !hc || !vb || !vc) return fail(c, APSE_E_INVALID, "null table");
    const apse_config& g = c->cfg;
    std::vector<int> a(hb, hb + 2 * g.image_w), b(hc, hc + (size_t)hk * g.image_w), d(vb, vb + 2 * g.image_h),
        e(vc, vc + (size_t)vk * g.image_h);
    for (int i = 0; i < g.image_w; ++i)
        if (a[2 * i] < 0 || a[2 * i + 1] > hk || a[2 * i] + a[2 * i + 1] > g.frame_w) return fail(c, APSE_E_INVALID, "bad horizontal bounds");
    for (int i = 0; i < g.image_h; ++i)
        if (d[2 * i] < 0 || d[2 * i + 1] > vk || d[2 * i] + d[2 * i + 1] > g.frame_h) return fail(c, APSE_E_INVALID, "bad vertical bounds");
    c->hb = dupload(c, a); c->hc = dupload(c, b); c->vb = dupload(c, d); c->vc = dupload(c, e);
    c->hk = hk; c->vk = vk;
    c->hcT = nullptr;
    if (hk <= 8) {
        std::vector<int> t((size_t)8 * g.image_w, 0);
        for (int i = 0; i < g.image_w; ++i)
            for (int j = 0; j < a[2 * i + 1] && j < hk; ++j) t[(size_t)j * g.image_w + i] = b[(size_t)i * hk + j];
        c->hcT = dupload(c, t);
    }
    return APSE_OK;
}

#define NEED_READY(c, batch)                                                                     \
    if (!(c) || !(c)->finalized) return fail(c, APSE_E_STATE, "weights not finalized");          \
    if ((batch) < 1 || (batch) > (c)->cfg.max_batch) return fail(c, APSE_E_INVALID, "batch out of range");

int apse_preprocess_frames(apse_ctx* c, const uint8_t* frames, int batch, void* stream) {
    NEED_READY(c, batch);
    if (!c->hb) return fail(c, APSE_E_STATE, "resize tables not set");
    const apse_config& g = c->cfg;
    int rc = apse_k_pil_resize(frames, c->rs_tmp, c->t["input"].p, c->t["input"].st, nullptr, c->hb, c->hc, c->hk, c->vb, c->vc, c->vk, batch,
                               g.frame_h, g.frame_w, g.image_h, g.image_w, c->PH, c->PW, g.pixel_mean, c->cam_on ? &c->cam : nullptr, c->cam_lut, c->cam_map_ok ? c->cam_map : nullptr,
                               c->hcT, c->rs_pitch, (hipStream_t)stream);
    return rc ? fail(c, rc, "pil resize launch failed") : APSE_OK;
}

int apse_preprocess_images(apse_ctx* c, const float* images, int batch, void* stream) {
    NEED_READY(c, batch);
    const apse_config& g = c->cfg;
    int rc = apse_k_chw_norm(images, c->t["input"].p, c->t["input"].st, batch, g.image_h, g.image_w, c->PH, c->PW, g.pixel_mean,
                             (hipStream_t)stream);
    return rc ? fail(c, rc, "chw normalise launch failed") : APSE_OK;
}

int apse_backbone(apse_ctx* c, int batch, void* stream) {
    NEED_READY(c, batch);
    return run_plan(c, c->backbone, batch, (hipStream_t)stream);
}

int apse_rpn_levels(apse_ctx* c, int batch, int level_mask, void* stream) {
    NEED_READY(c, batch);
    level_mask &= 31;
    if (!level_mask) return fail(c, APSE_E_INVALID, "empty RPN level mask");
    int first_level = 0;
    while (!((level_mask >> first_level) & 1)) ++first_level;
    hipStream_t s = (hipStream_t)stream;
    const apse_config& g = c->cfg;
    int rc = run_plan(c, c->rpnhead, batch, s);
    if (rc) return rc;
    for (size_t i = 0; i < c->stages.size(); ++i) {
        rc = apse_k_rpn_topk_stage(c->rl_dev, c->stage_dev[i], (int)c->stages[i].size(), c->lists, c->nslots, batch,
                                   i == 0 ? c->maxc : nullptr, s);
        if (rc) return fail(c, rc, "rpn top-k stage launch failed");
    }
    rc = apse_k_rpn_decode(c->rl_dev, g.rpn_pre_topk, c->lists, c->nslots, c->final_slot_dev, (float)g.image_h, (float)g.image_w,
                           (float)log(1000.0 / 16.0), c->dec_boxes, c->dec_scores, c->dec_valid, c->maxc, level_mask, batch, s);
    if (rc) return fail(c, rc, "rpn decode launch failed");
    rc = apse_k_nms_percat(c->dec_boxes, c->dec_scores, c->dec_valid, 5 * g.rpn_pre_topk, g.rpn_pre_topk, 0, c->maxc, g.rpn_nms,
                           c->keep_idx, c->keep_cnt, 5, c->nms_scratch, first_level, batch, 1, s);
    if (rc) return fail(c, rc, "rpn nms launch failed");
    int* propcnt = reinterpret_cast<int*>(c->res + c->lay.prop_count);
    rc = apse_k_rank_final(c->dec_boxes, c->dec_scores, 5 * g.rpn_pre_topk, c->keep_idx, c->keep_cnt, 5, g.rpn_post_topk, c->props,
                           c->prop_scores, c->prop_entry, propcnt, c->maxc + g.max_batch, batch, s);
    c->box_maxc_clean = true;
    if (rc) return fail(c, rc, "rpn rank launch failed");
    return APSE_OK;
}

int apse_rpn(apse_ctx* c, int batch, void* stream) { return apse_rpn_levels(c, batch, 31, stream); }

static int pack_from_dets(apse_ctx* c, int batch, hipStream_t s) {
    const apse_config& g = c->cfg;
    uint8_t* r = c->res;
    c->sums_dirty = false;                 // pack_detections clears the integer sums of the mask tail
    return apse_k_pack_detections(c->det_boxes, c->det_scores, c->det_entry, c->det_cnt, batch, g.dets_per_image, g.num_classes,
                                  (float*)(r + c->lay.box_resized), (float*)(r + c->lay.score), (int*)(r + c->lay.cls),
                                  (int*)(r + c->lay.img), (int*)(r + c->lay.roi), (int*)(r + c->lay.total),
                                  (int*)(r + c->lay.offset), c->sums, s);
}

int apse_box_head(apse_ctx* c, int batch, void* stream) {
    NEED_READY(c, batch);
    hipStream_t s = (hipStream_t)stream;
    const apse_config& g = c->cfg;
    int* propcnt = reinterpret_cast<int*>(c->res + c->lay.prop_count);
    const int P = g.rpn_post_topk, K = g.num_classes;
    int rc = apse_k_roi_align(&c->fm, c->props, nullptr, propcnt, nullptr, P, batch * P, 7, c->t["box_pooled"].p,
                              c->t["box_pooled"].st, s);
    if (rc) return fail(c, rc, "roi_align(7) launch failed");
    rc = run_plan(c, c->boxhead, batch, s);
    if (rc) return rc;
    const float wts[4] = {10.f, 10.f, 5.f, 5.f};
    if (!c->box_maxc_clean) hipMemsetAsync(c->maxc + g.max_batch, 0, sizeof(uint32_t) * g.max_batch, s);
    c->box_maxc_clean = false;
    rc = apse_k_box_candidates(c->t["box_pred"].p, 32, K, c->props, propcnt, P, (float)g.image_h, (float)g.image_w, g.score_thresh,
                               wts, (float)log(1000.0 / 16.0), c->cand_boxes, c->cand_scores, c->cand_valid,
                               c->maxc + g.max_batch, c->probs, batch, s);
    if (rc) return fail(c, rc, "box candidates launch failed");
    rc = apse_k_nms_percat(c->cand_boxes, c->cand_scores, c->cand_valid, P * K, 0, K, c->maxc + g.max_batch, g.box_nms, c->keep_idx,
                           c->keep_cnt, K, c->nms_scratch, 0, batch, 0, s);
    if (rc) return fail(c, rc, "box nms launch failed");
    rc = apse_k_rank_final(c->cand_boxes, c->cand_scores, P * K, c->keep_idx, c->keep_cnt, K, g.dets_per_image, c->det_boxes,
                           c->det_scores, c->det_entry, c->det_cnt, nullptr, batch, s);
    if (rc) return fail(c, rc, "box rank launch failed");
    rc = pack_from_dets(c, batch, s);
    return rc ? fail(c, rc, "pack launch failed") : APSE_OK;
}

int apse_set_detections(apse_ctx* c, const float* boxes, const int* classes, const float* scores, const int* counts, int batch,
                        void* stream) {
    NEED_READY(c, batch);
    hipStream_t s = (hipStream_t)stream;
    const apse_config& g = c->cfg;
    const int KD = g.dets_per_image, K = g.num_classes;
    const size_t nb = (size_t)g.max_batch * KD;
    const size_t bytes = nb * 16 + nb * 4 + nb * 4 + (size_t)g.max_batch * 4;
    const int slot = c->given_k ^= 1;
    if (!c->given_host[slot]) {
        HIPCHK(c, hipHostMalloc((void**)&c->given_host[slot], bytes, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&c->given_ev[slot], hipEventDisableTiming));
    } else {
        HIPCHK(c, hipEventSynchronize(c->given_ev[slot]));       // the copies of the call before last have left this block
    }
    float* db = (float*)c->given_host[slot];
    float* ds = db + nb * 4;
    int* de = (int*)(ds + nb);
    int* dc = de + nb;
    memset(db, 0, nb * 16 + nb * 4);
    for (size_t i = 0; i < nb; ++i) de[i] = -1;
    for (int b = 0; b < g.max_batch; ++b) dc[b] = 0;
    int o = 0;
    for (int b = 0; b < batch; ++b) {
        if (counts[b] < 0 || counts[b] > KD) return fail(c, APSE_E_INVALID, "too many given detections");
        dc[b] = counts[b];
        for (int k = 0; k < counts[b]; ++k, ++o) {
            memcpy(&db[((size_t)b * KD + k) * 4], boxes + (size_t)o * 4, 16);
            ds[(size_t)b * KD + k] = scores ? scores[o] : 1.0f;
            if (classes[o] < 0 || classes[o] >= K) return fail(c, APSE_E_INVALID, "class out of range");
            de[(size_t)b * KD + k] = k * K + classes[o];      // entry % K = class; entry / K = local index
        }
    }
    const size_t nbb = (size_t)batch * KD;
    HIPCHK(c, hipMemcpyAsync(c->det_boxes, db, nbb * 16, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->det_scores, ds, nbb * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->det_entry, de, nbb * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipMemcpyAsync(c->det_cnt, dc, (size_t)batch * 4, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipEventRecord(c->given_ev[slot], s));
    int rc = pack_from_dets(c, batch, s);
    return rc ? fail(c, rc, "pack launch failed") : APSE_OK;
}

int apse_mask_tail(apse_ctx* c, int batch, void* stream) {
    NEED_READY(c, batch);
    hipStream_t s = (hipStream_t)stream;
    const apse_config& g = c->cfg;
    uint8_t* r = c->res;
    const int NM = batch * g.dets_per_image;
    int* total = (int*)(r + c->lay.total);
    int rc = apse_k_roi_align(&c->fm, (float*)(r + c->lay.box_resized), (int*)(r + c->lay.img), nullptr, total, 0, NM, 14,
                              c->t["mask_pooled"].p, c->t["mask_pooled"].st, s);
    if (rc) return fail(c, rc, "roi_align(14) launch failed");
    rc = run_plan(c, c->maskhead, batch, s);
    if (rc) return rc;
    PasteParams p;
    p.boxes = (float*)(r + c->lay.box_resized); p.cls = (int*)(r + c->lay.cls); p.total = total;
    p.logits = c->t["mask_logits"].p; p.M = 28; p.ldc = c->t["mask_logits"].C;
    p.sx = (float)((double)g.frame_w / (double)g.image_w); p.sy = (float)((double)g.frame_h / (double)g.image_h);
    p.out_h = g.frame_h; p.out_w = g.frame_w; p.words_per_row = c->wpr; p.thresh = g.mask_thresh;
    p.boxes_out = (float*)(r + c->lay.box); p.valid = (int*)(r + c->lay.valid); p.rect = (int*)(r + c->lay.rect);
    // Two sets of bit planes, alternating per forward: a caller may enqueue the NEXT forward right behind apse_read_results_begin
    // (nothing in it depends on this one) and still copy this forward's mask windows out afterwards (apse_copy_mask_window reads
    // the set that belongs to the results last read).
    // the integer sums paste_masks adds into are cleared by pack_detections (apse_box_head / apse_set_detections); a caller that
    // repeats apse_mask_tail on one detection list (a timing loop) gets them cleared here, so the call is idempotent
    if (c->sums_dirty) HIPCHK(c, hipMemsetAsync(c->sums, 0, (size_t)g.max_batch * g.dets_per_image * 3 * sizeof(unsigned long long), s));
    c->sums_dirty = true;
    c->bits_cur ^= 1;
    if (!c->read_pending) c->bits_read = c->bits_cur;
    uint64_t* bits = c->bits2[c->bits_cur];
    p.bits = bits; p.sums = c->sums;
    unsigned long long* keys = (unsigned long long*)(r + c->lay.closest);      // raw (distance, index) keys; decoded on the host
    rc = apse_k_mask_paste(&p, NM, keys, g.dets_per_image, s);
    if (rc) return fail(c, rc, "mask paste launch failed");
    rc = apse_k_closest_points(bits, p.rect, p.valid, c->sums, (int*)(r + c->lay.img), (int*)(r + c->lay.offset), total, NM,
                               g.dets_per_image, g.frame_h, g.frame_w, c->wpr, (int*)(r + c->lay.centroid), (int*)(r + c->lay.mass),
                               keys, c->hint_total, s);
    return rc ? fail(c, rc, "closest points launch failed") : APSE_OK;
}

int apse_embed(apse_ctx* c, int batch, void* stream) {
    NEED_READY(c, batch);
    hipStream_t s = (hipStream_t)stream;
    const apse_config& g = c->cfg;
    uint8_t* r = c->res;
    const int NM = batch * g.dets_per_image;
    int* total = (int*)(r + c->lay.total);
    const Tens& p2 = c->t["p2"];
    int rc = apse_k_roi_pool(p2.p, p2.st, p2.H, p2.W, (float*)(r + c->lay.box), (int*)(r + c->lay.img), total, NM, g.assoc_roi,
                             g.assoc_scale, (float*)c->t["assoc_pooled"].p, 0, 0, s);
    if (rc) return fail(c, rc, "roi_pool launch failed");
    if (c->ws_assoc) {
        // K-sliced FC + ordered reduction + normalise (roi.hip): the filters of the plan's convolution step, its own two kernels
        const ConvParams& fp = c->embedfc[0].c.p;
        rc = apse_k_assoc_fc((const float*)c->t["assoc_pooled"].p, fp.w, fp.bias, c->ws_assoc, total, NM, fp.KH * fp.KWCp, g.embed_dim,
                             c->emb_raw, (float*)(r + c->lay.embedding), s);
        return rc ? fail(c, rc, "association FC launch failed") : APSE_OK;
    }
    rc = run_plan(c, c->embedfc, batch, s);
    if (rc) return rc;
    rc = apse_k_l2_normalize(c->emb_raw, (float*)(r + c->lay.embedding), g.embed_dim, total, NM, s);
    return rc ? fail(c, rc, "l2 normalise launch failed") : APSE_OK;
}

int apse_forward(apse_ctx* c, int batch, void* stream) {
    int rc;
    if ((rc = apse_backbone(c, batch, stream))) return rc;
    if ((rc = apse_rpn(c, batch, stream))) return rc;
    if ((rc = apse_box_head(c, batch, stream))) return rc;
    if ((rc = apse_mask_tail(c, batch, stream))) return rc;
    return apse_embed(c, batch, stream);
}

int apse_results_describe(apse_ctx* c, apse_results_layout* out) {
    if (!c || !out || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    *out = c->lay;
    return APSE_OK;
}

// The results block goes to the host in two halves so that a caller can put work behind the copy and still get the results
// as soon as the copy has landed: _begin enqueues the D2H and records an event right behind it, _end waits for THAT event
// (not for the stream: kernels enqueued after _begin -- the next frame's resize -- are not waited for).
int apse_read_results_begin(apse_ctx* c, void* host_dst, size_t bytes, void* stream) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    if (bytes < c->lay.bytes) return fail(c, APSE_E_INVALID, "results buffer too small");
    if (!c->read_ev) HIPCHK(c, hipEventCreateWithFlags(&c->read_ev, hipEventDisableTiming));
    HIPCHK(c, hipMemcpyAsync(host_dst, c->res, c->lay.bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(c, hipEventRecord(c->read_ev, (hipStream_t)stream));
    c->read_pending = host_dst;
    c->bits_read = c->bits_cur;                  // the mask windows that belong to these results
    if (c->prof_on) {                            // this forward's event pairs go to _end; the next forward records into the other half
        c->pending_read.swap(c->pending);
        c->pending.clear();
        c->cal_read = c->ev_used >= 2 ? c->ev_base : -1;
        c->ev_base ^= APSE_EV_HALF;
        c->ev_used = 0;
    }
    return APSE_OK;
}

int apse_read_results_end(apse_ctx* c, void* host_dst) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    if (!c->read_pending || c->read_pending != host_dst) return fail(c, APSE_E_STATE, "apse_read_results_end without a matching _begin");
    HIPCHK(c, hipEventSynchronize(c->read_ev));
    c->read_pending = nullptr;
    c->hint_total = *reinterpret_cast<const int*>(reinterpret_cast<const uint8_t*>(host_dst) + c->lay.total);
    {
        // closest-point table: the device leaves (f32 distance bits << 32 | row-major pixel index) keys, all ones = no point;
        // the record the caller sees holds 1-based (x, y) or (-1, -1), entries past the live detections (-1, -1)
        uint8_t* h = reinterpret_cast<uint8_t*>(host_dst) + c->lay.closest;
        const int kd = c->cfg.dets_per_image, W = c->cfg.frame_w;
        int live = c->hint_total < 0 ? 0 : c->hint_total;
        if (live > c->lay.n_max) live = c->lay.n_max;
        for (size_t t = 0; t < (size_t)c->lay.n_max * kd; ++t) {
            unsigned long long k;
            memcpy(&k, h + 8 * t, 8);
            int xy[2] = {-1, -1};
            if (t < (size_t)live * kd && k != ~0ull) {
                const unsigned lin = (unsigned)(k & 0xffffffffu);
                xy[0] = (int)(lin % (unsigned)W) + 1; xy[1] = (int)(lin / (unsigned)W) + 1;
            }
            memcpy(h + 8 * t, xy, 8);
        }
    }
    if (c->prof_on) {
        const uint8_t* h = reinterpret_cast<const uint8_t*>(host_dst);
        const int total = *reinterpret_cast<const int*>(h + c->lay.total);
        const int* pc = reinterpret_cast<const int*>(h + c->lay.prop_count);
        float cal = 0.f;
        if (c->cal_read >= 0 && hipEventElapsedTime(&cal, c->ev_pool[c->cal_read], c->ev_pool[c->cal_read + 1]) != hipSuccess) cal = 0.f;
        for (auto& q : c->pending_read) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c->ev_pool[q.e0], c->ev_pool[q.e1]) != hipSuccess) continue;
            ms = ms > cal ? ms - cal : ms;
            double items = q.batch;
            if (q.count_kind == 1) { items = 0; for (int b = 0; b < q.batch; ++b) items += pc[b]; }
            else if (q.count_kind == 2) items = total;
            c->prof[q.cfg][0] += ms; c->prof[q.cfg][1] += q.flops_per_item * items; c->prof[q.cfg][2] += 1;
        }
        c->pending_read.clear();
        c->cal_read = -1;
    }
    return APSE_OK;
}

int apse_read_results(apse_ctx* c, void* host_dst, size_t bytes, void* stream) {
    int rc = apse_read_results_begin(c, host_dst, bytes, stream);
    return rc ? rc : apse_read_results_end(c, host_dst);
}

int apse_profile(apse_ctx* c, int enable) {
    if (!c) return APSE_E_INVALID;
    if (enable && c->ev_pool.empty()) {
        c->ev_pool.resize(2 * APSE_EV_HALF);
        for (auto& e : c->ev_pool) if (hipEventCreate(&e) != hipSuccess) return fail(c, APSE_E_HIP, "hipEventCreate");
    }
    c->prof_on = enable != 0;
    c->pending.clear();
    c->pending_read.clear();
    c->cal_read = -1;
    c->ev_used = 0;
    return APSE_OK;
}

int apse_profile_read(apse_ctx* c, double* out42, int reset) {
    if (!c || !out42) return APSE_E_INVALID;
    memcpy(out42, c->prof, sizeof(c->prof));
    if (reset) memset(c->prof, 0, sizeof(c->prof));
    return APSE_OK;
}

int apse_copy_mask_window(apse_ctx* c, int det, int x0, int y0, int x1, int y1, uint64_t* dst, void* stream) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    const apse_config& g = c->cfg;
    if (det < 0 || det >= g.max_batch * g.dets_per_image || x0 < 0 || y0 < 0 || x1 > g.frame_w || y1 > g.frame_h || x1 <= x0 || y1 <= y0)
        return fail(c, APSE_E_INVALID, "bad mask window");
    const int w0 = x0 >> 6, w1 = (x1 + 63) >> 6;
    const uint64_t* src = c->bits2[c->bits_read] + ((size_t)det * g.frame_h + y0) * c->wpr + w0;
    HIPCHK(c, hipMemcpy2DAsync(dst, (size_t)(w1 - w0) * 8, src, (size_t)c->wpr * 8, (size_t)(w1 - w0) * 8, (size_t)(y1 - y0),
                               hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return APSE_OK;
}

int apse_copy_mask_windows(apse_ctx* c, int n, const int* dets, const int* rects, uint64_t* dst, const long long* dst_off, void* stream) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    const apse_config& g = c->cfg;
    if (n < 0 || n > 100 || (n > 0 && (!dets || !rects || !dst || !dst_off))) return fail(c, APSE_E_INVALID, "bad mask windows");
    long long src[100]; int nw[100], rows[100];
    for (int k = 0; k < n; ++k) {
        const int x0 = rects[k * 4], y0 = rects[k * 4 + 1], x1 = rects[k * 4 + 2], y1 = rects[k * 4 + 3];
        if (dets[k] < 0 || dets[k] >= g.max_batch * g.dets_per_image || x0 < 0 || y0 < 0 || x1 > g.frame_w || y1 > g.frame_h || x1 <= x0 ||
            y1 <= y0 || dst_off[k] < 0)
            return fail(c, APSE_E_INVALID, "bad mask window");
        const int w0 = x0 >> 6, w1 = (x1 + 63) >> 6;
        src[k] = ((long long)dets[k] * g.frame_h + y0) * c->wpr + w0;
        nw[k] = w1 - w0; rows[k] = y1 - y0;
    }
    int rc = apse_k_copy_mask_windows(c->bits2[c->bits_read], dst, n, src, dst_off, nw, rows, c->wpr, (hipStream_t)stream);
    return rc ? fail(c, rc, "mask windows launch failed") : APSE_OK;
}

int apse_feature_shape(apse_ctx* c, const char* name, int* chw3) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    auto it = c->t.find(name);
    if (it == c->t.end()) return fail(c, APSE_E_MISSING, std::string("no tensor ") + name);
    chw3[0] = it->second.C; chw3[1] = it->second.H; chw3[2] = it->second.W;
    return APSE_OK;
}

int apse_export_feature(apse_ctx* c, const char* name, float* dst, int batch, void* stream) {
    NEED_READY(c, batch);
    auto it = c->t.find(name);
    if (it == c->t.end()) return fail(c, APSE_E_MISSING, std::string("no tensor ") + name);
    const Tens& t = it->second;
    int rc = apse_k_nhwc_to_nchw(t.p, dst, batch, t.H * t.W, t.C, t.st, (hipStream_t)stream);
    return rc ? fail(c, rc, "export launch failed") : APSE_OK;
}

int apse_roi_features(apse_ctx* c, int image, const float* rois, const uint8_t* masks, int n, int roi_size, float* out, void* stream) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    const apse_config& g = c->cfg;
    if (image < 0 || image >= g.max_batch || n < 0 || roi_size < 1 || roi_size > 32 || (n > 0 && (!rois || !out)))
        return fail(c, APSE_E_INVALID, "bad roi_features arguments");
    if (n == 0) return APSE_OK;
    hipStream_t s = (hipStream_t)stream;
    const Tens& p2 = c->t["p2"];
    // spatial_scale = feature width / original width (roi_features_generator.py:105; the padded width, like rcnn_tracker.py:165)
    const float scale = (float)p2.W / (float)g.frame_w;
    int rc;
    if (!masks) {
        rc = apse_k_roi_pool(p2.p, p2.st, p2.H, p2.W, rois, nullptr, nullptr, n, roi_size, scale, out, image, 1, s);
        return rc ? fail(c, rc, "roi_pool launch failed") : APSE_OK;
    }
    const size_t need = (size_t)n * p2.H * p2.W;
    if (need > c->rf_mask_floats) {
        HIPCHK(c, hipStreamSynchronize(s));
        if (c->rf_mask) hipFree(c->rf_mask);
        c->rf_mask = nullptr; c->rf_mask_floats = 0;
        if (hipMalloc(reinterpret_cast<void**>(&c->rf_mask), need * sizeof(float)) != hipSuccess) return fail(c, APSE_E_NOMEM, "roi_features scratch");
        c->rf_mask_floats = need;
    }
    rc = apse_k_mask_resize(masks, n, g.frame_h, g.frame_w, p2.H, p2.W, c->rf_mask, s);
    if (rc) return fail(c, rc, "mask resize launch failed");
    rc = apse_k_roi_align_masked(p2.p, p2.st, p2.H, p2.W, image, rois, c->rf_mask, n, roi_size, 4, scale, out, s);
    return rc ? fail(c, rc, "masked roi_align launch failed") : APSE_OK;
}

int apse_debug_tensor(apse_ctx* c, const char* name, void* dst, size_t max_bytes, size_t* bytes, void* stream) {
    if (!c || !c->finalized) return fail(c, APSE_E_STATE, "not finalized");
    const apse_config& g = c->cfg;
    const void* src = nullptr;
    size_t n = 0;
    const std::string nm = name;
    const int B = g.max_batch;
    if (nm == "proposals") { src = c->props; n = (size_t)B * g.rpn_post_topk * 16; }
    else if (nm == "proposal_scores") { src = c->prop_scores; n = (size_t)B * g.rpn_post_topk * 4; }
    else if (nm == "proposal_entry") { src = c->prop_entry; n = (size_t)B * g.rpn_post_topk * 4; }
    else if (nm == "rpn_decoded") { src = c->dec_boxes; n = (size_t)B * 5 * g.rpn_pre_topk * 16; }
    else if (nm == "rpn_decoded_scores") { src = c->dec_scores; n = (size_t)B * 5 * g.rpn_pre_topk * 4; }
    else if (nm == "rpn_decoded_valid") { src = c->dec_valid; n = (size_t)B * 5 * g.rpn_pre_topk * 4; }
    else if (nm == "box_probs") { src = c->probs; n = (size_t)B * g.rpn_post_topk * (g.num_classes + 1) * 4; }
    else if (nm == "cand_boxes") { src = c->cand_boxes; n = (size_t)B * g.rpn_post_topk * g.num_classes * 16; }
    else if (nm == "det_boxes") { src = c->det_boxes; n = (size_t)B * g.dets_per_image * 16; }
    else if (nm == "det_scores") { src = c->det_scores; n = (size_t)B * g.dets_per_image * 4; }
    else if (nm == "det_entry") { src = c->det_entry; n = (size_t)B * g.dets_per_image * 4; }
    else if (nm == "det_count") { src = c->det_cnt; n = (size_t)B * 4; }
    else if (nm == "embedding_raw") { src = c->emb_raw; n = (size_t)B * g.dets_per_image * g.embed_dim * 4; }
    else {
        auto it = c->t.find(nm);
        if (it == c->t.end()) return fail(c, APSE_E_MISSING, "no tensor " + nm);
        const Tens& t = it->second;
        int items = B;
        if (nm == "box_pooled" || nm == "box_fc1" || nm == "box_fc2" || nm == "box_pred") items = B * g.rpn_post_topk;
        else if (nm.rfind("mask_", 0) == 0 || nm.rfind("assoc_", 0) == 0) items = B * g.dets_per_image;
        src = t.p; n = (size_t)items * t.H * t.W * t.C * (t.st ? 2 : 4);
    }
    if (bytes) *bytes = n;
    if (!dst) return APSE_OK;
    if (n > max_bytes) return fail(c, APSE_E_INVALID, "debug buffer too small for " + nm);
    HIPCHK(c, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return APSE_OK;
}

double apse_flops(apse_ctx* c, int batch, double proposals, double detections) {
    if (!c || !c->finalized) return 0.0;
    double f = 0;
    for (auto& s : c->backbone) if (s.kind == S_CONV || s.kind == S_BNECK) f += s.c.flops_per_item * batch;
    for (auto& s : c->rpnhead) f += s.c.flops_per_item * batch;
    for (auto& s : c->boxhead) f += s.c.flops_per_item * proposals;
    for (auto& s : c->maskhead) f += s.c.flops_per_item * detections;
    for (auto& s : c->embedfc) f += s.c.flops_per_item * detections;
    return f;
}

// ------------------------------------------------------------------------------------------------ stateless ops
size_t apse_conv_packed_elems(const apse_conv_desc* d) {
    const int cin_p = pow2_at_least(d->Cin);
    return (size_t)apse_roundup(d->Cout, 128) * d->KH * apse_roundup(d->KW * cin_p, 32);
}

int apse_conv_pack_weight(const apse_conv_desc* d, const float* w, int cin_real, const float* scale, float* packed) {
    if (!d || !w || !packed) return APSE_E_INVALID;
    const int cin_p = pow2_at_least(d->Cin);
    const int KWCp = apse_roundup(d->KW * cin_p, 32);
    memset(packed, 0, apse_conv_packed_elems(d) * sizeof(float));
    pack_oihw(w, d->Cout, cin_real, d->KH, d->KW, cin_p, scale, packed, KWCp);
    return APSE_OK;
}

int apse_conv2d(const apse_conv_desc* d, const float* x, const float* w, const float* bias, const float* res, float* y, float* ws,
                size_t ws_bytes, void* stream) {
    if (!d || !x || !w || !y) return APSE_E_INVALID;
    const int cin_p = pow2_at_least(d->Cin);
    if (cin_p != d->Cin) return APSE_E_INVALID;
    ConvParams p;
    memset(&p, 0, sizeof p);
    p.x = x; p.w = w; p.bias = bias; p.res = res; p.y = y; p.ws = ws;
    p.B = d->B; p.H = d->H; p.W = d->W; p.cin_log2 = apse_ilog2(cin_p);
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.KWCp = apse_roundup(d->KW * cin_p, 32);
    p.OH = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    p.OW = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    p.Cout = d->Cout; p.relu = d->relu; p.res_mode = d->res_mode;
    p.M = p.B * p.OH * p.OW; p.m_per_item = p.OH * p.OW;
    p.y_ld = d->Cout; p.steps_total = p.KH * (p.KWCp / 32);
    p.prec = (d->prec == 1 || d->prec == 2) ? d->prec : 0;
    p.x_st = d->x_st; p.res_st = d->res_st; p.y_st = d->y_st;
    if (p.x_st < 0 || p.x_st > 2 || p.res_st < 0 || p.res_st > 2 || p.y_st < 0 || p.y_st > 2) return APSE_E_INVALID;
    if (p.prec && p.x_st && p.x_st != p.prec) return APSE_E_INVALID;       // 16-bit x must already be the operand type
    if (p.x_st && cin_p < 8) return APSE_E_INVALID;
    int sk = 1;
    int cfg = apse_conv_pick_cfg(p.M, p.Cout, p.steps_total, &sk);
    if (d->cfg >= 0) { cfg = d->cfg; sk = 1; p.no_stream = (d->cfg != APSE_CFG_STREAM && d->cfg != APSE_CFG_GLDS && d->cfg != APSE_CFG_SKINNY); }
    if (d->splitk > 0) sk = d->splitk;
    if (sk > p.steps_total) sk = p.steps_total;
    p.splitk = sk;
    if (sk > 1 && (size_t)sk * p.M * p.Cout * sizeof(float) > ws_bytes) return APSE_E_INVALID;
    if (sk > 1 && d->fuse_reduce) {
        static int* cnt = nullptr;
        if (!cnt) { if (hipMalloc(reinterpret_cast<void**>(&cnt), 65536 * sizeof(int)) != hipSuccess) return APSE_E_NOMEM; hipMemset(cnt, 0, 65536 * sizeof(int)); }
        p.tile_cnt = cnt;
    }
    if (!p.prec) return apse_launch_conv(p, cfg, (hipStream_t)stream);
    // 16-bit operands: round the filters like a context does at load (this stateless entry is a test / tool helper: the
    // rounded copy is rebuilt by a small kernel on the caller's stream in front of every call, in a buffer that only grows)
    const size_t ne = (size_t)apse_roundup(p.Cout, 128) * p.KH * p.KWCp;
    static uint16_t* d16 = nullptr;
    static size_t d16_cap = 0;
    if (ne > d16_cap) {
        hipDeviceSynchronize();
        if (d16) hipFree(d16);
        d16 = nullptr; d16_cap = 0;
        if (hipMalloc(reinterpret_cast<void**>(&d16), ne * 2) != hipSuccess) return APSE_E_NOMEM;
        d16_cap = ne;
    }
    int rc = apse_k_round16(w, d16, ne, p.prec, (hipStream_t)stream);
    if (rc) return rc;
    p.w16 = d16;
    return apse_launch_conv(p, cfg, (hipStream_t)stream);
}

int apse_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    return apse_k_maxpool3x3s2(x, y, B, H, W, C, 0, (hipStream_t)stream);
}

int apse_maxpool3x3s2_typed(const void* x, void* y, int B, int H, int W, int C, int storage, void* stream) {
    if (!x || !y || B < 1 || H < 1 || W < 1 || C < 4 || (C & 3) || storage < 0 || storage > 2) return APSE_E_INVALID;
    return apse_k_maxpool3x3s2(x, y, B, H, W, C, storage, (hipStream_t)stream);
}

static int roi_align_stateless(const void* const* feats, const int* hs, const int* ws, const float* rois, int n, int per_img,
                               int out_size, int st, void* out, void* stream) {
    if (!feats || !hs || !ws || !rois || !out || n < 0 || out_size < 1 || st < 0 || st > 2) return APSE_E_INVALID;
    FpnMaps F;
    static const float sc[4] = {0.25f, 0.125f, 0.0625f, 0.03125f};
    for (int l = 0; l < 4; ++l) { F.p[l] = feats[l]; F.H[l] = hs[l]; F.W[l] = ws[l]; F.scale[l] = sc[l]; }
    F.st = st;
    // all rois live: a one-element count array is not available here, so use a device int holding n via total
    static int* total_dev = nullptr;
    if (!total_dev) hipMalloc(reinterpret_cast<void**>(&total_dev), sizeof(int));
    hipMemcpyAsync(total_dev, &n, sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream);
    hipStreamSynchronize((hipStream_t)stream);
    // roi_img derived from per_img: build on device via a tiny host vector
    std::vector<int> img(n);
    for (int i = 0; i < n; ++i) img[i] = per_img > 0 ? i / per_img : 0;
    int* img_dev = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&img_dev), sizeof(int) * (n > 0 ? n : 1)) != hipSuccess) return APSE_E_NOMEM;
    hipMemcpy(img_dev, img.data(), sizeof(int) * n, hipMemcpyHostToDevice);
    int rc = apse_k_roi_align(&F, rois, img_dev, nullptr, total_dev, 0, n, out_size, out, st, (hipStream_t)stream);
    hipStreamSynchronize((hipStream_t)stream);
    hipFree(img_dev);
    return rc;
}

int apse_roi_align(const float* const* feats, const int* hs, const int* ws, const float* rois, int n, int per_img, int out_size,
                   float* out, void* stream) {
    return roi_align_stateless(reinterpret_cast<const void* const*>(feats), hs, ws, rois, n, per_img, out_size, 0, out, stream);
}

int apse_roi_align_typed(const void* const* feats, const int* hs, const int* ws, const float* rois, int n, int per_img,
                         int out_size, int storage, void* out, void* stream) {
    return roi_align_stateless(feats, hs, ws, rois, n, per_img, out_size, storage, out, stream);
}

int apse_roi_pool(const float* feat, int H, int W, const float* rois, const int* roi_img, int n, int out_size, float scale,
                  float* out, void* stream) {
    static int* total_dev = nullptr;
    if (!total_dev) hipMalloc(reinterpret_cast<void**>(&total_dev), sizeof(int));
    hipMemcpyAsync(total_dev, &n, sizeof(int), hipMemcpyHostToDevice, (hipStream_t)stream);
    hipStreamSynchronize((hipStream_t)stream);
    return apse_k_roi_pool(feat, 0, H, W, rois, roi_img, total_dev, n, out_size, scale, out, 0, 0, (hipStream_t)stream);
}

int apse_nms_rank(const float* boxes, const float* scores, const int* valid, int n, int cat_div, int cat_mod, int ncat, float thr,
                  int topk, float* out_boxes, float* out_scores, int* out_index, int* out_count, void* stream) {
    if (ncat < 1 || ncat > 8 || n > 8192) return APSE_E_INVALID;
    hipStream_t s = (hipStream_t)stream;
    int *keep_idx = nullptr, *keep_cnt = nullptr;
    uint32_t* maxc = nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&keep_idx), sizeof(int) * 8 * NMS_SLOT) != hipSuccess) return APSE_E_NOMEM;
    hipMalloc(reinterpret_cast<void**>(&keep_cnt), sizeof(int) * 8);
    hipMalloc(reinterpret_cast<void**>(&maxc), sizeof(uint32_t));
    // max coordinate over the valid boxes (torchvision batched_nms): computed on the host for this stateless op
    std::vector<float> hb((size_t)n * 4);
    std::vector<int> hv(n);
    hipMemcpy(hb.data(), boxes, hb.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hv.data(), valid, hv.size() * 4, hipMemcpyDeviceToHost);
    float m = 0.f;
    bool any = false;
    for (int i = 0; i < n; ++i)
        if (hv[i]) for (int k = 0; k < 4; ++k) { m = (!any || hb[i * 4 + k] > m) ? hb[i * 4 + k] : m; any = true; }
    uint32_t mb;
    memcpy(&mb, &m, 4);
    hipMemcpy(maxc, &mb, 4, hipMemcpyHostToDevice);
    void* scratch = nullptr;
    if (hipMalloc(&scratch, apse_nms_scratch_bytes(8)) != hipSuccess) return APSE_E_NOMEM;
    int rc = apse_k_nms_percat(boxes, scores, valid, n, cat_div, cat_mod, maxc, thr, keep_idx, keep_cnt, ncat, scratch, 0, 1, 0, s);
    if (!rc) rc = apse_k_rank_final(boxes, scores, n, keep_idx, keep_cnt, ncat, topk, out_boxes, out_scores, out_index, out_count, nullptr, 1, s);
    hipStreamSynchronize(s);
    hipFree(keep_idx); hipFree(keep_cnt); hipFree(maxc); hipFree(scratch);
    return rc;
}

static int dense_scratch(int H, int W, uint64_t** bits, unsigned long long** sums) {
    static uint64_t* b = nullptr;
    static unsigned long long* sm = nullptr;
    static size_t words = 0;
    const size_t need = (size_t)H * ((W + 63) / 64);
    if (need > words) {
        if (b) hipFree(b);
        if (hipMalloc(reinterpret_cast<void**>(&b), need * 8) != hipSuccess) return APSE_E_NOMEM;
        words = need;
    }
    if (!sm && hipMalloc(reinterpret_cast<void**>(&sm), 4 * sizeof(unsigned long long)) != hipSuccess) return APSE_E_NOMEM;
    *bits = b; *sums = sm;
    return APSE_OK;
}

int apse_mask_centroid_dense(const uint8_t* mask, int H, int W, int* out3, void* stream) {
    uint64_t* bits; unsigned long long* sums;
    int rc = dense_scratch(H, W, &bits, &sums);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = apse_k_dense_to_bits(mask, H, W, (W + 63) / 64, bits, sums, s);
    if (rc) return rc;
    unsigned long long h[3];
    if (hipMemcpyAsync(h, sums, sizeof h, hipMemcpyDeviceToHost, s) != hipSuccess) return APSE_E_HIP;
    hipStreamSynchronize(s);
    out3[2] = (int)h[0];
    out3[0] = h[0] ? (int)(h[1] / h[0]) : -1;
    out3[1] = h[0] ? (int)(h[2] / h[0]) : -1;
    return APSE_OK;
}

int apse_mask_closest_dense(const uint8_t* mask, int H, int W, float px, float py, int* out2, void* stream) {
    uint64_t* bits; unsigned long long* sums;
    int rc = dense_scratch(H, W, &bits, &sums);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = apse_k_dense_to_bits(mask, H, W, (W + 63) / 64, bits, sums, s);
    if (rc) return rc;
    rc = apse_k_closest_single(bits, H, W, (W + 63) / 64, px, py, sums + 3, s);
    if (rc) return rc;
    unsigned long long best;
    if (hipMemcpyAsync(&best, sums + 3, sizeof best, hipMemcpyDeviceToHost, s) != hipSuccess) return APSE_E_HIP;
    hipStreamSynchronize(s);
    if (best == ~0ull) { out2[0] = out2[1] = -1; return APSE_OK; }
    const unsigned lin = (unsigned)(best & 0xffffffffu);
    out2[0] = (int)(lin % (unsigned)W) + 1;
    out2[1] = (int)(lin / (unsigned)W) + 1;
    return APSE_OK;
}

int apse_l2_normalize(const float* x, float* y, int n, int D, void* stream) {
    return apse_k_l2_normalize(x, y, D, nullptr, n, (hipStream_t)stream);
}
int apse_sqdist(const float* a, const float* b, int O, int N, int D, float* out, void* stream) {
    return apse_k_sqdist(a, b, O, N, D, out, (hipStream_t)stream);
}
static int fill_camera(UndistortParams& p, int H, int W, const double* m, const double* dist, int ndist, int do_undistort, int do_gamma) {
    memset(&p, 0, sizeof p);
    if (!m || ndist > 14 || ndist < 0 || (ndist > 0 && !dist)) return APSE_E_INVALID;
    for (int i = 0; i < ndist && i < 12; ++i) p.k[i] = dist[i];
    if (ndist > 12 && (dist[12] != 0.0 || (ndist > 13 && dist[13] != 0.0))) return APSE_E_INVALID;   // tilt model not built
    // inverse of the 3x3 camera matrix (double, adjugate / determinant)
    const double a = m[0], b = m[1], c = m[2], dd = m[3], e = m[4], f = m[5], g = m[6], h = m[7], k = m[8];
    const double det = a * (e * k - f * h) - b * (dd * k - f * g) + c * (dd * h - e * g);
    if (det == 0.0) return APSE_E_INVALID;
    const double id = 1.0 / det;
    p.ir[0] = (e * k - f * h) * id; p.ir[1] = (c * h - b * k) * id; p.ir[2] = (b * f - c * e) * id;
    p.ir[3] = (f * g - dd * k) * id; p.ir[4] = (a * k - c * g) * id; p.ir[5] = (c * dd - a * f) * id;
    p.ir[6] = (dd * h - e * g) * id; p.ir[7] = (b * g - a * h) * id; p.ir[8] = (a * e - b * dd) * id;
    p.fx = m[0]; p.fy = m[4]; p.u0 = m[2]; p.v0 = m[5];
    p.H = H; p.W = W; p.do_undistort = do_undistort; p.do_gamma = do_gamma;
    return APSE_OK;
}

// Lab tables of a gamma LUT, device-resident for the stateless operator: one copy PER DEVICE (keyed by hipGetDevice), rebuilt
// when the LUT changes, guarded by a mutex (the operator is a test / tool entry; a context keeps its own copy)
static int lab_tables_device(const uint8_t* lut, hipStream_t s, LabTables** out) {
    struct PerDev { LabTables* dev = nullptr; uint8_t lut[256]; bool have = false; };
    static std::mutex mu;
    static std::map<int, PerDev> cache;
    static LabTables host;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return APSE_E_HIP;
    std::lock_guard<std::mutex> lock(mu);
    PerDev& e = cache[device];
    if (!e.dev && hipMalloc(reinterpret_cast<void**>(&e.dev), sizeof(LabTables)) != hipSuccess) return APSE_E_NOMEM;
    if (!e.have || memcmp(e.lut, lut, 256) != 0) {
        hipStreamSynchronize(s);                       // an earlier launch may still read the previous tables
        lab_tables_build(&host, lut);
        if (hipMemcpy(e.dev, &host, sizeof(LabTables), hipMemcpyHostToDevice) != hipSuccess) return APSE_E_HIP;
        memcpy(e.lut, lut, 256);
        e.have = true;
    }
    *out = e.dev;
    return APSE_OK;
}

int apse_undistort_gamma(const uint8_t* src, uint8_t* dst, int B, int H, int W, const double* m, const double* dist, int ndist,
                         const uint8_t* lut, int do_undistort, int do_gamma, void* stream) {
    if (!src || !dst || (do_gamma && !lut)) return APSE_E_INVALID;
    UndistortParams p;
    int rc = fill_camera(p, H, W, m, dist, ndist, do_undistort, do_gamma);
    if (rc) return rc;
    LabTables* lab = nullptr;
    if (do_gamma && (rc = lab_tables_device(lut, (hipStream_t)stream, &lab))) return rc;
    return apse_k_undistort_gamma(&p, src, dst, lab, B, (hipStream_t)stream);
}

size_t apse_lab_tables_host(const uint8_t* lut256, void* out, size_t cap) {
    if (!lut256 || !out || cap < sizeof(LabTables)) return sizeof(LabTables);
    lab_tables_build(reinterpret_cast<LabTables*>(out), lut256);
    return sizeof(LabTables);
}

int apse_set_camera(apse_ctx* c, const double* m, const double* dist, int ndist, const uint8_t* lut_host, int do_undistort, int do_gamma) {
    if (!c) return APSE_E_INVALID;
    if (!do_undistort && !do_gamma) { c->cam_on = false; return APSE_OK; }
    if (do_gamma && !lut_host) return fail(c, APSE_E_INVALID, "gamma needs a 256-entry LUT");
    UndistortParams p;
    int rc = fill_camera(p, c->cfg.frame_h, c->cfg.frame_w, m, dist, ndist, do_undistort, do_gamma);
    if (rc) return fail(c, rc, "bad camera parameters (3x3 matrix, <= 14 distortion coefficients, tilt terms zero)");
    hipSetDevice(c->cfg.device);
    if (!c->cam_lut) {
        c->cam_lut = dalloc<LabTables>(c, 1);
        if (!c->cam_lut) return fail(c, APSE_E_NOMEM, "camera Lab tables alloc");
    }
    if (lut_host) {
        // the Lab step is integer arithmetic on these tables (preproc_pixel.h); built here once, on the host, in double
        LabTables host;
        lab_tables_build(&host, lut_host);
        HIPCHK(c, hipDeviceSynchronize());
        HIPCHK(c, hipMemcpy(c->cam_lut, &host, sizeof(LabTables), hipMemcpyHostToDevice));
    }
    // the remap table depends on the camera only: built here once (f64 rational model per pixel), read per frame -- 4 bytes per
    // pixel: the source position relative to the pixel in 1/32 px (preproc_pixel.h).  A camera that displaces a pixel inside the
    // frame by 1024 px or more does not fit; it keeps the per-pixel model (slower, same bytes).
    c->cam_map_ok = false;
    if (do_undistort) {
        if (!c->cam_map) {
            c->cam_map = dalloc<uint32_t>(c, (size_t)c->cfg.frame_h * c->cfg.frame_w + 4, false);
            if (!c->cam_map) return fail(c, APSE_E_NOMEM, "camera map alloc");
        }
        int* ovf = reinterpret_cast<int*>(reinterpret_cast<uint32_t*>(c->cam_map) + (size_t)c->cfg.frame_h * c->cfg.frame_w);
        HIPCHK(c, hipMemset(ovf, 0, sizeof(int)));
        rc = apse_k_undistort_build_map_compact(&p, c->cam_map, ovf, nullptr);
        if (rc) return fail(c, rc, "camera map launch failed");
        int overflow = 0;
        HIPCHK(c, hipMemcpy(&overflow, ovf, sizeof(int), hipMemcpyDeviceToHost));
        c->cam_map_ok = overflow == 0;
    }
    HIPCHK(c, hipDeviceSynchronize());
    c->cam = p;
    c->cam_on = true;
    return APSE_OK;
}
int apse_resize_normalize(const uint8_t* frames, uint8_t* tmp, float* out, uint8_t* resized, const int* hb, const int* hc, int hk,
                          const int* vb, const int* vc, int vk, int B, int H, int W, int OH, int OW, int PH, int PW,
                          const float* mean3, void* stream) {
    return apse_k_pil_resize(frames, tmp, out, 0, resized, hb, hc, hk, vb, vc, vk, B, H, W, OH, OW, PH, PW, mean3, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
}

}  // extern "C"
