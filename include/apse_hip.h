/* apse_hip.h -- C ABI of libapse_hip.so: the MI355X (gfx950) implementation of the per-frame
 * hot path of vision-agh/apse_uav's dcnn/ subsystem.
 *
 * The reference has no FFI layer: its boundary is the Python surface
 *   dcnn/engines/track_predictor.py:31-52  TrackPredictor.__call__   (resize + TrackRCNN.inference)
 *   dcnn/networks/track_rcnn.py:16-58      TrackRCNN.inference       (preprocess, backbone+FPN, RPN, ROI heads, postprocess)
 *   dcnn/engines/rcnn_tracker.py:156-221   get_features_rois, association head, distance matrix
 *   dcnn/utils/mask_utils.py:6-38          get_mask_centroid, compute_closest_point
 * Each entry point below names the reference code it replaces.  apse_uav_amd/_lib.py is the ctypes
 * binding; INTEGRATION.md shows the stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 (APSE_OK) or a negative APSE_E_* code; apse_last_error()
 * gives the text.  Pointers named *_dev are device pointers owned by the caller; `stream` is a
 * hipStream_t passed as void*.  Calls enqueue work and return; only apse_read_results (or its _end half) waits.
 * A context is bound to one device and is not thread-safe (one per process rank).
 */
#ifndef APSE_HIP_H
#define APSE_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define APSE_OK 0
#define APSE_E_INVALID (-1)
#define APSE_E_HIP (-2)
#define APSE_E_NOMEM (-3)
#define APSE_E_STATE (-4)
#define APSE_E_MISSING (-5)

typedef struct apse_ctx apse_ctx;

/* Hyper-parameters that define results (dcnn/configs/Base-RCNN-FPN.yaml,
 * dcnn/scripts/tests/visualize_uav.py:30-48, dcnn/engines/rcnn_tracker.py:33). */
typedef struct apse_config {
    int struct_size;              /* sizeof(apse_config), ABI check */
    int device;
    int max_batch;                /* frames per forward (reference: 1) */
    int frame_h, frame_w;         /* original frame, e.g. 2160 x 3840 */
    int image_h, image_w;         /* after ResizeShortestEdge, e.g. 750 x 1333 */
    int blocks[4];                /* bottlenecks per stage, R-101 = 3,4,23,3 */
    int num_classes;              /* 4 */
    float score_thresh;           /* 0.5 */
    float box_nms;                /* 0.5 */
    float rpn_nms;                /* 0.7 */
    float mask_thresh;            /* 0.5 */
    int rpn_pre_topk;             /* 1000 (<= 1000) */
    int rpn_post_topk;            /* 1000 (<= 1000) */
    int dets_per_image;           /* 100  (<= 100) */
    float pixel_mean[3];          /* BGR means */
    int assoc_roi;                /* 10 */
    int embed_dim;                /* 128 */
    float assoc_scale;            /* roi_pool spatial scale = p2 width / frame width (rcnn_tracker.py:165) */
    int compute_dtype;            /* 0 = exact f32 MFMA everywhere (reference numerics); 1 = bf16, 2 = f16 matrix cores with
                                     f32 accumulate for the trunk / head GEMMs (decision layers stay f32) */
    int storage16;                /* with compute_dtype 1/2: must be 1 = activations live in HBM in the 16-bit operand type (half
                                     the traffic, no conversion on the way to the matrix cores).  The f32-storage variant of
                                     the 16-bit modes was removed in round 4 (apse_create refuses it); ignored for f32 */
} apse_config;

/* Byte offsets of the per-forward results block (one D2H copy, apse_read_results). n = max_batch*dets_per_image. */
typedef struct apse_results_layout {
    size_t bytes;
    int n_max, dets_per_image, embed_dim, max_batch;
    size_t total;        /* int                 number of packed detections                           */
    size_t offset;       /* int  [max_batch+1]  first packed index of each image                      */
    size_t prop_count;   /* int  [max_batch]    RPN proposals kept per image                          */
    size_t img;          /* int  [n]            image of each detection                               */
    size_t cls;          /* int  [n]            predicted class                                       */
    size_t roi;          /* int  [n]            proposal index the detection came from (-1 given box) */
    size_t score;        /* f32  [n]                                                                  */
    size_t box_resized;  /* f32  [n][4]         box in resized-image pixels                           */
    size_t box;          /* f32  [n][4]         box in frame pixels (scaled + clipped)                */
    size_t valid;        /* int  [n]            box non-empty after scaling (detector_postprocess)    */
    size_t rect;         /* int  [n][4]         paste window x0,y0,x1,y1                              */
    size_t mass;         /* int  [n]            mask pixel count                                      */
    size_t centroid;     /* int  [n][2]         1-based floor centroid, -1 if empty                   */
    size_t closest;      /* int  [n][dets_per_image][2]  closest mask pixel of det i to the centroid of
                                                 the j-th detection of the same image, -1 if none     */
    size_t embedding;    /* f32  [n][embed_dim] L2-normalised association embedding                   */
} apse_results_layout;

/* ---- lifetime ---- */
int apse_create(const apse_config* cfg, apse_ctx** out);
void apse_destroy(apse_ctx* ctx);
const char* apse_last_error(apse_ctx* ctx);          /* ctx may be NULL: last creation error */
const char* apse_version(void);

/* ---- weights: replaces DetectionCheckpointer.load / load_state_dict (track_predictor.py:20-21,
 * rcnn_tracker.py:55-57).  Names are detectron2 state_dict keys, plus "association.fc.weight|bias".
 * Host f32 arrays in PyTorch layout.  apse_finalize_weights folds FrozenBN into the convolutions,
 * re-lays the filters out for the implicit-GEMM kernels, uploads them and builds the launch plan. */
int apse_set_weight(apse_ctx* ctx, const char* name, const float* host, const int64_t* shape, int ndim);
int apse_finalize_weights(apse_ctx* ctx);

/* Pillow resampling tables for frame->image (host int32; computed by the caller exactly as Pillow's
 * precompute_coeffs/normalize_coeffs_8bpc do): bounds [out][2] = (first, count), coef [out][ksize]. */
int apse_set_resize_tables(apse_ctx* ctx, const int* hbounds, const int* hcoef, int hksize, const int* vbounds,
                           const int* vcoef, int vksize);

/* Optional: preprocess_img of dcnn/scripts/tests/visualize_uav.py:56-71 (cv2.undistort with data/cam_params.json + gamma on the
 * Lab L channel) fused into apse_preprocess_frames -- the raw frame is undistorted / gamma-corrected while the horizontal
 * resize pass stages its rows, so no pre-processed 4K frame is ever written to HBM.  m: 3x3 camera matrix (row-major, host
 * f64), dist: up to 14 coefficients (k1 k2 p1 p2 k3 k4 k5 k6 s1..s4, tilt terms must be 0), lut: 256-entry L table (host).
 * do_undistort = do_gamma = 0 switches it off.  Synchronous (small host -> device copy). */
int apse_set_camera(apse_ctx* ctx, const double* m9, const double* dist, int ndist, const uint8_t* lut256_host, int do_undistort,
                    int do_gamma);

/* ---- per-frame stages (all enqueue on `stream`) ---- */
/* ResizeShortestEdge.apply_image (PIL bilinear) + preprocess_image: u8 BGR frames [B][frame_h][frame_w][3]
 * -> internal normalised, /32-padded network input (track_predictor.py:48-49, track_rcnn.py:35). */
int apse_preprocess_frames(apse_ctx* ctx, const uint8_t* frames_dev, int batch, void* stream);
/* Same from already-resized f32 CHW images [B][3][image_h][image_w] (the reference's model input). */
int apse_preprocess_images(apse_ctx* ctx, const float* images_dev, int batch, void* stream);
/* ResNet-101 + FPN (track_rcnn.py:42). */
int apse_backbone(apse_ctx* ctx, int batch, void* stream);
/* RPN head + proposal selection (track_rcnn.py:46). */
int apse_rpn(apse_ctx* ctx, int batch, void* stream);
/* Same with proposals taken only from the FPN levels in `level_mask` (bit 0 = p2 .. bit 4 = p6); mask 16 is
 * SelectiveRPN.gen_partial_proposals (dcnn/networks/selective_rpn.py:14-86: last level only). */
int apse_rpn_levels(apse_ctx* ctx, int batch, int level_mask, void* stream);
/* Box branch + FastRCNNOutputs.inference (track_rcnn.py:51) -> packed detection list. */
int apse_box_head(apse_ctx* ctx, int batch, void* stream);
/* roi_heads.forward_with_given_boxes (track_rcnn.py:52-54): host arrays, boxes in resized-image pixels,
 * counts[batch] detections per image, concatenated.  Replaces apse_rpn + apse_box_head.  Enqueues only: the arrays are
 * copied into pinned staging owned by the context before the call returns and may be freed by the caller at once. */
int apse_set_detections(apse_ctx* ctx, const float* boxes_host, const int* classes_host, const float* scores_host,
                        const int* counts_host, int batch, void* stream);
/* Mask branch + detector_postprocess/paste + centroids + closest-point table
 * (track_rcnn.py:51,57; mask_utils.py:6-38).  Idempotent on one detection list: a repeated call (a timing loop) gives the
 * same masses / centroids, not doubled ones. */
int apse_mask_tail(apse_ctx* ctx, int batch, void* stream);
/* roi_pool(p2) + AssociationHead (rcnn_tracker.py:156-189, association_head.py:16-27). */
int apse_embed(apse_ctx* ctx, int batch, void* stream);
/* RoiFeaturesGenerator.get_rois_features (dcnn/engines/roi_features_generator.py:68-117), to be called after
 * apse_backbone: RoI features of image `image` of the batch on p2 for the association-head training batches.
 * rois_dev: [n][4] x1,y1,x2,y2 in ORIGINAL-frame pixels.  masks_dev == NULL: torchvision roi_pool (:113);
 * else [n][frame_h][frame_w] u8 (non-zero = inside): p2 * bilinear-resized mask, then roi_align(aligned=False,
 * sampling_ratio=4) (:95-111).  out_dev: f32 [n][256][roi_size][roi_size]. */
int apse_roi_features(apse_ctx* ctx, int image, const float* rois_dev, const uint8_t* masks_dev, int n, int roi_size,
                      float* out_dev, void* stream);
/* backbone + rpn + box_head + mask_tail + embed. */
int apse_forward(apse_ctx* ctx, int batch, void* stream);

/* ---- results ---- */
int apse_results_describe(apse_ctx* ctx, apse_results_layout* out);
/* Copies the results block to host memory (layout->bytes) and waits for the copy (the implicit device sync of the reference loop:
 * rcnn_tracker.py:132 `.cpu()`, mask_utils.py:19,23,38 `.item()`).  = _begin + _end. */
int apse_read_results(apse_ctx* ctx, void* host_dst, size_t bytes, void* stream);
/* The same in two halves: _begin enqueues the copy (and an event behind it) and returns; _end waits for that copy only -- work the
 * caller enqueued on the stream in between is not waited for.  That may be the NEXT frame's apse_preprocess_frames AND its whole
 * forward: nothing in it depends on this frame, stream order keeps this copy in front of everything the forward overwrites, and the
 * mask bit planes alternate between two sets per forward, so apse_copy_mask_window after _end still reads THIS frame's masks.
 * host_dst must stay valid and unread until _end returns.  The closest-point table is decoded on the host inside _end (the device
 * leaves (distance, pixel index) keys in that field). */
int apse_read_results_begin(apse_ctx* ctx, void* host_dst, size_t bytes, void* stream);
int apse_read_results_end(apse_ctx* ctx, void* host_dst);
/* Copies detection i's mask window (rows rect.y0..y1, 64-bit words (x0>>6)..((x1+63)>>6)) to dst_dev: the masks of the forward whose
 * results were last read (apse_read_results / _begin), or of the last forward when no read has been started since. */
int apse_copy_mask_window(apse_ctx* ctx, int det, int x0, int y0, int x1, int y1, uint64_t* dst_dev, void* stream);
/* The same for n detections in ONE launch (n <= 100): window k = detection dets[k], rect rects[k] = x0,y0,x1,y1, written as
 * (y1 - y0) rows of ((x1 + 63) >> 6) - (x0 >> 6) words at dst_dev + dst_word_offsets[k].  Host arrays, read before the call returns. */
int apse_copy_mask_windows(apse_ctx* ctx, int n, const int* dets_host, const int* rects_host, uint64_t* dst_dev,
                           const long long* dst_word_offsets_host, void* stream);
/* Named internal tensor -> caller buffer as NCHW f32 (p2..p6, res2..res5, stem): the feature dict
 * TrackRCNN.inference returns (track_rcnn.py:57-58).  dims (B,C,H,W) via apse_feature_shape. */
int apse_feature_shape(apse_ctx* ctx, const char* name, int* chw3);
int apse_export_feature(apse_ctx* ctx, const char* name, float* dst_nchw_dev, int batch, void* stream);
/* Debug/parity taps: copy a named intermediate (device, raw layout) to dst_dev; returns element count in *count. */
int apse_debug_tensor(apse_ctx* ctx, const char* name, void* dst_dev, size_t max_bytes, size_t* bytes, void* stream);
/* Algorithmic FLOPs of one forward for `batch` images with the given proposal/detection totals. */
double apse_flops(apse_ctx* ctx, int batch, double proposals, double detections);

/* Per-kernel timing with HIP events on the caller's stream (one pair per convolution launch, collected at
 * apse_read_results).  out42 = [14 kernel shapes][3] = {sum ms, sum algorithmic FLOPs, launches}; shapes
 * 0..3 = 128x128, 64x64, 128x32, 128x64 implicit-GEMM tiles, 4..7 their 32-deep / 8-wave variants, 8 = 256x128,
 * 9 / 10 = the memory-streaming 1x1 kernels, A strip resident / streamed, 11 = the LDS-DMA
 * 256x128 kernel of the 16-bit modes, 12 = the fused stem + max-pool kernel of the 16-bit modes, 13 = the <= 16-channel
 * head kernel (a split-K launch includes its reduce pass).  With the two-half read a forward enqueued between
 * apse_read_results_begin and _end keeps its own event pairs: _end accounts exactly the forward whose results it returns. */
int apse_profile(apse_ctx* ctx, int enable);
int apse_profile_read(apse_ctx* ctx, double* out42, int reset);

/* ---- stage-level operators (stateless; used by the parity tests and by host-side helpers) ---- */
typedef struct apse_conv_desc {
    int B, H, W, Cin;       /* NHWC input, Cin a power of two >= 4 */
    int Cout, KH, KW, stride, pad;
    int relu;
    int res_mode;           /* 0 none, 1 same-shape residual, 2 nearest-2x upsampled residual */
    int cfg;                /* -1 auto, else tile shape 0..8: 128x128, 64x64, 128x32, 128x64 (64-deep steps for the small
                               ones), 4/5 = 64x64 / 128x32 with 32-deep steps, 6/7 = 64x64 with 8 waves in two k groups, 8 = 256x128
                               (16-bit operands stored 16-bit only; otherwise 128x128) */
    int splitk;             /* 0 auto */
    int prec;               /* 0 f32 MFMA; 1 bf16 / 2 f16 MFMA with f32 accumulate: x must be STORED in that type (x_st == prec),
                               filter rows whole 64-element steps, no fuse_reduce -- anything else is refused */
    int fuse_reduce;        /* split-K: 1 = last-arriving block reduces in the launch, 0 = separate reduce kernel */
    int x_st, res_st, y_st; /* storage type of x / residual / y: 0 f32, 1 bf16, 2 f16 (16-bit tensors need C % 8 == 0;
                               with prec 1/2 a 16-bit x must be stored in the operand type) */
} apse_conv_desc;
size_t apse_conv_packed_elems(const apse_conv_desc* d);
/* OIHW host filter (+ optional per-channel scale) -> packed host filter for apse_conv2d. */
int apse_conv_pack_weight(const apse_conv_desc* d, const float* w_oihw, int cin_real, const float* scale, float* packed);
int apse_conv2d(const apse_conv_desc* d, const float* x_dev, const float* w_packed_dev, const float* bias_dev,
                const float* res_dev, float* y_dev, float* ws_dev, size_t ws_bytes, void* stream);
int apse_maxpool3x3s2(const float* x_dev, float* y_dev, int B, int H, int W, int C, void* stream);
/* Same on a storage type (0 f32, 1 bf16, 2 f16): the form the 16-bit modes run after the stem. */
int apse_maxpool3x3s2_typed(const void* x_dev, void* y_dev, int B, int H, int W, int C, int storage, void* stream);
/* ROIAlignV2 over 4 levels (NHWC, C = 256); rois [n][4], batch index roi/per_img, all rois live. */
int apse_roi_align(const float* const* feats_dev, const int* hs, const int* ws, const float* rois_dev, int n, int per_img,
                   int out_size, float* out_dev, void* stream);
/* Same with the maps and the output in a storage type (0 f32, 1 bf16, 2 f16): the form the 16-bit modes run (two map cells
   per load instruction); arithmetic is f32 either way. */
int apse_roi_align_typed(const void* const* feats_dev, const int* hs, const int* ws, const float* rois_dev, int n, int per_img,
                         int out_size, int storage, void* out_dev, void* stream);
int apse_roi_pool(const float* feat_dev, int H, int W, const float* rois_dev, const int* roi_img_dev, int n, int out_size,
                  float scale, float* out_dev, void* stream);
/* Generic per-category NMS + ranking on [n] boxes (category = cat_dev[i] given as entry % cat_mod or / cat_div). */
int apse_nms_rank(const float* boxes_dev, const float* scores_dev, const int* valid_dev, int n, int cat_div, int cat_mod,
                  int ncat, float thr, int topk, float* out_boxes_dev, float* out_scores_dev, int* out_index_dev,
                  int* out_count_dev, void* stream);
/* mask_utils on a dense bool frame (uint8 0/1): centroid (1-based floor) and closest point to (px, py). */
int apse_mask_centroid_dense(const uint8_t* mask_dev, int H, int W, int* out_xy_mass_host3, void* stream);
int apse_mask_closest_dense(const uint8_t* mask_dev, int H, int W, float px, float py, int* out_xy_host2, void* stream);
/* F.normalize(p=2) rows and the squared-distance matrix of rcnn_tracker.py:192-221. */
int apse_l2_normalize(const float* x_dev, float* y_dev, int n, int D, void* stream);
int apse_sqdist(const float* a_dev, const float* b_dev, int O, int N, int D, float* out_dev, void* stream);
/* PIL resize + normalise as a stand-alone op (tables as in apse_set_resize_tables, device pointers). */
/* preprocess_img (visualize_uav.py:56-71): cv2.undistort + Lab-L gamma on u8 BGR frames [B][H][W][3].
 * mtx3x3 row-major, dist up to 14 coefficients (k1 k2 p1 p2 k3 k4 k5 k6 s1..s4 tx ty; tilt must be 0),
 * lut256_host = the 256-entry L-channel table (HOST memory since round 3: the Lab step is integer arithmetic on tables derived
 * from it on the host, see apse_lab_tables_host).  Either stage can be disabled. */
int apse_undistort_gamma(const uint8_t* src_dev, uint8_t* dst_dev, int B, int H, int W, const double* mtx3x3_host,
                         const double* dist_host, int ndist, const uint8_t* lut256_host, int do_undistort, int do_gamma,
                         void* stream);
/* Host only (no GPU touched): the integer tables of the Lab step (cvtColor RGB2LAB / LAB2RGB of visualize_uav.py:63-69 in the
 * manner of OpenCV's 8-bit path; layout = struct LabTables of csrc/preproc_pixel.h) for a 256-entry L table, written to `out`.
 * Returns the size of the block in bytes (call with out = NULL to query it).  tests/test_host_logic.py compares every entry
 * with the oracle's numpy-built tables. */
size_t apse_lab_tables_host(const uint8_t* lut256_host, void* out, size_t cap);
int apse_resize_normalize(const uint8_t* frames_dev, uint8_t* tmp_dev, float* out_nhwc4_dev, uint8_t* resized_u8_dev,
                          const int* hb_dev, const int* hc_dev, int hk, const int* vb_dev, const int* vc_dev, int vk, int B,
                          int H, int W, int OH, int OW, int PH, int PW, const float* mean3_host, void* stream);

/* ---- host-only: native replay of the sequential association from per-frame records (rank 0 of a sharded run).
 * Same rules as RcnnTracker.associate_detections_to_objects / ObjectInstances / generate_log_oneline
 * (rcnn_tracker.py:122-147, object_instances.py:48-162, visualize_uav.py:117-141); no GPU is touched. */
typedef struct apse_replay apse_replay;
apse_replay* apse_replay_create(int host_id, int embed_dim, float dist_thresh, int max_unseen_frames);
void apse_replay_destroy(apse_replay* r);
int apse_replay_step(apse_replay* r, int frame_idx, int n, const float* emb_host, const int* centroid_host,
                     const int* closest_host, char* line_out, int line_cap, int* det_ids_out);
/* nrec records laid end to end in the wire format of the sharded gather (apse_uav_amd/sharding.py::pack_record: count-prefixed,
 * `total_floats` in all; a frame with n detections is 1 + 13 n + 2 n^2 + n * embed_dim floats): one call for a whole run;
 * lines separated by '\n'; returns bytes written. */
long long apse_replay_packed(apse_replay* r, const float* records_host, long long total_floats, int nrec, int dets_per_image,
                             int first_frame, char* lines_out, long long cap);
int apse_replay_max_id(const apse_replay* r);
int apse_replay_next_id(const apse_replay* r);


/* ---- host-only: staging copy of the ingest path.  memcpy(dst, src, bytes) split over a small persistent pool of threads
 * (the caller's thread takes one part): a pageable frame from cv2.VideoCapture.read (visualize_uav.py:188-191) into the pinned
 * buffer its H2D starts from.  threads <= 1 or < 1 MB: plain memcpy.  Calls are serialised. */
int apse_host_copy(void* dst, const void* src, size_t bytes, int threads);

#ifdef __cplusplus
}
#endif
#endif
