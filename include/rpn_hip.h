/*
 * rpn_hip.h -- C ABI of librpn_hip.so: the MI355X (gfx950) Region Proposal Network
 * forward / proposal path.
 *
 * The reference (FurkanOM/tf-rpn) is pure Python on TensorFlow 2.0 and has no FFI of
 * its own; the boundary it offers is a handful of Python call signatures.  Every entry
 * point below names the reference interface (file:line under /root/reference) it
 * replaces; tf_rpn_amd/ binds them with ctypes under the reference's own function
 * names (INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - all `d_` pointers are DEVICE pointers (hipMalloc / torch.cuda storage), float32,
 *     row-major, box order [y1, x1, y2, x2]; the caller allocates every output.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every call
 *     is asynchronous on that stream and never synchronises the device.
 *   - return value: 0 on success, a negative rpn_status otherwise; rpn_last_error()
 *     returns a thread-local human-readable message for the last failure.
 *   - there is NO CPU fallback: without a HIP device every compute call fails with
 *     RPN_ERR_NO_DEVICE.
 *
 * Environment knobs of the library (read once per process).  These are ALL the names the
 * shipped library reads (tests/test_host.py compares this list with the binary's strings);
 * every setting computes the same arithmetic contract -- integer / index outputs bit-exact,
 * floats within the documented bound -- they select between implementations:
 *   RPN_KNOB RPN_B1_FUSE     (1)  VGG16 block 1 as one launch under f16x3; 0: two kernels
 *   RPN_KNOB RPN_HEAD_SPLITK (1)  rpn_reg | rpn_cls on the split-K head kernel; 0: generic f32 implicit GEMM
 *   RPN_KNOB RPN_KSPLIT      (1)  rpn_conv (split-precision modes) as a K tree: four fixed leaves of K, value (l0 + l1) +
 *                                 (l2 + l3) at every batch size, computed by 1, 2 or 4 workgroups per tile as the grid
 *                                 allows (same bits); 0: one accumulation chain, one workgroup per tile
 *   RPN_KNOB RPN_S16_DYN     (0)  1: dynamic tile queue in the persistent split-precision conv kernel
 *   RPN_KNOB RPN_S16_C64     (0)  1: block1_conv2 on the persistent kernel's 64-wide tiles
 *   RPN_KNOB RPN_NMS_LINEAR  (1)  NMS band selection from the linear score histogram; 0: radix select +
 *                                 bitonic sort only; 2: histogram select, sorted the old way
 *   RPN_KNOB RPN_NMS_CLUSTER (0 = automatic)  workgroups per (image, class) that share the NMS band selection
 *                                 when few images have many anchors; 1: always one workgroup
 *   RPN_KNOB RPN_NMS_PRUNE   (1)  NMS candidates meet only the selected boxes whose AREA allows the IoU threshold
 *                                 (area-ordered copy of the selected list); 0: the whole list
 *   RPN_KNOB RPN_MN_FUSE     (1)  MobileNetV2: one launch per inverted-residual block; 0: layer by layer
 *   RPN_KNOB RPN_MN_X3       (1)  MobileNetV2 under f16x3: 16-bit MFMA GEMMs inside the fused blocks; 0: f32 MFMA
 * Kernel / tile selection switches for A/B timing and the timing experiments whose results are
 * wrong on purpose (RPN_NMS_STOP, RPN_IOU_EXP, RPN_SPLIT_*, RPN_IOU_*, ...) exist only in a
 * laboratory build (`make -C tf_rpn_amd/csrc lab` -> librpn_hip_lab.so, -DRPN_LAB); the product
 * library does not contain their names.
 */
#ifndef RPN_HIP_H
#define RPN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPN_ABI_VERSION 1

typedef enum rpn_status {
    RPN_OK = 0,
    RPN_ERR_INVALID = -1,      /* bad argument (shape, null pointer, unsupported size) */
    RPN_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime error */
    RPN_ERR_WORKSPACE = -3,    /* workspace too small */
    RPN_ERR_UNSUPPORTED = -4   /* configuration outside the implemented range */
} rpn_status;

typedef enum rpn_backbone { RPN_BACKBONE_VGG16 = 0, RPN_BACKBONE_MOBILENET_V2 = 1 } rpn_backbone;

/* arithmetic of the conv stack.  F32: exact float32 MFMA (v_mfma_f32_32x32x2_f32), bit-for-bit
 * an ordered fmaf chain.  BF16X3 / F16X3: in the 3x3 stride-1 layers every float32 operand is carried
 * as hi + lo 16-bit halves (bfloat16 / float16) and each product is formed as
 * hi*hi + hi*lo + lo*hi on the 16-bit MFMA with float32 accumulation (product error ~2^-16 / ~2^-21
 * relative; measured against the 1e-4 parity bound in tests/, never assumed).  F16X3 requires
 * |activation| < 65504.  All other layers stay on the float32 kernels. */
typedef enum rpn_precision { RPN_PRECISION_F32 = 0, RPN_PRECISION_BF16X3 = 1, RPN_PRECISION_F16X3 = 2,
                             RPN_PRECISION_F32W = 3 /* float32 Winograd for the 3x3 stride-1 convs -- F(4x4,3x3) where its tiles
                                                       fill the chip, F(2x2,3x3) otherwise, chosen per model handle: float32
                                                       operands and accumulation, 1/4 / 1/2.25 of the multiply-adds, another
                                                       summation order (1e-4 contract; measured 3e-6 ... 6e-6 on the heads) */
} rpn_precision;

int rpn_abi_version(void);
const char *rpn_last_error(void);
/* number of visible HIP devices (0 when there is none); never fails */
int rpn_device_count(void);
/* Stream diagnostic: enqueue ONE wave on `stream` that sleeps for `microseconds` (0 .. 10000) of the device's real-time counter.
 * HIP maps streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4); two streams that share a queue run one behind
 * the other.  The predictor loop (predictor.py:46-60 restated with the NMS of batch k beside the convs of batch k + 1) needs its
 * two streams on DIFFERENT queues: spin both and compare the elapsed time (tf_rpn_amd/predictor.py: _streams_overlap). */
int rpn_stream_spin(void *stream, int microseconds);

/* ------------------------------------------------------------------------------------
 * generate_anchors(hyper_params) -> (A,4)            utils/bbox_utils.py:23-46 (+ :3-21)
 *   A = feature_map_shape^2 * n_ratios * n_scales, flat index (y*F + x)*K + k,
 *   k = scale_idx*n_ratios + ratio_idx.  ratios / scales are the python floats of
 *   hyper_params["anchor_ratios"/"anchor_scales"] (HOST arrays of double).
 * ---------------------------------------------------------------------------------- */
int rpn_generate_anchors(double img_size, int feature_map_shape, const double *ratios, int n_ratios,
                         const double *scales, int n_scales, float *d_anchors, void *stream);

/* ------------------------------------------------------------------------------------
 * get_bboxes_from_deltas(anchors, deltas) -> (B,A,4)          utils/bbox_utils.py:72-96
 *   fused with the caller's `rpn_bbox_deltas *= variances`       predictor.py:55
 *   d_anchors is (A,4) when anchors_batched == 0 (the predictor.py:56 call shape), else (B,A,4).
 *   variances: HOST pointer to 4 floats, or NULL for no scaling.
 * ---------------------------------------------------------------------------------- */
int rpn_decode(const float *d_anchors, int anchors_batched, const float *d_deltas, const float *variances,
               int B, int A, float *d_boxes, void *stream);

/* get_deltas_from_bboxes(bboxes, gt_boxes) -> (B,A,4)        utils/bbox_utils.py:98-124 */
int rpn_encode(const float *d_bboxes, int bboxes_batched, const float *d_gt_boxes, int B, int A,
               float *d_deltas, void *stream);

/* normalize_bboxes / denormalize_bboxes(bboxes, height, width)     utils/bbox_utils.py:152-166 / :168-182
 *   nboxes boxes of 4 floats; denormalize != 0 multiplies and rounds half-to-even (tf.round), else divides */
int rpn_scale_boxes(const float *d_boxes, long long nboxes, float height, float width, int denormalize,
                    float *d_out, void *stream);

/* ------------------------------------------------------------------------------------
 * generate_iou_map(bboxes, gt_boxes) -> (B,A,G)              utils/bbox_utils.py:126-150
 *   d_bboxes is (A,4) when bboxes_batched == 0 (the utils/train_utils.py:106 call shape).
 * ---------------------------------------------------------------------------------- */
int rpn_iou_map(const float *d_bboxes, int bboxes_batched, int A, const float *d_gt_boxes, int B, int G,
                float *d_iou, void *stream);

/* ------------------------------------------------------------------------------------
 * non_max_suppression(pred_bboxes, pred_labels, **kwargs)      utils/bbox_utils.py:48-70
 *   == tf.image.combined_non_max_suppression (TF 2.0.0 kernel semantics).
 *   d_boxes (B,N,q,4) with q in {1,C}; d_scores (B,N,C).
 *   Outputs (M = max_total; the pad_per_class rule is applied by the caller):
 *   d_out_boxes (B,M,4), d_out_scores (B,M), d_out_classes (B,M) float32,
 *   d_out_valid (B) int32, and -- not returned by TF, for parity checks -- d_out_idx (B,M)
 *   int32 box indices (-1 padded; may be NULL).
 *   d_workspace: rpn_nms_workspace_bytes(...) bytes of device scratch (may be NULL if 0).  For C > 1 the
 *   staging part is REQUIRED (RPN_ERR_WORKSPACE otherwise).  For few (image, class) pairs with many
 *   candidates (<= 32 pairs, N >= 16384) the size also covers the "cluster" scratch with which several
 *   workgroups per pair share the passes over the scores; a call that passes less (or NULL) still
 *   succeeds with identical results on one workgroup per pair.  rpn_decode_nms takes the same scratch:
 *   rpn_nms_workspace_bytes(B, A, 1, max_total, max_total).
 * ---------------------------------------------------------------------------------- */
size_t rpn_nms_workspace_bytes(int B, int N, int C, int max_per_class, int max_total);
int rpn_combined_nms(const float *d_boxes, const float *d_scores, int B, int N, int q, int C,
                     int max_per_class, int max_total, float iou_threshold, float score_threshold,
                     int clip_boxes, float *d_out_boxes, float *d_out_scores, float *d_out_classes,
                     int32_t *d_out_idx, int32_t *d_out_valid, void *d_workspace, size_t workspace_bytes,
                     void *stream);

/* ------------------------------------------------------------------------------------
 * predictor.py:52-56 + NMS in one launch sequence: (reg,cls) head outputs -> proposals.
 *   d_deltas (B,A,4) raw head output, d_scores (B,A) objectness, d_anchors (A,4);
 *   boxes are decoded lazily inside the NMS kernel (never materialised in HBM).
 * ---------------------------------------------------------------------------------- */
int rpn_decode_nms(const float *d_anchors, const float *d_deltas, const float *variances,
                   const float *d_scores, int B, int A, int max_total, float iou_threshold,
                   float score_threshold, int clip_boxes, float *d_out_boxes, float *d_out_scores,
                   int32_t *d_out_idx, int32_t *d_out_valid, void *d_workspace, size_t workspace_bytes,
                   void *stream);

/* ------------------------------------------------------------------------------------
 * calculate_rpn_actual_outputs(anchors, gt_boxes, gt_labels, hyper_params)   utils/train_utils.py:84-144
 *   (+ randomly_select_xyz_mask :50-65) -- the consumer of generate_iou_map; the (B,A,G) map is never written.
 *   d_gt_labels (B,G) int32, -1 = padding.  d_random_pos / d_random_neg (B,A) int32 >= 1 replace the two
 *   tf.random.uniform draws (priority order: larger first, ties lower index).  variances: HOST pointer, 4 floats.
 *   Outputs: d_bbox_deltas (B,A,4) = encoded deltas / variances (zero for non-positive anchors),
 *   d_bbox_labels (B,A) float32 in {1, 0, -1} (caller views it as (B,F,F,K)).
 * ---------------------------------------------------------------------------------- */
size_t rpn_targets_workspace_bytes(int B, int A, int G);
int rpn_rpn_targets(const float *d_anchors, const float *d_gt_boxes, const int32_t *d_gt_labels, int B, int A, int G,
                    int total_pos, int total_neg, const float *variances, const int32_t *d_random_pos,
                    const int32_t *d_random_neg, float *d_bbox_deltas, float *d_bbox_labels, void *d_workspace,
                    size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * preprocessing(image_data, final_height, final_width)            utils/data_utils.py:25-28
 *   one image: uint8 (H,W,3) -> float32 (out_h,out_w,3) in [0,1]: tf.image.convert_image_dtype (x * 1/255),
 *   tf.image.resize (bilinear, half-pixel centres, no antialias), optional tf.image.flip_left_right
 *   (flip_horizontally, utils/data_utils.py:63).  d_out may point into a (B,out_h,out_w,3) batch.
 * ---------------------------------------------------------------------------------- */
int rpn_preprocess_image(const unsigned char *d_img_u8, int H, int W, int out_h, int out_w, int flip,
                         float *d_out, void *stream);

/* ------------------------------------------------------------------------------------
 * get_model(hyper_params) -> rpn_model            models/rpn_vgg16.py:6-22,
 *                                                 models/rpn_mobilenet_v2.py:6-22
 * rpn_model.predict_on_batch(imgs) -> [reg, cls]  predictor.py:50
 *   The handle owns the packed device weights and the activation workspace.
 *   Weights are addressed by the Keras layer names the reference's
 *   load_weights(by_name=True) uses (predictor.py:44): "block1_conv1" ... "rpn_conv",
 *   "rpn_cls", "rpn_reg"; MobileNetV2: "Conv1", "bn_Conv1", "block_3_depthwise_BN", ...
 * ---------------------------------------------------------------------------------- */
typedef struct rpn_model rpn_model;

int rpn_model_create(int backbone, int img_size, int anchor_count, int precision, int max_batch,
                     rpn_model **out);
void rpn_model_destroy(rpn_model *m);
/* feature-map side F (31 for vgg16@500, 32 for mobilenet_v2@500) and head widths */
int rpn_model_feature_map_shape(const rpn_model *m);
/* number of weight-carrying layers, and the i-th layer's name / kernel shape (R,S,Cin,Cout) /
 * kind: 0 conv+bias, 1 conv (no bias) followed by BatchNorm, 2 depthwise conv followed by BatchNorm */
int rpn_model_num_layers(const rpn_model *m);
int rpn_model_layer_info(const rpn_model *m, int i, char *name, int name_len, int shape[4], int *kind);
/* Keras name of the BatchNormalization layer that follows layer i ("" when there is none) */
int rpn_model_layer_bn_name(const rpn_model *m, int i, char *name, int name_len);
/* device memory the handle needs (packed weights; activation arena for max_batch images) */
int rpn_model_memory_bytes(const rpn_model *m, size_t *weights, size_t *arena);
/* keep every intermediate activation alive (unique arena offsets) so that rpn_model_get_activation can
 * read any layer after a forward pass; must be called before the first set_layer / forward */
int rpn_model_keep_activations(rpn_model *m, int keep);
/* HOST pointers: kernel HWIO float32 (depthwise: (R,S,C,1)); bias (Cout) or NULL;
 * bn = {gamma, beta, moving_mean, moving_variance} each (Cout) or NULL (folded on the host, eps 1e-3) */
int rpn_model_set_layer(rpn_model *m, const char *name, const float *kernel, const float *bias,
                        const float *bn_gamma, const float *bn_beta, const float *bn_mean,
                        const float *bn_var);
/* d_imgs (B,img,img,3) NHWC float32 in [0,1] (utils/data_utils.py:25-26);
 * d_reg (B,F,F,4K), d_cls (B,F,F,K) -- the reference's output order is [reg, cls]
 * (models/rpn_vgg16.py:21).
 * A handle owns ONE activation arena and one set of inter-workgroup scratch (K-split partials and tickets): the forwards
 * of a handle must be ordered on one stream (or by events); for concurrent forwards on several streams create one handle
 * per stream (the reference's Keras model is not re-entrant either).
 * Batch invariance: an image's outputs are bit-identical at every batch size B <= max_batch OF ONE HANDLE (split factors follow
 * the grid, summation trees are fixed).  Two handles created with different max_batch may choose different weight packings
 * (MobileNetV2 under F16X3: block 3's chunk size follows the grid at max_batch) and then agree within the float bound, not
 * bit for bit. */
int rpn_model_forward(rpn_model *m, const float *d_imgs, int B, float *d_reg, float *d_cls, void *stream);
/* Sticky status flags of the forwards run so far (no reference counterpart: TF computes in float32 throughout).
 * RPN_STATUS_F16_RANGE: under RPN_PRECISION_F16X3 some activation did not fit float16 (|x| > 65504 or non-finite) when
 * it was written in split form -- outputs of that forward are invalid (use BF16X3 or F32 for such weights).  The flag
 * is raised on the device by the kernel that hits it; a forward never reads it back (no host synchronisation on the
 * hot path).  This call copies the word to the host (synchronising on `stream`) and clears it when `reset` != 0. */
#define RPN_STATUS_F16_RANGE 1u
int rpn_model_status(rpn_model *m, unsigned *flags, int reset, void *stream);
/* debug / test hook: copy the activation of layer `name` (NHWC float32) into d_out */
int rpn_model_get_activation(rpn_model *m, const char *name, float *d_out, size_t out_bytes, int shape[4],
                             void *stream);
/* algorithmic FLOPs (2*MACs) of one image through the conv stack */
double rpn_model_flops_per_image(const rpn_model *m);
/* per-op timing with HIP events recorded on the caller's stream around every launch of a forward:
 * keep the events of the last n_forwards forwards (0 = off), run forwards, then read the mean elapsed
 * milliseconds of each op over the kept forwards.
 * rpn_model_op_info names op i, the kernel that runs it and its algorithmic FLOPs / HBM bytes per image. */
int rpn_model_set_profiling(rpn_model *m, int n_forwards);
/* time only the ops with mask[i] != 0 (n = rpn_model_num_ops; NULL: every op); untimed ops read back as 0 ms */
int rpn_model_set_profiling_mask(rpn_model *m, const unsigned char *mask, int n);
/* with a mask: time ONE marked op per forward, round robin (2 events per forward); rpn_model_get_profile then averages
   each op over the forwards in which it was the one timed */
int rpn_model_set_profiling_rotate(rpn_model *m, int on);
/* arithmetic of op i's matrix work: an rpn_precision value (RPN_PRECISION_F32 for the float32-MFMA / vector-ALU kernels,
 * the model's split precision for the kernels that form each product from three 16-bit MFMAs); -1 for a bad index.
 * A 2x2 max-pool that runs inside the previous conv's epilogue is reported by rpn_model_op_info with the kernel name
 * "fused:maxpool_split" and zero bytes: it is not a launch. */
int rpn_model_op_arith(const rpn_model *m, int i);
int rpn_model_num_ops(const rpn_model *m);
int rpn_model_op_info(const rpn_model *m, int i, char *name, int name_len, char *kernel, int kernel_len,
                      double *flops_per_image, double *bytes_per_image);
int rpn_model_get_profile(rpn_model *m, float *ms, int n, int *n_forwards);

/* ------------------------------------------------------------------------------------
 * single conv layer, for kernel-level parity tests and micro-benchmarks.
 *   x (B,H,W,Cin) NHWC, w HWIO (device), bias (Cout, device, may be NULL).
 *   pad_t / pad_l: zero rows/cols added before the first row/col; OH/OW given by the caller.
 *   act: 0 linear, 1 relu, 2 sigmoid, 3 relu6.
 * ---------------------------------------------------------------------------------- */
int rpn_conv2d(const float *d_x, int B, int H, int W, int Cin, const float *d_w, const float *d_bias,
               int R, int S, int Cout, int stride, int pad_t, int pad_l, int OH, int OW, int act,
               int precision, float *d_out, void *stream);
int rpn_maxpool2x2(const float *d_x, int B, int H, int W, int C, float *d_out, void *stream);
/* depthwise 3x3 (MobileNetV2): d_w (3,3,C), d_bias (C) or NULL, C % 4 == 0 */
int rpn_dwconv3x3(const float *d_x, int B, int H, int W, int C, const float *d_w, const float *d_bias,
                  int stride, int pad_t, int pad_l, int OH, int OW, int act, float *d_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RPN_HIP_H */
