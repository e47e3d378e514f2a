// conv_kernels.h -- host-side interface of the conv / pool kernels (internal to librpn_hip.so).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "rpn_knobs.h"

namespace rpn {

enum Act { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SIGMOID = 2, ACT_RELU6 = 3 };

// Packed weight matrix of one dense conv layer, K-major: rows x cout_pad floats, zero padded.
//   fast layout  (Cin % 4 == 0): row = tap * cin_pad + c,  cin_pad = round_up(Cin, 16)
//   generic layout (any Cin)   : row = tap * Cin + c,      rows padded to a multiple of 16
// cout_pad = round_up(Cout, 32).
struct PackedShape {
    int R, S, Cin, Cout;
    int generic;      // 1 -> generic (flattened-K) layout
    int cin_pad;      // fast layout only
    int rows;         // total rows of the packed matrix
    int cout_pad;
    __host__ __device__ size_t floats() const { return (size_t)rows * cout_pad; }
};

PackedShape packed_shape(int R, int S, int Cin, int Cout);

// pack HWIO (R,S,Cin,Cout) -> packed, on the host (scale[n] multiplies output channel n; may be null)
void pack_weights_host(const PackedShape &ps, const float *hwio, const float *scale, float *dst);
// same on the device (d_hwio, d_dst device pointers)
void pack_weights_device(const PackedShape &ps, const float *d_hwio, float *d_dst, hipStream_t stream);

struct ConvArgs {
    const float *x;         // (B,H,W,Cin) NHWC
    const float *w;         // packed weights (PackedShape)
    const float *bias;      // (Cout) or null
    const float *residual;  // (B,OH,OW,Cout) added before the activation, or null
    float *out;             // (B,OH,OW,ld1): channels [0, split)
    float *out2;            // (B,OH,OW,ld2): channels [split, Cout), or null
    int B, H, W, Cin, OH, OW, Cout;
    int R, S, stride, pad_t, pad_l;
    int act, act2, split;   // split == Cout when there is a single output
    int ld1, ld2;
    PackedShape ps;
    int out_split;          // 0: float32 NHWC; 1 / 2: `out` is SPLIT16 with bfloat16 / float16 halves (needs Cout % 32 == 0,
                            // a single output, no residual): the consumer is a split-precision 3x3 conv
    unsigned *status;       // float16 range flag (out_split == 2), or null
    int pool;               // 1: `out` is the layer's output after MaxPooling2D(2, 2) 'valid', (B, OH/2, OW/2, ld1), pooled in the
                            // epilogue (needs a single float32 output, no residual, a monotone activation)
};

// dense convolution as an implicit GEMM on the f32 MFMA; returns hipGetLastError()
int conv_f32_tile_n(int B, int OH, int OW, int Cout);   // N tile (128 | 64 | 32) launch_conv_f32 picks for a layer
hipError_t launch_conv_f32(const ConvArgs &a, hipStream_t stream);
bool conv_f32_uses_dma(int tile_n, int generic);       // whether that tile runs on conv_igemm_f32_dma (LDS-DMA staging)

// RPN head (rpn_reg | rpn_cls 1x1 convs on a 512-channel float32 NHWC tensor of P pixels) as one split-K launch:
// w_packed = pack_head_weights_host(matrix [512][ld] whose columns are reg then cls) -- 512 * 16 * ceil(ncols / 16) floats;
// bias[ncols]; reg (P, n_reg) linear, cls (P, n_cls) sigmoid.  ncols = n_reg + n_cls <= 96.
bool rpn_head_supported(int Cin, int ncols);
void pack_head_weights_host(const float *w, int ld, int ncols, float *dst);
// n_slabs = 2 | 4: x is that many RAW partial-sum slabs of the preceding conv (slab_floats apart); the head adds them in TREE
// order -- s0 + s1, or (s0 + s1) + (s2 + s3) -- then + conv_bias, ReLU (rpn_conv's epilogue), while it loads its operand.
hipError_t launch_rpn_head(const float *x, long long P, const float *w_packed, const float *bias, int n_reg, int n_cls,
                           float *reg, float *cls, hipStream_t stream, int n_slabs = 1, long long slab_floats = 0,
                           const float *conv_bias = nullptr);

// ---- 3x3 stride-1 'same' conv as Winograd F(2x2, 3x3) on the float32 MFMA (conv_wino_kernels.hip; precision "f32w") ----
// variant: 2 = F(2x2, 3x3), 4 = F(4x4, 3x3) on 16 x 32-pixel x 64-channel tiles, 16 = F(4x4, 3x3) on 16 x 16-pixel x 128-channel tiles
// (Cout % 128 == 0), 8 = the 64-channel form with the input channels split over two workgroups per tile (wino_variant picks per
// layer from the grid at the handle's largest batch).
// u = pack_weights_wino_host(HWIO weights, variant): wino_weight_floats floats; out: (B,H,W,Cout) or, pool, (B,H/2,W/2,Cout).
// workspace (variant 8 only): wino_workspace_bytes bytes of device memory, zeroed ONCE by the owner (partial tiles + tickets; the
// kernel leaves the tickets at zero), private to one stream at a time.
bool wino_supported(int Cin, int Cout);                  // Cin % 8 == 0, Cout % 32 == 0
int wino_variant(int B, int H, int W, int Cin, int Cout);   // 16 | 4 | 8 | 2, from the grid and the CURRENT device's CU count
bool wino_launchable(int B, int H, int W, int Cin, int Cout, int variant);   // launch_conv3x3_wino's limits as a predicate
size_t wino_weight_floats(int Cin, int Cout, int variant);
size_t wino_workspace_bytes(int B, int H, int W, int Cin, int Cout, int variant);
void pack_weights_wino_host(const float *hwio, const float *scale, int Cin, int Cout, float *dst, int variant);
hipError_t launch_conv3x3_wino(const float *x, const float *u, const float *bias, float *out, int B, int H, int W, int Cin,
                               int Cout, int act, bool pool, hipStream_t s, int variant, void *workspace = nullptr);

// MaxPooling2D(2,2) 'valid' (floors odd sizes), NHWC, C % 4 == 0
hipError_t launch_maxpool2x2(const float *x, int B, int H, int W, int C, float *out, hipStream_t stream);

// depthwise 3x3 (+folded BN bias) + activation, NHWC; w is (3,3,C) with the BN scale folded in
hipError_t launch_dwconv3x3(const float *x, int B, int H, int W, int C, const float *w, const float *bias,
                            int stride, int pad_t, int pad_l, int OH, int OW, int act, float *out,
                            hipStream_t stream);

// ---- MobileNetV2 inverted-residual block in one launch (mnv2_block_kernels.hip) --------------------------------
// expand 1x1 + BN + ReLU6 -> depthwise 3x3 + BN + ReLU6 -> project 1x1 + BN (+ residual); `stem`: Conv1 (3x3 s2 from the
// image) takes the place of the expand stage (stem + expanded_conv block).  Weights (BatchNorm folded): we [KP][cexp]
// (KP = cin; stem: 28 = 27 im2col rows + a zero row), wd [9][cexp], wp [cexp][round_up(cout, 16)] + their biases.
bool ir_block_supported(int cin, int cexp, int cout, int stride, bool residual);
hipError_t launch_ir_block(const float *x, int B, int H, int W, int cin, int cexp, int cout, int stride, bool residual,
                           bool stem, int pad, int OH, int OW, const float *we, const float *be, const float *wd,
                           const float *bd, const float *wp, const float *bp, float *out, hipStream_t s);

// f16x3 variant for the low-resolution blocks (Cin = 64 / 96, stride 1; precision F16X3 only): both GEMMs on the 16-bit
// MFMA with hi + lo float16 operands.  we / wp: fragment-major images from pack_ir_x3_expand / _project (same byte
// counts as the float32 matrices), scale = 2^-shift of their power-of-two pre-scale.
bool ir_block_x3_supported(int cin, int cexp, int cout, int stride, bool residual);
void pack_ir_x3_expand(const float *w, int K, int CEXP, int shift, unsigned short *dst);
void pack_ir_x3_project(const float *w, int CEXP, int COUT, int shift, unsigned short *dst);
// depthwise weights [9][cexp] + bias [cexp] -> one 48-byte record per channel (what `wd` of launch_ir_block_x3 points to; `bd` unused)
size_t ir_x3_dw_floats(int cexp);
void pack_ir_x3_dw(const float *wd, const float *bd, int cexp, float *dst);
// scratch: ir_block_x3_scratch_floats() zero-initialised floats (partials + tickets of the K-split used on small grids), or
// null = never split.  The kernel leaves the tickets at zero.
size_t ir_block_x3_scratch_floats();
int ir_block_x3_ksplit(long long tiles);
hipError_t launch_ir_block_x3(const float *x, int B, int H, int W, int cin, int cexp, int cout, bool residual,
                              const void *we, const float *be, const float *wd, const float *bd, const void *wp,
                              const float *bp, float scale_e, float scale_p, float *out, unsigned *status, float *scratch,
                              hipStream_t s);

// pointwise conv (1 x 1) + bias + ReLU6 on the 16-bit MFMA (f16x3), float32 NHWC (P, cin) in -> SPLIT16 (P, cout) out:
// MobileNetV2's block_13_expand in front of rpn_conv.  w = pack_ir_x3_expand image, scale = 2^-shift.
bool pw_x3_supported(int cin, int cout);
hipError_t launch_pw_x3(const float *x, long long P, int cin, int cout, const void *w, const float *bias, float scale,
                        void *out, unsigned *status, hipStream_t s);

// f16x3 variant of the high-resolution blocks 1-3 (Cin = 16 / 24; mnv2_block_kernels.hip: ir_block_hrx3_kernel): the
// expand GEMM's K is zero-padded to one 32-deep step, so the fragment images are LARGER than the float32 matrices --
// ir_hrx3_*_floats give their sizes in floats.
bool ir_block_hrx3_supported(int cin, int cexp, int cout, int stride, bool residual);
// blocks 4, 5 are supported by both f16x3 block kernels: whether this one is the faster at the handle's largest grid
bool ir_block_hrx3_preferred(int cin, int stride, long long tiles_at_max_batch);
size_t ir_hrx3_expand_floats(int cexp);
// ce_ov: the block's chunk size when the model chose one (ir_hrx3_chunk_for: block 3, by the grid at the handle's max_batch);
// 0 = the kernel family's default.  The same value must go to the packers and to the launcher.
int ir_hrx3_chunk_for(int cin, int stride, long long tiles_at_max_batch);
size_t ir_hrx3_project_floats(int cin, int cexp, int cout, int stride, int ce_ov = 0);
void pack_ir_hrx3_expand(const float *w, int cin, int cexp, int stride, int shift, unsigned short *dst, int ce_ov = 0);
void pack_ir_hrx3_project(const float *w, int cin, int cexp, int cout, int stride, int shift, unsigned short *dst, int ce_ov = 0);
hipError_t launch_ir_block_hrx3(const float *x, int B, int H, int W, int cin, int cexp, int cout, int stride, bool residual,
                                int pad, int OH, int OW, const void *we, const float *be, const float *wd, const float *bd,
                                const void *wp, const float *bp, float scale_e, float scale_p, float *out, unsigned *status,
                                float *scratch, hipStream_t s, int ce_ov = 0);   // scratch: ir_block_x3_scratch_floats() (K tree of blocks 3 / 6), or null

// Device word into which the split-format writers launched from THIS host thread flag float16 range violations
// (RPN_STATUS_F16_RANGE); null = no reporting.  rpn_model_forward sets it around its launches.
void set_range_status(unsigned *p);
unsigned *range_status();

// ---- x3-split 16-bit MFMA path (conv_split_kernels.hip) ----------------------------------------
// SPLIT16 activation layout: [B][H][W][C/16][64 B] = {hi[0:8], lo[0:8], hi[8:16], lo[8:16]} (bf16 or f16).
// Split weights: [Cin/16][9][cout_pad][64 B], cout_pad = 64 for Cout <= 64, else round_up(Cout, 128).
inline int split_cout_pad(int Cout) { return Cout <= 64 ? 64 : (Cout + 127) / 128 * 128; }   // a multiple of the N tile
inline size_t split_weight_bytes(int Cin, int Cout) { return (size_t)9 * (Cin / 16) * split_cout_pad(Cout) * 64; }
// power-of-two pre-scale of the weights (f16 only; 0 for bf16) so that their lo halves stay normal
int split_weight_shift(const float *hwio, size_t count, bool f16);
void pack_weights_split_host(const float *hwio, const float *scale, int Cin, int Cout, int cout_pad, bool f16,
                             int shift, unsigned short *dst);
// 16x16x32-MFMA variant: weights in "split32" packing [Cin/32][9][cout_pad][128 B] (same byte count)
void pack_weights_split32_host(const float *hwio, const float *scale, int Cin, int Cout, int cout_pad, bool f16,
                               int shift, unsigned short *dst);
// (ktree: the layer as a K TREE -- conv3x3_split16_ktree_ok -- unsplit: float32 output, no pool, persistent 64-wide kernel)
hipError_t launch_conv3x3_split16(const void *x, const void *w, const float *bias, void *out, int B, int H, int W,
                                  int Cin, int Cout, int cout_pad, float out_scale, int act, bool out_f32, bool f16,
                                  bool pool, hipStream_t s, bool ktree = false);
// which 16x16x32 kernel launch_conv3x3_split16 picks for a layer: "dma,128" | "dma,64" (persistent LDS-DMA kernel,
// tile width in channels) or "reg,128" | "reg,64" (register-staged kernels); nullptr: not launchable
const char *conv3x3_split16_variant(int B, int H, int W, int Cin, int Cout, int cout_pad, bool pool);
// K tree of a layer whose consumer adds partial-sum slabs (rpn_conv -> the RPN head): the layer's value is (l0 + l1) + (l2 +
// l3) over four fixed leaves of K at every batch size, computed by one workgroup per tile (launch_conv3x3_split16, ktree) or by
// 2 / 4 (launch_conv3x3_split16_ksplit: slabs l0 + l1 | l2 + l3, or the four leaves; the head adds them in tree order).
// _ktree_ok: whether the layer can run that way at every batch up to B; _ksplit: the split factor (1 | 2 | 4) at batch B.
// Leaf boundaries of the K tree over `chunks` 32-channel slices: leaf k = slices [ktree_cut(k), ktree_cut(k + 1)).  The middle cut
// is EVEN, so that the two halves of a 2-way split are whole slice PAIRS (what the persistent LDS-DMA kernel walks): 18 slices
// (MobileNetV2's 576 channels) -> 4 4 5 5 (round 3: 4 5 4 5); 16 slices (VGG16) -> 4 4 4 4, unchanged.
__host__ __device__ inline int ktree_cut(int chunks, int k)
{
    return k <= 0 ? 0 : (k == 1 ? chunks / 4 : (k == 2 ? ((chunks / 2) & ~1) : (k == 3 ? 3 * chunks / 4 : chunks)));
}
bool conv3x3_split16_ktree_ok(int B, int H, int W, int Cin, int Cout, int cout_pad);
// whether the 2-way split of such a layer at batch B runs on the persistent LDS-DMA kernel (8 x 32 x 64 tiles, two workgroups
// per tile: grids of 65 .. 128 tiles -- configs[4]'s 64 x 64 feature map) instead of the register-staged 4 x 32 x 64 kernel
bool conv3x3_split16_ksplit_dma(int B, int H, int W, int Cin, int Cout, int cout_pad);
int conv3x3_split16_ksplit(int B, int H, int W, int Cin, int Cout, int cout_pad);
hipError_t launch_conv3x3_split16_ksplit(const void *x, const void *w, float *out, long long slab_floats, int B, int H, int W,
                                         int Cin, int Cout, int cout_pad, float out_scale, bool f16, int ksplit, hipStream_t s);
hipError_t launch_f32_to_split(const float *x, long long npix, int C, bool f16, void *out, hipStream_t s);
hipError_t launch_split_to_f32(const void *x, long long npix, int C, bool f16, float *out, hipStream_t s);
hipError_t launch_maxpool_split(const void *x, int B, int H, int W, int C, bool f16, void *out, hipStream_t s);
hipError_t launch_conv3x3_split(const void *x, const void *w, const float *bias, void *out, int B, int H, int W,
                                int Cin, int Cout, int cout_pad, float out_scale, int act, bool out_f32, bool f16,
                                bool pool, hipStream_t s, bool ktree = false);

// VGG16 block 1 (block1_conv1 + block1_conv2 + block1_pool) in one launch: the 64-channel full-resolution tensor stays
// on chip.  w1 = pack_weights_cin3_mfma_host records, w2 = pack_weights_split_host records; out: SPLIT16 (B,H/2,W/2,64).
hipError_t launch_vgg_block1(const float *img, const void *w1, const float *b1, float scale1, const void *w2,
                             const float *b2, float scale2, void *out, int B, int H, int W, bool f16, hipStream_t s);

// first layer (Cin = 3, 3x3): direct conv on the vector ALU; w (27,Cout) float32; out_fmt 0 = float32 NHWC,
// 1 = SPLIT16.  Cout % 16 == 0.
hipError_t launch_conv_cin3(const float *x, const float *w, const float *bias, void *out, int B, int H, int W,
                            int OH, int OW, int Cout, int stride, int pad_t, int pad_l, int act, int out_fmt, bool f16,
                            hipStream_t s);
// true when launch_conv_cin3 runs the layer on the float32 matrix pipe (conv_cin3_f32_mfma_kernel: float32 output, 64 channels,
// stride 1, 'same', ReLU / linear, the image's output below 2 GiB) instead of the vector-ALU kernel
bool conv_cin3_uses_f32_mfma(int B, int H, int W, int OH, int OW, int Cout, int stride, int pad_t, int pad_l, int act, int out_fmt);

// first layer on the matrix cores (split precisions only): Cout in {32, 64}; w = pack_weights_cin3_mfma_host output
void pack_weights_cin3_mfma_host(const float *hwio, const float *scale, int Cout, int cout_pad, bool f16, int shift,
                                 unsigned short *dst);
hipError_t launch_conv_cin3_mfma(const float *x, const void *w, const float *bias, void *out, int B, int H, int W,
                                 int OH, int OW, int Cout, int stride, int pad_t, int pad_l, int act, float out_scale,
                                 int out_fmt, bool f16, hipStream_t s);

}  // namespace rpn
