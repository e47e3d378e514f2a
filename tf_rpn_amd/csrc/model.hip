// model.hip -- native graph builder + executor for the two RPN models of the reference:
//   models/rpn_vgg16.py:15-21        VGG16 (to block5_conv3) + rpn_conv / rpn_cls / rpn_reg
//   models/rpn_mobilenet_v2.py:15-21 MobileNetV2 alpha=1 (to block_13_expand_relu) + the same head
// The Keras graphs themselves live in keras-applications 1.0.8 (not vendored by the reference);
// layer names follow Keras so that weights can be addressed the way load_weights(by_name=True)
// does (predictor.py:44).
//
// The handle owns (a) one device blob with every layer's packed weights (BatchNorm folded on
// the host at load time) and (b) an activation arena planned by tensor liveness.  A forward
// pass is a fixed sequence of kernel launches on the caller's stream: no allocation, no
// synchronisation, so it can be captured into a hipGraph by the caller.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "conv_kernels.h"
#include "rpn_common.h"

namespace rpn {

constexpr float kBnEps = 1e-3f;

enum OpKind { OP_CONV = 0, OP_DWCONV = 1, OP_POOL = 2, OP_HEAD = 3, OP_TOSPLIT = 4, OP_IRBLOCK = 5, OP_VGGB1 = 6 };

struct Tensor {
    std::string name;
    int H = 0, W = 0, C = 0;
    int def = -1, last_use = -1;       // op indices
    size_t offset = 0;                 // floats per image inside the arena (x max_batch)
    bool external = false;             // input images / head outputs
    bool split_fmt = false;            // SPLIT16 (hi/lo 16-bit records) instead of float32 NHWC; same byte size
    int slabs = 1;                     // room for this many copies (split-K partial sums of rpn_conv at small batches)
    size_t floats() const { return (size_t)H * W * C * slabs; }
};

struct Param {
    std::string name, bn_name;
    int kind = 0;                      // 0 conv+bias, 1 conv+BN, 2 depthwise+BN
    int R = 1, S = 1, Cin = 0, Cout = 0;
    int op = -1;
    int col_off = 0;                   // first output column inside the op's packed matrix
    bool loaded = false;
};

struct Op {
    int kind = OP_CONV;
    std::string name;
    int in = -1, out = -1, residual = -1;
    int Cin = 0, Cout = 0, R = 1, S = 1, stride = 1, pad_t = 0, pad_l = 0;
    int H = 0, W = 0, OH = 0, OW = 0;
    int act = ACT_LINEAR;
    PackedShape ps{};
    size_t w_off = 0, b_off = 0;       // floats into the weight blob
    std::vector<int> params;
    bool cin3 = false;                 // first layer (Cin = 3, 3x3): direct vector-ALU kernel ...
    bool cin3_mfma = false;            // ... or, in the split-precision modes, the one-step MFMA kernel
    bool split = false;                // runs on the x3-split 16-bit MFMA kernel (SPLIT16 input)
    bool k16 = false;                  // split conv on the 16x16x32-MFMA kernel ("split32" weight packing)
    bool out_f32 = false;              // split conv writing float32 NHWC (feeds the float32 head)
    bool f32_out_split = false;        // float32 implicit-GEMM conv writing SPLIT16 directly (its consumer is a split conv)
    bool wino = false;                 // precision F32W: 3x3 stride-1 'same' conv as float32 Winograd (conv_wino_kernels.hip)
    int wino_f = 2;                    // ... F(2x2, 3x3) or F(4x4, 3x3): wino_variant at the handle's max_batch
    float out_scale = 1.0f;            // 2^-shift of the pre-scaled split weights
    // OP_IRBLOCK (one fused MobileNetV2 block; Cin / Cout = block input / output channels, stride = the depthwise's):
    int cexp = 0;                      // expanded channels
    bool ir_stem = false;              // Conv1 + expanded_conv block: the expand stage is the 3x3 s2 stem conv
    bool ir_res = false;               // + block input
    int ir_pad = 0;                    // top/left padding of the depthwise (stem: of Conv1)
    size_t ir_off[6] = {0, 0, 0, 0, 0, 0};   // we, be, wd, bd, wp, bp (floats into the weight blob)
    bool ir_x3 = false;                // F16X3 only: the block's two GEMMs on the 16-bit MFMA (hi + lo float16 operands)
    bool ir_hrx3 = false;              // ... the high-resolution form (blocks 1-3: ir_block_hrx3_kernel; implies ir_x3)
    float ir_scale[2] = {1.0f, 1.0f};  // ... 2^-shift of the pre-scaled expand / projection weights
    int ir_ce = 0;                     // ... chunk size chosen for this handle (ir_hrx3_chunk_for; 0: the kernel family's default)
    // OP_VGGB1 (VGG16 block 1 in one launch) reuses ir_off[0..3] = w1, b1, w2, b2 and ir_scale[0..1] = the two 2^-shift
};

}  // namespace rpn

using namespace rpn;

struct rpn_model {
    int backbone = 0, img_size = 0, K = 0, precision = 0, max_batch = 0;
    bool use_split = false, f16 = false;   // precision BF16X3 / F16X3: eligible 3x3 convs use the split kernel
    bool wino = false;                     // precision F32W: eligible 3x3 convs use the float32 Winograd kernel
    int F = 0, feat_tensor = -1;
    bool keep_all = false;
    std::vector<Tensor> tensors;
    std::vector<Op> ops;
    std::vector<Param> params;
    size_t weight_floats = 0, arena_floats = 0;   // arena: per-image floats
    float *d_weights = nullptr, *d_arena = nullptr;
    unsigned *d_status = nullptr;          // RPN_STATUS_* flags raised by the kernels (sticky until rpn_model_status resets)
    // MobileNetV2 f16x3 blocks on small grids: partial projections + tickets (launch_ir_block_x3).  ONE buffer per handle,
    // shared by every such launch of a forward: the tickets are left at zero by the last arriver of each tile, which holds
    // only while the forwards of a handle run in stream order.  Two concurrent rpn_model_forward calls on the SAME handle
    // from different streams are not supported (rpn_hip.h says so); use one handle per stream.
    float *d_ksplit = nullptr;
    // f32w layers on F(4x4, 3x3) with split input channels (wino_variant 8): partial tiles + tickets (conv_kernels.h), one buffer for
    // all such layers of the handle, under the same one-stream-at-a-time rule
    void *d_wino_ws = nullptr;
    const float *last_input = nullptr;
    double flops = 0.0;
    // optional per-op timing: one hipEvent before the first op and one after every op
    int profiling = 0;                 // number of forwards whose events are kept (ring)
    std::vector<hipEvent_t> events;    // profiling x (ops + 1)
    std::vector<unsigned char> prof_mask;   // empty: time every op; else only the ops marked non-zero
    int prof_rotate = 0;                    // with a mask: each forward times ONE of the marked ops, round robin
    std::vector<int> prof_timed;            // per kept forward: the op timed in it (-1: every marked op)
    long long profiled_forwards = 0;
    std::vector<float> head_host;           // host copy of the fused head matrix [512][cout_pad] (source of the fragment-ordered copy)
};

namespace rpn {

// The 16x16x32-MFMA split kernels ("split32" weight packing) take every 3x3 layer whose channel count gives an even
// number of 32-channel slices (the persistent LDS-DMA kernel: Cin = 64, 128, 256, 512 in VGG16) and, register-staged,
// the other Cin >= 128 layers with 128-wide output tiles.  RPN_SPLIT_MFMA16=0 switches them off (32x32x16 kernels).
static bool use_mfma16(int Cin, int Cout, int H, int W, int B)
{
    static const int mode = RPN_LAB_KNOB("RPN_SPLIT_MFMA16", 1);
    static const int dma = RPN_LAB_KNOB("RPN_S16_DMA", 1);
    (void)H; (void)W; (void)B;
    if (mode == 0 || Cin % 32 != 0) return false;
    // block1_conv2 (64 -> 64, fused pool) stays on the 32x32x16 kernel: alone, the 64-wide persistent tiles are faster
    // (0.34 vs 0.37 ms), but this is the layer the overlapped NMS of the previous step runs beside, and a persistent
    // workgroup that has to share its CU delays a fixed share of the tiles (2737 vs 2814 images/s).  RPN_S16_C64=1
    // (with RPN_S16_DYN=1, the dynamic tile schedule) moves it over: 2800 images/s.
    static const int c64 = RPN_KNOB("RPN_S16_C64", 0);
    if (Cout <= 64 && !(c64 != 0 && dma != 0 && Cin % 64 == 0)) return false;
    if (dma && Cin % 64 == 0) return true;
    return Cin >= 128;
}

// RPN_HEAD_SPLITK=0: the head through the generic float32 implicit GEMM (one 128 x 32 tile walking K in 32 steps)
static bool head_splitk()
{
    static const int on = RPN_KNOB("RPN_HEAD_SPLITK", 1);
    return on != 0;
}

static int add_tensor(rpn_model *m, const std::string &name, int H, int W, int C, bool external = false)
{
    Tensor t;
    t.name = name; t.H = H; t.W = W; t.C = C; t.external = external;
    m->tensors.push_back(t);
    return (int)m->tensors.size() - 1;
}

static int add_param(rpn_model *m, int op, const std::string &name, const std::string &bn, int kind, int R, int S,
                     int Cin, int Cout, int col_off = 0)
{
    Param p;
    p.name = name; p.bn_name = bn; p.kind = kind; p.R = R; p.S = S; p.Cin = Cin; p.Cout = Cout; p.op = op;
    p.col_off = col_off;
    m->params.push_back(p);
    m->ops[op].params.push_back((int)m->params.size() - 1);
    return (int)m->params.size() - 1;
}

// F16X3 only: a pointwise conv + ReLU6 whose output goes to a split conv as SPLIT16 (MobileNetV2: block_13_expand in front
// of rpn_conv) runs on the 16-bit MFMA (pw_x3_kernel) instead of the float32 implicit GEMM.  Decided where it is used: the
// producer only learns that its output is SPLIT16 when its consumer is added.
static bool is_pw_x3(const rpn_model *m, const Op &op)
{
    static const int knob = RPN_KNOB("RPN_MN_X3", 1);      // read ONCE per process: weight packing (set_layer), kernel choice
                                                           // (forward) and op_info must agree whatever happens to the environment
    return op.kind == OP_CONV && op.f32_out_split && m->f16 && op.R == 1 && op.S == 1 && op.stride == 1 && op.act == ACT_RELU6 &&
           op.residual < 0 && !op.ps.generic && pw_x3_supported(op.Cin, op.Cout) && op.ps.floats() == (size_t)op.Cin * op.Cout &&
           knob != 0;
}

// Float32 graph: a float32 implicit-GEMM conv directly followed by its 2 x 2 max-pool (VGG16 block*_pool) runs as ONE kernel,
// the pool inside the conv's epilogue (ConvArgs::pool; same bits as the separate pool kernel), unless every activation must be
// kept.  True for op i = the conv; op i + 1 is then the pool that does not launch.
static bool f32_pool_fused(const rpn_model *m, size_t i)
{
    if (m->keep_all || i + 1 >= m->ops.size()) return false;
    const Op &op = m->ops[i], &nx = m->ops[i + 1];
    // the un-pooled tensor is never written when fused: the pool must be its ONLY reader (no residual, tap or returned feature map)
    if (op.out == m->feat_tensor) return false;
    for (size_t j = 0; j < m->ops.size(); ++j)
        if (j != i + 1 && (m->ops[j].in == op.out || m->ops[j].residual == op.out)) return false;
    return op.kind == OP_CONV && !op.split && !op.cin3 && !op.f32_out_split && op.residual < 0 && op.act != ACT_SIGMOID &&
           nx.kind == OP_POOL && !nx.split && nx.in == op.out && (RPN_LAB_KNOB("RPN_F32_POOLFUSE", 1) != 0);
}

// dense conv op; returns the output tensor id
static int add_conv(rpn_model *m, const std::string &name, const std::string &bn, int in, int Cout, int R,
                    int stride, int pad_t, int pad_l, int OH, int OW, int act, int residual = -1,
                    bool force_f32_out = false, bool cin3_out_split = false)
{
    const bool split = m->use_split && R == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && residual < 0 &&
                       act != ACT_SIGMOID &&
                       m->tensors[in].C % 16 == 0 && Cout % 16 == 0 && !m->tensors[in].external;
    if (split && !m->tensors[in].split_fmt && !m->keep_all && !m->ops.empty() && m->ops.back().out == in &&
        m->ops.back().kind == OP_CONV && !m->ops.back().split && !m->ops.back().cin3 && m->ops.back().residual < 0 &&
        m->ops.back().Cout % 32 == 0 && m->ops.back().act != ACT_SIGMOID) {
        // the float32 producer (MobileNetV2: block_13_expand) writes SPLIT16 from its own epilogue: no conversion pass
        m->ops.back().f32_out_split = true;
        m->tensors[in].split_fmt = true;
    }
    if (split && !m->tensors[in].split_fmt) {          // float32 producer -> SPLIT16 copy for the split kernel
        const Tensor tsrc = m->tensors[in];
        Op cv;
        cv.kind = OP_TOSPLIT; cv.name = tsrc.name + "/split"; cv.in = in;
        cv.Cin = cv.Cout = tsrc.C; cv.H = cv.OH = tsrc.H; cv.W = cv.OW = tsrc.W;
        cv.out = add_tensor(m, cv.name, tsrc.H, tsrc.W, tsrc.C);
        m->tensors[cv.out].split_fmt = true;
        m->ops.push_back(cv);
        in = cv.out;
    }
    const Tensor ti = m->tensors[in];
    Op op;
    op.split = split;
    op.cin3 = ti.C == 3 && R == 3 && (stride == 1 || stride == 2) && Cout % 16 == 0 && Cout <= 256 &&
              256 % (Cout / 16) == 0 && residual < 0 && act != ACT_SIGMOID;
    // f16x3 only: with bfloat16 halves the first layer's 2^-16 product error, carried through 40 MobileNetV2 layers,
    // measured 1.25e-4 on the objectness (bound 1e-4); with float16 halves it is indistinguishable from exact f32
    op.cin3_mfma = op.cin3 && m->use_split && m->f16 && (Cout == 32 || Cout == 64) &&
                   (RPN_LAB_KNOB("RPN_CIN3_MFMA", 1) != 0);
    op.out_f32 = split && force_f32_out;
    op.wino = m->wino && R == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && OH == ti.H && OW == ti.W && residual < 0 &&
              act != ACT_SIGMOID && !op.cin3 && !ti.external && wino_supported(ti.C, Cout);
    if (op.wino) {
        op.wino_f = wino_variant(m->max_batch, ti.H, ti.W, ti.C, Cout);
        // a layer the Winograd launcher would refuse (32-bit buffer offsets, ticket block) stays on the direct float32 kernel
        op.wino = wino_launchable(m->max_batch, ti.H, ti.W, ti.C, Cout, op.wino_f);
    }
    op.kind = OP_CONV; op.name = name; op.in = in; op.residual = residual;
    op.Cin = ti.C; op.Cout = Cout; op.R = R; op.S = R; op.stride = stride; op.pad_t = pad_t; op.pad_l = pad_l;
    op.H = ti.H; op.W = ti.W; op.OH = OH; op.OW = OW; op.act = act;
    op.ps = packed_shape(R, R, ti.C, Cout);
    op.k16 = split && use_mfma16(ti.C, Cout, ti.H, ti.W, m->max_batch);
    op.out = add_tensor(m, name, OH, OW, Cout);
    m->tensors[op.out].split_fmt = (split && !force_f32_out) || (op.cin3 && cin3_out_split);
    m->ops.push_back(op);
    const int oi = (int)m->ops.size() - 1;
    add_param(m, oi, name, bn, bn.empty() ? 0 : 1, R, R, ti.C, Cout);
    m->flops += 2.0 * OH * OW * Cout * R * R * ti.C;
    return op.out;
}

static int add_dwconv(rpn_model *m, const std::string &name, const std::string &bn, int in, int stride, int pad_t,
                      int pad_l, int OH, int OW, int act)
{
    const Tensor ti = m->tensors[in];
    Op op;
    op.kind = OP_DWCONV; op.name = name; op.in = in;
    op.Cin = ti.C; op.Cout = ti.C; op.R = 3; op.S = 3; op.stride = stride; op.pad_t = pad_t; op.pad_l = pad_l;
    op.H = ti.H; op.W = ti.W; op.OH = OH; op.OW = OW; op.act = act;
    op.out = add_tensor(m, name, OH, OW, ti.C);
    m->ops.push_back(op);
    const int oi = (int)m->ops.size() - 1;
    add_param(m, oi, name, bn, 2, 3, 3, ti.C, ti.C);
    m->flops += 2.0 * OH * OW * ti.C * 9;
    return op.out;
}

static int add_pool(rpn_model *m, const std::string &name, int in)
{
    const Tensor ti = m->tensors[in];
    Op op;
    op.kind = OP_POOL; op.name = name; op.in = in;
    op.Cin = op.Cout = ti.C; op.H = ti.H; op.W = ti.W; op.OH = ti.H / 2; op.OW = ti.W / 2;   // 'valid' floors
    op.out = add_tensor(m, name, op.OH, op.OW, ti.C);
    op.split = ti.split_fmt;
    m->tensors[op.out].split_fmt = ti.split_fmt;
    m->ops.push_back(op);
    return op.out;
}

// rpn_conv + fused (rpn_reg | rpn_cls) 1x1 head (models/rpn_vgg16.py:18-21)
static void add_head(rpn_model *m, int feat)
{
    const Tensor tf = m->tensors[feat];
    const int x = add_conv(m, "rpn_conv", "", feat, 512, 3, 1, 1, 1, tf.H, tf.W, ACT_RELU, -1, true);
    {   // split-K of rpn_conv at small batches (its partial-sum slabs are added by the head): room for the largest S * B
        const Op &cv = m->ops.back();
        if (cv.split && cv.k16 && cv.out_f32 && !m->keep_all && rpn_head_supported(512, 5 * m->K) && head_splitk() &&
            conv3x3_split16_ktree_ok(m->max_batch, cv.H, cv.W, cv.Cin, cv.Cout, split_cout_pad(cv.Cout))) {
            int cap = m->max_batch;
            for (int b = 1; b <= m->max_batch; ++b) {
                const int sk = conv3x3_split16_ksplit(b, cv.H, cv.W, cv.Cin, cv.Cout, split_cout_pad(cv.Cout));
                if (sk * b > cap) cap = sk * b;
            }
            m->tensors[x].slabs = (cap + m->max_batch - 1) / m->max_batch;
        }
    }
    Op op;
    op.kind = OP_HEAD; op.name = "rpn_head"; op.in = x;
    op.Cin = 512; op.Cout = 5 * m->K; op.R = op.S = 1; op.stride = 1;
    op.H = op.OH = tf.H; op.W = op.OW = tf.W;
    op.ps = packed_shape(1, 1, 512, 5 * m->K);
    op.out = -1;
    m->ops.push_back(op);
    const int oi = (int)m->ops.size() - 1;
    add_param(m, oi, "rpn_reg", "", 0, 1, 1, 512, 4 * m->K, 0);          // columns [0,4K): linear
    add_param(m, oi, "rpn_cls", "", 0, 1, 1, 512, m->K, 4 * m->K);       // columns [4K,5K): sigmoid
    m->flops += 2.0 * tf.H * tf.W * 5 * m->K * 512;
}

// VGG16 block 1 as one launch (f16x3 only; the layer-by-layer graph when every activation must exist, RPN_B1_FUSE=0):
// block1_conv1 + block1_conv2 + block1_pool, the 64-channel full-resolution tensors never reach HBM.
static int add_vgg_block1(rpn_model *m, int in)
{
    const Tensor ti = m->tensors[in];
    Op op;
    op.kind = OP_VGGB1; op.name = "block1_pool"; op.in = in;
    op.Cin = 3; op.Cout = 64; op.R = op.S = 3; op.stride = 1; op.pad_t = op.pad_l = 1;
    op.H = ti.H; op.W = ti.W; op.OH = ti.H / 2; op.OW = ti.W / 2; op.act = ACT_RELU;
    op.out = add_tensor(m, "block1_pool", op.OH, op.OW, 64);
    m->tensors[op.out].split_fmt = true;
    m->ops.push_back(op);
    const int oi = (int)m->ops.size() - 1;
    add_param(m, oi, "block1_conv1", "", 0, 3, 3, 3, 64);
    add_param(m, oi, "block1_conv2", "", 0, 3, 3, 64, 64);
    m->flops += 2.0 * ti.H * ti.W * 64 * 9.0 * (3 + 64);
    return op.out;
}

static void build_vgg16(rpn_model *m)
{
    int t = add_tensor(m, "input", m->img_size, m->img_size, 3, true);
    const int cfg[5][2] = {{2, 64}, {2, 128}, {3, 256}, {3, 512}, {3, 512}};
    const bool fuse_b1 = m->use_split && m->f16 && !m->keep_all &&
                         (RPN_KNOB("RPN_B1_FUSE", 1) != 0);
    if (fuse_b1) t = add_vgg_block1(m, t);
    for (int blk = fuse_b1 ? 1 : 0; blk < 5; ++blk) {
        for (int c = 0; c < cfg[blk][0]; ++c) {
            char name[64];
            snprintf(name, sizeof name, "block%d_conv%d", blk + 1, c + 1);
            const Tensor ti = m->tensors[t];
            // block1_conv1 (Cin = 3) writes SPLIT16 directly when the next layer runs on the split kernel
            t = add_conv(m, name, "", t, cfg[blk][1], 3, 1, 1, 1, ti.H, ti.W, ACT_RELU, -1, false,
                         m->use_split && ti.C == 3);
        }
        if (blk < 4) {                       // block5_pool is never executed (tap = block5_conv3)
            char name[64];
            snprintf(name, sizeof name, "block%d_pool", blk + 1);
            t = add_pool(m, name, t);
        }
    }
    m->feat_tensor = t;
    add_head(m, t);
}

// One fused inverted-residual block (mnv2_block_kernels.hip).  Parameters keep their Keras names, in the order
// expand (or Conv1), depthwise, project -- set_layer addresses them by that position.
static int add_irblock(rpn_model *m, const std::string &name, int in, int cexp, int cout, int stride, int pad, bool res,
                       bool stem, int OH, int OW, const std::string (&pn)[3], const std::string (&bn)[3])
{
    const Tensor ti = m->tensors[in];
    Op op;
    op.kind = OP_IRBLOCK; op.name = name; op.in = in;
    op.Cin = ti.C; op.Cout = cout; op.cexp = cexp; op.stride = stride; op.ir_pad = pad; op.ir_res = res; op.ir_stem = stem;
    op.R = op.S = 3;
    op.H = ti.H; op.W = ti.W; op.OH = OH; op.OW = OW; op.act = ACT_LINEAR;
    op.ir_x3 = m->f16 && !stem && ir_block_x3_supported(ti.C, cexp, cout, stride, res) &&
               (RPN_KNOB("RPN_MN_X3", 1) != 0);
    op.ir_hrx3 = m->f16 && !stem && ir_block_hrx3_supported(ti.C, cexp, cout, stride, res) &&
                 ir_block_hrx3_preferred(ti.C, stride, (long long)((OW + 7) / 8) * ((OH + 3) / 4) * m->max_batch) &&
                 (RPN_KNOB("RPN_MN_X3", 1) != 0) && (RPN_LAB_KNOB("RPN_MN_HRX3", 1) != 0);
    if (op.ir_hrx3) {
        op.ir_x3 = true;
        op.ir_ce = ir_hrx3_chunk_for(ti.C, stride, (long long)((OW + 7) / 8) * ((OH + 3) / 4) * m->max_batch);
    }
    op.out = add_tensor(m, name, OH, OW, cout);
    m->ops.push_back(op);
    const int oi = (int)m->ops.size() - 1;
    // expanded-tensor size (the depthwise's input grid)
    const int DH = stem ? (ti.H + pad + 1 - 3) / 2 + 1 : ti.H, DW = stem ? (ti.W + pad + 1 - 3) / 2 + 1 : ti.W;
    if (stem) add_param(m, oi, pn[0], bn[0], 1, 3, 3, 3, cexp);
    else add_param(m, oi, pn[0], bn[0], 1, 1, 1, ti.C, cexp);
    add_param(m, oi, pn[1], bn[1], 2, 3, 3, cexp, cexp);
    add_param(m, oi, pn[2], bn[2], 1, 1, 1, cexp, cout);
    m->flops += 2.0 * DH * DW * cexp * (stem ? 27 : ti.C) + 2.0 * OH * OW * cexp * 9 + 2.0 * OH * OW * cexp * cout;
    return op.out;
}

// keras_applications.correct_pad for a 3x3 kernel: (before, after) per spatial dim
static void correct_pad(int n, int *before, int *after)
{
    const int adjust = 1 - n % 2;
    *before = 1 - adjust;
    *after = 1;
}

static void build_mobilenet_v2(rpn_model *m)
{
    int t = add_tensor(m, "input", m->img_size, m->img_size, 3, true);
    // One launch per inverted-residual block unless every intermediate activation must exist (keep_all) or
    // RPN_MN_FUSE=0 (the unfused layer-by-layer path: the cross-check of the fused kernels in the tests).
    const bool fuse = !m->keep_all && (RPN_KNOB("RPN_MN_FUSE", 1) != 0);
    const int blocks[13][4] = {{32, 1, 16, 1}, {16, 6, 24, 2}, {24, 6, 24, 1}, {24, 6, 32, 2}, {32, 6, 32, 1},
                               {32, 6, 32, 1}, {32, 6, 64, 2}, {64, 6, 64, 1}, {64, 6, 64, 1}, {64, 6, 64, 1},
                               {64, 6, 96, 1}, {96, 6, 96, 1}, {96, 6, 96, 1}};
    int first_block = 0;
    if (fuse) {     // Conv1_pad + Conv1 + bn_Conv1 + ReLU6 + expanded_conv (depthwise, project): one launch
        const Tensor ti = m->tensors[t];
        int pb, pa;
        correct_pad(ti.H, &pb, &pa);
        const int OH = (ti.H + pb + pa - 3) / 2 + 1;
        const std::string pn[3] = {"Conv1", "expanded_conv_depthwise", "expanded_conv_project"};
        const std::string bn[3] = {"bn_Conv1", "expanded_conv_depthwise_BN", "expanded_conv_project_BN"};
        t = add_irblock(m, "expanded_conv_project", t, 32, 16, 1, pb, false, true, OH, OH, pn, bn);
        first_block = 1;
    } else {   // Conv1_pad + Conv1 (3x3 s2 valid, no bias) + bn_Conv1 + ReLU6
        const Tensor ti = m->tensors[t];
        int pb, pa;
        correct_pad(ti.H, &pb, &pa);
        const int OH = (ti.H + pb + pa - 3) / 2 + 1;
        t = add_conv(m, "Conv1", "bn_Conv1", t, 32, 3, 2, pb, pb, OH, OH, ACT_RELU6);
    }
    for (int bid = first_block; bid < 13; ++bid) {
        const int cin = blocks[bid][0], expand = blocks[bid][1], cout = blocks[bid][2], stride = blocks[bid][3];
        char pre[32];
        if (bid == 0) snprintf(pre, sizeof pre, "expanded_conv_");
        else snprintf(pre, sizeof pre, "block_%d_", bid);
        const std::string prefix = pre;
        const int inp = t;
        if (fuse && bid != 0 && ir_block_supported(cin, cin * expand, cout, stride, cin == cout && stride == 1)) {
            const Tensor ti = m->tensors[t];
            int pb = 1, pa = 1, OH = ti.H;
            if (stride == 2) {
                correct_pad(ti.H, &pb, &pa);
                OH = (ti.H + pb + pa - 3) / 2 + 1;
            }
            const std::string pn[3] = {prefix + "expand", prefix + "depthwise", prefix + "project"};
            const std::string bn[3] = {prefix + "expand_BN", prefix + "depthwise_BN", prefix + "project_BN"};
            t = add_irblock(m, prefix + "project", t, cin * expand, cout, stride, pb, cin == cout && stride == 1, false,
                            OH, OH, pn, bn);
            continue;
        }
        if (bid != 0) {
            const Tensor ti = m->tensors[t];
            t = add_conv(m, prefix + "expand", prefix + "expand_BN", t, cin * expand, 1, 1, 0, 0, ti.H, ti.W,
                         ACT_RELU6);
        }
        {
            const Tensor ti = m->tensors[t];
            if (stride == 2) {
                int pb, pa;
                correct_pad(ti.H, &pb, &pa);
                const int OH = (ti.H + pb + pa - 3) / 2 + 1;
                t = add_dwconv(m, prefix + "depthwise", prefix + "depthwise_BN", t, 2, pb, pb, OH, OH, ACT_RELU6);
            } else {
                t = add_dwconv(m, prefix + "depthwise", prefix + "depthwise_BN", t, 1, 1, 1, ti.H, ti.W, ACT_RELU6);
            }
        }
        {
            const Tensor ti = m->tensors[t];
            const int res = (cin == cout && stride == 1) ? inp : -1;
            t = add_conv(m, prefix + "project", prefix + "project_BN", t, cout, 1, 1, 0, 0, ti.H, ti.W, ACT_LINEAR,
                         res);
        }
    }
    {
        const Tensor ti = m->tensors[t];
        t = add_conv(m, "block_13_expand", "block_13_expand_BN", t, 576, 1, 1, 0, 0, ti.H, ti.W, ACT_RELU6);
    }
    m->feat_tensor = t;
    add_head(m, t);
}

// liveness-based arena plan (first fit); offsets in floats per image
static void plan_arena(rpn_model *m)
{
    for (auto &t : m->tensors) t.def = t.last_use = -1;
    for (int i = 0; i < (int)m->ops.size(); ++i) {
        const Op &op = m->ops[i];
        if (op.out >= 0) m->tensors[op.out].def = i;
        if (op.in >= 0) m->tensors[op.in].last_use = i;
        if (op.residual >= 0) m->tensors[op.residual].last_use = i;
    }
    // a split conv fused with its max-pool writes the POOL's output while it still reads its own input:
    // the pooled tensor becomes live one op earlier
    if (!m->keep_all)
        for (size_t i = 0; i + 1 < m->ops.size(); ++i) {
            const Op &op = m->ops[i], &nx = m->ops[i + 1];
            if (op.kind == OP_CONV && op.split && !op.out_f32 && nx.kind == OP_POOL && nx.split && nx.in == op.out)
                m->tensors[nx.out].def = (int)i;
            if (f32_pool_fused(m, i)) m->tensors[nx.out].def = (int)i;      // (the float32 graph's fused pools likewise)
        }
    size_t top = 0;
    std::vector<int> placed;
    for (int ti = 0; ti < (int)m->tensors.size(); ++ti) {
        Tensor &t = m->tensors[ti];
        if (t.external || t.def < 0) continue;
        if (t.last_use < t.def) t.last_use = t.def;
        const size_t sz = (t.floats() + 63) & ~(size_t)63;          // 256-byte granules
        size_t off = 0;
        if (m->keep_all) {
            off = top;
        } else {
            bool moved = true;
            while (moved) {
                moved = false;
                for (int pj : placed) {
                    const Tensor &o = m->tensors[pj];
                    const bool live_overlap = !(o.last_use < t.def || t.last_use < o.def);
                    const size_t osz = (o.floats() + 63) & ~(size_t)63;
                    if (live_overlap && off < o.offset + osz && o.offset < off + sz) {
                        off = o.offset + osz;
                        moved = true;
                    }
                }
            }
        }
        t.offset = off;
        placed.push_back(ti);
        if (off + sz > top) top = off + sz;
    }
    m->arena_floats = top;
}

static void plan_weights(rpn_model *m)
{
    size_t off = 0;
    for (auto &op : m->ops) {
        if (op.kind == OP_CONV && op.cin3) {
            op.w_off = off;
            off += ((size_t)32 * op.Cout + 63) & ~(size_t)63;        // 27 x Cout floats, or Cout x 128 B of split records
            op.b_off = off;
            off += ((size_t)op.Cout + 63) & ~(size_t)63;
        } else if (op.kind == OP_CONV && op.split) {
            op.w_off = off;
            off += (split_weight_bytes(op.Cin, op.Cout) / sizeof(float) + 63) & ~(size_t)63;
            op.b_off = off;
            off += (size_t)split_cout_pad(op.Cout);
        } else if (op.kind == OP_CONV && op.wino) {
            op.w_off = off;
            off += (wino_weight_floats(op.Cin, op.Cout, op.wino_f) + 63) & ~(size_t)63;
            op.b_off = off;
            off += (size_t)op.ps.cout_pad;
        } else if (op.kind == OP_CONV || op.kind == OP_HEAD) {
            op.w_off = off;
            off += (op.ps.floats() + 63) & ~(size_t)63;
            op.b_off = off;
            off += (size_t)op.ps.cout_pad;
            if (op.kind == OP_HEAD && rpn_head_supported(op.Cin, op.Cout)) {      // fragment-ordered copy for rpn_head_kernel
                op.ir_off[0] = off;
                off += (size_t)512 * 16 * ((op.Cout + 15) / 16);
            }
        } else if (op.kind == OP_DWCONV) {
            op.w_off = off;
            off += ((size_t)9 * op.Cin + 63) & ~(size_t)63;
            op.b_off = off;
            off += ((size_t)op.Cin + 63) & ~(size_t)63;
        } else if (op.kind == OP_VGGB1) {
            const size_t sizes[4] = {(size_t)32 * 64, 64, split_weight_bytes(64, 64) / sizeof(float), 64};
            op.w_off = off;
            for (int i = 0; i < 4; ++i) {
                op.ir_off[i] = off;
                off += (sizes[i] + 63) & ~(size_t)63;
            }
        } else if (op.kind == OP_IRBLOCK) {
            const size_t kp = op.ir_stem ? 28 : (size_t)op.Cin, coutp = ((size_t)op.Cout + 15) / 16 * 16;
            size_t sizes[6] = {kp * op.cexp, (size_t)op.cexp, (size_t)9 * op.cexp, (size_t)op.cexp,
                               (size_t)op.cexp * coutp, coutp};
            if (op.ir_hrx3) {                                             // zero-padded fragment images: larger than the matrices
                sizes[0] = ir_hrx3_expand_floats(op.cexp);
                sizes[4] = ir_hrx3_project_floats(op.Cin, op.cexp, op.Cout, op.stride, op.ir_ce);
            }
            if (op.ir_x3) sizes[2] = ir_x3_dw_floats(op.cexp);           // depthwise taps + bias as one record per channel
            op.w_off = off;
            for (int i = 0; i < 6; ++i) {
                op.ir_off[i] = off;
                off += (sizes[i] + 63) & ~(size_t)63;
            }
        }
    }
    m->weight_floats = off;
}

static int ensure_device(rpn_model *m)
{
    if (!have_device()) return RPN_ERR_NO_DEVICE;
    if (!m->d_weights) {
        RPN_HIP_CHECK(hipMalloc(&m->d_weights, m->weight_floats * sizeof(float)));
        RPN_HIP_CHECK(hipMemset(m->d_weights, 0, m->weight_floats * sizeof(float)));
    }
    if (!m->d_arena) {
        RPN_HIP_CHECK(hipMalloc(&m->d_arena, m->arena_floats * (size_t)m->max_batch * sizeof(float)));
    }
    if (!m->d_status) {
        RPN_HIP_CHECK(hipMalloc(&m->d_status, 64));
        RPN_HIP_CHECK(hipMemset(m->d_status, 0, 64));
    }
    if (!m->d_wino_ws) {
        size_t need = 0;
        for (const Op &op : m->ops)
            if (op.kind == OP_CONV && op.wino) need = std::max(need, wino_workspace_bytes(m->max_batch, op.H, op.W, op.Cin, op.Cout, op.wino_f));
        if (need) {
            RPN_HIP_CHECK(hipMalloc(&m->d_wino_ws, need));
            // the tickets must read zero before the first forward, which runs on the CALLER's stream: a non-blocking stream is not
            // ordered behind the null stream's memset, so drain it here (allocation path, once per handle)
            RPN_HIP_CHECK(hipMemset(m->d_wino_ws, 0, need));
            RPN_HIP_CHECK(hipStreamSynchronize(nullptr));
        }
    }
    if (!m->d_ksplit) {
        bool any = false;
        for (const Op &op : m->ops) any = any || (op.kind == OP_IRBLOCK && op.ir_x3);
        if (any) {
            RPN_HIP_CHECK(hipMalloc(&m->d_ksplit, ir_block_x3_scratch_floats() * sizeof(float)));
            RPN_HIP_CHECK(hipMemset(m->d_ksplit, 0, ir_block_x3_scratch_floats() * sizeof(float)));
        }
    }
    return RPN_OK;
}

static float *tensor_ptr(rpn_model *m, int id, const float *d_input)
{
    if (id == 0) return const_cast<float *>(d_input);
    return m->d_arena + m->tensors[id].offset * (size_t)m->max_batch;
}

}  // namespace rpn

extern "C" int rpn_model_create(int backbone, int img_size, int anchor_count, int precision, int max_batch,
                                rpn_model **out)
{
    RPN_REQUIRE(out, "rpn_model_create: null out");
    RPN_REQUIRE(backbone == RPN_BACKBONE_VGG16 || backbone == RPN_BACKBONE_MOBILENET_V2,
                "rpn_model_create: unknown backbone %d", backbone);
    RPN_REQUIRE(img_size >= 32 && img_size <= 8192, "rpn_model_create: img_size %d out of range", img_size);
    RPN_REQUIRE(anchor_count >= 1 && anchor_count <= 64, "rpn_model_create: anchor_count %d out of range",
                anchor_count);
    RPN_REQUIRE(max_batch >= 1, "rpn_model_create: max_batch must be >= 1");
    RPN_REQUIRE(precision == RPN_PRECISION_F32 || precision == RPN_PRECISION_BF16X3 || precision == RPN_PRECISION_F16X3 ||
                    precision == RPN_PRECISION_F32W,
                "rpn_model_create: unknown precision %d", precision);
    rpn_model *m = new rpn_model();
    m->backbone = backbone; m->img_size = img_size; m->K = anchor_count; m->precision = precision;
    m->max_batch = max_batch;
    m->use_split = precision == RPN_PRECISION_BF16X3 || precision == RPN_PRECISION_F16X3;
    m->f16 = precision == RPN_PRECISION_F16X3;
    m->wino = precision == RPN_PRECISION_F32W;
    if (backbone == RPN_BACKBONE_VGG16) build_vgg16(m);
    else build_mobilenet_v2(m);
    m->F = m->tensors[m->feat_tensor].H;
    plan_arena(m);
    plan_weights(m);
    *out = m;
    return RPN_OK;
}

extern "C" void rpn_model_destroy(rpn_model *m)
{
    if (!m) return;
    if (m->d_weights) (void)hipFree(m->d_weights);
    if (m->d_arena) (void)hipFree(m->d_arena);
    if (m->d_status) (void)hipFree(m->d_status);
    if (m->d_ksplit) (void)hipFree(m->d_ksplit);
    if (m->d_wino_ws) (void)hipFree(m->d_wino_ws);
    for (auto &ev : m->events) (void)hipEventDestroy(ev);
    delete m;
}

extern "C" int rpn_model_feature_map_shape(const rpn_model *m) { return m ? m->F : 0; }

extern "C" int rpn_model_num_layers(const rpn_model *m) { return m ? (int)m->params.size() : 0; }

extern "C" int rpn_model_layer_info(const rpn_model *m, int i, char *name, int name_len, int shape[4], int *kind)
{
    RPN_REQUIRE(m && i >= 0 && i < (int)m->params.size(), "rpn_model_layer_info: bad index %d", i);
    const Param &p = m->params[i];
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", p.name.c_str());
    if (shape) {
        shape[0] = p.R; shape[1] = p.S; shape[2] = p.Cin; shape[3] = (p.kind == 2) ? 1 : p.Cout;
    }
    if (kind) *kind = p.kind;
    return RPN_OK;
}

// name of the BatchNormalization layer that follows layer i ("" when there is none)
extern "C" int rpn_model_layer_bn_name(const rpn_model *m, int i, char *name, int name_len)
{
    RPN_REQUIRE(m && i >= 0 && i < (int)m->params.size(), "rpn_model_layer_bn_name: bad index %d", i);
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", m->params[i].bn_name.c_str());
    return RPN_OK;
}

extern "C" double rpn_model_flops_per_image(const rpn_model *m) { return m ? m->flops : 0.0; }

// bytes of device memory the handle needs (weights, activation arena)
extern "C" int rpn_model_memory_bytes(const rpn_model *m, size_t *weights, size_t *arena)
{
    RPN_REQUIRE(m, "rpn_model_memory_bytes: null model");
    if (weights) *weights = m->weight_floats * sizeof(float);
    if (arena) *arena = m->arena_floats * (size_t)m->max_batch * sizeof(float);
    return RPN_OK;
}

// keep every intermediate activation alive (unique arena offsets) so that
// rpn_model_get_activation can read any layer after a forward pass; must precede the first forward
extern "C" int rpn_model_keep_activations(rpn_model *m, int keep)
{
    RPN_REQUIRE(m, "rpn_model_keep_activations: null model");
    RPN_REQUIRE(!m->d_arena && !m->d_weights, "rpn_model_keep_activations: must precede the first set_layer / forward");
    m->keep_all = keep != 0;
    // the graph itself depends on it (MobileNetV2 blocks are fused into one launch only when their intermediate
    // activations need not exist): rebuild
    m->tensors.clear(); m->ops.clear(); m->params.clear(); m->flops = 0.0;
    if (m->backbone == RPN_BACKBONE_VGG16) build_vgg16(m);
    else build_mobilenet_v2(m);
    m->F = m->tensors[m->feat_tensor].H;
    plan_arena(m);
    plan_weights(m);
    return RPN_OK;
}

extern "C" int rpn_model_set_layer(rpn_model *m, const char *name, const float *kernel, const float *bias,
                                   const float *bn_gamma, const float *bn_beta, const float *bn_mean,
                                   const float *bn_var)
{
    RPN_REQUIRE(m && name && kernel, "rpn_model_set_layer: null argument");
    int pi = -1;
    for (int i = 0; i < (int)m->params.size(); ++i)
        if (m->params[i].name == name) pi = i;
    RPN_REQUIRE(pi >= 0, "rpn_model_set_layer: no layer named '%s'", name);
    Param &p = m->params[pi];
    Op &op = m->ops[p.op];
    const bool has_bn = p.kind != 0;
    RPN_REQUIRE(!has_bn || (bn_gamma && bn_beta && bn_mean && bn_var),
                "rpn_model_set_layer: layer '%s' is followed by BatchNorm '%s'; its parameters are required", name,
                p.bn_name.c_str());
    RPN_REQUIRE(has_bn || !bn_gamma, "rpn_model_set_layer: layer '%s' has no BatchNorm", name);
    const int st = ensure_device(m);
    if (st != RPN_OK) return st;

    // fold BatchNorm (inference mode, eps = 1e-3): y = conv(x) * scale + shift
    std::vector<float> scale, shift(p.Cout, 0.0f);
    if (has_bn) {
        scale.resize(p.Cout);
        for (int n = 0; n < p.Cout; ++n) {
            scale[n] = bn_gamma[n] / sqrtf(bn_var[n] + kBnEps);
            shift[n] = bn_beta[n] - bn_mean[n] * scale[n];
        }
    }
    if (bias)
        for (int n = 0; n < p.Cout; ++n) shift[n] += has_bn ? bias[n] * scale[n] : bias[n];

    if (op.kind == OP_VGGB1) {
        // parameter 0 = block1_conv1 (one-MFMA-step records), 1 = block1_conv2 (split records); no BatchNorm in VGG16
        const bool first = op.params[0] == pi;
        const size_t count = (size_t)9 * p.Cin * p.Cout;
        const int wshift = split_weight_shift(kernel, count, m->f16);
        std::vector<unsigned short> packed(first ? (size_t)64 * 64 : split_weight_bytes(64, 64) / sizeof(unsigned short));
        if (first) pack_weights_cin3_mfma_host(kernel, nullptr, 64, 64, m->f16, wshift, packed.data());
        else pack_weights_split_host(kernel, nullptr, 64, 64, 64, m->f16, wshift, packed.data());
        op.ir_scale[first ? 0 : 1] = ldexpf(1.0f, -wshift);
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.ir_off[first ? 0 : 2], packed.data(), packed.size() * sizeof(unsigned short),
                                hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.ir_off[first ? 1 : 3], shift.data(), (size_t)64 * sizeof(float),
                                hipMemcpyHostToDevice));
    } else if (op.kind == OP_IRBLOCK) {
        // role = position among the op's parameters: 0 expand / Conv1, 1 depthwise, 2 project.  Zero padding (the stem's
        // 28th im2col row, projection columns beyond Cout) comes from the memset of the weight blob.
        int role = 0;
        while (role < 3 && op.params[(size_t)role] != pi) ++role;
        std::vector<float> w;
        size_t w_at, b_at, nb;
        if (role == 0) {                                   // (R,S,Cin,Cexp) row-major == [K][Cexp]
            const size_t rows = op.ir_stem ? 27 : (size_t)p.Cin;
            w.resize(rows * p.Cout);
            for (size_t k = 0; k < rows; ++k)
                for (int n = 0; n < p.Cout; ++n) w[k * p.Cout + n] = kernel[k * p.Cout + n] * scale[n];
            if (op.ir_hrx3) {
                const int sh = split_weight_shift(w.data(), w.size(), true);
                std::vector<float> packed(ir_hrx3_expand_floats(p.Cout));
                pack_ir_hrx3_expand(w.data(), p.Cin, p.Cout, op.stride, sh, reinterpret_cast<unsigned short *>(packed.data()), op.ir_ce);
                w.swap(packed);
                op.ir_scale[0] = ldexpf(1.0f, -sh);
            } else if (op.ir_x3) {                         // hi + lo float16 fragments, same byte count
                const int sh = split_weight_shift(w.data(), w.size(), true);
                std::vector<float> packed(w.size());
                pack_ir_x3_expand(w.data(), p.Cin, p.Cout, sh, reinterpret_cast<unsigned short *>(packed.data()));
                w.swap(packed);
                op.ir_scale[0] = ldexpf(1.0f, -sh);
            }
            w_at = op.ir_off[0]; b_at = op.ir_off[1]; nb = (size_t)p.Cout;
        } else if (role == 1) {                            // (3,3,C,1) == [9][C]
            w.resize((size_t)9 * p.Cin);
            for (int t = 0; t < 9; ++t)
                for (int c = 0; c < p.Cin; ++c) w[(size_t)t * p.Cin + c] = kernel[(size_t)t * p.Cin + c] * scale[c];
            if (op.ir_x3) {                                // [channel][9 taps, bias, 0, 0] (both f16x3 block kernels)
                std::vector<float> packed(ir_x3_dw_floats(p.Cin));
                pack_ir_x3_dw(w.data(), shift.data(), p.Cin, packed.data());
                w.swap(packed);
            }
            w_at = op.ir_off[2]; b_at = op.ir_off[3]; nb = (size_t)p.Cin;
        } else {                                           // (1,1,Cexp,Cout) -> [Cexp][coutp]
            const size_t coutp = ((size_t)p.Cout + 15) / 16 * 16;
            w.assign((size_t)p.Cin * coutp, 0.0f);
            for (int k = 0; k < p.Cin; ++k)
                for (int n = 0; n < p.Cout; ++n) w[(size_t)k * coutp + n] = kernel[(size_t)k * p.Cout + n] * scale[n];
            if (op.ir_hrx3) {
                const int sh = split_weight_shift(w.data(), w.size(), true);
                std::vector<float> packed(ir_hrx3_project_floats(op.Cin, p.Cin, p.Cout, op.stride, op.ir_ce));
                pack_ir_hrx3_project(w.data(), op.Cin, p.Cin, p.Cout, op.stride, sh, reinterpret_cast<unsigned short *>(packed.data()), op.ir_ce);
                w.swap(packed);
                op.ir_scale[1] = ldexpf(1.0f, -sh);
            } else if (op.ir_x3) {                         // (Cout % 16 == 0 for these blocks: coutp == Cout)
                const int sh = split_weight_shift(w.data(), w.size(), true);
                std::vector<float> packed(w.size());
                pack_ir_x3_project(w.data(), p.Cin, p.Cout, sh, reinterpret_cast<unsigned short *>(packed.data()));
                w.swap(packed);
                op.ir_scale[1] = ldexpf(1.0f, -sh);
            }
            w_at = op.ir_off[4]; b_at = op.ir_off[5]; nb = (size_t)p.Cout;
        }
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + w_at, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + b_at, shift.data(), nb * sizeof(float), hipMemcpyHostToDevice));
    } else if (p.kind == 2) {
        std::vector<float> w((size_t)9 * p.Cin);
        for (int t = 0; t < 9; ++t)
            for (int c = 0; c < p.Cin; ++c) w[(size_t)t * p.Cin + c] = kernel[(size_t)t * p.Cin + c] * scale[c];
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float),
                                hipMemcpyHostToDevice));
    } else if (op.kind == OP_HEAD) {
        // this parameter fills columns [col_off, col_off + Cout) of the fused head matrix
        const PackedShape &ps = op.ps;
        RPN_HIP_CHECK(hipMemcpy2D(m->d_weights + op.w_off + p.col_off, (size_t)ps.cout_pad * sizeof(float), kernel,
                                  (size_t)p.Cout * sizeof(float), (size_t)p.Cout * sizeof(float), (size_t)p.Cin,
                                  hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off + p.col_off, shift.data(), (size_t)p.Cout * sizeof(float),
                                hipMemcpyHostToDevice));
        if (rpn_head_supported(op.Cin, op.Cout)) {
            m->head_host.resize((size_t)512 * ps.cout_pad, 0.0f);
            for (int k = 0; k < 512; ++k)
                memcpy(&m->head_host[(size_t)k * ps.cout_pad + p.col_off], kernel + (size_t)k * p.Cout, (size_t)p.Cout * sizeof(float));
            std::vector<float> packed((size_t)512 * 16 * ((op.Cout + 15) / 16));
            pack_head_weights_host(m->head_host.data(), ps.cout_pad, op.Cout, packed.data());
            RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.ir_off[0], packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    } else if (op.cin3_mfma) {
        // the power-of-two pre-scale is derived from the FOLDED weights (a BatchNorm scale of 30 on top of a shift
        // chosen for the raw kernel would push the float16 hi halves past 65504)
        std::vector<float> folded(kernel, kernel + (size_t)27 * p.Cout);
        if (has_bn)
            for (size_t i = 0; i < folded.size(); ++i) folded[i] *= scale[i % (size_t)p.Cout];
        const int wshift = split_weight_shift(folded.data(), folded.size(), m->f16);
        std::vector<unsigned short> packed((size_t)p.Cout * 64);
        pack_weights_cin3_mfma_host(folded.data(), nullptr, p.Cout, p.Cout, m->f16, wshift, packed.data());
        op.out_scale = ldexpf(1.0f, -wshift);
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, packed.data(), packed.size() * sizeof(unsigned short),
                                hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float),
                                hipMemcpyHostToDevice));
    } else if (op.cin3) {
        std::vector<float> w((size_t)27 * p.Cout);
        for (int k = 0; k < 27; ++k)
            for (int n = 0; n < p.Cout; ++n)
                w[(size_t)k * p.Cout + n] = kernel[(size_t)k * p.Cout + n] * (has_bn ? scale[n] : 1.0f);
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float),
                                hipMemcpyHostToDevice));
    } else if (op.split) {
        const int cpad = split_cout_pad(op.Cout);
        std::vector<float> folded(kernel, kernel + (size_t)9 * p.Cin * p.Cout);       // BatchNorm folded BEFORE the shift
        if (has_bn)
            for (size_t i = 0; i < folded.size(); ++i) folded[i] *= scale[i % (size_t)p.Cout];
        const int wshift = split_weight_shift(folded.data(), folded.size(), m->f16);
        std::vector<unsigned short> packed(split_weight_bytes(op.Cin, op.Cout) / sizeof(unsigned short));
        if (op.k16)
            pack_weights_split32_host(folded.data(), nullptr, p.Cin, p.Cout, cpad, m->f16, wshift, packed.data());
        else
            pack_weights_split_host(folded.data(), nullptr, p.Cin, p.Cout, cpad, m->f16, wshift, packed.data());
        op.out_scale = ldexpf(1.0f, -wshift);
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, packed.data(), packed.size() * sizeof(unsigned short),
                                hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float),
                                hipMemcpyHostToDevice));
    } else if (op.wino) {
        std::vector<float> packed(wino_weight_floats(p.Cin, p.Cout, op.wino_f));
        pack_weights_wino_host(kernel, has_bn ? scale.data() : nullptr, p.Cin, p.Cout, packed.data(), op.wino_f);
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float), hipMemcpyHostToDevice));
    } else if (is_pw_x3(m, op)) {
        // [Cin][Cout] with BatchNorm folded, then the power-of-two pre-scale and the fragment-major hi / lo float16 image
        std::vector<float> w((size_t)p.Cin * p.Cout);
        for (int k = 0; k < p.Cin; ++k)
            for (int n = 0; n < p.Cout; ++n) w[(size_t)k * p.Cout + n] = kernel[(size_t)k * p.Cout + n] * (has_bn ? scale[n] : 1.0f);
        const int sh = split_weight_shift(w.data(), w.size(), true);
        std::vector<float> packed(w.size());
        pack_ir_x3_expand(w.data(), p.Cin, p.Cout, sh, reinterpret_cast<unsigned short *>(packed.data()));
        op.out_scale = ldexpf(1.0f, -sh);
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float), hipMemcpyHostToDevice));
    } else {
        std::vector<float> packed(op.ps.floats());
        pack_weights_host(op.ps, kernel, has_bn ? scale.data() : nullptr, packed.data());
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.w_off, packed.data(), packed.size() * sizeof(float),
                                hipMemcpyHostToDevice));
        RPN_HIP_CHECK(hipMemcpy(m->d_weights + op.b_off, shift.data(), (size_t)p.Cout * sizeof(float),
                                hipMemcpyHostToDevice));
    }
    p.loaded = true;
    return RPN_OK;
}

extern "C" int rpn_model_forward(rpn_model *m, const float *d_imgs, int B, float *d_reg, float *d_cls, void *stream)
{
    RPN_REQUIRE(m && d_imgs && d_reg && d_cls, "rpn_model_forward: null argument");
    RPN_REQUIRE(B >= 1 && B <= m->max_batch, "rpn_model_forward: batch %d outside [1, %d]", B, m->max_batch);
    for (const Param &p : m->params)
        RPN_REQUIRE(p.loaded, "rpn_model_forward: weights of layer '%s' were never set", p.name.c_str());
    const int st = ensure_device(m);
    if (st != RPN_OK) return st;
    hipStream_t s = as_stream(stream);
    m->last_input = d_imgs;
    struct StatusScope {                   // the split-format writers launched below flag float16 range violations here
        explicit StatusScope(unsigned *p) { set_range_status(p); }
        ~StatusScope() { set_range_status(nullptr); }
    } status_scope(m->f16 ? m->d_status : nullptr);
    hipEvent_t *evs = nullptr;
    int sel_op = -1;
    if (m->profiling > 0) {
        const size_t per = m->ops.size() + 1;
        if (m->events.size() != per * (size_t)m->profiling) {
            for (auto &ev : m->events) (void)hipEventDestroy(ev);
            m->events.assign(per * (size_t)m->profiling, nullptr);
            for (auto &ev : m->events) RPN_HIP_CHECK(hipEventCreate(&ev));
            m->profiled_forwards = 0;
        }
        const size_t slot = (size_t)(m->profiled_forwards % m->profiling);
        evs = m->events.data() + per * slot;
        m->prof_timed.resize((size_t)m->profiling, -1);
        m->prof_timed[slot] = -1;
        if (m->prof_rotate && !m->prof_mask.empty()) {
            std::vector<int> marked;
            for (size_t i = 0; i < m->ops.size(); ++i)
                if (m->prof_mask[i]) marked.push_back((int)i);
            if (!marked.empty()) m->prof_timed[slot] = marked[(size_t)(m->profiled_forwards % (long long)marked.size())];
        }
        sel_op = m->prof_timed[slot];
        ++m->profiled_forwards;
    }
    // event i+1 closes op i and opens op i+1: with a mask only the boundaries of marked ops are recorded
    // (with rotation: of the one op selected for this forward)
    auto want_event = [&](size_t boundary) {
        if (m->prof_mask.empty()) return true;
        auto marked = [&](size_t i) { return sel_op >= 0 ? (int)i == sel_op : m->prof_mask[i] != 0; };
        return (boundary > 0 && marked(boundary - 1)) || (boundary < m->ops.size() && marked(boundary));
    };
    if (evs && want_event(0)) RPN_HIP_CHECK(hipEventRecord(evs[0], s));
    int op_index = 0;
    bool skip_next = false;
    int head_slabs = 1;                     // > 1: the head's input is that many partial-sum slabs of a split-K rpn_conv
    size_t head_conv_bias = 0;
    // split-K factor of op i at this batch size: > 1 only for rpn_conv feeding the slab-adding head, when its output tensor
    // has room for factor * B images (RPN_KSPLIT=1; conv3x3_split16_ksplit says when it pays)
    // rpn_conv feeding the slab-adding head runs as a K TREE (conv_kernels.h) at every batch size of this model, or at none
    auto ktree_for = [&](size_t i) -> bool {
        const Op &o = m->ops[i];
        return o.kind == OP_CONV && o.split && o.k16 && o.out_f32 && i + 1 < m->ops.size() && m->ops[i + 1].kind == OP_HEAD &&
               m->ops[i + 1].in == o.out && rpn_head_supported(512, 5 * m->K) && head_splitk() &&
               conv3x3_split16_ktree_ok(m->max_batch, o.H, o.W, o.Cin, o.Cout, split_cout_pad(o.Cout));
    };
    auto ksplit_for = [&](size_t i) -> int {
        const Op &o = m->ops[i];
        if (!ktree_for(i) || m->keep_all) return 1;     // (every activation kept: the tree inside one workgroup's tile loop)
        const int f = conv3x3_split16_ksplit(B, o.H, o.W, o.Cin, o.Cout, split_cout_pad(o.Cout));
        return (size_t)f * B <= (size_t)m->tensors[o.out].slabs * m->max_batch ? f : 1;
    };
    for (size_t oi = 0; oi < m->ops.size(); ++oi) {
        const Op &op = m->ops[oi];
        const float *x = tensor_ptr(m, op.in, d_imgs);
        hipError_t e = hipSuccess;
        // a split conv directly followed by its 2x2 max-pool runs as ONE kernel (pool fused into the
        // epilogue; the un-pooled activation never reaches HBM) unless every activation must be kept
        const bool fuse_pool = op.kind == OP_CONV && op.split && !op.out_f32 && !m->keep_all &&
                               oi + 1 < m->ops.size() && m->ops[oi + 1].kind == OP_POOL &&
                               m->ops[oi + 1].split && m->ops[oi + 1].in == op.out;
        if (skip_next) {
            skip_next = false;                      // the pool already ran inside the previous conv
        } else if (op.kind == OP_TOSPLIT) {
            e = launch_f32_to_split(x, (long long)B * op.H * op.W, op.Cin, m->f16, tensor_ptr(m, op.out, d_imgs), s);
        } else if (op.kind == OP_CONV && op.cin3_mfma) {
            e = launch_conv_cin3_mfma(x, m->d_weights + op.w_off, m->d_weights + op.b_off, tensor_ptr(m, op.out, d_imgs),
                                      B, op.H, op.W, op.OH, op.OW, op.Cout, op.stride, op.pad_t, op.pad_l, op.act,
                                      op.out_scale, m->tensors[op.out].split_fmt ? 1 : 0, m->f16, s);
        } else if (op.kind == OP_CONV && op.cin3) {
            e = launch_conv_cin3(x, m->d_weights + op.w_off, m->d_weights + op.b_off, tensor_ptr(m, op.out, d_imgs), B,
                                 op.H, op.W, op.OH, op.OW, op.Cout, op.stride, op.pad_t, op.pad_l, op.act,
                                 m->tensors[op.out].split_fmt ? 1 : 0, m->f16, s);
        } else if (ksplit_for(oi) > 1) {
            // rpn_conv at a small batch: split-K, raw partial sums; the head adds the slabs (+ this layer's bias, ReLU)
            head_slabs = ksplit_for(oi);
            head_conv_bias = op.b_off;
            e = launch_conv3x3_split16_ksplit(x, m->d_weights + op.w_off, tensor_ptr(m, op.out, d_imgs),
                                              (long long)B * op.H * op.W * op.Cout, B, op.H, op.W, op.Cin, op.Cout,
                                              split_cout_pad(op.Cout), op.out_scale, m->f16, head_slabs, s);
        } else if (ktree_for(oi)) {
            e = launch_conv3x3_split16(x, m->d_weights + op.w_off, m->d_weights + op.b_off, tensor_ptr(m, op.out, d_imgs), B, op.H,
                                       op.W, op.Cin, op.Cout, split_cout_pad(op.Cout), op.out_scale, op.act, true, m->f16, false,
                                       s, true);
        } else if (op.kind == OP_CONV && op.split) {
            const int dst = fuse_pool ? m->ops[oi + 1].out : op.out;
            e = (op.k16 ? launch_conv3x3_split16 : launch_conv3x3_split)(
                x, m->d_weights + op.w_off, m->d_weights + op.b_off, tensor_ptr(m, dst, d_imgs), B, op.H, op.W, op.Cin,
                op.Cout, split_cout_pad(op.Cout), op.out_scale, op.act, op.out_f32, m->f16, fuse_pool, s, false);
            skip_next = fuse_pool;
        } else if (op.kind == OP_POOL && op.split) {
            e = launch_maxpool_split(x, B, op.H, op.W, op.Cin, m->f16, tensor_ptr(m, op.out, d_imgs), s);
        } else if (op.kind == OP_HEAD && rpn_head_supported(op.Cin, op.Cout) && head_splitk()) {
            e = launch_rpn_head(x, (long long)B * op.H * op.W, m->d_weights + op.ir_off[0], m->d_weights + op.b_off, 4 * m->K,
                                m->K, d_reg, d_cls, s, head_slabs, (long long)B * op.H * op.W * 512,
                                head_slabs > 1 ? m->d_weights + head_conv_bias : nullptr);
            head_slabs = 1;
        } else if (is_pw_x3(m, op)) {
            e = launch_pw_x3(x, (long long)B * op.H * op.W, op.Cin, op.Cout, m->d_weights + op.w_off, m->d_weights + op.b_off,
                             op.out_scale, tensor_ptr(m, op.out, d_imgs), m->d_status, s);
        } else if (op.kind == OP_CONV && op.wino) {
            const bool pool = f32_pool_fused(m, oi);                    // + block*_pool: a Winograd tile's 2 x 2 outputs are one window
            e = launch_conv3x3_wino(x, m->d_weights + op.w_off, m->d_weights + op.b_off,
                                    tensor_ptr(m, pool ? m->ops[oi + 1].out : op.out, d_imgs), B, op.H, op.W, op.Cin, op.Cout, op.act,
                                    pool, s, op.wino_f, m->d_wino_ws);
            skip_next = pool;
        } else if (op.kind == OP_CONV || op.kind == OP_HEAD) {
            ConvArgs a{};
            a.x = x;
            a.w = m->d_weights + op.w_off;
            a.bias = m->d_weights + op.b_off;
            a.residual = op.residual >= 0 ? tensor_ptr(m, op.residual, d_imgs) : nullptr;
            a.B = B; a.H = op.H; a.W = op.W; a.Cin = op.Cin; a.OH = op.OH; a.OW = op.OW; a.Cout = op.Cout;
            a.R = op.R; a.S = op.S; a.stride = op.stride; a.pad_t = op.pad_t; a.pad_l = op.pad_l;
            a.ps = op.ps;
            if (op.kind == OP_HEAD) {
                a.out = d_reg; a.ld1 = 4 * m->K; a.act = ACT_LINEAR; a.split = 4 * m->K;
                a.out2 = d_cls; a.ld2 = m->K; a.act2 = ACT_SIGMOID;
            } else {
                const bool pool = f32_pool_fused(m, oi);                // + block*_pool in the epilogue: writes the pool's tensor
                a.out = tensor_ptr(m, pool ? m->ops[oi + 1].out : op.out, d_imgs); a.ld1 = op.Cout; a.act = op.act; a.split = op.Cout;
                a.out2 = nullptr; a.ld2 = 0; a.act2 = ACT_LINEAR;
                a.out_split = op.f32_out_split ? (m->f16 ? 2 : 1) : 0;
                a.status = op.f32_out_split && m->f16 ? m->d_status : nullptr;
                a.pool = pool ? 1 : 0;
                skip_next = pool;
            }
            e = launch_conv_f32(a, s);
        } else if (op.kind == OP_VGGB1) {
            const float *wb = m->d_weights;
            e = launch_vgg_block1(x, wb + op.ir_off[0], wb + op.ir_off[1], op.ir_scale[0], wb + op.ir_off[2],
                                  wb + op.ir_off[3], op.ir_scale[1], tensor_ptr(m, op.out, d_imgs), B, op.H, op.W, m->f16, s);
        } else if (op.kind == OP_IRBLOCK) {
            const float *wb = m->d_weights;
            if (op.ir_hrx3)
                e = launch_ir_block_hrx3(x, B, op.H, op.W, op.Cin, op.cexp, op.Cout, op.stride, op.ir_res, op.ir_pad, op.OH, op.OW,
                                         wb + op.ir_off[0], wb + op.ir_off[1], wb + op.ir_off[2], wb + op.ir_off[3],
                                         wb + op.ir_off[4], wb + op.ir_off[5], op.ir_scale[0], op.ir_scale[1],
                                         tensor_ptr(m, op.out, d_imgs), m->d_status, m->d_ksplit, s, op.ir_ce);
            else if (op.ir_x3)
                e = launch_ir_block_x3(x, B, op.H, op.W, op.Cin, op.cexp, op.Cout, op.ir_res, wb + op.ir_off[0],
                                       wb + op.ir_off[1], wb + op.ir_off[2], wb + op.ir_off[3], wb + op.ir_off[4],
                                       wb + op.ir_off[5], op.ir_scale[0], op.ir_scale[1], tensor_ptr(m, op.out, d_imgs),
                                       m->d_status, m->d_ksplit, s);
            else
            e = launch_ir_block(x, B, op.H, op.W, op.Cin, op.cexp, op.Cout, op.stride, op.ir_res, op.ir_stem, op.ir_pad,
                                op.OH, op.OW, wb + op.ir_off[0], wb + op.ir_off[1], wb + op.ir_off[2], wb + op.ir_off[3],
                                wb + op.ir_off[4], wb + op.ir_off[5], tensor_ptr(m, op.out, d_imgs), s);
        } else if (op.kind == OP_DWCONV) {
            e = launch_dwconv3x3(x, B, op.H, op.W, op.Cin, m->d_weights + op.w_off, m->d_weights + op.b_off,
                                 op.stride, op.pad_t, op.pad_l, op.OH, op.OW, op.act, tensor_ptr(m, op.out, d_imgs), s);
        } else {
            e = launch_maxpool2x2(x, B, op.H, op.W, op.Cin, tensor_ptr(m, op.out, d_imgs), s);
        }
        if (e != hipSuccess)
            return fail(RPN_ERR_NO_DEVICE, "rpn_model_forward: layer '%s' failed to launch: %s", op.name.c_str(),
                        hipGetErrorString(e));
        ++op_index;
        if (evs && want_event((size_t)op_index)) RPN_HIP_CHECK(hipEventRecord(evs[op_index], s));
    }
    return RPN_OK;
}

extern "C" int rpn_model_status(rpn_model *m, unsigned *flags, int reset, void *stream)
{
    RPN_REQUIRE(m && flags, "rpn_model_status: null argument");
    *flags = 0;
    if (!m->d_status) return RPN_OK;                                // no forward has run
    hipStream_t s = as_stream(stream);
    RPN_HIP_CHECK(hipMemcpyAsync(flags, m->d_status, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    if (reset) RPN_HIP_CHECK(hipMemsetAsync(m->d_status, 0, sizeof(unsigned), s));
    RPN_HIP_CHECK(hipStreamSynchronize(s));
    return RPN_OK;
}

// ---- per-op timing with HIP events on the caller's stream ------------------------------------
extern "C" int rpn_model_set_profiling(rpn_model *m, int n_forwards)
{
    RPN_REQUIRE(m && n_forwards >= 0 && n_forwards <= 4096, "rpn_model_set_profiling: bad argument");
    m->profiling = n_forwards;
    m->profiled_forwards = 0;
    return RPN_OK;
}

// Restrict the per-op events to the ops with mask[i] != 0 (NULL: every op).  A throughput run that only needs the
// dominant kernel's durations then records 2 events per marked launch instead of one per op.
extern "C" int rpn_model_set_profiling_mask(rpn_model *m, const unsigned char *mask, int n)
{
    RPN_REQUIRE(m, "rpn_model_set_profiling_mask: null model");
    if (!mask) { m->prof_mask.clear(); return RPN_OK; }
    RPN_REQUIRE(n == (int)m->ops.size(), "rpn_model_set_profiling_mask: mask has %d entries, the model %d ops", n, (int)m->ops.size());
    m->prof_mask.assign(mask, mask + n);
    m->profiled_forwards = 0;
    return RPN_OK;
}

// With a mask: time only ONE of the marked ops per forward, round robin over the forwards (2 events per forward
// instead of 2 per marked launch; every marked op is still timed live, in every (number of marked ops)-th forward).
extern "C" int rpn_model_set_profiling_rotate(rpn_model *m, int on)
{
    RPN_REQUIRE(m, "rpn_model_set_profiling_rotate: null model");
    m->prof_rotate = on ? 1 : 0;
    m->profiled_forwards = 0;
    return RPN_OK;
}

extern "C" int rpn_model_num_ops(const rpn_model *m) { return m ? (int)m->ops.size() : 0; }

// name, kernel family ("conv128x128", "conv128x64", "conv128x32", "conv_generic*", "dwconv", "maxpool") and
// algorithmic FLOPs (2*MACs, whole batch of the last profiled forward... per image here) of op i
extern "C" int rpn_model_op_info(const rpn_model *m, int i, char *name, int name_len, char *kernel, int kernel_len,
                                 double *flops_per_image, double *bytes_per_image)
{
    RPN_REQUIRE(m && i >= 0 && i < (int)m->ops.size(), "rpn_model_op_info: bad index %d", i);
    const Op &op = m->ops[i];
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", op.name.c_str());
    double fl = 0.0, by = 0.0;
    const char *k = "maxpool2x2";
    const double in_b = 4.0 * op.H * op.W * op.Cin, out_b = 4.0 * op.OH * op.OW * op.Cout;
    // a 2x2 max-pool that runs inside the epilogue of the split conv in front of it (rpn_model_forward: fuse_pool) is not a
    // launch: no kernel, no bytes of its own (the conv's row carries the pooled output)
    const bool pool_fused_away = op.kind == OP_POOL && op.split && !m->keep_all && i > 0 && m->ops[i - 1].kind == OP_CONV &&
                                 m->ops[i - 1].split && !m->ops[i - 1].out_f32 && !m->ops[i - 1].cin3 && op.in == m->ops[i - 1].out;
    if (op.kind == OP_TOSPLIT) {
        by = in_b + out_b;
        k = "f32_to_split";
    } else if (pool_fused_away) {
        k = "fused:maxpool_split";
    } else if (op.kind == OP_POOL && i > 0 && f32_pool_fused(m, (size_t)i - 1)) {
        k = "fused:maxpool_f32";                   // runs inside the previous float32 conv's epilogue: no launch, no bytes of its own
    } else if (op.kind == OP_POOL && op.split) {
        by = in_b + out_b;
        k = "maxpool_split";
    } else if (op.kind == OP_CONV && op.cin3) {
        fl = 2.0 * op.OH * op.OW * op.Cout * 27.0;
        by = in_b + out_b;
        k = op.cin3_mfma ? "conv_cin3_mfma"
            : conv_cin3_uses_f32_mfma(1, op.H, op.W, op.OH, op.OW, op.Cout, op.stride, op.pad_t, op.pad_l, op.act,
                                      m->tensors[op.out].split_fmt ? 1 : 0) ? "conv_cin3_f32_mfma" : "conv_cin3_direct";
    } else if (op.kind == OP_CONV && op.split) {
        fl = 2.0 * op.OH * op.OW * op.Cout * 9.0 * op.Cin;
        by = in_b + out_b + 4.0 * 9 * op.Cin * op.Cout;
        // the kernel template that runs this layer (both POOL instantiations under one name)
        static thread_local char kname[64];
        const char *prec = m->f16 ? "f16x3" : "bf16x3";
        if (op.k16) {
            const bool fused_pool = i + 1 < (int)m->ops.size() && m->ops[i + 1].kind == OP_POOL && m->ops[i + 1].split;
            const char *v = conv3x3_split16_variant(m->max_batch, op.H, op.W, op.Cin, op.Cout, split_cout_pad(op.Cout), fused_pool);
            // (rpn_conv as a K tree: the 64-wide persistent kernel whenever one workgroup per tile walks all of K)
            if (v && v[0] == 'd' && op.out_f32 && i + 1 < (int)m->ops.size() && m->ops[i + 1].kind == OP_HEAD &&
                rpn_head_supported(512, 5 * m->K) && head_splitk() &&
                conv3x3_split16_ktree_ok(m->max_batch, op.H, op.W, op.Cin, op.Cout, split_cout_pad(op.Cout)))
                v = "dma,64";
            if (v && v[0] == 'd') snprintf(kname, sizeof kname, "conv3x3_split16_dma<%s,%s>", prec, v + 4);
            else snprintf(kname, sizeof kname, "conv3x3_split16<%s,%s>", prec, v ? v + 4 : "?");
        } else {
            snprintf(kname, sizeof kname, "conv3x3_split<%s>", prec);
        }
        k = kname;
    } else if (op.kind == OP_CONV || op.kind == OP_HEAD) {
        fl = 2.0 * op.OH * op.OW * op.Cout * op.R * op.S * op.Cin;
        by = in_b + out_b + 4.0 * op.R * op.S * op.Cin * op.Cout;
        static thread_local char kname32[64];
        if (op.kind == OP_HEAD && rpn_head_supported(op.Cin, op.Cout) && head_splitk())
            snprintf(kname32, sizeof kname32, "rpn_head_splitk<%d>", (op.Cout + 15) / 16);
        else if (is_pw_x3(m, op))
            snprintf(kname32, sizeof kname32, "pw_f16x3<%d,%d>", op.Cin, op.Cout);
        else if (op.kind == OP_CONV && op.wino)
            snprintf(kname32, sizeof kname32, op.wino_f == 16 ? "conv3x3_wino4_f32<16x16x128>" : op.wino_f == 4 ? "conv3x3_wino4_f32<16x32x64>"
                                              : (op.wino_f == 8 ? "conv3x3_wino4_f32<16x32x64,k2>" : "conv3x3_wino_f32<16x16x64>"));   // (fl above: the direct conv's flops; 4 x / 2.25 x fewer are executed)
        else
        {
            const int bn32 = conv_f32_tile_n(m->max_batch, op.OH, op.OW, op.Cout);
            snprintf(kname32, sizeof kname32, "conv_igemm_f32%s<128x%d%s>", conv_f32_uses_dma(bn32, op.ps.generic) ? "_dma" : "", bn32,
                     op.ps.generic ? ",generic" : "");
        }
        k = kname32;
    } else if (op.kind == OP_VGGB1) {
        fl = 2.0 * op.H * op.W * 64 * 9.0 * (3 + 64);
        by = in_b + out_b + 4.0 * 9 * (3 + 64) * 64;
        static thread_local char kb1[64];
        snprintf(kb1, sizeof kb1, "vgg_block1<%s>", m->f16 ? "f16x3" : "bf16x3");
        k = kb1;
    } else if (op.kind == OP_IRBLOCK) {
        const int DH = op.ir_stem ? (op.H + op.ir_pad + 1 - 3) / 2 + 1 : op.H, DW = op.ir_stem ? (op.W + op.ir_pad + 1 - 3) / 2 + 1 : op.W;
        const double kin = op.ir_stem ? 27.0 : (double)op.Cin;
        fl = 2.0 * DH * DW * op.cexp * kin + 2.0 * op.OH * op.OW * op.cexp * 9.0 + 2.0 * op.OH * op.OW * (double)op.cexp * op.Cout;
        by = in_b + out_b + 4.0 * (kin * op.cexp + 10.0 * op.cexp + (double)op.cexp * op.Cout + op.Cout);
        static thread_local char kir[64];
        if (op.ir_stem) snprintf(kir, sizeof kir, "ir_block<stem,32,16,s1>");
        else snprintf(kir, sizeof kir, "ir_block%s<%d,%d,%d,s%d%s>", op.ir_hrx3 ? "_hr_f16x3" : (op.ir_x3 ? "_f16x3" : ""), op.Cin, op.cexp, op.Cout, op.stride,
                      op.ir_res ? ",res" : "");
        k = kir;
    } else if (op.kind == OP_DWCONV) {
        fl = 2.0 * op.OH * op.OW * op.Cin * 9;
        by = in_b + out_b;
        k = "dwconv3x3";
    } else {
        by = in_b + out_b;
    }
    if (kernel && kernel_len > 0) snprintf(kernel, (size_t)kernel_len, "%s", k);
    if (flops_per_image) *flops_per_image = fl;
    if (bytes_per_image) *bytes_per_image = by;
    return RPN_OK;
}

// arithmetic the op's matrix work runs in: RPN_PRECISION_F32 (float32 MFMA / vector ALU) or the model's split precision
// (three 16-bit MFMAs per product) -- what a roofline report must price the op's flops against
extern "C" int rpn_model_op_arith(const rpn_model *m, int i)
{
    if (!m || i < 0 || i >= (int)m->ops.size()) return -1;
    const Op &op = m->ops[i];
    const bool x3 = (op.kind == OP_CONV && (op.split || op.cin3_mfma)) || op.kind == OP_VGGB1 || (op.kind == OP_IRBLOCK && op.ir_x3) ||
                    is_pw_x3(m, op);
    if (op.kind == OP_CONV && op.wino) return RPN_PRECISION_F32W;     // float32 MFMA, 1 / 2.25 of the direct conv's multiply-adds
    return x3 ? (m->f16 ? RPN_PRECISION_F16X3 : RPN_PRECISION_BF16X3) : RPN_PRECISION_F32;
}

// mean elapsed milliseconds of every op over the kept forwards (synchronises on their last events);
// returns the number of forwards averaged in *n_forwards (may be NULL)
extern "C" int rpn_model_get_profile(rpn_model *m, float *ms, int n, int *n_forwards)
{
    RPN_REQUIRE(m && ms, "rpn_model_get_profile: null argument");
    RPN_REQUIRE(m->profiling > 0 && m->profiled_forwards > 0 && !m->events.empty(),
                "rpn_model_get_profile: profiling is off or no forward has run");
    RPN_REQUIRE(n >= (int)m->ops.size(), "rpn_model_get_profile: need room for %d ops", (int)m->ops.size());
    const size_t per = m->ops.size() + 1;
    const int kept = (int)(m->profiled_forwards < m->profiling ? m->profiled_forwards : m->profiling);
    std::vector<double> sum(m->ops.size(), 0.0);
    std::vector<int> cnt(m->ops.size(), 0);
    for (int f = 0; f < kept; ++f) {
        hipEvent_t *evs = m->events.data() + per * (size_t)f;
        const int sel = (size_t)f < m->prof_timed.size() ? m->prof_timed[(size_t)f] : -1;
        for (size_t i = 0; i < m->ops.size(); ++i) {
            if (!m->prof_mask.empty() && !m->prof_mask[i]) continue;          // not timed: stays 0
            if (sel >= 0 && (int)i != sel) continue;                          // rotation: not the op timed in this forward
            RPN_HIP_CHECK(hipEventSynchronize(evs[i + 1]));
            float t = 0.0f;
            RPN_HIP_CHECK(hipEventElapsedTime(&t, evs[i], evs[i + 1]));
            sum[i] += t;
            ++cnt[i];
        }
    }
    for (size_t i = 0; i < m->ops.size(); ++i) ms[i] = cnt[i] ? (float)(sum[i] / cnt[i]) : 0.0f;
    if (n_forwards) *n_forwards = kept;
    return RPN_OK;
}

extern "C" int rpn_model_get_activation(rpn_model *m, const char *name, float *d_out, size_t out_bytes,
                                        int shape[4], void *stream)
{
    RPN_REQUIRE(m && name, "rpn_model_get_activation: null argument");
    int ti = -1;
    for (int i = 1; i < (int)m->tensors.size(); ++i)
        if (m->tensors[i].name == name) ti = i;
    RPN_REQUIRE(ti >= 0, "rpn_model_get_activation: no activation named '%s'", name);
    const Tensor &t = m->tensors[ti];
    if (shape) {
        shape[0] = m->max_batch; shape[1] = t.H; shape[2] = t.W; shape[3] = t.C;
    }
    if (!d_out) return RPN_OK;                                      // shape query only
    RPN_REQUIRE(m->keep_all || ti == m->feat_tensor,
                "rpn_model_get_activation: call rpn_model_keep_activations(m, 1) before the first forward");
    RPN_REQUIRE(m->d_arena, "rpn_model_get_activation: no forward pass has run");
    const size_t bytes = t.floats() * (size_t)m->max_batch * sizeof(float);
    RPN_REQUIRE(out_bytes >= bytes, "rpn_model_get_activation: %zu bytes needed, %zu given", bytes, out_bytes);
    if (t.split_fmt) {
        const hipError_t e = launch_split_to_f32(m->d_arena + t.offset * (size_t)m->max_batch,
                                                 (long long)m->max_batch * t.H * t.W, t.C, m->f16, d_out,
                                                 as_stream(stream));
        if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_model_get_activation: %s", hipGetErrorString(e));
        return RPN_OK;
    }
    RPN_HIP_CHECK(hipMemcpyAsync(d_out, m->d_arena + t.offset * (size_t)m->max_batch, bytes,
                                 hipMemcpyDeviceToDevice, as_stream(stream)));
    return RPN_OK;
}

// ---- single-layer entry points (kernel-level parity tests, micro-benchmarks) -----------------
extern "C" int rpn_conv2d(const float *d_x, int B, int H, int W, int Cin, const float *d_w, const float *d_bias,
                          int R, int S, int Cout, int stride, int pad_t, int pad_l, int OH, int OW, int act,
                          int precision, float *d_out, void *stream)
{
    RPN_REQUIRE(d_x && d_w && d_out, "rpn_conv2d: null pointer");
    RPN_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cin >= 1 && Cout >= 1 && R >= 1 && S >= 1 && stride >= 1 &&
                    OH >= 1 && OW >= 1 && pad_t >= 0 && pad_l >= 0,
                "rpn_conv2d: bad geometry");
    RPN_REQUIRE(act >= 0 && act <= 3, "rpn_conv2d: bad activation %d", act);
    RPN_REQUIRE(precision == RPN_PRECISION_F32 || precision == RPN_PRECISION_BF16X3 || precision == RPN_PRECISION_F16X3 ||
                    precision == RPN_PRECISION_F32W,
                "rpn_conv2d: unknown precision %d", precision);
    RPN_REQUIRE_DEVICE();
    hipStream_t s = as_stream(stream);
    if (precision == RPN_PRECISION_F32W) {
        // float32 Winograd (the form wino_variant picks for this grid: F(4x4, 3x3) wide / 64-channel / split-channel, or F(2x2, 3x3)):
        // 3x3 / stride 1 / pad 1 only; transforms and packs the weights on the host
        if (!(R == 3 && S == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && OH == H && OW == W && wino_supported(Cin, Cout) &&
              act != ACT_SIGMOID))
            return fail(RPN_ERR_UNSUPPORTED, "rpn_conv2d: the Winograd path needs 3x3 s1 'same', Cin %% 8 == 0, Cout %% 32 == 0");
        const size_t wcount = (size_t)9 * Cin * Cout;
        const int wf = wino_variant(B, H, W, Cin, Cout);
        if (!wino_launchable(B, H, W, Cin, Cout, wf))
            return fail(RPN_ERR_UNSUPPORTED, "rpn_conv2d: the Winograd path needs H * W * Cin * 4 bytes per image below 2 GiB");
        std::vector<float> hw(wcount), packed(wino_weight_floats(Cin, Cout, wf));
        RPN_HIP_CHECK(hipMemcpy(hw.data(), d_w, wcount * sizeof(float), hipMemcpyDeviceToHost));
        pack_weights_wino_host(hw.data(), nullptr, Cin, Cout, packed.data(), wf);
        float *d_u = nullptr;
        void *d_ws = nullptr;
        RPN_HIP_CHECK(hipMalloc(&d_u, packed.size() * sizeof(float)));
        hipError_t e = hipMemcpy(d_u, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice);
        const size_t ws_bytes = wino_workspace_bytes(B, H, W, Cin, Cout, wf);
        if (e == hipSuccess && ws_bytes) e = hipMalloc(&d_ws, ws_bytes);
        if (e == hipSuccess && ws_bytes) e = hipMemset(d_ws, 0, ws_bytes);
        if (e == hipSuccess) e = launch_conv3x3_wino(d_x, d_u, d_bias, d_out, B, H, W, Cin, Cout, act, false, s, wf, d_ws);
        const hipError_t e2 = hipStreamSynchronize(s);
        (void)hipFree(d_u);
        if (d_ws) (void)hipFree(d_ws);
        if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d(f32w): launch failed: %s", hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d(f32w): kernel failed: %s", hipGetErrorString(e2));
        return RPN_OK;
    }
    if (precision != RPN_PRECISION_F32) {
        // x3-split path: 3x3 / stride 1 / pad 1 only; converts x to SPLIT16 and packs the weights on the host
        if (!(R == 3 && S == 3 && stride == 1 && pad_t == 1 && pad_l == 1 && OH == H && OW == W && Cin % 16 == 0 &&
              Cout % 16 == 0 && act != ACT_SIGMOID))
            return fail(RPN_ERR_UNSUPPORTED, "rpn_conv2d: the split path needs 3x3 s1 'same', Cin,Cout %% 16 == 0");
        const bool f16 = precision == RPN_PRECISION_F16X3;
        const size_t wcount = (size_t)9 * Cin * Cout;
        std::vector<float> hw(wcount);
        RPN_HIP_CHECK(hipMemcpy(hw.data(), d_w, wcount * sizeof(float), hipMemcpyDeviceToHost));
        const int cpad = split_cout_pad(Cout);
        const int wshift = split_weight_shift(hw.data(), wcount, f16);
        std::vector<unsigned short> packed(split_weight_bytes(Cin, Cout) / 2);
        const bool k16 = use_mfma16(Cin, Cout, H, W, B);
        if (k16) pack_weights_split32_host(hw.data(), nullptr, Cin, Cout, cpad, f16, wshift, packed.data());
        else pack_weights_split_host(hw.data(), nullptr, Cin, Cout, cpad, f16, wshift, packed.data());
        void *d_ws = nullptr, *d_xs = nullptr;
        hipError_t e = hipMalloc(&d_ws, packed.size() * 2);
        if (e == hipSuccess) e = hipMalloc(&d_xs, (size_t)B * H * W * Cin * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(d_ws, packed.data(), packed.size() * 2, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = launch_f32_to_split(d_x, (long long)B * H * W, Cin, f16, d_xs, s);
        if (e == hipSuccess)
            e = (k16 ? launch_conv3x3_split16 : launch_conv3x3_split)(d_xs, d_ws, d_bias, d_out, B, H, W, Cin, Cout, cpad,
                                                                      ldexpf(1.0f, -wshift), act, true, f16, false, s, false);
        const hipError_t e2 = hipStreamSynchronize(s);
        if (d_ws) (void)hipFree(d_ws);                  // freed on every path
        if (d_xs) (void)hipFree(d_xs);
        if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d(split): launch failed: %s", hipGetErrorString(e));
        if (e2 != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d(split): kernel failed: %s", hipGetErrorString(e2));
        return RPN_OK;
    }
    if (Cin == 3 && R == 3 && S == 3 && (stride == 1 || stride == 2) && Cout % 16 == 0 && Cout <= 256 &&
        256 % (Cout / 16) == 0 && act != ACT_SIGMOID) {   // first-layer direct kernel
        const hipError_t e = launch_conv_cin3(d_x, d_w, d_bias, d_out, B, H, W, OH, OW, Cout, stride, pad_t, pad_l, act,
                                              0, false, s);
        if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d(cin3): launch failed: %s", hipGetErrorString(e));
        RPN_HIP_CHECK(hipStreamSynchronize(s));
        return RPN_OK;
    }
    ConvArgs a{};
    a.ps = packed_shape(R, S, Cin, Cout);
    float *d_packed = nullptr;
    RPN_HIP_CHECK(hipMalloc(&d_packed, a.ps.floats() * sizeof(float)));   // test entry point: allocates + syncs
    pack_weights_device(a.ps, d_w, d_packed, s);
    a.x = d_x; a.w = d_packed; a.bias = d_bias; a.residual = nullptr;
    a.out = d_out; a.out2 = nullptr;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.OH = OH; a.OW = OW; a.Cout = Cout;
    a.R = R; a.S = S; a.stride = stride; a.pad_t = pad_t; a.pad_l = pad_l;
    a.act = act; a.act2 = ACT_LINEAR; a.split = Cout; a.ld1 = Cout; a.ld2 = 0;
    const hipError_t e = launch_conv_f32(a, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(d_packed);
    if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d: launch failed: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_conv2d: kernel failed: %s", hipGetErrorString(e2));
    return RPN_OK;
}

extern "C" int rpn_maxpool2x2(const float *d_x, int B, int H, int W, int C, float *d_out, void *stream)
{
    RPN_REQUIRE(d_x && d_out, "rpn_maxpool2x2: null pointer");
    RPN_REQUIRE(B >= 1 && H >= 2 && W >= 2 && C >= 4 && C % 4 == 0, "rpn_maxpool2x2: bad shape (C %% 4 must be 0)");
    RPN_REQUIRE_DEVICE();
    const hipError_t e = launch_maxpool2x2(d_x, B, H, W, C, d_out, as_stream(stream));
    if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_maxpool2x2: launch failed: %s", hipGetErrorString(e));
    return RPN_OK;
}

// depthwise 3x3 single-layer entry (w: (3,3,C) device, bias (C) device or NULL)
extern "C" int rpn_dwconv3x3(const float *d_x, int B, int H, int W, int C, const float *d_w, const float *d_bias,
                             int stride, int pad_t, int pad_l, int OH, int OW, int act, float *d_out, void *stream)
{
    RPN_REQUIRE(d_x && d_w && d_out, "rpn_dwconv3x3: null pointer");
    RPN_REQUIRE(B >= 1 && C >= 4 && C % 4 == 0 && (stride == 1 || stride == 2), "rpn_dwconv3x3: bad shape");
    RPN_REQUIRE_DEVICE();
    const hipError_t e = launch_dwconv3x3(d_x, B, H, W, C, d_w, d_bias, stride, pad_t, pad_l, OH, OW, act, d_out,
                                          as_stream(stream));
    if (e != hipSuccess) return fail(RPN_ERR_NO_DEVICE, "rpn_dwconv3x3: launch failed: %s", hipGetErrorString(e));
    return RPN_OK;
}
