"""torch-CPU restatement of the reference's model forward (conv backbone + RPN head).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  PARITY UNPINNED: the
arithmetic of these layers lives in TensorFlow 2.0.0 (Conv2D / MaxPool /
DepthwiseConv2D / BatchNormalization kernels) and the graphs in
keras-applications 1.0.8 (``environment.yml:22,49-52``), neither of which is under
/root/reference or importable here.  What IS pinned by the reference:

* models/rpn_vgg16.py:16-21       VGG16(include_top=False) tapped at ``block5_conv3``, then
                                  ``rpn_conv`` 3x3x512 relu same, ``rpn_cls`` 1x1xK sigmoid,
                                  ``rpn_reg`` 1x1x4K linear; outputs ``[reg, cls]``
* models/rpn_mobilenet_v2.py:16-21 MobileNetV2(include_top=False) tapped at ``block_13_expand_relu``,
                                  same three head layers
* utils/data_utils.py:25-26       input is NHWC float32 in [0,1], no mean subtraction

The Keras layer semantics restated here: Conv2D = cross-correlation, NHWC x HWIO,
bias add then activation; 'same' 3x3 stride 1 = zero pad 1; MaxPooling2D(2,2) 'valid'
floors; MobileNetV2 stride-2 convs = ZeroPadding2D(correct_pad) + 'valid'; BatchNorm in
inference mode with eps = 1e-3; ReLU6 after expand / depthwise, none after project.

``torch.nn.functional.conv2d`` (oneDNN) is an implementation of the same maths that is
independent of the HIP kernels.  ``dtype=torch.float64`` gives the error-budget reference.
"""
import numpy as np
import torch
import torch.nn.functional as Fnn

# ---------------------------------------------------------------- VGG16 ----
# (name, cin, cout) conv 3x3 s1 same relu; "pool" = MaxPooling2D(2,2) valid
VGG16_LAYERS = [
    ("block1_conv1", 3, 64), ("block1_conv2", 64, 64), "pool",
    ("block2_conv1", 64, 128), ("block2_conv2", 128, 128), "pool",
    ("block3_conv1", 128, 256), ("block3_conv2", 256, 256), ("block3_conv3", 256, 256), "pool",
    ("block4_conv1", 256, 512), ("block4_conv2", 512, 512), ("block4_conv3", 512, 512), "pool",
    ("block5_conv1", 512, 512), ("block5_conv2", 512, 512), ("block5_conv3", 512, 512),
]

# -------------------------------------------------------- MobileNetV2 ----
# keras-applications 1.0.8 mobilenet_v2.py, alpha = 1.0, up to block_13_expand_relu.
# (block_id, in_ch, expansion, out_ch, stride)
MNV2_BLOCKS = [
    (0, 32, 1, 16, 1),
    (1, 16, 6, 24, 2), (2, 24, 6, 24, 1),
    (3, 24, 6, 32, 2), (4, 32, 6, 32, 1), (5, 32, 6, 32, 1),
    (6, 32, 6, 64, 2), (7, 64, 6, 64, 1), (8, 64, 6, 64, 1), (9, 64, 6, 64, 1),
    (10, 64, 6, 96, 1), (11, 96, 6, 96, 1), (12, 96, 6, 96, 1),
]
BN_EPS = 1e-3


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _conv(x, w_hwio, bias, stride=1, padding=0, groups=1, dtype=torch.float32):
    """x NCHW; w HWIO (or HWC1 for depthwise, Keras depthwise_kernel layout (R,S,Cin,1))."""
    w = _t(w_hwio, dtype)
    if groups == 1:
        w = w.permute(3, 2, 0, 1).contiguous()          # OIHW
    else:
        w = w.permute(2, 3, 0, 1).contiguous()          # (Cin,1,R,S)
    b = _t(bias, dtype) if bias is not None else None
    return Fnn.conv2d(x, w, b, stride=stride, padding=padding, groups=groups)


def _bn(x, p, dtype):
    g, b, m, v = (_t(p[k], dtype).view(1, -1, 1, 1) for k in ("gamma", "beta", "mean", "var"))
    return g * (x - m) / torch.sqrt(v + BN_EPS) + b


def _correct_pad(n, k=3):
    """keras_applications.correct_pad for one spatial dim: (before, after)."""
    adjust = 1 - n % 2
    correct = k // 2
    return correct - adjust, correct


def _head(x, weights, dtype):
    """models/rpn_vgg16.py:18-20 == models/rpn_mobilenet_v2.py:18-20."""
    x = torch.relu(_conv(x, weights["rpn_conv"]["kernel"], weights["rpn_conv"]["bias"], padding=1, dtype=dtype))
    cls = torch.sigmoid(_conv(x, weights["rpn_cls"]["kernel"], weights["rpn_cls"]["bias"], dtype=dtype))
    reg = _conv(x, weights["rpn_reg"]["kernel"], weights["rpn_reg"]["bias"], dtype=dtype)
    return reg, cls


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(torch.float32).numpy()


def vgg16_rpn_forward(imgs_nhwc, weights, dtype=torch.float32, return_features=False):
    """models/rpn_vgg16.py:16-21.  Returns [reg (B,F,F,4K), cls (B,F,F,K)] as float32 numpy."""
    with torch.no_grad():
        x = _t(imgs_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
        for layer in VGG16_LAYERS:
            if layer == "pool":
                x = Fnn.max_pool2d(x, 2, 2)                       # 'valid': floors 125 -> 62
            else:
                name = layer[0]
                x = torch.relu(_conv(x, weights[name]["kernel"], weights[name]["bias"], padding=1, dtype=dtype))
        feat = x
        reg, cls = _head(x, weights, dtype)
        out = [_nhwc(reg), _nhwc(cls)]
        return out + [_nhwc(feat)] if return_features else out


def mobilenet_v2_rpn_forward(imgs_nhwc, weights, dtype=torch.float32, return_features=False):
    """models/rpn_mobilenet_v2.py:16-21 (tap: block_13_expand_relu, 576 channels, stride 16)."""
    relu6 = lambda t: torch.clamp(t, 0.0, 6.0)
    with torch.no_grad():
        x = _t(imgs_nhwc, dtype).permute(0, 3, 1, 2).contiguous()
        # Conv1_pad + Conv1 (3x3 s2 valid, no bias) + bn_Conv1 + Conv1_relu
        ph, pw = _correct_pad(x.shape[2]), _correct_pad(x.shape[3])
        x = Fnn.pad(x, (pw[0], pw[1], ph[0], ph[1]))
        x = relu6(_bn(_conv(x, weights["Conv1"]["kernel"], None, stride=2, dtype=dtype), weights["bn_Conv1"], dtype))
        for (bid, cin, t, cout, stride) in MNV2_BLOCKS:
            prefix = "expanded_conv_" if bid == 0 else "block_%d_" % bid
            inp = x
            if bid != 0:
                x = relu6(_bn(_conv(x, weights[prefix + "expand"]["kernel"], None, dtype=dtype),
                              weights[prefix + "expand_BN"], dtype))
            ch = x.shape[1]
            if stride == 2:
                ph, pw = _correct_pad(x.shape[2]), _correct_pad(x.shape[3])
                x = Fnn.pad(x, (pw[0], pw[1], ph[0], ph[1]))
                x = _conv(x, weights[prefix + "depthwise"]["kernel"], None, stride=2, groups=ch, dtype=dtype)
            else:
                x = _conv(x, weights[prefix + "depthwise"]["kernel"], None, padding=1, groups=ch, dtype=dtype)
            x = relu6(_bn(x, weights[prefix + "depthwise_BN"], dtype))
            x = _bn(_conv(x, weights[prefix + "project"]["kernel"], None, dtype=dtype),
                    weights[prefix + "project_BN"], dtype)
            if cin == cout and stride == 1:
                x = inp + x
        x = relu6(_bn(_conv(x, weights["block_13_expand"]["kernel"], None, dtype=dtype),
                      weights["block_13_expand_BN"], dtype))
        feat = x
        reg, cls = _head(x, weights, dtype)
        out = [_nhwc(reg), _nhwc(cls)]
        return out + [_nhwc(feat)] if return_features else out


def rpn_forward(backbone, imgs_nhwc, weights, dtype=torch.float32, return_features=False):
    fn = vgg16_rpn_forward if backbone == "vgg16" else mobilenet_v2_rpn_forward
    return fn(imgs_nhwc, weights, dtype=dtype, return_features=return_features)


def conv2d_nhwc(x, w_hwio, bias=None, stride=1, pad=(0, 0, 0, 0), act=None, depthwise=False,
                dtype=torch.float32):
    """Single-layer helper for kernel unit tests.  pad = (top, bottom, left, right)."""
    with torch.no_grad():
        xt = _t(x, dtype).permute(0, 3, 1, 2).contiguous()
        xt = Fnn.pad(xt, (pad[2], pad[3], pad[0], pad[1]))
        y = _conv(xt, w_hwio, bias, stride=stride, groups=xt.shape[1] if depthwise else 1, dtype=dtype)
        if act == "relu":
            y = torch.relu(y)
        elif act == "relu6":
            y = torch.clamp(y, 0.0, 6.0)
        elif act == "sigmoid":
            y = torch.sigmoid(y)
        return _nhwc(y)
