"""Kernel time of one VGG16 layer as exact-float32 implicit GEMM vs float32 Winograd (run under rocprofv3 --kernel-trace --stats:
rpn_conv2d is a test entry that packs weights and synchronises, so only the profiler's kernel durations mean anything)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd import _lib as L
shapes = {"block3_conv2": (8, 125, 256, 256), "block2_conv2": (8, 250, 128, 128), "block5_conv1": (8, 31, 512, 512),
          "block1_conv2": (8, 500, 64, 64), "block4_conv2": (8, 62, 512, 512)}
for name in sys.argv[1:] or ["block3_conv2"]:
    B, H, Cin, Cout = shapes[name]
    torch.manual_seed(0)
    x = (torch.rand((B, H, H, Cin), device="cuda") - 0.5).contiguous()
    w = (torch.randn((3, 3, Cin, Cout), device="cuda") * (2.0 / (9 * Cin)) ** 0.5).contiguous()
    b = torch.rand((Cout,), device="cuda") - 0.5
    outs = {}
    for prec in ("f32", "f32w"):
        out = torch.empty((B, H, H, Cout), device="cuda")
        for _ in range(5):
            L.check(L.lib().rpn_conv2d(L.ptr(x), B, H, H, Cin, L.ptr(w), L.ptr(b), 3, 3, Cout, 1, 1, 1, H, H, L.ACTS["relu"],
                                       L.PRECISIONS[prec], L.ptr(out), L.stream_ptr()), "rpn_conv2d")
        torch.cuda.synchronize()
        outs[prec] = out
    ref = torch.nn.functional.conv2d(x[:1].permute(0, 3, 1, 2).double(), w.permute(3, 2, 0, 1).double(), b.double(), padding=1).relu().permute(0, 2, 3, 1)
    print(name, "max |f32w - f32| %.3e   |f32 - f64| %.3e   |f32w - f64| %.3e   max |y| %.2f" % (
        (outs["f32w"] - outs["f32"]).abs().max().item(), (outs["f32"][:1].double() - ref).abs().max().item(),
        (outs["f32w"][:1].double() - ref).abs().max().item(), ref.abs().max().item()), flush=True)
