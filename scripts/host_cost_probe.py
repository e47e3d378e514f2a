"""Host time of one MobileNetV2 one-image step, by part (enqueue only; the device is drained between parts)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf_rpn_amd.models._rpn_model import synthetic_weights
from tf_rpn_amd.predictor import Proposer
from tf_rpn_amd.utils import train_utils
hp = dict(train_utils.get_hyper_params("mobilenet_v2"))
w = synthetic_weights("mobilenet_v2", hp, seed=1)
prop = Proposer("mobilenet_v2", hyper_params=hp, weights=w, precision="f16x3", max_batch=1, overlap_nms=True)
x = torch.rand((1, 500, 500, 3), device="cuda")
for _ in range(20): prop.propose_async(x)
prop.wait(); torch.cuda.synchronize()
def host(fn, n=200):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6
reg, cls = prop._reg[:1], prop._cls[:1]
print("forward_into (16 launches, C++): %.1f us" % host(lambda: prop.rpn_model.forward_into(x, reg, cls)))
ob, osc, oi, ov = prop._boxes[:1], prop._scores[:1], prop._idx[:1], prop._valid[:1]
print("decode_nms (ctypes + launches):   %.1f us" % host(lambda: prop.decode_nms(reg.view(1, -1, 4), cls.view(1, -1), 1, ob, osc, oi, ov)))
print("propose_async (whole step):       %.1f us" % host(lambda: prop.propose_async(x)))
ev = torch.cuda.Event(); s = torch.cuda.current_stream()
print("event record + wait_event:        %.1f us" % host(lambda: (ev.record(s), s.wait_event(ev))))
def ctx():
    with torch.cuda.stream(prop._nms_stream): pass
print("with torch.cuda.stream(...):      %.1f us" % host(ctx))
print("_check_imgs:                      %.1f us" % host(lambda: prop._check_imgs(x)))
