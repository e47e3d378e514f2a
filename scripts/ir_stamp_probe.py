"""In-kernel phase timing of the fused MobileNetV2 block kernel (-DRPN_STAMP build of mnv2_block_kernels.hip):
RPN_HIP_LIB=tf_rpn_amd/csrc/librpn_hip_irstamp.so python scripts/ir_stamp_probe.py [block ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bbox_oracle as bo
from tf_rpn_amd import _lib as L
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights

raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_ir_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
hp = dict(bo.get_hyper_params("mobilenet_v2"))
B = 8
m = RPNModel("mobilenet_v2", hp, precision=os.environ.get("RPN_PROBE_PRECISION", "f32"), max_batch=B)
m.set_weights(synthetic_weights("mobilenet_v2", hp, seed=1))
x = torch.rand((B, 500, 500, 3), device="cuda")
F, K = m.feature_map_shape, m.anchor_count
reg = torch.empty((B, F, F, 4 * K), device="cuda"); cls = torch.empty((B, F, F, K), device="cuda")
ops = m.ops()
# run the model op by op is not possible from here: every ir_block launch overwrites the stamps, so the LAST ir_block
# (block_12) is what remains after a forward; RPN_IR_STAMP_ONLY=<op index> makes the kernel of that op the last one
# by running forwards with profiling to find steady state, then reading the stamps of the selected op via the env knob.
for _ in range(3): m.forward_into(x, reg, cls)
torch.cuda.synchronize()
n = 512 * 128
st = np.zeros(n, dtype=np.uint64)
assert raw.rpn_debug_read_ir_stamps(st.ctypes.data, n) == 0
st = st.reshape(512, 128).astype(np.int64)
live = st[:, 1] > 0
s = st[live]
nit = int(((s[0, 2:64] > 0).sum() + 3) // 4)
print("stamped op: whichever ir_block ran last with RPN_IR_STAMP_OP=%s; %d workgroups, %d steps" % (os.environ.get("RPN_IR_STAMP_OP"), live.sum(), nit))
med = lambda a: int(np.median(a))
print("  tile load: %d cycles" % med(s[:, 1] - s[:, 0]))
prev = s[:, 1]
for it in range(nit):
    e, p, d, b = s[:, 2 + 4 * it], s[:, 3 + 4 * it], s[:, 4 + 4 * it], s[:, 5 + 4 * it]
    x = [s[:, 64 + 4 * it + j] for j in range(4)]
    xs = " ".join("%6d" % (med(v - prev) if (v > 0).all() else -1) for v in x)
    print("  step %2d: E done +%6d  P done +%6d  D done +%6d  barrier +%6d   | service: window read, fetch issued, copied, stored: %s" % (it, med(e - prev) if (e > 0).all() else -1, med(p - prev), med(d - prev) if (d > 0).all() else -1, med(b - prev), xs))
    prev = b
print("  total %d cycles" % med(prev - s[:, 0]))
