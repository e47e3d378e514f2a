"""In-kernel phase timing of ir_block_x3_kernel at a given batch (default ONE image; -DRPN_STAMP build):
RPN_HIP_LIB=tf_rpn_amd/csrc/librpn_hip_irstamp.so RPN_IR_STAMP_OP=64,1 python scripts/ir_stamp_probe1.py [B] [img_size]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bbox_oracle as bo
from tf_rpn_amd import _lib as L
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights

raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_ir_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
size = int(sys.argv[2]) if len(sys.argv) > 2 else 500
hp = dict(bo.get_hyper_params("mobilenet_v2", img_size=size, feature_map_shape={500: 32, 1024: 64}[size]))
m = RPNModel("mobilenet_v2", hp, precision="f16x3", max_batch=B)
m.set_weights(synthetic_weights("mobilenet_v2", hp, seed=1))
x = torch.rand((B, size, size, 3), device="cuda")
F, K = m.feature_map_shape, m.anchor_count
reg = torch.empty((B, F, F, 4 * K), device="cuda"); cls = torch.empty((B, F, F, K), device="cuda")
for _ in range(5): m.forward_into(x, reg, cls)
torch.cuda.synchronize()
raw.rpn_debug_clear_ir_stamps()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
m.set_profiling(1)
m.forward_into(x, reg, cls)
torch.cuda.synchronize()
ms, _ = m.profile_ms()
print("per-op event times (us):", " ".join("%s=%.1f" % (o["name"].replace("_project", ""), 1e3 * t) for o, t in zip(m.ops(), ms)))
n = 512 * 128
st = np.zeros(n, dtype=np.uint64)
assert raw.rpn_debug_read_ir_stamps(st.ctypes.data, n) == 0
st = st.reshape(512, 128).astype(np.int64)
live = st[:, 1] > 0
s = st[live]
t0 = s[:, 0].min()
nit = int(((s[0, 2:64] > 0).sum() + 3) // 4)
med = lambda a: int(np.median(a))
print("op %s, B=%d: %d workgroups, %d steps; workgroup starts spread over %d cycles" % (os.environ.get("RPN_IR_STAMP_OP"), B, live.sum(), nit, s[:, 0].max() - t0))
print("  tile load: %d cycles" % med(s[:, 1] - s[:, 0]))
prev = s[:, 1]
for it in range(nit):
    e, p, d, b = s[:, 2 + 4 * it], s[:, 3 + 4 * it], s[:, 4 + 4 * it], s[:, 5 + 4 * it]
    print("  step %2d: E done +%6d  P done +%6d  D done +%6d  barrier +%6d" % (it, med(e - prev) if (e > 0).all() else -1, med(p - prev), med(d - prev) if (d > 0).all() else -1, med(b - prev)))
    prev = b
print("  loop end at %d cycles from the workgroup's start (median)" % med(prev - s[:, 0]))
if (s[:, 100] > 0).any():
    print("  seam: partial store starts +%d, ticket known +%d (after loop end)" % (med(s[:, 100] - prev), med(s[:, 101] - prev)))
lastw = s[:, 103] > 0
if lastw.any():
    l = s[lastw]
    print("  last arrivers (%d): partials summed +%d after ticket; outputs stored +%d; whole workgroup %d cycles" % (
        lastw.sum(), med(l[:, 102] - l[:, 101]) if (l[:, 101] > 0).all() else -1, med(l[:, 103] - l[:, 102]), med(l[:, 103] - l[:, 0])))
    print("  kernel span: first start -> last output %d cycles" % (s[:, 103].max() - t0))

rt = s[:, 104]
end = s[:, 105]
ok = (rt > 0) & (end > 0)
print("  wall clock (100 MHz ticks -> us): workgroup starts spread %.2f us; first start -> last end %.2f us; median workgroup %.2f us" % (
    (rt[ok].max() - rt[ok].min()) / 100.0, (end[ok].max() - rt[ok].min()) / 100.0, np.median(end[ok] - rt[ok]) / 100.0))
