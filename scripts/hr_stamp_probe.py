"""In-kernel phase timing of ir_block_hrx3_kernel (MobileNetV2 blocks 1-3 under f16x3; -DRPN_STAMP build of
mnv2_block_kernels.hip: scripts/build_ir_stamp.sh):  RPN_HIP_LIB=tf_rpn_amd/csrc/librpn_hip_irstamp.so python scripts/hr_stamp_probe.py [B]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import bbox_oracle as bo
from tf_rpn_amd import _lib as L
from tf_rpn_amd.models._rpn_model import RPNModel, synthetic_weights
raw = ctypes.CDLL(L.LIB_PATH)
raw.rpn_debug_read_hr_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
hp = dict(bo.get_hyper_params("mobilenet_v2"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
m = RPNModel("mobilenet_v2", hp, precision="f16x3", max_batch=B)
m.set_weights(synthetic_weights("mobilenet_v2", hp, seed=1))
x = torch.rand((B, 500, 500, 3), device="cuda")
for _ in range(3): m.predict_on_batch(x)
torch.cuda.synchronize()
n = 3 * 256 * 40
st = np.zeros(n, dtype=np.uint64)
assert raw.rpn_debug_read_hr_stamps(st.ctypes.data, n) == 0
st = st.reshape(3, 256, 40).astype(np.int64)
med = lambda a: int(np.median(a))
for v, name in enumerate(("block 1 <16,96,16,24,s2>", "block 2 <24,144,48,24,s1,res>", "block 3 <24,144,16,32,s2>")):
    s = st[v][st[v][:, 39] > 0]
    if not len(s): continue
    nch = int((s[0, 2:38] > 0).sum() // 3)
    print("%s: %d workgroups stamped, %d chunks, total %d cycles" % (name, len(s), nch, med(s[:, 39] - s[:, 0])))
    print("   A operand (halo load + split) %d" % med(s[:, 1] - s[:, 0]))
    prev = s[:, 1]
    E = []; D = []; P = []
    for c in range(nch):
        E.append(med(s[:, 2 + 3 * c] - prev)); D.append(med(s[:, 3 + 3 * c] - s[:, 2 + 3 * c])); P.append(med(s[:, 4 + 3 * c] - s[:, 3 + 3 * c]))
        prev = s[:, 4 + 3 * c]
    print("   per chunk  E (+barrier): %s" % E); print("              D (+barrier): %s" % D); print("              P: %s" % P)
    print("   outputs %d" % med(s[:, 39] - prev))
