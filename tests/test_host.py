"""CPU suite, part 2: host logic and the C-ABI boundary (no compute calls without a GPU)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

import __graft_entry__ as entry
from tf_rpn_amd import _lib as L
from tf_rpn_amd.utils import train_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        entry.build()
    return L.lib()


# ---- C ABI ------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "rpn_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(rpn_[a-z0-9_]+)\s*\(", header)))
    assert len(declared) >= 25
    raw = ctypes.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), "librpn_hip.so does not export %s" % name
    assert declared == L.exported_symbols(), "ctypes table and include/rpn_hip.h disagree"
    assert lib.rpn_abi_version() == 1


def test_shipped_library_reads_only_the_documented_knobs(lib):
    """Product vs laboratory: the environment names compiled into librpn_hip.so are exactly the `RPN_KNOB` list of
    include/rpn_hip.h.  A/B switches and the timing experiments that produce wrong results on purpose (RPN_NMS_STOP,
    RPN_IOU_EXP, RPN_SPLIT_*, ...) exist only in `make lab` builds (-DRPN_LAB), so a stray environment variable cannot
    change what the product computes."""
    header = open(os.path.join(ROOT, "include", "rpn_hip.h")).read()
    documented = set(re.findall(r"RPN_KNOB (RPN_[A-Z0-9_]+)", header))
    assert len(documented) >= 8
    blob = open(L.LIB_PATH, "rb").read()
    in_binary = {m.group(1).decode() for m in re.finditer(rb"\x00(RPN_[A-Z0-9_]+)(?=\x00)", blob)}
    assert in_binary == documented, (sorted(in_binary - documented), sorted(documented - in_binary))
    for name in ("RPN_NMS_STOP", "RPN_IOU_EXP", "RPN_SPLIT_TILE", "RPN_S16_DMA", "RPN_IR_STAMP_OP"):
        assert name.encode() not in blob


def test_nms_kernels_use_no_scratch_memory(lib):
    """Code-object metadata of the shipped library: the NMS kernels keep everything in registers (round 2: 46 spilled VGPRs =
    180 B of scratch per thread = 11.8 MB written per launch at 64 images), and no kernel of the library uses scratch at all."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import codeobj
    ks = codeobj.kernels(L.LIB_PATH)
    nms = {n: k for n, k in ks.items() if "nms_kernel" in n or "nms_merge" in n}
    assert len(nms) == 3
    for name, k in nms.items():
        assert k[".private_segment_fixed_size"] == 0 and k.get(".vgpr_spill_count", 0) == 0, (name, k)
        assert k[".vgpr_count"] <= 128                       # 1024-thread workgroups: 4 waves per SIMD
    worst = max(ks.values(), key=lambda k: k[".private_segment_fixed_size"])
    assert worst[".private_segment_fixed_size"] == 0, worst[".name"]      # no kernel of the library spills to scratch


# Register / LDS budgets of the kernels whose speed depends on how many workgroups a CU holds (512 registers per SIMD lane, 160 KB of
# LDS per CU).  {kernel-name regex: (max VGPRs incl. AGPRs, max SGPR spills, max LDS bytes)}; every kernel matching a pattern must
# fit.  Round 4 shipped the exact-f32 dominant kernel at 227 registers instead of 92 (a run-time branch in its epilogue) and nothing
# noticed: the figures DESIGN.md quotes are held here.
KERNEL_BUDGETS = {
    r"conv_igemm_f32_dma<2,2,2,2,(false|true),(false|true)>": (96, 0, 32768),   # five 4-wave workgroups per CU (LEAN / general epilogue)
    r"conv_igemm_f32<2,2,2,2,false,(false|true),(false|true)>": (128, 0, 33024),     # four
    r"conv_igemm_f32(_dma)?<4,1,1,[12],": (80, 0, 24832),               # six
    r"conv3x3_split16_dma_kernel<(false|true),(false|true),128,false>": (256, 52, 163840),   # one 8-wave workgroup, 2 waves / SIMD: the cliff is 256 (round 6, halo_swz = hx & 6: 247-249, was 226-227)
    r"conv3x3_split16_dma_kernel<(false|true),(false|true),64,false>": (176, 52, 113664),
    r"conv3x3_split16_dma_kernel<(false|true),false,64,true>": (240, 88, 113664),            # K-tree fold: two more accumulator sets
    r"conv3x3_split16_kernel<8,2,8,": (240, 0, 136448),
    r"nms_kernel<(false|true)>": (128, 104, 0),                          # 1024 threads = 4 waves per SIMD (dynamic LDS); SGPR spills go to VGPR lanes
    r"stem_block_kernel<8>": (96, 0, 27648),                             # five workgroups per CU
    r"stem_block_kernel<4>": (80, 0, 16128),
    r"ir_block_hrx3_kernel<16,96,32,24,2,false,1,1>": (112, 0, 27136),   # MobileNetV2 block 1 at batch 8: four per CU
    r"ir_block_hrx3_kernel<24,144,48,24,1,true,1,1>": (104, 0, 21504),   # block 2
    r"ir_block_x3_kernel<": (224, 0, 100368),
    r"pw_x3_kernel<96,96,8>": (80, 0, 51200),
    r"iou_map_rows_kernel<": (128, 0, 0),
    r"rpn_head_kernel<": (192, 0, 49152),
    r"conv3x3_wino_f32_kernel": (168, 0, 160 * 1024),                   # twelve waves (8 MFMA + 4 staging): three per SIMD
    r"conv3x3_wino4_f32_kernel": (128, 180, 160 * 1024),                # sixteen waves (12 MFMA + 4 staging): four per SIMD; persistent (round 6): 174 scalar spill instructions into VGPR lanes, all in per-TILE code -- none inside the slice loops (checked on the ISA)
    r"conv_cin3_f32_mfma_kernel": (168, 0, 20 * 1024),                  # first layer of the float32 graphs: three persistent workgroups per CU
    r"conv3x3_wino4n_f32_kernel": (128, 8, 160 * 1024),                 # the wide form: the same sixteen waves (persistent: four scalars parked in lanes)
}


def test_kernel_register_budgets(lib):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import codeobj
    tab = codeobj.table(L.LIB_PATH)
    assert len(tab) >= 150
    for pattern, (max_vgpr, max_sspill, max_lds) in KERNEL_BUDGETS.items():
        hits = {n: r for n, r in tab.items() if re.match(pattern, n)}
        assert hits, "no kernel of librpn_hip.so matches %r" % pattern
        for name, (vgpr, sspill, vspill, scratch, lds, _wg) in hits.items():
            assert vgpr <= max_vgpr, "%s: %d VGPRs > %d" % (name, vgpr, max_vgpr)
            assert sspill <= max_sspill, "%s: %d SGPR spills > %d" % (name, sspill, max_sspill)
            assert vspill == 0 and scratch == 0, "%s spills to scratch" % name
            assert lds <= max_lds, "%s: %d B of LDS > %d" % (name, lds, max_lds)


def test_no_unguarded_store_data_hazard(lib, tmp_path):
    """gfx950 needs one wait state between a 16-byte buffer store whose scalar offset is a REGISTER and a vector instruction that
    overwrites the store's data registers; hipcc inserts none (it guards the constant-offset form only) and the first float32-MFMA
    first-layer kernel stored an LDS address into 0.5 % of its outputs (scripts/micro/store_hazard.hip measures the rule).  The shipped
    library's disassembly must not contain the pattern."""
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import codeobj
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        objdump = shutil.which("llvm-objdump")
    if not objdump:
        pytest.skip("no llvm-objdump")
    bad = codeobj.store_data_hazards("_Z3foov:\n buffer_store_dwordx4 v[18:21], v26, s[12:15], s54 offen\n v_add_u32_e32 v18, 0x1000, v120\n")
    assert len(bad) == 1                                    # (the scanner itself: the instruction pair of the bug)
    assert not codeobj.store_data_hazards("_Z3foov:\n buffer_store_dwordx4 v[18:21], v26, s[12:15], s54 offen\n s_nop 0\n v_add_u32_e32 v18, 0x1000, v120\n")
    n = stores = 0
    for elf in codeobj.code_objects(L.LIB_PATH):
        f = tmp_path / ("co%d.elf" % n)
        f.write_bytes(elf)
        text = subprocess.run([objdump, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, check=True).stdout
        stores += text.count("buffer_store_dwordx4")
        found = codeobj.store_data_hazards(text)
        assert not found, found[:3]
        n += 1
    assert n >= 8 and stores > 500                          # (every code object of the library was disassembled)


def test_no_torch_types_in_the_abi():
    header = open(os.path.join(ROOT, "include", "rpn_hip.h")).read()
    code = re.sub(r"/\*.*?\*/", "", header, flags=re.S)              # comments may mention torch storage
    assert "torch" not in code.lower() and "at::" not in code and "#include <hip" not in code


def test_product_never_imports_the_oracle():
    for base, _dirs, files in os.walk(os.path.join(ROOT, "tf_rpn_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(base, f)).read()
                assert "oracle" not in src, "%s mentions the oracle" % os.path.join(base, f)


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_compute_calls_fail_loudly_without_a_device(lib):
    assert lib.rpn_device_count() == 0
    buf = (ctypes.c_float * 16)()
    st = lib.rpn_decode(ctypes.cast(buf, L.vp), 0, ctypes.cast(buf, L.vp), None, 1, 1, ctypes.cast(buf, L.vp), None)
    assert st == L.RPN_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.rpn_last_error()
    from tf_rpn_amd.utils import bbox_utils
    with pytest.raises(RuntimeError):
        bbox_utils.generate_anchors(train_utils.get_hyper_params("vgg16"))


def test_argument_validation_precedes_device_use(lib):
    st = lib.rpn_combined_nms(None, None, 1, 10, 2, 3, 5, 5, 0.5, 0.0, 1, None, None, None, None, None, None, 0, None)
    assert st == L.RPN_ERR_INVALID and b"q must be 1 or C" in lib.rpn_last_error()
    h = L.vp(0)
    assert lib.rpn_model_create(7, 500, 9, 0, 1, ctypes.byref(h)) == L.RPN_ERR_INVALID


# ---- native graph builder (runs without a GPU: device buffers are allocated lazily) --------------
@pytest.mark.parametrize("backbone,img,K,F,gflops,nlayers", [
    ("vgg16", 500, 9, 31, 156.552, 16),
    ("mobilenet_v2", 500, 9, 32, 7.737, 43),
    ("vgg16", 1024, 15, 64, 661.07, 16),
    ("mobilenet_v2", 1024, 15, 64, 31.282, 43),
])
def test_graph_builder_shapes_and_flops(lib, backbone, img, K, F, gflops, nlayers):
    from tf_rpn_amd.models._rpn_model import RPNModel
    m = RPNModel(backbone, {"img_size": img, "anchor_count": K}, max_batch=2)
    assert m.feature_map_shape == F
    assert abs(m.flops_per_image / 1e9 - gflops) < 0.01
    assert len(m.layers) == nlayers
    names = [l["name"] for l in m.layers]
    assert names[-3:] == ["rpn_conv", "rpn_reg", "rpn_cls"]
    assert m.layers[-2]["shape"] == (1, 1, 512, 4 * K) and m.layers[-1]["shape"] == (1, 1, 512, K)
    if backbone == "vgg16":
        assert names[0] == "block1_conv1" and m.layers[0]["shape"] == (3, 3, 3, 64)
        assert m.activation_shape("block3_pool")[1:] == (img // 8, img // 8, 256)
    else:
        assert m.layers[0]["bn_name"] == "bn_Conv1" and m.layers[1]["kind"] == 2
        assert m.activation_shape("block_13_expand")[1:] == (F, F, 576)
        # the intermediate tensors of a block exist only on the layer-by-layer graph (fused blocks keep them on chip)
        mk = RPNModel(backbone, {"img_size": img, "anchor_count": K}, max_batch=2, keep_activations=True)
        sides = [mk.activation_shape(n)[1] for n in ("Conv1", "block_1_depthwise", "block_3_depthwise", "block_6_depthwise")]
        assert len(m.ops()) == 16 and len(mk.ops()) > 40     # one launch per block
        assert sides == ([250, 125, 63, 32] if img == 500 else [512, 256, 128, 64])
    w_bytes, arena_bytes = m.memory_bytes()
    assert w_bytes > 0 and arena_bytes > 0


# ---- config --------------------------------------------------------------------------------
def test_get_hyper_params_contract():
    import copy
    saved = copy.deepcopy(train_utils.RPN)
    try:
        hp = train_utils.get_hyper_params("vgg16")
        assert hp is train_utils.RPN["vgg16"]                       # mutates the module dict, as the reference
        assert (hp["img_size"], hp["feature_map_shape"], hp["anchor_count"], hp["test_nms_topn"]) == (500, 31, 9, 300)
        assert hp["variances"] == [0.1, 0.1, 0.2, 0.2] and hp["anchor_ratios"] == [1.0, 2.0, 0.5]
        hp = train_utils.get_hyper_params("mobilenet_v2", img_size=1024, feature_map_shape=64,
                                          anchor_ratios=[1., 2., .5, 3., 1 / 3.], unknown_key=5, test_nms_topn=0)
        assert hp["anchor_count"] == 15 and hp["img_size"] == 1024 and "unknown_key" not in hp
        assert hp["test_nms_topn"] == 300                          # falsy override ignored
    finally:
        train_utils.RPN.clear()
        train_utils.RPN.update(saved)


def test_hyper_params_agree_with_oracle():
    import copy
    from oracle import bbox_oracle as bo
    saved = copy.deepcopy(train_utils.RPN)
    try:
        for bb in ("vgg16", "mobilenet_v2"):
            assert train_utils.get_hyper_params(bb) == bo.get_hyper_params(bb)
    finally:
        train_utils.RPN.clear()
        train_utils.RPN.update(saved)


# ---- sharding / record packing ----------------------------------------------------------------
def test_shard_bounds_cover_the_batch():
    from tf_rpn_amd.predictor import shard_bounds
    for total in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_record_pack_unpack_round_trip():
    from tf_rpn_amd.predictor import Proposer
    B, M = 3, 7
    boxes, scores = torch.rand(B, M, 4), torch.rand(B, M)
    valid = torch.tensor([7, 0, 3], dtype=torch.int32)
    rec = Proposer.pack_records(None, boxes, scores, valid)
    assert rec.shape == (B, M * 5 + 1)
    b2, s2, v2 = Proposer.unpack_records(rec, M)
    assert torch.equal(b2, boxes) and torch.equal(s2, scores) and torch.equal(v2, valid)


def test_io_utils_mirror(tmp_path, monkeypatch):
    """io_utils.py:17-50 of the reference: model path (creates ./trained), the two CLI flags, backbone check."""
    from tf_rpn_amd.utils import io_utils
    monkeypatch.chdir(tmp_path)
    assert io_utils.get_model_path("rpn", "vgg16") == os.path.join("trained", "rpn_vgg16_model_weights.h5")
    assert os.path.isdir(tmp_path / "trained")
    assert io_utils.get_model_path("faster_rcnn", "mobilenet_v2").endswith("faster_rcnn_mobilenet_v2_model_weights.h5")
    a = io_utils.handle_args([])
    assert a.backbone == "mobilenet_v2" and a.handle_gpu is False
    a = io_utils.handle_args(["-handle-gpu", "--backbone", "vgg16"])
    assert a.backbone == "vgg16" and a.handle_gpu is True
    io_utils.is_valid_backbone("vgg16")
    with pytest.raises(AssertionError):
        io_utils.is_valid_backbone("resnet50")
    assert io_utils.handle_gpu_compatibility() is None


def test_custom_image_generator(tmp_path):
    """data_utils.py:106-136: files of the folder (not recursive), PIL Lanczos resize, float32 [0,1], empty gt."""
    from PIL import Image
    from tf_rpn_amd.utils import data_utils
    rng = np.random.RandomState(0)
    (tmp_path / "sub").mkdir()
    for name, (h, w) in (("a.png", (30, 50)), ("b.png", (64, 40))):
        Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(tmp_path / name)
    Image.fromarray(np.zeros((8, 8, 3), np.uint8)).save(tmp_path / "sub" / "nested.png")
    paths = sorted(data_utils.get_custom_imgs(str(tmp_path)))
    assert [os.path.basename(p) for p in paths] == ["a.png", "b.png"]
    out = list(data_utils.custom_data_generator(paths, 20, 24))
    assert len(out) == 2
    for (img, boxes, labels), path in zip(out, paths):
        assert img.shape == (20, 24, 3) and img.dtype == np.float32 and 0.0 <= img.min() and img.max() <= 1.0
        ref = np.asarray(Image.open(path).resize((24, 20), Image.LANCZOS), dtype=np.uint8)
        np.testing.assert_array_equal(img, ref.astype(np.float32) * np.float32(1.0 / 255.0))
        assert boxes.shape == (1, 0) and boxes.dtype == np.float32 and labels.shape == (0,) and labels.dtype == np.int32


def test_get_step_size():
    """train_utils.py:40-48: math.ceil(total_items / batch_size)."""
    import math
    for total, bs in ((4952, 4), (4952, 8), (1, 8), (16, 8), (17, 8), (5011, 3)):
        assert train_utils.get_step_size(total, bs) == math.ceil(total / bs)


# ---- bench.py launches its own ranks (python bench.py --gpus N without torchrun) -------------------------------------
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_spawn_ranks_environment_and_failure_handling():
    import time
    bench = _bench_module()
    code = "import os; print(os.environ['RANK'], os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']) > 0)"
    rc, out = bench.spawn_ranks(3, [sys.executable, "-c", code])
    assert rc == 0 and out.strip() == "0 0 3 127.0.0.1 True"          # rank 0's stdout only
    # one rank fails: its code is returned and the sleeping ranks are terminated, not waited for
    t0 = time.time()
    rc, _ = bench.spawn_ranks(3, [sys.executable, "-c",
                                  "import os, sys, time; sys.exit(7) if os.environ['RANK'] == '2' else time.sleep(60)"])
    assert rc == 7 and time.time() - t0 < 30
    t0 = time.time()
    rc, _ = bench.spawn_ranks(2, [sys.executable, "-c", "import time; time.sleep(60)"], timeout=1.0)
    assert rc == 124 and time.time() - t0 < 30


def test_bench_gpus_n_without_launcher_fails_cleanly_without_devices():
    """`python bench.py --gpus 2` on a box with fewer than 2 HIP devices: non-zero exit, a message, no hang, no JSON."""
    import subprocess
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has two devices")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == "" and "--gpus 2" in r.stderr
