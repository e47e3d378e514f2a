"""Keras .h5 weights reader (tf_rpn_amd/utils/h5_weights.py) against files written by the real h5py / libhdf5
(tests/golden/*.h5; generator tests/golden/make_h5_fixtures.py, run with /opt/conda/bin/python3.9)."""
import os
import zlib

import numpy as np
import pytest

from tf_rpn_amd.utils import h5_weights as hw

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def expected_array(path, shape):                      # same formula as the fixture generator
    n = int(np.prod(shape)) if len(shape) else 1
    k = zlib.crc32(path.encode()) % 97
    return ((np.arange(n, dtype=np.float64) * 0.25 + k) * (-1.0) ** k).astype(np.float32).reshape(shape)


VGG_TINY_SHAPES = {"block1_conv1": {"kernel": (3, 3, 3, 4), "bias": (4,)}, "rpn_conv": {"kernel": (3, 3, 4, 8), "bias": (8,)},
                   "rpn_cls": {"kernel": (1, 1, 8, 9), "bias": (9,)}, "rpn_reg": {"kernel": (1, 1, 8, 36), "bias": (36,)}}


@pytest.mark.parametrize("name,full", [("keras_weights_fixed_strings", False), ("keras_weights_vlen_strings", False),
                                       ("keras_full_model", True), ("keras_weights_latest_libver", False),
                                       ("keras_weights_chunked_attrs", False)])
def test_reads_keras_layout_written_by_h5py(name, full):
    w, info = hw.read_keras_weights(os.path.join(GOLD, name + ".h5"))
    assert info["layer_names"] == ["input_1", "block1_conv1", "block1_pool", "rpn_conv", "rpn_cls", "rpn_reg"]
    assert info["backend"] == "tensorflow" and info["keras_version"] == "2.2.4-tf" and info["full_model"] is full
    assert w["input_1"] == {} and w["block1_pool"] == {}
    for layer, params in VGG_TINY_SHAPES.items():
        assert list(w[layer]) == list(params)                     # file order: kernel, bias
        for p, shape in params.items():
            got = w[layer][p]
            assert got.dtype == np.float32 and got.shape == shape
            np.testing.assert_array_equal(got, expected_array("%s/%s/%s:0" % (layer, layer, p), shape))


def test_many_groups_batchnorm_and_depthwise_names():
    w, info = hw.read_keras_weights(os.path.join(GOLD, "keras_weights_many_layers.h5"))
    assert len(info["layer_names"]) == 3 + 3 * 16 and info["layer_names"][:3] == ["input_1", "Conv1", "bn_Conv1"]
    n = 0
    for layer, params in w.items():
        for p, got in params.items():
            np.testing.assert_array_equal(got, expected_array("%s/%s/%s:0" % (layer, layer, p), got.shape))
            n += 1
    assert n == 1 + 4 + 16 * (1 + 1 + 4)
    assert set(w["bn_Conv1"]) == {"gamma", "beta", "moving_mean", "moving_variance"}
    assert w["block_7_depthwise"]["depthwise_kernel"].shape == (3, 3, 16, 1)
    arrays = hw.to_layer_arrays(w)
    assert "input_1" not in arrays
    assert set(arrays["block_7_depthwise_BN"]) == {"gamma", "beta", "mean", "var"}
    assert arrays["block_7_depthwise"]["kernel"].shape == (3, 3, 16, 1)


def test_unsupported_features_fail_loudly(tmp_path):
    with pytest.raises(NotImplementedError, match="compression|chunked"):
        hw.read_keras_weights(os.path.join(GOLD, "keras_weights_chunked_gzip.h5"))
    bad = tmp_path / "not.h5"
    bad.write_bytes(b"PK\x03\x04" + b"\0" * 100)
    with pytest.raises(hw.H5FormatError):
        hw.read_keras_weights(str(bad))
    good = open(os.path.join(GOLD, "keras_weights_fixed_strings.h5"), "rb").read()
    cut = tmp_path / "cut.h5"
    cut.write_bytes(good[:6000])                                   # datasets live past 8 KB: truncated file
    with pytest.raises((hw.H5FormatError, IndexError, ValueError)):
        hw.read_keras_weights(str(cut))


def test_reader_accepts_bytes():
    data = open(os.path.join(GOLD, "keras_weights_vlen_strings.h5"), "rb").read()
    w, _ = hw.read_keras_weights(data)
    assert w["rpn_reg"]["bias"].shape == (36,)
