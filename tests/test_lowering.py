"""CPU: the Keras-layer-list -> op-program lowering (keras_model.compile_layers), executed by tests/prog_interp.py
(torch-CPU float64), against the Keras-semantics oracle -- every sweep topology, including the channel-padded ones."""
import numpy as np
import pytest

from inaspeechsegmenter_amd import keras_model as KM, _native as N
from oracle import keras_cnn as ocnn
import prog_interp
import topologies as TP


@pytest.mark.parametrize('name', sorted(TP.SPECS))
def test_lowered_program_equals_oracle(name):
    rng = np.random.default_rng(3)
    for net, (layers, shp) in TP.nets(name).items():
        x = rng.normal(0, 1, (3,) + shp).astype(np.float32)
        comp = KM.compile_layers(layers, shp)
        got = prog_interp.run(comp, x)
        want = ocnn.forward(layers, x)
        assert got.shape == want.shape and np.abs(got - want).max() < 2e-5, (name, net, np.abs(got - want).max())
        # algorithmic flops ignore channel padding (fused 'valid' pools may drop an odd last row / column)
        assert KM.compile_layers(layers, shp, fuse_pool=False).flops_per_sample == ocnn.flops_per_sample(layers, shp)
        assert comp.flops_per_sample <= ocnn.flops_per_sample(layers, shp)
        # every conv behind the first layer reads a multiple of 32 channels (vectorised MFMA kernels)
        for R in comp.prog[1:]:
            if R[N.C_OP] == N.OP_CONV:
                assert R[N.C_CIN] % 32 == 0, (name, net, R[N.C_CIN])


def test_channel_padding_is_transparent():
    layers, shp = TP.build(TP.SPECS['ch48_96'], 21, 3, 5)
    a = KM.compile_layers(layers, shp, pad_channels=True)
    b = KM.compile_layers(layers, shp, pad_channels=False)
    assert [int(r[N.C_COUT]) for r in a.prog if r[N.C_OP] == N.OP_CONV][:4] == [64, 64, 96, 96]
    assert [int(r[N.C_COUT]) for r in b.prog if r[N.C_OP] == N.OP_CONV][:4] == [48, 48, 96, 96]
    x = np.random.default_rng(0).normal(0, 1, (2,) + shp).astype(np.float32)
    assert np.abs(prog_interp.run(a, x) - prog_interp.run(b, x)).max() < 1e-12
    assert a.flops_per_sample == b.flops_per_sample and a.out_dim == b.out_dim == 3


def test_resnet101_lowering_equals_oracle():
    """keras_model.compile_resnet101 (BatchNorm folding, residual wiring with in-place adds, stride-2 shortcuts, statistics
    pooling order, embedding layer) executed on CPU against the torch restatement of resnet.py:115-130 (oracle/vbx.py)."""
    from oracle import vbx as ovbx
    params = KM.synthetic_resnet101(3)
    comp = KM.compile_resnet101(params)
    x = np.random.default_rng(5).normal(0, 1, (2, 64, 144)).astype(np.float32)
    ref = ovbx.resnet101_forward(params, x)
    got = prog_interp.run(comp, x[..., None])
    assert got.shape == ref.shape == (2, 256)
    assert np.abs(got - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())

