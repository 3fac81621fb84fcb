"""Row f1: the capacitance CNN definitions and the batched inference engine.

Pinning: torchvision is absent, so the MobileNetV3 restatement is checked against the
published parameter counts of torchvision's mobilenet_v3_small / mobilenet_v3_large
(2 542 856 / 5 483 032 with their 1000-class classifiers => 927 008 / 2 971 952 in
`features`) and against the state_dict key layout the reference's checkpoints use.
Numerical parity of a *trained* model is unpinned: no checkpoint ships with the reference.
"""
import os

import numpy as np
import pytest
import torch

from qadapt_hip import capacitance_cnn as M


def _randomise_bn(model, seed=0):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)


@pytest.mark.parametrize("arch,features_params,feature_dim", [("small", 927_008, 576), ("large", 2_971_952, 960)])
def test_mobilenet_backbone_matches_published_parameter_counts(arch, features_params, feature_dim):
    b = M.MobileNetV3Backbone(arch, in_channels=3)
    assert sum(p.numel() for p in b.features.parameters()) == features_params
    assert b.feature_dim == feature_dim
    one = M.MobileNetV3Backbone(arch, in_channels=1)          # CapacitancePrediction.py:135-149
    assert sum(p.numel() for p in one.parameters()) == features_params - 2 * 16 * 9


def test_state_dict_layout_is_the_reference_checkpoint_layout():
    m = M.CapacitancePredictionModel(3)
    sd = m.state_dict()
    for k, shape in {"backbone.features.0.0.weight": (16, 1, 3, 3),
                     "backbone.features.0.1.running_var": (16,),
                     "backbone.features.1.block.0.0.weight": (16, 1, 3, 3),       # depthwise, no expand conv
                     "backbone.features.1.block.1.fc1.weight": (8, 16, 1, 1),
                     "backbone.features.1.block.2.0.weight": (16, 16, 1, 1),
                     "backbone.features.2.block.0.0.weight": (72, 16, 1, 1),
                     "backbone.features.4.block.2.fc2.bias": (96,),
                     "backbone.features.12.0.weight": (576, 96, 1, 1),
                     "value_head.0.weight": (256, 576), "value_head.3.weight": (128, 256),
                     "value_head.6.weight": (3, 128), "confidence_head.6.bias": (3,)}.items():
        assert tuple(sd[k].shape) == shape, k
    assert not any(k.startswith("backbone.classifier") for k in sd)
    imp = M.IMPALACapacitanceModel(3).state_dict()
    assert tuple(imp["backbone.cnn.0.weight"].shape) == (16, 1, 3, 3)
    assert tuple(imp["backbone.cnn.2.conv1.weight"].shape) == (16, 16, 3, 3)
    assert tuple(imp["backbone.cnn.5.weight"].shape) == (32, 16, 3, 3)
    assert tuple(imp["value_head.0.weight"].shape) == (128, 512)
    sep = M.SeparateHeadMobileNet().state_dict()
    assert tuple(sep["nnn_value_head.5.weight"].shape) == (2, 64) and tuple(sep["nn_value_head.5.weight"].shape) == (1, 64)
    sepi = M.SeparateHeadIMPALA().state_dict()
    assert tuple(sepi["nnn_confidence_head.3.weight"].shape) == (2, 64)


@pytest.mark.parametrize("res", [64, 100, 32])
def test_forward_shapes(res):
    x = torch.randn(3, 1, res, res)
    for model in (M.create_model(3), M.create_model(3, backbone="impala"), M.create_model(2, mobilenet="large")):
        model.eval()
        v, l = model(x)
        assert v.shape == l.shape == (3, model.output_size)
    for model in (M.create_model(3, separate_heads=True), M.create_model(3, backbone="impala", separate_heads=True)):
        model.eval()
        out = model(x)
        assert out["nn"][0].shape == (3, 1) and out["nnn"][1].shape == (3, 2)
        v, l = model.forward_combined(x)
        assert v.shape == (3, 3) and torch.equal(v[:, :1], out["nn"][0])


def test_folded_batchnorm_is_the_same_function():
    torch.manual_seed(1)
    m = M.CapacitancePredictionModel(3).eval()
    _randomise_bn(m)
    f = M.fold_batchnorm(m)
    assert not any(isinstance(x, torch.nn.BatchNorm2d) for x in f.modules())
    x = torch.rand(6, 1, 64, 64)
    with torch.no_grad():
        v, l = m(x); v2, l2 = f(x)
    assert torch.allclose(v, v2, atol=2e-5, rtol=1e-4) and torch.allclose(l, l2, atol=2e-5, rtol=1e-4)
    assert v.abs().max() > 0


def test_checkpoint_round_trip_both_forms(tmp_path):
    torch.manual_seed(2)
    src = M.CapacitancePredictionModel(3).eval()
    _randomise_bn(src)
    bare = tmp_path / "bare.pth"; full = tmp_path / "full.pth"
    torch.save(src.state_dict(), bare)
    torch.save({"model_state_dict": src.state_dict(), "epoch": 7, "val_loss": 0.1}, full)      # env.py:741-745
    x = torch.rand(2, 1, 64, 64)
    for p in (bare, full):
        dst = M.load_checkpoint(M.CapacitancePredictionModel(3), str(p)).eval()
        with torch.no_grad():
            assert torch.equal(dst(x)[0], src(x)[0])
    with pytest.raises(FileNotFoundError):
        M.load_checkpoint(M.CapacitancePredictionModel(3), str(tmp_path / "missing.pth"))
    with pytest.raises(RuntimeError):                       # wrong architecture: strict key matching
        M.load_checkpoint(M.IMPALACapacitanceModel(3), str(bare))


def test_engine_chunking_and_combined_heads_cpu():
    torch.manual_seed(3)
    m = M.CapacitancePredictionModel(3).eval()
    _randomise_bn(m)
    x = torch.rand(11, 1, 32, 32)
    eng1 = M.DeviceCapacitanceModel(m, device="cpu", chunk_images=4)
    eng2 = M.DeviceCapacitanceModel(m, device="cpu", chunk_images=64, channels_last=False, fold_bn=False)
    v1, l1 = eng1(x); v2, l2 = eng2(x)
    assert v1.shape == (11, 3) and v1.dtype == torch.float32
    assert torch.allclose(v1, v2, atol=2e-5, rtol=1e-4) and torch.allclose(l1, l2, atol=2e-5, rtol=1e-4)
    sep = M.DeviceCapacitanceModel(M.SeparateHeadIMPALA().eval(), device="cpu")
    v, l = sep(x)
    assert v.shape == l.shape == (11, 3)
    a = M.build_device_model(seed=5, device="cpu"); b = M.build_device_model(seed=5, device="cpu")
    assert torch.equal(a(x)[0], b(x)[0])


# ----------------------------------------------------------------------------- on the MI355X

def _env(B, N, R, model, **kw):
    from qadapt_hip.vec_env import VecQuantumDeviceEnv
    return VecQuantumDeviceEnv(B, num_dots=N, resolution=R, seed=77, validate=True, capacitance_model=model, **kw)


@pytest.mark.gpu
def test_cnn_in_the_step_equals_precomputed_outputs():
    """The step with the CNN on the device (images never leave the GPU) gives the same state as
    the step fed with outputs computed on the host from the same images by the unfused fp32 model."""
    torch.manual_seed(11)
    model = M.CapacitancePredictionModel(3).eval()
    _randomise_bn(model)
    for head in (model.value_head, model.confidence_head):      # outputs in the Kalman filter's working range
        head[6].weight.data.mul_(30.0)
    model.confidence_head[6].bias.data.fill_(-4.0)
    B, N, R = 6, 4, 32
    eng = M.DeviceCapacitanceModel(model, chunk_images=8)
    a = _env(B, N, R, eng); b = _env(B, N, R, eng)
    oa = a.reset(); img0 = oa["barrier_images"].cpu()
    with torch.no_grad():
        v0, l0 = model(img0.reshape(B * (N - 1), 1, R, R))
    b.reset(cnn_outputs=(v0.reshape(B, N - 1, 3).cuda(), l0.reshape(B, N - 1, 3).cuda()))
    sa, _ = a.get_state(); sb, _ = b.get_state()
    np.testing.assert_allclose(sa, sb, rtol=1e-4, atol=1e-5)
    rng = np.random.default_rng(0)
    for _ in range(2):
        act = torch.as_tensor(rng.uniform(-0.05, 0.05, (B, 2 * N - 1)).astype(np.float32)).cuda()
        oa, ra, _, _ = a.step(act)
        # feed b with host-computed outputs of a's images (b's own images agree to ~1e-6)
        with torch.no_grad():
            v, l = model(oa["barrier_images"].cpu().reshape(B * (N - 1), 1, R, R))
        ob, rb, _, _ = b.step(act, cnn_outputs=(v.reshape(B, N - 1, 3).cuda(), l.reshape(B, N - 1, 3).cuda()))
        np.testing.assert_allclose(ra.cpu().numpy(), rb.cpu().numpy(), rtol=1e-4, atol=1e-6)
        sa, _ = a.get_state(); sb, _ = b.get_state()
        np.testing.assert_allclose(sa, sb, rtol=1e-3, atol=1e-4)
    km = sa[:, a.L.s_kmean:a.L.s_kmean + N * N].reshape(B, N, N)
    assert np.abs(km[:, 0, 1] - 0.3).max() > 1e-3, "the Kalman filter never accepted a CNN output"
    a.close(); b.close()


@pytest.mark.gpu
def test_single_env_mirror_loads_a_checkpoint(tmp_path):
    """QuantumDeviceEnv(capacitance_model_checkpoint=...) as env.py:680-802: model built, weights loaded
    (weights_only), CNN run on the device inside reset()/step()."""
    from qadapt_hip.env import QuantumDeviceEnv
    torch.manual_seed(4)
    src = M.CapacitancePredictionModel(3)
    ck = tmp_path / "mobilenet_barrier_weights.pth"
    torch.save({"model_state_dict": src.state_dict()}, ck)
    env = QuantumDeviceEnv(num_dots=4, capacitance_model_checkpoint=str(ck))
    obs, info = env.reset()
    obs, rew, term, trunc, info = env.step({"action_gate_voltages": np.zeros(4, np.float32),
                                            "action_barrier_voltages": np.zeros(3, np.float32)})
    assert obs["image"].shape == (env.resolution, env.resolution, 3) and np.isfinite(rew["gates"]).all()
    env.close()
    with pytest.raises(RuntimeError, match="Model weights not found"):
        QuantumDeviceEnv(num_dots=4, capacitance_model_checkpoint=str(tmp_path / "nope.pth"))


@pytest.mark.gpu
def test_bf16_engine_close_to_fp32():
    torch.manual_seed(9)
    model = M.CapacitancePredictionModel(3).eval()
    _randomise_bn(model)
    x = torch.rand(64, 1, 64, 64, device="cuda")
    v32, l32 = M.DeviceCapacitanceModel(model)(x)
    v16, l16 = M.DeviceCapacitanceModel(model, dtype=torch.bfloat16)(x)
    scale = v32.abs().max().item() + 1e-6
    assert (v32 - v16).abs().max().item() < 0.1 * scale + 0.02
    with torch.no_grad():
        vc, lc = model(x.cpu())
    assert torch.allclose(v32.cpu(), vc, atol=1e-4, rtol=1e-3)
