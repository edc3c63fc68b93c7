"""CPU checks of the perceptual-loss oracle (restated LPIPS graph; PARITY UNPINNED: the lpips package and its weights
are absent from the reference tree) and of the host logic that maps user-supplied LPIPS / torchvision state dicts."""
import torch

from oracle import orbit2_oracle as O


def test_lpips_oracle_properties():
    sd = O.init_lpips_weights(0)
    g = torch.Generator().manual_seed(1)
    a = torch.randn(2, 3, 32, 48, generator=g)
    b = torch.randn(2, 3, 32, 48, generator=g)
    dab, dba = O.lpips_vgg(a, b, sd), O.lpips_vgg(b, a, sd)
    assert dab.shape == (2,) and (dab > 0).all()
    assert torch.allclose(dab, dba, rtol=1e-5)                       # symmetric
    assert float(O.lpips_vgg(a, a, sd).abs().max()) == 0.0            # identical images
    # every tap contributes at most sum(lin) (unit vectors: |n0 - n1|^2 <= 2 per pixel, weighted)
    bound = sum(float(sd["lin%d.weight" % k].sum()) * 2.0 for k in range(5))
    assert float(dab.max()) <= bound
    want = (a - b).abs().mean() + 0.5 * dab.mean()
    assert torch.allclose(O.perceptual(a, b, sd), want)


def test_lpips_oracle_taps_are_the_documented_vgg16_layers():
    # relu1_2, relu2_2, relu3_3, relu4_3, relu5_3 = conv indices 1, 3, 6, 9, 12 with 64,128,256,512,512 channels
    sd = O.init_lpips_weights(0)
    assert [sd["lin%d.weight" % k].numel() for k in range(5)] == [64, 128, 256, 512, 512]
    assert [sd["conv%d.weight" % i].shape[0] for i in O.VGG16_TAPS] == [64, 128, 256, 512, 512]
    assert sum(1 for c in O.VGG16_CFG if c != "M") == 13 and O.VGG16_CFG.count("M") == 4


def test_state_dict_name_mapping_and_stand_in_weights():
    from climate_learn.metrics.lpips_hip import _canonical, random_lpips_state, _TV_CONV_IDX
    ours, orc = random_lpips_state(3), O.init_lpips_weights(3)
    assert ours.keys() == orc.keys() and all(torch.equal(ours[k], orc[k]) for k in ours)
    # lpips naming: net.slice{s}.{torchvision index}.weight, lin{k}.model.1.weight [1,C,1,1]
    lp = {}
    slices = (1, 1, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5)
    for i, idx in enumerate(_TV_CONV_IDX):
        lp["net.slice%d.%d.weight" % (slices[i], idx)] = orc["conv%d.weight" % i]
        lp["net.slice%d.%d.bias" % (slices[i], idx)] = orc["conv%d.bias" % i]
    for k in range(5):
        lp["lin%d.model.1.weight" % k] = orc["lin%d.weight" % k].view(1, -1, 1, 1)
    lp["scaling_layer.shift"] = torch.zeros(1, 3, 1, 1)
    got = _canonical(lp)
    assert got.keys() == orc.keys() and all(torch.equal(got[k], orc[k]) for k in orc)
    tv = {"features.%d.weight" % idx: orc["conv%d.weight" % i] for i, idx in enumerate(_TV_CONV_IDX)}
    got = _canonical(tv)
    assert all(torch.equal(got["conv%d.weight" % i], orc["conv%d.weight" % i]) for i in range(13))
