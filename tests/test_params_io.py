"""Checkpoint ingestion: MXNet `.params` files, the layout conversion into the graph, and the
init_from_resnet rules.  The byte layouts below are written out by hand from MXNet's NDArray::Save / LegacyLoad."""
import os
import struct

import numpy as np
import pytest

from dspnet_amd import model, multi_init


def _hex(*parts):
    return b"".join(bytes.fromhex(p.replace(" ", "")) for p in parts)


F32 = lambda *v: struct.pack("<%df" % len(v), *v)  # noqa: E731

# list header: magic 0x112, reserved 0, two arrays
HEAD2 = _hex("1201000000000000", "0000000000000000", "0200000000000000")
# V2 record: magic f993fac9, storage 0, ndim 2, dims (2, 3) as int64, context cpu(0) = (1, 0), type flag 0
REC_V2 = _hex("c9fa93f9", "00000000", "02000000", "0200000000000000", "0300000000000000", "01000000", "00000000",
              "00000000") + F32(1, 2, 3, 4, 5, 6)
# legacy record (before 0.12): first word is ndim = 1, dims as uint32 (3,), context gpu(0) = (2, 0), type flag 0
REC_LEGACY = _hex("01000000", "03000000", "02000000", "00000000", "00000000") + F32(-1, 0.5, 7)
# V1 record: magic f993fac8, ndim 1, dim 2 as int64, context, type flag 4 (int32)
REC_V1 = _hex("c8fa93f9", "01000000", "0200000000000000", "01000000", "00000000", "04000000") + struct.pack("<2i", 9, -9)
NAMES2 = _hex("0200000000000000", "0c00000000000000") + b"arg:fc_weigh" + _hex("0a00000000000000") + b"aux:bn_mea"


def test_reader_v2_and_legacy_records(tmp_path):
    f = tmp_path / "a-0001.params"
    f.write_bytes(HEAD2 + REC_V2 + REC_LEGACY + NAMES2)
    d = model.nd_load(str(f))
    assert list(d) == ["arg:fc_weigh", "aux:bn_mea"]
    np.testing.assert_array_equal(d["arg:fc_weigh"], np.arange(1, 7, dtype=np.float32).reshape(2, 3))
    np.testing.assert_array_equal(d["aux:bn_mea"], np.array([-1, 0.5, 7], np.float32))
    _, args, auxs = model.load_checkpoint(str(tmp_path / "a"), 1)
    assert list(args) == ["fc_weigh"] and list(auxs) == ["bn_mea"]


def test_reader_v1_unnamed_and_errors(tmp_path):
    f = tmp_path / "b.params"
    f.write_bytes(HEAD2 + REC_V1 + REC_V2 + _hex("0000000000000000"))
    lst = model.nd_load(str(f))
    assert isinstance(lst, list) and lst[0].dtype == np.int32 and lst[0].tolist() == [9, -9]
    f.write_bytes(HEAD2 + REC_V2)                                   # second array missing
    with pytest.raises(model.ParamsFormatError, match="truncated"):
        model.nd_load(str(f))
    f.write_bytes(_hex("1301000000000000", "0000000000000000", "0000000000000000", "0000000000000000"))
    with pytest.raises(model.ParamsFormatError, match="not an NDArray list"):
        model.nd_load(str(f))
    sparse = _hex("c9fa93f9", "01000000")
    f.write_bytes(_hex("1201000000000000", "0000000000000000", "0100000000000000") + sparse)
    with pytest.raises(model.ParamsFormatError, match="sparse"):
        model.nd_load(str(f))


def test_writer_emits_v2_bytes(tmp_path):
    f = tmp_path / "c.params"
    model.nd_save(str(f), {"arg:fc_weigh": np.arange(1, 7, dtype=np.float32).reshape(2, 3)})
    want = _hex("1201000000000000", "0000000000000000", "0100000000000000") + REC_V2 + \
        _hex("0100000000000000", "0c00000000000000") + b"arg:fc_weigh"
    assert f.read_bytes() == want
    g = np.random.Generator(np.random.PCG64(0))
    blob = {"arg:w%d" % i: g.standard_normal(s).astype(t) for i, (s, t) in
            enumerate([((4, 3, 3, 3), np.float32), ((7,), np.float64), ((2, 5), np.float16)])}
    blob["aux:i"] = np.arange(5, dtype=np.int64)
    model.nd_save(str(f), blob)
    back = model.nd_load(str(f))
    assert list(back) == list(blob)
    for k in blob:
        assert back[k].dtype == blob[k].dtype
        np.testing.assert_array_equal(back[k], blob[k])


def test_upsample_filt_and_name_rules():
    np.testing.assert_allclose(multi_init.upsample_filt(4), np.outer([.25, .75, .75, .25], [.25, .75, .75, .25]))
    np.testing.assert_allclose(multi_init.upsample_filt(3), np.outer([.5, 1, .5], [.5, 1, .5]))
    added = ["score_weight", "score2_pool4_bn_beta", "score3_conv_weight", "score4_conv_weight", "res5_reduced_bn_beta",
             "res4_reduced2_weight", "res3_bn_gamma", "_plus12_cls_pred_conv_bias", "_plus6_loc_pred_conv_weight",
             "multi_feat_2_conv_1x1_conv_weight", "multi_feat_5_conv_3x3_relu_loc_pred_conv_bias", "bigscore_weight"]
    kept = ["conv0_weight", "bn_data_beta", "bn0_gamma", "stage1_unit1_conv1_weight", "stage4_unit3_bn3_beta", "bn1_gamma",
            "fc1_weight", "stage3_unit1_sc_weight"]
    assert all(multi_init.is_added_layer(n) for n in added)
    assert not any(multi_init.is_added_layer(n) for n in kept)


@pytest.mark.gpu
def test_pretrained_ingestion_and_checkpoint_round_trip(gpu_device, tmp_path):
    import torch
    from dspnet_amd import synthetic
    from dspnet_amd.symbol.multitask_symbol_factory import get_multi_symbol_train
    B, S = 1, 128
    net = get_multi_symbol_train("resnet-50", S, num_classes=8, batch_size=B, device=gpu_device, seed=1)
    g = net.g
    logical = g.get_params()
    # a stand-in for the model-zoo ResNet-50: every backbone array of the graph with fresh values, plus what such a
    # file also carries and the graph has no use for (classifier, fix_gamma gamma, moving statistics)
    rng = np.random.Generator(np.random.PCG64(42))
    zoo = {}
    for p in g.param_order:
        if not multi_init.is_added_layer(p.name) and p.name != "affine_matrix":     # (a model-zoo file has no such array)
            zoo["arg:" + p.name] = rng.standard_normal(logical[p.name].shape).astype(np.float32) * 0.05
    assert "arg:conv0_weight" in zoo and zoo["arg:conv0_weight"].shape == (64, 3, 7, 7)
    assert zoo["arg:bn_data_beta"].shape == (3,)
    zoo["arg:fc1_weight"] = rng.standard_normal((1000, 2048)).astype(np.float32)
    zoo["arg:bn_data_gamma"] = np.ones(3, np.float32)
    zoo["aux:bn0_moving_mean"] = rng.standard_normal(64).astype(np.float32)
    model.nd_save(str(tmp_path / "resnet-50-0000.params"), zoo)

    _, args, auxs = model.load_checkpoint(str(tmp_path / "resnet-50"), 0)
    args, auxs = multi_init.init_from_resnet(net, args, auxs, seed=3)
    assert args["affine_matrix"].tolist() == [[1, 0, 0, 0, 1, 0]]
    g.set_params(args)
    now = g.get_params()
    n_added = 0
    for p in g.param_order:
        v = now[p.name]
        if p.name == "affine_matrix":                    # multi_init.py:72, in the checkpoint's (1, 6) shape
            assert v.tolist() == [[1, 0, 0, 0, 1, 0]]
            continue
        if not multi_init.is_added_layer(p.name):
            np.testing.assert_array_equal(v, zoo["arg:" + p.name])
            continue
        n_added += 1
        if p.kind == "deconv":
            assert p.name == "score4_conv_weight" and v.shape == (19, 19, 4, 4)
            for i in range(19):
                np.testing.assert_allclose(v[i, i], multi_init.upsample_filt(4), rtol=1e-7)
            assert v[0, 1].max() == 0
        elif p.name.endswith("_weight"):
            lim = 1.0 / np.sqrt(max(v.shape))
            assert np.abs(v).max() <= lim and np.abs(v).max() > 0.5 * lim and abs(float(v.mean())) < 0.1 * lim
        elif p.name.endswith("_gamma"):
            assert (v == 1).all()
        else:
            assert (v == 0).all(), p.name
    assert n_added > 40
    # pad lanes of the device layout stay zero (conv0 reads a 3 -> 4 channel input)
    assert float(g.params["conv0_weight"].data[..., 3].abs().max()) == 0.0
    # a wrong shape and a missing argument are errors, as in Module.set_params / executor bind
    bad = dict(args); bad["conv0_weight"] = np.zeros((64, 3, 3, 3), np.float32)
    with pytest.raises(ValueError, match="conv0_weight"):
        g.set_params(bad)
    short = {k: v for k, v in args.items() if k != "stage1_unit1_conv1_weight"}
    with pytest.raises(KeyError, match="stage1_unit1_conv1_weight"):
        g.set_params(short)
    with pytest.raises(KeyError, match="neither"):
        multi_init.init_from_resnet(net, {k: v for k, v in args.items() if not k.startswith("stage2_unit1_conv1")}, auxs)

    # the ingested network runs; checkpoint round trip through the reference's file format restores every bit
    gen = synthetic.rng(233)
    net.data.data.copy_(torch.from_numpy(synthetic.images(B, S, S, gen)).to(gpu_device))
    net.label_det.data.copy_(torch.from_numpy(synthetic.det_labels(B, gen=gen, height=S, width=S)).to(gpu_device))
    net.label_seg.data.copy_(torch.from_numpy(synthetic.seg_labels(B, S, S, gen=gen)).to(gpu_device))
    g.forward()
    out1 = [o.clone() for o in net.outputs()]
    assert all(bool(torch.isfinite(o).all()) for o in out1)
    arena1 = g.arena.clone()
    model.save_checkpoint(str(tmp_path / "dspnet"), 7, net, aux_params=auxs)
    assert os.path.exists(str(tmp_path / "dspnet-0007.params"))
    saved = model.nd_load(str(tmp_path / "dspnet-0007.params"))
    assert saved["arg:bn_data_gamma"].tolist() == [1, 1, 1] and saved["arg:conv0_weight"].shape == (64, 3, 7, 7)
    np.testing.assert_array_equal(saved["aux:bn0_moving_mean"], zoo["aux:bn0_moving_mean"])
    assert saved["aux:bn0_moving_var"].tolist() == [1.0] * 64
    g.arena.zero_()
    _, args2, _ = model.load_checkpoint(str(tmp_path / "dspnet"), 7)
    g.set_params(args2)
    # vectors narrower than their device buffer keep pad lanes: compare through the logical view and the outputs
    for k, v in g.get_params().items():
        np.testing.assert_array_equal(v, now[k])
    g.forward()
    for a, b in zip(out1, net.outputs()):
        assert torch.equal(a, b)
    assert arena1.shape == g.arena.shape


def test_symbol_factory_exposes_the_six_builders():
    """symbol/multitask_symbol_factory.py:104-224 and the callers' imports (multi_eval.py, train_multi.py)"""
    from dspnet_amd.symbol import multitask_symbol_builder as b, multitask_symbol_factory as f
    for name in ("get_det_symbol_train", "get_det_symbol", "get_seg_symbol_train", "get_seg_symbol",
                 "get_multi_symbol_train", "get_multi_symbol"):
        assert callable(getattr(f, name)) and callable(getattr(b, name)), name
    cfg = f.get_config("resnet-50", 512)
    assert cfg["network"] == "resnet" and cfg["from_layers"][2] == "_plus15"      # what the seg-only graphs read
