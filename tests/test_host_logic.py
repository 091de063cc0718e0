"""CPU tests of the host side: the C-ABI library loads and exports every symbol the
header declares (no compute without a GPU), host-only queries, the reference-compatible
RNG replay, the module API surface / state-dict keys, and the loud failure on CPU tensors."""
import os
import re

import pytest
import torch

import torch_tts_amd as T
from oracle import tacotron_oracle as O
from torch_tts_amd import _lib
from torch_tts_amd.rng import MaskStream

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ttsdec.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(tts(?:dec|enc|vits)_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    lib = _lib.load()
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert lib.ttsdec_version() == _lib.ABI_VERSION == 2
    assert lib.ttsdec_strerror(0) == b"ok"
    assert b"multiples of 4" in lib.ttsdec_strerror(_lib.ERR_DIMS)


def test_host_only_queries_and_layout_sizes():
    e = T.Engine(T.EngineDims(postnet_layers=3), None)
    assert e.num_weight_tensors() == 21 + 5 * 3 + 1
    # 18 577 489 decoder params + 4 initial-state vectors + postnet, padded per tensor to 256 B
    n_dec = 86528 + 7348224 + 524288 + 10493952 + 124497
    assert n_dec == 18577489
    # + the split-fp16 planes (hi, lo) of the four LSTM weight matrices, row-major and in the chunked layout:
    # another 2 x 4 bytes per LSTM weight
    n_lstm_w = 2 * (4096 * (768 + 1024) + 4096 * (1536 + 1024))
    assert e.packed_bytes() >= 4 * (n_dec + n_lstm_w) and e.packed_bytes() % 256 == 0
    # + postnet fp32 (2 870 272 params incl. BN folded to alpha/beta) and its bf16 / fp16-hi / fp16-lo planes
    n_post_w = 512 * 80 * 5 + 2 * 512 * 512 * 5 + 80 * 512
    # + the planes of the query and mel/stop projection weights
    n_small_w = 512 * 1024 + 81 * 1536
    assert e.packed_bytes() < 4 * (n_dec + n_lstm_w + n_small_w + 2870272) * 1.01 + 6 * n_post_w + 128 * 256
    w1, w2 = e.workspace_bytes(64, 120), e.workspace_bytes(256, 120)
    assert 0 < w1 < w2 and w2 % 256 == 0
    assert e.postnet_workspace_bytes(256, 600) == 2 * 256 * 600 * 512 * 4 + 256 * 600 * 80 * 4
    assert T.Engine(T.EngineDims(), None).num_weight_tensors() == 21
    # precision switch: split-fp16 needs 8-aligned feature dims, otherwise the handle stays on fp32
    assert e.precision() == "f32"
    e.set_precision("split_f16")
    assert e.precision() == "split_f16"
    e2 = T.Engine(T.EngineDims(d_pre=36, d_ctx=40, h_att=72, h_dec=88), None)
    e2.set_precision("split_f16")
    assert e2.precision() == "f32"
    e.close()


def test_bad_dims_and_unbound_handle_are_errors_not_crashes():
    with pytest.raises(_lib.TtsdecError) as ei:
        T.Engine(T.EngineDims(d_pre=255), None)
    assert ei.value.code == _lib.ERR_DIMS
    with pytest.raises(_lib.TtsdecError):
        T.Engine(T.EngineDims(postnet_layers=9), None)
    e = T.Engine(T.EngineDims(), None)
    lib = _lib.load()
    dummy = 256  # never dereferenced: the call must stop at the unbound-weights check
    rc = lib.ttsdec_decode(e._h, dummy, 2, 3, 0, 1, 1, -2.0, 1, 0, None, 0, None, 0, None, dummy, dummy, dummy, None, dummy, 1 << 30, None)
    assert rc == _lib.ERR_NOT_BOUND
    assert lib.ttsdec_decode(e._h, None, 2, 3, 0, 1, 1, -2.0, 1, 0, None, 0, None, 0, None, dummy, dummy, dummy, None, dummy, 0, None) == _lib.ERR_INVALID_ARG
    assert lib.ttsdec_postnet(e._h, dummy, 1, 1, 0, dummy, dummy, 0, None) == _lib.ERR_DIMS  # no postnet in these dims
    e.close()


def test_modules_refuse_cpu_tensors_loudly():
    dims = O.DecoderDims(d_mel=8, d_pre=16, d_ctx=32, h_att=32, h_dec=32)
    cell = T.Taco2ProdDecoderCell(dims.d_ctx, dims.d_mel, 1, [dims.h_att, dims.h_dec], dim_pre=dims.d_pre)
    dec = T.Decoder(cell, 1, dims.d_mel).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(torch.zeros(2, 5, 32), None, None, 3)
    with pytest.raises(RuntimeError, match="HIP path only"):
        T.MelPostnet(8, 16, 5, 2).eval()(torch.zeros(1, 4, 8))
    # the sub-modules run as torch ops when called on their own - on a ROCm device only, like everything else
    with pytest.raises(RuntimeError, match="no CPU path"):
        cell.pre_net(torch.zeros(1, 8))
    # training mode / grad-enabled calls take the torch-op device path: CPU tensors are refused there too
    with pytest.raises(RuntimeError, match="no CPU"):
        dec.train()(torch.zeros(2, 5, 32), None, torch.zeros(2, 4, 8))


def test_state_dict_keys_match_the_reference(golden):
    d = golden["meta"]["small_dims"]
    cell = T.Taco2ProdDecoderCell(d["d_ctx"], d["d_mel"], d["r"], [d["h_att"], d["h_dec"]], dim_pre=d["d_pre"], dim_att=d["h_att"])
    dec = T.Decoder(cell, d["r"], d["d_mel"])
    assert set(dec.state_dict().keys()) == set(golden["dec"].keys())
    for k, v in dec.state_dict().items():
        assert tuple(v.shape) == tuple(golden["dec"][k].shape), k
    dec.load_state_dict(golden["dec"], strict=True)
    pn = T.MelPostnet(d["d_mel"], d["postnet_hidden"], 5, d["postnet_layers"])
    keys = {k for k in pn.state_dict().keys() if not k.endswith("num_batches_tracked")}
    assert keys == set(golden["post"].keys())
    # attribute surface the callers read (decoder.py:6-14, decoder_cell.py:150)
    for attr in ("decoder_cell", "r", "dim_mel", "stop_threshold", "fc_mel", "fc_stop"):
        assert hasattr(dec, attr)
    assert cell.dim_output == d["h_dec"] + d["d_ctx"]
    w0, ctx0, hc = cell.initial_state(3, 7, torch.float32, "cpu")
    assert w0.shape == (3, 7) and float(w0[:, 0].min()) == 1.0 and float(w0[:, 1:].abs().max()) == 0.0
    assert ctx0.shape == (3, d["d_ctx"]) and hc[0][0].shape == (3, d["h_att"]) and hc[1][1].shape == (3, d["h_dec"])


def test_taco2_and_postnet2_state_dict_keys_match_the_reference(golden_taco2):
    d = golden_taco2["meta"]["dims"]
    cell = T.Taco2DecoderCell(d["d_ctx"], d["d_mel"], d["r"], [d["h_att"], d["h_dec"]], dim_pre=d["d_pre"])
    dec = T.Decoder(cell, d["r"], d["d_mel"])
    assert set(dec.state_dict().keys()) == set(golden_taco2["dec"].keys())
    dec.load_state_dict(golden_taco2["dec"], strict=True)
    assert dec.fc_mel.out_features == d["r"] * d["d_mel"] and dec.fc_stop.out_features == d["r"]
    pn = T.MelPostnet2(d["d_mel"], d["postnet_hidden"], d["postnet_layers"])
    keys = {k for k in pn.state_dict().keys() if not k.endswith("num_batches_tracked")}
    assert keys == set(golden_taco2["post"].keys())
    ts = pn.weight_tensors()
    assert len(ts) == 21 + 11 * d["postnet_layers"] and tuple(ts[21].shape) == (d["postnet_hidden"], d["d_mel"], 5)
    e = T.Engine(pn.engine_dims(), None)
    assert e.num_weight_tensors() == len(ts)
    e.close()


def test_encoder_module_keys_and_weight_order():
    import json

    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "encoder_meta.json")))
    enc = T.Encoder2(meta["dims"]["alphabet"], dim_out=meta["dims"]["d_out"], dim_emb=meta["dims"]["d_emb"])
    keys = {k for k in enc.state_dict().keys() if not k.endswith("num_batches_tracked")}
    assert keys == set(meta["keys"])
    ts = enc.weight_tensors()
    assert len(ts) == _lib.ENC_W_COUNT == 20
    E, H = meta["dims"]["d_emb"], meta["dims"]["d_out"] // 2
    assert tuple(ts[0].shape) == (meta["dims"]["alphabet"], E) and tuple(ts[14].shape) == (4 * H, 2 * E) and tuple(ts[17].shape) == (4 * H, H)
    # CPU tensors / training keep the stock PyTorch path (autograd); padded rows come out exactly zero
    ids = torch.tensor([[3, 1, 2, 5], [4, 2, 0, 0]])
    out = enc.eval()(ids, torch.tensor([4, 2]))
    assert out.shape == (2, 4, meta["dims"]["d_out"]) and float(out[1, 2:].abs().max()) == 0.0


def test_weight_tensor_order_matches_header_enum(golden):
    d = golden["meta"]["small_dims"]
    cell = T.Taco2ProdDecoderCell(d["d_ctx"], d["d_mel"], d["r"], [d["h_att"], d["h_dec"]], dim_pre=d["d_pre"])
    dec = T.Decoder(cell, d["r"], d["d_mel"])
    ts = dec.weight_tensors()
    assert len(ts) == _lib.W_DECODER_COUNT
    shapes = [tuple(t.shape) for t in ts]
    P, D, Ha, Hd, M = d["d_pre"], d["d_ctx"], d["h_att"], d["h_dec"], d["d_mel"]
    assert shapes == [
        (P, M), (P,), (P, P), (P,), (D, Ha),
        (4 * Ha, P + D), (4 * Ha, Ha), (4 * Ha,), (4 * Ha,),
        (4 * Hd, Ha + D), (4 * Hd, Hd), (4 * Hd,), (4 * Hd,),
        (1, Ha), (1, Hd), (1, Ha), (1, Hd),
        (M, Hd + D), (M,), (1, Hd + D), (1,),
    ]
    pn = T.MelPostnet(M, 64, 5, 3)
    pts = pn.weight_tensors()
    assert len(pts) == 21 + 16 and all(t is None for t in pts[:21]) and tuple(pts[21].shape) == (64, M, 5)


def test_mask_stream_replays_reference_draws_and_rewinds():
    B, P = 3, 36
    torch.manual_seed(7)
    ref = [torch.stack(O.draw_prenet_masks(B, P, P)) for _ in range(6)]
    ref_next = torch.rand(1)
    torch.manual_seed(7)
    ms = MaskStream(B, P)
    m1, f1 = ms.draw(4)
    m2, _ = ms.draw(2)
    assert torch.equal(torch.cat([m1, m2]), torch.stack(ref)) and bool(f1.all())
    assert torch.equal(torch.rand(1), ref_next)
    # early stop after 3 of 10 drawn steps: generator must be where the reference leaves it
    torch.manual_seed(7)
    ms = MaskStream(B, P)
    ms.draw(10)
    ms.rewind_to(3)
    torch.manual_seed(7)
    for _ in range(3):
        O.draw_prenet_masks(B, P, P)
    expect = torch.rand(1)
    torch.manual_seed(7)
    ms = MaskStream(B, P)
    ms.draw(10)
    ms.rewind_to(3)
    assert torch.equal(torch.rand(1), expect)


def test_mask_stream_teacher_flags_follow_decoder_py_65(golden):
    c, m = golden["cases"], golden["meta"]["teacher_p"]
    torch.manual_seed(m["seed"])
    ms = MaskStream(3, 36, 0.5, p_no_forcing=m["p_no_forcing"], teacher_steps=m["Tx"])
    masks, flags = ms.draw(m["Tx"])
    assert torch.equal(masks, c["teacher_p/masks"])
    assert torch.equal(flags[:-1], c["teacher_p/flags"]) and int(flags[-1]) == 1


def test_philox_restatement_is_balanced_and_keyed():
    a = O.philox_keep_masks(1, 0, 64, 256)
    b = O.philox_keep_masks(1, 1, 64, 256)
    c = O.philox_keep_masks(2, 0, 64, 256)
    assert a.shape == (2, 64, 256)
    assert 0.45 < float(a.float().mean()) < 0.55
    assert not torch.equal(a, b) and not torch.equal(a, c) and not torch.equal(a[0], a[1])
    assert torch.equal(a, O.philox_keep_masks(1, 0, 64, 256))


def test_lengths_to_mask_and_build_tacotron_surface():
    m = T.lengths_to_mask(torch.tensor([3, 1, 2]))
    assert m.tolist() == [[True, True, True], [True, False, False], [True, True, False]]
    cfg = {
        "text": {"alphabet": "abc"},
        "audio": {"num_mels": 8},
        "model": {
            "encoder": {"dim_emb": 8, "dim_out": 16},
            "decoder": {"type": "tacotron2prod", "r": 1, "dim_pre": 8, "dim_att": 16, "dim_rnn": [16, 12]},
            "postnet": {"type": "tacotron2", "dim_hidden": 16, "num_layers": 2},
        },
    }
    model = T.build_tacotron(cfg)
    assert isinstance(model.decoder, T.Decoder) and isinstance(model.postnet, T.MelPostnet)
    assert model.decoder.decoder_cell.dim_output == 12 + 16
    # encoder is stock PyTorch and runs on CPU (outside the hot path); padded rows are exactly zero
    ids = torch.tensor([[1, 2, 3, 1], [2, 1, 0, 0]])
    mem = model.eval().encoder(ids, torch.tensor([4, 2]))
    assert mem.shape == (2, 4, 16) and float(mem[1, 2:].abs().max()) == 0.0
    import copy
    import pickle

    model2 = pickle.loads(pickle.dumps(model))  # nn.DataParallel / checkpointing need this
    assert set(model2.state_dict().keys()) == set(model.state_dict().keys())
    copy.deepcopy(model)
    cfg["model"]["decoder"]["type"] = "tacotron2"  # rdh / sandra / template configs
    m2 = T.build_tacotron(cfg)
    assert isinstance(m2.decoder.decoder_cell, T.Taco2DecoderCell)
    assert m2.decoder.decoder_cell.dim_output == 16 + 12 + 16 and m2.decoder.fc_mel.in_features == 44
    assert m2.decoder.decoder_cell.pre_net.layers[0].out_features == 128
    w0, hc = m2.decoder.decoder_cell.initial_state(2, 5, torch.float32, "cpu")
    assert w0.shape == (2, 5) and len(hc) == 2 and hc[1][0].shape == (2, 12)
    cfg["model"]["decoder"]["type"] = "tacotron1"  # dead code in the reference
    with pytest.raises(NotImplementedError):
        T.build_tacotron(cfg)


@pytest.mark.skipif(not os.path.isdir("/root/reference/tacotron"), reason="reference checkout not present (GPU box)")
def test_integration_recipe_drops_in_under_the_reference_tacotron_py():
    """alias the three hot-path classes in the reference's own modules and
    let the reference's build_tacotron assemble the model - run in a subprocess so the reference's
    top-level module names (decoder, modules, ...) do not leak into this test session."""
    import subprocess
    import sys

    code = r'''
import sys, yaml
sys.dont_write_bytecode = True
sys.path.insert(0, %r)
import torch_tts_amd as T
sys.path.insert(0, "/root/reference/tacotron")
import decoder_cell, decoder
import modules.modules as ref_modules
ref_keys = None
import tacotron as ref_tacotron
cfg = yaml.safe_load(open("/root/reference/configs/config-ljspeech.yaml"))
ref_keys = set(ref_tacotron.build_tacotron(cfg).state_dict().keys())
ref_tacotron.Taco2ProdDecoderCell = decoder_cell.Taco2ProdDecoderCell = T.Taco2ProdDecoderCell
ref_tacotron.Decoder = decoder.Decoder = T.Decoder
ref_tacotron.MelPostnet = ref_modules.MelPostnet = T.MelPostnet
model = ref_tacotron.build_tacotron(cfg)
assert isinstance(model.decoder, T.Decoder) and isinstance(model.decoder.decoder_cell, T.Taco2ProdDecoderCell)
assert isinstance(model.postnet, T.MelPostnet)
assert set(model.state_dict().keys()) == ref_keys, set(model.state_dict().keys()) ^ ref_keys
print("OK", len(ref_keys))
''' % ROOT
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr


def test_vits2_modules_keys_tensor_counts_and_refusals():
    import json
    import warnings

    import numpy as np

    warnings.filterwarnings("ignore", category=FutureWarning)
    z = np.load(os.path.join(ROOT, "tests", "golden", "vits2_small.npz"))
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "vits2_meta.json")))["dims"]
    ref_keys = {k[2:] for k in z.files if k.startswith("w/")}
    te = T.vits2.TextEncoder(d["n_vocab"], d["inter_channels"], d["hidden_channels"], d["filter_channels"], d["n_heads"], d["n_layers"], d["kernel_size"], 0.1)
    assert {"enc_p." + k for k in te.state_dict()} == {k for k in ref_keys if k.startswith("enc_p.")}
    fl = T.vits2.ResidualCouplingTransformersBlock(d["inter_channels"], d["flow_hidden"], d["flow_kernel"], 1, d["flow_wn_layers"], n_flows=d["n_flows"],
                                                   use_transformer_flows=True)
    mine = {"flow." + k for k in fl.state_dict() if "post_transformer" not in k}
    assert mine == {k for k in ref_keys if k.startswith("flow.")}
    # speaker-conditioned variants (gin_channels > 0): the reference's extra keys, and two more tensors per owner in the C ABI's list
    dg = json.load(open(os.path.join(ROOT, "tests", "golden", "vits2_meta.json")))["dims_g"]
    ref_keys_g = {k[3:] for k in z.files if k.startswith("wg/")}
    te_g = T.vits2.TextEncoder(dg["n_vocab"], dg["inter_channels"], dg["hidden_channels"], dg["filter_channels"], dg["n_heads"], dg["n_layers"],
                               dg["kernel_size"], 0.1, gin_channels=dg["gin_channels"])
    assert {"enc_p." + k for k in te_g.state_dict()} == {k for k in ref_keys_g if k.startswith("enc_p.")}
    fl_g = T.vits2.ResidualCouplingTransformersBlock(dg["inter_channels"], dg["flow_hidden"], dg["flow_kernel"], 1, dg["flow_wn_layers"],
                                                     n_flows=dg["n_flows"], gin_channels=dg["gin_channels"], use_transformer_flows=True)
    assert {"flow." + k for k in fl_g.state_dict() if "post_transformer" not in k} == {k for k in ref_keys_g if k.startswith("flow.")}
    assert te_g.encoder.cond_layer_idx == 2 and len(fl_g.flows[0].weight_tensors()) == len(fl.flows[0].weight_tensors()) + 2
    # tensor counts the C ABI expects (include/ttsdec.h): 1 + 18/layer + 2 ; per flow 16/tf-layer + 2 + 4/WN-layer + 2
    lib = _lib.load()
    import ctypes as C

    h = C.c_void_p()
    dims = _lib.VitsDims(*[int(d.get(n, 0)) for n, _ in _lib.VitsDims._fields_])
    assert lib.ttsvits_create(C.byref(dims), C.byref(h)) == _lib.OK
    n_text = 1 + 18 * d["n_layers"] + 2
    n_flow = d["n_flows"] * (16 * d["flow_tf_layers"] + 2 + 4 * d["flow_wn_layers"] + 2)
    assert lib.ttsvits_num_weight_tensors(h) == n_text + n_flow
    assert len([te.emb.weight] + te.encoder.weight_tensors() + [te.proj.weight, te.proj.bias]) == n_text
    assert sum(len(fl.flows[2 * i].weight_tensors()) for i in range(d["n_flows"])) == n_flow
    assert lib.ttsvits_packed_bytes(h) % 256 == 0 and lib.ttsvits_text_encoder_workspace_bytes(h, 2, 13) > 0
    assert lib.ttsvits_flow_reverse(h, 256, 256, None, 1, 4, 256, 256, 1 << 30, None) == _lib.ERR_NOT_BOUND
    lib.ttsvits_destroy(h)
    dims_g = _lib.VitsDims(*[int(dg.get(n, 0)) for n, _ in _lib.VitsDims._fields_])
    assert lib.ttsvits_create(C.byref(dims_g), C.byref(h)) == _lib.OK
    assert lib.ttsvits_num_weight_tensors(h) == (1 + 2 + 18 * dg["n_layers"] + 2) + dg["n_flows"] * (16 * dg["flow_tf_layers"] + 2 + 2 + 4 * dg["flow_wn_layers"] + 2)
    lib.ttsvits_destroy(h)
    dims_g.cond_layer_idx = dg["n_layers"]  # attentions.py:50-52
    assert lib.ttsvits_create(C.byref(dims_g), C.byref(h)) == _lib.ERR_DIMS
    bad = _lib.VitsDims(*[int(d.get(n, 0)) for n, _ in _lib.VitsDims._fields_])
    bad.inter_channels = 18  # half = 9 is not a multiple of 4
    assert lib.ttsvits_create(C.byref(bad), C.byref(h)) == _lib.ERR_DIMS
    with pytest.raises(RuntimeError):
        te.eval()(torch.zeros(1, 4, dtype=torch.long), torch.tensor([4]))  # CPU tensors: no fallback


def test_blob_key_sees_invalidate_but_not_data_edits():
    """engine.weights_fingerprint: `.data` edits leave Parameter._version alone (why invalidate() exists);
    invalidate() and load_state_dict change the key; modules with the hook still pickle / deepcopy."""
    import copy
    import pickle

    import torch

    import torch_tts_amd as T
    from torch_tts_amd.engine import weights_fingerprint

    cell = T.Taco2ProdDecoderCell(16, 8, 1, [16, 16], dim_pre=8, dim_att=16)
    dec = T.Decoder(cell, 1, 8)
    k0 = weights_fingerprint(dec.weight_tensors())
    dec.fc_mel.weight.data.mul_(2.0)
    assert weights_fingerprint(dec.weight_tensors()) == k0, "(documented blind spot: .data edits are invisible)"
    dec.invalidate()
    k1 = weights_fingerprint(dec.weight_tensors())
    assert k1 != k0
    dec.load_state_dict(dec.state_dict())
    assert weights_fingerprint(dec.weight_tensors()) != k1
    with torch.no_grad():
        dec.fc_mel.weight.mul_(2.0)  # a tracked in-place op is seen without help
    k2 = weights_fingerprint(dec.weight_tensors())
    assert k2 != weights_fingerprint([t.clone() for t in dec.weight_tensors()])
    for m in (pickle.loads(pickle.dumps(dec)), copy.deepcopy(dec)):
        assert set(m.state_dict()) == set(dec.state_dict())
        m.load_state_dict(dec.state_dict())


def test_options_api_and_env_presets_host_only(monkeypatch):
    """ttsdec_set_option / get_option / option_name and the TTSDEC_OPTIONS preset (include/ttsdec.h): host-only, no GPU."""
    ids = _lib.option_ids()
    assert {"graph", "overlap", "chunk_a", "chunk_b", "proj_regw", "head_proj", "query_role", "profile_ablation", "debug_flags", "spin_limit"} <= set(ids)
    assert _lib.load().ttsdec_option_name(len(ids)) is None and _lib.load().ttsdec_option_name(-1) is None
    e = T.Engine(T.EngineDims(postnet_layers=3), None)
    assert all(e.get_option(n) in (-1, 0) for n in ids)  # library defaults
    e.set_option("overlap", 1)
    e.set_option("spin_limit", 77)
    assert e.get_option("overlap") == 1 and e.get_option("spin_limit") == 77 and e.get_option("graph") == -1
    lib = _lib.load()
    import ctypes as C
    assert lib.ttsdec_set_option(e._h, 999, 1) == _lib.ERR_INVALID_ARG and lib.ttsdec_set_option(None, 0, 1) == _lib.ERR_INVALID_ARG
    assert lib.ttsdec_get_option(e._h, 0, None) == _lib.ERR_INVALID_ARG
    v = C.c_int(5)
    assert lib.ttsdec_get_option(e._h, -3, C.byref(v)) == _lib.ERR_INVALID_ARG
    # presets: well-formed items apply, junk is ignored (never a crash: the string comes from the environment)
    monkeypatch.setenv("TTSDEC_OPTIONS", "overlap=2,,graph=0,=5,nonsense,head_proj=,query_role=0=1,spin_limit=123456789012345,chunk_b=0,")
    e2 = T.Engine(T.EngineDims(postnet_layers=0), None)
    assert (e2.get_option("overlap"), e2.get_option("graph"), e2.get_option("chunk_b"), e2.get_option("chunk_a")) == (2, 0, 0, -1)
    assert e2.get_option("head_proj") == 0 and e2.get_option("query_role") == 0 and e2.get_option("spin_limit") == 0x7FFFFFFF
    monkeypatch.setenv("TTSDEC_OPTIONS", "x" * 5000 + ",overlap=1")
    assert T.Engine(T.EngineDims(), None).get_option("overlap") == 1


def test_workspace_and_blob_layouts_over_many_shapes_host_only():
    """The carving arithmetic of api.hip (run under ASan / UBSan by tools/build_asan.sh): sizes are positive, 256-byte granular,
    monotone in B and L, and zero / error for nonsense - for every cell and postnet type."""
    for cell, ph in ((_lib.CELL_TACO2PROD, 0), (_lib.CELL_TACO2, 128)):
        for pt, layers in ((_lib.POSTNET_TYPE_MEL, 5), (_lib.POSTNET_TYPE_MEL2, 2), (_lib.POSTNET_TYPE_MEL, 0)):
            d = T.EngineDims(r=2 if cell == _lib.CELL_TACO2 else 1, cell_type=cell, d_pre_hidden=ph, postnet_type=pt, postnet_layers=layers)
            e = T.Engine(d, None)
            assert e.packed_bytes() > 0 and e.packed_bytes() % 256 == 0 and e.num_weight_tensors() >= 21
            prev = 0
            for B in (1, 3, 31, 32, 33, 64, 65, 255, 256, 257, 2048, 8192):
                sizes = [e.workspace_bytes(B, L) for L in (1, 2, 120, 521)]
                assert all(s > 0 and s % 256 == 0 for s in sizes) and sizes == sorted(sizes), (B, sizes)
                assert sizes[2] >= prev
                prev = sizes[2]
                if layers:
                    assert e.postnet_workspace_bytes(B, 600) > e.postnet_workspace_bytes(B, 1) > 0
            assert e.workspace_bytes(0, 5) == 0 and e.workspace_bytes(5, 0) == 0 and e.workspace_bytes(-1, -1) == 0
            assert e.postnet_workspace_bytes(0, 1) == 0


def test_bench_byte_and_flop_model_follows_the_selected_config():
    """bench.py's algorithmic bytes / FLOPs per decode step: the LJSpeech cell reproduces SURVEY.md 8(d)'s figures; the other
    shipped configs get their own widths (d_ctx 256, PreNet hidden 128, r = 2, the Taco2 cell's query / projection inputs)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lj = bench.cell_model(bench.LJSPEECH)
    for B in (1, 64, 256):
        assert bench.step_bytes(B, 120, lj)["step"] == 74_309_956 + B * 285_220
    assert bench.step_flops(1, 120, lj) == 2 * (18_577_489 - 16_977) + 4 * 120 * 512  # 37.37 MFLOP per frame
    sd = bench.cell_model(bench.SANDRA)
    assert (sd["D"], sd["P0"], sd["r"], sd["Ha"], sd["Kq"], sd["Kproj"], sd["Nproj"]) == (256, 128, 2, 512, 1024, 1024, 162)
    b = bench.step_bytes(256, 120, sd)
    assert b["attention"] == 256 * (120 * 256 * 4 + 3 * 120 * 4 + 256 * 4)  # one pass over a 256-wide memory
    assert b["query"] == 256 * 1024 * 4 and b["proj"] == (162 * 1024 + 162) * 4 + 256 * (2 * 80 * 4 + 2 * 4)
    rd = bench.cell_model(bench.RDH)
    assert rd["taco2"] and rd["P0"] == 128 and rd["D"] == 512 and rd["Kq"] == 2048
    # MelPostnet of the LJSpeech / rdh configs: 3 x conv(k = 5) 80 -> 512 -> 512 -> 512 and the 512 -> 80 Linear, 2 FLOP per weight and frame
    assert bench.postnet_flops_per_frame(bench.CONFIGS["ljspeech"]) == 2 * (5 * 80 * 512 + 2 * 5 * 512 * 512 + 512 * 80)


def test_no_kernel_of_the_library_uses_scratch_memory():
    """The compiler's own resource report of the build (torch-tts_amd/lib/*.resources.txt, written by build.py): every kernel
    of every source file with ScratchSize 0 and no spilled vector registers.  (Round 4: six extra 64-bit base computations in a
    loader put the split-fp16 two-role kernels - already at the scalar-register limit - into scratch and cost the step 45 %;
    round 3 shipped a row GEMM with 14 spilled registers.)"""
    import glob

    from torch_tts_amd import _lib  # noqa: F401  (the build has run)

    files = sorted(glob.glob(os.path.join(ROOT, "torch-tts_amd", "lib", "*.resources.txt")))
    if not files:
        build = os.path.join(ROOT, "torch-tts_amd", "build.py")
        pytest.skip(f"no resource reports beside the objects (library built by something other than {build})")
    assert len(files) >= 6, files
    kernels = bad = 0
    for fn in files:
        name = None
        for line in open(fn):
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                name = m.group(1)
                kernels += 1
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
            if m and int(m.group(1)) != 0:
                bad += 1
                print("scratch:", os.path.basename(fn), name, m.group(1))
            m = re.search(r"VGPRs Spill: (\d+)", line)
            if m and int(m.group(1)) != 0:
                bad += 1
                print("spill:", os.path.basename(fn), name, m.group(1))
    assert kernels > 100 and bad == 0, (kernels, bad)


def test_exact_fp32_is_every_handles_default_arithmetic():
    """The reference computes in fp32: every handle of the C ABI and every drop-in module starts in exact fp32 and split-fp16 is
    an explicit opt-in (ttsdec_ / ttsenc_ / ttsvits_set_precision; Tacotron.fast_inference() sets all of a model's modules)."""
    import ctypes as C
    import json

    lib = _lib.load()
    h = C.c_void_p()
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "vits2_meta.json")))["dims"]
    dims = _lib.VitsDims(*[int(d.get(n, 0)) for n, _ in _lib.VitsDims._fields_])
    assert lib.ttsvits_create(C.byref(dims), C.byref(h)) == _lib.OK
    assert lib.ttsvits_get_precision(h) == _lib.PREC_F32
    assert lib.ttsvits_set_precision(h, _lib.PREC_SPLIT_F16) == _lib.OK and lib.ttsvits_get_precision(h) == _lib.PREC_SPLIT_F16
    assert lib.ttsvits_set_precision(h, 7) == _lib.ERR_INVALID_ARG
    lib.ttsvits_destroy(h)
    e = C.c_void_p()
    assert lib.ttsenc_create(C.byref(_lib.EncDims(40, 64, 64, 5, 1e-5)), C.byref(e)) == _lib.OK
    assert lib.ttsenc_get_precision(e) == _lib.PREC_F32
    assert lib.ttsenc_set_precision(e, _lib.PREC_SPLIT_F16) == _lib.OK and lib.ttsenc_get_precision(e) == _lib.PREC_SPLIT_F16
    assert lib.ttsenc_set_precision(e, -1) == _lib.ERR_INVALID_ARG and lib.ttsenc_set_precision(None, 0) == _lib.ERR_INVALID_ARG
    lib.ttsenc_destroy(e)
    model = T.build_tacotron({"text": {"alphabet": "abc"}, "audio": {"num_mels": 80}, "model": {
        "encoder": {"type": "tacotron2", "dim_emb": 64, "dim_out": 64}, "decoder": {"type": "tacotron2prod", "r": 1, "dim_pre": 128, "dim_att": 128, "dim_rnn": [128, 128]},
        "postnet": {"type": "tacotron2", "dim_hidden": 64, "num_layers": 3}}})
    assert (model.decoder.precision, model.postnet.precision, model.encoder.precision) == ("f32", "f32", "f32")
    model.fast_inference(3)
    assert (model.decoder.precision, model.postnet.precision, model.encoder.precision) == ("split_f16",) * 3
    model.reference_compatible()
    assert (model.decoder.precision, model.postnet.precision, model.encoder.precision) == ("f32", "f32", "f32")
    te = T.vits2.TextEncoder(20, 16, 16, 32, 2, 1, 3, 0.1)
    fl = T.vits2.ResidualCouplingTransformersBlock(16, 16, 5, 1, 2, use_transformer_flows=True)
    assert te.precision == "f32" and fl.precision == "f32"


def test_committed_traffic_captures_belong_to_these_kernel_sources():
    """bench.py prints roofline.traffic only from a PMC capture whose digest of the kernel sources equals the running sources'
    (profiles/r04_traffic_*.json, tools/summarize_pmc.py): a kernel edit without a new capture would silently turn the field
    into null on the driver's line.  This keeps the two together."""
    import importlib.util
    import json

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    dig = bench.sources_digest()
    files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("r04_traffic") and f.endswith(".json"))
    assert len(files) >= 3, files
    seen = set()
    for fn in files:
        tj = json.load(open(os.path.join(ROOT, "profiles", fn)))
        assert tj["sources_digest"] == dig, (fn, tj["sources_digest"], dig)
        seen.add((tj["precision"], tj["batch"]))
        for name, v in tj["per_launch"].items():
            assert v["hbm_bytes"] > 0, (fn, name)
    assert {("f32", 256), ("f32", 64), ("split_f16", 256)} <= seen, seen
