#!/usr/bin/env python3
"""Golden vectors for SURVEY 8f rank 1 - Taco2DecoderCell (r = 2) and MelPostnet2 - produced
by running the REFERENCE (kgoba/torch-tts at /root/reference), reduced dims, config-sandra-style.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_taco2.py

Writes tests/golden/taco2_model.npz, taco2_cases.npz, taco2_meta.json (data only)."""
import copy
import json
import os
import sys

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True
import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "tacotron"))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import tacotron as ref_tacotron  # noqa: E402
from make_golden import perturb, replay_masks, maxerr  # noqa: E402
from oracle import tacotron_oracle as O  # noqa: E402

torch.set_num_threads(4)


def main():
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/config-sandra.yaml")))
    cfg = copy.deepcopy(cfg)
    cfg["audio"]["num_mels"] = 24
    cfg["model"]["encoder"].update(dim_emb=24, dim_out=40)
    cfg["model"]["decoder"].update(r=2, dim_pre=32, dim_att=64, dim_rnn=[72, 88])
    cfg["model"]["postnet"] = {"dim_hidden": 64, "num_layers": 2}  # no "type" -> MelPostnet2 (tacotron.py:201-212)
    assert cfg["model"]["decoder"]["type"] == "tacotron2"
    torch.manual_seed(42)
    model = ref_tacotron.build_tacotron(cfg).eval()
    assert type(model.decoder.decoder_cell).__name__ == "Taco2DecoderCell" and type(model.postnet).__name__ == "MelPostnet2"
    perturb(model, 9)
    # MelPostnet2's BN stats / affine and Conv1dFix weights: make them non-trivial too
    g0 = torch.Generator().manual_seed(10)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.startswith("postnet.") and (k.endswith("running_mean") or k.endswith(".bias")):
            v.add_(0.1 * torch.randn(v.shape, generator=g0))
        if k.startswith("postnet.") and k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g0))
    model.load_state_dict(sd)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dec_w = {k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}
    post_w = {k[len("postnet."):]: v for k, v in sd.items() if k.startswith("postnet.") and not k.endswith("num_batches_tracked")}
    dims = O.DecoderDims(d_mel=24, r=2, d_pre=32, d_ctx=40, h_att=72, h_dec=88)
    d_pre_hidden = dec_w["decoder_cell.pre_net.layers.0.weight"].shape[0]
    assert d_pre_hidden == 128

    B, L = 3, 13
    lengths = torch.tensor([13, 8, 4])
    g = torch.Generator().manual_seed(77)
    ids = torch.randint(1, 30, (B, L), generator=g)
    for b in range(B):
        ids[b, lengths[b]:] = 0
    with torch.no_grad():
        memory = model.encoder(ids, lengths)
    mmask = ref_tacotron.lengths_to_mask(lengths)
    cases, errs = {}, {}

    def masks_for(seed, T):
        torch.manual_seed(seed)
        ms = []
        for _ in range(T):
            m0 = torch.empty(B, 128).bernoulli_(0.5).to(torch.uint8)
            m1 = torch.empty(B, 32).bernoulli_(0.5).to(torch.uint8)
            ms.append((m0, m1))
        return ms

    # inference, max_steps = 8 -> 9 steps, 18 frames
    with torch.no_grad():
        torch.manual_seed(5)
        y, s, w = model.decoder(memory, mmask, None, 8, p_no_forcing=0.1)
        y_post = model.postnet(y)
    T = w.shape[1]
    assert T == 9 and y.shape[1] == 18
    ms = masks_for(5, T)
    # layer-0 masks are [B,128], layer-1 [B,32]: store separately
    m0 = torch.stack([a for a, _ in ms]); m1 = torch.stack([b for _, b in ms])

    class M:  # masks[step][layer]
        def __getitem__(self, t):
            return [m0[t], m1[t]]
    oy, os_, ow = O.taco2_decode(dec_w, dims, memory, max_steps=8, masks=M())
    errs["infer"] = {"y": maxerr(oy, y), "s": maxerr(os_, s), "w": maxerr(ow, w)}
    opost = O.mel_postnet2(oy, post_w, 2)
    errs["postnet2"] = maxerr(opost, y_post)
    cases.update({"infer/m0": m0, "infer/m1": m1, "infer/y": y, "infer/s": s, "infer/w": w, "infer/y_post": y_post})

    # teacher forcing, 11 teacher frames -> 5 steps
    gx = torch.Generator().manual_seed(3)
    x = torch.randn(B, 11, 24, generator=gx) * 0.5
    with torch.no_grad():
        torch.manual_seed(6)
        y2, s2, w2 = model.decoder(memory, mmask, x, 0, p_no_forcing=None)
    assert w2.shape[1] == 5
    ms2 = masks_for(6, 5)
    t0 = torch.stack([a for a, _ in ms2]); t1 = torch.stack([b for _, b in ms2])

    class M2:
        def __getitem__(self, t):
            return [t0[t], t1[t]]
    oy2, os2, ow2 = O.taco2_decode(dec_w, dims, memory, masks=M2(), x=x)
    errs["teacher"] = {"y": maxerr(oy2, y2), "w": maxerr(ow2, w2)}
    cases.update({"teacher/x": x, "teacher/m0": t0, "teacher/m1": t1, "teacher/y": y2, "teacher/s": s2, "teacher/w": w2})

    # MelPostnet2 unit on a longer sequence
    yp = torch.randn(2, 29, 24, generator=gx)
    with torch.no_grad():
        ypo = model.postnet(yp)
    errs["postnet2_unit"] = maxerr(O.mel_postnet2(yp, post_w, 2), ypo)
    cases.update({"unit/post_y": yp, "unit/post_out": ypo})

    meta = {"dims": {"d_mel": 24, "r": 2, "d_pre": 32, "d_pre_hidden": 128, "d_ctx": 40, "h_att": 72, "h_dec": 88,
                     "postnet_hidden": 64, "postnet_layers": 2, "B": B, "L": L, "lengths": lengths.tolist()},
            "oracle_vs_reference_maxabs": errs, "seeds": {"infer": 5, "teacher": 6}}
    npz = {"memory": memory.numpy(), "lengths": lengths.numpy()}
    for k, v in dec_w.items():
        npz["dec/" + k] = v.numpy()
    for k, v in post_w.items():
        npz["post/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "taco2_model.npz"), **npz)
    np.savez_compressed(os.path.join(HERE, "taco2_cases.npz"), **{k: v.detach().numpy() for k, v in cases.items()})
    json.dump(meta, open(os.path.join(HERE, "taco2_meta.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(errs, indent=1))
    for fn in ("taco2_model.npz", "taco2_cases.npz"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)))


if __name__ == "__main__":
    main()
