#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE itself (kgoba/torch-tts,
mounted read-only at /root/reference) on CPU.  Run in the build container only:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Outputs (committed; data only - inputs, weights at reduced dims, expected
outputs; no reference source):
    tests/golden/small_model.npz   reduced-dims weights (encoder, decoder, postnet) + ids + encoder memory
    tests/golden/small_cases.npz   decode / stop / teacher / postnet / unit vectors
    tests/golden/meta.json         sizes, seeds, oracle-vs-reference errors
                                   (incl. a full LJSpeech-dims check whose
                                   86 MB of weights are NOT committed)

The reference's Prenet dropout is always on (modules.py:40) and draws from the
default CPU generator; the masks are captured by replaying the same draws under
the same seed, and the replay is verified by the oracle reproducing the
reference's outputs.
"""
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

import tacotron as ref_tacotron  # noqa: E402  (the reference)
from modules.modules import PreNet as RefPreNet  # noqa: E402,F401
from oracle import tacotron_oracle as O  # noqa: E402

torch.set_num_threads(4)


def small_config():
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/config-ljspeech.yaml")))
    cfg = copy.deepcopy(cfg)
    cfg["audio"]["num_mels"] = 20
    cfg["model"]["encoder"]["dim_emb"] = 24
    cfg["model"]["encoder"]["dim_out"] = 40
    cfg["model"]["decoder"]["dim_pre"] = 36
    cfg["model"]["decoder"]["dim_att"] = 72
    cfg["model"]["decoder"]["dim_rnn"] = [72, 88]
    cfg["model"]["postnet"]["dim_hidden"] = 64
    cfg["model"]["postnet"]["num_layers"] = 3
    return cfg


def perturb(model, seed):
    """Make biases, initial states and BN statistics non-trivial (the
    reference's init zeroes them, which would hide indexing bugs)."""
    g = torch.Generator().manual_seed(seed)
    sd = model.state_dict()
    for k, v in sd.items():
        if k.startswith("encoder"):
            continue
        if k.endswith("num_batches_tracked"):
            continue
        if k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
        elif k.endswith("running_mean") or k.endswith(".bias") or "initial_decoder" in k or k.endswith("bias_ih") or k.endswith("bias_hh"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("1.weight") and "postnet.conv" in k:
            v.add_(0.1 * torch.randn(v.shape, generator=g))
    model.load_state_dict(sd)


def split_weights(model):
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    dec = {k[len("decoder.") :]: v for k, v in sd.items() if k.startswith("decoder.")}
    post = {k[len("postnet.") :]: v for k, v in sd.items() if k.startswith("postnet.") and not k.endswith("num_batches_tracked")}
    return dec, post


def replay_masks(seed, T, B, d_pre, teacher_p=None):
    """Replay the default-generator draws Decoder.forward makes: per step two
    Bernoulli(0.5) tensors [B, d_pre] (PreNet layers), then - teacher mode with
    p_no_forcing - one torch.rand(1) (decoder.py:65), except after the last step."""
    torch.manual_seed(seed)
    masks, flags = [], []
    for t in range(T):
        m0, m1 = O.draw_prenet_masks(B, d_pre, d_pre)
        masks.append(torch.stack([m0, m1]))
        if teacher_p is not None and t < T - 1:
            flags.append(bool(torch.rand(1) > teacher_p))
    return torch.stack(masks), flags


def maxerr(a, b):
    return float((a - b).abs().max())


def main():
    meta = {"reference": "kgoba/torch-tts @ 2024_10_08", "torch": torch.__version__}
    cfg = small_config()
    torch.manual_seed(42)
    model = ref_tacotron.build_tacotron(cfg).eval()
    perturb(model, 7)
    dec_w, post_w = split_weights(model)
    dims = O.DecoderDims(d_mel=20, r=1, d_pre=36, d_ctx=40, h_att=72, h_dec=88)
    n_layers = cfg["model"]["postnet"]["num_layers"]

    # ---- encoder memory with ragged lengths (padded rows are exactly zero) ----
    B, L = 3, 17
    lengths = torch.tensor([17, 11, 5])
    g = torch.Generator().manual_seed(1234)
    ids = torch.randint(1, 40, (B, L), generator=g)
    for b in range(B):
        ids[b, lengths[b] :] = 0
    with torch.no_grad():
        memory = model.encoder(ids, lengths)
    assert memory.shape == (B, L, 40)
    assert float(memory[1, 11:].abs().max()) == 0.0
    mmask = ref_tacotron.lengths_to_mask(lengths)

    cases = {}
    errs = {}

    # ---- case 1: inference, T = max_steps + 1 = 13 ----
    seed1, max_steps = 7, 12
    with torch.no_grad():
        torch.manual_seed(seed1)
        y, s, w = model.decoder(memory, mmask, None, max_steps, p_no_forcing=0.1)
        y_post = model.postnet(y)
    T = y.shape[1]
    assert T == max_steps + 1
    masks, _ = replay_masks(seed1, T, B, dims.d_pre)
    oy, os_, ow = O.decode(dec_w, dims, memory, max_steps=max_steps, masks=masks)
    opost = O.mel_postnet(oy, post_w, n_layers)
    errs["infer"] = {"y": maxerr(oy, y), "s": maxerr(os_, s), "w": maxerr(ow, w), "y_post": maxerr(opost, y_post)}
    cases.update({"infer/masks": masks, "infer/y": y, "infer/s": s, "infer/w": w, "infer/y_post": y_post})
    meta["infer"] = {"seed": seed1, "max_steps": max_steps, "T": T}

    # ---- case 1b: the same through Tacotron.forward itself (tacotron.py:29-56): ids -> encoder -> decoder -> postnet ----
    with torch.no_grad():
        torch.manual_seed(seed1)
        ye, ype, se, oute = model(ids, lengths, max_steps=max_steps)
    assert maxerr(ye, y) == 0.0 and maxerr(ype, y_post) == 0.0  # (the eval-mode encoder draws nothing from the generator)
    cases.update({"e2e/y": ye, "e2e/y_post": ype, "e2e/s": se, "e2e/w": oute["w"], "e2e/kl_loss": oute["kl_loss"].reshape(1)})
    meta["e2e"] = {"seed": seed1, "max_steps": max_steps, "T": int(ye.shape[1])}

    # ---- case 2: batch-global, inclusive stop rule fired mid-sequence ----
    m_t = s[:, :, 0].min(dim=0).values  # per-step batch minimum of the stop logit
    stop_step, thr = None, None
    for t in range(3, T - 2):
        if m_t[t] < m_t[:t].min():
            cand = float((m_t[t] + m_t[:t].min()) / 2)
            stop_step, thr = t, cand
            break
    assert stop_step is not None, "no usable stop step; change seed"
    dec2 = copy.deepcopy(model.decoder)
    dec2.stop_threshold = thr  # ctor arg stop_threshold (decoder.py:6,9)
    with torch.no_grad():
        torch.manual_seed(seed1)
        y2, s2, w2 = dec2(memory, mmask, None, 40, p_no_forcing=0.1)
    assert y2.shape[1] == stop_step + 1, (y2.shape, stop_step)
    oy2, os2, ow2 = O.decode(dec_w, dims, memory, max_steps=40, stop_threshold=thr, masks=masks)
    assert oy2.shape == y2.shape
    errs["stop"] = {"y": maxerr(oy2, y2), "s": maxerr(os2, s2), "w": maxerr(ow2, w2)}
    cases.update({"stop/y": y2, "stop/s": s2, "stop/w": w2})
    meta["stop"] = {"seed": seed1, "max_steps": 40, "threshold": thr, "T": int(y2.shape[1])}

    # ---- case 3: max_steps=1 edge (T = 2) ----
    with torch.no_grad():
        torch.manual_seed(seed1)
        y3, s3, w3 = model.decoder(memory, mmask, None, 1, p_no_forcing=0.1)
    assert y3.shape[1] == 2
    cases.update({"t2/y": y3, "t2/s": s3, "t2/w": w3})

    # ---- case 4: teacher forcing, always forced (p_no_forcing=None) ----
    gx = torch.Generator().manual_seed(99)
    Tx = 9
    x = torch.randn(B, Tx, 20, generator=gx) * 0.5
    seed4 = 11
    with torch.no_grad():
        torch.manual_seed(seed4)
        y4, s4, w4 = model.decoder(memory, mmask, x, 0, p_no_forcing=None)
    assert y4.shape[1] == Tx
    masks4, _ = replay_masks(seed4, Tx, B, dims.d_pre)
    oy4, os4, ow4 = O.decode(dec_w, dims, memory, masks=masks4, x=x, p_no_forcing=None)
    errs["teacher"] = {"y": maxerr(oy4, y4), "s": maxerr(os4, s4), "w": maxerr(ow4, w4)}
    cases.update({"teacher/x": x, "teacher/masks": masks4, "teacher/y": y4, "teacher/s": s4, "teacher/w": w4})
    meta["teacher"] = {"seed": seed4, "Tx": Tx}

    # ---- case 5: teacher forcing with p_no_forcing=0.5 (rand(1) interleaved with the dropout draws) ----
    seed5 = 13
    with torch.no_grad():
        torch.manual_seed(seed5)
        y5, s5, w5 = model.decoder(memory, mmask, x, 0, p_no_forcing=0.5)
    masks5, flags5 = replay_masks(seed5, Tx, B, dims.d_pre, teacher_p=0.5)
    assert 0 < sum(flags5) < len(flags5), flags5
    oy5, os5, ow5 = O.decode(dec_w, dims, memory, masks=masks5, x=x, teacher_flags=flags5)
    errs["teacher_p"] = {"y": maxerr(oy5, y5), "s": maxerr(os5, s5), "w": maxerr(ow5, w5)}
    torch.manual_seed(seed5)
    oy5b, _, _ = O.decode(dec_w, dims, memory, dropout="rng", x=x, p_no_forcing=0.5)
    errs["teacher_p_rng"] = {"y": maxerr(oy5b, y5)}
    cases.update(
        {
            "teacher_p/masks": masks5,
            "teacher_p/flags": torch.tensor(flags5, dtype=torch.uint8),
            "teacher_p/y": y5,
            "teacher_p/s": s5,
            "teacher_p/w": w5,
        }
    )
    meta["teacher_p"] = {"seed": seed5, "Tx": Tx, "p_no_forcing": 0.5}

    # ---- unit vectors for rows a5-a9 (reference submodules called directly) ----
    cell = model.decoder.decoder_cell
    gu = torch.Generator().manual_seed(5)
    with torch.no_grad():
        # a5 prenet
        xin = torch.randn(B, 20, generator=gu)
        torch.manual_seed(21)
        pre_out = cell.pre_net(xin)
        torch.manual_seed(21)
        pm0, pm1 = O.draw_prenet_masks(B, 36, 36)
        cases.update({"unit/prenet_x": xin, "unit/prenet_masks": torch.stack([pm0, pm1]), "unit/prenet_out": pre_out})
        errs["unit_prenet"] = maxerr(O.prenet(xin, dec_w, torch.stack([pm0, pm1])), pre_out)
        # a6 LSTM zoneout cells
        for name, mod, din, H in (("lstm1", cell.attention_rnn, 36 + 40, 72), ("lstm2", cell.decoder_rnn, 72 + 40, 88)):
            xi = torch.randn(B, din, generator=gu)
            hi = torch.randn(B, H, generator=gu) * 0.5
            ci = torch.randn(B, H, generator=gu) * 0.5
            ho, co = mod(xi, (hi, ci))
            cases.update({f"unit/{name}_x": xi, f"unit/{name}_h": hi, f"unit/{name}_c": ci, f"unit/{name}_ho": ho, f"unit/{name}_co": co})
            pref = "decoder_cell.attention_rnn" if name == "lstm1" else "decoder_cell.decoder_rnn"
            oh, oc = O.lstm_zoneout_cell(xi, hi, ci, dec_w, pref, 0.1)
            errs[f"unit_{name}"] = max(maxerr(oh, ho), maxerr(oc, co))
        # a7 attention
        ha = torch.randn(B, 72, generator=gu)
        wi = torch.rand(B, L, generator=gu)
        wi = wi / wi.sum(dim=1, keepdim=True)
        wo = cell.attention_module(ha, wi.clone(), memory, mmask)
        cases.update({"unit/att_h": ha, "unit/att_w": wi, "unit/att_wo": wo})
        errs["unit_att"] = maxerr(O.stepwise_monotonic_attention(ha, wi, memory, dec_w), wo)
        # a9 postnet on a longer random sequence
        yp = torch.randn(2, 37, 20, generator=gu)
        ypo = model.postnet(yp)
        cases.update({"unit/post_y": yp, "unit/post_out": ypo})
        errs["unit_postnet"] = maxerr(O.mel_postnet(yp, post_w, n_layers), ypo)

    # ---- full LJSpeech dims: oracle vs reference (weights not committed) ----
    cfg_full = yaml.safe_load(open(os.path.join(REF, "configs/config-ljspeech.yaml")))
    torch.manual_seed(42)
    full = ref_tacotron.build_tacotron(cfg_full).eval()
    perturb(full, 8)
    fdec, fpost = split_weights(full)
    fd = O.DecoderDims()
    Bf, Lf = 4, 31
    lens_f = torch.tensor([31, 25, 31, 9])
    idsf = torch.randint(1, 40, (Bf, Lf), generator=g)
    for b in range(Bf):
        idsf[b, lens_f[b] :] = 0
    with torch.no_grad():
        memf = full.encoder(idsf, lens_f)
        torch.manual_seed(3)
        yf, sf, wf = full.decoder(memf, ref_tacotron.lengths_to_mask(lens_f), None, 24, p_no_forcing=0.1)
        ypf = full.postnet(yf)
    mf, _ = replay_masks(3, yf.shape[1], Bf, 256)
    oyf, osf, owf = O.decode(fdec, fd, memf, max_steps=24, masks=mf)
    opf = O.mel_postnet(oyf, fpost, 3)
    errs["ljspeech_dims"] = {
        "T": int(yf.shape[1]),
        "y": maxerr(oyf, yf),
        "s": maxerr(osf, sf),
        "w": maxerr(owf, wf),
        "y_post": maxerr(opf, ypf),
        "argmax_equal": bool((owf.argmax(-1) == wf.argmax(-1)).all()),
        "y_absmax": float(yf.abs().max()),
    }
    meta["memory_range_full"] = [float(memf.min()), float(memf.max())]
    meta["stop_logit_range_full"] = [float(sf.min()), float(sf.max())]

    meta["oracle_vs_reference_maxabs"] = errs
    meta["small_dims"] = dims.__dict__ if hasattr(dims, "__dict__") else None
    meta["small_dims"] = {"d_mel": 20, "r": 1, "d_pre": 36, "d_ctx": 40, "h_att": 72, "h_dec": 88, "postnet_hidden": 64, "postnet_layers": 3, "B": B, "L": L, "lengths": lengths.tolist()}

    model_npz = {"memory": memory.numpy(), "lengths": lengths.numpy(), "ids": ids.numpy()}
    for k, v in dec_w.items():
        model_npz["dec/" + k] = v.numpy()
    for k, v in post_w.items():
        model_npz["post/" + k] = v.numpy()
    for k, v in model.state_dict().items():  # the encoder's parameters too: the end-to-end Tacotron.forward test starts from ids
        if k.startswith("encoder.") and not k.endswith("num_batches_tracked"):
            model_npz["enc/" + k[len("encoder."):]] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "small_model.npz"), **model_npz)
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **{k: v.detach().numpy() for k, v in cases.items()})
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(json.dumps(errs, indent=1))
    for fn in ("small_model.npz", "small_cases.npz", "meta.json"):
        print(fn, os.path.getsize(os.path.join(HERE, fn)), "bytes")


if __name__ == "__main__":
    main()
