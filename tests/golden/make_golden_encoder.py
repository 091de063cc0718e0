#!/usr/bin/env python3
"""Golden vectors for SURVEY 8f rank 2 - Encoder2 - produced by running the REFERENCE.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_encoder.py
Writes tests/golden/encoder_small.npz (weights at reduced dims, ids, lengths, memory) + encoder_meta.json."""
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
import tacotron as ref_tacotron  # noqa: E402
from oracle import tacotron_oracle as O  # noqa: E402

torch.set_num_threads(4)


def main():
    cfg = copy.deepcopy(yaml.safe_load(open(os.path.join(REF, "configs/config-ljspeech.yaml"))))
    cfg["audio"]["num_mels"] = 20
    cfg["model"]["encoder"].update(dim_emb=24, dim_out=40)
    cfg["model"]["decoder"].update(dim_pre=36, dim_att=72, dim_rnn=[72, 88])
    cfg["model"]["postnet"].update(dim_hidden=64)
    torch.manual_seed(42)
    enc = ref_tacotron.build_tacotron(cfg).eval().encoder
    g = torch.Generator().manual_seed(15)
    sd = enc.state_dict()
    for k, v in sd.items():
        if k.endswith("running_var"):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
        elif k.endswith("running_mean") or k.endswith(".bias") or k in ("rnn_h0", "rnn_c0"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith("1.weight") or k.endswith("4.weight"):
            v.add_(0.1 * torch.randn(v.shape, generator=g))
    enc.load_state_dict(sd)
    wts = {k: v.detach().clone() for k, v in enc.state_dict().items() if not k.endswith("num_batches_tracked")}
    B, L = 5, 19
    lengths = torch.tensor([19, 11, 5, 19, 1])
    ids = torch.randint(1, 40, (B, L), generator=g)
    for b in range(B):
        ids[b, lengths[b]:] = 0
    with torch.no_grad():
        memory = enc(ids, lengths)
    # a second case where no utterance fills the padded length: output is max(lengths) long
    lengths2 = torch.tensor([7, 12, 3])
    ids2 = torch.randint(1, 40, (3, 16), generator=g)
    for b in range(3):
        ids2[b, lengths2[b]:] = 0
    with torch.no_grad():
        memory2 = enc(ids2, lengths2)
    assert memory2.shape[1] == 12
    errs = {"case1": float((O.encoder2(ids, lengths, wts) - memory).abs().max()),
            "case2": float((O.encoder2(ids2, lengths2, wts) - memory2).abs().max())}
    npz = {"ids": ids.numpy(), "lengths": lengths.numpy(), "memory": memory.numpy(),
           "ids2": ids2.numpy(), "lengths2": lengths2.numpy(), "memory2": memory2.numpy()}
    for k, v in wts.items():
        npz["w/" + k] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "encoder_small.npz"), **npz)
    json.dump({"dims": {"alphabet": int(wts["emb.weight"].shape[0]), "d_emb": 24, "d_out": 40}, "oracle_vs_reference_maxabs": errs,
               "keys": sorted(wts.keys())}, open(os.path.join(HERE, "encoder_meta.json"), "w"), indent=1)
    print(errs, os.path.getsize(os.path.join(HERE, "encoder_small.npz")))
    print(sorted(wts.keys()))


if __name__ == "__main__":
    main()
