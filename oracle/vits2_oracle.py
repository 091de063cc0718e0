"""CPU oracle for the VITS2 second hot path: TextEncoder and the
reverse pass of the transformer-coupling flow.

TEST INFRASTRUCTURE ONLY (see oracle/tacotron_oracle.py): torch-CPU fp32 restatement over flat weight
dicts keyed like the reference's state dicts; nothing in the product package may import it.

Parity pin: the building blocks (attentions.Encoder with and without the relative-position window,
attentions.FFN, modules.LayerNorm, modules.WN with weight-norm, commons.fused_add_tanh_sigmoid_multiply,
modules.Flip) are checked against vectors produced by importing those reference modules directly
(tests/golden/make_golden_vits2.py).  vits2/models.py itself cannot be imported here (it needs the
unbuilt `monotonic_align` Cython extension, and no stand-in is written for it), so the ~20 lines of glue
in TextEncoder.forward (models.py:369-380) and ResidualCouplingTransformersLayer.forward (models.py:506-531)
/ the block driver (models.py:803-810) are restated below and pinned only through the golden script's own
composition of the reference blocks: parity of that glue is "pinned by composition", not by a reference run.

All tensors are channel-first [B, C, T] like the reference."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Weights = Dict[str, Tensor]


@dataclass(frozen=True)
class Vits2Dims:
    """ModelConfig defaults (vits2/cli.py:159-180) and the flow constants of SynthesizerTrn (models.py:1191-1200)."""

    n_vocab: int = 178
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3
    window_size: int = 4
    # flow: ResidualCouplingTransformersBlock(inter, hidden, 5, 1, 4, n_flows=4, "pre_conv")
    flow_hidden: int = 192
    flow_kernel: int = 5
    flow_wn_layers: int = 4
    n_flows: int = 4
    flow_tf_layers: int = 2  # pre_transformer: Encoder(half, half, n_heads=2, n_layers=2, kernel_size=3, window_size=None)
    flow_tf_heads: int = 2
    flow_tf_kernel: int = 3
    # speaker conditioning (0 = none): WN.cond_layer per coupling layer (modules.py:149-153) and the text encoder's
    # spk_emb_linear, added at the input of layer cond_layer_idx (attentions.py:41-52, 80-84)
    gin_channels: int = 0
    cond_layer_idx: int = 2


def sequence_mask(lengths: Tensor, T: int) -> Tensor:
    """commons.sequence_mask: [B, T] bool, True inside the utterance."""
    return torch.arange(T)[None, :] < lengths[:, None]


def layer_norm_c(x: Tensor, gamma: Tensor, beta: Tensor, eps: float = 1e-5) -> Tensor:
    """modules.LayerNorm.forward (modules.py:24-27): layer norm over the CHANNEL dim of [B, C, T]."""
    return F.layer_norm(x.transpose(1, -1), (x.shape[1],), gamma, beta, eps).transpose(1, -1)


def mha(x: Tensor, attn_mask: Tensor, wts: Weights, prefix: str, n_heads: int, window: Optional[int]) -> Tensor:
    """attentions.MultiHeadAttention.forward / .attention (attentions.py:234-295) for self-attention.
    The reference builds the relative-position terms by pad/reshape tricks (:320-368); stated directly:
      scores[i, j] += (q_i / sqrt(dk)) . E_k[j - i + w]      for |j - i| <= w
      out[i]       += sum_{|j-i|<=w} p[i, j] * E_v[j - i + w]
    with E_k / E_v the [2w+1, dk] tables shared by all heads (heads_share=True)."""
    B, C, T = x.shape
    dk = C // n_heads
    q = F.conv1d(x, wts[prefix + ".conv_q.weight"], wts[prefix + ".conv_q.bias"])
    k = F.conv1d(x, wts[prefix + ".conv_k.weight"], wts[prefix + ".conv_k.bias"])
    v = F.conv1d(x, wts[prefix + ".conv_v.weight"], wts[prefix + ".conv_v.bias"])
    q = q.view(B, n_heads, dk, T).transpose(2, 3) / math.sqrt(dk)  # [B, h, T, dk]
    k = k.view(B, n_heads, dk, T).transpose(2, 3)
    v = v.view(B, n_heads, dk, T).transpose(2, 3)
    scores = torch.matmul(q, k.transpose(-2, -1))  # [B, h, T, T]
    if window is not None:
        ek = wts[prefix + ".emb_rel_k"][0]  # [2w+1, dk]
        rel = torch.matmul(q, ek.t())  # [B, h, T, 2w+1]
        i = torch.arange(T)[:, None]
        j = torch.arange(T)[None, :]
        r = j - i + window
        ok = (r >= 0) & (r <= 2 * window)
        local = torch.gather(rel, 3, r.clamp(0, 2 * window).expand(B, n_heads, T, T)) * ok
        scores = scores + local
    scores = scores.masked_fill(attn_mask == 0, -1e4)
    p = F.softmax(scores, dim=-1)
    out = torch.matmul(p, v)  # [B, h, T, dk]
    if window is not None:
        ev = wts[prefix + ".emb_rel_v"][0]
        pr = torch.gather(p, 3, (torch.arange(T)[:, None] + torch.arange(2 * window + 1)[None, :] - window).clamp(0, T - 1).expand(B, n_heads, T, 2 * window + 1))
        jj = torch.arange(T)[:, None] + torch.arange(2 * window + 1)[None, :] - window
        pr = pr * ((jj >= 0) & (jj < T))
        out = out + torch.matmul(pr, ev)
    out = out.transpose(2, 3).contiguous().view(B, C, T)
    return F.conv1d(out, wts[prefix + ".conv_o.weight"], wts[prefix + ".conv_o.bias"])


def ffn(x: Tensor, x_mask: Tensor, wts: Weights, prefix: str, kernel: int) -> Tensor:
    """attentions.FFN.forward with same-padding and ReLU (attentions.py:411-419, 432-440)."""
    pl, pr = (kernel - 1) // 2, kernel // 2
    y = F.conv1d(F.pad(x * x_mask, (pl, pr)), wts[prefix + ".conv_1.weight"], wts[prefix + ".conv_1.bias"])
    y = torch.relu(y)
    y = F.conv1d(F.pad(y * x_mask, (pl, pr)), wts[prefix + ".conv_2.weight"], wts[prefix + ".conv_2.bias"])
    return y * x_mask


def encoder_stack(x: Tensor, x_mask: Tensor, wts: Weights, prefix: str, n_layers: int, n_heads: int,
                  window: Optional[int], kernel: int, g: Optional[Tensor] = None, cond_layer_idx: int = -1) -> Tensor:
    """attentions.Encoder.forward in eval mode (attentions.py:76-93); g [B, gin, 1] enters at layer cond_layer_idx (:80-84)."""
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)
    x = x * x_mask
    for i in range(n_layers):
        if i == cond_layer_idx and g is not None:
            gp = F.linear(g.transpose(1, 2), wts[prefix + ".spk_emb_linear.weight"], wts[prefix + ".spk_emb_linear.bias"]).transpose(1, 2)
            x = (x + gp) * x_mask
        y = mha(x, attn_mask, wts, f"{prefix}.attn_layers.{i}", n_heads, window)
        x = layer_norm_c(x + y, wts[f"{prefix}.norm_layers_1.{i}.gamma"], wts[f"{prefix}.norm_layers_1.{i}.beta"])
        y = ffn(x, x_mask, wts, f"{prefix}.ffn_layers.{i}", kernel)
        x = layer_norm_c(x + y, wts[f"{prefix}.norm_layers_2.{i}.gamma"], wts[f"{prefix}.norm_layers_2.{i}.beta"])
    return x * x_mask


def text_encoder(ids: Tensor, lengths: Tensor, wts: Weights, dims: Vits2Dims, prefix: str = "enc_p",
                 g: Optional[Tensor] = None) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """TextEncoder.forward (models.py:369-380): returns (x, m, logs, x_mask)."""
    H = dims.hidden_channels
    x = F.embedding(ids, wts[prefix + ".emb.weight"]) * math.sqrt(H)  # [B, T, H]
    x = x.transpose(1, -1)
    x_mask = sequence_mask(lengths, x.shape[2]).unsqueeze(1).to(x.dtype)
    x = encoder_stack(x * x_mask, x_mask, wts, prefix + ".encoder", dims.n_layers, dims.n_heads, dims.window_size, dims.kernel_size,
                      g=g, cond_layer_idx=dims.cond_layer_idx if dims.gin_channels else -1)
    stats = F.conv1d(x, wts[prefix + ".proj.weight"], wts[prefix + ".proj.bias"]) * x_mask
    m, logs = torch.split(stats, dims.inter_channels, dim=1)
    return x, m, logs, x_mask


def weight_norm_weight(wts: Weights, prefix: str) -> Tensor:
    """Effective weight of a torch.nn.utils.weight_norm'd conv (dim=0): g * v / ||v|| per output channel."""
    if prefix + ".weight" in wts:
        return wts[prefix + ".weight"]
    v, g = wts[prefix + ".weight_v"], wts[prefix + ".weight_g"]
    return v * (g / v.flatten(1).norm(dim=1).view(-1, 1, 1))


def wn(x: Tensor, x_mask: Tensor, wts: Weights, prefix: str, n_layers: int, kernel: int, dilation_rate: int = 1,
       g: Optional[Tensor] = None) -> Tensor:
    """modules.WN.forward (modules.py:185-210) and commons.fused_add_tanh_sigmoid_multiply (:102-109); g [B, gin, 1 or T]."""
    H = x.shape[1]
    output = torch.zeros_like(x)
    if g is not None:
        g = F.conv1d(g, weight_norm_weight(wts, f"{prefix}.cond_layer"), wts[f"{prefix}.cond_layer.bias"])  # :189-190
    for i in range(n_layers):
        d = dilation_rate**i
        pad = (kernel * d - d) // 2
        x_in = F.conv1d(x, weight_norm_weight(wts, f"{prefix}.in_layers.{i}"), wts[f"{prefix}.in_layers.{i}.bias"], padding=pad, dilation=d)
        if g is not None:
            x_in = x_in + g[:, i * 2 * H:(i + 1) * 2 * H, :]  # g_l, :193-196
        acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
        rs = F.conv1d(acts, weight_norm_weight(wts, f"{prefix}.res_skip_layers.{i}"), wts[f"{prefix}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :H]) * x_mask
            output = output + rs[:, H:]
        else:
            output = output + rs
    return output * x_mask


def coupling_reverse(x: Tensor, x_mask: Tensor, wts: Weights, prefix: str, dims: Vits2Dims, g: Optional[Tensor] = None) -> Tensor:
    """ResidualCouplingTransformersLayer.forward(reverse=True), mean_only (models.py:506-531)."""
    half = dims.inter_channels // 2
    x0, x1 = torch.split(x, [half, half], 1)
    x0_ = encoder_stack(x0 * x_mask, x_mask, wts, prefix + ".pre_transformer", dims.flow_tf_layers, dims.flow_tf_heads, None, dims.flow_tf_kernel)
    x0_ = x0_ + x0
    h = F.conv1d(x0_, wts[prefix + ".pre.weight"], wts[prefix + ".pre.bias"]) * x_mask
    h = wn(h, x_mask, wts, prefix + ".enc", dims.flow_wn_layers, dims.flow_kernel, g=g)
    m = F.conv1d(h, wts[prefix + ".post.weight"], wts[prefix + ".post.bias"]) * x_mask
    x1 = (x1 - m) * x_mask  # logs = 0 in mean-only mode: exp(-logs) = 1
    return torch.cat([x0, x1], 1)


def flow_reverse(z: Tensor, y_mask: Tensor, wts: Weights, dims: Vits2Dims, prefix: str = "flow", g: Optional[Tensor] = None) -> Tensor:
    """ResidualCouplingTransformersBlock.forward(reverse=True) (models.py:803-810): flows are
    [layer_0, Flip, layer_1, Flip, ...]; reversed: Flip, layer_{n-1}, ..., Flip, layer_0."""
    x = z
    for i in reversed(range(dims.n_flows)):
        x = torch.flip(x, [1])  # modules.Flip (modules.py:374-381)
        x = coupling_reverse(x, y_mask, wts, f"{prefix}.flows.{2 * i}", dims, g=g)
    return x


# --------------------------------------------------------------------------
# synthetic weights (the reference zero-initialises `post`; a random init there keeps the test honest)
# --------------------------------------------------------------------------
def _encoder_weights(w: Weights, prefix: str, C: int, Fc: int, n_layers: int, n_heads: int, window: Optional[int], kernel: int, g: torch.Generator):
    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    dk = C // n_heads
    for i in range(n_layers):
        a = f"{prefix}.attn_layers.{i}"
        for nm in ("conv_q", "conv_k", "conv_v", "conv_o"):
            w[f"{a}.{nm}.weight"] = rn(C, C, 1, scale=C**-0.5)
            w[f"{a}.{nm}.bias"] = rn(C, scale=0.1)
        if window is not None:
            w[f"{a}.emb_rel_k"] = rn(1, 2 * window + 1, dk, scale=dk**-0.5)
            w[f"{a}.emb_rel_v"] = rn(1, 2 * window + 1, dk, scale=dk**-0.5)
        for nl in ("norm_layers_1", "norm_layers_2"):
            w[f"{prefix}.{nl}.{i}.gamma"] = 1.0 + rn(C, scale=0.1)
            w[f"{prefix}.{nl}.{i}.beta"] = rn(C, scale=0.1)
        f = f"{prefix}.ffn_layers.{i}"
        w[f"{f}.conv_1.weight"] = rn(Fc, C, kernel, scale=(C * kernel) ** -0.5)
        w[f"{f}.conv_1.bias"] = rn(Fc, scale=0.1)
        w[f"{f}.conv_2.weight"] = rn(C, Fc, kernel, scale=(Fc * kernel) ** -0.5)
        w[f"{f}.conv_2.bias"] = rn(C, scale=0.1)


def random_vits2_weights(dims: Vits2Dims, seed: int = 0) -> Weights:
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, scale=1.0):
        return torch.randn(*shape, generator=g) * scale

    w: Weights = {}
    H, I = dims.hidden_channels, dims.inter_channels
    w["enc_p.emb.weight"] = rn(dims.n_vocab, H, scale=H**-0.5)
    _encoder_weights(w, "enc_p.encoder", H, dims.filter_channels, dims.n_layers, dims.n_heads, dims.window_size, dims.kernel_size, g)
    gin = dims.gin_channels
    w["enc_p.proj.weight"] = rn(2 * I, H, 1, scale=H**-0.5)
    w["enc_p.proj.bias"] = rn(2 * I, scale=0.1)
    half, Fh = I // 2, dims.flow_hidden
    for i in range(dims.n_flows):
        p = f"flow.flows.{2 * i}"
        _encoder_weights(w, p + ".pre_transformer", half, half, dims.flow_tf_layers, dims.flow_tf_heads, None, dims.flow_tf_kernel, g)
        w[p + ".pre.weight"] = rn(Fh, half, 1, scale=half**-0.5)
        w[p + ".pre.bias"] = rn(Fh, scale=0.1)
        for j in range(dims.flow_wn_layers):
            co = 2 * Fh
            w[f"{p}.enc.in_layers.{j}.weight_v"] = rn(co, Fh, dims.flow_kernel, scale=(Fh * dims.flow_kernel) ** -0.5)
            w[f"{p}.enc.in_layers.{j}.weight_g"] = 0.5 + torch.rand(co, 1, 1, generator=g)
            w[f"{p}.enc.in_layers.{j}.bias"] = rn(co, scale=0.1)
            cr = 2 * Fh if j < dims.flow_wn_layers - 1 else Fh
            w[f"{p}.enc.res_skip_layers.{j}.weight_v"] = rn(cr, Fh, 1, scale=Fh**-0.5)
            w[f"{p}.enc.res_skip_layers.{j}.weight_g"] = 0.5 + torch.rand(cr, 1, 1, generator=g)
            w[f"{p}.enc.res_skip_layers.{j}.bias"] = rn(cr, scale=0.1)
        w[p + ".post.weight"] = rn(half, Fh, 1, scale=Fh**-0.5)
        w[p + ".post.bias"] = rn(half, scale=0.1)
    if gin:  # (drawn last: the unconditioned tensors above do not depend on gin_channels)
        w["enc_p.encoder.spk_emb_linear.weight"] = rn(H, gin, scale=gin**-0.5)
        w["enc_p.encoder.spk_emb_linear.bias"] = rn(H, scale=0.1)
        for i in range(dims.n_flows):
            p = f"flow.flows.{2 * i}.enc.cond_layer"
            nc = 2 * Fh * dims.flow_wn_layers
            w[p + ".weight_v"] = rn(nc, gin, 1, scale=gin**-0.5)
            w[p + ".weight_g"] = 0.5 + torch.rand(nc, 1, 1, generator=g)
            w[p + ".bias"] = rn(nc, scale=0.1)
    return w
