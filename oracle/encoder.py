"""Oracle: the `-m enc-only` model as a plain functional PyTorch-CPU forward.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates
  EncoderOnlyTransformer   /root/reference/protein_transformer/models/encoder_only.py:10-45
  Encoder / EncoderLayer   .../models/transformer/Encoder.py:8-54
  MultiHeadedAttention     .../models/transformer/Attention.py:5-69
  SublayerConnection, PositionwiseFeedForward, PositionalEncoding, Embeddings
                           .../models/transformer/Sublayers.py:5-72
over a dict of tensors that uses the reference's state_dict keys (SURVEY.md
Appendix E), so a state_dict saved from the reference model loads unchanged.
Dropout is always 0 here (RNG streams cannot match).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

PAD_ID = 20
VOCAB_SIZE = 22
NUM_OUT = 24


def positional_table(max_len, dm):
    # Sublayers.py:48-56
    pe = torch.zeros(max_len, dm)
    position = torch.arange(0., max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0., dm, 2) * -(np.log(10000.0) / dm))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def init_params(nlayers, dmodel, dff, max_seq_len, angle_means, seed=None):
    """Parameters initialised the way encoder_only.py:24-34 does it.

    xavier_uniform on every >=2-D parameter, torch defaults for 1-D ones
    (Linear bias U(+-1/sqrt(fan_in)), LayerNorm 1/0), then the output layer is
    weight = 0, bias = arctanh(angle_means).
    """
    if seed is not None:
        torch.manual_seed(seed)
    p = {}

    def linear(prefix, fin, fout):
        w = torch.empty(fout, fin)
        torch.nn.init.xavier_uniform_(w)
        b = torch.empty(fout).uniform_(-1 / math.sqrt(fin), 1 / math.sqrt(fin))
        p[prefix + ".weight"], p[prefix + ".bias"] = w, b

    emb = torch.empty(VOCAB_SIZE, dmodel)
    torch.nn.init.xavier_uniform_(emb)
    p["encoder.input_embedding.emb.weight"] = emb
    p["encoder.positional_enc.pe"] = positional_table(max_seq_len, dmodel)
    for i in range(nlayers):
        base = f"encoder.enc_layers.{i}."
        for nm in ("wq", "wk", "wv", "wo"):
            linear(base + "self_attn." + nm, dmodel, dmodel)
        linear(base + "pwff.layer1", dmodel, dff)
        linear(base + "pwff.layer2", dff, dmodel)
        for j in (0, 1):
            p[base + f"sublayer_connections.{j}.norm.weight"] = torch.ones(dmodel)
            p[base + f"sublayer_connections.{j}.norm.bias"] = torch.zeros(dmodel)
    p["output_projection.weight"] = torch.zeros(NUM_OUT, dmodel)
    p["output_projection.bias"] = torch.tensor(np.arctanh(np.asarray(angle_means)), dtype=torch.float32)
    return p


def n_layers_of(params):
    return 1 + max(int(k.split(".")[2]) for k in params if k.startswith("encoder.enc_layers."))


def attention(p, base, x, key_mask, nhead):
    # Attention.py:47-69 with ScaledDotProductAttention :14-22, dropout off
    B, L, D = x.shape
    dk = D // nhead
    q = F.linear(x, p[base + "wq.weight"], p[base + "wq.bias"])
    k = F.linear(x, p[base + "wk.weight"], p[base + "wk.bias"])
    v = F.linear(x, p[base + "wv.weight"], p[base + "wv.bias"])
    q, k, v = (t.view(B, -1, nhead, dk).transpose(1, 2) for t in (q, k, v))
    scores = torch.matmul(q, k.transpose(-2, -1)) / np.sqrt(dk)
    scores = scores.masked_fill(key_mask[:, None, None, :] == 0, -np.inf)
    probs = torch.softmax(scores, dim=-1)
    o = torch.matmul(probs, v).transpose(1, 2).contiguous().view(B, -1, D)
    return F.linear(o, p[base + "wo.weight"], p[base + "wo.bias"])


def encoder_forward(p, seq, nhead, return_hidden=False, use_tanh_out=True):
    """[B,L] int64 -> [B,L,24] tanh'ed (cos,sin) predictions.

    enc-only: encoder_only.py:36-42.  conv-enc (keys `encoder.conv_layers.j.*` present, possibly without an
    embedding): convolutional_encoder.py:41-47,106-123 - Conv1d stack with no activation between the embedding
    (or the one-hot input) and the encoder layers; without embedding the positional term is added after the
    convolutions as out + (out + pe).
    """
    key_mask = seq != PAD_ID                                   # encoder_only.py:37
    pe = p["encoder.positional_enc.pe"][:, :seq.shape[1]]
    has_emb = "encoder.input_embedding.emb.weight" in p
    if has_emb:
        emb = p["encoder.input_embedding.emb.weight"]
        x0 = emb[seq] * np.sqrt(emb.shape[1])                  # Sublayers.py:72
        x = x0 + (x0 + pe)                                     # Encoder.py:30 + Sublayers.py:59-62
    else:
        x = F.one_hot(seq, num_classes=VOCAB_SIZE).float()     # convolutional_encoder.py:110-111
    n_conv = len([k for k in p if k.startswith("encoder.conv_layers.") and k.endswith(".weight")])
    if n_conv:
        x = x.transpose(-1, -2)
        for j in range(n_conv):
            w = p[f"encoder.conv_layers.{j}.weight"]
            x = F.conv1d(x, w, p[f"encoder.conv_layers.{j}.bias"], padding=(w.shape[2] - 1) // 2)
        x = x.transpose(-1, -2)
    if not has_emb:
        x = x + (x + pe)                                       # convolutional_encoder.py:118-119
    D = x.shape[-1]
    for i in range(n_layers_of(p)):
        base = f"encoder.enc_layers.{i}."
        n0w, n0b = p[base + "sublayer_connections.0.norm.weight"], p[base + "sublayer_connections.0.norm.bias"]
        n1w, n1b = p[base + "sublayer_connections.1.norm.weight"], p[base + "sublayer_connections.1.norm.bias"]
        h = F.layer_norm(x, (D,), n0w, n0b, 1e-5)              # Sublayers.py:17
        x = x + attention(p, base + "self_attn.", h, key_mask, nhead)
        h = F.layer_norm(x, (D,), n1w, n1b, 1e-5)
        h = F.linear(torch.relu(F.linear(h, p[base + "pwff.layer1.weight"], p[base + "pwff.layer1.bias"])),
                     p[base + "pwff.layer2.weight"], p[base + "pwff.layer2.bias"])   # Sublayers.py:34
        x = x + h
    out = F.linear(x, p["output_projection.weight"], p["output_projection.bias"])
    if use_tanh_out:                                           # encoder_only.py:40-41; False = `-m conv-enc-linear-out`
        out = torch.tanh(out)
    return (out, x) if return_hidden else out
