"""TEST-ONLY CPU stand-ins for the three HIP ops, built from the oracle, so that the HOST-side logic of the
product package (module API, state-dict keys, glue arithmetic, matcher, loss, autograd bridges) can be checked on
a machine without a GPU.  Installed only through the ``cpu_kernels`` pytest fixture (monkeypatch); the product
never imports this file and has no CPU path of its own -- the GPU tests (-m gpu) exercise the real kernels."""
import torch

from oracle import msda as OM


class OracleMSDA:
    @staticmethod
    def ms_deform_attn_forward(value, shapes, lsi, loc, attn, im2col_step):
        return OM.msda_forward(value, shapes, lsi, loc, attn)

    @staticmethod
    def ms_deform_attn_backward(value, shapes, lsi, loc, attn, grad_output, im2col_step):
        return OM.msda_backward(value, shapes, lsi, loc, attn, grad_output)


def decoder_self_attention(q, k, v, num_heads, want_maps=True):
    B, N, MD = q.shape
    D = MD // num_heads

    def heads(t):
        return t.view(B, N, num_heads, D).transpose(1, 2).contiguous()

    qh, kh, vh = heads(q), heads(k), heads(v)
    w = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)), dim=-1)
    o = torch.matmul(w, vh).transpose(1, 2).reshape(B, N, MD)
    return (o, qh, kh) if want_maps else (o, None, None)


def relation_head(gate_q, gate_k, uq, uk, b1, w2r, b2r, w3r, b3r, w2c, b2c, w3c, b3c, triplet_dist=None,
                  node_cls=None, want_gate_mean=False, owner=None, sigmoid=False):
    F = torch.nn.functional
    Hd = w2r.shape[1]
    g = torch.sigmoid(gate_q[:, :, None, :] + gate_k[:, None, :, :])
    h1 = torch.relu(torch.einsum("bijt,bitc->bijc", g, uq) + torch.einsum("bijt,bjtc->bijc", g, uk) + b1)
    rel = F.linear(torch.relu(F.linear(h1[..., :Hd], w2r, b2r)), w3r, b3r)
    conn = F.linear(torch.relu(F.linear(h1[..., Hd:], w2c, b2c)), w3c, b3c)
    if triplet_dist is not None:
        rel = rel + torch.stack([triplet_dist[node_cls[i]][:, node_cls[i]] for i in range(rel.shape[0])], 0)
    gm = g.reshape(-1, g.shape[-1]).mean(0) if want_gate_mean else None
    if sigmoid:
        rel, conn = rel.sigmoid(), conn.sigmoid()
    return rel, conn, gm
