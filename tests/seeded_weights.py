"""Weights that a fixture does not have to store: a TextEncoder state dict drawn from numpy's legacy RandomState (whose
stream is fixed across numpy versions) under a given seed.  tests/golden/make_golden.py loads it into the REFERENCE module
to mint outputs and gradients; the tests load the same values into this package's module."""
import numpy as np


def text_encoder_state(seed, vocab, d, layers, proj):
    rs = np.random.RandomState(seed)

    def mat(n, k, scale=None):
        return (rs.standard_normal((n, k)) * (scale if scale is not None else k ** -0.5)).astype(np.float32)

    def vec(n, base=0.0, scale=0.05):
        return (base + rs.standard_normal(n) * scale).astype(np.float32)

    sd = {"embedding.weight": mat(vocab, d, 1.0)}
    for i in range(layers):
        p = f"transformer.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = mat(3 * d, d)
        sd[p + "self_attn.in_proj_bias"] = vec(3 * d)
        sd[p + "self_attn.out_proj.weight"] = mat(d, d)
        sd[p + "self_attn.out_proj.bias"] = vec(d)
        sd[p + "linear1.weight"] = mat(4 * d, d)
        sd[p + "linear1.bias"] = vec(4 * d)
        sd[p + "linear2.weight"] = mat(d, 4 * d)
        sd[p + "linear2.bias"] = vec(d)
        sd[p + "norm1.weight"] = vec(d, 1.0)
        sd[p + "norm1.bias"] = vec(d)
        sd[p + "norm2.weight"] = vec(d, 1.0)
        sd[p + "norm2.bias"] = vec(d)
    sd["norm.weight"] = vec(d, 1.0)
    sd["norm.bias"] = vec(d)
    sd["projection.weight"] = mat(proj, d)
    sd["projection.bias"] = vec(proj)
    return sd
