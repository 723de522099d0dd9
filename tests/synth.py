"""Seeded synthetic inputs of SURVEY.md 8(d): HF-layout GPTQ tensors, activations, KV."""
import numpy as np


def gptq_hf(rng, k, n, g, sym=False):
    """qweight (K/8,N) int32 uniform nibbles; qzeros (K/G,N/8) int32 nibbles 0..14 stored as zero-1
    (sym: all 7 => zero 8); scales (K/G,N) fp16 = |N(0,1)|*0.02/8 + 1e-4 (returned as uint16 bits)."""
    nib = rng.integers(0, 16, size=(k, n), dtype=np.uint32)
    qweight = np.zeros((k // 8, n), np.uint32)
    for j in range(8):
        qweight |= nib[j::8, :] << np.uint32(4 * j)
    ng = k // g
    zn = np.full((ng, n), 7, np.uint32) if sym else rng.integers(0, 15, size=(ng, n), dtype=np.uint32)
    qzeros = np.zeros((ng, n // 8), np.uint32)
    for j in range(8):
        qzeros |= zn[:, j::8] << np.uint32(4 * j)
    scales = (np.abs(rng.standard_normal((ng, n))) * 0.02 / 8 + 1e-4).astype(np.float16)
    return qweight, qzeros, scales.view(np.uint16)


def act(rng, m, k, scale=1.0):
    return (rng.standard_normal((m, k)) * scale).astype(np.float16)


def ulp_diff_f16(a_bits, b_bits):
    """|a-b| in fp16 ulps computed on the monotonic integer line (raw uint16 inputs)."""
    def key(u):
        u = u.astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7FFF), u + 0x8000)  # -0 == +0
    return np.abs(key(np.asarray(a_bits)) - key(np.asarray(b_bits)))


AWQ_PACK_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)   # AutoAWQ: nibble i of a packed word holds column 8c + order[i]


def awq_hf(rng, k, n, g):
    """An AWQ ("gemm" version) checkpoint triple and the integers behind it: qweight (K, N/8) int32, qzeros
    (K/G, N/8) int32 (zero points as used, no -1), scales (K/G, N) fp16 bits; plus q (K,N) and z (K/G,N)."""
    q = rng.integers(0, 16, size=(k, n), dtype=np.uint32)
    ng = k // g
    z = rng.integers(0, 16, size=(ng, n), dtype=np.uint32)

    def pack(a):
        out = np.zeros((a.shape[0], n // 8), np.uint32)
        for i, o in enumerate(AWQ_PACK_ORDER):
            out |= a[:, o::8] << np.uint32(4 * i)
        return out
    scales = (np.abs(rng.standard_normal((ng, n))) * 0.02 / 8 + 1e-4).astype(np.float16)
    return pack(q), pack(z), scales.view(np.uint16), q, z


def gptq_act_order_hf(rng, k, n, g):
    """A desc_act GPTQ checkpoint: rows are assigned to groups through a random permutation (g_idx); returns
    (qweight, qzeros, scales_bits, g_idx) in the HF layout plus the dense fp16 matrix W16[n, k] =
    rn16(rn16(q - z) * s) the reference's dequantiser would produce for it."""
    qweight, qzeros, scales_bits = gptq_hf(rng, k, n, g)
    order = rng.permutation(k)
    g_idx = np.empty(k, np.int32)
    g_idx[order] = np.arange(k, dtype=np.int32) // g
    q = np.zeros((k, n), np.int32)
    for j in range(8):
        q[j::8] = (qweight >> np.uint32(4 * j)) & 0xF
    z = np.zeros((k // g, n), np.int32)
    for j in range(8):
        z[:, j::8] = ((qzeros >> np.uint32(4 * j)) & 0xF) + 1
    s = scales_bits.view(np.float16)
    d = (q - z[g_idx]).astype(np.float16)                              # exact
    w16 = (d.astype(np.float32) * s[g_idx].astype(np.float32)).astype(np.float16)   # one rounding of an exact product
    return qweight, qzeros, scales_bits, g_idx, np.ascontiguousarray(w16.T)
