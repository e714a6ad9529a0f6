"""oracle/sampler_oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy) of the sampling step behind `generate(do_sample=True, temperature=0.2, ...)` as the reference
calls it (/root/reference/gpt4roi/app.py:293-300).  The sampling arithmetic is not under /root/reference: it lives in HF
transformers (pinned at git cae78c46, pyproject.toml:19) -- `TemperatureLogitsWarper`, `TopKLogitsWarper`
(removes scores < the k-th largest, GenerationConfig default top_k = 50), `TopPLogitsWarper` (default top_p = 1.0) -- and
ends in `torch.multinomial`, whose random stream is an implementation detail of the device.  What this oracle pins:

  * the Philox4x32-10 counter-based generator (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11)
    against the Random123 known-answer vectors (`PHILOX_KAT`, checked in tests/test_generation_cpu.py);
  * the warper chain on a logits row (kept set identical to HF's warpers, checked against the container's transformers);
  * the draw: inverse CDF over the kept tokens in ascending vocabulary order with u = philox(counter=(step,0,0,0),
    key=seed)[0] >> 8 scaled to [0,1) -- a sampler of the same distribution as HF's, with a reproducible stream.
"""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF

# Random123 kat_vectors, philox4x32 with 10 rounds: (counter, key) -> output
PHILOX_KAT = [
    ((0x00000000, 0x00000000, 0x00000000, 0x00000000), (0x00000000, 0x00000000),
     (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff), (0xffffffff, 0xffffffff),
     (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def philox4x32_10(counter, key):
    c0, c1, c2, c3 = [int(x) & MASK for x in counter]
    k0, k1 = [int(x) & MASK for x in key]
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def uniform(step, seed):
    """The uniform of decode step `step` for a 64-bit seed: 24 bits in [0, 1), as a float32."""
    x = philox4x32_10((step, 0, 0, 0), (seed & MASK, (seed >> 32) & MASK))[0]
    return np.float32(x >> 8) * np.float32(1.0 / 16777216.0)


def kept_and_weights(logits, temperature=1.0, top_k=50, top_p=1.0):
    """-> (ascending kept indices, their unnormalised weights exp((l - max) / T) as float32)."""
    l = np.asarray(logits, dtype=np.float32)
    n = l.size
    m = l.max()
    inv_t = np.float32(1.0) / np.float32(temperature)
    if 1 <= top_k < n:
        kth = np.partition(l, n - top_k)[n - top_k]            # k-th largest
        keep = np.nonzero(l >= kth)[0]                         # ties with the k-th stay (scores < kth are removed)
    else:
        keep = np.arange(n)
    e = np.exp(((l[keep] - m) * inv_t).astype(np.float32)).astype(np.float32)
    if top_p < 1.0:
        z = e.astype(np.float64).sum()
        stay = np.zeros(keep.size, dtype=bool)
        for a in range(keep.size):
            more = (e > e[a]) | ((e == e[a]) & (keep < keep[a]))
            stay[a] = e[more].astype(np.float64).sum() < float(np.float32(top_p)) * z
        keep, e = keep[stay], e[stay]
    return keep, e


def sample(logits, step, seed, temperature=1.0, top_k=50, top_p=1.0, u=None):
    """One draw; `u` overrides the generator (to replay the uniforms a kernel reports)."""
    keep, e = kept_and_weights(logits, temperature, top_k, top_p)
    if u is None:
        u = uniform(step, seed)
    target = float(np.float32(u)) * float(e.astype(np.float64).cumsum()[-1])
    acc = 0.0
    for idx, w in zip(keep, e):
        acc += float(w)
        if acc > target:
            return int(idx)
    return int(keep[-1])
