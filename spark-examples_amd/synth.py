"""Deterministic synthetic genotypes (bench / tests only; not part of the reference).

Balding-Nichols population structure so that the spectrum of B has clear gaps:
K populations, ancestral frequency p_v log-uniform on [0.001, 0.5], per-population frequency
p_vk ~ Beta(p_v (1-F)/F, (1-p_v)(1-F)/F), F = 0.1, diploid carrier probability 1 - (1 - p_vk)^2.
The per-(variant, population) carrier probabilities are turned into uint32 thresholds on the host;
the genotype itself is pure integer arithmetic,
    X[v, i] = 1  iff  philox4x32-10(key = seed, counter = (v_lo, v_hi, i // 4, 0))[i % 4] < T[v, pop(i)],
so the device kernel (csrc/gram_aux.hip: synth_fill_kernel) and the numpy twin below agree bit for
bit and the data do not depend on how variants are sharded over GPUs.
"""
import numpy as np

# 1000 Genomes phase 3 super-population sizes (AFR, AMR, EAS, EUR, SAS), sum = 2504
KG_POP_SIZES = (661, 347, 504, 503, 489)

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011).  Inputs broadcastable uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = (p1 & _MASK).astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = (p0 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def pop_offsets(n_samples, sizes=KG_POP_SIZES):
    """Population boundaries for n_samples, proportional to `sizes` (exact for n = 2504)."""
    sizes = np.asarray(sizes, dtype=np.float64)
    k = len(sizes)
    if n_samples < k:
        cuts = np.minimum(np.arange(k + 1), n_samples)
        cuts[-1] = n_samples
        return cuts.astype(np.int32)
    cum = np.concatenate([[0.0], np.cumsum(sizes)]) / sizes.sum()
    offs = np.rint(cum * n_samples).astype(np.int64)
    offs[0], offs[-1] = 0, n_samples
    for i in range(1, k + 1):  # every population non-empty
        offs[i] = max(offs[i], offs[i - 1] + (1 if i < k else 0))
    offs[-1] = n_samples
    return offs.astype(np.int32)


def thresholds(seed, first_variant, n_variants, n_pops=5, fst=0.1):
    """uint32 [n_variants][n_pops] carrier thresholds for variants [first, first + n).
    Generated per 65,536-variant block from a block-keyed numpy Philox stream, so any sub-range
    aligned or not gives the same table as the whole."""
    out = np.empty((n_variants, n_pops), dtype=np.uint32)
    blk = 1 << 16
    v = first_variant
    end = first_variant + n_variants
    while v < end:
        b = v // blk
        lo, hi = b * blk, (b + 1) * blk
        rng = np.random.Generator(np.random.Philox(key=[int(seed) & (2**64 - 1), b]))
        pv = np.exp(rng.uniform(np.log(0.001), np.log(0.5), size=blk))
        a = pv * (1.0 - fst) / fst
        bb = (1.0 - pv) * (1.0 - fst) / fst
        pvk = rng.beta(a[:, None], bb[:, None], size=(blk, n_pops))
        q = 1.0 - (1.0 - pvk) ** 2
        t = np.minimum(np.floor(q * 4294967296.0), 4294967295.0).astype(np.uint32)
        s0, s1 = max(v, lo), min(end, hi)
        out[s0 - first_variant:s1 - first_variant] = t[s0 - lo:s1 - lo]
        v = s1
    return out


def genotypes(seed, first_variant, thr, offsets, dtype=np.float32, cols=None):
    """Host twin of synth_fill_kernel: dense [n_variants][n_samples] 0/1 matrix.
    cols = (c0, c1): only sample columns [c0, c1) (the Philox counter is (variant, column // 4), so a column block of a
    very wide cohort -- N = 100,000 -- can be produced without the rest)."""
    thr = np.asarray(thr, dtype=np.uint32)
    offsets = np.asarray(offsets, dtype=np.int64)
    nv = thr.shape[0]
    n = int(offsets[-1])
    c0, c1 = (0, n) if cols is None else (int(cols[0]), int(cols[1]))
    if not (0 <= c0 <= c1 <= n):
        raise ValueError("cols outside [0, n_samples]")
    g0, g1 = c0 // 4, (c1 + 3) // 4
    v = (np.arange(nv, dtype=np.uint64) + np.uint64(first_variant))[:, None]
    g = np.arange(g0, g1, dtype=np.uint32)[None, :]
    w = philox4x32_10((v & _MASK).astype(np.uint32), (v >> np.uint64(32)).astype(np.uint32), g, np.uint32(0),
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=-1).reshape(nv, (g1 - g0) * 4)[:, c0 - 4 * g0:c1 - 4 * g0]
    pop = np.zeros(n, dtype=np.int64)
    for p in range(len(offsets) - 1):
        pop[offsets[p]:offsets[p + 1]] = p
    return (words < thr[:, pop[c0:c1]]).astype(dtype)
