"""Randomised parity sweep of every boundary of include/pcoa.h onto the Gram kernels (GPU only).

Seeded: the same cases every run.  Sizes are ragged on purpose (tile edges at 32 / 64 / 256 samples, k-block edges at
16 / 32 variants, odd row strides, padding filled with garbage), densities from almost empty to almost full."""
import os

import numpy as np
import pytest

from conftest import int_gram, load_pkg

pytestmark = pytest.mark.gpu

N_CASES = int(os.environ.get("PCOA_FUZZ_CASES", "48"))


@pytest.fixture(scope="module")
def P():
    return load_pkg()


def _case(seed):
    rng = np.random.default_rng(seed)
    edges_n = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 1025]
    edges_v = [1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 129, 1023, 1025]
    n = int(rng.choice(edges_n)) if rng.random() < 0.5 else int(rng.integers(1, 1300))
    v = int(rng.choice(edges_v)) if rng.random() < 0.5 else int(rng.integers(1, 4000))
    dens = float(rng.choice([0.01, 0.1, 0.3, 0.6, 0.97]))
    x = (rng.random((v, n)) < dens).astype(np.uint8)
    return rng, n, v, x


@pytest.mark.parametrize("seed", range(N_CASES))
def test_every_boundary_agrees_with_integer_matmul(P, seed):
    import torch
    ingest = load_pkg("ingest")
    rng, n, v, x = _case(1000 + seed)
    want = int_gram(x)
    pad = int(rng.integers(0, 9))
    kernel = ["auto", "auto", "fp4", "i8"][seed % 4]
    operand = "fp4" if seed % 3 == 2 else "bits"   # form of the binary-tile operand in HBM (PCOA_FLAG_OPERAND_FP4 / default)
    with P.PcoaEngine(n, gram_kernel=kernel, operand=operand) as eng:
        # fp32, device pointer, padded stride, NaN in the padding
        buf = torch.full((v, n + pad), float("nan"), dtype=torch.float32, device="cuda")
        buf[:, :n] = torch.from_numpy(x.astype(np.float32)).cuda()
        eng.accumulate_dense(buf)
        assert np.array_equal(eng.gram(), want), ("f32 device", n, v, pad, kernel)
        eng.reset()
        # uint8, device pointer, padded stride, 0xff in the padding
        b8 = torch.full((v, n + pad), 255, dtype=torch.uint8, device="cuda")
        b8[:, :n] = torch.from_numpy(x).cuda()
        eng.accumulate_dense_u8(b8)
        assert np.array_equal(eng.gram(), want), ("u8 device", n, v, pad, kernel)
        eng.reset()
        # host tiles
        eng.accumulate_dense(x.astype(np.float32))
        eng.accumulate_dense_u8(x)
        assert np.array_equal(eng.gram(), 2 * want), ("host", n, v, kernel)
        eng.reset()
        # carrier bitsets (device, padded stride with garbage) and CSR carrier lists
        bits = ingest.pack_bits(x, pad_words=pad % 3)
        if pad % 3:
            bits[:, (n + 31) // 32:] = 0xa5a5a5a5
        eng.accumulate_bits(torch.from_numpy(bits.view(np.int32)).cuda())
        eng.accumulate_callsets([list(np.nonzero(r)[0]) for r in x])
        assert np.array_equal(eng.gram(), 2 * want), ("bits + csr", n, v, kernel)
        # the same bitsets from HOST memory (r06: through two device slots on the copy stream), padded stride included
        eng.accumulate_bits(bits)
        assert np.array_equal(eng.gram(), 3 * want), ("host bits", n, v, kernel)
        if kernel != "i8":
            assert eng.timings()["operand_bits"] == (4 if operand == "fp4" else 1)


@pytest.mark.parametrize("seed", range(8))
def test_multiplicities_take_the_int8_kernel_and_stay_exact(P, seed):
    rng, n, v, x = _case(5000 + seed)
    xm = x.astype(np.int64) * rng.integers(1, 128, size=x.shape)
    want = xm.T @ xm
    with P.PcoaEngine(n) as eng:
        eng.accumulate_dense(xm.astype(np.float32))
        eng.accumulate_dense_u8(xm.astype(np.uint8))
        assert np.array_equal(eng.gram(), 2 * want)
        assert eng.timings()["gram_kernel_kind"] == (2 if (xm > 1).any() else 3)
