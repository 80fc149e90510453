"""GPU parity of the individual HIP kernels (through the C ABI) against the NumPy oracle.

Tolerances: f32 path 2e-5 relative to the output's max magnitude (exact-f32 MFMA, only the
summation order differs from NumPy); bf16 path 2e-2 (inputs rounded to bf16, f32 accumulate).
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from conftest import pkg
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 2e-2}


def relerr(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def bf16_round(a):
    return torch.tensor(a, dtype=torch.float32).bfloat16().float().numpy()


CONV_CASES = [
    # n, h, w, cin, cout, k, stride
    (2, 32, 32, 3, 16, 5, 2),      # thin input: scalar gather (c1)
    (3, 16, 16, 8, 24, 5, 2),      # vector gather, ragged N
    (2, 8, 8, 40, 104, 5, 2),
    (2, 8, 8, 200, 400, 5, 2),     # c2-like channels, BN=208 tile
    (4, 4, 4, 16, 8, 1, 1),        # 1x1 (VAE c5/c6)
    (2, 16, 16, 4, 64, 4, 2),      # pix2pix k4 s2
    (5, 1, 1, 24, 136, 1, 1),      # dense as 1x1 conv on 1x1 images (fc1)
    (2, 7, 9, 12, 20, 5, 2),       # odd spatial sizes
    (2, 16, 16, 3, 100, 5, 2),     # dc4 geometry: big side c=3 (scalar), small side 100 -> stride 104
    (3, 7, 9, 3, 24, 5, 2),        # thin big side, odd sizes: fused-class backward-data with ragged parity classes
    (2, 32, 32, 1, 40, 5, 2),      # one channel (mnist): fused-class backward-data, row-tiled
    (3, 32, 32, 3, 200, 5, 2),     # c1 itself: thin-input forward kernel (compact K = 75 -> 96), 16 x 16 output, N = 200
    (2, 64, 64, 3, 64, 5, 2),      # VAE / autoencoder c1: thin-input forward, 4 output rows per workgroup
    (2, 12, 20, 2, 232, 3, 1),     # thin input, stride 1, two column tiles, ragged last row tile
    (2, 64, 64, 1, 128, 4, 2),     # pix2pix d8 (deconv 128 -> 1): one-channel big side, 128-column LDS-DMA tile
    (3, 32, 32, 64, 128, 4, 2),    # pix2pix e2 / m2: 128-column LDS-DMA tile, 64-channel input
    (2, 16, 16, 128, 256, 4, 2),   # 256 columns = two 128-column tiles
    (1, 256, 256, 1, 128, 4, 2),   # pix2pix d8 at full width: fused-class backward-data with COLUMN tiles (128 anchors x 128 channels)
    (80, 16, 16, 8, 400, 5, 2),    # 400 columns on an under-filled grid: 128 x 112 tiles (four column tiles, the last one 64 wide);
                                   # the smaller 400-column cases above take the 64 x 112 tile
    # GEMM + col2im backward-data ((tap, channel) pairs <= 64 columns; the k4 / one-channel cases above take it too)
    (3, 7, 9, 1, 24, 5, 2),        # 25 columns (two MFMA column tiles), odd sizes: ragged parity classes and halo
    (2, 9, 11, 2, 16, 4, 2),       # 32 columns, odd sizes
    (3, 30, 30, 4, 72, 3, 2),      # k3: 36 columns (three column tiles), K = 72 padded to 96
    (2, 70, 40, 1, 32, 5, 2),      # several row tiles per image (35 anchor rows)
    # thin input, 32 .. 128 output channels on large images: thin-input forward kernel with 64-column tiles
    (2, 128, 128, 3, 64, 4, 2),    # pix2pix e1 geometry at half size: two output rows per workgroup
    (1, 256, 256, 4, 64, 4, 2),    # pix2pix m1: one 128-pixel output row per workgroup
    (1, 130, 250, 1, 128, 5, 2),   # two column tiles, ragged rows (125 pixels), one input channel
    (2, 64, 128, 2, 40, 3, 1),     # stride 1, N = 40 inside one tile, K = 18 padded to 32
    (66, 64, 64, 3, 64, 5, 2),     # VAE c1 (32 x 32 outputs, four output rows per workgroup) at a batch past the 64k-pixel threshold
    # 256-column filter-gradient tiles (N % 256 == 0) in each loader mode
    (16, 8, 8, 64, 256, 5, 2),     # whole-image steps (4 x 4 outputs)
    (2, 32, 32, 64, 512, 4, 2),    # rectangle steps (16 x 16 outputs), two column tiles
    (2, 14, 18, 32, 256, 3, 1),    # generic loader: odd sizes
    # one output channel: the wave-per-pixel forward kernel (M >= 1024)
    (16, 16, 16, 64, 1, 4, 2),     # pix2pix m5 geometry at 64 channels
    (5, 15, 17, 24, 1, 3, 1),      # stride 1, odd sizes, C / 8 = 3 chunks per tap
]


def make_conv(K, dtype, n, h, w, cin, cout, k, s, dev, compact=False):
    oh, pt, _ = T.same_pad(h, k, s)
    ow, pl, _ = T.same_pad(w, k, s)
    big = K.Act(n, h, w, cin, dtype, dev, cs=(cin if compact else None))
    small = K.Act(n, oh, ow, cout, dtype, dev)
    conv = K.Conv(big, small, k, k, s, pt, pl)
    return big, small, conv


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd_bwd(case, dtype):
    K = pkg('kernels')
    n, h, w, cin, cout, k, s = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(hash(case) % 1000)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    Wt = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    big, small, conv = make_conv(K, dtype, n, h, w, cin, cout, k, s, dev)
    dy = rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32)
    if dtype == 1:
        x, Wt, dy = bf16_round(x), bf16_round(Wt), bf16_round(dy)
    wd = torch.tensor(Wt, device=dev)
    bd = torch.tensor(b, device=dev)
    conv.pack(wd)
    # ---- forward with bias + lrelu
    big.set(x)
    conv.fwd(big.ptr(), small.ptr(), n, K.epilogue(bias=bd, act=K.ACT_LRELU, leak=0.2))
    ref = T.lrelu(T.conv2d(x.astype(np.float64), Wt.astype(np.float64), s) + b)
    assert relerr(small.get(), ref) < TOL[dtype]
    # padding channels must stay zero
    full = small.buf.float().reshape(n, small.h, small.w, small.cs).cpu().numpy()
    assert np.all(full[..., cout:] == 0)
    # ---- backward data with lrelu mask taken from x
    small.set(dy)
    mask_src = big.like().set(x)
    out = big.like()
    conv.bwd_data(small.ptr(), out.ptr(), n, K.epilogue(mask_mode=K.MASK_LRELU, mask_src=mask_src.ptr(), leak=0.2))
    ref = T.conv2d_backprop_input(x.shape, Wt.astype(np.float64), dy.astype(np.float64), s) * T.lrelu_grad_mask(x.astype(np.float64))
    assert relerr(out.get(), ref) < TOL[dtype]
    # ---- backward filter (accumulating: beta = 1 on a pre-filled gradient)
    dw = torch.full((k, k, cin, cout), 0.5, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n, beta=1.0)
    ref = T.conv2d_backprop_filter(x.astype(np.float64), Wt.shape, dy.astype(np.float64), s) + 0.5
    assert relerr(dw.cpu().numpy(), ref) < TOL[dtype]


VALID_CASES = [
    # the gen-2 VALID-padded k5 s2 stack (hem/models/paper_cgan.py:221-224: 65 -> 31 -> 14 -> 5 -> 1) and its mirror
    # (deconv2d with explicit output_shape, :237-241: the transpose of the same geometry; 14 -> 5 leaves big row 13 untouched)
    (2, 65, 65, 3, 64, 5, 2),
    (2, 31, 31, 64, 128, 5, 2),
    (3, 14, 14, 128, 256, 5, 2),
    (3, 5, 5, 256, 512, 5, 2),
    (2, 9, 12, 8, 24, 3, 1),       # VALID stride 1
    (2, 10, 7, 16, 40, 4, 2),      # k4 s2 with a remainder column that no window reads
]


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', VALID_CASES)
def test_conv_valid_padding_all_forms(case, dtype):
    """padding='VALID' (out = ceil((in - k + 1) / stride), no pad) through the same three GEMM forms:
    forward, backward-data (= conv2d_transpose VALID with an explicit output_shape) and backward-filter."""
    K = pkg('kernels')
    n, h, w, cin, cout, k, s = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(sum(case))
    oh, ow = -(-(h - k + 1) // s), -(-(w - k + 1) // s)
    big = K.Act(n, h, w, cin, dtype, dev)
    small = K.Act(n, oh, ow, cout, dtype, dev)
    conv = K.Conv(big, small, k, k, s, 0, 0)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    Wt = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32)
    dy = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
    if dtype == 1:
        x, Wt, dy = bf16_round(x), bf16_round(Wt), bf16_round(dy)
    conv.pack(torch.tensor(Wt, device=dev))
    big.set(x)
    conv.fwd(big.ptr(), small.ptr(), n, K.epilogue(bias=torch.tensor(b, device=dev), act=K.ACT_RELU))
    ref = np.maximum(T.conv2d(x.astype(np.float64), Wt.astype(np.float64), s, 'VALID') + b, 0)
    assert ref.shape == (n, oh, ow, cout)
    assert relerr(small.get(), ref) < TOL[dtype]
    small.set(dy)
    out = big.like()
    bb = rng.standard_normal(cin).astype(np.float32)
    conv.bwd_data(small.ptr(), out.ptr(), n, K.epilogue(bias=torch.tensor(bb, device=dev)))
    ref = T.conv2d_backprop_input(x.shape, Wt.astype(np.float64), dy.astype(np.float64), s, 'VALID') + bb
    assert relerr(out.get(), ref) < TOL[dtype]
    dw = torch.zeros(k, k, cin, cout, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n)
    ref = T.conv2d_backprop_filter(x.astype(np.float64), Wt.shape, dy.astype(np.float64), s, 'VALID')
    assert relerr(dw.cpu().numpy(), ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(5, 16, 16, 200, 400, 5, 2), (9, 8, 8, 400, 200, 5, 2), (3, 9, 7, 16, 200, 4, 2)])
def test_conv_lds_dma_variant(case, dtype, monkeypatch):
    """The LDS-DMA tiles (normally chosen for large M only) forced on small, ragged problems, in every wave layout."""
    for mode in ('4', '3', '0'):                 # 256-row tile, 192-row tile, register-staged kernel
        monkeypatch.setenv('TDG_DMA', mode)
        test_conv_fwd_bwd(case, dtype)
    monkeypatch.setenv('TDG_DMA', '3')
    for nw in ('8',):                            # the 192-row tile with every wave loading and computing (default: wave-specialised)
        monkeypatch.setenv('TDG_DMA_NW', nw)
        test_conv_fwd_bwd(case, dtype)
    monkeypatch.delenv('TDG_DMA')
    monkeypatch.delenv('TDG_DMA_NW')
    monkeypatch.setenv('TDG_PATCH', '2')         # the patch-resident kernel (whole-image row tiles) where its plan applies
    test_conv_fwd_bwd(case, dtype)


@pytest.mark.parametrize('case', [(7, 16, 16, 200, 400, 5, 2), (25, 8, 8, 400, 800, 5, 2), (13, 8, 8, 200, 200, 5, 2), (6, 16, 16, 40, 200, 5, 2),
                                  (5, 8, 8, 200, 400, 3, 1), (4, 16, 16, 400, 200, 4, 2)])
def test_conv_patch_resident_kernel(case, monkeypatch):
    """igemm_fwd_patch_kernel forced onto small problems: ragged last row tile (images % images-per-tile != 0), every parity
    class of the backward-data GEMM, a stride-1 conv, 4x4 filters, one and many K slices; the dispatch is asserted."""
    K = pkg('kernels')
    monkeypatch.setenv('TDG_PATCH', '2')
    test_conv_fwd_bwd(case, 1)
    monkeypatch.setenv('TDG_PATCH_BM', '128')     # the 128-row tile (two images of 8 x 8 outputs, eight of 4 x 4)
    test_conv_fwd_bwd(case, 1)
    monkeypatch.delenv('TDG_PATCH_BM')
    n, h, w, cin, cout, k, s = case
    dev = torch.device('cuda:0')
    oh, pt, _ = T.same_pad(h, k, s)
    ow, pl, _ = T.same_pad(w, k, s)
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
    conv = K.Conv(big, small, k, k, s, pt, pl)
    conv.pack(torch.zeros(k, k, cin, cout, device=dev))
    conv.fwd(big.ptr(), small.ptr(), n)
    assert 'igemm_fwd_patch_kernel' in pkg('_lib').load().tdg_last_kernel().decode()


def _wgrad_random_cases(n_cases=28, seed=20261005):
    """Seeded random geometries inside the patch-resident filter gradient's plan (whole images per 64-row step, 8-channel
    slices, 208-column tiles): output grids whose size divides 64, stride 1 / 2, 2x2 .. 5x5 filters, 1 - 13 slices, ragged
    column tiles, image counts that leave ragged last steps and single-step splits."""
    rng = np.random.default_rng(seed)
    grids = [(1, 1), (2, 2), (4, 4), (8, 8), (2, 4), (4, 2), (8, 4), (4, 8), (2, 8), (1, 4), (16, 4), (4, 16)]
    out = []
    while len(out) < n_cases:
        oh, ow = grids[rng.integers(len(grids))]
        s = int(rng.integers(1, 3))
        k = int(rng.integers(max(2, s), 6))
        cin = int(rng.choice([8, 16, 24, 40, 64, 104]))
        cout = int(rng.choice([168, 184, 200, 208, 328, 400, 416]))
        n = int(rng.integers(1, 3 * max(1, 64 // (oh * ow)) + 3))
        if k * k * cin < 128 or k * k > 32:
            continue                                          # (below the LDS-DMA kernels' K threshold / the tap table)
        out.append((n, oh * s, ow * s, cin, cout, k, s))
    return out


@pytest.mark.parametrize('case', _wgrad_random_cases())
def test_wgrad_patch_random_geometries(case):
    """Each case must (a) take igemm_wgrad_patch_kernel or, where its plan refuses, a slab kernel -- never fault -- and (b)
    match the float64 oracle; two launches are bit-equal."""
    K = pkg('kernels')
    n, h, w, cin, cout, k, s = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(hash(case) % 100000)
    x = bf16_round(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    big, small, conv = make_conv(K, 1, n, h, w, cin, cout, k, s, dev)
    dy = bf16_round(rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32))
    big.set(x)
    small.set(dy)
    ref = T.conv2d_backprop_filter(x.astype(np.float64), (k, k, cin, cout), dy.astype(np.float64), s)
    dw = [torch.zeros(k, k, cin, cout, device=dev) for _ in range(2)]
    for d in dw:
        conv.bwd_filter(big.ptr(), small.ptr(), d, n)
    kern = pkg('_lib').load().tdg_last_kernel().decode()
    assert relerr(dw[0].cpu().numpy(), ref) < TOL[1], (case, kern)
    assert torch.equal(dw[0], dw[1]), (case, kern)
    if 64 % (small.h * small.w) == 0 and cin % 8 == 0 and pkg('_lib') and 'igemm_wgrad_dma' in kern:
        print('note: slab kernel on', case)


BLOCK_PATCH_CASES = [
    # n, h, w, cin, cout, k, stride, forward dispatch expected
    (2, 64, 64, 64, 128, 4, 2, 'igemm_fwd_bp_kernel<bf16,256,128>'),      # pix2pix e2 at quarter size: 32 x 32 outputs = four 16 x 16 blocks per image; bwd-data: 64 columns
    (3, 32, 32, 128, 256, 4, 2, 'igemm_fwd_bp_kernel<bf16,256,128>'),     # e3: one block per image (halo'd whole-image patch), two column tiles
    (5, 16, 16, 256, 512, 4, 2, 'igemm_fwd_patch_kernel<bf16,256,128>'),  # e4 at 8 x 8 outputs: WHOLE-image tiles (four images), ragged last tile; bwd-data: blocks
    (2, 64, 64, 64, 128, 5, 2, 'igemm_fwd_patch_kernel<bf16,256,128>'),   # VAE c2 geometry (5 x 5, stride 2): tap groups of 9 / 6 / 6 / 4, halo 1
    (2, 32, 32, 64, 128, 3, 1, 'igemm_fwd_patch_kernel<bf16,256,128>'),   # stride 1: one tap group of nine, blocks with a halo on every side
    (2, 128, 128, 64, 64, 4, 2, 'igemm_fwd_bp_kernel<bf16,256,64>'),      # 64-column tiles both ways (d7-like widths)
    (1, 256, 256, 64, 128, 4, 2, 'igemm_fwd_bp_kernel<bf16,256,128>'),    # e2 at full size: 64 blocks per image, every border block
    (2, 48, 32, 64, 128, 4, 2, None),          # 24 x 16 outputs: no 16 x 16 blocks -> the slab kernel (plan refused)
]


@pytest.mark.parametrize('case', BLOCK_PATCH_CASES)
def test_conv_block_patch_kernel(case, monkeypatch):
    """igemm_fwd_bp_kernel<128 | 64> (every wave loads and multiplies; 4x4 / stride-2 filters and their backward-data classes on
    16 x 16-anchor blocks) and igemm_fwd_patch_kernel<256, 128 | 64> (32-channel K slices, 28 KiB patches: the other geometries): row
    tiles of one 16 x 16 block of an image with a halo'd patch, or of four whole images, forced onto small problems; forward, backward-data and filter gradient
    against the oracle (the filter gradient runs on the 4-chunk K order of the packed operand), dispatch asserted, a second
    launch bit-equal."""
    K = pkg('kernels')
    monkeypatch.setenv('TDG_PATCH', '2')
    monkeypatch.setenv('TDG_BLOCKPATCH', '1')    # (opt-in: on pix2pix / VAE the slab kernel is the faster one)
    n, h, w, cin, cout, k, s, want = case
    test_conv_fwd_bwd((n, h, w, cin, cout, k, s), 1)
    dev = torch.device('cuda:0')
    oh, pt, _ = T.same_pad(h, k, s)
    ow, pl, _ = T.same_pad(w, k, s)
    big, small = K.Act(n, h, w, cin, K.BF16, dev), K.Act(n, oh, ow, cout, K.BF16, dev)
    conv = K.Conv(big, small, k, k, s, pt, pl)
    rng = np.random.default_rng(3)
    conv.pack(torch.tensor(rng.standard_normal((k, k, cin, cout)).astype(np.float32) * 0.05, device=dev))
    big.set(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    conv.fwd(big.ptr(), small.ptr(), n)
    kern = pkg('_lib').load().tdg_last_kernel().decode()
    if want is None:
        assert 'igemm_fwd_patch_kernel' not in kern and 'igemm_fwd_bp_kernel' not in kern, kern
    else:
        assert want in kern, kern
    first = small.get().copy()
    small.set(np.zeros_like(first))
    conv.fwd(big.ptr(), small.ptr(), n)
    assert np.array_equal(small.get(), first)


WGRAD_PATCH_CASES = [
    (7, 16, 16, 200, 400, 5, 2),      # c2 geometry: one image per 64-row step, 25 slices, 20 K tiles of 32 (slice, tap) units
    (25, 8, 8, 400, 800, 5, 2),       # c3: four images per step, ragged last step (25 images), four column tiles
    (13, 8, 8, 200, 200, 5, 2),       # one column tile, ragged
    (6, 16, 16, 40, 200, 5, 2),       # 5 slices: the last K tile is mostly padding units
    (37, 4, 4, 400, 800, 5, 2),       # dc1 geometry: 2 x 2 outputs, 16 images per step; most taps fall outside the image
    (5, 8, 8, 200, 400, 3, 1),        # stride 1, 9 taps: a K tile spans 5 slices (80-byte patch pixels)
    (4, 16, 16, 104, 200, 5, 2),      # dc3: 13 slices (100 channels in a stride of 104)
    (9, 16, 16, 48, 400, 4, 2),       # 4 x 4 filters (16 taps: a tile is exactly two slices)
]


@pytest.mark.parametrize('nw', ['8', '4'])
@pytest.mark.parametrize('case', WGRAD_PATCH_CASES)
def test_wgrad_patch_kernel(case, nw, monkeypatch):
    """igemm_wgrad_patch_kernel (the gathered operand of the filter gradient resident in LDS as a patch of source pixels) in
    both wave layouts against the float64 oracle and, bit for bit, against a second launch; the dispatch is asserted.
    beta = 1 on a pre-filled gradient; the slab kernel (TDG_WPATCH=0) must agree to summation-order rounding."""
    K = pkg('kernels')
    monkeypatch.setenv('TDG_WPATCH_NW', nw)
    n, h, w, cin, cout, k, s = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(11)
    x = bf16_round(rng.standard_normal((n, h, w, cin)).astype(np.float32))
    big, small, conv = make_conv(K, 1, n, h, w, cin, cout, k, s, dev)
    dy = bf16_round(rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32))
    big.set(x)
    small.set(dy)
    ref = T.conv2d_backprop_filter(x.astype(np.float64), (k, k, cin, cout), dy.astype(np.float64), s) + 0.5
    outs = []
    for rep in range(2):
        dw = torch.full((k, k, cin, cout), 0.5, device=dev)
        conv.bwd_filter(big.ptr(), small.ptr(), dw, n, beta=1.0)
        assert 'igemm_wgrad_patch_kernel<bf16,256,208,%s' % nw in pkg('_lib').load().tdg_last_kernel().decode()
        outs.append(dw.cpu().numpy())
    assert relerr(outs[0], ref) < TOL[1]
    assert np.array_equal(outs[0], outs[1])
    monkeypatch.setenv('TDG_WPATCH', '0')
    dw = torch.full((k, k, cin, cout), 0.5, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n, beta=1.0)
    assert 'igemm_wgrad_dma_kernel' in pkg('_lib').load().tdg_last_kernel().decode()
    assert relerr(dw.cpu().numpy(), outs[0]) < 1e-5


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(2, 8, 8, 200, 400, 5, 2), (2, 16, 16, 128, 256, 4, 2), (3, 4, 4, 512, 512, 4, 2)])
def test_conv_split_k_small_m(case, dtype, monkeypatch):
    """Small-M forward-type GEMMs (pix2pix's bottleneck layers: a handful of workgroups, each 32 - 128 K steps deep) are cut
    along K into f32 partial tiles plus a finishing kernel: default split count, forced ragged counts, and never."""
    for ks in (None, '3', '7', '1'):
        if ks is None:
            monkeypatch.delenv('TDG_KSPLIT', raising=False)
        else:
            monkeypatch.setenv('TDG_KSPLIT', ks)
        test_conv_fwd_bwd(case, dtype)


@pytest.mark.parametrize('dtype', [0, 1])
def test_conv_compact_thin_input_takes_the_scalar_gather(dtype):
    """A caller-provided compact layout (channel stride 3) cannot use 16-byte gathers."""
    K = pkg('kernels')
    dev = torch.device('cuda:0')
    n, h, w, cin, cout, k, s = 3, 16, 16, 3, 24, 5, 2
    rng = np.random.default_rng(9)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    Wt = (rng.standard_normal((k, k, cin, cout)) / 8).astype(np.float32)
    if dtype == 1:
        x, Wt = bf16_round(x), bf16_round(Wt)
    big, small, conv = make_conv(K, dtype, n, h, w, cin, cout, k, s, dev, compact=True)
    assert big.cs == 3
    conv.pack(torch.tensor(Wt, device=dev))
    big.set(x)
    conv.fwd(big.ptr(), small.ptr(), n)
    assert relerr(small.get(), T.conv2d(x.astype(np.float64), Wt.astype(np.float64), s)) < TOL[dtype]
    dy = rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32)
    if dtype == 1:
        dy = bf16_round(dy)
    small.set(dy)
    dw = torch.zeros(k, k, cin, cout, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n)
    assert relerr(dw.cpu().numpy(), T.conv2d_backprop_filter(x.astype(np.float64), Wt.shape, dy.astype(np.float64), s)) < TOL[dtype]


@pytest.mark.parametrize('dtype', [0, 1])
def test_conv_subbatch_and_classes(dtype):
    """n_images < capacity with a pointer offset (the slot scheme of the D passes)."""
    K = pkg('kernels')
    dev = torch.device('cuda:0')
    n, h, w, cin, cout, k, s = 6, 8, 8, 16, 32, 5, 2
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    Wt = (rng.standard_normal((k, k, cin, cout)) / 20).astype(np.float32)
    if dtype == 1:
        x, Wt = bf16_round(x), bf16_round(Wt)
    big, small, conv = make_conv(K, dtype, n, h, w, cin, cout, k, s, dev)
    conv.pack(torch.tensor(Wt, device=dev))
    big.set(x)
    conv.fwd(big.ptr(2), small.ptr(2), 3)
    got = small.get()
    ref = T.conv2d(x[2:5].astype(np.float64), Wt.astype(np.float64), s)
    assert relerr(got[2:5], ref) < TOL[dtype]
    assert np.all(got[:2] == 0) and np.all(got[5:] == 0)


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('shape', [(512, 200), (96, 3), (1000, 1), (64, 800)])
def test_batch_norm(shape, dtype):
    K = pkg('kernels')
    dev = torch.device('cuda:0')
    rows, c = shape
    rng = np.random.default_rng(3)
    u = (rng.standard_normal((rows, c)) * 2 + 0.7).astype(np.float32)
    beta = rng.standard_normal(c).astype(np.float32)
    dh = rng.standard_normal((rows, c)).astype(np.float32)
    if dtype == 1:
        u, dh = bf16_round(u), bf16_round(dh)
    ws = K.Workspace(dev)
    ua = K.Act(rows, 1, 1, c, dtype, dev).set(u)
    pre, hh, du = ua.like(), ua.like(), ua.like()
    stats = torch.zeros(2 * c, device=dev)
    bd = torch.tensor(beta, device=dev)
    K.bn_fwd(ws, ua, c, bd, K.ACT_RELU, pre, hh, stats)
    ref_pre, cache = T.batch_norm_train(u.astype(np.float64), beta.astype(np.float64))
    assert relerr(pre.get().reshape(rows, c), ref_pre) < TOL[dtype]
    assert relerr(hh.get().reshape(rows, c), T.relu(ref_pre)) < TOL[dtype]
    dha = K.Act(rows, 1, 1, c, dtype, dev).set(dh)
    dbeta = torch.zeros(c, device=dev)
    # backward consumes the device's own `pre` (what the product path does)
    K.bn_bwd(ws, dha, pre, c, bd, stats, K.ACT_RELU, du, dbeta)
    pre_dev = pre.get().reshape(rows, c).astype(np.float64)
    dpre = dh * (pre_dev > 0)
    xhat = pre_dev - beta
    rstd = cache[1]
    ref_du = rstd * (dpre - dpre.mean(0) - xhat * (dpre * xhat).mean(0))
    assert relerr(du.get().reshape(rows, c), ref_du) < 5 * TOL[dtype]
    assert relerr(dbeta.cpu().numpy(), dpre.sum(0)) < 5 * TOL[dtype]
    # the same pass with the bias gradient of the conv in front (column sums of the stored du, accumulated onto 0.25):
    # identical du, and the sums a separate pass over du finds (analytically zero: compared on the scale of sum |du|)
    du2, dbeta2 = ua.like(), torch.zeros(c, device=dev)
    dbias = torch.full((c,), 0.25, device=dev)
    K.bn_bwd(ws, dha, pre, c, bd, stats, K.ACT_RELU, du2, dbeta2, dbias=dbias, dbias_acc=1.0)
    got = du2.get().reshape(rows, c)
    assert np.array_equal(got, du.get().reshape(rows, c))
    assert np.array_equal(dbeta2.cpu().numpy(), dbeta.cpu().numpy())
    want = got.astype(np.float64).sum(0)
    scale = np.abs(got).sum(0).max() + 1e-30
    assert np.abs(dbias.cpu().numpy() - 0.25 - want).max() / scale < 1e-6


@pytest.mark.parametrize('dtype', [0, 1])
def test_row_ops_and_reductions(dtype):
    K = pkg('kernels')
    L = pkg('_lib')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(5)
    rows, cols = 48, 1280
    x = rng.standard_normal((rows, cols)).astype(np.float32)
    if dtype == 1:
        x = bf16_round(x)
    w = rng.standard_normal(cols).astype(np.float32)
    b = np.array([0.3], np.float32)
    xd = torch.tensor(x, device=dev).to(K.TORCH_DTYPE[dtype])
    wd, bd = torch.tensor(w, device=dev), torch.tensor(b, device=dev)
    out = torch.zeros(rows, device=dev)
    L.call('tdg_rowdot', dtype, K.ptr(xd), rows, cols, K.ptr(wd), K.ptr(bd), K.ACT_NONE, K.ptr(out), K.stream())
    assert relerr(out.cpu().numpy(), x.astype(np.float64) @ w + 0.3) < TOL[dtype]
    # rowouter with lrelu mask
    dout = torch.tensor(rng.standard_normal(rows).astype(np.float32), device=dev)
    dx = torch.zeros(rows, cols, device=dev, dtype=K.TORCH_DTYPE[dtype])
    L.call('tdg_rowouter', dtype, K.ptr(dout), K.ptr(wd), rows, cols, K.MASK_LRELU, 0.2, K.ptr(xd), K.ptr(dx), K.stream())
    ref = np.outer(dout.cpu().numpy(), w) * np.where(x > 0, 1.0, 0.2)
    assert relerr(dx.float().cpu().numpy(), ref) < TOL[dtype]
    # weighted column sum
    ws = K.Workspace(dev)
    dw = torch.ones(cols, device=dev)
    K.colsum_weighted(ws, dtype, K.ptr(xd), rows, cols, cols, dout, dw, beta=1.0)
    assert relerr(dw.cpu().numpy(), 1.0 + dout.cpu().numpy().astype(np.float64) @ x) < TOL[dtype]
    # sumsq + gp scalars
    acc = torch.zeros(1, device=dev)
    K.sumsq(ws, dtype, K.ptr(xd), rows * cols, acc)
    ss = float((x.astype(np.float64) ** 2).sum())
    assert abs(acc.item() - ss) / ss < 1e-5
    scal = torch.zeros(2, device=dev)
    L.call('tdg_gp_scalars', K.ptr(acc), 10.0, K.ptr(scal), K.stream())
    s = np.sqrt(ss)
    assert np.allclose(scal.cpu().numpy(), [(s - 1) ** 2, 20 * (s - 1) / s], rtol=1e-5)
    # interpolation
    g = rng.standard_normal((rows, cols)).astype(np.float32)
    al = rng.uniform(0, 1, rows).astype(np.float32)
    gd = torch.tensor(g, device=dev).to(K.TORCH_DTYPE[dtype])
    ad = torch.tensor(al, device=dev)
    xh = torch.zeros_like(xd)
    L.call('tdg_gp_interp', dtype, K.ptr(xd), K.ptr(gd), K.ptr(ad), rows, cols, K.ptr(xh), K.stream())
    gq = gd.float().cpu().numpy()
    assert relerr(xh.float().cpu().numpy(), x + al[:, None] * (gq - x)) < TOL[dtype]
    # bias grad on a thin tensor
    ya = K.Act(rows, 4, 4, 3, dtype, dev)
    yv = rng.standard_normal((rows, 4, 4, 3)).astype(np.float32)
    if dtype == 1:
        yv = bf16_round(yv)
    ya.set(yv)
    db = torch.zeros(3, device=dev)
    K.bias_grad(ws, ya, 3, db)
    assert relerr(db.cpu().numpy(), yv.astype(np.float64).reshape(-1, 3).sum(0)) < TOL[dtype]


def test_optimizers_match_tf_rules():
    K = pkg('kernels')
    L = pkg('_lib')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(11)
    n = 1000
    p0 = rng.standard_normal(n).astype(np.float32)
    grads = [rng.standard_normal(n).astype(np.float32) for _ in range(3)]
    # Adam
    ref = {'p': p0.copy().astype(np.float64)}
    opt = T.Adam(1e-2, 0.5, 0.9)
    p = torch.tensor(p0, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev)
    for t, g in enumerate(grads, start=1):
        opt.apply(ref, {'p': g.astype(np.float64)})
        lr_t = 1e-2 * np.sqrt(1 - 0.9 ** t) / (1 - 0.5 ** t)
        L.call('tdg_adam_step', K.ptr(p), K.ptr(torch.tensor(g, device=dev)), K.ptr(m), K.ptr(v), n, lr_t, 0.5, 0.9, 1e-8, 1.0, K.stream())
    assert relerr(p.cpu().numpy(), ref['p']) < 1e-5
    # RMSProp with the reference's defaults (momentum 0.01, rms slot starts at 1)
    ref = {'p': p0.copy().astype(np.float64)}
    opt = T.RMSProp(1e-3, 0.9, 0.01)
    p = torch.tensor(p0, device=dev); rms = torch.ones(n, device=dev); mom = torch.zeros(n, device=dev)
    for g in grads:
        opt.apply(ref, {'p': g.astype(np.float64)})
        L.call('tdg_rmsprop_step', K.ptr(p), K.ptr(torch.tensor(g, device=dev)), K.ptr(rms), K.ptr(mom), n, 1e-3, 0.9, 0.01, 1e-10, 1.0, K.stream())
    assert relerr(p.cpu().numpy(), ref['p']) < 1e-5


@pytest.mark.parametrize('name', ['rmsprop_centered', 'adagrad', 'padagrad', 'adadelta', 'ftrl', 'sgd', 'momentum'])
def test_optimizer_branches_match_oracle(name):
    """util.py:150-183: every optimizer branch the CLI exposes, through util.init_optimizer on a flat parameter store,
    against the numpy restatement in float64 over 5 steps (with a gradient scale, as the N-tower average uses)."""
    E = pkg('engine')
    U = pkg('util')
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(11)
    n = 4096
    opt_name = 'rmsprop' if name == 'rmsprop_centered' else name
    args = SimpleNamespace(optimizer=opt_name, lr=1e-2, decay=0.9, momentum=0.3, centered=name == 'rmsprop_centered',
                           beta1=0.5, beta2=0.9)
    store = E.ParamStore(dev)
    store.declare('p', (n,))
    store.allocate()
    p0 = rng.standard_normal(n).astype(np.float32)
    store.load({'p': p0})
    opt = U.init_optimizer(args, store)
    ref_opt = T.init_optimizer(args)
    ref = {'p': p0.astype(np.float64)}
    for _ in range(5):
        g = rng.standard_normal(n).astype(np.float32)
        store.grad('p').copy_(torch.tensor(g) * 2.0)
        opt.step(grad_scale=0.5)
        ref_opt.apply(ref, {'p': g.astype(np.float64)})
    got = store.state_dict()['p']
    assert np.isfinite(got).all()
    assert relerr(got, ref['p']) < 1e-5
    assert set(opt.state_tensors()) == {'rmsprop_centered': {'rms', 'mom', 'mg'}, 'adagrad': {'acc'}, 'padagrad': {'acc'},
                                        'adadelta': {'acc', 'acc_update'}, 'ftrl': {'acc', 'linear'}, 'sgd': {'acc'},
                                        'momentum': {'acc'}}[name]


def test_ticketed_kernels_keep_to_one_stream():
    """Kernels that hand their last block a ticket from a device-global slot (the *_dev RNG draws, Adam, sumsq) must not overlap
    on the device: the library remembers, per device, the first stream that launched one and refuses any other with TDG_EINVAL."""
    _lib, K = pkg('_lib'), pkg('kernels')
    dev = torch.device('cuda:0')
    out = torch.zeros(1024, device=dev)
    draws = torch.zeros(1, dtype=torch.int32, device=dev)
    cur = torch.cuda.current_stream(dev).cuda_stream
    _lib.call('tdg_random_uniform_f32_dev', 1, 2, K.ptr(draws), 1024, K.ptr(out), cur)
    torch.cuda.synchronize()
    assert int(draws.item()) == 1 and float(out.min()) >= 0.0 and float(out.max()) < 1.0
    other = torch.cuda.Stream(device=dev)
    with pytest.raises(_lib.TdgError, match='ONE stream per device'):
        _lib.call('tdg_random_uniform_f32_dev', 1, 2, K.ptr(draws), 1024, K.ptr(out), other.cuda_stream)
    _lib.call('tdg_random_uniform_f32_dev', 1, 2, K.ptr(draws), 1024, K.ptr(out), cur)          # the first stream still works
    torch.cuda.synchronize()
    assert int(draws.item()) == 2


def test_rng_statistics():
    K = pkg('kernels')
    L = pkg('_lib')
    dev = torch.device('cuda:0')
    n = 1 << 20
    z = torch.zeros(n, device=dev)
    L.call('tdg_random_normal', 0, 1234, 1, 0, n, K.ptr(z), K.stream())
    assert abs(z.mean().item()) < 5e-3 and abs(z.std().item() - 1) < 5e-3
    u = torch.zeros(n, device=dev)
    L.call('tdg_random_uniform_f32', 1234, 2, 0, n, K.ptr(u), K.stream())
    assert 0 <= u.min().item() and u.max().item() < 1 and abs(u.mean().item() - 0.5) < 2e-3
    z2 = torch.zeros(n, device=dev)
    L.call('tdg_random_normal', 0, 1234, 1, 0, n, K.ptr(z2), K.stream())
    assert torch.equal(z, z2)            # counter-based: same key/counter -> same stream


def test_pack_filters_batch_equals_single_packs_and_timing_records():
    """tdg_pack_filters == tdg_pack_filter_fwd + tdg_pack_filter_bwd per job (bit-identical packed bytes, including the
    fused-class backward form of a thin big side); tdg_timing_* returns one record per conv GEMM kernel launch."""
    K = pkg('kernels')
    dev = torch.device('cuda:0')
    convs, ws = [], []
    for (n, h, w, cin, cout, k, s) in [(2, 16, 16, 200, 400, 5, 2), (2, 32, 32, 3, 200, 5, 2), (2, 8, 8, 40, 104, 5, 2)]:
        big, small, conv = make_conv(K, 1, n, h, w, cin, cout, k, s, dev)
        wt = torch.randn(k, k, cin, cout, device=dev) * 0.05
        conv.pack(wt)
        convs.append((conv, big, small, n))
        ws.append(wt)
    single = [(c.w_fwd.clone(), c.w_bwd.clone()) for c, _, _, _ in convs]
    for c, _, _, _ in convs:
        c.w_fwd.zero_()
        c.w_bwd.zero_()
    K.pack_all(K.make_pack_jobs([c.pack_job(wt) for (c, _, _, _), wt in zip(convs, ws)]))
    torch.cuda.synchronize()
    for (c, _, _, _), (f, b) in zip(convs, single):
        assert torch.equal(c.w_fwd, f) and torch.equal(c.w_bwd, b)
    K.timing_begin()
    conv, big, small, n = convs[0]
    conv.fwd(big.ptr(), small.ptr(), n)
    conv.bwd_data(small.ptr(), big.ptr(), n)
    dw = torch.zeros(5, 5, 200, 400, device=dev)
    conv.bwd_filter(big.ptr(), small.ptr(), dw, n)
    rec = K.timing_end()
    names = [r[0] for r in rec]
    assert len(rec) >= 4 and any('wgrad' in x for x in names) and 'slab_reduce' in names
    gemm = [r for r in rec if r[2] > 0]
    assert all(r[1] > 0 for r in rec) and abs(sum(r[2] for r in gemm) - 3 * conv.flops(n)) < 1e-6 * conv.flops(n)
    assert K.timing_end() == []                  # nothing is recorded once timing is off


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(6, 16, 16, 200, 400, 5, 2, 4), (5, 8, 8, 40, 104, 5, 2, 2), (4, 32, 32, 3, 200, 5, 2, 3),
                                  (6, 16, 16, 64, 256, 4, 2, 2)])      # the last: 256-column LDS-DMA tile
def test_bwd_filter_two_sources(case, dtype):
    """tdg_conv2d_bwd_filter2 (rows of the first n_first images from one tensor, the rest from another) against the
    oracle's conv2d_backprop_filter on the concatenated input, with accumulation into an existing dw."""
    K = pkg('kernels')
    n, h, w, cin, cout, k, s, n_first = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(7)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    big, small, conv = make_conv(K, dtype, n, h, w, cin, cout, k, s, dev)
    dy = rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32)
    dw0 = rng.standard_normal((k, k, cin, cout)).astype(np.float32)
    if dtype == 1:
        x, dy = bf16_round(x), bf16_round(dy)
    first, second = big.like().set(x), big.like()
    second.set(np.concatenate([x[n_first:], np.zeros_like(x[:n_first])]))      # its image 0 is image n_first
    first.set(np.concatenate([x[:n_first], 9.0 * np.ones_like(x[n_first:])]))  # rows past n_first must not be read
    small.set(dy)
    dw = torch.tensor(dw0, device=dev)
    conv.bwd_filter2(first.ptr(), n_first, second.ptr(), small.ptr(), dw, n, beta=0.5)
    ref = 0.5 * dw0 + T.conv2d_backprop_filter(x.astype(np.float64), (k, k, cin, cout), dy.astype(np.float64), s)
    assert relerr(dw.cpu().numpy(), ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(3, 32, 32, 3, 200, 5, 2), (2, 16, 16, 200, 400, 5, 2), (2, 20, 12, 4, 168, 3, 1)])
def test_conv_fwd_with_mask_epilogue(case, dtype):
    """Forward conv whose epilogue multiplies by the (l)relu derivative of a mask tensor laid out like the output: the
    tangent pass of the gradient penalty (t_i = lrelu'(h_i) * conv_i(t_{i-1}), no bias).  Covers the thin-input kernel
    (first case, bf16) and the LDS-DMA kernel."""
    K = pkg('kernels')
    n, h, w, cin, cout, k, s = case
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(11)
    x = rng.standard_normal((n, h, w, cin)).astype(np.float32)
    Wt = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    big, small, conv = make_conv(K, dtype, n, h, w, cin, cout, k, s, dev)
    m = rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32)
    if dtype == 1:
        x, Wt, m = bf16_round(x), bf16_round(Wt), bf16_round(m)
    conv.pack(torch.tensor(Wt, device=dev))
    big.set(x)
    mask = small.like().set(m)
    conv.fwd(big.ptr(), small.ptr(), n, K.epilogue(mask_mode=K.MASK_LRELU, leak=0.2, mask_src=mask.ptr()))
    ref = T.conv2d(x.astype(np.float64), Wt.astype(np.float64), s) * np.where(m > 0, 1.0, 0.2)
    assert relerr(small.get(), ref) < TOL[dtype]


@pytest.mark.parametrize('dtype', [0, 1])
@pytest.mark.parametrize('case', [(2, 32, 32, 3, 104, 5, 2), (2, 16, 16, 200, 400, 5, 2), (3, 14, 10, 1, 64, 4, 2)])
def test_deconv_forward_bias_tanh(case, dtype):
    """conv2d_transpose + bias + tanh in one launch (the generator's image layer: backward-data form with bias and
    activation in the epilogue).  Covers the fused-class kernel (thin big sides, bf16) and the per-class kernels."""
    K = pkg('kernels')
    n, h, w, cin, cout, k, s = case           # big side h x w x cin (the deconv OUTPUT), small side cout channels
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(13)
    big, small, conv = make_conv(K, dtype, n, h, w, cin, cout, k, s, dev)
    y = rng.standard_normal((n, small.h, small.w, cout)).astype(np.float32)
    Wt = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cout)).astype(np.float32)
    b = rng.standard_normal(cin).astype(np.float32)
    if dtype == 1:
        y, Wt = bf16_round(y), bf16_round(Wt)
    conv.pack(torch.tensor(Wt, device=dev))
    small.set(y)
    conv.bwd_data(small.ptr(), big.ptr(), n, K.epilogue(bias=torch.tensor(b, device=dev), act=K.ACT_TANH))
    ref = np.tanh(T.conv2d_transpose(y.astype(np.float64), Wt.astype(np.float64), (n, h, w, cin), s) + b)
    assert relerr(big.get(), ref) < TOL[dtype]
    full = big.buf.float().reshape(n, h, w, big.cs).cpu().numpy()
    assert np.all(full[..., cin:] == 0)        # channel padding stays zero
