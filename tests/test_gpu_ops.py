"""GPU parity of the single ops behind the C-ABI against plain PyTorch fp32 on the CPU."""
import ctypes
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device('cuda:0')


def run_conv(x_nchw, w, scale=None, shift=None, residual=None, stride=1, pad=0, act=0, tile=(0, 0), ksplit=0, kwaves=0, stages=0,
             counters=None, repeat=1, tail=(0, 0), grid_wgs=0):
    """Runs ym_conv2d_fwd on NHWC data; returns NCHW cpu tensor."""
    from yolact_minimal_amd import hip
    dev = _dev()
    b, cin, h, wd = x_nchw.shape
    cout, _, kh, kw = w.shape
    stem = cin == 3
    if stem:
        xin = torch.zeros(b, h, wd, 4)
        xin[..., :3] = x_nchw.permute(0, 2, 3, 1)
    else:
        xin = x_nchw.permute(0, 2, 3, 1)
    xin = xin.contiguous().to(dev)
    cin_pad = 4 if stem else cin
    k_pad = (kh * kw * cin_pad + 31) // 32 * 32
    wp = hip.pack_conv_weight(w.to(dev), cin_pad, k_pad)
    ho, wo = (h + 2 * pad - kh) // stride + 1, (wd + 2 * pad - kw) // stride + 1
    out = torch.full((b, ho, wo, cout), float('nan'), device=dev)
    d = hip.ConvDesc()
    d.inp, d.weight = xin.data_ptr(), wp.data_ptr()
    sc = scale.to(dev) if scale is not None else None
    sh = shift.to(dev) if shift is not None else None
    rs = residual.permute(0, 2, 3, 1).contiguous().to(dev) if residual is not None else None
    d.scale = sc.data_ptr() if sc is not None else None
    d.shift = sh.data_ptr() if sh is not None else None
    d.residual = rs.data_ptr() if rs is not None else None
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, wd, cin_pad, cout, kh, kw
    d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = stride, pad, ho, wo, k_pad, 1
    d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
    d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = ho * wo * cout, cout, act
    d.tile_m, d.tile_n = tile
    d.ksplit = ksplit
    d.kwaves = kwaves
    d.stages = stages
    if counters is not None:
        d.tile_counters = counters.data_ptr()
    d.tail_tiles, d.tail_ksplit = tail
    d.grid_wgs = grid_wgs
    nbytes = hip.conv_workspace_bytes(d)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=dev)
    for _ in range(repeat):
        hip.conv2d_fwd(d, ws)
    torch.cuda.synchronize()
    return out.cpu().permute(0, 3, 1, 2)


def ref_conv(x, w, scale, shift, residual, stride, pad, act):
    y = F.conv2d(x.double(), w.double(), None, stride, pad)
    if scale is not None:
        y = y * scale.double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + shift.double().view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual.double()
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = torch.tanh(y)
    return y.float()


CONV_CASES = [
    # b, cin, h, w, cout, k, stride, pad, act, residual, tile, ksplit
    (1, 64, 17, 17, 64, 1, 1, 0, 1, False, (0, 0), 0),
    (2, 64, 20, 23, 256, 1, 1, 0, 1, True, (128, 128), 1),
    (1, 128, 19, 19, 128, 3, 2, 1, 1, False, (128, 64), 1),
    (1, 256, 9, 9, 96, 3, 1, 1, 2, False, (64, 128), 1),
    (2, 256, 12, 10, 255, 3, 1, 1, 0, False, (64, 64), 3),
    (1, 512, 7, 7, 2048, 1, 2, 0, 0, False, (64, 64), 4),
    (1, 32, 40, 40, 64, 3, 1, 1, 1, True, (0, 0), 0),
    (2, 3, 64, 64, 64, 7, 2, 3, 1, False, (0, 0), 0),       # stem mode
    (1, 3, 33, 47, 64, 7, 2, 3, 1, False, (0, 0), 0),       # stem, odd sizes
    (1, 256, 5, 5, 256, 3, 2, 1, 1, False, (0, 0), 0),      # P7-like, heuristic split-K
]


@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_parity(case):
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, ksplit = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit)
    want = ref_conv(x, wt, scale, shift, res, stride, pad, act)
    assert not torch.isnan(got).any()
    # tolerance: fp32 accumulation-order differences only
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', [
    # b, cin, h, w, cout, k, stride, pad, act, residual, tile, ksplit   (all with the 3-deep LDS ring)
    (1, 64, 17, 17, 64, 1, 1, 0, 1, False, (64, 64), 1),        # 2 K tiles
    (1, 32, 9, 9, 64, 1, 1, 0, 0, False, (64, 64), 1),          # 1 K tile
    (2, 96, 12, 10, 128, 1, 1, 0, 1, True, (64, 64), 1),        # 3 K tiles
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, (64, 64), 6),      # split-K, 12 tiles per slice
    (1, 128, 19, 19, 128, 3, 2, 1, 1, False, (128, 64), 1),
    (1, 256, 9, 9, 96, 3, 1, 1, 2, False, (64, 128), 2),
    (1, 1024, 13, 13, 256, 1, 1, 0, 1, True, (64, 64), 5),      # uneven slices (32 tiles / 5)
])
def test_conv_three_stage_ring_parity(case):
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, ksplit = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit, 0, 3)
    base = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit, 0, 2)
    assert torch.equal(got, base)          # same summation order as the 2-stage kernel -> bit-identical
    torch.testing.assert_close(got, ref_conv(x, wt, scale, shift, res, stride, pad, act), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', [
    # b, cin, h, w, cout, k, stride, pad, act, residual, tile, ksplit, stages
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, (64, 64), 6, 3),
    (1, 1024, 34, 34, 256, 1, 1, 0, 1, False, (64, 64), 4, 2),
    (1, 256, 34, 34, 1024, 1, 1, 0, 1, True, (64, 64), 2, 2),
    (1, 1024, 13, 13, 256, 1, 1, 0, 1, True, (64, 64), 5, 2),       # uneven slices
    (2, 256, 12, 10, 252, 3, 1, 1, 0, False, (64, 128), 3, 2),      # Cout not a multiple of the tile
    (1, 512, 17, 17, 512, 3, 1, 1, 2, False, (128, 128), 8, 2),
    (1, 256, 5, 5, 256, 3, 2, 1, 1, False, (0, 0), 0, 0),           # heuristic split
])
def test_conv_splitk_fused_finish(case):
    """K split with `tile_counters`: the last-arriving workgroup reduces the slices in slice order and runs the epilogue in
    the same launch -> bit-identical to the separate reduce launch, counters back at zero (re-launchable)."""
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, ksplit, stages = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    counters = torch.zeros(4096, dtype=torch.int32, device=_dev())
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit, 0, stages, counters=counters, repeat=3)
    base = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit, 0, stages)
    assert int(counters.abs().sum()) == 0
    assert torch.equal(got, base)
    torch.testing.assert_close(got, ref_conv(x, wt, scale, shift, res, stride, pad, act), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('case', [
    # b, cin, h, w, cout, k, stride, pad, act, residual, tile, tail (tiles, slices), stages
    (2, 256, 34, 34, 256, 3, 1, 1, 1, False, (64, 64), (20, 4), 2),      # 37x4 = 148 tiles, the last 20 in quarters
    (2, 256, 34, 34, 256, 3, 1, 1, 1, True, (64, 64), (148, 3), 3),      # every tile in the tail
    (1, 1024, 34, 34, 256, 1, 1, 0, 1, False, (128, 64), (7, 8), 2),
    (2, 128, 40, 40, 252, 3, 2, 1, 2, False, (64, 128), (5, 6), 2),      # ragged M and N edges live in the tail tiles
    (1, 64, 30, 30, 64, 1, 1, 0, 0, False, (64, 64), (3, 8), 2),         # 2 K tiles: slices clamp to 2
])
def test_conv_tail_split(case):
    """`tail_tiles`/`tail_ksplit`: the last output tiles are computed as K slices by several workgroups and finished by the last
    arriver; the other tiles are untouched by the split (bit-identical to the plain launch)."""
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, tail, stages = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    counters = torch.zeros(4096, dtype=torch.int32, device=_dev())
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, 1, 0, stages, counters=counters, repeat=2, tail=tail)
    base = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, 1, 0, stages)
    assert int(counters.abs().sum()) == 0
    torch.testing.assert_close(got, ref_conv(x, wt, scale, shift, res, stride, pad, act), rtol=1e-4, atol=1e-5)
    tiles_n = -(-cout // tile[1])
    tiles = -(-(b * ho * wo) // tile[0]) * tiles_n
    main_rows = (tiles - tail[0]) // tiles_n * tile[0]        # rows whose every tile is outside the tail
    gm = got.permute(0, 2, 3, 1).reshape(-1, cout)[:main_rows]
    bm = base.permute(0, 2, 3, 1).reshape(-1, cout)[:main_rows]
    assert torch.equal(gm, bm)
    assert not torch.equal(got, base) or tail[1] == 1


@pytest.mark.parametrize('case', [
    # b, cin, h, w, cout, k, stride, pad, act, residual, tile, ksplit
    (1, 64, 17, 17, 64, 1, 1, 0, 1, False, (64, 64), 1),        # 2 K tiles
    (1, 32, 9, 9, 64, 1, 1, 0, 0, False, (64, 64), 1),          # 1 K tile: every prefetch is past the end
    (2, 96, 12, 10, 128, 1, 1, 0, 1, True, (64, 128), 1),       # 3 K tiles
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, (64, 64), 6),      # split-K, 12 tiles per slice, padding taps
    (1, 128, 19, 19, 128, 3, 2, 1, 1, False, (128, 64), 1),     # stride 2
    (1, 256, 9, 9, 96, 3, 1, 1, 2, False, (64, 128), 2),        # ragged N
    (2, 256, 20, 20, 256, 3, 1, 1, 1, True, (128, 128), 1),
    (1, 1024, 13, 13, 256, 1, 1, 0, 1, True, (64, 64), 5),      # uneven slices
])
def test_conv_direct_to_lds_parity(case):
    """stages 22 / 23 / 24: operand tiles DMA'd global -> LDS (`buffer_load ... lds`, XOR-swizzled chunks, ring of 2 / 3 / 4);
    33 / 34: the same with software-pipelined fragment reads across the tile barrier.
    Same MFMA order as the register-staged kernel -> bit-identical output."""
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, ksplit = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    base = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit, 0, 2)
    for stages in (22, 23) + ((24, 33, 34) if tile == (64, 64) else ()):
        got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, ksplit, 0, stages)
        assert torch.equal(got, base), stages
    torch.testing.assert_close(base, ref_conv(x, wt, scale, shift, res, stride, pad, act), rtol=1e-4, atol=1e-5)


PERS_CASES = [
    # b, cin, h, w, cout, k, stride, pad, act, residual, ksplit, tail, grid_wgs (0 = library choice)
    (1, 256, 34, 34, 1024, 1, 1, 0, 1, True, 1, (48, 2), 0),      # layer3 conv3 at bs=1 as tuned: tail split, ~1 item per workgroup
    (1, 1024, 34, 34, 256, 1, 1, 0, 1, False, 3, (0, 0), 0),      # layer3 conv1 at bs=1: K split 3, fused finish
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, 6, (0, 0), 64),      # 3x3 (padding taps, tap changes inside an item), 7 items per workgroup
    (2, 256, 34, 34, 1024, 1, 1, 0, 1, True, 1, (0, 0), 40),      # 592 tiles on 40 workgroups: the operand stream crosses 14 item boundaries
    (2, 128, 19, 19, 128, 3, 2, 1, 0, False, 1, (0, 0), 8),       # stride 2, no activation, ragged M (last tile 41 rows)
    (1, 256, 9, 9, 96, 3, 1, 1, 1, False, 2, (0, 0), 8),          # ragged N (96 = 64 + 32), K split, 1 K tile items inside a ring of 8
    (1, 64, 12, 12, 64, 1, 1, 0, 1, True, 2, (0, 0), 8),          # items of ONE K tile: the loader runs several items ahead of the MFMAs
    (3, 512, 17, 17, 2048, 1, 1, 0, 1, True, 1, (30, 4), 0),      # layer4 conv3, tail of 30 tiles x 4 slices
]


@pytest.mark.parametrize('case', PERS_CASES)
def test_conv_persistent_kernel_parity(case):
    """stages 42 / 43 / 44 / 46 / 48 (csrc/conv_persist.hip): `grid_wgs` persistent workgroups walk the (tile, K slice) items with
    ONE direct-to-LDS ring whose loader runs on across item boundaries.  Same work items, same MFMA order, same slice-sum order
    and the same epilogue arithmetic as the per-item kernel (stages 22) -> BIT-IDENTICAL output for every ring depth and every
    grid size (1 ... many items per workgroup), and within 1e-4 of an fp64 convolution."""
    from yolact_minimal_amd import hip
    b, cin, h, w, cout, k, stride, pad, act, use_res, ksplit, tail, grid = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    counters = torch.zeros(hip.TILE_COUNTERS, device=_dev(), dtype=torch.int32)
    base = run_conv(x, wt, scale, shift, res, stride, pad, act, (64, 64), ksplit, 0, 22, counters=counters, tail=tail)
    for stages in (42, 43, 44, 46, 48):
        for gw in sorted({grid, 8 if grid else 0}):
            got = run_conv(x, wt, scale, shift, res, stride, pad, act, (64, 64), ksplit, 0, stages, counters=counters, tail=tail,
                           grid_wgs=gw, repeat=2)
            assert not torch.isnan(got).any(), (stages, gw)
            assert torch.equal(got, base), (stages, gw, float((got - base).abs().max()))
    assert int(counters.abs().sum()) == 0
    torch.testing.assert_close(base, ref_conv(x, wt, scale, shift, res, stride, pad, act), rtol=1e-4, atol=1e-5)


WS_CASES = [
    # b, cin, h, w, cout, act, residual, scale/shift
    (2, 64, 37, 41, 256, 1, True, True),         # layer1 conv3 shape class: K = 64, ragged M (3034 rows: no tile divides it)
    (1, 64, 30, 30, 64, 1, False, True),         # layer1.0 conv1
    (2, 256, 23, 19, 64, 1, False, True),        # layer1 conv1: K = 256 -> only the 64-channel slice fits the LDS
    (1, 128, 33, 33, 512, 0, True, False),       # layer2 conv3: four 128-channel slices, no activation
    (3, 96, 20, 20, 288, 0, False, True),        # Swin stage-1 qkv: K = 96 (3 K tiles), N = 288 = 2 x 128 + 32
    (1, 32, 9, 7, 36, 1, True, True),            # one K tile, fewer rows than a block, N % 32 != 0
]


@pytest.mark.parametrize('case', WS_CASES)
def test_conv_weight_stationary_kernel_parity(case):
    """stages 52 / 53 / 54 (csrc/conv_ws.hip): the 1x1 / stride-1 convolution as a GEMM whose filter slice stays in LDS while the
    workgroup walks M blocks.  Every tile (64x256, 128x128, 256x64) whose slice fits, every ring depth, the library's grid and a
    grid of 8 workgroups (many blocks per workgroup: the A stream crosses block boundaries), against an fp64 convolution (1e-4)
    and against the 64x64 direct-to-LDS kernel (fp32 summation order); repeated launches give the same bits.  A request the
    kernel does not cover (a 3x3 filter) runs as the 64x64 kernel: the same bits as asking for that directly."""
    b, cin, h, w, cout, act, use_res, affine = case
    g = torch.Generator().manual_seed(cin * 7 + cout + h)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) * (1.0 / cin ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5 if affine else None
    shift = torch.randn(cout, generator=g) * 0.1 if affine else None
    res = torch.randn(b, cout, h, w, generator=g) if use_res else None
    want = ref_conv(x, wt, scale, shift, res, 1, 0, act)
    base = run_conv(x, wt, scale, shift, res, 1, 0, act, (64, 64), 1, 0, 22)
    ran = 0
    for tile in ((64, 256), (128, 128), (256, 64)):
        if tile[1] * cin * 4 > 64 * 1024:
            continue                                  # the slice does not fit: covered by the fallback check below
        for stages in (52, 53, 54):
            for gw in (0, 8):
                got = run_conv(x, wt, scale, shift, res, 1, 0, act, tile, 1, 0, stages, grid_wgs=gw, repeat=2)
                assert not torch.isnan(got).any(), (tile, stages, gw)
                torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-4, msg=lambda m: f'{tile} {stages} {gw}: {m}')
                torch.testing.assert_close(got, base, rtol=2e-5, atol=2e-5)
                ran += 1
    assert ran >= 6
    # not covered: a slice that does not fit the LDS, a filter with taps -> the 64x64 direct-to-LDS kernel
    if 256 * cin * 4 > 64 * 1024:
        got = run_conv(x, wt, scale, shift, res, 1, 0, act, (64, 256), 1, 0, 53)
        assert torch.equal(got, base)
    w3 = torch.randn(cout, cin, 3, 3, generator=g) * (1.0 / (9 * cin) ** 0.5)
    a = run_conv(x, w3, scale, shift, res, 1, 1, act, (64, 64), 1, 0, 22)
    b3 = run_conv(x, w3, scale, shift, res, 1, 1, act, (128, 128), 1, 0, 53)
    assert torch.equal(a, b3)


def test_conv_weight_stationary_kernel_batchnorm_sums():
    """ym_conv_desc.bn_sum with the weight-stationary kernel: per-channel sum / sum of squares of the conv output, kept in
    registers for the workgroup's whole life and flushed once -- against fp64 sums of the kernel's own output (ragged last block:
    rows past M must not count)."""
    from yolact_minimal_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    b, cin, h, w, cout = 2, 64, 37, 41, 256
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).to(dev)
    for tile, stages, gw in (((64, 256), 53, 0), ((128, 128), 52, 8), ((256, 64), 54, 0)):
        out = torch.full((b, h, w, cout), float('nan'), device=dev)
        sums = torch.zeros(2, cout, dtype=torch.float64, device=dev)
        d = hip.ConvDesc()
        d.inp, d.weight = x.data_ptr(), wt.data_ptr()
        d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout, 1, 1
        d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = 1, 0, h, w, cin, 1
        d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
        d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = h * w * cout, cout, 0
        d.tile_m, d.tile_n, d.ksplit, d.stages, d.grid_wgs = tile[0], tile[1], 1, stages, gw
        assert hip.lib().ym_conv2d_fuses_bn_stats(ctypes.byref(d)) == 1
        d.bn_sum, d.bn_sumsq = sums[0].data_ptr(), sums[1].data_ptr()
        hip.conv2d_fwd(d, None)
        torch.cuda.synchronize()
        y = out.double().reshape(-1, cout)
        torch.testing.assert_close(out.reshape(-1, cout), (x.reshape(-1, cin) @ wt.t()), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(sums[0], y.sum(0), rtol=1e-5, atol=1e-3)          # (fp32 over the 32 rows of a block, fp64 across)
        torch.testing.assert_close(sums[1], (y * y).sum(0), rtol=1e-5, atol=1e-3)


WAVE_CASES = [
    # b, cin, h, w, cout, k, stride, pad, act, residual, wave tile, kwaves
    (1, 64, 17, 17, 64, 1, 1, 0, 1, False, (32, 32), 1),
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, (32, 32), 4),
    (1, 1024, 9, 11, 256, 1, 1, 0, 1, True, (32, 32), 8),
    (2, 128, 19, 19, 128, 3, 2, 1, 1, False, (64, 32), 2),
    (1, 256, 13, 13, 1024, 1, 1, 0, 1, True, (32, 64), 1),
    (1, 512, 7, 7, 512, 3, 1, 1, 0, False, (64, 64), 4),
    (2, 64, 20, 23, 96, 1, 1, 0, 2, False, (64, 64), 1),
    (1, 256, 5, 5, 255, 3, 2, 1, 1, False, (32, 32), 8),   # Cout % 4 != 0 -> scalar epilogue
    (1, 64, 6, 6, 64, 3, 1, 1, 1, False, (64, 32), 8),     # more K waves than useful tiles
]


@pytest.mark.parametrize('case', WAVE_CASES)
def test_conv_wave_kernel_parity(case):
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, kwaves = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, 0, kwaves)
    want = ref_conv(x, wt, scale, shift, res, stride, pad, act)
    assert not torch.isnan(got).any()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    # deterministic: the K-wave partials are combined in a fixed order
    again = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, 0, kwaves)
    assert torch.equal(got, again)


WAVE_DMA_CASES = [
    # b, cin, h, w, cout, k, stride, pad, act, residual, wave tile, kwaves, ring (stages 22 / 23 / 24)
    (1, 64, 17, 17, 64, 1, 1, 0, 1, False, (32, 32), 1, 22),
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, (32, 32), 4, 22),        # layer3's 3x3 at batch 1: the tuned choice
    (1, 1024, 34, 34, 256, 1, 1, 0, 1, True, (32, 32), 4, 22),        # layer3's conv1
    (1, 256, 34, 34, 1024, 1, 1, 0, 1, True, (32, 32), 1, 22),        # layer3's conv3 (+ residual): four tiles per workgroup
    (1, 1024, 9, 11, 256, 1, 1, 0, 1, True, (32, 32), 2, 24),         # ring of 4
    (2, 128, 19, 19, 128, 3, 2, 1, 1, False, (64, 32), 2, 23),        # stride 2, padding taps, ring of 3
    (1, 256, 13, 13, 1024, 1, 1, 0, 1, True, (32, 64), 1, 22),
    (1, 512, 7, 7, 512, 3, 1, 1, 0, False, (32, 64), 4, 23),
    (2, 64, 20, 23, 96, 1, 1, 0, 2, False, (64, 32), 1, 22),          # tanh epilogue, K = 2 tiles
    (1, 256, 5, 5, 255, 3, 2, 1, 1, False, (32, 32), 4, 22),          # Cout % 4 != 0 -> scalar epilogue, weight rows past Cout
    (1, 32, 6, 6, 64, 1, 1, 0, 1, False, (32, 32), 4, 22),            # ONE K tile for four K waves: three waves hold nothing
    (1, 96, 9, 9, 64, 3, 1, 1, 1, False, (64, 32), 4, 23),            # Cin = 96: a filter tap is three K tiles, ranges cut inside taps
    # fewer waves per workgroup (ym_conv_desc.grid_wgs = 1 / 2): 13th field
    (1, 256, 34, 34, 1024, 1, 1, 0, 1, True, (32, 32), 1, 22, 1),     # layer3's conv3: one wave = one tile = one workgroup
    (1, 256, 34, 34, 1024, 1, 1, 0, 1, True, (32, 32), 1, 23, 2),
    (2, 128, 19, 19, 128, 3, 2, 1, 1, False, (32, 64), 2, 22, 2),
    (1, 256, 5, 5, 255, 3, 2, 1, 0, False, (64, 32), 1, 22, 1),
]


@pytest.mark.parametrize('case', WAVE_DMA_CASES)
def test_conv_wave_dma_ring_kernel_parity(case):
    """conv_wdma_f32 (round 4): a wave owns a tile and a share of K, its operands stream through a wave-private LDS ring filled
    by global->LDS DMA, the K waves of a tile combine through LDS in a fixed order.  fp64 reference; bit-reproducible run to run
    (a ring hazard -- a stage re-filled while its fragments are still being read -- would show as run-to-run differences)."""
    b, cin, h, w, cout, k, stride, pad, act, use_res, tile, kwaves, stages = case[:13]
    wpb = case[13] if len(case) > 13 else 0
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, 0, kwaves, stages, grid_wgs=wpb)
    want = ref_conv(x, wt, scale, shift, res, stride, pad, act)
    assert not torch.isnan(got).any()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    for _ in range(3):
        again = run_conv(x, wt, scale, shift, res, stride, pad, act, tile, 0, kwaves, stages, repeat=20, grid_wgs=wpb)
        assert torch.equal(got, again)


@pytest.mark.parametrize('case', [
    # b, cin, h, w, cout, k, stride, pad, act, residual, tail tiles, tail split
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, 40, 6),           # layer3's 3x3 at batch 1: 296 tiles = 256 whole + 40 in sixths
    (1, 1024, 34, 34, 256, 1, 1, 0, 1, True, 40, 4),           # layer3's conv1 (+ residual in the last arriver's epilogue)
    (1, 256, 34, 34, 256, 3, 1, 1, 1, False, 296, 3),          # every tile in the tail
    (1, 64, 9, 9, 96, 3, 1, 1, 0, False, 5, 8),                # rows past M, 18 K tiles in 8 slices (the last ones short)
    (2, 128, 19, 19, 128, 3, 2, 1, 1, False, 7, 16),           # more slices asked for than the gather takes: clamped to 8
])
def test_conv_wave_dma_ring_tail_split(case):
    """conv_wdma_f32 with `tail_tiles`: the last tiles are computed as K slices by extra workgroups (second slot of every CU) whose
    partial tiles meet through the workspace -- fp64 reference, bit-reproducible, counters left at zero."""
    from yolact_minimal_amd import hip
    b, cin, h, w, cout, k, stride, pad, act, use_res, tail_tiles, tail_split = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(b, cout, ho, wo, generator=g) if use_res else None
    counters = torch.zeros(hip.TILE_COUNTERS, device=_dev(), dtype=torch.int32)
    kw = dict(tile=(32, 32), kwaves=4, stages=22, counters=counters, tail=(tail_tiles, tail_split))
    got = run_conv(x, wt, scale, shift, res, stride, pad, act, **kw)
    want = ref_conv(x, wt, scale, shift, res, stride, pad, act)
    assert not torch.isnan(got).any()
    torch.testing.assert_close(got, want, rtol=1e-4, atol=1e-5)
    for _ in range(3):
        assert torch.equal(got, run_conv(x, wt, scale, shift, res, stride, pad, act, repeat=30, **kw))
    assert int(counters.abs().sum()) == 0


_TILE_ORDER_PROBE = r"""
import sys, torch
sys.path.insert(0, {repo!r})
from tests.test_gpu_ops import run_conv, _dev
from yolact_minimal_amd import hip
g = torch.Generator().manual_seed(77)
out = []
# (cin, h, w, cout, k, pad, kwargs): tile columns that the group counts 2 / 3 / 4 / 8 do not divide, a tail, a K split, the walker
cases = [
    (64, 34, 34, 160, 3, 1, dict(tile=(32, 32), kwaves=4, stages=22)),                       # 37 x 5 wave tiles
    (256, 34, 34, 288, 1, 0, dict(tile=(32, 32), kwaves=4, stages=22, tail=(77, 4))),        # 37 x 9 = 333 tiles, 77 of them in the tail
    (128, 34, 34, 224, 1, 0, dict(tile=(32, 32), kwaves=1, stages=22, grid_wgs=1)),          # one-wave workgroups, 37 x 7
    (64, 40, 40, 320, 3, 1, dict(tile=(64, 64), ksplit=1, stages=22)),                       # conv_igemm_f32, 25 x 5
    (256, 20, 20, 448, 1, 0, dict(tile=(64, 64), ksplit=3, stages=22)),                      # K slices meet in the fused finish, 7 x 7
    (64, 40, 40, 320, 3, 1, dict(tile=(64, 64), ksplit=1, stages=43, grid_wgs=16)),          # conv_igemm_pers
]
for cin, h, w, cout, k, pad, kw in cases:
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) * (1.0 / (cin * k * k) ** 0.5)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    counters = torch.zeros(hip.TILE_COUNTERS, device=_dev(), dtype=torch.int32)
    y = run_conv(x, wt, sc, sh, None, 1, pad, 1, counters=counters, **kw)
    assert not torch.isnan(y).any() and int(counters.abs().sum()) == 0
    out.append(y.contiguous())
torch.save(out, sys.argv[1])
"""


def test_tile_order_groups_do_not_change_results(tmp_path):
    """The order in which tile ids map to (tile row, tile column) (csrc/conv_common.h: ym_tile_decode, groups of tile columns chosen per
    launch from the operand bytes) decides which workgroup computes a tile, never how: every forced group count -- including ones
    that leave a narrower last group -- must reproduce the default's outputs bit for bit, for the wave kernels (one-wave
    workgroups too), the per-item kernel (with K slices) and the persistent walker.  One exception by construction: with a TAIL
    split the last tile ids are computed slice by slice (another summation order than a whole tile's), and which tiles those are
    depends on the order -- there the outputs agree to rounding, not to the bit (the per-item kernel keeps the plain order under a
    tail for that reason).  YM_TILE_GROUPS is read once per process, hence the subprocesses."""
    import subprocess
    import sys
    from tests.conftest import REPO
    seen = {}
    for setting in (None, '1', '2', '3', '4', '8', '64', 'legacy'):
        env = dict(os.environ)
        env.pop('YM_TILE_GROUPS', None)
        if setting is not None:
            env['YM_TILE_GROUPS'] = setting
        path = str(tmp_path / f'out_{setting}.pt')
        out = subprocess.run([sys.executable, '-c', _TILE_ORDER_PROBE.format(repo=REPO), path], env=env, capture_output=True, text=True,
                             timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        seen[setting] = torch.load(path)
    base = seen[None]
    assert len(base) == 6
    for setting, outs in seen.items():
        for i, (a, b) in enumerate(zip(outs, base)):
            if i == 1:                                     # the launch with a tail
                torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6, msg=f'YM_TILE_GROUPS={setting}, case {i}')
            else:
                assert torch.equal(a, b), (setting, i, float((a - b).abs().max()))
    # and the tail case is not vacuous: some forced order does move tiles into / out of the tail
    assert any(not torch.equal(outs[1], base[1]) for outs in seen.values())


def test_conv_wave_dma_ring_rejects_what_it_does_not_cover():
    """The DMA-ring variant needs Cin % 32 == 0 and a 32x32 / 64x32 / 32x64 wave tile; anything else is an error, not a silent
    fallback."""
    x = torch.randn(1, 3, 32, 32)
    w = torch.randn(64, 3, 7, 7)
    with pytest.raises(RuntimeError):
        run_conv(x, w, stride=2, pad=3, tile=(32, 32), kwaves=1, stages=22)          # stem (Cin = 4)
    x = torch.randn(1, 64, 8, 8)
    w = torch.randn(64, 64, 1, 1)
    with pytest.raises(RuntimeError):
        run_conv(x, w, tile=(64, 64), kwaves=1, stages=22)


def test_conv_identity_asymmetric():
    """Transpose-detecting check (cdna guide G9): identity 1x1 weight must return the input exactly."""
    x = torch.arange(2 * 64 * 5 * 7, dtype=torch.float32).reshape(2, 64, 5, 7) * 0.01
    w = torch.eye(64).reshape(64, 64, 1, 1)
    got = run_conv(x, w)
    assert torch.equal(got, x)


def test_conv_three_segments():
    """Head-style routing: 351 output channels into three tensors with different pitch / activation."""
    from yolact_minimal_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    b, h, w, cin = 2, 9, 9, 256
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(351, cin, 3, 3, generator=g) * 0.02
    bias = torch.randn(351, generator=g) * 0.1
    n_total, off = 400, 37
    conf = torch.zeros(b, n_total, 81, device=dev)
    box = torch.zeros(b, n_total, 4, device=dev)
    coef = torch.zeros(b, n_total, 32, device=dev)
    xin = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wp = hip.pack_conv_weight(wt.to(dev), cin, 9 * cin)
    sh = bias.to(dev)
    d = hip.ConvDesc()
    d.inp, d.weight, d.shift = xin.data_ptr(), wp.data_ptr(), sh.data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, 351, 3, 3
    d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = 1, 1, h, w, 9 * cin, 3
    for i, (n0, n1, t, c, act) in enumerate([(0, 243, conf, 81, 0), (243, 255, box, 4, 0), (255, 351, coef, 32, 2)]):
        d.seg[i].n_begin, d.seg[i].n_end = n0, n1
        d.seg[i].out = t.data_ptr() + off * c * 4
        d.seg[i].batch_stride, d.seg[i].pitch, d.seg[i].act = n_total * c, 3 * c, act
    ws = torch.empty(max(hip.conv_workspace_bytes(d), 256), dtype=torch.uint8, device=dev)
    hip.conv2d_fwd(d, ws)
    torch.cuda.synchronize()
    y = F.conv2d(x, wt, bias, 1, 1).permute(0, 2, 3, 1)            # [b,h,w,351]
    want_conf = y[..., :243].reshape(b, -1, 81)
    want_box = y[..., 243:255].reshape(b, -1, 4)
    want_coef = torch.tanh(y[..., 255:]).reshape(b, -1, 32)
    n = h * w * 3
    torch.testing.assert_close(conf[:, off:off + n].cpu(), want_conf, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(box[:, off:off + n].cpu(), want_box, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(coef[:, off:off + n].cpu(), want_coef, rtol=1e-4, atol=1e-5)
    assert float(conf[:, :off].abs().sum()) == 0 and float(conf[:, off + n:].abs().sum()) == 0


def test_conv_rejects_bad_args():
    from yolact_minimal_amd import hip
    d = hip.ConvDesc()
    with pytest.raises(RuntimeError):
        hip.conv2d_fwd(d, None)


def test_maxpool():
    from yolact_minimal_amd import hip
    x = torch.randn(2, 64, 31, 30)
    xin = x.permute(0, 2, 3, 1).contiguous().to(_dev())
    out = torch.empty(2, 16, 15, 64, device=_dev())
    hip.maxpool3x3s2(xin, out)
    want = F.max_pool2d(x, 3, 2, 1)
    assert torch.equal(out.cpu().permute(0, 3, 1, 2), want)


@pytest.mark.parametrize('b,h,w', [(2, 64, 64), (1, 33, 47), (1, 70, 54), (2, 97, 161), (1, 544, 544)])
def test_fused_stem_equals_conv_then_maxpool(b, h, w):
    """`ym_stem_conv_bn_relu_maxpool` (the eval-mode ResNet stem in one launch, modules/resnet.py:86-91, reading the NCHW image
    itself) against the three launches it replaces -- `ym_nchw_to_nhwc4`, `ym_conv2d_fwd` in stem mode with folded BN + ReLU,
    `ym_maxpool3x3s2_fwd`: same MFMA order and epilogue arithmetic -> the SAME BITS (pooled tiles that overhang the image, odd
    conv / pool sizes, image borders where the window is clipped); and against PyTorch fp64 on the CPU."""
    from yolact_minimal_amd import hip
    g = torch.Generator().manual_seed(b * 1000 + h + w)
    x = torch.randn(b, 3, h, w, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) * 0.08
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    conv = run_conv(x, wt, scale, shift, None, 2, 3, 1)                       # NCHW cpu, [b, 64, ho, wo]
    ho, wo = conv.shape[2:]
    hp, wp = (ho + 2 - 3) // 2 + 1, (wo + 2 - 3) // 2 + 1
    pooled = torch.empty(b, hp, wp, 64, device=_dev())
    hip.maxpool3x3s2(conv.permute(0, 2, 3, 1).contiguous().to(_dev()), pooled)
    wpk = hip.pack_conv_weight(wt.to(_dev()), 4, 224)
    out = torch.full((b, hp, wp, 64), float('nan'), device=_dev())
    hip.stem_conv_bn_relu_maxpool(x.to(_dev()), wpk, scale.to(_dev()), shift.to(_dev()), out)
    torch.cuda.synchronize()
    assert torch.equal(out, pooled), float((out - pooled).abs().max())
    want = F.max_pool2d(F.relu(F.conv2d(x.double(), wt.double(), None, 2, 3) * scale.double().view(1, -1, 1, 1)
                               + shift.double().view(1, -1, 1, 1)), 3, 2, 1)
    torch.testing.assert_close(out.cpu().permute(0, 3, 1, 2).double(), want, rtol=1e-4, atol=1e-4)
    # NaN / inf in the image propagate like the two-kernel path (max-pool takes NaN; padded K taps contribute nothing)
    x2 = x.clone()
    x2[0, 1, h // 2, w // 3] = float('nan')
    x2[0, 2, 0, 0] = float('inf')
    conv2 = run_conv(x2, wt, scale, shift, None, 2, 3, 1)
    hip.maxpool3x3s2(conv2.permute(0, 2, 3, 1).contiguous().to(_dev()), pooled)
    hip.stem_conv_bn_relu_maxpool(x2.to(_dev()), wpk, scale.to(_dev()), shift.to(_dev()), out)
    torch.cuda.synchronize()
    assert torch.equal(torch.isnan(out), torch.isnan(pooled)) and bool(torch.isnan(out).any())
    assert torch.equal(torch.nan_to_num(out, 7.0), torch.nan_to_num(pooled, 7.0))
    with pytest.raises(RuntimeError):
        hip.stem_conv_bn_relu_maxpool(x.to(_dev()), torch.zeros(64, 256, device=_dev()), scale.to(_dev()), shift.to(_dev()), out)


@pytest.mark.parametrize('align', [False, True])
def test_bilinear2x(align):
    from yolact_minimal_amd import hip
    x = torch.randn(2, 8, 9, 7)
    xin = x.permute(0, 2, 3, 1).contiguous().to(_dev())
    out = torch.empty(2, 18, 14, 8, device=_dev())
    hip.bilinear2x(xin, out, align)
    want = F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=align)
    torch.testing.assert_close(out.cpu().permute(0, 3, 1, 2), want, rtol=1e-6, atol=1e-6)


def test_softmax_rows():
    from yolact_minimal_amd import hip
    x = torch.randn(3, 1000, 81) * 4
    xd = x.to(_dev())
    out = torch.empty_like(xd)
    hip.softmax_rows(xd, out)
    torch.testing.assert_close(out.cpu(), F.softmax(x, -1), rtol=1e-5, atol=1e-7)


def test_fold_bn_and_pack():
    from yolact_minimal_amd import hip
    dev = _dev()
    g, b, m, v = torch.rand(70) + 0.5, torch.randn(70), torch.randn(70), torch.rand(70) + 0.5
    sc, sh = hip.fold_bn(g.to(dev), b.to(dev), m.to(dev), v.to(dev), 1e-5)
    want_sc = g / torch.sqrt(v + 1e-5)
    torch.testing.assert_close(sc.cpu(), want_sc, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(sh.cpu(), b - m * want_sc, rtol=1e-5, atol=1e-6)
    w = torch.randn(5, 3, 7, 7)
    p = hip.pack_conv_weight(w.to(dev), 4, 224).cpu()
    want = torch.zeros(5, 7, 7, 4)
    want[..., :3] = w.permute(0, 2, 3, 1)
    assert torch.equal(p[:, :196], want.reshape(5, 196)) and float(p[:, 196:].abs().sum()) == 0


@pytest.mark.parametrize('h,w,dtype', [(480, 640, torch.uint8), (640, 427, torch.float32), (544, 544, torch.uint8), (37, 91, torch.float32)])
def test_val_aug_preprocess(h, w, dtype):
    from oracle import yolact_ref as R
    from yolact_minimal_amd.utils.augmentations import val_aug
    g = torch.Generator().manual_seed(h + w)
    img = torch.randint(0, 256, (h, w, 3), generator=g).to(dtype)
    got = val_aug(img.to(_dev()), 544)
    want = R.val_aug(img, 544)
    assert got.shape == (3, 544, 544)
    torch.testing.assert_close(got.cpu(), want, rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize('tile,ksplit,stages,tail', [((64, 64), 1, 22, (50, 2)), ((64, 64), 1, 22, (0, 0)), ((64, 64), 4, 22, (0, 0)),
                                                     ((64, 64), 3, 0, (0, 0)), ((128, 64), 1, 23, (37, 3)), ((64, 64), 1, 34, (0, 0)),
                                                     ((128, 128), 1, 103, (0, 0)), ((64, 64), 1, 103, (50, 2)), ((64, 64), 3, 103, (0, 0)), ((128, 64), 1, 106, (0, 0)),
                                                     # the persistent kernel (conv_persist.hip): ring of 3 / 4 / 8, ~3.4 items per workgroup, K-slice exchange, tail
                                                     ((64, 64), 1, 43, (0, 0)), ((64, 64), 1, 44, (50, 2)), ((64, 64), 3, 43, (0, 0)), ((64, 64), 1, 48, (0, 0)), ((64, 64), 2, 42, (0, 0))])
def test_conv_launches_are_race_free(tile, ksplit, stages, tail):
    """400 back-to-back launches of a chip-filling shape (2610 tiles: the Swin-T bs=8 qkv conv, M9248_N1152_C384) must all be
    bit-identical, and equal to the plain launch of the same tile up to the association of the K sum.  Regression test for two
    in-launch hazards the bs=8 @544 golden tests exposed as run-to-run differences in ~1 % of the launches:
      * direct-to-LDS ring: the buffer a DMA re-stages right after the tile barrier was still being read by a slower wave (the
        barrier did not retire the ds_reads: WAR on LDS) -> one wrong 16-byte operand chunk = 32 wrong outputs;
      * split-K / tail exchange: the arrival counter could overtake slice stores still in flight (no vmcnt drain)."""
    from yolact_minimal_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    b, h, w, cin, cout = 8, 34, 34, 384, 1152
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * 0.05).to(dev)
    wp = hip.pack_conv_weight(wt, cin, cin)
    out = torch.empty(b, h, w, cout, device=dev)
    counters = torch.zeros(hip.TILE_COUNTERS, device=dev, dtype=torch.int32)
    d = hip.ConvDesc()
    d.inp, d.weight = x.data_ptr(), wp.data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout, 1, 1
    d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = 1, 0, h, w, cin, 1
    d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
    d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = h * w * cout, cout, 0
    d.tile_counters = counters.data_ptr()
    d.tile_m, d.tile_n, d.ksplit, d.stages = tile[0], tile[1], 1, 0
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    hip.conv2d_fwd(d, ws)
    want = out.clone()
    d.ksplit, d.stages = ksplit, stages % 100
    if stages >= 100:                               # 103 / 106: split-bf16 x3 / x6 (register staging)
        d.mma, d.stages = stages - 100, 0
    d.tail_tiles, d.tail_ksplit = tail
    if 42 <= stages <= 48:
        d.grid_wgs = 768                            # 2610+ items on 768 persistent workgroups
    assert hip.conv_workspace_bytes(d) <= ws.numel()
    hip.conv2d_fwd(d, ws)
    first = out.clone()
    bad_t = torch.zeros((), device=dev, dtype=torch.int64)
    for it in range(400):                      # (ksplit > 1 changes the association of the K sum: launches are compared with each other)
        hip.conv2d_fwd(d, ws)
        bad_t += (out != first).any()
    bad = int(bad_t)
    assert bad == 0, f'{bad} launches differed from the first one'
    assert int(counters.abs().sum()) == 0                                     # arrival counters left at zero
    torch.testing.assert_close(first, want, rtol=1e-5, atol=1e-5 if d.mma != 3 else 1e-4)


@pytest.mark.parametrize('tile,kwaves,stages', [((32, 32), 4, 22), ((32, 32), 1, 24), ((64, 32), 2, 23), ((32, 64), 4, 22)])
def test_conv_wave_dma_ring_launches_are_race_free(tile, kwaves, stages):
    """The same stress as `test_conv_launches_are_race_free` for conv_wdma_f32: 400 launches of a chip-filling shape (M9248_N1152_C384,
    10 k+ wave tiles) must be bit-identical -- the wave-private rings have no barrier, only counted vmcnt waits, so a stage
    re-filled too early or a fragment read before its tile landed would show up here -- and equal to the LDS-tiled kernel up to
    the association of the K sum."""
    from yolact_minimal_amd import hip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    b, h, w, cin, cout = 8, 34, 34, 384, 1152
    x = torch.randn(b, h, w, cin, generator=g).to(dev)
    wt = (torch.randn(cout, cin, 1, 1, generator=g) * 0.05).to(dev)
    wp = hip.pack_conv_weight(wt, cin, cin)
    out = torch.empty(b, h, w, cout, device=dev)
    d = hip.ConvDesc()
    d.inp, d.weight = x.data_ptr(), wp.data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW = b, h, w, cin, cout, 1, 1
    d.stride, d.pad, d.Ho, d.Wo, d.k_pad, d.nseg = 1, 0, h, w, cin, 1
    d.seg[0].n_begin, d.seg[0].n_end, d.seg[0].out = 0, cout, out.data_ptr()
    d.seg[0].batch_stride, d.seg[0].pitch, d.seg[0].act = h * w * cout, cout, 0
    d.tile_m, d.tile_n = 64, 64
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    hip.conv2d_fwd(d, ws)
    want = out.clone()
    d.tile_m, d.tile_n, d.kwaves, d.stages = tile[0], tile[1], kwaves, stages
    hip.conv2d_fwd(d, ws)
    first = out.clone()
    bad_t = torch.zeros((), device=dev, dtype=torch.int64)
    for it in range(400):
        hip.conv2d_fwd(d, ws)
        bad_t += (out != first).any()
    assert int(bad_t) == 0, f'{int(bad_t)} launches differed from the first one'
    torch.testing.assert_close(first, want, rtol=1e-5, atol=1e-5)
