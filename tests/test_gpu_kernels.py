"""GPU: each HIP kernel behind the C ABI against a plain fp32 restatement of the same op
(torch-CPU / the oracle's cell).  Tolerances are absolute fp32 bounds written next to each check."""
import os

import numpy as np
import pytest
import torch

from oracle import s2vt_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _r(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("M,N,K", [(5, 7, 3), (128, 128, 32), (200, 300, 100), (129, 1000, 1000), (64, 50, 50),
                                   (1, 1, 1), (257, 130, 33)])
def test_gemm_all_layouts(lib, M, N, K):
    from s2vt_video_caption_amd import ops
    a, b, bias = _r(M, K, seed=1), _r(N, K, seed=2), _r(N, seed=3)
    ref = a.double() @ b.double().t()
    tol = 2e-6 * K ** 0.5 * 4 + 1e-6          # fp32 accumulation over K terms of O(1) products
    got = ops.gemm(a.to(DEV), b.to(DEV), bias=bias.to(DEV)).cpu()
    assert (got.double() - (ref + bias.double())).abs().max().item() < tol
    got = ops.gemm(a.to(DEV), b.t().contiguous().to(DEV), b_kmajor=False).cpu()               # A[M,K]·B[K,N]
    assert (got.double() - ref).abs().max().item() < tol
    got = ops.gemm(a.t().contiguous().to(DEV), b.t().contiguous().to(DEV), a_kmajor=False, b_kmajor=False).cpu()
    assert (got.double() - ref).abs().max().item() < tol                                      # A^T stored
    c0 = _r(M, N, seed=4)
    out = c0.to(DEV).clone()
    ops.gemm(a.to(DEV), b.to(DEV), out=out, accumulate=True)
    assert (out.cpu().double() - (ref + c0.double())).abs().max().item() < tol


def test_gemm_is_exact_on_integers(lib):
    """Small-integer operands: every partial sum is exactly representable, so the result must be exact —
    this catches any wrong fragment / k mapping (asymmetric B)."""
    from s2vt_video_caption_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-4, 5, (150, 77), generator=g).float()
    b = torch.randint(-4, 5, (93, 77), generator=g).float()
    for ak, bk, A, B in ((True, True, a, b), (True, False, a, b.t().contiguous()),
                         (False, False, a.t().contiguous(), b.t().contiguous())):
        got = ops.gemm(A.to(DEV), B.to(DEV), a_kmajor=ak, b_kmajor=bk).cpu()
        assert torch.equal(got, a @ b.t())


@pytest.mark.parametrize("M,N,K", [(5, 7, 3), (200, 300, 104), (257, 513, 1000), (640, 256, 2048), (64, 1000, 4100),
                                   (1, 1, 1), (300, 70, 65)])
def test_split_precision_gemm_blocked_planes(lib, M, N, K):
    """3 bf16 planes x 6 plane products (split.hip + gemm_x3.hip) against fp64: fp32-equivalent (the bound is the one
    of test_gemm_all_layouts), in both split orientations, with bias / accumulate and with split-K scratch."""
    from s2vt_video_caption_amd import ops
    a, b, bias = _r(M, K, seed=1), _r(N, K, seed=2), _r(N, seed=3)
    ref = a.double() @ b.double().t()
    tol = 4e-6 * ref.abs().max().item() + 1e-6      # fp32-equivalent: a few ulp of the largest output (fp32 MFMA: 1.5-2e-6)
    pa, pb = ops.split_planes(a.to(DEV)), ops.split_planes(b.to(DEV))
    got = ops.gemm_planes(pa, pb, M, N, bias=bias.to(DEV)).cpu()
    assert (got.double() - (ref + bias.double())).abs().max().item() < tol
    # operands given transposed ([K, rows]): the transposing split must produce the same planes
    paT = ops.split_planes(a.t().contiguous().to(DEV), transpose=True)
    pbT = ops.split_planes(b.t().contiguous().to(DEV), transpose=True)
    assert torch.equal(paT[0][:M].cpu(), pa[0][:M].cpu()) if M % 64 == 0 else True
    got_t = ops.gemm_planes(paT, pbT, M, N, bias=bias.to(DEV)).cpu()
    assert torch.equal(got_t, got)
    c0 = _r(M, N, seed=4)
    out = c0.to(DEV).clone()
    ws = torch.empty(8 * M * N + 1, device=DEV)
    ops.gemm_planes(pa, pb, M, N, out=out, accumulate=True, splitk_ws=ws)
    assert (out.cpu().double() - (ref + c0.double())).abs().max().item() < tol


@pytest.mark.parametrize("M,N,K", [(5, 7, 3), (200, 300, 104), (257, 513, 1000), (640, 256, 2048), (64, 1000, 4100), (1, 1, 1),
                                   (1000, 1000, 1000), (2048, 1000, 4000), (4000, 1000, 5056), (1203, 4000, 1000)])
def test_plain_bf16_gemm_rows(lib, M, N, K):
    """gemm_b1_kernel (config 3: bf16 operand rows, fp32 accumulate) with and without split-K, full and clamped tiles:
    against fp64 products of the bf16-ROUNDED operands - the only error left is the fp32 accumulation order (<= 2e-5 of
    the largest output, measured ~1e-6) - with bias, accumulate and split-K scratch; integer operands must come out exact."""
    from s2vt_video_caption_amd import ops
    a, b, bias = _r(M, K, seed=11), _r(N, K, seed=12), _r(N, seed=13)
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref = ar @ br.t()
    tol = 2e-5 * ref.abs().max().item() + 1e-6
    pa, pb = ops.split_planes(a.to(DEV), 1), ops.split_planes(b.to(DEV), 1)
    assert torch.equal(pa[0][:M, :K].cpu().view(torch.bfloat16), a.bfloat16())          # the plane IS the rounded operand
    got = ops.gemm_planes(pa, pb, M, N, nplanes=1, bias=bias.to(DEV)).cpu()
    assert (got.double() - (ref + bias.double())).abs().max().item() < tol
    c0 = _r(M, N, seed=14)
    out = c0.to(DEV).clone()
    ws = torch.empty(8 * M * N + 1, device=DEV)
    ops.gemm_planes(pa, pb, M, N, nplanes=1, out=out, accumulate=True, splitk_ws=ws)
    assert (out.cpu().double() - (ref + c0.double())).abs().max().item() < tol
    g = torch.Generator().manual_seed(M + N)
    ai = torch.randint(-4, 5, (M, K), generator=g).float()
    bi = torch.randint(-4, 5, (N, K), generator=g).float()
    goti = ops.gemm_planes(ops.split_planes(ai.to(DEV), 1), ops.split_planes(bi.to(DEV), 1), M, N, nplanes=1, splitk_ws=ws).cpu()
    assert torch.equal(goti, ai @ bi.t())




def test_plain_bf16_gemm_transposed_reads(lib):
    """gemm_b1_kernel<4, true>: C = X_A^T X_B from bf16 ROW images (what the bf16 recurrence kernels write), both read transposed:
    integers exact, random operands against fp64 products of the bf16-rounded operands, ragged M / N, split-K, bias, accumulate, a
    k range that starts 64 rows into the images."""
    from s2vt_video_caption_amd import ops
    for (K, M, N, ns) in ((192, 300, 77, 0), (1024, 4000, 1000, 4), (640, 1003, 517, 0), (128, 12000, 1000, 0)):
        lib.s2vt_gemm_tune(1, 0, ns)
        try:
            g = torch.Generator().manual_seed(K + M)
            xa = torch.randint(-4, 5, (K + 64, M), generator=g).float()
            xb = torch.randint(-4, 5, (K + 64, N), generator=g).float()
            bias = torch.randint(-8, 9, (N,), generator=g).float()
            pa, pb = ops.split_planes(xa.to(DEV), 1), ops.split_planes(xb.to(DEV), 1)
            ws = torch.empty(8 * M * N + 1, device=DEV)
            got = ops.gemm_planes_tt(pa, pb, M, N, K, bias=bias.to(DEV), splitk_ws=ws, nplanes=1).cpu()
            assert torch.equal(got, xa[:K].t() @ xb[:K] + bias), (K, M, N, ns)
            pa1, pb1 = (pa[0][64:], pa[1], pa[2]), (pb[0][64:], pb[1], pb[2])
            c0 = torch.randint(-8, 9, (M, N), generator=g).float()
            out = c0.to(DEV).clone()
            ops.gemm_planes_tt(pa1, pb1, M, N, K, out=out, accumulate=True, splitk_ws=ws, nplanes=1)
            assert torch.equal(out.cpu(), xa[64:64 + K].t() @ xb[64:64 + K] + c0), (K, M, N, ns, "offset")
            a, b = _r(K, M, seed=K), _r(K, N, seed=M)
            ref = a.bfloat16().double().t() @ b.bfloat16().double()
            got = ops.gemm_planes_tt(ops.split_planes(a.to(DEV), 1), ops.split_planes(b.to(DEV), 1), M, N, K, splitk_ws=ws, nplanes=1).cpu()
            assert (got.double() - ref).abs().max().item() < 2e-5 * ref.abs().max().item() + 1e-6, (K, M, N, ns)
        finally:
            lib.s2vt_gemm_tune(1, 0, 0)


@pytest.mark.parametrize("tile_rows", [0, 128, 192, 256])
def test_split_precision_gemm_transposed_reads(lib, tile_rows):
    """gemm_x3_kernel<MI, true> (every tile height; 0 = the launcher's choice): C = X_A^T X_B with both operands read transposed from their ROW plane images (the weight-gradient
    form dW = dG^T h): integer operands exact (any wrong lane / row / chunk of the transposed fragment reads or of the gathered
    LDS-DMA shows), random operands at the fp32-equivalent bound, ragged M / N (columns not multiples of 16 / 64 / 256), split-K,
    bias, accumulate, and a k range that starts at a later 64-row block of the images (dW_hh skips the first timestep)."""
    from s2vt_video_caption_amd import ops
    for (K, M, N, ns) in ((192, 300, 77, 0), (1024, 4000, 1000, 4), (640, 1003, 517, 0), (128, 12000, 1000, 0)):
        lib.s2vt_gemm_tune(3, tile_rows, ns)
        try:
            g = torch.Generator().manual_seed(K + M)
            xa = torch.randint(-4, 5, (K + 64, M), generator=g).float()
            xb = torch.randint(-4, 5, (K + 64, N), generator=g).float()
            bias = torch.randint(-8, 9, (N,), generator=g).float()
            pa, pb = ops.split_planes(xa.to(DEV), 3), ops.split_planes(xb.to(DEV), 3)
            ws = torch.empty(8 * M * N + 1, device=DEV)
            got = ops.gemm_planes_tt(pa, pb, M, N, K, bias=bias.to(DEV), splitk_ws=ws).cpu()
            assert torch.equal(got, xa[:K].t() @ xb[:K] + bias), (K, M, N, ns)
            # rows 64 .. 64 + K of the images (a later row block): pointers advanced by one 64-row block
            pa1 = (pa[0][64:], pa[1], pa[2])
            pb1 = (pb[0][64:], pb[1], pb[2])
            c0 = torch.randint(-8, 9, (M, N), generator=g).float()
            out = c0.to(DEV).clone()
            ops.gemm_planes_tt(pa1, pb1, M, N, K, out=out, accumulate=True, splitk_ws=ws)
            assert torch.equal(out.cpu(), xa[64:64 + K].t() @ xb[64:64 + K] + c0), (K, M, N, ns, "offset")
            a, b = _r(K, M, seed=K), _r(K, N, seed=M)
            ref = a.double().t() @ b.double()
            got = ops.gemm_planes_tt(ops.split_planes(a.to(DEV), 3), ops.split_planes(b.to(DEV), 3), M, N, K, splitk_ws=ws).cpu()
            assert (got.double() - ref).abs().max().item() < 4e-6 * ref.abs().max().item() + 1e-6, (K, M, N, ns)
        finally:
            lib.s2vt_gemm_tune(3, 0, 0)


@pytest.mark.parametrize("planes", [3, 1])
def test_transposed_read_gemm_of_an_image_beyond_the_32_bit_offsets(lib, planes):
    """A row image larger than the transposed-read kernels' 32-bit offsets reach (4 GB for gemm_x3_kernel<MI, true>, 2 GB for
    gemm_b1_kernel<4, true> - dlogits from B = 768 on at V = 12000): the launcher cuts it into k slices (split-K, fixed-order
    combine) instead of refusing it (round-4 advisor finding).  Operands in {-1, 0, 1}: every sum is an integer below 2^24, the
    result must be exact; without split-K scratch the call is refused with a message."""
    from s2vt_video_caption_amd import capi, ops
    M, N = 64, 64
    K = (11_300_000 if planes == 3 else 16_900_000) // 64 * 64        # x 192 elements x 2 B = 4.3 GB / x 64 x 2 B = 2.16 GB per image
    step = 1 << 20
    pa = ops.split_planes(torch.zeros(64, M, device=DEV), planes)      # (shape probe: planes of a 64-row image)
    ld = pa[0].numel() // 64
    ia = torch.empty(K * ld, dtype=pa[0].dtype, device=DEV)
    ib = torch.empty(K * ld, dtype=pa[0].dtype, device=DEV)
    ref = torch.zeros(M, N, dtype=torch.float64, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    for k0 in range(0, K, step):
        k1 = min(K, k0 + step)
        xa = torch.randint(-1, 2, (k1 - k0, M), generator=g, device=DEV).float()
        xb = torch.randint(-1, 2, (k1 - k0, N), generator=g, device=DEV).float()
        ref += xa.double().t() @ xb.double()
        ia[k0 * ld:k1 * ld].copy_(ops.split_planes(xa, planes)[0].reshape(-1)[:(k1 - k0) * ld])
        ib[k0 * ld:k1 * ld].copy_(ops.split_planes(xb, planes)[0].reshape(-1)[:(k1 - k0) * ld])
    assert K * ld * 2 > (4 << 30 if planes == 3 else 2 << 30)
    ws = torch.empty(8 * M * N + 1, device=DEV)
    got = ops.gemm_planes_tt((ia, pa[1], pa[2]), (ib, pa[1], pa[2]), M, N, K, splitk_ws=ws, nplanes=planes)
    assert torch.equal(got.double(), ref)
    with pytest.raises(capi.S2VTHipError, match="k slices"):
        ops.gemm_planes_tt((ia, pa[1], pa[2]), (ib, pa[1], pa[2]), M, N, K, nplanes=planes)


@pytest.mark.parametrize("tile_rows", [128, 192, 256])
def test_split_precision_gemm_every_tile_height_persistent(lib, tile_rows):
    """gemm_x3_kernel<MI> for every tile height as a PERSISTENT launch (more tiles than compute units, the 3-slot ring running on
    across the tiles of a workgroup), ragged M / N, bias, accumulate, split-K, NaN planes in the padding rows of the 64-row
    blocks: integer operands must come out exact (every plane product and every fp32 sum is exact then), random operands within
    the fp32-equivalent bound of test_split_precision_gemm_blocked_planes."""
    from s2vt_video_caption_amd import capi, ops
    try:
        for (M, N, K, ns) in ((6001, 2900, 192, 0), (4100, 1000, 2048, 3), (333, 77, 64, 0), (5120, 1000, 512, 0)):
            lib.s2vt_gemm_tune(3, tile_rows, ns)
            g = torch.Generator().manual_seed(M + N + tile_rows)
            ai = torch.randint(-4, 5, (M, K), generator=g).float()
            bi = torch.randint(-4, 5, (N, K), generator=g).float()
            bias = torch.randint(-8, 9, (N,), generator=g).float()
            pa, pb = ops.split_planes(ai.to(DEV), 3), ops.split_planes(bi.to(DEV), 3)
            ws = torch.empty(4 * M * N + 1, device=DEV)
            got = ops.gemm_planes(pa, pb, M, N, nplanes=3, bias=bias.to(DEV), splitk_ws=ws).cpu()
            assert torch.equal(got, ai @ bi.t() + bias), (M, N, K, ns)
            c0 = torch.randint(-8, 9, (M, N), generator=g).float()
            out = c0.to(DEV).clone()
            ops.gemm_planes(pa, pb, M, N, nplanes=3, out=out, accumulate=True, splitk_ws=ws)
            assert torch.equal(out.cpu(), ai @ bi.t() + c0), (M, N, K, ns, "accumulate")
            a, b = _r(M, K, seed=M), _r(N, K, seed=N)
            ref = a.double() @ b.double().t()
            got = ops.gemm_planes(ops.split_planes(a.to(DEV), 3), ops.split_planes(b.to(DEV), 3), M, N, nplanes=3, splitk_ws=ws).cpu()
            assert (got.double() - ref).abs().max().item() < 4e-6 * ref.abs().max().item() + 1e-6, (M, N, K, ns)
        capi.check_async_error()
    finally:
        lib.s2vt_gemm_tune(3, 0, 0)


@pytest.mark.parametrize("tile_rows", [128, 192, 256, 320])
def test_plain_bf16_gemm_every_tile_height_persistent(lib, tile_rows):
    """gemm_b1_kernel<MI> for every tile height, as a PERSISTENT launch (more tiles than compute units: a workgroup walks
    several, its stage pipeline running on across them), with ragged M / N, a gathered / permuted C row map, bias, split-K -
    and with NaNs planted in the operand rows beyond M and N (the padding rows of the plane buffers, which the tiles' loads
    do touch): nothing of them may reach the output.  Integer operands: exact."""
    from s2vt_video_caption_amd import capi, ops
    lib.s2vt_gemm_tune(1, tile_rows, 0)
    try:
        for (M, N, K, ns) in ((6001, 2900, 192, 0), (4100, 1000, 2048, 3), (333, 77, 64, 0), (20480 // 4, 1000, 512, 0)):
            lib.s2vt_gemm_tune(1, tile_rows, ns)
            g = torch.Generator().manual_seed(M + N + tile_rows)
            ai = torch.randint(-4, 5, (M, K), generator=g).float()
            bi = torch.randint(-4, 5, (N, K), generator=g).float()
            bias = torch.randint(-8, 9, (N,), generator=g).float()
            pa, pb = ops.split_planes(ai.to(DEV), 1), ops.split_planes(bi.to(DEV), 1)
            pa[0][M:].fill_(0x7FC0)                                  # bf16 NaN in every padding row
            pb[0][N:].fill_(0x7FC0)
            ws = torch.empty(4 * M * N + 1, device=DEV)
            got = ops.gemm_planes(pa, pb, M, N, nplanes=1, bias=bias.to(DEV), splitk_ws=ws).cpu()
            assert torch.equal(got, ai @ bi.t() + bias), (M, N, K, ns)
            c0 = torch.randint(-8, 9, (M, N), generator=g).float()
            out = c0.to(DEV).clone()
            ops.gemm_planes(pa, pb, M, N, nplanes=1, out=out, accumulate=True, splitk_ws=ws)
            assert torch.equal(out.cpu(), ai @ bi.t() + c0), (M, N, K, ns, "accumulate")
        capi.check_async_error()
    finally:
        lib.s2vt_gemm_tune(1, 0, 0)


def test_split_precision_gemm_at_the_gx1_shape_of_config2(lib):
    """gx1 of config 2 (5120 x 4000 x 1000): 320 tiles of 256 rows = 1.25 rounds of the chip, the shape the launcher's time model
    picks a tile height for (a persistent launch in which some workgroups walk two tiles): every output row against fp64 (a row
    range served twice or not at all would show), integers exact, bias and accumulate."""
    from s2vt_video_caption_amd import ops
    M, N, K = 5120, 4000, 1000
    g = torch.Generator().manual_seed(9)
    ai = torch.randint(-3, 4, (M, K), generator=g).float()
    bi = torch.randint(-3, 4, (N, K), generator=g).float()
    got = ops.gemm_planes(ops.split_planes(ai.to(DEV)), ops.split_planes(bi.to(DEV)), M, N)
    assert torch.equal(got, ai.to(DEV) @ bi.to(DEV).t())          # (integer products: any fp32 matmul is exact on them)
    a, b, bias = _r(M, K, seed=1), _r(N, K, seed=2), _r(N, seed=3)
    ref = (a.double() @ b.double().t()) + bias.double()
    tol = 4e-6 * ref.abs().max().item() + 1e-6
    pa, pb = ops.split_planes(a.to(DEV)), ops.split_planes(b.to(DEV))
    got = ops.gemm_planes(pa, pb, M, N, bias=bias.to(DEV))
    assert (got.cpu().double() - ref).abs().max().item() < tol
    out = torch.ones(M, N, device=DEV)
    ops.gemm_planes(pa, pb, M, N, out=out, accumulate=True)
    assert (out.cpu().double() - (ref - bias.double() + 1.0)).abs().max().item() < tol


def test_split_precision_gemm_is_exact_on_integers(lib):
    """Small integers are exact in the hi plane (mid/lo planes zero) and every partial sum is representable: the
    result must be exact - catches any wrong piece / fragment / k mapping of the blocked layout."""
    from s2vt_video_caption_amd import ops
    g = torch.Generator().manual_seed(5)
    a = torch.randint(-4, 5, (333, 200), generator=g).float()
    b = torch.randint(-4, 5, (517, 200), generator=g).float()
    got = ops.gemm_planes(ops.split_planes(a.to(DEV)), ops.split_planes(b.to(DEV)), 333, 517).cpu()
    assert torch.equal(got, a @ b.t())
    got = ops.gemm_planes(ops.split_planes(a.t().contiguous().to(DEV), transpose=True),
                          ops.split_planes(b.t().contiguous().to(DEV), transpose=True), 333, 517).cpu()
    assert torch.equal(got, a @ b.t())


@pytest.mark.parametrize("B,H", [(1, 32), (4, 500), (33, 40), (64, 1000), (16, 8), (17, 36), (8, 1000), (5, 1000), (2, 1500),
                                 (3, 30)])
def test_lstm_step_fwd_matches_cell(lib, B, H):
    """Batches from 1 to 64, including the small batches (B <= 8) of the reference's plumbing configuration and of decode."""
    from s2vt_video_caption_amd import ops
    k = 1.0 / H ** 0.5
    w_hh = (torch.rand(4 * H, H, generator=torch.Generator().manual_seed(1)) * 2 - 1) * k
    gx, h0, c0 = _r(B, 4 * H, seed=2), _r(B, H, seed=3, scale=0.5), _r(B, H, seed=4)
    zeros = torch.zeros(4 * H)
    h_ref, c_ref = orc.lstm_cell(None, h0, c0, None, w_hh, gx, zeros)      # gx plays b_ih (+ x-part)
    h, c, st = ops.lstm_step_fwd(gx.to(DEV), None, w_hh.to(DEV), h0.to(DEV), c0.to(DEV), want_stash=True)
    assert (h.cpu() - h_ref).abs().max().item() < 2e-6
    assert (c.cpu() - c_ref).abs().max().item() < 4e-6
    g = gx + h0 @ w_hh.t()
    i, f, gg, o = g.chunk(4, 1)
    st_ref = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], 1)
    assert (st.cpu() - st_ref).abs().max().item() < 2e-6
    # zero state + bias only (first step of a padded sequence)
    bias = _r(4 * H, seed=6)
    h_ref2, c_ref2 = orc.lstm_cell(None, torch.zeros(1, H), torch.zeros(1, H), None, w_hh, bias, zeros)
    hb, cb = ops.lstm_step_fwd(bias.to(DEV).expand(B, 4 * H).contiguous(), None, w_hh.to(DEV), None, None)
    assert (hb.cpu() - h_ref2.expand(B, H)).abs().max().item() < 1e-6



@pytest.mark.parametrize("B,H,E,V", [(1, 1000, 1000, 300), (2, 64, 40, 50), (3, 520, 500, 90), (5, 1000, 1000, 300), (8, 1000, 1000, 300),
                                     (8, 512, 512, 200), (7, 8, 4, 30)])
def test_small_batch_gemv_step_against_fp64_and_the_tile_kernel(lib, B, H, E, V):
    """lstm_step_fwd_gemv_kernel (option gemv, B <= 8: h staged in LDS, weight rows streamed to registers, wavefront shuffle
    reductions) through s2vt_lstm_step_fwd and s2vt_lstm_step_fwd_token: against the fp64 cell (fp32 rounding only: 2e-6) and
    against the 16-row MFMA tile kernel it replaces (option gemv = 0), with the gate stash, the zero-state first step (h_prev
    null), the embedded-word segment with int32 tokens / one constant token, and an id beyond the table (IndexError, token 0
    read)."""
    from s2vt_video_caption_amd import capi, ops
    w_hh, w_ih, emb = _r(4 * H, H, seed=2, scale=H ** -0.5), _r(4 * H, E + H, seed=3, scale=(E + H) ** -0.5), _r(V, E, seed=4)
    gx, hp, cp = _r(B, 4 * H, seed=1), _r(B, H, seed=5, scale=0.5), _r(B, H, seed=6, scale=0.5)
    tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(7), dtype=torch.int32)

    def ref(pre, cprev):
        i, f, g, o = pre.chunk(4, dim=1)
        c = torch.sigmoid(f) * cprev + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c), c, torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)], 1)
    dev = [t.to(DEV) for t in (gx, w_hh, hp, cp, emb, w_ih)]
    outs = {}
    prev = lib.s2vt_set_option(b"gemv", -1)
    try:
        for on in (1, 0):
            lib.s2vt_set_option(b"gemv", 2 if on else 0)        # (2: every B <= 8 takes the GEMV kernel; the default, 1, stops at B = 4)
            a = ops.lstm_step_fwd(dev[0], None, dev[1], dev[2], dev[3], want_stash=True)
            b0 = ops.lstm_step_fwd(dev[0], None, dev[1], None, None)                       # zero state: no recurrent segment
            t1 = ops.lstm_step_fwd_token(*dev, tok=tok.to(DEV))
            t2 = ops.lstm_step_fwd_token(*dev, tok_const=V - 1)
            capi.check_async_error()
            outs[on] = (a, b0, t1, t2)
            bad = ops.lstm_step_fwd_token(*dev, tok=tok.clone().index_fill_(0, torch.tensor([B - 1]), V).to(DEV))
            with pytest.raises(IndexError):
                torch.cuda.synchronize()
                capi.check_async_error()
            ok_rows = slice(0, B - 1)
            assert torch.equal(bad[0][ok_rows], t1[0][ok_rows])
    finally:
        lib.s2vt_set_option(b"gemv", prev)
    d = lambda x: x.double()
    hr, cr, sr = ref(d(gx) + d(hp) @ d(w_hh).t(), d(cp))
    h0r, c0r, _ = ref(d(gx), torch.zeros(B, H, dtype=torch.float64))
    e1 = d(emb)[tok.long()] @ d(w_ih)[:, :E].t()
    h1r, c1r, _ = ref(d(gx) + d(hp) @ d(w_hh).t() + e1, d(cp))
    e2 = d(emb)[V - 1:V].expand(B, E) @ d(w_ih)[:, :E].t()
    h2r, c2r, _ = ref(d(gx) + d(hp) @ d(w_hh).t() + e2, d(cp))
    for on in (1, 0):
        (h, c, st), (hb, cb), (ht, ct), (hk, ck) = outs[on]
        for got, want in ((h, hr), (c, cr), (st, sr), (hb, h0r), (cb, c0r), (ht, h1r), (ct, c1r), (hk, h2r), (ck, c2r)):
            assert (got.cpu().double() - want).abs().max().item() < 4e-6, on
    for x, y in zip(outs[1][0] + outs[1][2], outs[0][0] + outs[0][2]):
        assert (x - y).abs().max().item() < 2e-6                 # the two kernels differ by the order of fp32 additions only


def test_decode_step_token_segment_and_poisoned_token_word(lib):
    """The decode step's token path (S2VTModel.py:100-103: Emb[prev word] in front of vid_out) through s2vt_lstm_step_fwd_token:
    (a) int32 tokens and packed argmax words against the fp64 cell; (b) a packed word that no producer wrote (0 -> token
    0xFFFFFFFF) or an int32 id >= V must come back as IndexError from capi.check_async_error() - the row is computed with
    token 0 and nothing outside the table is addressed - where the kernel used to read emb + 4e9 rows (a queue abort)."""
    from s2vt_video_caption_amd import capi, ops
    B, H, E, V = 37, 72, 40, 50
    gx, w_hh, w_ih, emb = _r(B, 4 * H, seed=1), _r(4 * H, H, seed=2, scale=0.2), _r(4 * H, E + H, seed=3, scale=0.2), _r(V, E, seed=4)
    hp, cp = _r(B, H, seed=5, scale=0.5), _r(B, H, seed=6, scale=0.5)
    tok = torch.randint(0, V, (B,), generator=torch.Generator().manual_seed(7), dtype=torch.int32)

    def ref(tk):
        pre = gx.double() + hp.double() @ w_hh.double().t() + emb[tk.long()].double() @ w_ih[:, :E].double().t()
        i, f, g, o = pre.chunk(4, dim=1)
        c = torch.sigmoid(f) * cp.double() + torch.sigmoid(i) * torch.tanh(g)
        return torch.sigmoid(o) * torch.tanh(c), c
    args = [t.to(DEV) for t in (gx, w_hh, hp, cp, emb, w_ih)]
    h, c = ops.lstm_step_fwd_token(*args, tok=tok.to(DEV))
    capi.check_async_error()
    hr, cr = ref(tok)
    assert (h.cpu().double() - hr).abs().max().item() < 2e-6 and (c.cpu().double() - cr).abs().max().item() < 2e-6
    packed = ((torch.arange(B, dtype=torch.int64) + 1234) << 32) | (0xFFFFFFFF - tok.long())     # (ordered logit << 32 | ~index)
    h2, c2 = ops.lstm_step_fwd_token(*args, tok_packed=packed.to(DEV))
    capi.check_async_error()
    assert torch.equal(h2, h) and torch.equal(c2, c)
    h3, _ = ops.lstm_step_fwd_token(*args, tok_const=9)
    capi.check_async_error()
    assert (h3.cpu().double() - ref(torch.full((B,), 9))[0]).abs().max().item() < 2e-6
    # poisoned inputs: row 5's packed word never written; an int32 id == V; a negative id
    for kw in (dict(tok_packed=packed.clone().index_fill_(0, torch.tensor([5]), 0).to(DEV)),
               dict(tok=tok.clone().index_fill_(0, torch.tensor([5]), V).to(DEV)),
               dict(tok=tok.clone().index_fill_(0, torch.tensor([5]), -1).to(DEV)),
               dict(tok_const=V)):
        hb, cb = ops.lstm_step_fwd_token(*args, **kw)
        with pytest.raises(IndexError):
            capi.check_async_error()
        if "tok_const" not in kw:                       # every other row is untouched, the bad row ran with token 0
            tz = tok.clone()
            tz[5] = 0
            assert (hb.cpu().double() - ref(tz)[0]).abs().max().item() < 2e-6
    h4, _ = ops.lstm_step_fwd_token(*args, tok=tok.to(DEV))        # the flag does not stick
    capi.check_async_error()
    assert torch.equal(h4, h)


@pytest.mark.parametrize("B,H", [(3, 32), (20, 100), (64, 500)])
def test_lstm_step_bwd_matches_autograd(lib, B, H):
    from s2vt_video_caption_amd import ops
    k = 1.0 / H ** 0.5
    w_hh = ((torch.rand(4 * H, H, generator=torch.Generator().manual_seed(1)) * 2 - 1) * k)
    gx = _r(B, 4 * H, seed=2).requires_grad_()
    c_prev = _r(B, H, seed=4).requires_grad_()
    g = gx
    i, f, gg, o = g.chunk(4, 1)
    st = torch.cat([torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)], 1)
    c = torch.sigmoid(f) * c_prev + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    dh_out, dc_in, dg_next = _r(B, H, seed=5), _r(B, H, seed=6), _r(B, 4 * H, seed=7, scale=0.1)
    dh = dh_out + dg_next @ w_hh
    (h * dh).sum().backward(retain_graph=True, inputs=[gx, c_prev])
    dgx1, dcp1 = gx.grad.clone(), c_prev.grad.clone()
    gx.grad = None; c_prev.grad = None
    (c * dc_in).sum().backward(inputs=[gx, c_prev])
    dg_ref, dcp_ref = dgx1 + gx.grad, dcp1 + c_prev.grad
    dc = dc_in.to(DEV).clone()
    dg = ops.lstm_step_bwd(dg_next.to(DEV), w_hh.t().contiguous().to(DEV), dh_out.to(DEV), st.detach().to(DEV),
                           c.detach().to(DEV), c_prev.detach().to(DEV), dc, False)
    assert (dg.cpu() - dg_ref).abs().max().item() < 5e-6
    assert (dc.cpu() - dcp_ref).abs().max().item() < 5e-6


def test_lstm_seq_fwd_bwd_vs_autograd(lib):
    from s2vt_video_caption_amd import ops
    T, B, H, n_gx = 7, 5, 24, 4
    k = 1.0 / H ** 0.5
    gen = torch.Generator().manual_seed(11)
    w_hh = ((torch.rand(4 * H, H, generator=gen) * 2 - 1) * k).requires_grad_()
    bias = ((torch.rand(4 * H, generator=gen) * 2 - 1) * k)
    gx = torch.randn(n_gx, B, 4 * H, generator=gen).requires_grad_()
    h = torch.zeros(B, H); c = torch.zeros(B, H)
    hs = []
    for t in range(T):
        g = (gx[t] if t < n_gx else bias) + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, 1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        hs.append(h)
    hs = torch.stack(hs)
    dh_out = torch.randn(T - 2, B, H, generator=gen)          # gradient arrives for steps >= 2 only
    (hs[2:] * dh_out).sum().backward()
    h_all, c_all, stash = ops.lstm_seq_fwd(T, B, gx.detach().reshape(n_gx * B, 4 * H).to(DEV), n_gx, bias.to(DEV),
                                           w_hh.detach().to(DEV), want_stash=True)
    assert (h_all.cpu().view(T, B, H) - hs.detach()).abs().max().item() < 2e-6
    dg = ops.lstm_seq_bwd(T, B, w_hh.detach().to(DEV), dh_out.reshape(-1, H).to(DEV), 2, c_all, stash)
    dg = dg.cpu().view(T, B, 4 * H)
    assert (dg[:n_gx] - gx.grad).abs().max().item() < 5e-6
    h_prev = torch.cat([torch.zeros(1, B, H), hs.detach()[:-1]])
    dw = torch.einsum("tbg,tbh->gh", dg, h_prev)
    assert (dw - w_hh.grad).abs().max().item() < 2e-5


@pytest.mark.parametrize("B,Lm1,V", [(3, 7, 50), (4, 79, 100), (2, 5, 12000), (1, 1, 3)])
def test_mean_ce_fwd_bwd(lib, B, Lm1, V):
    from s2vt_video_caption_amd import functional as F
    logits = _r(B, Lm1, V, seed=1, scale=3.0)
    target = torch.randint(0, V, (B, Lm1 + 1), generator=torch.Generator().manual_seed(2))
    lg = logits.clone().requires_grad_()
    ref = torch.nn.functional.cross_entropy(lg.reshape(-1, V), target[:, 1:].reshape(-1))
    (ref * 1.7).backward()
    x = logits.to(DEV).requires_grad_()
    loss = F.mean_cross_entropy(x, target.to(DEV))
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 2e-6 * max(1.0, abs(float(ref)))
    assert (x.grad.cpu() - lg.grad).abs().max().item() < 1e-7 + 2e-6 / (B * Lm1)


@pytest.mark.parametrize("B,Lm1,V,kind", [(3, 7, 50, "captions"), (4, 79, 100, "ones"), (64, 79, 300, "captions"), (2, 5, 40, "weights"),
                                          (2, 5, 40, "empty"), (1, 1, 3, "ones")])
def test_mask_criterion_is_the_reference_arithmetic(lib, B, Lm1, V, kind):
    """s2vt_mask_criterion_forward / _backward (two launches) against the reference's three lines around nn.CrossEntropyLoss
    (utils.py:22-25) on the CPU: 0/1 caption masks, all ones, arbitrary weights, and the all-zero mask (NaN, as upstream)."""
    import utils
    g = torch.Generator().manual_seed(11)
    logits = _r(B, Lm1, V, seed=1, scale=3.0)
    target = torch.randint(0, V, (B, Lm1 + 1), generator=g)
    if kind == "ones":
        mask = torch.ones(B, Lm1 + 1)
    elif kind == "weights":
        mask = torch.rand(B, Lm1 + 1, generator=g) * 3.0
    elif kind == "empty":
        mask = torch.zeros(B, Lm1 + 1)
    else:
        lens = torch.randint(1, Lm1 + 2, (B,), generator=g)
        mask = (torch.arange(Lm1 + 1)[None, :] < lens[:, None]).float()
    lg = logits.clone().requires_grad_()
    ce = torch.nn.functional.cross_entropy(lg.reshape(-1, V), target[:, 1:].reshape(-1))
    w = mask[:, 1:].reshape(-1)
    ref = (ce * w).sum() / w.sum()
    x = logits.to(DEV).requires_grad_()
    loss = utils.MaskCriterion()(x, target.to(DEV), mask.to(DEV))
    if kind == "empty" or float(w.sum()) == 0.0:
        assert torch.isnan(ref) and torch.isnan(loss.cpu())
        return
    (ref * 1.7).backward()
    (loss * 1.7).backward()
    assert abs(float(loss) - float(ref)) < 3e-6 * max(1.0, abs(float(ref)))
    assert (x.grad.cpu() - lg.grad).abs().max().item() < 1e-7 + 3e-6 / (B * Lm1)
    # a strided mask view (row stride > L) and an integer mask go through the same entry
    wide = torch.zeros(B, Lm1 + 5)
    wide[:, :Lm1 + 1] = mask
    loss2 = utils.MaskCriterion()(logits.to(DEV), target.to(DEV), wide.to(DEV)[:, :Lm1 + 1])
    assert float(loss2) == float(loss)
    if kind in ("ones", "captions"):
        loss3 = utils.MaskCriterion()(logits.to(DEV), target.to(DEV), mask.long().to(DEV))
        assert float(loss3) == float(loss)


@pytest.mark.parametrize("n", [1, 5, 4096, 1000003])
def test_flat_adam_step_is_torch_adam(lib, n):
    """s2vt_adam_step against torch.optim.Adam (the reference's optimizer, train.py:89-93) on the CPU, five steps with fresh
    gradients each: the same arithmetic operation for operation, so within a couple of fp32 roundings of the update."""
    import ctypes
    g = torch.Generator().manual_seed(n)
    p0 = torch.randn(n, generator=g)
    ref = p0.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=1e-3)
    p = p0.to(DEV)
    m = torch.zeros(n, device=DEV)
    v = torch.zeros(n, device=DEV)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** float(torch.randint(-4, 2, (1,), generator=g)))
        ref.grad = grad.clone()
        opt.step()
        gd = grad.to(DEV)
        rc = lib.s2vt_adam_step(ptr(p), ptr(gd), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, step, None)
        assert rc == 0
        torch.cuda.synchronize()
        st = opt.state[ref]
        assert (m.cpu() - st["exp_avg"]).abs().max().item() <= 4e-7 * st["exp_avg"].abs().max().item()
        assert (v.cpu() - st["exp_avg_sq"]).abs().max().item() <= 4e-7 * st["exp_avg_sq"].abs().max().item()
        assert (p.cpu() - ref.detach()).abs().max().item() <= 4e-7 * step       # updates of ~1e-3 each, parameters of O(1)
    assert lib.s2vt_adam_step(ptr(p), ptr(gd), ptr(m), ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0, None) != 0     # steps count from 1


def test_flat_adam_trains_the_model_as_torch_adam_does(lib):
    """optim.FlatAdam (parameters / gradients / moments in flat buffers, one launch per step) against torch.optim.Adam over four
    train steps of one small model: same losses, same parameters; state_dict keys and shapes unchanged by the flattening."""
    import S2VTModel, utils
    from s2vt_video_caption_amd import dp, optim, synth
    L, Fd, H, E, V, B = 12, 64, 96, 80, 200, 16
    sd = synth.make_state_dict(V, Fd, H, E, seed=5)
    feats, caps, mask = (t.to(DEV) for t in synth.make_batch(B, L, Fd, V, seed=6))
    crit = utils.MaskCriterion()
    runs = []
    for kind in ("torch", "flat"):
        m = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
        m.load_state_dict(sd)
        m.to(DEV)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3) if kind == "torch" else optim.FlatAdam(m, lr=1e-3)
        losses = [float(dp.train_step(m, crit, opt, feats, caps, mask, None)) for _ in range(4)]
        runs.append((losses, {k: t.detach().cpu().clone() for k, t in m.state_dict().items()}))
        if kind == "flat":
            assert all(p.grad is not None and p.grad.data_ptr() >= opt.flat_g.data_ptr() for p in m.parameters())
            ids = m(feats, mode="test")                      # the decode sees the updated weights (version counters were bumped)
            m2 = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
            m2.load_state_dict(m.state_dict())
            assert torch.equal(ids, m2.to(DEV)(feats, mode="test"))
            # resume: a second optimizer loaded from this one's state takes the same next step
            st = opt.state_dict()
            m3 = S2VTModel.S2VT(V, Fd, L, dim_hid=H, dim_embed=E)
            m3.load_state_dict(m.state_dict())
            m3.to(DEV)
            opt3 = optim.FlatAdam(m3, lr=1e-3)
            opt3.load_state_dict(st)
            la = float(dp.train_step(m, crit, opt, feats, caps, mask, None))
            lb = float(dp.train_step(m3, crit, opt3, feats, caps, mask, None))
            assert la == lb and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))
    (l0, s0), (l1, s1) = runs
    assert l0[0] > l0[-1] and max(abs(a - b) for a, b in zip(l0, l1)) < 2e-5
    assert list(s0) == list(s1)
    for k in s0:
        assert s0[k].shape == s1[k].shape and (s0[k] - s1[k]).abs().max().item() < 2e-5, k


@pytest.mark.parametrize("planes", [False, True])
def test_decode_argmax_first_max_wins(lib, planes):
    from s2vt_video_caption_amd import ops as _ops

    class ops:      # both kernels behind one name: fp32-input MFMA (lstm.hip) / bf16 x 3 planes (argmax_x3.hip)
        decode_step_argmax = staticmethod(lambda h, w, b: _ops.decode_step_argmax(h, w, b, planes=planes))
    B, H, V = 37, 16, 1000
    h = torch.zeros(B, H); h[:, 0] = 1.0
    w = torch.zeros(V, H)
    bias = torch.zeros(V)
    # logits = w[:,0] + bias ; plant ties: the same maximum at two/three indices -> lowest index must win
    g = torch.Generator().manual_seed(3)
    w[:, 0] = torch.randint(-50, 50, (V,), generator=g).float()
    w[[7, 500, 999], 0] = 60.0
    ids = ops.decode_step_argmax(h.to(DEV), w.to(DEV), bias.to(DEV)).cpu()
    assert (ids == 7).all()
    logits = _r(B, V, seed=4)
    hh = _r(B, H, seed=5); ww = _r(V, H, seed=6); bb = _r(V, seed=7)
    ids = ops.decode_step_argmax(hh.to(DEV), ww.to(DEV), bb.to(DEV)).cpu()
    ref = (hh.double() @ ww.double().t() + bb.double())
    top2 = ref.topk(2, 1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(ids[safe], ref.argmax(1)[safe])
    # all-negative logits and -0.0 handling
    ids = ops.decode_step_argmax(hh.to(DEV), ww.to(DEV), (bb - 1000).to(DEV)).cpu()
    assert torch.equal(ids[safe], ref.argmax(1)[safe])


@pytest.mark.parametrize("planes", [False, True])
@pytest.mark.parametrize("B,H,V", [(128, 1000, 12000), (100, 72, 1000), (48, 500, 97), (300, 128, 530), (64, 1000, 12000)])
def test_decode_argmax_at_decode_sizes(lib, B, H, V, planes):
    """logits_argmax_kernel / logits_argmax_x3_kernel at decode sizes (B = 128, V = 12000, H = 1000) and ragged B / V / H:
    planted ties resolve to the lowest index, random logits give the fp64 argmax wherever the top-2 gap exceeds fp32
    rounding."""
    from s2vt_video_caption_amd import ops as _ops

    class ops:
        decode_step_argmax = staticmethod(lambda h, w, b: _ops.decode_step_argmax(h, w, b, planes=planes))
    h = torch.zeros(B, H); h[:, 0] = 1.0
    w = torch.zeros(V, H)
    g = torch.Generator().manual_seed(3)
    w[:, 0] = torch.randint(-50, 50, (V,), generator=g).float()
    w[[5, V // 2, V - 1], 0] = 60.0
    ids = ops.decode_step_argmax(h.to(DEV), w.to(DEV), torch.zeros(V).to(DEV)).cpu()
    assert (ids == 5).all()
    hh = _r(B, H, seed=5); ww = _r(V, H, seed=6, scale=H ** -0.5); bb = _r(V, seed=7)
    ids = ops.decode_step_argmax(hh.to(DEV), ww.to(DEV), bb.to(DEV)).cpu()
    ref = (hh.double() @ ww.double().t() + bb.double())
    top2 = ref.topk(2, 1).values
    safe = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert safe.sum() > B // 2
    assert torch.equal(ids[safe], ref.argmax(1)[safe])
    ids = ops.decode_step_argmax(hh.to(DEV), ww.to(DEV), (bb - 1000).to(DEV)).cpu()          # all-negative logits
    assert torch.equal(ids[safe], ref.argmax(1)[safe])


def test_feat_proj_fwd_bwd(lib):
    from s2vt_video_caption_amd import ops
    B, L, Fd, H = 3, 5, 70, 20
    feats, w, bias = _r(B, L, Fd, seed=1), _r(H, Fd, seed=2, scale=0.1), _r(H, seed=3)
    x1 = ops.feat_proj_fwd(feats.to(DEV), w.to(DEV), bias.to(DEV)).cpu()
    ref = (feats @ w.t() + bias).transpose(0, 1).reshape(L * B, H)          # time-major
    assert (x1 - ref).abs().max().item() < 5e-6
    dx1 = _r(L * B, H, seed=4)
    dw, db, dfe = ops.feat_proj_bwd(feats.to(DEV), w.to(DEV), dx1.to(DEV), need_dfeats=True)
    dx_bm = dx1.view(L, B, H).transpose(0, 1)                                # [B,L,H]
    assert (dw.cpu() - torch.einsum("blh,blf->hf", dx_bm, feats)).abs().max().item() < 2e-5
    assert (db.cpu() - dx1.sum(0)).abs().max().item() < 1e-5
    assert (dfe.cpu() - dx_bm @ w).abs().max().item() < 1e-5



@pytest.mark.parametrize("bw", [5, 3, 8])
@pytest.mark.parametrize("ties", [False, True])
def test_device_beam_queues_match_reference_heap(lib, ties, bw):
    """csrc/beam_queue.hip against the literal heap bookkeeping of S2VTModel.py:186-236 (beam.HeapQueues): the same synthetic
    top-20 tables (ascending token ids, random log-probs - small integers in the `ties` case, where WHICH entries pop first
    depends on the binary heap's internal layout) are fed to both, depth by depth; popped tokens, the set of frozen samples at
    every depth and the back-traced sequences must be identical.  <eos> = 2 occurs among the tokens, so entries finish, are
    re-inserted unchanged and samples stop early."""
    import ctypes
    from s2vt_video_caption_amd import beam
    from s2vt_video_caption_amd.functional import _ptr, _stream
    B, D, V, sos, eos = 37, 12, 23, 1, 2            # (bw = 8: the kernel's widest beam, 168 candidates per depth = 3 per lane)
    R = B * bw
    for seed in range(4):
        rng = np.random.default_rng(seed + (100 if ties else 0))
        q = beam.HeapQueues(B, bw, sos, eos)
        nbytes = lib.s2vt_beam_queue_bytes(B, bw, D)
        state = torch.zeros(nbytes, dtype=torch.uint8, device=DEV)
        rows = torch.zeros(3, R, dtype=torch.int32, device=DEV)
        tix = torch.zeros(R, 20, dtype=torch.int32, device=DEV)
        tlp = torch.zeros(R, 20, dtype=torch.float32, device=DEV)

        def qstep(depth):
            from s2vt_video_caption_amd import capi
            capi.check(lib.s2vt_beam_queue_step(B, bw, D, sos, eos, depth, _ptr(state), nbytes, _ptr(tix), _ptr(tlp), _ptr(rows[0]),
                                                _ptr(rows[1]), _ptr(rows[2]), _stream(torch.device(DEV))), "s2vt_beam_queue_step")
        depth = 0
        while depth < D and not q.all_done():
            depth += 1
            rb, rs, rt = q.pop()
            qstep(depth)
            # the device's fixed rows r = b * bw + slot against the host's compact (sample, slot) rows of this depth
            dev_tok = rows[2].cpu().numpy().reshape(B, bw)
            want = np.zeros((B, bw), dtype=np.int64)
            slots = []
            for b in range(B):
                if q.beams[b] is None:
                    continue
                for j, (key, n) in enumerate(q.beams[b]):
                    if not (n.wordid == eos and n.prevNode is not None):
                        want[b, j] = n.wordid
                        slots.append(b * bw + j)
            np.testing.assert_array_equal(dev_tok, want, err_msg="depth %d seed %d" % (depth, seed))
            assert len(slots) == len(rb)
            T_ix = np.sort(np.stack([rng.choice(V, 20, replace=False) for _ in range(R)]), axis=1).astype(np.int32)
            T_lp = (-rng.integers(1, 6, size=(R, 20)).astype(np.float32) if ties else -rng.random((R, 20), dtype=np.float32) * 10)
            tix.copy_(torch.from_numpy(T_ix))
            tlp.copy_(torch.from_numpy(T_lp))
            q.push(T_ix[slots].astype(np.int64) if slots else None, T_lp[slots] if slots else None)
        qstep(0)
        frozen = int(state[:4].view(torch.int32).item())
        assert frozen == sum(q.done)
        out = torch.zeros(B, D + 2, dtype=torch.int32, device=DEV)
        out_len = torch.zeros(B, dtype=torch.int32, device=DEV)
        from s2vt_video_caption_amd import capi
        capi.check(lib.s2vt_beam_queue_result(B, bw, D, _ptr(state), nbytes, _ptr(out), D + 2, _ptr(out_len), _stream(torch.device(DEV))),
                   "s2vt_beam_queue_result")
        got = [out[b, :int(out_len[b])].cpu().tolist() for b in range(B)]
        assert got == q.finish(), seed


def test_beam_step_matches_cell_and_topk(lib):
    """s2vt_beam_step (vid step, gathered word step with in-kernel embedding rows, out_linear, log_softmax + top-20 in
    ascending token order) against the oracle's cell and torch's log_softmax/topk on the CPU."""
    from s2vt_video_caption_amd import ops, synth, capi
    B, L, F, H, E, V = 5, 8, 64, 40, 24, 300
    sd = synth.make_state_dict(V, F, H, E, seed=9)
    params = [sd[k] for k in capi.PARAM_KEYS]
    g = torch.Generator().manual_seed(3)
    R = 11
    row_b = torch.randint(0, B, (R,), generator=g, dtype=torch.int32)
    row_state = torch.randint(0, 7, (R,), generator=g, dtype=torch.int32)
    tok = torch.randint(0, V, (R,), generator=g, dtype=torch.int32)
    vid_h, vid_c = _r(B, H, seed=1, scale=0.5), _r(B, H, seed=2, scale=0.5)
    word_h, word_c = _r(7, H, seed=3, scale=0.5), _r(7, H, seed=4, scale=0.5)
    out = ops.beam_step([p.to(DEV) for p in params], (B, L, F, H, E, V), row_b.to(DEV), row_state.to(DEV), tok.to(DEV),
                        vid_h.to(DEV), vid_c.to(DEV), word_h.to(DEV), word_c.to(DEV))
    vh, vc, wh, wc, tix, tlp = [t.cpu() for t in out]
    # CPU restatement (S2VTModel.py:208-219)
    w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, w_f, b_f, w_o, b_o, emb = params
    rvh, rvc = orc.lstm_cell(torch.zeros(B, H), vid_h, vid_c, w_ih1, w_hh1, b_ih1, b_hh1)
    x = torch.cat([emb[tok.long()], rvh[row_b.long()]], dim=1)
    rwh, rwc = orc.lstm_cell(x, word_h[row_state.long()], word_c[row_state.long()], w_ih2, w_hh2, b_ih2, b_hh2)
    logp = torch.log_softmax(rwh @ w_o.t() + b_o, dim=1)
    rix = logp.topk(20, dim=1).indices.sort(dim=1).values
    assert (vh - rvh).abs().max() < 2e-6 and (vc - rvc).abs().max() < 2e-6
    assert (wh - rwh).abs().max() < 2e-6 and (wc - rwc).abs().max() < 2e-6
    assert torch.equal(tix.long(), rix)
    assert (tlp - logp.gather(1, rix)).abs().max() < 5e-6


def test_beam_step_at_config5_size(lib):
    """s2vt_beam_step at the size BASELINE configs[4] runs it at - R = 128 samples x beam 5 = 640 rows, H = E = 1000,
    V = 12000 (the 64-row step-kernel instantiation, the split-K logits GEMM, the 640-row top-20 kernel) - against torch-CPU
    fp32: cell states, the 20 token ids per row (ascending) and their log-probs.  A token may only differ from torch's
    top-20 where the 20th and 21st log-probs are closer than the stated log-prob tolerance."""
    from s2vt_video_caption_amd import ops, synth, capi
    B, L, F, H, E, V = 128, 80, 4096, 1000, 1000, 12000
    sd = synth.make_state_dict(V, F, H, E, seed=13, out_scale=16.0)
    params = [sd[k] for k in capi.PARAM_KEYS]
    g = torch.Generator().manual_seed(7)
    R, S = 640, 640
    row_b = torch.arange(B, dtype=torch.int32).repeat_interleave(5)
    row_state = torch.randperm(S, generator=g).to(torch.int32)
    tok = torch.randint(0, V, (R,), generator=g, dtype=torch.int32)
    vid_h, vid_c = _r(B, H, seed=1, scale=0.5), _r(B, H, seed=2, scale=0.5)
    word_h, word_c = _r(S, H, seed=3, scale=0.5), _r(S, H, seed=4, scale=0.5)
    out = ops.beam_step([p.to(DEV) for p in params], (B, L, F, H, E, V), row_b.to(DEV), row_state.to(DEV), tok.to(DEV),
                        vid_h.to(DEV), vid_c.to(DEV), word_h.to(DEV), word_c.to(DEV))
    vh, vc, wh, wc, tix, tlp = [t.cpu() for t in out]
    w_ih1, w_hh1, b_ih1, b_hh1, w_ih2, w_hh2, b_ih2, b_hh2, w_f, b_f, w_o, b_o, emb = params
    rvh, rvc = orc.lstm_cell(torch.zeros(B, H), vid_h, vid_c, w_ih1, w_hh1, b_ih1, b_hh1)          # S2VTModel.py:208-210
    x = torch.cat([emb[tok.long()], rvh[row_b.long()]], dim=1)
    rwh, rwc = orc.lstm_cell(x, word_h[row_state.long()], word_c[row_state.long()], w_ih2, w_hh2, b_ih2, b_hh2)   # :211-212
    logp = torch.log_softmax(rwh @ w_o.t() + b_o, dim=1)                                           # :213-214
    assert (vh - rvh).abs().max() < 5e-6 and (vc - rvc).abs().max() < 5e-6
    assert (wh - rwh).abs().max() < 5e-6 and (wc - rwc).abs().max() < 5e-6
    top21 = logp.topk(21, dim=1)
    rix = top21.indices[:, :20].sort(dim=1).values
    tol = 5e-5                      # |logits| reach ~25 at out_scale 16: 2e-6 relative
    cut_gap = top21.values[:, 19] - top21.values[:, 20]
    same = (tix.long() == rix).all(dim=1)
    assert bool((cut_gap[~same] < tol).all()), (int((~same).sum()), cut_gap[~same])
    assert int((~same).sum()) <= 2
    assert (tix[:, 1:] > tix[:, :-1]).all()                                                        # ascending token order
    assert (tlp - logp.gather(1, tix.long())).abs().max() < tol


# ---------------------------------------------------------------------------------------------- config-3 (bf16) kernels
def _bf16r(x):
    return x.to(torch.bfloat16).to(torch.float64)


def _check_seq_bf16_teacher_forced(h_all, c_all, gates, gx, n_gx, bias, w_hh, T, B, H, tol=2e-5):
    """Every step checked on its own against fp64 math on the bf16-rounded operands the kernel itself saw: the h the
    kernel wrote for step t-1 (rounded to bf16 as the kernel rounds it) and its fp32 c_{t-1}.  A wrong k mapping, a
    stale or torn hand-off of h_{t-1}, a lost row or column shows as an O(0.1) error; fp32 accumulation order as ~1e-6."""
    wb = _bf16r(w_hh.cpu())
    h_all, c_all, gates = h_all.cpu().double(), c_all.cpu().double(), gates.cpu().double()
    gx, bias = gx.cpu().double(), bias.cpu().double()
    worst = 0.0
    for t in range(T):
        hp = _bf16r(h_all[(t - 1) * B:t * B].float()) if t else torch.zeros(B, H, dtype=torch.float64)
        cp = c_all[(t - 1) * B:t * B] if t else torch.zeros(B, H, dtype=torch.float64)
        pre = (gx[t * B:(t + 1) * B] if t < n_gx else bias[None, :]) + hp @ wb.t()
        i, f, g, o = pre.chunk(4, dim=1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * cp + i * g
        h = o * torch.tanh(c)
        worst = max(worst, (h_all[t * B:(t + 1) * B] - h).abs().max().item(), (c_all[t * B:(t + 1) * B] - c).abs().max().item(),
                    (gates[t * B:(t + 1) * B] - torch.cat([i, f, g, o], dim=1)).abs().max().item())
    assert worst < tol, worst
    return worst


@pytest.mark.parametrize("T,B,H,n_gx", [(3, 37, 1000, 2), (4, 5, 72, 4), (3, 129, 1000, 3), (5, 64, 200, 3)])
def test_bf16_step_kernels_h1000_odd_batch(lib, T, B, H, n_gx):
    """lstm_step_fwd_bf16_kernel at the config-3 hidden size (H = 1000: k padded to 1024, gate tiles ragged) and batch
    sizes that are not multiples of any tile."""
    from s2vt_video_caption_amd import ops
    gx, bias, w = _r(T * B, 4 * H, seed=11), _r(4 * H, seed=12, scale=0.3), _r(4 * H, H, seed=13, scale=H ** -0.5)
    h_all, c_all, gates = ops.lstm_seq_fwd_bf16(gx.to(DEV), n_gx, bias.to(DEV), w.to(DEV), T, B, H)
    _check_seq_bf16_teacher_forced(h_all, c_all, gates, gx, n_gx, bias, w, T, B, H)


@pytest.mark.parametrize("T,B,H,n_gx,block", [(6, 32, 128, 4, 0), (7, 128, 1000, 4, 3), (5, 256, 1000, 3, 0),
                                              (9, 96, 520, 9, 4), (12, 256, 1000, 6, 5), (4, 512, 1000, 2, 0)])
def test_persistent_bf16_recurrence(lib, T, B, H, n_gx, block):
    """lstm_seq_fwd_bf16_persist_kernel (one launch per block of steps, W_hh slice resident per CU, cross-workgroup
    hand-off of h_t): every step teacher-forced against fp64 math, the run repeated bit for bit (a stale hand-off is
    timing dependent), and against the one-launch-per-step kernels."""
    from s2vt_video_caption_amd import ops
    gx, bias, w = _r(T * B, 4 * H, seed=21), _r(4 * H, seed=22, scale=0.3), _r(4 * H, H, seed=23, scale=H ** -0.5)
    args = (gx.to(DEV), n_gx, bias.to(DEV), w.to(DEV), T, B, H)
    h1, c1, g1 = ops.lstm_seq_fwd_bf16(*args, persistent=True, block=block)
    _check_seq_bf16_teacher_forced(h1, c1, g1, gx, n_gx, bias, w, T, B, H)
    h2, c2, g2 = ops.lstm_seq_fwd_bf16(*args, persistent=True, block=block)
    assert torch.equal(h1, h2) and torch.equal(c1, c2) and torch.equal(g1, g2)
    h0, c0, g0 = ops.lstm_seq_fwd_bf16(*args, persistent=False)
    # same arithmetic, different fp32 summation order; a bf16 rounding flip of one h value moves a later pre-activation
    # by ~|w| * 2^-9 |h|: bounded far below a structural error
    assert (h1 - h0).abs().max().item() < 2e-3 and (c1 - c0).abs().max().item() < 2e-3


def test_persistent_bf16_recurrence_under_load(lib):
    """The hand-off again while another stream keeps the memory system and the other compute units busy (uneven load is
    what exposes a missing release / acquire): result identical to the quiet run, every step teacher-forced."""
    from s2vt_video_caption_amd import ops
    T, B, H, n_gx = 40, 256, 1000, 20
    gx, bias, w = _r(T * B, 4 * H, seed=31), _r(4 * H, seed=32, scale=0.3), _r(4 * H, H, seed=33, scale=H ** -0.5)
    args = (gx.to(DEV), n_gx, bias.to(DEV), w.to(DEV), T, B, H)
    quiet = ops.lstm_seq_fwd_bf16(*args, persistent=True, block=16)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(30):
            big = big * 1.0001 + 1.0          # 512 MB of traffic per pass on the CUs the recurrence leaves free
    loaded = ops.lstm_seq_fwd_bf16(*args, persistent=True, block=16)
    torch.cuda.synchronize()
    for a, b in zip(quiet, loaded):
        assert torch.equal(a, b)
    _check_seq_bf16_teacher_forced(*loaded, gx, n_gx, bias, w, T, B, H)


_CORESIDENCY_CHILD = r"""
import ctypes, sys, time
import torch
sys.path.insert(0, sys.argv[1])
from s2vt_video_caption_amd import capi, ops
hold_us = int(sys.argv[2])
lib = capi.load()
DEV = "cuda:0"
def r(*shape, seed, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)
T, B, H, n_gx = 10, 256, 1000, 5
a = (r(T * B, 4 * H, seed=1), r(T * B, 4 * H, seed=2), n_gx, r(4 * H, seed=3, scale=0.3), r(4 * H, seed=4, scale=0.3),
     r(4 * H, H, seed=5, scale=H ** -0.5), r(4 * H, H, seed=6, scale=H ** -0.5), T, B, H)
quiet = ops.lstm_seq_fwd_bf16_pair(*a, block=5)          # 504 workgroups, two per compute unit
torch.cuda.synchronize()
side = torch.cuda.Stream()
# one workgroup per compute unit holding 84 KB of LDS (two do not fit one CU; 84 + 74 KB leave room for ONE workgroup of
# the persistent launch beside it instead of two): half of the launch cannot become resident while this kernel runs
capi.check(lib.s2vt_test_occupy_cus(256, 84 * 1024, hold_us, ctypes.c_void_p(side.cuda_stream)), "occupy")
time.sleep(0.005)
t0 = time.time()
try:
    loaded = ops.lstm_seq_fwd_bf16_pair(*a, block=5)
    torch.cuda.synchronize()
except capi.S2VTHipError as e:
    print("CLEAN_TIMEOUT after %.2f s: %s" % (time.time() - t0, e))
    torch.cuda.synchronize()
    sys.exit(3)
same = all(torch.equal(x, y) for p, q in zip(quiet, loaded) for x, y in zip(p, q))
print("COMPLETED in %.2f s, identical=%s" % (time.time() - t0, same))
sys.exit(0 if same else 4)
"""


@pytest.mark.parametrize("hold_ms,allowed", [(30, (0,)), (2500, (0, 3))])
def test_persistent_launch_beside_an_lds_holding_kernel(lib, hold_ms, allowed):
    """A persistent launch whose workgroups cannot all become resident: a foreign kernel (the stand-in for an RCCL kernel
    on a communication stream) holds 84 KB of LDS on every compute unit, so only one of the two workgroups per CU fits.
    Short hold (30 ms, far below the 1-s spin bound): the launch must complete with the bits of the quiet run once the
    foreign kernel has left.  Long hold (2.5 s): either that, or a CLEAN time-out - the library reports S2VT_ERR_TIMEOUT
    (the child exits 3) - never a hang and never a silently wrong result.  Runs in a child process under a hard time limit,
    once."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _CORESIDENCY_CHILD, root, str(hold_ms * 1000)], capture_output=True, text=True,
                       timeout=180)
    print(r.stdout, r.stderr[-2000:])
    assert r.returncode in allowed, (r.returncode, r.stdout, r.stderr[-2000:])


def test_persistent_bf16_recurrence_two_layers_one_launch(lib):
    """Two layers side by side in ONE persistent launch (504 workgroups, two per CU at B = 256): each must equal its own
    single-layer launch bit for bit, and both pass the teacher-forced check."""
    from s2vt_video_caption_amd import ops
    T, B, H, n_gx = 20, 256, 1000, 12
    ins = []
    for k in range(2):
        ins.append((_r(T * B, 4 * H, seed=41 + k), _r(4 * H, seed=43 + k, scale=0.3), _r(4 * H, H, seed=45 + k, scale=H ** -0.5)))
    (gx0, b0, w0), (gx1, b1, w1) = ins
    pair = ops.lstm_seq_fwd_bf16_pair(gx0.to(DEV), gx1.to(DEV), n_gx, b0.to(DEV), b1.to(DEV), w0.to(DEV), w1.to(DEV), T, B, H, block=8)
    for (gx, b, w), got in zip(ins, pair):
        solo = ops.lstm_seq_fwd_bf16(gx.to(DEV), n_gx, b.to(DEV), w.to(DEV), T, B, H, persistent=True, block=8)
        for x, y in zip(got, solo):
            assert torch.equal(x, y)
        _check_seq_bf16_teacher_forced(*got, gx, n_gx, b, w, T, B, H)


def _bptt_bf16_reference(w_hh, dh_out, dh_first, c_all, gates, T, B, H):
    """fp64 BPTT with the kernels' operand rounding: W_hh and the dG fed to the next (earlier) step rounded to bf16."""
    wb = _bf16r(w_hh.cpu())                       # [4H, H]
    c_all, gates = c_all.cpu().double(), gates.cpu().double()
    dh_out = dh_out.cpu().double() if dh_out is not None else None
    dG = torch.zeros(T * B, 4 * H, dtype=torch.float64)
    dc = torch.zeros(B, H, dtype=torch.float64)
    nxt = None
    for t in range(T - 1, -1, -1):
        dh = torch.zeros(B, H, dtype=torch.float64)
        if nxt is not None:
            dh += _bf16r(nxt.float()) @ wb
        if dh_out is not None and t >= dh_first:
            dh += dh_out[(t - dh_first) * B:(t - dh_first + 1) * B]
        i, f, g, o = gates[t * B:(t + 1) * B].chunk(4, dim=1)
        c = c_all[t * B:(t + 1) * B]
        cp = c_all[(t - 1) * B:t * B] if t else torch.zeros_like(c)
        tc = torch.tanh(c)
        dct = dh * o * (1 - tc * tc) + dc
        nxt = torch.cat([dct * g * i * (1 - i), dct * cp * f * (1 - f), dct * i * (1 - g * g), dh * tc * o * (1 - o)], dim=1)
        dG[t * B:(t + 1) * B] = nxt
        dc = dct * f
    return dG


def _bptt_inputs(T, B, H, seed):
    w = _r(4 * H, H, seed=seed, scale=H ** -0.5)
    gates = torch.sigmoid(_r(T * B, 4 * H, seed=seed + 1))
    gates[:, 2 * H:3 * H] = torch.tanh(_r(T * B, H, seed=seed + 2))
    c_all = _r(T * B, H, seed=seed + 3, scale=0.7)
    return w, gates, c_all


@pytest.mark.parametrize("T,B,H,dh_first", [(3, 37, 1000, 1), (4, 129, 1000, 0), (5, 5, 72, 2)])
def test_bf16_bptt_step_kernels_h1000_odd_batch(lib, T, B, H, dh_first):
    """lstm_step_bwd_bf16_kernel at H = 1000 (k = 4000 padded to 4032) and ragged batches against fp64 math on
    bf16-rounded operands.  Bound: a bf16 re-rounding flip of one dG element (2^-9 relative) seen through W (|w| ~ 0.03)."""
    from s2vt_video_caption_amd import ops
    w, gates, c_all = _bptt_inputs(T, B, H, 50)
    dh = _r((T - dh_first) * B, H, seed=59, scale=0.1)
    got = ops.lstm_seq_bwd_bf16(w.to(DEV), dh.to(DEV), dh_first, c_all.to(DEV), gates.to(DEV), T, B, H).cpu().double()
    ref = _bptt_bf16_reference(w, dh, dh_first, c_all, gates, T, B, H)
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-7


@pytest.mark.parametrize("bptt_units", [0, 16, 32])
@pytest.mark.parametrize("T,B,H,dh_first,block", [(6, 32, 128, 2, 0), (7, 128, 1000, 0, 3), (5, 256, 1000, 1, 0), (9, 96, 520, 3, 4)])
def test_persistent_bf16_bptt(lib, T, B, H, dh_first, block, bptt_units):
    """lstm_seq_bwd_bf16_persist_kernel against the fp64 reference, the launch-per-timestep kernels, and itself (bitwise) - in
    both workgroup shapes (option bptt_units: 32 hidden units x 8 waves, one workgroup per compute unit, and the 16-unit x 4-wave
    fallback for shapes whose two layers do not fit the device that way; 0 = the launcher's rule)."""
    from s2vt_video_caption_amd import ops
    w, gates, c_all = _bptt_inputs(T, B, H, 60)
    dh = _r((T - dh_first) * B, H, seed=69, scale=0.1)
    args = (w.to(DEV), dh.to(DEV), dh_first, c_all.to(DEV), gates.to(DEV), T, B, H)
    prev = lib.s2vt_set_option(b"bptt_units", bptt_units)
    try:
        got = ops.lstm_seq_bwd_bf16(*args, persistent=True, block=block)
        again = ops.lstm_seq_bwd_bf16(*args, persistent=True, block=block)
    finally:
        lib.s2vt_set_option(b"bptt_units", prev)
    assert torch.equal(got, again)
    ref = _bptt_bf16_reference(w, dh, dh_first, c_all, gates, T, B, H)
    scale = ref.abs().max().item()
    assert (got.cpu().double() - ref).abs().max().item() < 2e-3 * scale + 1e-7
    per_step = ops.lstm_seq_bwd_bf16(*args, persistent=False)
    assert (got - per_step).abs().max().item() < 2e-3 * scale + 1e-7


def test_persistent_bf16_bptt_two_layers_one_launch_under_load(lib):
    from s2vt_video_caption_amd import ops
    T, B, H, dh_first = 24, 256, 1000, 4
    ins = [_bptt_inputs(T, B, H, 70 + 10 * k) for k in range(2)]
    dhs = [_r((T - dh_first) * B, H, seed=79 + k, scale=0.1) for k in range(2)]
    dev = [tuple(x.to(DEV) for x in i) for i in ins]
    ddh = [d.to(DEV) for d in dhs]
    solo = [ops.lstm_seq_bwd_bf16(dev[k][0], ddh[k], dh_first, dev[k][2], dev[k][1], T, B, H, persistent=True, block=8) for k in range(2)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.randn(32 * 1024 * 1024, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(20):
            big = big * 1.0001 + 1.0
    pair = ops.lstm_seq_bwd_bf16_pair(dev[0][0], dev[1][0], ddh[0], ddh[1], dh_first, dev[0][2], dev[1][2], dev[0][1], dev[1][1],
                                      T, B, H, block=8)
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(pair[k], solo[k])
        ref = _bptt_bf16_reference(ins[k][0], dhs[k], dh_first, ins[k][2], ins[k][1], T, B, H)
        assert (pair[k].cpu().double() - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-7


# ------------------------------------------------------------------------------------- fp32 persistent recurrence
def _cell_seq_fp64(gx, n_gx, bias, w, T, B, H):
    gx, bias, w = gx.double(), bias.double(), w.double()
    h = torch.zeros(B, H, dtype=torch.float64)
    c = torch.zeros(B, H, dtype=torch.float64)
    hs, cs, gs = [], [], []
    for t in range(T):
        pre = (gx[t * B:(t + 1) * B] if t < n_gx else bias[None, :]) + h @ w.t()
        i, f, g, o = pre.chunk(4, dim=1)
        i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
        c = f * c + i * g
        h = o * torch.tanh(c)
        hs.append(h); cs.append(c); gs.append(torch.cat([i, f, g, o], dim=1))
    return torch.cat(hs), torch.cat(cs), torch.cat(gs)


@pytest.mark.parametrize("T,B,H,n_gx,block", [(6, 32, 128, 4, 0), (7, 64, 1000, 4, 3), (5, 128, 1000, 3, 0), (9, 96, 520, 9, 4),
                                              (4, 256, 1000, 2, 0), (5, 32, 8, 5, 2), (12, 64, 1000, 6, 5), (5, 64, 1024, 2, 0),
                                              (4, 32, 70, 1, 2), (5, 64, 37, 3, 2), (3, 32, 999, 1, 0)])
def test_persistent_split_precision_recurrence(lib, T, B, H, n_gx, block):
    """lstm_seq_fwd_x3_persist_kernel: h_{t-1} . W_hh^T as six bf16 plane products (fp32-equivalent), W_hh planes resident in
    384 registers per lane, hand-off of h_t as three bf16 planes.  Bounds of the fp32 launch-per-timestep kernel: against fp64
    cell math, the launch-per-timestep kernels, and itself bit for bit.  The workspace is handed over full of bf16 NaN patterns:
    a plane element read before it was written (k padding, the pad columns of the last column slice) would poison the result."""
    from s2vt_video_caption_amd import ops
    gx, bias, w = _r(n_gx * B, 4 * H, seed=81), _r(4 * H, seed=82, scale=0.3), _r(4 * H, H, seed=83, scale=H ** -0.5)
    args = (T, B, gx.to(DEV), n_gx, bias.to(DEV), w.to(DEV))
    h1, c1, g1 = ops.lstm_seq_fwd_persist(*args, block=block)
    h2, c2, g2 = ops.lstm_seq_fwd_persist(*args, block=block)
    assert torch.equal(h1, h2) and torch.equal(c1, c2) and torch.equal(g1, g2)
    rh, rc, rg = _cell_seq_fp64(gx, n_gx, bias, w, T, B, H)
    assert (h1.cpu().double() - rh).abs().max().item() < 2e-6
    assert (c1.cpu().double() - rc).abs().max().item() < 4e-6
    assert (g1.cpu().double() - rg).abs().max().item() < 2e-6
    h0, c0, g0 = ops.lstm_seq_fwd(*args, want_stash=True)
    assert (h1 - h0).abs().max().item() < 2e-6


def test_persistent_split_precision_two_layers_one_launch_under_load(lib):
    """Two layers per launch at the config-2 shape (252 workgroups, one per compute unit) while another stream loads the chip:
    each layer equals its solo run bit for bit (a stale hand-off is timing dependent)."""
    from s2vt_video_caption_amd import ops
    T, B, H, n_gx = 24, 64, 1000, 12
    ins = [(_r(n_gx * B, 4 * H, seed=111 + k), _r(4 * H, seed=113 + k, scale=0.3), _r(4 * H, H, seed=115 + k, scale=H ** -0.5))
           for k in range(2)]
    dev = [tuple(x.to(DEV) for x in i) for i in ins]
    solo = [ops.lstm_seq_fwd_persist(T, B, dev[k][0], n_gx, dev[k][1], dev[k][2], block=9) for k in range(2)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.randn(32 * 1024 * 1024, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(20):
            big = big * 1.0001 + 1.0
    pair = ops.lstm_seq_fwd_persist(T, B, dev[0][0], n_gx, dev[0][1], dev[0][2], block=9, second=dev[1])
    torch.cuda.synchronize()
    for k in range(2):
        for a, b in zip(pair[k], solo[k]):
            assert torch.equal(a, b)
        rh, rc, rg = _cell_seq_fp64(*[ins[k][0], n_gx, ins[k][1], ins[k][2]], T, B, H)
        assert (pair[k][0].cpu().double() - rh).abs().max().item() < 2e-6


def _bptt_fp64(w, gates, c_all, dh, dh_first, T, B, H):
    wd, cd, gd, dhd = w.double(), c_all.double(), gates.double(), dh.double()
    ref = torch.zeros(T * B, 4 * H, dtype=torch.float64)
    dc = torch.zeros(B, H, dtype=torch.float64)
    nxt = None
    for t in range(T - 1, -1, -1):
        d = torch.zeros(B, H, dtype=torch.float64)
        if nxt is not None:
            d += nxt @ wd
        if t >= dh_first:
            d += dhd[(t - dh_first) * B:(t - dh_first + 1) * B]
        i, f, g, o = gd[t * B:(t + 1) * B].chunk(4, dim=1)
        c = cd[t * B:(t + 1) * B]
        cp = cd[(t - 1) * B:t * B] if t else torch.zeros_like(c)
        tc = torch.tanh(c)
        dct = d * o * (1 - tc * tc) + dc
        nxt = torch.cat([dct * g * i * (1 - i), dct * cp * f * (1 - f), dct * i * (1 - g * g), d * tc * o * (1 - o)], dim=1)
        ref[t * B:(t + 1) * B] = nxt
        dc = dct * f
    return ref


@pytest.mark.parametrize("T,B,H,dh_first,block", [(6, 32, 128, 2, 0), (7, 64, 1000, 0, 3), (5, 128, 1000, 1, 0), (9, 96, 520, 3, 4),
                                                  (4, 256, 1000, 0, 0), (5, 32, 8, 0, 2), (12, 64, 1000, 5, 5), (5, 64, 1024, 2, 0),
                                                  (4, 32, 70, 1, 2), (5, 64, 37, 3, 2), (3, 32, 999, 1, 0)])
def test_persistent_split_precision_bptt(lib, T, B, H, dh_first, block):
    """lstm_seq_bwd_x3_persist_kernel: the contraction dG_{t+1} . W_hh split over the gate columns (every workgroup multiplies its
    own dG tile with its 64 rows of W_hh, planes resident in registers, and the fp32 partial sums are scattered / gathered per
    consumer in a fixed order).  Bounds of the fp32 launch-per-timestep kernel: fp64 BPTT, the launch-per-timestep kernels,
    itself bit for bit; the workspace arrives full of NaN patterns."""
    from s2vt_video_caption_amd import ops
    w, gates, c_all = _bptt_inputs(T, B, H, 90)
    dh = _r((T - dh_first) * B, H, seed=99, scale=0.1)
    args = (T, B, w.to(DEV), dh.to(DEV), dh_first, c_all.to(DEV), gates.to(DEV))
    got = ops.lstm_seq_bwd_persist(*args, block=block)
    again = ops.lstm_seq_bwd_persist(*args, block=block)
    assert torch.equal(got, again)
    ref = _bptt_fp64(w, gates, c_all, dh, dh_first, T, B, H)
    scale = ref.abs().max().item()
    assert (got.cpu().double() - ref).abs().max().item() < 4e-6 * scale + 1e-9
    per_step = ops.lstm_seq_bwd(T, B, w.to(DEV), dh.to(DEV), dh_first, c_all.to(DEV), gates.to(DEV).clone())
    assert (got - per_step).abs().max().item() < 4e-6 * scale + 1e-9


def test_persistent_split_precision_bptt_two_layers_one_launch_under_load(lib):
    """Two layers per launch at the config-2 shape while another stream loads the chip: each layer equals its solo run bit for
    bit (a stale or early hand-off of the partial sums is timing dependent)."""
    from s2vt_video_caption_amd import ops
    T, B, H, dh_first = 20, 64, 1000, 6
    ins = [_bptt_inputs(T, B, H, 170 + 10 * k) for k in range(2)]
    dhs = [_r((T - dh_first) * B, H, seed=179 + k, scale=0.1) for k in range(2)]
    dev = [tuple(x.to(DEV) for x in i) for i in ins]
    ddh = [d.to(DEV) for d in dhs]
    solo = [ops.lstm_seq_bwd_persist(T, B, dev[k][0], ddh[k], dh_first, dev[k][2], dev[k][1], block=7) for k in range(2)]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.randn(32 * 1024 * 1024, device=DEV)
    with torch.cuda.stream(side):
        for _ in range(20):
            big = big * 1.0001 + 1.0
    pair = ops.lstm_seq_bwd_persist(T, B, dev[0][0], ddh[0], dh_first, dev[0][2], dev[0][1], block=7,
                                    second=(dev[1][0], ddh[1], dev[1][2], dev[1][1]))
    torch.cuda.synchronize()
    for k in range(2):
        assert torch.equal(pair[k], solo[k])
        ref = _bptt_fp64(ins[k][0], ins[k][1], ins[k][2], dhs[k], dh_first, T, B, H)
        assert (pair[k].cpu().double() - ref).abs().max().item() < 4e-6 * ref.abs().max().item() + 1e-9


