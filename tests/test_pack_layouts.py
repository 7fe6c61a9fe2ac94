"""Host-side packing of operands into MFMA fragment order (coponerf_amd/render.py) against the element maps the kernels document
(include/coponerf_hip.h: cpn_lightfield_decode wpack, cpn_encode_key group 4, cpn_local_mlp rows_frag).  CPU only."""
import torch

from coponerf_amd.render import frag_order_f32, pack_key_ring, rows_from_frag_order


def test_frag_order_f32_matches_the_documented_element_map():
    torch.manual_seed(0)
    for n, k in ((128, 416), (128, 32), (16, 128)):
        m = torch.randn(n, k)
        flat = frag_order_f32(m).reshape(-1)
        nkb = k // 16
        for t, kb, lane, e in ((0, 0, 0, 0), (n // 16 - 1, nkb - 1, 63, 3), (n // 32, nkb // 2, 37, 2), (0, nkb - 1, 16, 1)):
            want = m[16 * t + (lane & 15), 16 * kb + 4 * (lane >> 4) + e]
            assert flat[((t * nkb + kb) * 64 + lane) * 4 + e] == want
        assert flat.numel() == n * k


def test_pack_key_ring_matches_the_documented_element_map():
    torch.manual_seed(1)
    wk = torch.randn(128, 1664).half()
    flat = pack_key_ring(wk).reshape(-1)
    assert flat.numel() == wk.numel()
    for j, n, t, k, lane, e in ((0, 0, 0, 0, 0, 0), (1, 12, 7, 1, 63, 7), (1, 0, 3, 0, 21, 5), (0, 5, 6, 1, 48, 2)):
        step, piece = j * 13 + n, t * 2 + k
        want = wk[16 * t + (lane & 15), 832 * j + 64 * n + 32 * k + 8 * (lane >> 4) + e]
        assert flat[((step * 16 + piece) * 64 + lane) * 8 + e] == want


def test_rows_from_frag_order_inverts_the_accumulator_layout():
    torch.manual_seed(2)
    rows = 16 * 5
    x = torch.randn(rows, 128).half()
    # [group][32-column block p][lane = row + 16 * 8-column group fg][8]: element (row, p * 32 + fg * 8 + e)
    packed = x.reshape(rows // 16, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
    flat = packed.reshape(-1)
    for row, col in ((0, 0), (17, 45), (79, 127), (33, 64)):
        g, a, p, fg, e = row // 16, row % 16, col // 32, (col % 32) // 8, col % 8
        assert flat[(((g * 4 + p) * 64) + (a + 16 * fg)) * 8 + e] == x[row, col]
    assert torch.equal(rows_from_frag_order(packed, rows), x)
    assert torch.equal(rows_from_frag_order(packed, rows - 3), x[:rows - 3])
