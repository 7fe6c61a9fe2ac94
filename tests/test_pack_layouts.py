"""Host-side packing of operands into MFMA fragment order (coponerf_amd/render.py) against the element maps the kernels document
(include/coponerf_hip.h: cpn_lightfield_decode wpack, cpn_encode_key kwring / kh_units).  CPU only."""
import torch

from coponerf_amd.render import frag_order_f32, pack_key_ring


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


def test_unit_rows_is_the_encoders_row_map():
    """render.unit_rows against the unit / row arithmetic of csrc/encode_fused.hip and csrc/local_units.hip restated as loops: unit
    u = ((ray group - first group) * V + view) * ceil(S/4) + sample block; slot c = (sample & 3) * 4 + (ray & 3); rows of a partial
    unit outside the ray range (or past R / S) are dead; every live row appears exactly once."""
    from coponerf_amd.render import rows_from_unit_order, unit_rows
    V = 2
    for B, R, S, ray0, n in ((1, 8, 8, 0, 8), (2, 7, 6, 3, 9), (3, 5, 9, 4, 7), (1, 64, 32, 13, 40), (2, 10, 4, 9, 3)):
        gpb, nsblk = (R + 3) // 4, (S + 3) // 4
        b_lo, b_hi = ray0 // R, (ray0 + n - 1) // R
        g0 = b_lo * gpb + (ray0 - b_lo * R) // 4
        g1 = b_hi * gpb + (ray0 + n - 1 - b_hi * R) // 4
        want = []
        for gq in range(g0, g1 + 1):
            b, rg = gq // gpb, gq % gpb
            for v in range(V):
                for sb in range(nsblk):
                    for c in range(16):
                        r, s = rg * 4 + (c & 3), sb * 4 + (c >> 2)
                        ray = b * R + r
                        live = r < R and s < S and ray0 <= ray < ray0 + n
                        want.append(((ray - ray0) * V + v) * S + s if live else -1)
        got = unit_rows(B, R, S, ray0, n).tolist()
        assert got == want, (B, R, S, ray0, n)
        assert sorted(x for x in got if x >= 0) == list(range(n * V * S))
        # and the inverse used by the tests: a unit-order matrix built from rows comes back as those rows
        rows = n * V * S
        x = torch.randn(rows, 128).half()
        idx = torch.tensor(got)
        units = len(got) // 16
        tmp = torch.zeros(units * 16, 128, dtype=torch.float16)
        tmp[idx >= 0] = x[idx[idx >= 0]]
        xu = tmp.view(units, 16, 4, 4, 8).permute(0, 2, 3, 1, 4).contiguous()
        assert torch.equal(rows_from_unit_order(xu, B, R, S, ray0, n), x)
