"""CPU tier: the index algebra of the SH backward on the matrix pipe (raster_bwd.hip: raster_backward_mfma_sh_kernel), run as a
NumPy model of the wave's 64 lanes against the plain per-pixel recursion it replaces.

The kernel rests on three layout facts that no compiler checks: (1) with lane l = (Gaussian l & 15, pixel quad l >> 4) the
fp32 MFMA v_mfma_f32_16x16x4_f32 (A[l & 15][l >> 4], B[l >> 4][l & 15], D[4 (l >> 4) + reg][l & 15]: cdna_hip_programming.md)
puts the colour logit of (Gaussian, pixel 4 (l >> 4) + reg) into register `reg` of exactly that lane; (2) the same lanes'
registers, fed back as B with the SH table read transposed as A, accumulate dL/dcoef[g][4 (l >> 4) + reg] over the pixel
quads, the quad's pixels and the pixel rows with no cross-lane reduction; (3) the transmittance / rho recursions over the
16 Gaussians of a group are an exclusive product / inclusive sum over the 16 lanes of a DPP row, with UNMASKED alphas in
the product and the reference's stop (gaussian.cu:906) as a mask.  The model below performs exactly the kernel's lane-level
steps for one group of 16 Gaussians on one 16 x 16 tile and is compared with the sequential front-to-back loop."""
import numpy as np

T_STOP = np.float32(1e-4)


def mfma_16x16x4(a, b, c):
    """D = A B + C in the lane layout of v_mfma_f32_16x16x4_f32: a, b [64]; c [64][4] (lane, register)."""
    A = np.zeros((16, 4))
    B = np.zeros((4, 16))
    for l in range(64):
        A[l & 15, l >> 4] = a[l]
        B[l >> 4, l & 15] = b[l]
    D = A @ B
    out = c.copy()
    for l in range(64):
        for r in range(4):
            out[l, r] += D[4 * (l >> 4) + r, l & 15]
    return out


def row_scan(v, op, identity):
    """Inclusive scan over the 16 lanes of every DPP row (row_shr 1, 2, 4, 8; lanes outside the row keep the identity)."""
    v = v.copy()
    for sh in (1, 2, 4, 8):
        src = np.full_like(v, identity)
        for l in range(64):
            if (l & 15) >= sh:
                src[l] = v[l - sh]
        v = op(v, src)
    return v


def test_lane_model_equals_the_sequential_recursion():
    rng = np.random.default_rng(5)
    NB = 16
    sh = rng.normal(size=(256, NB))                       # sh'_k of pixel 16 y + x
    coef = rng.normal(size=(16, 3, NB)) * 0.5             # the group's 16 Gaussians
    G = rng.normal(size=(3, 256))                         # dL/dC per pixel
    alpha_raw = rng.uniform(0.0, 0.9, size=(16, 256))     # G(pixel) sigma(opa) of (Gaussian, pixel)
    alpha_raw[5] = rng.uniform(0.95, 0.999, size=256)     # an opaque one: pixels stop inside the group
    T_in = rng.uniform(0.0, 1.0, size=256)
    T_in[::7] = 5e-5                                       # pixels that had stopped before the group
    T_in[32:48] = 5e-5                                     # ... and a whole pixel row of them
    rho_in = rng.normal(size=256)

    # ---- reference: pixel by pixel, front to back
    dcoef_ref = np.zeros((16, 3, NB))
    s_ref = np.zeros((16, 256))
    T_ref, rho_ref = T_in.copy(), rho_in.copy()
    for p in range(256):
        T, rho = T_in[p], rho_in[p]
        for g in range(16):
            live = T > T_STOP
            a = alpha_raw[g, p] if live else 0.0
            w = a * T
            c = 1.0 / (1.0 + np.exp2(coef[g] @ sh[p]))     # [3]; the table is pre-scaled: sigma(x) = 1 / (1 + 2^x')
            gc = float(G[:, p] @ c)
            rho = rho - w * gc
            d_alpha = (T * gc - rho / (1.0 - alpha_raw[g, p] + 1e-7)) if live else 0.0
            s_ref[g, p] = d_alpha * a
            dcoef_ref[g] += (G[:, p] * w * c * (1.0 - c))[:, None] * sh[p][None, :]
            T = T - w
        T_ref[p], rho_ref[p] = T, rho

    # ---- the kernel's lanes: lane l = (Gaussian gq = l & 15, pixel quad jq = l >> 4), pixel row s per step
    lanes = np.arange(64)
    gq, jq = lanes & 15, lanes >> 4
    acc = np.zeros((3, 64, 4))                             # MFMA accumulators of the coefficient sums
    s_model = np.zeros((16, 256))
    T_state, rho_state = T_in.copy(), rho_in.copy()
    for s in range(16):
        prow = 16 * s
        # colour logits: A = sh'_(4 kk + jq)(pixel prow + (l & 15)), B = coef[gq][ch][4 kk + jq]
        lg = np.zeros((3, 64, 4))
        for kk in range(NB // 4):
            a_op = sh[prow + (lanes & 15), 4 * kk + jq]
            for ch in range(3):
                lg[ch] = mfma_16x16x4(a_op, coef[gq, ch, 4 * kk + jq], lg[ch])
        dv = np.zeros((3, 64, 4))
        Tout = np.zeros((64, 4))
        Rout = np.zeros((64, 4))
        for i in range(4):
            pix = prow + 4 * jq + i                       # this lane's pixel i
            araw = alpha_raw[gq, pix]
            pin = row_scan(np.clip(1.0 - araw, 0.0, 1.0), np.multiply, 1.0)
            pex = np.ones(64)
            pex[gq > 0] = pin[lanes[gq > 0] - 1]          # row_shr:1 with 1.0 in lane 0 of the row
            Tb = T_state[pix] * pex
            live = Tb > T_STOP
            alpha = np.where(live, araw, 0.0)
            w = alpha * Tb
            c = 1.0 / (1.0 + np.exp2(lg[:, lanes, i]))    # [3][64]: register i of the lane
            # (1): the logit in register i of lane (gq, jq) is that of (Gaussian gq, pixel 4 jq + i)
            assert np.allclose(lg[:, lanes, i], np.einsum("lck,lk->cl", coef[gq], sh[pix]), atol=1e-12)
            gc = (G[:, pix] * c).sum(0)
            rho = rho_state[pix] - row_scan(w * gc, np.add, 0.0)
            d_alpha = np.where(live, Tb * gc - rho / (1.0 - araw + 1e-7), 0.0)
            s_model[gq, pix] = d_alpha * alpha
            dv[:, :, i] = G[:, pix] * w * c * (1.0 - c)
            Tout[:, i] = T_state[pix] * pin
            Rout[:, i] = rho
        # coefficient sums: A' = sh'_(l & 15)(pixel prow + 4 jq + i), B' = register i of the lane
        for i in range(4):
            tb = sh[prow + 4 * jq + i, lanes & 15]
            for ch in range(3):
                acc[ch] = mfma_16x16x4(tb, dv[ch, :, i], acc[ch])
        for l in lanes[gq == 15]:                          # the lanes of Gaussian 15 carry the row's states on
            for i in range(4):
                T_state[prow + 4 * (l >> 4) + i] = Tout[l, i]
                rho_state[prow + 4 * (l >> 4) + i] = Rout[l, i]

    # (2): lane (gq, jq) register r holds dL/dcoef[gq][ch][4 jq + r]
    dcoef_model = np.zeros((16, 3, NB))
    for l in range(64):
        for r in range(4):
            dcoef_model[l & 15, :, 4 * (l >> 4) + r] = acc[:, l, r]
    assert np.allclose(dcoef_model, dcoef_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(s_model, s_ref, rtol=1e-10, atol=1e-12)
    assert np.allclose(rho_state, rho_ref, rtol=1e-10, atol=1e-12)
    # (3): the carried transmittance is the UNMASKED product: equal to the recursion's while the pixel lives, and at most
    # the stop threshold -- like the recursion's frozen value -- once it has stopped
    alive = T_ref > T_STOP
    assert np.allclose(T_state[alive], T_ref[alive], rtol=1e-10)
    assert np.all(T_state[~alive] <= T_STOP) and np.all(T_ref[~alive] <= T_STOP)
    # a pixel row whose 16 pixels had all stopped contributes exact zeros (the kernel leaves such rows out)
    dead_rows = [s for s in range(16) if np.all(T_in[16 * s:16 * s + 16] <= T_STOP)]
    assert dead_rows == [2]
    for s in dead_rows:
        assert np.all(s_model[:, 16 * s:16 * s + 16] == 0.0)


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: how a group's 16 gradient rows leave the wave.  Lane (g, jq) = g + 16 jq holds the coefficient sums
# acc[ch][e] = dL/dcoef[g][ch][4 jq + e] (fact (2) above) and -- after the quad sums -- every lane of Gaussian g holds its
# geometry sums.  The rows are stored as whole 64-byte lines: for line `it`, lane l offers one 16-byte piece, destination lane
# 4 r + q fetches the piece of lane r + 16 q (ds_bpermute) and stores it as float4 number 4 it + q of row r.  The row layout
# this produces must be the one the readers index through (csrc/gs_frame_layout.h: gs_row_geo, gs_row_col, gs_row_compact --
# restated here and checked against the header's text so that the two cannot drift apart).
def _row_geo(cd, m):
    return (12 + m if m < 4 else 28 + (m - 4)) if cd == 27 else m


def _row_col(cd, c):
    return 16 * (c // 9) + c % 9 if cd == 27 else (16 + c if cd == 48 else 7 + c)


def _row_floats(cd):
    return {3: 16, 27: 48, 48: 64}[cd]


def _store_rows_model(cd, header, acc):
    """header [16][7] (dx, dy, da, db, dc, dd, dopa) per Gaussian, acc [64][3][4] per lane -> rows [16][RW] as the kernel's
    closing section writes them (NaN where nothing was written)."""
    NB = cd // 3
    RW = _row_floats(cd)
    rows = np.full((16, RW), np.nan)
    h0 = lambda g: [header[g, 0], header[g, 1], header[g, 2], header[g, 3]]  # noqa: E731
    h1 = lambda g: [header[g, 4], header[g, 5], header[g, 6], 0.0]  # noqa: E731
    lines = range(4) if NB == 16 else range(3)
    for it in lines:
        offered = np.zeros((64, 4))
        for l in range(64):
            g, jq = l & 15, l >> 4
            if NB == 16:
                if it == 0:
                    offered[l] = h0(g) if jq == 0 else h1(g) if jq == 1 else [0, 0, 0, 0]
                else:
                    offered[l] = acc[l, it - 1]
            else:
                ch = it
                offered[l] = (h0(g) if ch == 0 else h1(g) if ch == 1 else [0, 0, 0, 0]) if jq == 3 else acc[l, ch]
        for l in range(64):  # destination lanes
            dr, dq = l >> 2, l & 3
            rows[dr, 4 * (4 * it + dq):4 * (4 * it + dq) + 4] = offered[dr + 16 * dq]
    return rows


def test_row_store_permutation_writes_the_documented_layout():
    import os
    import re

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "3d-gaussian-splatting_amd", "csrc",
                            "gs_frame_layout.h")).read()
    # the header's definitions are the ones restated above
    assert "return color_dim == 27 ? (m < 4 ? 12 + m : 28 + (m - 4)) : m;" in hdr
    assert "return color_dim == 27 ? 16 * (c / 9) + c % 9 : color_dim == 48 ? 16 + c : 7 + c;" in hdr
    assert re.search(r"color_dim == 3 \? 16 : color_dim == 27 \? 48 : color_dim == 48 \? 64", hdr)
    rng = np.random.default_rng(11)
    for cd in (27, 48):
        NB, RW = cd // 3, _row_floats(cd)
        header = rng.normal(size=(16, 7))
        dcoef = rng.normal(size=(16, 3, NB))
        acc = np.zeros((64, 3, 4))
        for l in range(64):
            g, jq = l & 15, l >> 4
            for ch in range(3):
                for e in range(4):
                    k = 4 * jq + e
                    acc[l, ch, e] = dcoef[g, ch, k] if k < NB else 0.0  # (k >= NB: the SH table's zero entries)
        rows = _store_rows_model(cd, header, acc)
        assert not np.isnan(rows).any()  # every float of every line is written: whole lines, nothing left to a memset
        used = np.zeros(RW, bool)
        for m in range(7):
            assert np.array_equal(rows[:, _row_geo(cd, m)], header[:, m]), (cd, "geometry", m)
            used[_row_geo(cd, m)] = True
        for c in range(cd):
            assert np.array_equal(rows[:, _row_col(cd, c)], dcoef[:, c // NB, c % NB]), (cd, "coefficient", c)
            assert not used[_row_col(cd, c)]
            used[_row_col(cd, c)] = True
        assert np.all(rows[:, ~used] == 0.0)  # padding floats are exact zeros (the pre-pass adds whole rows)
        assert used.sum() == 7 + cd
    # rgb rows (one line) keep the plain order
    assert [_row_geo(3, m) for m in range(7)] == list(range(7)) and [_row_col(3, c) for c in range(3)] == [7, 8, 9]


# ---------------------------------------------------------------------------------------------------------------------
# Round 5: the rgb backward in the row layout (raster_bwd.hip: raster_backward_rows_kernel) -- the same lanes = (Gaussian,
# pixel quad) and DPP row scans as above, without the matrix products, plus three identities of its own:
#   (a) the opacity rides in the exponent: alpha = 2^-(q - log2 sigma) and sum s q = sum s q' + log2 sigma sum s;
#   (b) T in front of Gaussian g = T_in x (inclusive product scan of lane g - 1), lane 0 keeps T_in (one v_mul_f32_dpp
#       row_shr:1 whose disabled lanes keep the destination);
#   (c) s = dL/dalpha alpha = w gc - rho beta with beta = alpha / (1 - alpha + 1e-7) -- the prefix sum of w gc is taken out
#       of place (first scan step with bound_ctrl:0: lanes without a source add 0), so the unscanned w gc is still there.
def _row_shr(v, sh, fill):
    """DPP row_shr:sh over the 16 lanes of every row: lane l reads lane l - sh of its row; `fill` where there is none."""
    out = np.array(fill, dtype=np.float64, copy=True) if np.ndim(fill) else np.full_like(v, fill)
    for l in range(64):
        if (l & 15) >= sh:
            out[l] = v[l - sh]
    return out


def test_rgb_row_layout_model_equals_the_sequential_recursion():
    rng = np.random.default_rng(9)
    n_g = 16
    gx, gy = rng.uniform(-0.2, 0.2, n_g), rng.uniform(-0.2, 0.2, n_g)
    A, C = rng.uniform(20, 400, n_g), rng.uniform(20, 400, n_g)
    B = rng.uniform(-0.9, 0.9, n_g) * 2 * np.sqrt(A * C)
    sig = rng.uniform(0.02, 0.99, n_g)
    sig[6] = 0.999
    col = rng.uniform(0, 1, (n_g, 3))
    px, py = (np.arange(16) - 7.5) / 40.0, (np.arange(16) - 7.5) / 40.0
    Gimg = rng.normal(size=(3, 16, 16))        # dL/dC [ch][y][x]
    T_in = rng.uniform(0.0, 1.0, (16, 16))
    T_in[3, :] = 5e-5                           # a pixel row that had stopped before the group: skipped
    T_in[::5, ::3] = 2e-5
    rho_in = rng.normal(size=(16, 16))

    # ---- reference: pixel by pixel, Gaussian by Gaussian (raster_backward_pixel_kernel's arithmetic, float64)
    ref = np.zeros((n_g, 10))  # Sx Sy Sxx Sxy Syy Su' (= sum s q) Sopa Sc0 Sc1 Sc2
    T_ref, rho_ref = T_in.copy(), rho_in.copy()
    for y in range(16):
        for x in range(16):
            T, rho = T_ref[y, x], rho_ref[y, x]
            for g in range(n_g):
                dx, dy = px[x] - gx[g], py[y] - gy[g]
                q = A[g] * dx * dx - B[g] * dx * dy + C[g] * dy * dy
                Gv = 2.0 ** -q
                live = T > T_STOP
                alpha = Gv * sig[g] if live else 0.0
                w = alpha * T
                gc = Gimg[:, y, x] @ col[g]
                rho = rho - w * gc
                d_alpha = (T * gc - rho / (1.00000011920928955 - Gv * sig[g])) if live else 0.0  # (the kernels' fp32 constant)
                s = d_alpha * alpha
                ref[g] += [s * dx, s * dy, s * dx * dx, s * dx * dy, s * dy * dy, s * q, d_alpha * Gv,
                           Gimg[0, y, x] * w, Gimg[1, y, x] * w, Gimg[2, y, x] * w]
                T = T - w
            T_ref[y, x], rho_ref[y, x] = T, rho

    # ---- the kernel's lanes: lane l = (g = l & 15, jq = l >> 4), four pixels x = 4 jq + i of row s per step
    lanes = np.arange(64)
    g_of, jq_of = lanes & 15, lanes >> 4
    lopa = np.log2(sig)
    S1, Sy = np.zeros((64, 4)), np.zeros((64, 4))
    Syy, Sq, Sc = np.zeros(64), np.zeros(64), np.zeros((64, 3))
    sT, sR = T_in.copy(), rho_in.copy()
    skipped = 0
    for s in range(16):
        Tin = np.stack([sT[s, 4 * jq_of + i] for i in range(4)], 1)   # [lane][i]
        Rin = np.stack([sR[s, 4 * jq_of + i] for i in range(4)], 1)
        if not (Tin > T_STOP).any():
            skipped += 1
            continue
        dy = py[s] - gy[g_of]
        for i in range(4):
            x = 4 * jq_of + i
            dx = px[x] - gx[g_of]
            qp = (C[g_of] * dy - B[g_of] * dx) * dy + (A[g_of] * dx * dx - lopa[g_of])  # q' = q - log2 sigma
            araw = 2.0 ** -qp
            pin = np.clip(1.0 - araw, 0.0, 1.0)
            scan = pin.copy()
            for sh in (1, 2, 4, 8):                      # in-place inclusive product, disabled lanes keep their value
                scan = scan * _row_shr(scan, sh, 1.0)
            Tb = Tin[:, i] * _row_shr(scan, 1, 1.0)      # (b): lane 0 keeps T_in
            alpha = np.where(Tb > T_STOP, araw, 0.0)
            w = alpha * Tb
            gc = sum(Gimg[ch, s, x] * col[g_of, ch] for ch in range(3))
            wg = w * gc
            ws = wg + _row_shr(wg, 1, 0.0)               # (c): first step with bound_ctrl:0
            for sh in (2, 4, 8):
                ws = ws + _row_shr(ws, sh, 0.0)
            rho = Rin[:, i] - ws
            beta = alpha / (1.00000011920928955 - araw)
            sv = wg - rho * beta
            for ch in range(3):
                Sc[:, ch] += Gimg[ch, s, x] * w
            S1[:, i] += sv
            Sy[:, i] += sv * dy
            Syy += sv * dy * dy
            Sq += sv * qp
            last = g_of == 15                             # lanes of Gaussian 15 hand the row's states on
            sT[s, x[last]] = (Tin[:, i] * scan)[last]
            sR[s, x[last]] = rho[last]
    assert skipped == 1
    # ---- the group's closing: the lane's four columns, then the Gaussian's four quads
    got = np.zeros((n_g, 10))
    for l in range(64):
        g, jq = g_of[l], jq_of[l]
        st = S1[l].sum()
        for i in range(4):
            dx = px[4 * jq + i] - gx[g]
            got[g, 0] += S1[l, i] * dx
            got[g, 2] += S1[l, i] * dx * dx
            got[g, 3] += Sy[l, i] * dx
            got[g, 1] += Sy[l, i]
        got[g, 4] += Syy[l]
        got[g, 5] += Sq[l] + lopa[g] * st                # (a): sum s q = sum s q' + log2 sigma sum s
        got[g, 6] += st / sig[g]                         # sum dL/dalpha G = (sum s) / sigma
        got[g, 7:10] += Sc[l]
    scale = np.abs(ref).max(axis=0) + 1e-30
    assert np.abs(got - ref).max(axis=0).max() < 1e-9 * scale.max(), np.abs(got - ref).max(axis=0) / scale
    # pixel states behind the group (dead pixels: the unmasked product keeps falling, harmlessly; rho must agree)
    live_end = T_ref > T_STOP
    assert np.allclose(sT[live_end], T_ref[live_end], rtol=1e-9, atol=0)
    assert np.allclose(np.delete(sR, 3, axis=0), np.delete(rho_ref, 3, axis=0), rtol=1e-9, atol=1e-12)


def test_fused_adam_hand_over_covers_every_element_once():
    """The optimizer step fused into the projection backward (cull_project.hip: frame_project_backward_kernel<3, 0, 256, ADAM>,
    round 5): a thread owns a GAUSSIAN, but the [N, 3] arrays are walked by ELEMENT -- lane l of wave w writes its Gaussian's
    three gradients to tr[3 l + k], then lanes with 4 l + 4 <= ne take elements 4 l .. 4 l + 3 of the wave's contiguous run
    (base 3 (pid0 + 64 w)) as one float4, and the lane that straddles the array's end takes the rest one by one.  The model
    walks every workgroup / wave / lane for array lengths around every boundary and checks that each element of the array
    is updated exactly once, with the gradient of ITS Gaussian and component, by a lane whose own Gaussian lies inside the
    array (the kernel's `valid` lanes are the only ones guaranteed to have live registers for the quaternion / opacity step,
    and the early return of invalid threads must not take a needed lane away)."""
    for n in (1, 2, 3, 5, 63, 64, 65, 127, 191, 255, 256, 257, 300, 511, 515, 1000, 1021):
        grad = np.arange(3 * n, dtype=np.int64) + 7  # gradient of element e = e + 7
        hits = np.zeros(3 * n, dtype=np.int64)
        got = np.zeros(3 * n, dtype=np.int64)
        for blk in range((n + 255) // 256):
            pid0 = blk * 256
            for wv in range(4):
                wbase = pid0 + 64 * wv
                left = n - wbase
                ne = 192 if left >= 64 else (3 * left if left > 0 else 0)
                tr = np.zeros(192, dtype=np.int64)
                for lane in range(64):  # every lane of the wave, valid or not (an invalid lane offers zeros)
                    pid = wbase + lane
                    for k in range(3):
                        tr[3 * lane + k] = grad[3 * pid + k] if pid < n else 0
                for lane in range(64):
                    e0 = 4 * lane
                    if e0 + 4 <= ne:
                        todo = range(e0, e0 + 4)
                        assert (3 * wbase + e0) % 4 == 0  # the float4 is aligned when the arrays are (wbase is a multiple of 64)
                    else:
                        todo = range(e0, ne)
                    for e in todo:
                        assert wbase + lane < n, "an element is left to a lane the kernel may have retired"
                        hits[3 * wbase + e] += 1
                        got[3 * wbase + e] = tr[e]
        assert (hits == 1).all(), n
        assert (got == grad).all(), n
