"""Developer tool (CPU): the numeric gate of VERDICT r03 item 4 — would Winograd F(4x4,3x3) be admissible for the head's 3x3 layers?
fp32 arithmetic emulated with numpy float32 at the kernel's granularity: input / filter transforms in fp32, the channel sum accumulated
in fp32 in the order an MFMA K loop adds (sequential over channels), output transform in fp32; reference: the direct convolution in
float64.  Compared: F(2x2,3x3) (what csrc/wino.hip runs: constants 0, +-1, +-1/2), F(4x4,3x3) with the standard interpolation points
(0, +-1, +-2) and with the better-conditioned points (0, +-1, +-1/2) of Barabasz et al. ("Error analysis and improving the accuracy of
Winograd convolution for deep neural networks").  Gate: max error of F(4x4) <= 2 x F(2x2)'s on head-like data (C = 256, unit-variance
activations after ReLU, He-scaled weights).
usage: python tools/wino_f4_error.py"""
import numpy as np


def cook_toom(points, m, r):
    """AT [m x a], G [a x r], BT [a x a] (a = m + r - 1) for the polynomial points (+ infinity), in float64 (Lavin / wincnn construction)"""
    from fractions import Fraction as Fr
    a = m + r - 1
    pts = [Fr(p) for p in points]
    assert len(pts) == a - 1

    def poly_mul(p, q):
        out = [Fr(0)] * (len(p) + len(q) - 1)
        for i, x in enumerate(p):
            for j, y in enumerate(q):
                out[i + j] += x * y
        return out
    # f[i] = prod_{j != i} (p_i - p_j)
    f = [Fr(1)] * (a - 1)
    for i in range(a - 1):
        for j in range(a - 1):
            if i != j:
                f[i] *= pts[i] - pts[j]
    AT = [[pts[j] ** i for j in range(a - 1)] + [Fr(1 if i == m - 1 else 0)] for i in range(m)]
    G = [[pts[i] ** j / f[i] for j in range(r)] for i in range(a - 1)] + [[Fr(1 if j == r - 1 else 0) for j in range(r)]]
    # BT: rows i < a-1: coefficients of prod_{j != i} (x - p_j); last row: prod_j (x - p_j)
    BT = []
    for i in range(a - 1):
        p = [Fr(1)]
        for j in range(a - 1):
            if j != i:
                p = poly_mul(p, [-pts[j], Fr(1)])
        BT.append(p + [Fr(0)] * (a - len(p)))
    p = [Fr(1)]
    for j in range(a - 1):
        p = poly_mul(p, [-pts[j], Fr(1)])
    BT.append(p)
    to = lambda M: np.array([[float(x) for x in row] for row in M], np.float64)
    return to(AT), to(G), to(BT)


def wino_conv(x, w, AT, G, BT, m):
    """x [C, H, W] (H, W multiples of m, zero padding 1), w [N, C, 3, 3]; every product and sum rounded to float32"""
    f32 = np.float32
    C, H, W = x.shape
    N = w.shape[0]
    a = m + 2
    AT32, G32, BT32 = AT.astype(f32), G.astype(f32), BT.astype(f32)
    U = np.einsum("ij,ncjk,lk->ncil", G32, w.astype(f32), G32).astype(f32)          # [N, C, a, a]   (filter image: made once per step)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1))).astype(f32)
    out = np.zeros((N, H, W), f32)
    for ty in range(H // m):
        for tx in range(W // m):
            d = xp[:, ty * m:ty * m + a, tx * m:tx * m + a]
            V = np.einsum("ij,cjk->cik", BT32, d).astype(f32)
            V = np.einsum("cik,lk->cil", V, BT32).astype(f32)                          # [C, a, a]
            M = np.zeros((N, a, a), f32)
            for c in range(C):                                                         # fp32 accumulation in K order
                M = (M + U[:, c] * V[c]).astype(f32)
            Y = np.einsum("ij,njk->nik", AT32, M).astype(f32)
            Y = np.einsum("nik,lk->nil", Y, AT32).astype(f32)
            out[:, ty * m:(ty + 1) * m, tx * m:(tx + 1) * m] = Y
    return out


def direct64(x, w):
    C, H, W = x.shape
    xp = np.pad(x.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
    out = np.zeros((w.shape[0], H, W))
    for u in range(3):
        for v in range(3):
            out += np.einsum("nc,chw->nhw", w[:, :, u, v].astype(np.float64), xp[:, u:u + H, v:v + W])
    return out


def main():
    rng = np.random.default_rng(0)
    C, N, H = 256, 32, 8
    x = np.maximum(rng.normal(0, 1, (C, H, H)), 0).astype(np.float32)                 # post-ReLU activations
    w = (rng.normal(0, 1, (N, C, 3, 3)) * np.sqrt(2.0 / (9 * C))).astype(np.float32)
    ref = direct64(x, w)
    scale = np.abs(ref).max()
    res = {}
    for name, pts, m in (("F(2x2,3x3) pts 0,1,-1", [0, 1, -1], 2), ("F(4x4,3x3) pts 0,1,-1,2,-2", [0, 1, -1, 2, -2], 4),
                         ("F(4x4,3x3) pts 0,1,-1,1/2,-1/2", [0, 1, -1, "1/2", "-1/2"], 4), ("F(4x4,3x3) pts 0,1,-1,1/2,-2", [0, 1, -1, "1/2", -2], 4)):
        AT, G, BT = cook_toom(pts, m, 3)
        y = wino_conv(x, w, AT, G, BT, m)
        e = np.abs(y - ref)
        res[name] = (e.max() / scale, np.sqrt((e ** 2).mean()) / scale)
        print(f"{name:36s} max|err| / max|y| = {res[name][0]:.2e}   rms = {res[name][1]:.2e}")
    # the direct fp32 convolution (what igemm.hip computes), same accumulation order
    acc = np.zeros_like(ref, dtype=np.float32)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for u in range(3):
        for v in range(3):
            for c in range(C):
                acc = (acc + w[:, c, u, v][:, None, None] * xp[c, u:u + H, v:v + H][None]).astype(np.float32)
    e = np.abs(acc - ref)
    print(f"{'direct fp32 (igemm order)':36s} max|err| / max|y| = {e.max() / scale:.2e}   rms = {np.sqrt((e ** 2).mean()) / scale:.2e}")
    base = res["F(2x2,3x3) pts 0,1,-1"][0]
    for k, v in res.items():
        if k.startswith("F(4x4"):
            print(f"gate: {k}: {v[0] / base:.1f} x F(2x2)'s max error (admissible: <= 2.0)")


if __name__ == "__main__":
    main()
