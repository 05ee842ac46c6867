"""Developer tool (CPU): END-TO-END numeric gate for Winograd F(4x4,3x3) on the head / pyramid-output 3x3 convolutions
(op-level error vs fp64 <= 1e-5 max-normalised on the head shapes AND the e2e forward bound of
tests/test_gpu_fullshape.py — abs 2e-3 against the reference golden g10_e2e_300_b16 — still green).

The fp32 CPU oracle (= the reference's arithmetic) runs the benchmark shape (ResNet-50 FPN, 300x300, B=16, train mode) three times:
  direct : every convolution by torch (the oracle as it is)                      -> distance to the golden = 0 by construction
  f2     : the stride-1 3x3 convolutions of head + pyramid outputs by an fp32 emulation of F(2x2,3x3)  (what csrc/wino.hip runs)
  f4     : the same layers by F(4x4,3x3) on the levels with >= 16 rows (38^2, 19^2), F(2x2,3x3) on the small ones
and prints max |out - golden| over the golden's sampled anchors.  The emulation rounds every transform and the channel sum to fp32
(the channel sum through torch's fp32 GEMM: blocked order, same error class as the MFMA K loop).
usage: python tools/wino_f4_e2e.py [points] [enc]     points: std (0,+-1,+-2; default) | mix (0,1,-1,1/2,-2); enc: also layer1's conv2"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zsg_oracle as O            # noqa: E402
from tools.wino_f4_error import cook_toom     # noqa: E402

_real_conv2d = F.conv2d
MODE = {"m": 0, "min_hw": 16, "only": None}
MATS = {}


def mats(m, pts):
    key = (m, tuple(map(str, pts)))
    if key not in MATS:
        AT, G, BT = cook_toom(pts, m, 3)
        MATS[key] = tuple(torch.from_numpy(x.astype(np.float32)) for x in (AT, G, BT))
    return MATS[key]


def wino_conv2d(x, w, bias, m, pts):
    AT, G, BT = mats(m, pts)
    B, C, H, W = x.shape
    N = w.shape[0]
    a = m + 2
    ty, tx = (H + m - 1) // m, (W + m - 1) // m
    xp = F.pad(x, (1, tx * m + 1 - W, 1, ty * m + 1 - H))
    d = xp.unfold(2, a, m).unfold(3, a, m)                          # [B, C, ty, tx, a, a]
    V = torch.einsum("ij,bcyxjk->bcyxik", BT, d)
    V = torch.einsum("bcyxik,lk->bcyxil", V, BT)                    # B^T d B
    U = torch.einsum("ij,ncjk->ncik", G, w)
    U = torch.einsum("ncik,lk->ncil", U, G)                         # G g G^T
    Vm = V.permute(4, 5, 1, 0, 2, 3).reshape(a * a, C, B * ty * tx)
    Um = U.permute(2, 3, 0, 1).reshape(a * a, N, C)
    M = torch.bmm(Um, Vm).reshape(a, a, N, B, ty, tx)               # fp32 channel sums
    Y = torch.einsum("ij,jknbyx->iknbyx", AT, M)
    Y = torch.einsum("iknbyx,lk->ilnbyx", Y, AT)                    # [m, m, N, B, ty, tx]
    y = Y.permute(3, 2, 4, 0, 5, 1).reshape(B, N, ty * m, tx * m)[:, :, :H, :W]
    if bias is not None:
        y = y + bias.view(1, -1, 1, 1)
    return y.contiguous()


PTS = {"std": [0, 1, -1, 2, -2], "mix": [0, 1, -1, "1/2", -2]}
WHICH = "std"


def patched(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
    s = stride if isinstance(stride, int) else stride[0]
    p = padding if isinstance(padding, int) else padding[0]
    if MODE["m"] and w.shape[2] == 3 and s == 1 and p == 1 and MODE["hit"](w):
        m = MODE["m"] if min(x.shape[2], x.shape[3]) >= MODE["min_hw"] else 2
        return wino_conv2d(x, w, bias, m, PTS[WHICH] if m == 4 else [0, 1, -1])
    return _real_conv2d(x, w, bias, stride, padding, dilation, groups)


def main():
    global WHICH
    if len(sys.argv) > 1:
        WHICH = sys.argv[1]
    torch.set_num_threads(os.cpu_count())
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_e2e_300_b16.npz"))
    sd = O.seeded_state_dict("resnet50", int(g["seed"][0]))
    bt = O.synthetic_batch(16, 300, 300, seed=int(g["batch_seed"][0]))
    h0, c0 = torch.from_numpy(g["h0"]), torch.from_numpy(g["c0"])
    head_w = {id(v) for k, v in sd.items() if k.startswith("att_reg_box") and k.endswith("weight")}
    fpn_w = {id(v) for k, v in sd.items() if k.startswith("backbone.fpn.P") and k.endswith("_2.weight")}
    enc_w = {id(v) for k, v in sd.items() if k.startswith("backbone.encoder.layer1.") and k.endswith("conv2.weight")} if "enc" in sys.argv else set()
    MODE["hit"] = lambda w: (id(w) in head_w) or (id(w) in fpn_w) or (id(w) in enc_w)
    O.F.conv2d = patched
    res = {}
    with torch.no_grad():
        for name, m in (("direct", 0), ("f2", 2), ("f4", 4)):
            MODE["m"] = m
            out = O.zsgnet_forward(sd, bt, h0, c0, arch="resnet50")
            att, bbx = out["att_out"].numpy(), out["bbx_out"].numpy()
            e_att = float(np.abs(att[:, ::37] - g["att_out_s"]).max())
            e_bbx = float(np.abs(bbx[:, ::37] - g["bbx_out_s"]).max())
            res[name] = (att, bbx)
            print(f"{name:7s} max |out - reference golden|: att {e_att:.3e}  bbx {e_bbx:.3e}   (max |att| {np.abs(att).max():.3f}, max |bbx| {np.abs(bbx).max():.3f})", flush=True)
    for a, b in (("f2", "direct"), ("f4", "direct"), ("f4", "f2")):
        print(f"{a} vs {b}: att {np.abs(res[a][0] - res[b][0]).max():.3e}  bbx {np.abs(res[a][1] - res[b][1]).max():.3e}")


if __name__ == "__main__":
    main()
