"""Developer tool (GPU box): the library's fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt) on the GEMM shapes of the step's 1x1
convolutions, as a yard-stick for igemm.hip / wgrad.hip (plain GEMM: no fused bias / ReLU / BatchNorm partials, no split-K slabs).
usage: python tools/gemm_yardstick.py"""
import torch

SHAPES = [  # (name, M = pixels, N = output channels, K = input channels)
    ("l1 conv1/conv3 64<->256", 90000, 64, 256), ("l1 conv3", 90000, 256, 64),
    ("l2 conv1", 23104, 128, 512), ("l2 conv3", 23104, 512, 128),
    ("l3 conv1", 5776, 256, 1024), ("l3 conv3", 5776, 1024, 256),
    ("l4 conv1", 1600, 512, 2048), ("l4 conv3", 1600, 2048, 512),
    ("l3.0 downsample", 5776, 1024, 512), ("l4.0 downsample", 1600, 2048, 1024),
    ("square 4096", 4096, 4096, 4096),
]


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]


torch.backends.cuda.matmul.allow_tf32 = False
print(f"{'shape':28s} {'fwd  y = x w^T':>22s} {'wgrad dw = dy^T x':>24s}")
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device="cuda")
    w = torch.randn(N, K, device="cuda")
    dy = torch.randn(M, N, device="cuda")
    gf = 2.0 * M * N * K / 1e9
    t_f = timeit(lambda: torch.mm(x, w.t()))
    t_w = timeit(lambda: torch.mm(dy.t(), x))
    print(f"{name:28s} {1e3 * t_f:8.1f} us {gf / t_f:7.1f} TF/s   {1e3 * t_w:8.1f} us {gf / t_w:7.1f} TF/s   ({M}x{N}x{K})", flush=True)
