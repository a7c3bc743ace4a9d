"""Calibration only (not part of the product path): what the vendor libraries reach on the
GEMM / conv shapes of yolov3@608 b=16 -- hipBLASLt through torch.matmul on the implicit-GEMM
shape, MIOpen through conv2d (channels_last bf16).  Gives a practical ceiling to compare the
hand-written kernels with (random data, so DVFS behaves like the real run)."""
import sys
import time
import torch

SHAPES = [  # name, H, Cin, Cout, k, stride
    ("s76_128-256_k3", 76, 128, 256, 3, 1),
    ("s38_256-512_k3", 38, 256, 512, 3, 1),
    ("s19_512-1024_k3", 19, 512, 1024, 3, 1),
    ("s152_64-128_k3", 152, 64, 128, 3, 1),
    ("s304_32-64_k3", 304, 32, 64, 3, 1),
    ("s76_256-128_k1", 76, 256, 128, 1, 1),
    ("s38_512-256_k1", 38, 512, 256, 1, 1),
    ("s19_1024-512_k1", 19, 1024, 512, 1, 1),
]


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = 16
    dev = torch.device("cuda")
    torch.manual_seed(0)
    for name, H, cin, cout, k, s in SHAPES:
        M, N, K = B * H * H, cout, k * k * cin
        flops = 2.0 * M * N * K
        a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
        t_mm = timeit(lambda: torch.matmul(a, w.t()))
        x = torch.randn(B, cin, H, H, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        wc = torch.randn(cout, cin, k, k, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
        try:
            t_cv = timeit(lambda: torch.nn.functional.conv2d(x, wc, None, s, (k - 1) // 2))
        except Exception as e:  # noqa
            t_cv = float("nan")
        print("%-18s M=%6d N=%4d K=%4d  matmul %.4f ms %7.1f TF   conv2d %.4f ms %7.1f TF" %
              (name, M, N, K, t_mm, flops / t_mm / 1e9, t_cv, flops / t_cv / 1e9))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
