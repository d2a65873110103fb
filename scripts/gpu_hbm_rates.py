"""What this board's HBM sustains, for scale next to the stencils: a read-only stream
(sum of a tensor), a write-only stream (fill) and a copy, at sizes inside and well beyond the
256 MB Infinity Cache."""
import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    for mb in (128, 240, 512, 1024, 4096):
        n = mb * (1 << 20) // 4
        x = torch.ones(n, dtype=torch.float32, device="cuda")
        y = torch.empty_like(x)
        r = timed(lambda: x.sum())
        w = timed(lambda: y.fill_(1.0))
        c = timed(lambda: y.copy_(x))
        print("%5d MB: read %.0f GB/s, write %.0f GB/s, copy %.0f GB/s (read + write)" % (
            mb, n * 4 / r / 1e9, n * 4 / w / 1e9, 2 * n * 4 / c / 1e9), flush=True)


if __name__ == "__main__":
    main()
