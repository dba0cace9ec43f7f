"""Time dg_bn_stats_from_partials (merge of the conv kernels' BatchNorm partial rows) in its one-launch and two-level forms.
python tools/bench_partials.py            (GPU)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from discogan_modernized_amd import ops, _lib


def main():
    dev = "cuda"
    L = _lib.load()
    print("# rows x channels: one launch us | two-level us | max |mean diff| / max |invstd rel diff|")
    for P, C in [(16384, 64), (4096, 128), (2048, 128), (1024, 256), (1024, 512), (512, 512), (256, 1024), (64, 2048)]:
        rows_per = 32
        M = P * rows_per
        g = torch.Generator(device=dev).manual_seed(P + C)
        stat = torch.empty((P, 3 * C + 4), device=dev)
        stat[:, 0] = rows_per
        stat[:, 4:4 + C] = torch.randn((P, C), device=dev, generator=g)
        stat[:, 4 + C:4 + 2 * C] = torch.randn((P, C), device=dev, generator=g) * 3
        stat[:, 4 + 2 * C:] = torch.rand((P, C), device=dev, generator=g) * 40 + 5
        y = torch.empty((1, C, M, 1), device=dev).permute(0, 1, 2, 3)[:, :, :1, :].expand(1, C, M, 1)   # only the shape is read
        res = []
        for one in (True, False):
            ops._PARTIALS_ONE_LAUNCH = one
            rm, rv, nbt = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(1, device=dev, dtype=torch.int64)
            for _ in range(5):
                saved = ops.bn_stats_from_partials(stat, y, rm, rv, nbt, 1e-5, 0.1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                saved = ops.bn_stats_from_partials(stat, y, rm, rv, nbt, 1e-5, 0.1)
            e1.record()
            torch.cuda.synchronize()
            res.append((e0.elapsed_time(e1) / 50 * 1e3, saved.clone()))
        dm = (res[0][1][0] - res[1][1][0]).abs().max().item()
        dv = ((res[0][1][1] - res[1][1][1]).abs() / res[0][1][1].abs()).max().item()
        print(f"{P:6d} x {C:5d}: {res[0][0]:8.1f} | {res[1][0]:8.1f} | {dm:.2e} / {dv:.2e}")


if __name__ == "__main__":
    main()
