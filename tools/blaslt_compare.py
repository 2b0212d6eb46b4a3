"""What the vendor library reaches on the encoder's GEMM shapes (torch.nn.functional.linear -> hipBLASLt / rocBLAS), as a yardstick
for the hand-written kernels.  Plain GEMM + bias only: none of the fused epilogues (LayerNorm fold, GELU, residual planes, row
statistics) the product kernels carry.  Not part of the product path.   python tools/blaslt_compare.py [batch]"""
import sys
import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
M = B * 192
dev = 'cuda'
shapes = {'qkv': (768, 2304), 'proj': (768, 768), 'fc1': (768, 3072), 'fc2': (3072, 768),
          'L.qkv': (1024, 3072), 'L.fc1': (1024, 4096), 'L.fc2': (4096, 1024)}
for dt in (torch.float16, torch.bfloat16):
    for name, (K, N) in shapes.items():
        x = torch.randn(M, K, device=dev, dtype=dt) * 0.5
        w = torch.randn(N, K, device=dev, dtype=dt) * 0.03
        b = torch.randn(N, device=dev, dtype=dt)
        for _ in range(5):
            y = F.linear(x, w, b)
        torch.cuda.synchronize()
        ts = []
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = F.linear(x, w, b)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        us = min(ts)
        print(f'{str(dt)[6:]:9s} {name:6s} M={M} K={K} N={N}: {us:7.1f} us  {2 * M * K * N / us / 1e9:6.3f} PFLOP/s', flush=True)
        if name == 'fc1':
            for _ in range(3):
                y = F.gelu(F.linear(x, w, b))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = F.gelu(F.linear(x, w, b))
            e1.record()
            torch.cuda.synchronize()
            print(f'{str(dt)[6:]:9s} fc1+gelu (separate elementwise pass): {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us', flush=True)
