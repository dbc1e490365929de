"""Calibration only (nothing here is product code): what the vendor's own bf16 GEMM (torch.matmul -> hipBLASLt) sustains
on THIS box at shapes whose bf16 matrix work equals one bf16x6 nn.Linear call of the training step (six bf16 partial
products per fp32 product: K' = 6 K), next to the box's register-only MFMA rate.  Random operands (the chip's clock under
matrix load depends on the data).  python tools/bf16_gemm_reference.py"""
import sys, time, torch
sys.path.insert(0, ".")
from gct_plus_amd import graphdiag

dev = "cuda"
shapes = [("ffn2-like", 32768, 512, 6 * 2048), ("ffn1-like", 32768, 2048, 6 * 512), ("qkv-like", 32768, 1536, 6 * 512),
          ("out-like", 32768, 512, 6 * 512), ("square 8k", 8192, 8192, 8192)]
print("box:", graphdiag.mfma_probe())
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        c = a @ b.t()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    t = sorted(ts)[len(ts) // 2] * 1e-3
    print(f"{name:10s} M={M} N={N} K'={K}: {t * 1e6:8.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} bf16 TFLOP/s (bf16 in, bf16 out)")
