"""What rocBLAS / hipBLASLt (through torch.matmul, bf16) reach on the step's GEMM shapes -- reference point only, not used
by the product."""
import torch
dev = torch.device("cuda:0")
T = 262144
shapes = [("qkv fwd", 2304, 768), ("oproj fwd", 768, 768), ("ffn1 fwd", 3072, 768), ("ffn2 fwd", 768, 3072), ("qkv dgrad", 768, 2304)]
for name, N, K in shapes:
    a = torch.randn(T, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("%-10s N=%4d K=%4d  %7.3f ms  %7.1f TF/s (plain GEMM, no epilogue)" % (name, N, K, ms, 2.0 * T * N * K / ms / 1e9))
# wgrad shape
x = torch.randn(T, 768, device=dev, dtype=torch.bfloat16); dy = torch.randn(T, 3072, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    g = dy.t() @ x
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    g = dy.t() @ x
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("w1 wgrad [3072,768] over T  %7.3f ms  %7.1f TF/s" % (ms, 2.0 * T * 3072 * 768 / ms / 1e9))
