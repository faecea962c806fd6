"""hipBLASLt (through torch.matmul) vs the persistent GEMM on the BACKWARD shapes of the fusion transformer (dX = dY W: NN,
dW = dY^T X: TN, fp32 out) -- a ceiling estimate, not product code."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bifold_amd.hip import ops
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
def timeit(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
M = 70400
for name, n_out, n_in in [("fc2", 512, 2048), ("fc1", 2048, 512), ("out", 512, 512), ("qkv", 1536, 512)]:
    dy = torch.randn(M, n_out, device=dev, generator=g).bfloat16()
    x = torch.randn(M, n_in, device=dev, generator=g).bfloat16()
    w = (torch.randn(n_out, n_in, device=dev, generator=g) * n_in ** -0.5).bfloat16()
    dw = torch.empty(n_out, n_in, device=dev, dtype=torch.float32)
    fl = 2.0 * M * n_out * n_in
    t_nn_lt = timeit(lambda: torch.matmul(dy, w))
    t_nn = timeit(lambda: ops.gemm(dy, w, b_kn=True))
    t_tn_lt = timeit(lambda: torch.matmul(dy.t(), x))                       # bf16 out
    t_tn = timeit(lambda: ops.gemm(dy, x, trans_a=True, b_kn=True, out=dw))
    print(f"{name}: dX {M}x{n_in}x{n_out}: hipBLASLt {t_nn_lt*1e3:6.1f} us {fl/t_nn_lt/1e9:5.0f} TF/s | own {t_nn*1e3:6.1f} us {fl/t_nn/1e9:5.0f} TF/s"
          f"   ||   dW {n_out}x{n_in}x{M}: hipBLASLt {t_tn_lt*1e3:6.1f} us {fl/t_tn_lt/1e9:5.0f} TF/s | own {t_tn*1e3:6.1f} us {fl/t_tn/1e9:5.0f} TF/s", flush=True)
