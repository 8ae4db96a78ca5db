"""The LDS-tiled GEMM (shape 4) at the sizes TQC's batched critic launches have (5 critics x [2048 x 512] = 1280 tiles of
64x64, emulated as one problem of 10240 rows): forward (both operands k-contiguous) and dX (B operand n-contiguous)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
import gcrl_amd

lib = gcrl_amd._ffi.lib
st = gcrl_amd._ffi.stream_handle()


def timed(args, flops, tag, reps=30):
    for _ in range(3):
        assert lib.gcrl_gemm_f32(*args) == 0
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        lib.gcrl_gemm_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{tag}: {us:8.1f} us  {flops / us / 1e6:6.1f} TFLOP/s", flush=True)


M, N, K = int(sys.argv[1]) if len(sys.argv) > 1 else 10240, 512, 512
X = torch.randn(M, K, device="cuda")
W = torch.randn(N, K, device="cuda")
Y = torch.zeros(M, N, device="cuda")
b = torch.randn(N, device="cuda")
# forward: Y = leaky(X W^T + b)
timed((X.data_ptr(), K, 1, W.data_ptr(), 1, K, Y.data_ptr(), N, b.data_ptr(), M, N, K, 1, 4, st), 2.0 * M * N * K, f"fwd  M={M} N={N} K={K}")
# dX = G W  (G [M][N], W [N][K]: k = N index; B(k, n) = W[k*K + n])
G = torch.randn(M, N, device="cuda")
dX = torch.zeros(M, K, device="cuda")
timed((G.data_ptr(), N, 1, W.data_ptr(), K, 1, dX.data_ptr(), K, None, M, K, N, 0, 4, st), 2.0 * M * N * K, f"dX   M={M} N={K} K={N}")
ref = torch.nn.functional.leaky_relu(X @ W.T + b)
print("fwd max err", float((Y - ref).abs().max()), "dX max err", float((dX - G @ W).abs().max()))
# yardstick: the library's fp32 GEMM at the same sizes (plain product, no epilogue)
for tag, fn in (("torch X@W.T", lambda: torch.matmul(X, W.T)), ("torch G@W  ", lambda: torch.matmul(G, W))):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 30
    print(f"{tag}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:6.1f} TFLOP/s")
print("sample", Y[3, :3].tolist(), ref[3, :3].tolist())
