"""Small-K problems of TQC's batched launches (first layers K = 25, scalar-head dX K = 1) through the GEMM forms."""
import sys
import torch
sys.path.insert(0, "/root/repo")
import gcrl_amd

lib = gcrl_amd._ffi.lib
st = gcrl_amd._ffi.stream_handle()


def timed(args, tag, reps=30):
    for _ in range(3):
        assert lib.gcrl_gemm_f32(*args) == 0
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps):
        lib.gcrl_gemm_f32(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"{tag}: {e0.elapsed_time(e1) * 1e3 / reps:8.1f} us", flush=True)


M, N = 10240, 512
for K, ldx in ((25, 28), (1, 1), (3, 8)):
    X = torch.randn(M, ldx, device="cuda")
    W = torch.randn(N, K, device="cuda")
    Y = torch.zeros(M, N, device="cuda")
    b = torch.randn(N, device="cuda")
    for shape in (2, 3, 4):
        timed((X.data_ptr(), ldx, 1, W.data_ptr(), 1, K, Y.data_ptr(), N, b.data_ptr(), M, N, K, 1, shape, st), f"fwd-like M={M} N={N} K={K} shape {shape}")
    ref = torch.nn.functional.leaky_relu(X[:, :K] @ W.T + b)
    print("   max err", float((Y - ref).abs().max()))
