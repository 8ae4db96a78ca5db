"""dW-shaped problems (batch-major operands, long K) through the GEMM shapes: us and TFLOP/s per shape."""
import sys, torch
sys.path.insert(0, "/root/repo")
import gcrl_amd
lib = gcrl_amd._ffi.lib
def run(M, N, K, shape, reps=50):
    G = torch.randn(K, M, device="cuda"); X = torch.randn(K, N, device="cuda"); out = torch.zeros(M, N, device="cuda")
    st = gcrl_amd._ffi.stream_handle()
    args = (G.data_ptr(), 1, M, X.data_ptr(), N, 1, out.data_ptr(), N, None, M, N, K, 0, shape, st)
    for _ in range(5): lib.gcrl_gemm_f32(*args)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(reps): lib.gcrl_gemm_f32(*args)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"M={M} N={N} K={K} shape {shape}: {us:8.1f} us  {2*M*N*K/us/1e6:6.1f} TFLOP/s")
for (M, N, K) in [(512, 513, 2048), (256, 257, 2048), (256, 257, 1024), (256, 257, 256)]:
    for s in (1, 2, 4): run(M, N, K, s)
