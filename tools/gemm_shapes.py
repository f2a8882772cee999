"""usage (GPU box): python tools/gemm_shapes.py - how fast does rocBLAS run the GEMMs of the wide LQ sweep
(csrc/ogsqp_lqwide.h: rows x L matrix A row-major, nb reflectors V row-major) in their possible formulations?
Times every variant with HIP events; prints TF/s and the share of the HBM roofline of each."""
import ctypes as C
import sys
import torch

lib = C.CDLL("librocblas.so.5", mode=C.RTLD_GLOBAL)
h = C.c_void_p()
assert lib.rocblas_create_handle(C.byref(h)) == 0
lib.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
N_, T_ = 111, 112
dp = C.c_void_p


def gemm(ta, tb, m, n, k, alpha, A, lda, B, ldb, beta, Cm, ldc):
    a, b = C.c_double(alpha), C.c_double(beta)
    rc = lib.rocblas_dgemm(h, ta, tb, m, n, k, C.byref(a), dp(A.data_ptr()), lda, dp(B.data_ptr()), ldb, C.byref(b),
                           dp(Cm.data_ptr()), ldc)
    assert rc == 0, rc


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for rows, L in ((10283, 6149), (8000, 4000), (6200, 2100)):
    ld = 6160
    A = torch.randn(rows, ld, dtype=torch.float64, device="cuda")
    for nb in (64, 128):
        V = torch.randn(nb, ld, dtype=torch.float64, device="cuda")
        W = torch.zeros(rows * nb, dtype=torch.float64, device="cuda")
        flops = 2.0 * rows * L * nb
        variants = {
            # W (rows x nb) = A V'
            "W: T,N m=nb n=rows (now)": lambda: gemm(T_, N_, nb, rows, L, 1.0, V, ld, A, ld, 0.0, W, nb),
            "W: T,N m=rows n=nb": lambda: gemm(T_, N_, rows, nb, L, 1.0, A, ld, V, ld, 0.0, W, rows),
            # A -= W2 V
            "U: N,N m=L n=rows k=nb (now)": lambda: gemm(N_, N_, L, rows, nb, -1.0, V, ld, W, nb, 1.0, A, ld),
            "U: N,T m=L n=rows k=nb": lambda: gemm(N_, T_, L, rows, nb, -1.0, V, ld, W, rows, 1.0, A, ld),
        }
        for name, fn in variants.items():
            t = timed(fn)
            traffic = 8.0 * rows * L * (1 if name.startswith("W") else 2)
            print("rows %5d L %4d nb %3d  %-30s %7.1f us  %5.1f TF/s  %4.2f TB/s" % (rows, L, nb, name, t * 1e6, flops / t / 1e12,
                                                                                traffic / t / 1e12), flush=True)
    del A
