"""hipBLASLt yardstick (libdvla_cmp.so, include/dvla_cmp.h) next to the hand-written kernels on the plain GEMMs of the training
step: same parameter block, same box, launches interleaved.  Measurement infrastructure (GPU box only, not a test, not product).
    python tests/library_yardstick.py > gpurun_out/library_yardstick.txt"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import _lib  # noqa: E402
from dreamvla_amd._lib import GemmParams  # noqa: E402

BF = torch.bfloat16
# (name, M, N, K, a_trans, b_trans, bias, out fp32): the step's plain / bias-only launches (profiles/r02_gemm_breakdown.json)
CASES = [("trunk fc2 dX", 20832, 1024, 4096, 0, 0, 0, 0), ("trunk c_attn dX", 20832, 1024, 3072, 0, 0, 0, 0),
         ("trunk c_attn fwd", 20832, 3072, 1024, 0, 1, 1, 0), ("vit qkv", 88256, 2304, 768, 0, 0, 1, 0),
         ("dec fc2 dX", 91840, 1024, 4096, 0, 1, 0, 0), ("dW fc1", 1024, 4096, 20832, 1, 1, 0, 0),
         ("dW fc2", 4096, 1024, 20832, 1, 1, 0, 0), ("dW dec fc1", 1024, 4096, 91840, 1, 1, 0, 0),
         ("square", 8192, 8192, 8192, 0, 0, 0, 0)]


def main():
    lib, cmp = _lib.load(), _lib.load_comparator()
    torch.manual_seed(0)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    print("GEMM launch time, microseconds [TFLOP/s]: hand-written (cost-model choice, variant 0; best forced variant) vs hipBLASLt")
    for name, M, N, K, at, bt, bias, f32 in CASES:
        a = torch.randn((K, M) if at else (M, K), device="cuda", dtype=BF)
        b = torch.randn((K, N) if bt else (N, K), device="cuda", dtype=BF)
        c = torch.empty((M, N), device="cuda", dtype=torch.float32 if f32 else BF)
        bv = torch.randn(N, device="cuda", dtype=BF) if bias else None
        p = GemmParams()
        p.A, p.lda, p.a_trans = a.data_ptr(), a.stride(0), at
        p.B, p.ldb, p.b_trans = b.data_ptr(), b.stride(0), bt
        p.C, p.ldc, p.c_dtype = c.data_ptr(), c.stride(0), 1 if f32 else 0
        p.M, p.N, p.K = M, N, K
        if bias:
            p.bias, p.bias_dtype = bv.data_ptr(), 0
        p.split_k = 1
        sk = None
        if at:      # the model's weight-gradient launches use split-K with a workspace
            from dreamvla_amd.ops import auto_split_k
            p.split_k = auto_split_k(M, N, K)
            sk = torch.empty((p.split_k, M, N), dtype=torch.float32, device="cuda")
            p.workspace = sk.data_ptr()
        q = GemmParams.from_buffer_copy(p)
        q.split_k, q.workspace = 1, None

        def ours(v):
            lib.dvla_set_gemm_variant(v)
            rc = lib.dvla_gemm_bf16(C.byref(p), None)
            lib.dvla_set_gemm_variant(0)
            return rc

        def theirs():
            return cmp.dvla_gemm_library_bf16(C.byref(q), ws.data_ptr(), ws.numel(), None)

        def time(fn, it=8):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(it):
                fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / it * 1e3

        rc_lib = theirs()
        best = {}
        for rnd in range(3):
            for v in (0, 4, 6, 7, 8, 9):
                best[v] = min(best.get(v, 1e9), time(lambda: ours(v)))
            if rc_lib == 0:
                best["lib"] = min(best.get("lib", 1e9), time(theirs))
        fl = 2.0 * M * N * K
        bv_ = min((v for v in best if v != "lib"), key=lambda v: best[v])
        tf = lambda us: fl / us / 1e6
        libs = f"{best['lib']:8.1f} [{tf(best['lib']):5.0f}]" if "lib" in best else f"unsupported (rc {rc_lib})"
        print(f"{name:18s} {M:6d} x {N:5d} x {K:6d} sk{p.split_k:2d}  v0 {best[0]:8.1f} [{tf(best[0]):5.0f}]   best v{bv_} {best[bv_]:8.1f} [{tf(best[bv_]):5.0f}]   "
              f"hipBLASLt {libs}", flush=True)


if __name__ == "__main__":
    main()
