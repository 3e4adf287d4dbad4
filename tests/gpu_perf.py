"""Micro-benchmarks of the individual kernels at the DreamVLA shapes (GPU box only); prints one line each
and writes gpurun_out/perf.json.  Not a test."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from dreamvla_amd import ops  # noqa: E402

BF = torch.bfloat16
DEV = "cuda"


def timeit(fn, iters=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    res = []
    torch.manual_seed(0)
    ops.GemmTuner.enabled = False      # time the library's own cost-model choice, not the tuner's trial calls

    def rec(name, sec, flops=None, nbytes=None):
        r = {"name": name, "us": sec * 1e6}
        if flops:
            r["TFLOPs"] = flops / sec / 1e12
        if nbytes:
            r["GBps"] = nbytes / sec / 1e9
        res.append(r)
        print(json.dumps(r), flush=True)

    # GEMMs: trunk (M = 32*651), ViT (M = 448*197)
    for (M, N, K) in [(20832, 3072, 1024), (20832, 1024, 1024), (20832, 4096, 1024), (20832, 1024, 4096),
                      (88256, 2304, 768), (88256, 3072, 768), (8192, 8192, 8192)]:
        a = torch.randn(M, K, device=DEV, dtype=BF)
        b = torch.randn(N, K, device=DEV, dtype=BF) * 0.03
        rec(f"gemm NT {M}x{N}x{K}", timeit(lambda: ops.gemm(a, b)), 2 * M * N * K)
        if M == 20832 and N == 1024 and K == 1024:
            bt = b.t().contiguous()
            rec(f"gemm NN(b_trans) {M}x{N}x{K}", timeit(lambda: ops.gemm(a, bt, b_trans=True)), 2 * M * N * K)
            bias = torch.randn(N, device=DEV, dtype=BF)
            resid = torch.randn(M, N, device=DEV, dtype=BF)
            rec(f"gemm NT+bias+gelu+res {M}x{N}x{K}", timeit(lambda: ops.gemm(a, b, bias=bias, act=1, residual=resid)), 2 * M * N * K)
            # weight-gradient shape: C[N,K] = dY^T X, contraction M
            dy = torch.randn(M, N, device=DEV, dtype=BF)
            for sk in (1, 4, 8):
                rec(f"gemm TT dW {N}x{K}x{M} splitk{sk}", timeit(lambda: ops.gemm(dy, a, a_trans=True, b_trans=True, split_k=sk)), 2 * M * N * K)
            rec(f"torch.matmul {M}x{N}x{K}", timeit(lambda: torch.matmul(a, b.t())), 2 * M * N * K)
        if M == 8192:
            rec(f"torch.matmul {M}x{N}x{K}", timeit(lambda: torch.matmul(a, b.t())), 2 * M * N * K)
        del a, b
    # LayerNorm
    for rows, cols in [(20832, 1024), (88256, 768)]:
        x = torch.randn(rows, cols, device=DEV, dtype=BF)
        w = torch.ones(cols, device=DEV, dtype=BF)
        bb = torch.zeros(cols, device=DEV, dtype=BF)
        rec(f"layernorm fwd {rows}x{cols}", timeit(lambda: ops.layernorm_fwd(x, w, bb, 1e-5, False)), nbytes=2 * rows * cols * 2)
        y, mean, rstd = ops.layernorm_fwd(x, w, bb, 1e-5, True)
        rec(f"layernorm bwd {rows}x{cols}", timeit(lambda: ops.layernorm_bwd(x, x, w, mean, rstd, True)), nbytes=3 * rows * cols * 2)
        rec(f"torch layer_norm fwd {rows}x{cols}", timeit(lambda: torch.nn.functional.layer_norm(x, (cols,), w, bb, 1e-5)), nbytes=2 * rows * cols * 2)
    # attention
    from tests.gpu_checks import make_block_mask
    for (B, H, L, mk) in [(32, 16, 651, "dense"), (32, 16, 651, "trunk"), (448, 12, 197, "dense"), (448, 16, 265, "dense")]:
        qkv = torch.randn(B, L, 3 * H * 64, device=DEV, dtype=BF)
        v5 = qkv.view(B, L, 3, H, 64)
        mt = None
        vis = 1.0
        if mk == "trunk":
            mask = make_block_mask(L, 93, 36)
            mt = ops.build_mask_tables(mask, device=DEV)
            vis = float((mask == 0).float().mean())
        fl = 4 * B * H * L * L * 64
        f = lambda: ops.attn_fwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], scale=0.125, mask_tables=mt)
        rec(f"attn fwd B{B} H{H} L{L} {mk} (dense-equivalent flops, visible {vis:.2f})", timeit(f), fl)
        o, lse = f()
        do = torch.randn_like(o)
        dqkv = torch.empty_like(qkv)
        d5 = dqkv.view(B, L, 3, H, 64)
        g = lambda: ops.attn_bwd_raw(v5[:, :, 0], v5[:, :, 1], v5[:, :, 2], o, lse, do, d5[:, :, 0], d5[:, :, 1], d5[:, :, 2],
                                     scale=0.125, mask_tables=mt)
        rec(f"attn bwd B{B} H{H} L{L} {mk}", timeit(g), 2.5 * fl)
        if mk == "dense":
            q, k, v = [t.permute(0, 2, 1, 3) for t in (v5[:, :, 0], v5[:, :, 1], v5[:, :, 2])]
            rec(f"torch sdpa fwd B{B} H{H} L{L}", timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v)), fl)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "perf.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
