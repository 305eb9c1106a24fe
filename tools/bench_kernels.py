"""Micro-benchmarks of the individual HIP kernels at the SDXL / SD1.5 problem shapes.
Prints one JSON line per case: achieved TFLOP/s or GB/s (algorithmic work / HIP-event time)."""
import json
import sys
import os
os.environ.setdefault("FMX_ALLOW_KNOBS", "1")   # this tool A/Bs the library's development knobs

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import forge_amd  # noqa
from forge_amd import hipops as ops

DEV = "cuda"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def timeit_graph(fns, reps=5):
    """Launch-overhead-free time per call: the calls of `fns` captured into ONE HIP graph (back to back on the device, as in the
    graph-replayed UNet forward), replayed `reps` times.  Give every call its own weights when cold operands matter."""
    from forge_amd.runtime import HipGraph
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = HipGraph()
    with torch.cuda.stream(side):
        g.capture(side, lambda: [f() for f in fns])
        g.launch(side)
        side.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(side)
        for _ in range(reps):
            g.launch(side)
        e.record(side)
        side.synchronize()
    g.destroy()
    return s.elapsed_time(e) / (reps * len(fns)) * 1e-3


def bench_linear_cold(m, n, k, tile=0, act=0, copies=None):
    """as bench_linear, but inside a graph and with enough distinct weight matrices that none is cache-resident when its turn comes (> 256 MB
    in total: the forward reads every weight once per step, from HBM)"""
    copies = copies or max(8, int(300e6 / (n * k * 2)) + 1)
    x, b = rnd(m, k), rnd(n)
    ws = [rnd(n, k, scale=k ** -0.5) for _ in range(copies)]
    if copies == 1:
        ws = ws * 40     # one matrix, forty launches in the graph: hot weights
    out = torch.empty(m, n // 2 if act else n, dtype=torch.float16, device=DEV)
    t = timeit_graph([(lambda w=w: ops.conv_gemm(x, w, n, bias=b, out=out, ld_out=out.shape[1], act=act, force_tile=tile)) for w in ws])
    print(json.dumps({"op": "linear (in graph, cold weights)" if copies > 1 else "linear (in graph, ONE weight matrix)", "m": m, "n": n, "k": k, "tile": tile, "act": act, "us": round(t * 1e6, 1),
                      "tflops": round(2 * m * n * k / t / 1e12, 1)}), flush=True)


def bench_linear_res(m, n, k, tile=0, copies=8, residual=True, inplace=True):
    """Residual-adding projection (attn `to_out`, ff.net.2) as the forward runs it: in a graph, every call with its own weights AND its own
    residual stream (written long ago: cold), x += f(x) in place or out of place.  Isolates the epilogue's row pass (round 3)."""
    x, b = rnd(m, k), rnd(n)
    ws = [rnd(n, k, scale=k ** -0.5) for _ in range(copies)]
    rs = [rnd(m, n) for _ in range(copies)] if residual else [None] * copies
    outs = rs if (inplace and residual) else [torch.empty(m, n, dtype=torch.float16, device=DEV) for _ in range(copies)]
    t = timeit_graph([(lambda w=w, r=r, o=o: ops.conv_gemm(x, w, n, bias=b, residual=r, out=o, ld_out=n, force_tile=tile)) for w, r, o in zip(ws, rs, outs)])
    print(json.dumps({"op": "linear + residual (in graph, cold)" if residual else "linear (in graph, cold, distinct outputs)", "m": m, "n": n, "k": k,
                      "tile": tile, "inplace": bool(inplace and residual), "us": round(t * 1e6, 1), "tflops": round(2 * m * n * k / t / 1e12, 1),
                      "lib": os.environ.get("FMX_LIB", "default"), "mfma": os.environ.get("FMX_GEMM_MFMA", "32")}), flush=True)


def _graph_of(stream, fns):
    from forge_amd.runtime import HipGraph
    g = HipGraph()
    with torch.cuda.stream(stream):
        for f in fns[:2]:
            f()
        stream.synchronize()
        g.capture(stream, lambda: [f() for f in fns])
        g.launch(stream)
        stream.synchronize()
    return g


def _wall(graphs_streams, reps):
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for g, st in graphs_streams:
            g.launch(st)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def bench_dual(name, make_chain, flops, reps=20):
    """Round 3: ONE chain of kernels over the whole UNet batch on one stream, against TWO chains over half the batch each on two streams
    (the uncond and cond halves of a CFG step are independent).  A half-batch launch fills half the CUs; the two chains drift out of phase, so
    one chain's epilogue / launch ramp overlaps the other's K loop -- the overlap a 160-accumulator tile cannot get inside a CU.
    make_chain(part, parts) -> list of callables for part `part` of `parts` (parts = 1: the whole batch)."""
    s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    full = _graph_of(s0, make_chain(0, 1))
    h1, h2 = _graph_of(s1, make_chain(0, 2)), _graph_of(s2, make_chain(1, 2))
    out = {"op": "one full-batch chain vs two half-batch chains on two streams", "case": name}
    for rnd in range(2):
        out[f"full_us_{rnd}"] = round(_wall([(full, s0)], reps) * 1e6, 1)
        out[f"dual_us_{rnd}"] = round(_wall([(h1, s1), (h2, s2)], reps) * 1e6, 1)
        out[f"half_alone_us_{rnd}"] = round(_wall([(h1, s1)], reps) * 1e6, 1)
    # the same without graphs: eager launches, the two chains interleaved call by call from one host thread on two streams
    import time
    ffull, f1, f2 = make_chain(0, 1), make_chain(0, 2), make_chain(1, 2)

    def eager_full():
        with torch.cuda.stream(s0):
            for f in ffull:
                f()

    def eager_dual():
        for a, b in zip(f1, f2):
            with torch.cuda.stream(s1):
                a()
            with torch.cuda.stream(s2):
                b()
    for nm, fn in (("eager_full_us", eager_full), ("eager_dual_us", eager_dual)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        out[nm] = round((time.perf_counter() - t0) / reps * 1e6, 1)
    out["full_tflops"] = round(flops / (out["full_us_1"] * 1e-6) / 1e12, 1)
    out["dual_tflops"] = round(flops / (out["dual_us_1"] * 1e-6) / 1e12, 1)
    print(json.dumps(out), flush=True)
    for g in (full, h1, h2):
        g.destroy()


def dual_cases():
    Bu = 16

    def transformer_like(m_img, c, heads, nblk=4):
        """per block: q+k projection (N = 2c), V^T projection, self-attention, to_out (+residual), to_q, (77-key attention skipped), to_out (+residual),
        GEGLU ff1, ff2 (+residual) -- the GEMM / attention sequence of a BasicTransformerBlock at the SDXL sizes, cold weights per block"""
        m = Bu * m_img
        d = c // heads
        x = rnd(m, c)
        w_qk = [rnd(2 * c, c, scale=c ** -0.5) for _ in range(nblk)]
        w_v = [rnd(c, c, scale=c ** -0.5) for _ in range(nblk)]
        w_o = [rnd(c, c, scale=c ** -0.5) for _ in range(2 * nblk)]
        w_q2 = [rnd(c, c, scale=c ** -0.5) for _ in range(nblk)]
        w_f1 = [rnd(8 * c, c, scale=c ** -0.5) for _ in range(nblk)]
        w_f2 = [rnd(c, 4 * c, scale=(4 * c) ** -0.5) for _ in range(nblk)]
        bias = {n: rnd(n) for n in (c, 2 * c, 8 * c)}
        stream_x = rnd(m, c)
        qk = torch.empty(m, 2 * c, dtype=torch.float16, device=DEV)
        vt = rnd(heads, c // heads, Bu, m_img)    # V^T as its projection GEMM leaves it: [head][channel][image][key]
        ao = torch.empty(m, c, dtype=torch.float16, device=DEV)
        q2 = torch.empty(m, c, dtype=torch.float16, device=DEV)
        hid = torch.empty(m, 4 * c, dtype=torch.float16, device=DEV)
        flops = nblk * (2 * m * c * (2 * c + c + c + c + c + 8 * c + 4 * c) + 4 * Bu * heads * m_img * m_img * d)

        def chain(part, parts):
            lo, hi = part * m // parts, (part + 1) * m // parts
            blo, bhi = part * Bu // parts, (part + 1) * Bu // parts
            fns = []
            for i in range(nblk):
                fns.append(lambda i=i: ops.conv_gemm(x[lo:hi], w_qk[i], 2 * c, bias=bias[2 * c], out=qk[lo:hi], ld_out=2 * c))
                fns.append(lambda i=i: ops.conv_gemm(x[lo:hi], w_v[i], c, bias=bias[c], out=q2[lo:hi], ld_out=c))   # (V projection; the transposed form has the same cost)
                fns.append(lambda: ops.attention(qk[lo:hi], qk[lo:hi, c:], vt[:, :, blo:bhi], batch=bhi - blo, heads=heads, nq=m_img, nk=m_img, nk_pad=m_img, dpad=d,
                                                 scale=d ** -0.5, q_bs=m_img * 2 * c, q_rs=2 * c, k_bs=m_img * 2 * c, k_rs=2 * c, vt_bs=m_img, vt_hs=d * Bu * m_img,
                                                 vt_ds=Bu * m_img, out=ao[lo:hi]))
                fns.append(lambda i=i: ops.conv_gemm(ao[lo:hi], w_o[2 * i], c, bias=bias[c], residual=stream_x[lo:hi], out=stream_x[lo:hi], ld_out=c))
                fns.append(lambda i=i: ops.conv_gemm(stream_x[lo:hi], w_q2[i], c, bias=bias[c], out=q2[lo:hi], ld_out=c))
                fns.append(lambda i=i: ops.conv_gemm(q2[lo:hi], w_o[2 * i + 1], c, bias=bias[c], residual=stream_x[lo:hi], out=stream_x[lo:hi], ld_out=c))
                fns.append(lambda i=i: ops.conv_gemm(stream_x[lo:hi], w_f1[i], 8 * c, bias=bias[8 * c], out=hid[lo:hi], ld_out=4 * c, act=1))
                fns.append(lambda i=i: ops.conv_gemm(hid[lo:hi], w_f2[i], c, bias=bias[c], residual=stream_x[lo:hi], out=stream_x[lo:hi], ld_out=c))
            return fns
        return chain, flops

    chain, fl = transformer_like(1024, 1280, 20)
    bench_dual("4 transformer blocks at 32x32 (1280 wide, 20 heads), UNet batch 16", chain, fl)
    chain, fl = transformer_like(4096, 640, 10, nblk=2)
    bench_dual("2 transformer blocks at 64x64 (640 wide, 10 heads), UNet batch 16", chain, fl)

    def conv_like(h, c, nconv=4):
        x = torch.randn(Bu, h, h, c, device=DEV).half()
        ws = [rnd(c, 9 * c, scale=(9 * c) ** -0.5) for _ in range(nconv)]
        b = rnd(c)
        ys = [torch.empty(Bu, h, h, c, dtype=torch.float16, device=DEV) for _ in range(2)]
        flops = nconv * 2 * Bu * h * h * c * 9 * c

        def chain(part, parts):
            blo, bhi = part * Bu // parts, (part + 1) * Bu // parts
            fns = []
            for i in range(nconv):
                src = x if i == 0 else ys[(i - 1) & 1]
                fns.append(lambda i=i, src=src: ops.conv_gemm(src[blo:bhi], ws[i], c, kh=3, pad=1, bias=b, out=ys[i & 1][blo:bhi].view(-1, c), ld_out=c))
            return fns
        return chain, flops
    for h, c in ((32, 1280), (64, 640), (128, 320)):
        chain, fl = conv_like(h, c)
        bench_dual(f"4 convolutions 3x3 {c}->{c} at {h}x{h}, UNet batch 16", chain, fl)


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, device=DEV) * scale).half()


def bench_linear(m, n, k, tile=0, act=0):
    x, w, b = rnd(m, k), rnd(n, k, scale=k ** -0.5), rnd(n)
    out = torch.empty(m, n // 2 if act else n, dtype=torch.float16, device=DEV)
    t = timeit(lambda: ops.conv_gemm(x, w, n, bias=b, out=out, ld_out=out.shape[1], act=act, force_tile=tile))
    print(json.dumps({"op": "linear", "m": m, "n": n, "k": k, "tile": tile, "act": act, "us": round(t * 1e6, 1), "tflops": round(2 * m * n * k / t / 1e12, 1)}), flush=True)


def bench_conv(n, h, w, c, co, tile=0, stride=1, up=None):
    x, wk, b = rnd(n, h, w, c), rnd(co, 9 * c, scale=(9 * c) ** -0.5), rnd(co)
    oh, ow = (up or (h, w))
    oh, ow = oh // stride, ow // stride
    out = torch.empty(n * oh * ow, co, dtype=torch.float16, device=DEV)
    t = timeit(lambda: ops.conv_gemm(x, wk, co, kh=3, pad=1, stride=stride, up=up, bias=b, out=out, ld_out=co, force_tile=tile))
    print(json.dumps({"op": "conv3x3", "n": n, "hw": [h, w], "c": c, "co": co, "stride": stride, "up": up, "tile": tile, "us": round(t * 1e6, 1),
                      "tflops": round(2 * n * oh * ow * co * 9 * c / t / 1e12, 1)}), flush=True)


def bench_conv_cold(n, h, w, c, co, tile=0, stride=1, up=None, copies=None, stats=False):
    """3x3 convolution in a graph, every launch on its own weights (cold), optionally with the output statistics"""
    copies = copies or max(4, int(300e6 / (co * 9 * c * 2)) + 1)
    x, b = rnd(n, h, w, c), rnd(co)
    ws = [rnd(co, 9 * c, scale=(9 * c) ** -0.5) for _ in range(copies)]
    oh, ow = (up or (h, w))
    oh, ow = oh // stride, ow // stride
    out = torch.empty(n * oh * ow, co, dtype=torch.float16, device=DEV)
    t = timeit_graph([(lambda wk=wk: ops.conv_gemm(x, wk, co, kh=3, pad=1, stride=stride, up=up, bias=b, out=out, ld_out=co, force_tile=tile, stats=stats)) for wk in ws])
    print(json.dumps({"op": "conv3x3 (in graph, cold weights)", "n": n, "hw": [h, w], "c": c, "co": co, "stride": stride, "up": up, "stats": stats, "tile": tile,
                      "M": n * oh * ow, "K": 9 * c, "us": round(t * 1e6, 1), "tflops": round(2 * n * oh * ow * co * 9 * c / t / 1e12, 1)}), flush=True)


def bench_attn(b, h, nq, nk, d, dpad, force32=False):
    nkp = -(-nk // 64) * 64
    q, k = rnd(b, nq, h, dpad), rnd(b, nkp, h, dpad)
    vt = rnd(h, dpad, b, nkp)
    out = torch.empty(b * nq, h * dpad, dtype=torch.float16, device=DEV)
    t = timeit(lambda: ops.attention(q, k, vt, batch=b, heads=h, nq=nq, nk=nk, nk_pad=nkp, dpad=dpad, scale=d ** -0.5, q_bs=nq * h * dpad,
                                     q_rs=h * dpad, k_bs=nkp * h * dpad, k_rs=h * dpad, vt_bs=nkp, vt_hs=dpad * b * nkp, vt_ds=b * nkp, out=out, force32=force32))
    print(json.dumps({"op": "attention", "b": b, "h": h, "nq": nq, "nk": nk, "d": d, "force32": force32, "us": round(t * 1e6, 1),
                      "tflops": round(4 * b * h * nq * nk * d / t / 1e12, 1), "q_plus_o_GBps": round(4 * b * h * nq * dpad / t / 1e9, 1),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("FMX_ATTN")}}), flush=True)


def bench_attn_graph(b, h, nq, nk, d, copies=4):
    """attention launches as the forward runs them: back to back inside ONE graph (no host launch gap), each on its own Q / O tensors"""
    nkp = -(-nk // 64) * 64
    k = rnd(b, nkp, h, d)
    vt = rnd(h, d, b, nkp)
    qs = [rnd(b, nq, h, d) for _ in range(copies)]
    outs = [torch.empty(b * nq, h * d, dtype=torch.float16, device=DEV) for _ in range(copies)]
    t = timeit_graph([(lambda q=q, o=o: ops.attention(q, k, vt, batch=b, heads=h, nq=nq, nk=nk, nk_pad=nkp, dpad=d, scale=d ** -0.5, q_bs=nq * h * d,
                                                       q_rs=h * d, k_bs=nkp * h * d, k_rs=h * d, vt_bs=nkp, vt_hs=d * b * nkp, vt_ds=b * nkp, out=o))
                      for q, o in zip(qs, outs)], reps=10)
    print(json.dumps({"op": "attention (in graph)", "b": b, "h": h, "nq": nq, "nk": nk, "d": d, "us": round(t * 1e6, 1),
                      "tflops": round(4 * b * h * nq * nk * d / t / 1e12, 1), "q_plus_o_GBps": round(4 * b * h * nq * d / t / 1e9, 1),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("FMX_ATTN")}}), flush=True)


def bench_attn512(b, n):
    """the VAE mid-block attention: one 512-wide head over n tokens, fused kernel vs the materialised-score path of round 1"""
    c = 512
    qk, vt = rnd(b * n, 2 * c), rnd(c, b * n)
    o = torch.empty(b * n, c, dtype=torch.float16, device=DEV)
    t = timeit(lambda: ops.attention_single_head512(qk, qk[:, c:], vt, o, batch=b, nq=n, nk=n, nk_pad=n, q_bs=n * 2 * c, q_rs=2 * c, k_bs=n * 2 * c,
                                                    k_rs=2 * c, vt_bs=n, vt_ds=b * n, scale=c ** -0.5), iters=5, warm=2)
    s = torch.empty(n, n, dtype=torch.float16, device=DEV)

    def materialised():
        for bi in range(b):
            ops.conv_gemm(qk[bi * n:(bi + 1) * n, :c], qk[bi * n:(bi + 1) * n, c:], n, alpha=c ** -0.5, out=s, ld_out=n)
            ops.softmax_rows_(s)
            ops.conv_gemm(s, vt[:, bi * n:(bi + 1) * n], c, out=o[bi * n:(bi + 1) * n], ld_out=c)
    t2 = timeit(materialised, iters=3, warm=1)
    fl = 4.0 * b * n * n * c
    print(json.dumps({"op": "attention_single_head_512", "b": b, "n": n, "fused_us": round(t * 1e6, 1), "fused_tflops": round(fl / t / 1e12, 1),
                      "materialised_us": round(t2 * 1e6, 1), "materialised_tflops": round(fl / t2 / 1e12, 1)}), flush=True)


def bench_gn(n, h, w, c):
    x, g, bb = rnd(n, h, w, c), rnd(c), rnd(c)
    out = torch.empty_like(x)
    t = timeit(lambda: ops.groupnorm(x, g, bb, 1e-5, silu=True, out=out))
    by = x.numel() * 2
    print(json.dumps({"op": "groupnorm_silu", "n": n, "hw": [h, w], "c": c, "us": round(t * 1e6, 1), "GBps_alg(1R+1W)": round(2 * by / t / 1e9, 0),
                      "GBps_touched(2R+1W)": round(3 * by / t / 1e9, 0)}), flush=True)


def bench_gn_apply_graph(n, h, w, c, copies=3):
    """GroupNorm apply (+SiLU) on statistics the producer left, as the forward runs it: in a graph, every launch on its own tensors"""
    g, bb = rnd(c), rnd(c)
    xs = [rnd(n, h, w, c) for _ in range(copies)]
    outs = [torch.empty_like(x) for x in xs]
    sts = [ops.groupnorm_stats(x) for x in xs]
    t = timeit_graph([(lambda x=x, o=o, st=st: ops.groupnorm(x, g, bb, 1e-5, silu=True, out=o, stats=st)) for x, o, st in zip(xs, outs, sts)], reps=10)
    by = xs[0].numel() * 2
    print(json.dumps({"op": "groupnorm finalize + apply + SiLU (in graph)", "n": n, "hw": [h, w], "c": c, "us": round(t * 1e6, 1), "GBps(1R+1W)": round(2 * by / t / 1e9, 0),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("FMX_GN")}}), flush=True)


def bench_gn_paths(n, h, w, c):
    """GroupNorm + SiLU three ways: own statistics pass (stats + finalize + apply), on statistics the producer left (finalize + apply),
    and the producer side: the same conv / linear with and without statistics in its epilogue."""
    x, g, bb = rnd(n, h, w, c), rnd(c), rnd(c)
    out = torch.empty_like(x)
    by = x.numel() * 2
    t_own = timeit(lambda: ops.groupnorm(x, g, bb, 1e-5, silu=True, out=out))
    st = ops.groupnorm_stats(x)
    t_st = timeit(lambda: ops.groupnorm_stats(x))
    t_app = timeit(lambda: ops.groupnorm(x, g, bb, 1e-5, silu=True, out=out, stats=st))
    print(json.dumps({"op": "groupnorm_silu", "n": n, "hw": [h, w], "c": c, "own_stats_us": round(t_own * 1e6, 1), "stats_pass_us": round(t_st * 1e6, 1),
                      "apply_only_us": round(t_app * 1e6, 1), "apply_GBps(1R+1W)": round(2 * by / t_app / 1e9, 0),
                      "own_GBps_alg(1R+1W)": round(2 * by / t_own / 1e9, 0), "stats_GBps(1R)": round(by / t_st / 1e9, 0)}), flush=True)


def bench_gemm_stats(n, h, w, cin, co, kh):
    x, wk, b = rnd(n, h, w, cin), rnd(co, kh * kh * cin, scale=(kh * kh * cin) ** -0.5), rnd(co)
    res = rnd(n * h * w, co)
    out = torch.empty(n * h * w, co, dtype=torch.float16, device=DEV)
    part = ops.stats_buffer(n, h * w, co, device=DEV)
    t0 = timeit(lambda: ops.conv_gemm(x, wk, co, kh=kh, pad=kh // 2, bias=b, residual=res, out=out, ld_out=co))
    t1 = timeit(lambda: ops.conv_gemm(x, wk, co, kh=kh, pad=kh // 2, bias=b, residual=res, out=out, ld_out=co, stats=True, stats_partial=part))
    _, st = ops.conv_gemm(x, wk, co, kh=kh, pad=kh // 2, bias=b, residual=res, out=out, ld_out=co, stats=True, stats_partial=part)
    fl = 2.0 * n * h * w * co * kh * kh * cin
    print(json.dumps({"op": "gemm_with_gn_stats", "n": n, "hw": [h, w], "cin": cin, "co": co, "kh": kh, "plain_us": round(t0 * 1e6, 1), "with_stats_us": round(t1 * 1e6, 1),
                      "delta_us": round((t1 - t0) * 1e6, 1), "plain_tflops": round(fl / t0 / 1e12, 1), "with_stats_tflops": round(fl / t1 / 1e12, 1),
                      "chunks": None if st is None else st.nchunks, "fused": st is not None and st.nchunks == (h * w) // 256}), flush=True)


def bench_ln(rows, c):
    x, g, bb = rnd(rows, c), rnd(c), rnd(c)
    out = torch.empty_like(x)
    t = timeit(lambda: ops.layernorm(x, g, bb, out=out))
    print(json.dumps({"op": "layernorm", "rows": rows, "c": c, "us": round(t * 1e6, 1), "GBps": round(2 * x.numel() * 2 / t / 1e9, 0)}), flush=True)


def gemm_sweep():
    """every GEMM-shaped problem of one SDXL 1024^2 B=8 (Bu=16) UNet forward, old tiles (1,2,3) vs 256x256 (4) vs auto (0)"""
    Bu = 16
    lin = [(Bu * 1024, 1280, 1280, 0), (Bu * 1024, 2560, 1280, 0), (1280, Bu * 1024, 1280, 0), (Bu * 1024, 10240, 1280, 1), (Bu * 1024, 1280, 5120, 0),
           (Bu * 4096, 640, 640, 0), (Bu * 4096, 1280, 640, 0), (640, Bu * 4096, 640, 0), (Bu * 4096, 5120, 640, 1), (Bu * 4096, 640, 2560, 0),
           (8192, 8192, 8192, 0), (4096, 4096, 4096, 0)]
    for m, n, k, act in lin:
        for tile in ((6, 7, 0) if act else (5, 6, 7, 0) if n % 320 == 0 else (1, 6, 7, 0)):
            bench_linear(m, n, k, tile, act=act)
    conv = [(Bu, 32, 32, 1280, 1280, 1, None), (Bu, 32, 32, 2560, 1280, 1, None), (Bu, 64, 64, 640, 640, 1, None), (Bu, 64, 64, 1920, 640, 1, None),
            (Bu, 128, 128, 320, 320, 1, None), (Bu, 128, 128, 960, 320, 1, None), (Bu, 32, 32, 1280, 1280, 1, (64, 64)), (Bu, 64, 64, 640, 640, 1, (128, 128)),
            (Bu, 128, 128, 320, 320, 2, None), (Bu, 64, 64, 640, 640, 2, None), (8, 256, 256, 512, 512, 1, None), (8, 512, 512, 256, 256, 1, None)]
    for n, h, w, c, co, stride, up in conv:
        for tile in ((5, 6, 7, 0) if co % 160 == 0 else (1, 6, 7, 0)):
            bench_conv(n, h, w, c, co, tile, stride=stride, up=up)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gemm4w":
        # round 4: the 256x160 tile with two 4-wave workgroups per CU (force_tile 10) against the 256x320 kernel (7), same process,
        # interleaved twice, in a graph on cold weights -- the K <= 1280 linear shapes of the SDXL forward, the long-K one for reference
        for rep in range(2):
            for (m, n, k, act) in [(16384, 1280, 1280, 0), (16384, 2560, 1280, 0), (16384, 10240, 1280, 1), (65536, 640, 640, 0), (65536, 5120, 640, 1),
                                   (16384, 1280, 5120, 0), (65536, 640, 2560, 0)]:
                for tile in (7, 10):
                    bench_linear_cold(m, n, k, tile=tile, act=act)
            for (m, n, k) in [(16384, 1280, 1280), (65536, 640, 640)]:
                for tile in (7, 10):
                    bench_linear_res(m, n, k, tile=tile)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gn":
        for shp in ((16, 128, 128, 320), (16, 128, 128, 640), (16, 64, 64, 640), (16, 64, 64, 1280), (16, 32, 32, 1280), (16, 32, 32, 2560), (8, 1024, 1024, 128),
                    (8, 512, 512, 256), (8, 256, 256, 512), (2, 64, 64, 320)):
            bench_gn_paths(*shp)
        for shp in ((16, 32, 32, 1280, 1280, 1), (16, 64, 64, 640, 640, 1), (16, 128, 128, 320, 320, 1), (16, 32, 32, 1280, 1280, 3), (16, 64, 64, 640, 640, 3),
                    (16, 128, 128, 320, 320, 3), (8, 512, 512, 256, 256, 3), (8, 1024, 1024, 128, 128, 3)):
            bench_gemm_stats(*shp)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gemmsched":
        # the shapes that carry the SDXL forward, on the 256x320 tile (run under FMX_GEMM_SCHED=0/1/2 to compare DMA issue schedules)
        Bu = 16
        for m, n, k, act in ((Bu * 1024, 1280, 1280, 0), (Bu * 1024, 10240, 1280, 1), (Bu * 1024, 1280, 5120, 0), (Bu * 1024, 2560, 1280, 0),
                             (Bu * 4096, 640, 640, 0), (Bu * 4096, 5120, 640, 1), (Bu * 4096, 640, 2560, 0)):
            bench_linear(m, n, k, 7, act=act)
        bench_conv(Bu, 32, 32, 1280, 1280, 7)
        bench_conv(Bu, 64, 64, 640, 640, 7)
        bench_conv(Bu, 128, 128, 320, 320, 7)
        bench_linear(Bu * 1024, 10240, 1280, 7, act=0)   # the GEGLU projection's shape without GEGLU: what the activation epilogue costs
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "splitk":
        # interactive-batch shapes (UNet batch 2 and 4): split-K factor sweep on the two 4-wave tiles (FMX_GEMM_SPLITK), and the dispatcher's choice
        import os
        shapes = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (1280, 2048, 1280), (8192, 640, 640), (8192, 640, 2560),
                  (4096, 1280, 1280), (4096, 1280, 5120)]
        for m, n, k in shapes:
            for tile in (1, 2):
                for s in (0, 2, 3, 4, 6, 8):
                    os.environ["FMX_GEMM_SPLITK"] = str(s)
                    print(json.dumps({"splitk": s}), end=" ")
                    bench_linear_cold(m, n, k, tile)
            for s in ("0", None):
                if s is None:
                    os.environ.pop("FMX_GEMM_SPLITK", None)
                else:
                    os.environ["FMX_GEMM_SPLITK"] = s
                print(json.dumps({"splitk": "auto" if s is None else "off", "dispatcher": True}), end=" ")
                bench_linear_cold(m, n, k, 0)
        for co, c, hw in ((1280, 1280, 32), (1280, 2560, 32), (640, 640, 64)):
            for s in ("0", None):
                if s is None:
                    os.environ.pop("FMX_GEMM_SPLITK", None)
                else:
                    os.environ["FMX_GEMM_SPLITK"] = s
                print(json.dumps({"splitk": "auto" if s is None else "off", "dispatcher": True}), end=" ")
                bench_conv(2, hw, hw, c, co, 0)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ring":
        # round 5: the 4-wave tiles on the 4-stage LDS ring (force_tile 11 / 12 / 13 = 128x128 / 128x160 / 128x64) against their 2-stage forms (1 / 5 / 2)
        # and the dispatcher's choice with the ring allowed / forbidden, at the small-M shapes of SD1.5 batch 4 and SDXL batch 1; in a graph, cold weights
        import os
        shapes = [(2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (1280, 2048, 1280), (2048, 1280, 2560), (512, 1280, 1280), (512, 1280, 5120),
                  (8192, 640, 640), (8192, 640, 2560), (8192, 1280, 640), (4096, 1280, 1280), (32768, 320, 320)]
        for m, n, k in shapes:
            os.environ["FMX_GEMM_SPLITK"] = "0"
            for tile in (1, 11, 2, 13) + ((5, 12) if n % 160 == 0 else ()):
                bench_linear_cold(m, n, k, tile)
            os.environ.pop("FMX_GEMM_SPLITK", None)
            for ring in ("0", "1"):
                os.environ["FMX_GEMM_RING"] = ring
                print(json.dumps({"dispatcher": True, "ring": ring}), end=" ")
                bench_linear_cold(m, n, k, 0)
            os.environ.pop("FMX_GEMM_RING", None)
        # the same launches on ONE weight matrix (hot in L2 / the memory-side cache): what a prefetch of the next layer's weights could buy
        os.environ["FMX_GEMM_SPLITK"] = "0"
        for m, n, k in ((2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280)):
            for tile in (1, 11, 12):
                print(json.dumps({"weights": "hot (one matrix)"}), end=" ")
                bench_linear_cold(m, n, k, tile, copies=1)
        os.environ.pop("FMX_GEMM_SPLITK", None)
        for s in (2, 3, 4):
            os.environ["FMX_GEMM_SPLITK"] = str(s)
            for m, n, k in ((2048, 1280, 5120), (512, 1280, 5120)):
                for tile in (1, 11):
                    print(json.dumps({"splitk": s}), end=" ")
                    bench_linear_cold(m, n, k, tile)
        os.environ.pop("FMX_GEMM_SPLITK", None)
        for co, c, hw, nb in ((1280, 1280, 32, 2), (1280, 2560, 32, 2), (640, 640, 64, 2), (1280, 1280, 16, 8), (1280, 2560, 16, 8), (1280, 1280, 8, 8), (1280, 2560, 8, 8),
                              (640, 640, 32, 8), (640, 1280, 32, 8)):
            for ring in ("0", "1"):
                os.environ["FMX_GEMM_RING"] = ring
                print(json.dumps({"dispatcher": True, "ring": ring}), end=" ")
                bench_conv(nb, hw, hw, c, co, 0)
            os.environ.pop("FMX_GEMM_RING", None)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ringord":
        # round 5: every ring tile (force_tile 11..15) on the small-M shapes, one process per FMX_RING_ORDERED setting (the knob is read once)
        import os
        os.environ["FMX_GEMM_SPLITK"] = "0"
        tag = {"FMX_RING_ORDERED": os.environ.get("FMX_RING_ORDERED", "default")}
        for m, n, k in ((2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (1280, 2048, 1280), (8192, 640, 640), (8192, 640, 2560), (512, 1280, 1280),
                        (4096, 1280, 1280), (32768, 320, 320), (32768, 320, 1280)):
            for tile in (11, 12, 13, 14, 15, 0):
                if tile == 12 and n % 160:
                    continue
                print(json.dumps(tag), end=" ")
                bench_linear_cold(m, n, k, tile)
        for (n, hw, c, co) in ((2, 32, 1280, 1280), (2, 64, 640, 640), (8, 8, 1280, 1280), (8, 16, 1280, 1280), (2, 128, 320, 320)):
            for tile in (11, 12, 13, 14, 0):
                print(json.dumps(tag), end=" ")
                bench_conv_cold(n, hw, hw, c, co, tile, stats=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "tile14":
        import os
        os.environ["FMX_GEMM_SPLITK"] = "0"
        for m, n, k in ((2048, 1280, 1280), (2048, 1280, 5120), (2048, 1280, 2560), (2048, 2560, 1280), (8192, 640, 640), (8192, 640, 2560), (512, 1280, 1280), (4096, 1280, 1280)):
            for tile in (11, 12, 14):
                bench_linear_cold(m, n, k, tile)
        for (n, hw, c, co) in ((2, 32, 1280, 1280), (2, 64, 640, 640), (8, 8, 1280, 1280)):
            for tile in (11, 12, 14):
                bench_conv_cold(n, hw, hw, c, co, tile)
        for m, n, k in ((1280, 2048, 1280), (640, 8192, 640), (1280, 512, 1280)):
            for tile in (11, 13, 15, 0):
                bench_linear_cold(m, n, k, tile)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "smallconv":
        # round 5: the convolutions of SD1.5 batch 4 / SDXL batch 1 that leave CUs idle, tile by tile (0 = dispatcher)
        import os
        os.environ["FMX_GEMM_SPLITK"] = "0"
        for (n, hw, c, co, up) in ((2, 32, 1280, 1280, (64, 64)), (2, 128, 320, 320, None), (2, 128, 640, 320, None), (2, 64, 640, 640, None), (2, 64, 1280, 640, None)):
            for tile in (5, 12, 7, 6, 1, 11):
                try:
                    bench_conv_cold(n, hw, hw, c, co, tile, up=up, stats=True)
                except Exception as e:
                    print(json.dumps({"hw": hw, "c": c, "co": co, "tile": tile, "error": str(e)[:100]}), flush=True)
        os.environ.pop("FMX_GEMM_SPLITK", None)
        for (n, hw, c, co, up) in ((2, 32, 1280, 1280, (64, 64)), (2, 128, 320, 320, None), (2, 128, 640, 320, None), (2, 64, 640, 640, None), (2, 64, 1280, 640, None),
                                   (2, 32, 1280, 1280, None), (8, 8, 1280, 1280, None)):
            print(json.dumps({"dispatcher": True}), end=" ")
            bench_conv_cold(n, hw, hw, c, co, 0, up=up, stats=True)
        for m, n, k in ((32768, 320, 512), (32768, 320, 1280), (32768, 320, 320), (8192, 640, 640), (8192, 1280, 640)):
            for tile in (5, 12, 7, 6, 2, 13, 0):
                bench_linear_cold(m, n, k, tile)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "coldhot":
        # the big linear shapes of the batch-8 forward: eager back-to-back on ONE weight matrix (hot in L2 / MALL) vs in a graph with a
        # different weight matrix per call (cold, as in the forward)
        for m, n, k, act in ((16384, 1280, 1280, 0), (16384, 10240, 1280, 1), (16384, 1280, 5120, 0), (16384, 2560, 1280, 0), (65536, 640, 640, 0)):
            bench_linear(m, n, k, 7, act)
            bench_linear_cold(m, n, k, 7, act)
            bench_linear_cold(m, n, k, 7, act, copies=1 if False else 2)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "narrow":
        # the VAE decoder's 128-output-channel level: 4-wave 128x128 tile (1), 256x256 (6, half of it padding), 512x128 (9), dispatcher (0)
        for c in (128, 256):
            for tile in (1, 6, 9, 0):
                bench_conv(4, 1024, 1024, c, 128, tile)
        for tile in (6, 7, 9, 0):
            bench_conv(8, 512, 512, 256, 256, tile)
            bench_conv(8, 256, 256, 512, 384, tile)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "convgn":
        # GroupNorm + SiLU + conv3x3 in one kernel (csrc/fmx_conv_patch.hip) against groupnorm + conv_gemm, the VAE decoder's full-resolution level
        import math
        for n, hh, ww, cin in ((8, 1024, 1024, 128), (4, 1024, 1024, 256), (8, 512, 512, 128)):
            x = (rnd(n, hh, ww, cin) + 0.3)
            g, b = 1 + 0.1 * rnd(cin), 0.1 * rnd(cin)
            wk = rnd(128, 9 * cin, scale=1 / math.sqrt(9 * cin))
            bias = rnd(128)
            res = rnd(n * hh * ww, 128)
            st = ops.groupnorm_stats(x)
            flops = 2.0 * n * hh * ww * 128 * 9 * cin
            for with_res in (False, True):
                r = res if with_res else None
                out = ops.empty((n * hh * ww, 128), torch.float16)
                t_f = timeit(lambda: ops.conv3x3_gn_silu(x, g, b, 1e-6, wk, bias, residual=r, out=out, stats=st), iters=10)

                def two():
                    gn = ops.groupnorm(x, g, b, 1e-6, silu=True, stats=st)
                    ops.conv_gemm(gn, wk, 128, kh=3, pad=1, bias=bias, residual=r, out=out, ld_out=128, stats=True)
                t_2 = timeit(two, iters=10)
                gn = ops.groupnorm(x, g, b, 1e-6, silu=True, stats=st)
                t_c = timeit(lambda: ops.conv_gemm(gn, wk, 128, kh=3, pad=1, bias=bias, residual=r, out=out, ld_out=128, stats=True), iters=10)
                print(json.dumps({"case": f"gn+silu+conv3x3 n={n} {hh}x{ww} cin={cin} cout=128 residual={with_res}", "fused_ms": round(t_f * 1e3, 3),
                                  "fused_tflops": round(flops / t_f / 1e12, 1), "groupnorm_plus_conv_gemm_ms": round(t_2 * 1e3, 3),
                                  "conv_gemm_alone_ms": round(t_c * 1e3, 3), "conv_gemm_alone_tflops": round(flops / t_c / 1e12, 1),
                                  "lib": os.environ.get("FMX_LIB", "")}), flush=True)
                del gn
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attn512":
        for b, n in ((8, 16384), (8, 4096), (1, 16384), (2, 1024)):
            bench_attn512(b, n)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attnpoly":
        # round 5: attn_q64v2 with a fraction of its exponentials as a packed-fp16 polynomial (FMX_ATTN_POLY = pairs per 16-key group, read once per process):
        # time at 4096 / 1024 keys and the error against fp32 softmax attention on peaked and flat score distributions
        import os
        bench_attn(16, 10, 4096, 4096, 64, 64)
        bench_attn(16, 20, 1024, 1024, 64, 64)
        bench_attn(16, 10, 4096, 4096, 64, 64)
        bench_attn(16, 20, 1024, 1024, 64, 64)
        for name, qs in (("peaked (|q| = 1)", 1.0), ("flat (|q| = 0.1)", 0.1)):
            g = torch.Generator().manual_seed(1)
            b, h, n, d = 2, 4, 1024, 64
            q, k, v = (torch.randn(b, n, h * d, generator=g) for _ in range(3))
            q = (q * qs).half().to(DEV); k = k.half().to(DEV); v = v.half().to(DEV)
            vt = v.view(b * n, h, d).permute(1, 2, 0).reshape(h * d, b * n).contiguous()
            o = ops.attention(q.view(b * n, h * d), k.view(b * n, h * d), vt, batch=b, heads=h, nq=n, nk=n, nk_pad=n, dpad=d, scale=d ** -0.5, q_bs=n * h * d, q_rs=h * d,
                              k_bs=n * h * d, k_rs=h * d, vt_bs=n, vt_hs=d * b * n, vt_ds=b * n)
            qf, kf, vf = (t.float().view(b, n, h, d).permute(0, 2, 1, 3) for t in (q, k, v))
            ref = torch.softmax(qf @ kf.transpose(-1, -2) * d ** -0.5, -1) @ vf
            got = o.view(b, n, h, d).permute(0, 2, 1, 3).float()
            print(json.dumps({"poly": os.environ.get("FMX_ATTN_POLY", "0"), "scores": name, "rms_rel_vs_fp32": float(((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())),
                              "max_rel_vs_fp32": float((got - ref).abs().max() / ref.abs().max())}), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attn128":
        # d_head 128 (Flux-dev at 1024^2: 4096 image + 256 text tokens, 24 heads) and two neighbours, in a graph on their own tensors
        for b, h, n in ((2, 24, 4352), (1, 24, 4352), (4, 24, 4352), (2, 24, 1280)):
            bench_attn(b, h, n, n, 128, 128, False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attn":
        for f32 in (True, False):
            bench_attn(16, 10, 4096, 4096, 64, 64, f32)
            bench_attn(16, 20, 1024, 1024, 64, 64, f32)
            bench_attn(16, 20, 1024, 77, 64, 64, f32)
            bench_attn(16, 10, 4096, 77, 64, 64, f32)
            bench_attn(2, 10, 4096, 4096, 64, 64, f32)
            bench_attn(8, 10, 4096, 4096, 64, 64, f32)   # batch 4 under CFG: 1280 tiles = 2.5 rounds
            bench_attn(8, 20, 1024, 1024, 64, 64, f32)   # 640 tiles = 1.25 rounds
            bench_attn(2, 20, 1024, 1024, 64, 64, f32)   # batch 1 under CFG: 160 tiles
            bench_attn(2, 24, 4352, 4352, 128, 128, f32)  # Flux-dev at 1024^2: 4096 image + 256 text tokens, 24 heads of 128
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "dual":
        dual_cases()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "attnshort":
        # the 77-token cross-attention launches of the SDXL forward at UNet batch 16 (A/B through FMX_ATTN_SHORT / FMX_ATTN_SHORT_WGS)
        for _ in range(2):
            bench_attn_graph(16, 20, 1024, 77, 64)
            bench_attn_graph(16, 10, 4096, 77, 64)
        bench_attn_graph(2, 20, 1024, 77, 64)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gnapply":
        for shp in ((16, 128, 128, 320), (16, 64, 64, 640), (16, 32, 32, 1280), (16, 32, 32, 2560), (8, 1024, 1024, 128), (8, 512, 512, 256)):
            bench_gn_apply_graph(*shp)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "epi":
        # round 3: what the epilogue costs.  Full chip (256 tiles), half chip (128 tiles: is the row pass contention-bound?), two rounds, long K
        for m, n, k in ((16384, 1280, 1280), (8192, 1280, 1280), (16384, 2560, 1280), (16384, 1280, 5120), (65536, 640, 640), (65536, 640, 2560)):
            bench_linear_res(m, n, k, residual=False)
            bench_linear_res(m, n, k, residual=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "smallm":
        # batch-1 shapes (UNet batch 2): which tile is fastest, and what does the dispatcher (tile 0) pick?
        for m, n, k in ((2048, 1280, 1280), (2048, 1280, 5120), (2048, 2560, 1280), (1280, 2048, 1280), (8192, 640, 640), (8192, 1280, 640),
                        (4096, 1280, 1280), (4096, 1280, 5120)):
            for tile in (0, 1, 2, 3, 5, 6, 7, 8):
                try:
                    bench_linear(m, n, k, tile)
                except Exception as e:
                    print(json.dumps({"m": m, "n": n, "k": k, "tile": tile, "error": str(e)[:80]}), flush=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gemm":
        gemm_sweep()
        sys.exit(0)
    Bu = 16  # SDXL B=8 with CFG
    for tile in (1, 2, 3):
        bench_linear(Bu * 1024, 1280, 1280, tile)
    bench_linear(Bu * 1024, 10240, 1280, 0, act=1)
    bench_linear(Bu * 1024, 1280, 5120)
    bench_linear(Bu * 4096, 640, 640)
    bench_linear(Bu * 4096, 5120, 640, 0, act=1)
    bench_linear(Bu * 4096, 640, 2560)
    bench_linear(8192, 8192, 8192, 1)
    for tile in (1, 2):
        bench_conv(Bu, 128, 128, 320, 320, tile)
    bench_conv(Bu, 64, 64, 640, 640)
    bench_conv(Bu, 32, 32, 1280, 1280)
    bench_conv(Bu, 32, 32, 2560, 1280)
    bench_conv(Bu, 32, 32, 1280, 1280, up=(64, 64))
    bench_conv(Bu, 128, 128, 320, 320, stride=2)
    bench_attn(Bu, 10, 4096, 4096, 64, 64)
    bench_attn(Bu, 20, 1024, 1024, 64, 64)
    bench_attn(Bu, 20, 1024, 77, 64, 64)
    bench_attn(8, 8, 4096, 4096, 40, 48)
    bench_attn(8, 8, 1024, 1024, 80, 80)
    bench_attn(8, 8, 256, 256, 160, 160)
    bench_gn(Bu, 128, 128, 320)
    bench_gn(Bu, 64, 64, 640)
    bench_gn(Bu, 32, 32, 1280)
    bench_ln(Bu * 4096, 640)
    bench_ln(Bu * 1024, 1280)
