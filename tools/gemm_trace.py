"""Per-block cycle stamps of the ring GEMM (debug build with -DPA_GEMM_TRACE3, see csrc/gemm.hip):
    PLANK_HIP_LIB=tools/ubench/libplank_trace.so python tools/gemm_trace.py
mean cycles per block: kernel entry -> setup done -> first K tile landed -> K loop done -> epilogue done (stores retired)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from plankassembly_amd import ops, _lib as L


def items(K):
    fn = getattr(L.lib(), 'pa_gemm3_items_read', None)
    if fn is None:
        return
    buf = np.zeros(64, dtype=np.uint64)
    fn.restype = C.c_int
    n = fn(buf.ctypes.data_as(C.c_void_p))
    if n <= 0:
        return
    st = [int(x) & ((1 << 63) - 1) for x in buf[:n + 1]]
    hot = ['H' if int(x) >> 63 else 'c' for x in buf[:n]]
    print('    block 0 items (H = steady-state item, c = item at a unit boundary): ' + ' '.join(f'{hot[i]}{st[i + 1] - st[i]}' for i in range(n)), flush=True)


def run(M, N, K, res=True, drop=0.0, alias_a=False):
    """alias_a: every row of A is the SAME 1 KB row (row stride 0) - the activation operand then lives in L2 like the weight,
    which separates the HBM latency of the A panel from everything else in the K loop."""
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    if alias_a:
        x = x[:1].expand(M, K)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    r = torch.randn(M, N, device="cuda").to(torch.bfloat16) if res else None
    for _ in range(3):
        ops.gemm(x, w, bias=b, residual=r, drop_p=drop, drop_seed=1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(x, w, bias=b, residual=r, drop_p=drop, drop_seed=1)
    e1.record(); torch.cuda.synchronize()
    buf = np.zeros(512 * 8, dtype=np.uint64)
    fn = L.lib().pa_gemm3_trace_read
    fn.restype = C.c_int
    assert fn(buf.ctypes.data_as(C.c_void_p), C.c_int32(buf.size)) == 0
    t = buf.reshape(-1, 8).astype(np.int64)
    tb = 64 if os.environ.get("SMALL") == "1" else 128
    nb = min(512, (M + tb - 1) // tb * ((N + tb - 1) // tb))
    t = t[:nb]
    t = t[t[:, 4] > 0]
    d = [t[:, i + 1] - t[:, i] for i in range(4)]
    spread = t[:, 0].max() - t[:, 0].min()
    span = t[:, 4].max() - t[:, 0].min()
    print(f"M {M:5d} N {N:5d} K {K:5d} res {int(res)} drop {drop}{' A aliased (L2-resident)' if alias_a else ''}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us/launch | blocks {len(t)}  setup {d[0].mean():6.0f}  "
          f"first tile {d[1].mean():6.0f}  K loop {d[2].mean():7.0f} ({d[2].mean() / (K // 64):5.0f}/tile)  epilogue {d[3].mean():6.0f} (stores issued after {(t[:, 5] - t[:, 3]).mean():6.0f})  "
          f"block total {(t[:, 4] - t[:, 0]).mean():7.0f}  first entry -> last exit {span}  entry spread {spread}", flush=True)


if os.environ.get("SMALL") == "1":
    for K in (512, 1024, 1536):
        run(2048, 512, K)
    run(2048, 1536, 512, res=False)
    run(2048, 1024, 512, res=False, drop=0.2)
    run(256, 512, 512)
    sys.exit(0)
if os.environ.get("ALIAS") == "1":
    for K in (512, 1536):
        run(7940, 512, K, res=False)
        items(K)
        run(7940, 512, K, res=False, alias_a=True)
        items(K)
    sys.exit(0)
for K in (512, 1024, 1536):
    run(7940, 512, K)
    items(K)
run(7940, 512, 512, res=False)
run(7940, 512, 1024, drop=0.2)
run(7940, 1024, 512, res=False)
run(2048, 2048, 4096, res=False)
