import ctypes, sys, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from demovlp_amd import ops
old = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "norm_old.so"))
M, D = 18496, 768
g = torch.Generator(device="cuda").manual_seed(0)
dy = torch.randn(M, D, device="cuda", generator=g).bfloat16(); x = torch.randn(M, D, device="cuda", generator=g).bfloat16()
gamma = torch.ones(D, device="cuda"); mean = torch.zeros(M, device="cuda"); rstd = torch.ones(M, device="cuda")
dres = torch.randn(M, D, device="cuda", generator=g).bfloat16()
dx = torch.empty_like(x); gb = torch.empty(2 * D, device="cuda"); ws = torch.empty(1025 * 2 * D + 1024 * D, device="cuda")
def p(t): return ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
def run_old():
    old.dvlp_layernorm_bwd(1, ctypes.c_int64(M), ctypes.c_int64(D), p(dy), p(x), p(gamma), p(mean), p(rstd), p(dres), p(dx), p(gb), ctypes.c_void_p(gb.data_ptr() + 4 * D), p(ws), 0, st)
def run_new():
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dres=dres, out_gamma=gb[:D], out_beta=gb[D:])
def bench(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("old (incl. reduce launch): %.1f us   new: %.1f us" % (bench(run_old), bench(run_new)))
