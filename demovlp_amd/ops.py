"""Thin tensor-level wrappers over the C ABI (include/demovlp_hip.h).  No autograd here; see functional.py.

Every function takes contiguous CUDA(HIP) tensors, launches on torch's current stream and returns torch tensors
allocated with the caching allocator (PyTorch is plumbing for device memory and streams only).
"""
from __future__ import annotations

import contextlib
import ctypes

import torch

from . import _lib
from ._lib import BF16, EPI_ACCUM, EPI_GELU, EPI_GELU_BWD, EPI_LEAKY, EPI_RELU_BWD, F32, call

EPI_OUT_F32 = 32
_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dt(t_or_dtype) -> int:
    d = t_or_dtype.dtype if isinstance(t_or_dtype, torch.Tensor) else t_or_dtype
    try:
        return _DT[d]
    except KeyError:
        raise _lib.DemoVLPHipError(f"unsupported compute dtype {d}") from None


def p(t):
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.DemoVLPHipError("demovlp_amd kernels need tensors on an MI355X device (no CPU fallback)")


# ----------------------------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------------------------
class GemmExt(ctypes.Structure):
    """include/demovlp_hip.h: dvlp_gemm_ext"""
    _fields_ = [("colsum", ctypes.c_void_p), ("colsum_fused", ctypes.c_int)]


class AttnExt(ctypes.Structure):
    """include/demovlp_hip.h: dvlp_attn_ext"""
    _fields_ = [("keep", ctypes.c_void_p), ("keepT", ctypes.c_void_p), ("keep_scale", ctypes.c_float), ("colsum", ctypes.c_void_p),
                ("colsum_fused", ctypes.c_int), ("folded", ctypes.c_int)]


def gemm(a, b, M, N, K, *, trans_a=False, trans_b=False, lda=None, ldb=None, out=None, ldc=None, bias=None, res=None,
         aux=None, flags=0, alpha=1.0, out_f32=False, dtype=None, ext=None):
    """C[M,N] = epi(alpha * op(A) op(B)^T).  a/b/out/res/aux are tensors whose data_ptr is the matrix origin.  ``ext``: a GemmExt."""
    d = dt(a) if dtype is None else dtype
    if d == BF16:
        ensure_gemm_workspace(a.device)
    lda = lda if lda is not None else (M if trans_a else K)
    ldb = ldb if ldb is not None else (N if trans_b else K)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    ldc = ldc if ldc is not None else N
    if out_f32:
        flags |= EPI_OUT_F32
    if ext is not None:
        call("dvlp_gemm_ex", d, int(trans_a), int(trans_b), M, N, K, p(a), lda, p(b), ldb, p(out), ldc, p(bias), p(res),
             N if res is not None else 0, p(aux), N if aux is not None else 0, flags, float(alpha), ctypes.byref(ext), stream())
        return out
    call("dvlp_gemm", d, int(trans_a), int(trans_b), M, N, K, p(a), lda, p(b), ldb, p(out), ldc, p(bias), p(res),
         N if res is not None else 0, p(aux), N if aux is not None else 0, flags, float(alpha), stream())
    return out


_GEMM_WS = {}

# Scratch buffers are cached per stream (two streams must not share scratch).  A hipGraph capture runs on torch's capture stream, and anything
# allocated while capturing comes out of THAT graph's private memory pool: a scratch buffer cached under the capture stream's id would be
# handed to the next capture -- of another shape, of another step object -- long after the pool it lives in was released.  So the thread
# that captures declares which eager stream its capture stream stands for (the replays run there): scratch is looked up under that
# stream's id, where the eager warm-up steps allocated it from the ordinary pool, and growing it during a capture is refused.
_CAPTURE_ALIAS = {}          # capture stream id -> id of the eager stream it stands for


class capture_on_behalf_of:
    """``with capture_on_behalf_of(eager_stream): with torch.cuda.graph(...): ...`` -- see above."""

    def __init__(self, eager_stream):
        self.eager = eager_stream.cuda_stream
        self.cap = None

    def note(self):
        """Call inside the capture context: binds the current (capture) stream to the eager one."""
        self.cap = torch.cuda.current_stream().cuda_stream
        if self.cap != self.eager:
            _CAPTURE_ALIAS[self.cap] = self.eager
        return self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if self.cap is not None:
            _CAPTURE_ALIAS.pop(self.cap, None)
        return False


@contextlib.contextmanager
def graph_capture(graph, pool=None):
    """``torch.cuda.graph`` for code that calls into this library: ``thread_local`` error mode (a loader thread may keep staging, copying and
    waiting on events while this thread captures) and the cached scratch of the CURRENT stream -- the one the replays will run on."""
    with capture_on_behalf_of(torch.cuda.current_stream()) as alias:
        with torch.cuda.graph(graph, capture_error_mode="thread_local", **({"pool": pool} if pool is not None else {})):
            alias.note()
            yield graph


def _scratch_stream_id():
    """(id the scratch caches are keyed by, id of the stream the kernels are launched on)"""
    sid = torch.cuda.current_stream().cuda_stream
    return _CAPTURE_ALIAS.get(sid, sid), sid


def ensure_gemm_workspace(device, mbytes=160):
    """Register (once per device and stream) the scratch the split-K GEMM uses for its fp32 partial slabs."""
    key_sid, sid = _scratch_stream_id()
    key = (str(device), key_sid)
    if key not in _GEMM_WS:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.DemoVLPHipError("GEMM scratch would be allocated inside a hipGraph capture: run the step eagerly once first")
        t = torch.empty(mbytes * 1024 * 1024, device=device, dtype=torch.uint8)
        _GEMM_WS[key] = t
        call("dvlp_set_workspace_stream", ctypes.c_void_p(key_sid), p(t), t.numel())
    t = _GEMM_WS[key]
    if sid != key_sid and _GEMM_WS.get(("alias", str(device), sid)) is not t:
        call("dvlp_set_workspace_stream", ctypes.c_void_p(sid), p(t), t.numel())     # the capture stream's launches find the same buffer
        _GEMM_WS[("alias", str(device), sid)] = t
    return t


@_lib.on_library_switch
def _drop_library_registrations():
    """Another build of the library is about to become the active one (tests / tools: _lib.use_dev_library): whatever this module
    registered with the current one -- split-K scratch per stream, colsum counters, the deferred-reduction queue -- is flushed and dropped;
    it is registered anew, lazily, with whichever library the next call goes to."""
    if _DEFER:
        torch.cuda.synchronize()
        disable_deferred_reductions()
    for key in list(_GEMM_WS):
        if key[0] != "alias":
            call("dvlp_set_workspace_stream", ctypes.c_void_p(key[1]), None, 0)
    _GEMM_WS.clear()
    _COLSUM_CNT.clear()
    _HINTED.clear()


_SIDE = {}


def side_stream(device=None):
    """Secondary HIP stream for work that is off the backward critical path (weight / bias gradients)."""
    device = torch.cuda.current_device() if device is None else device
    if device not in _SIDE:
        _SIDE[device] = torch.cuda.Stream(device=device)
    return _SIDE[device]


_TEXT = {}


_HINTED = set()


def text_stream(device=None):
    """HIP stream the text tower runs on when the two towers run concurrently (model.ObjectRelation.parallel_towers).  Registered with the
    library as CO-RUNNING (dvlp_stream_hint): its GEMMs share the chip with the object tower's, so the dispatch picks tiles for CU-time per
    FLOP rather than for the latency of their own grid."""
    device = torch.cuda.current_device() if device is None else device
    if device not in _TEXT:
        _TEXT[device] = torch.cuda.Stream(device=device)
    if device not in _HINTED:
        call("dvlp_stream_hint", ctypes.c_void_p(_TEXT[device].cuda_stream), 1)
        _HINTED.add(device)
    return _TEXT[device]


def linear_fwd(x2d, w, bias=None, res=None, gelu_aux=None):
    """y = x W^T (+ bias) (+ res); with gelu_aux given: aux <- pre-activation, y <- gelu(pre)."""
    M, K = x2d.shape
    N = w.shape[0]
    return gemm(x2d, w, M, N, K, bias=bias, res=res, aux=gelu_aux, flags=EPI_GELU if gelu_aux is not None else 0)


def linear_bwd_input(dy2d, w, *, res=None, gelu_pre=None, relu_pre=None, out=None, accumulate=False, colsum_to=None):
    """dx = dy W  (optionally * gelu'(pre) or masked by pre > 0, + res, or accumulated onto ``out``).
    ``colsum_to``: fp32 [K] destination for the column sums of dx (the bias gradient of the Linear that produced this GEMM's
    input side); fused into the GEMM epilogue on the deferred path -- final after :func:`flush_reductions` then."""
    M, N = dy2d.shape
    K = w.shape[1]
    flags, aux = 0, None
    if gelu_pre is not None:
        flags, aux = EPI_GELU_BWD, gelu_pre
    elif relu_pre is not None:
        flags, aux = EPI_RELU_BWD, relu_pre
    if accumulate:
        flags |= EPI_ACCUM
    # A = dy [M x N] (k = N contiguous), B(kin, n) = W[n][kin] -> form R with ld = K
    ext = None
    if colsum_to is not None:
        assert colsum_to.dtype == torch.float32 and colsum_to.numel() == K and not accumulate
        ensure_gemm_workspace(dy2d.device)
        ext = GemmExt(colsum=colsum_to.data_ptr())
    return gemm(dy2d, w, M, K, N, trans_b=True, ldb=K, bias=None, res=res, aux=aux, flags=flags, out=out, ext=ext)


def linear_bwd_weight(dy2d, x2d, out=None):
    """dW [N,K] = dy^T x, written as fp32 (directly usable as the master-parameter gradient)."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    if out is None:
        out = torch.empty((N, K), device=dy2d.device, dtype=torch.float32)
    # A(n, m) = dy[m][n] -> form R ld = N ; B(k, m) = x[m][k] -> form R ld = K ; reduction over m
    return gemm(dy2d, x2d, N, K, M, trans_a=True, trans_b=True, lda=N, ldb=K, out=out, out_f32=True)


class WgradExt(ctypes.Structure):
    """include/demovlp_hip.h: dvlp_wgrad_ext"""
    _fields_ = [("n", ctypes.c_int64), ("p", ctypes.c_void_p), ("g", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("hyper", ctypes.c_void_p), ("bf16_shadow", ctypes.c_void_p), ("fused", ctypes.c_int)]


def wgrad_grouped(problems, ride=None):
    """[(dy2d [T, N_p], x2d [T, K_p], out fp32 [N_p, K_p] or None)] -> list of dW_p = dy_p^T x_p (fp32).  One grouped launch for
    bf16 operands that suit the 256 x 256 GEMM kernel (a transformer layer's weight gradients), else per-problem GEMMs.
    ``ride``: (p, g, m, v, hyper, shadow | None, lo, hi) -- one fused-AdamW range update (adamw_range_dev's arguments) to be executed by the
    workgroups the grouped launch leaves idle (dvlp_wgrad_grouped_ex); launched on its own right behind when there is no room."""
    n = len(problems)
    outs = []
    I64, VP = ctypes.c_int64 * n, ctypes.c_void_p * n
    Ms, Ns, Ks, lda, ldb = I64(), I64(), I64(), I64(), I64()
    A, Bp, C = VP(), VP(), VP()
    d = dt(problems[0][0])
    if d == BF16:
        ensure_gemm_workspace(problems[0][0].device)
    for i, (dy2d, x2d, out) in enumerate(problems):
        T, N = dy2d.shape
        K = x2d.shape[1]
        assert x2d.shape[0] == T and dt(dy2d) == d and dt(x2d) == d and dy2d.stride(1) == 1 and x2d.stride(1) == 1
        if out is None:
            out = torch.empty((N, K), device=dy2d.device, dtype=torch.float32)
        assert out.is_contiguous() and out.dtype == torch.float32
        outs.append(out)
        Ms[i], Ns[i], Ks[i], lda[i], ldb[i] = N, K, T, dy2d.stride(0), x2d.stride(0)
        A[i], Bp[i], C[i] = dy2d.data_ptr(), x2d.data_ptr(), out.data_ptr()
    if ride is not None:
        fp, fg, fm, fv, hyper, shadow, lo, hi = ride
        ext = WgradExt(n=hi - lo, p=fp.data_ptr() + 4 * lo, g=fg.data_ptr() + 4 * lo, m=fm.data_ptr() + 4 * lo, v=fv.data_ptr() + 4 * lo,
                       hyper=hyper.data_ptr(), bf16_shadow=(shadow.data_ptr() + 2 * lo) if shadow is not None else None)
        call("dvlp_wgrad_grouped_ex", d, n, ctypes.addressof(Ms), ctypes.addressof(Ns), ctypes.addressof(Ks), ctypes.addressof(A), ctypes.addressof(lda),
             ctypes.addressof(Bp), ctypes.addressof(ldb), ctypes.addressof(C), 0, ctypes.byref(ext), stream())
        return outs
    call("dvlp_wgrad_grouped", d, n, ctypes.addressof(Ms), ctypes.addressof(Ns), ctypes.addressof(Ks), ctypes.addressof(A), ctypes.addressof(lda),
         ctypes.addressof(Bp), ctypes.addressof(ldb), ctypes.addressof(C), 0, stream())
    return outs


_WS = {}


def _workspace(key, nfloat, device):
    """Reusable fp32 scratch, one per (purpose, device, stream): use is ordered by the stream it belongs to."""
    k = (key, device, _scratch_stream_id()[0])
    t = _WS.get(k)
    if t is None or t.numel() < nfloat:
        if torch.cuda.is_current_stream_capturing():
            raise _lib.DemoVLPHipError(f"scratch '{key}' would be (re)allocated inside a hipGraph capture: run the step eagerly at this shape first")
        if t is not None:
            _WS_RETIRED.append(t)       # a captured graph may still point at the smaller buffer: it must never be handed to anyone else
        t = torch.empty(int(nfloat), device=device, dtype=torch.float32)
        _WS[k] = t
    return t


_WS_RETIRED = []


_COLSUM_CNT = {}


FUSED_COLSUM = False      # measured on MI355X: the agent-scope release per workgroup (L2 write-back) costs more than the
                          # second launch it saves (step 39.5 -> 43.4 ms), so the two-launch reduction stays the default


def _ensure_colsum_counters(device):
    if not FUSED_COLSUM:
        return
    key = str(device)
    if key not in _COLSUM_CNT:
        t = torch.zeros(4096, device=device, dtype=torch.int32)
        _COLSUM_CNT[key] = t
        call("dvlp_colsum_counters", p(t), t.numel())


_DEFER = {}


def enable_deferred_reductions(device, workspace_mb=1024, max_items=4096):
    """Let gradient column sums that are written straight into a gradient arena (``defer=True`` below) postpone their final
    reduction to ONE batched launch at :func:`flush_reductions`.  The trainer enables this and flushes before the
    optimizer step; without it every such call reduces immediately.  (A B = 64 step queues ~390 MB of partial rows -- 37 LayerNorm
    backward launches of 1024 workgroups x 2-3 planes x 768 floats are most of it; a call that finds the workspace full reduces on the
    spot, which costs a launch of its own.)"""
    key = str(device)
    if key not in _DEFER:
        ws = torch.empty(int(workspace_mb) << 18, device=device, dtype=torch.float32)
        table = torch.empty(max_items * 48, device=device, dtype=torch.uint8)
        _DEFER[key] = (ws, table)
        call("dvlp_reduce_defer", p(ws), ws.numel() * 4, p(table), table.numel())


def disable_deferred_reductions():
    flush_reductions()
    call("dvlp_reduce_defer", None, 0, None, 0)
    _DEFER.clear()


def flush_reductions():
    if _DEFER:
        call("dvlp_reduce_flush", stream())


def colsum(x2d, out=None, accumulate=False, defer=False):
    """fp32 [N] = sum over rows of x [M,N] (bias gradients).  ``defer``: the result is not read before flush_reductions()."""
    M, N = x2d.shape
    if out is None:
        out = torch.empty(N, device=x2d.device, dtype=torch.float32)
    _ensure_colsum_counters(x2d.device)
    ws = _workspace("colsum", call("dvlp_colsum_chunks", M) * N, x2d.device)
    call("dvlp_colsum", dt(x2d), M, N, p(x2d), x2d.stride(0), M, 0, 1, 0, p(out), p(ws), int(accumulate) | (2 if defer else 0), stream())
    return out


def colsum_grouped(x, M, N, ld, inner, ostride, groups, gstride, out=None):
    if out is None:
        out = torch.empty((groups, N), device=x.device, dtype=torch.float32)
    _ensure_colsum_counters(x.device)
    ws = _workspace("colsum", groups * call("dvlp_colsum_chunks", M) * N, x.device)
    call("dvlp_colsum", dt(x), M, N, p(x), ld, inner, ostride, groups, gstride, p(out), p(ws), 0, stream())
    return out


# ----------------------------------------------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------------------------------------------
# TIMING-ONLY ablation (DVLP_ABLATE_LN_FWD=1, tools/ln_fusion_bound.sh): the forward LayerNorm launches nothing and hands its INPUT on as its
# output -- what the step would take if every forward LayerNorm pass were free, i.e. an upper bound on what any fusion of the normalisation
# into a neighbouring product could buy in the step.  Results are wrong by construction; bench.py marks the line.
LN_FWD_ABLATE = bool(int(__import__("os").environ.get("DVLP_ABLATE_LN_FWD", "0")))
_LN_ABL = {}


def layernorm_fwd(x2d, gamma, beta, eps, want_relu=False):
    M, D = x2d.shape
    if LN_FWD_ABLATE and not want_relu:
        key = (M, str(x2d.device))
        if key not in _LN_ABL:                      # constant statistics (mean 0, rstd 1), allocated once: no launch per call
            _LN_ABL[key] = (torch.zeros(M, device=x2d.device, dtype=torch.float32), torch.ones(M, device=x2d.device, dtype=torch.float32))
        return x2d, None, _LN_ABL[key][0], _LN_ABL[key][1]
    y = torch.empty_like(x2d)
    yr = torch.empty_like(x2d) if want_relu else None
    mean = torch.empty(M, device=x2d.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    call("dvlp_layernorm_fwd", dt(x2d), M, D, p(x2d), p(gamma), p(beta), float(eps), p(y), p(yr), p(mean), p(rstd), stream())
    return y, yr, mean, rstd


def layernorm_bwd(dy2d, x2d, gamma, mean, rstd, dres=None, out_gamma=None, out_beta=None, defer=False, dx_colsum=None):
    """``out_gamma`` / ``out_beta``: optional fp32 destinations (e.g. gradient-arena slices); contiguous pairs reduce in one launch.
    ``defer``: dgamma / dbeta are not read before flush_reductions()."""
    M, D = x2d.shape
    dx = torch.empty_like(x2d)
    if out_gamma is not None and out_beta is not None:
        dgamma, dbeta = out_gamma, out_beta
    else:
        gb = torch.empty(2 * D, device=x2d.device, dtype=torch.float32)      # contiguous pair -> one reduction launch
        dgamma, dbeta = gb[:D], gb[D:]
    ws = _workspace("ln", (call("dvlp_layernorm_bwd_blocks", M) + 1) * 2 * D, x2d.device)
    call("dvlp_layernorm_bwd", dt(x2d), M, D, p(dy2d), p(x2d), p(gamma), p(mean), p(rstd), p(dres), p(dx), p(dgamma), p(dbeta),
         p(ws), 2 if defer else 0, p(dx_colsum), stream())
    return dx, dgamma, dbeta


# ----------------------------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------------------------
HEADS, HEAD_DIM = 12, 64
SCALE = HEAD_DIM ** -0.5


def space_attention_fwd(qkv, addmask, B, F, R, want_stats=False):
    """qkv [B*N, 2304] packed (q | k | v), addmask [B,N] fp32 -> [B*N, 768].  ``want_stats`` (bf16): also the CLS query's softmax
    statistics [B, H, 4] fp32 that :func:`space_attention_bwd` takes back with the output (the CLS query is then folded into the
    per-frame waves in both directions); None in their place where the fold does not apply."""
    N = 1 + F * R
    out = torch.empty((B * N, 768), device=qkv.device, dtype=qkv.dtype)
    es = qkv.element_size()
    base = qkv.data_ptr()
    stats, ws = None, None
    if want_stats and qkv.dtype == torch.bfloat16 and (R + 15) // 16 == (R + 16) // 16 and R + 1 <= 48:
        stats = torch.empty((B, HEADS, 4), device=qkv.device, dtype=torch.float32)
        ws = _workspace("attn_fwd", B * HEADS * F * 66, qkv.device)
    ext = AttnExt()
    call("dvlp_attention_fwd_ex", dt(qkv), 0, B, N, HEADS, F, R, ctypes.c_void_p(base), ctypes.c_void_p(base + 768 * es),
         ctypes.c_void_p(base + 1536 * es), 2304, p(addmask), p(out), 768, SCALE, p(ws), p(stats), ctypes.byref(ext), stream())
    if stats is not None and not ext.folded:
        stats = None                                  # the fold was switched off (dvlp_dev_attention_cls_fold): nothing was written
    return (out, stats) if want_stats else out


def space_attention_bwd(qkv, addmask, dout, B, F, R, out=None, stats=None, colsum_to=None):
    """``out`` / ``stats``: the forward's output and the statistics it returned (``want_stats``), or None.  ``colsum_to``: fp32 [2304]
    destination of the column sums of dqkv (the packed qkv bias gradient), queued from inside the kernels when possible (final after
    :func:`flush_reductions`); returns (dqkv, fused) then -- ``fused`` False means the caller still has to sum the columns."""
    N = 1 + F * R
    dqkv = torch.empty_like(qkv)
    es = qkv.element_size()
    b, db = qkv.data_ptr(), dqkv.data_ptr()
    ws = _workspace("attn", B * HEADS * (F * 3 * 64 + 4), qkv.device)
    if stats is None:
        out = None
    ext = AttnExt(colsum=colsum_to.data_ptr() if colsum_to is not None else None)
    call("dvlp_attention_bwd_ex", dt(qkv), 0, B, N, HEADS, F, R, ctypes.c_void_p(b), ctypes.c_void_p(b + 768 * es),
         ctypes.c_void_p(b + 1536 * es), 2304, p(addmask), p(dout), 768, ctypes.c_void_p(db), ctypes.c_void_p(db + 768 * es),
         ctypes.c_void_p(db + 1536 * es), 2304, p(ws), SCALE, p(out), 768, p(stats), ctypes.byref(ext), stream())
    if colsum_to is not None:
        return dqkv, bool(ext.colsum_fused)
    return dqkv


def full_attention_fwd(q, k, v, addmask, B, L, ld=768, keep=None):
    """``ld``: row stride of q / k / v (2304 when they are the three column blocks of one packed projection).
    ``keep``: (keep bytes [B*H, Ns, Ns], transposed keep bytes, 1 / (1 - p)) from :func:`attn_keep_masks` -- dropout of the
    attention probabilities."""
    out = torch.empty((B * L, 768), device=q.device, dtype=q.dtype)
    ext = AttnExt(keep=keep[0].data_ptr(), keepT=keep[1].data_ptr(), keep_scale=float(keep[2])) if keep is not None else AttnExt()
    call("dvlp_attention_fwd_ex", dt(q), 1, B, L, HEADS, 1, L, p(q), p(k), p(v), ld, p(addmask), p(out), 768, SCALE, None, None, ctypes.byref(ext), stream())
    return out


def full_attention_bwd(q, k, v, addmask, dout, B, L, ld=768, out=None, ld_out=768, keep=None):
    ext = AttnExt(keep=keep[0].data_ptr(), keepT=keep[1].data_ptr(), keep_scale=float(keep[2])) if keep is not None else AttnExt()
    if out is None:
        dq, dk, dv = (torch.empty((B * L, 768), device=q.device, dtype=q.dtype) for _ in range(3))
    else:
        dq, dk, dv = out
    call("dvlp_attention_bwd_ex", dt(q), 1, B, L, HEADS, 1, L, p(q), p(k), p(v), ld, p(addmask), p(dout), 768, p(dq), p(dk), p(dv),
         ld_out, None, SCALE, None, 0, None, ctypes.byref(ext), stream())
    return dq, dk, dv


# ----------------------------------------------------------------------------------------------------------------
# dropout (text tower, train mode)
# ----------------------------------------------------------------------------------------------------------------
def dropout_state(device, seed=0):
    """Device-resident Philox state {seed_lo, seed_hi, offset, -} (int32 storage of uint32 words)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    words = [seed & 0xFFFFFFFF, seed >> 32, 0, 0]
    return torch.tensor([w - (1 << 32) if w >= (1 << 31) else w for w in words], dtype=torch.int32, device=device)


def dropout_advance(state):
    call("dvlp_dropout_advance", p(state), stream())


def dropout_fwd(x, prob, state, site, res=None):
    """y = x * keep / (1 - p) (+ res).  Returns (y, keep bytes)."""
    y = torch.empty_like(x)
    keep = torch.empty(x.numel(), device=x.device, dtype=torch.uint8)
    call("dvlp_dropout_fwd", dt(x), x.numel(), p(x), p(res), p(y), p(keep), float(prob), p(state), int(site), stream())
    return y, keep


def dropout_bwd(dy, keep, prob):
    dx = torch.empty_like(dy)
    call("dvlp_dropout_bwd", dt(dy), dy.numel(), p(dy), p(keep), float(prob), p(dx), stream())
    return dx


def attn_keep_masks(B, L, prob, state, site, device):
    """Keep bytes of the [L, L] attention-probability maps of every (batch, head), in both orientations + the rescale factor."""
    Ns = (L + 15) // 16 * 16
    keep = torch.empty((B * HEADS, Ns, Ns), device=device, dtype=torch.uint8)
    keepT = torch.empty_like(keep)
    call("dvlp_dropout_attn_mask", B * HEADS, L, float(prob), p(state), int(site), p(keep), p(keepT), stream())
    return keep, keepT, 1.0 / (1.0 - prob)


# ----------------------------------------------------------------------------------------------------------------
# prologues
# ----------------------------------------------------------------------------------------------------------------
def obj_split(obj, dtype):
    """obj [B,F,R,2054] fp32 -> feat [M,2048] (compute dtype), box [M,6] fp32."""
    M = obj.numel() // 2054
    feat = torch.empty((M, 2048), device=obj.device, dtype=dtype)
    box = torch.empty((M, 6), device=obj.device, dtype=torch.float32)
    call("dvlp_obj_split", dt(dtype), M, p(obj), p(feat), p(box), stream())
    return feat, box


def embed_assemble(tok, box, Wp, bp, temporal, cls, pos0, mask01, B, F, R):
    N = 1 + F * R
    x = torch.empty((B * N, 768), device=tok.device, dtype=tok.dtype)
    addmask = torch.empty((B, N), device=tok.device, dtype=torch.float32)
    call("dvlp_embed_assemble", dt(tok), B, F, R, p(tok), p(box), p(Wp), p(bp), p(temporal), p(cls), p(pos0), p(mask01), p(x),
         p(addmask), stream())
    return x, addmask


def split_cls(x):
    """x [B,N,d] (contiguous) -> (x[:, 0] as [B,d], x[:, 1:] as [B,N-1,d]), both contiguous, one launch."""
    B, N, d = x.shape
    g = torch.empty((B, d), device=x.device, dtype=x.dtype)
    l = torch.empty((B, N - 1, d), device=x.device, dtype=x.dtype)
    call("dvlp_split_cls", B, N, d * x.element_size(), p(x), p(g), p(l), stream())
    return g, l


def merge_cls(dg, dl, B, N, d, dtype, device):
    """Backward of :func:`split_cls`: dx [B,N,d] from dg [B,d] / dl [B,N-1,d] (None = zeros), one launch."""
    dx = torch.empty((B, N, d), device=device, dtype=dtype)
    call("dvlp_merge_cls", B, N, d * dx.element_size(), p(dg), p(dl), p(dx), stream())
    return dx


def embed_unassemble(dx, B, F, R):
    dtok = torch.empty((B * F * R, 768), device=dx.device, dtype=dx.dtype)
    call("dvlp_embed_unassemble", dt(dx), B, F, R, p(dx), p(dtok), stream())
    return dtok


def token_transpose(src, B, F, R, res=None):
    """[B, 1 + F*R, D] frame-major tokens -> region-major ('b (f n) d -> b (n f) d', CLS in place), + ``res`` (destination
    order) when given.  With F = 1 this is the identity, i.e. a plain ``src + res``."""
    dst = torch.empty_like(src)
    D = src.shape[-1]
    call("dvlp_token_transpose", dt(src), B, F, R, D, p(src), p(res), p(dst), stream())
    return dst


def box_wgrad(dtok, box, out=None):
    M = dtok.shape[0]
    if out is None:
        out = torch.empty((768, 6), device=dtok.device, dtype=torch.float32)
    ws = _workspace("boxw", call("dvlp_box_wgrad_chunks", M) * 6 * 768, dtok.device)
    call("dvlp_box_wgrad", dt(dtok), M, p(dtok), p(box), p(out), p(ws), 0, stream())
    return out


def text_embed_fwd(ids, word, pos, gamma, beta, eps, dtype):
    B, L = ids.shape
    e = torch.empty((B * L, 768), device=ids.device, dtype=dtype)
    y = torch.empty_like(e)
    mean = torch.empty(B * L, device=ids.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    call("dvlp_text_embed_fwd", dt(dtype), B, L, p(ids), p(word), p(pos), p(gamma), p(beta), float(eps), p(e), p(y), p(mean),
         p(rstd), stream())
    return e, y, mean, rstd


def text_embed_bwd(ids, de, vocab):
    dword = torch.zeros((vocab, 768), device=de.device, dtype=torch.float32)
    call("dvlp_text_embed_bwd", dt(de), de.shape[0], p(ids), p(de), p(dword), stream())
    return dword


def text_key_mask(att):
    """int64 attention mask [B, L] -> DistilBERT's additive key mask, fp32 [B, L]: 0 for real tokens, -inf for padding (one launch)."""
    B, L = att.shape
    mask = torch.empty((B, L), device=att.device, dtype=torch.float32)
    call("dvlp_text_key_mask", B, L, p(att), p(mask), stream())
    return mask


def text_mask_len(att):
    """(text_length int64 [B], text_mask fp32 [B, L - 1]) = (att.sum(1), (att[:, 1:] - 1.0) * 100.0) in one launch (att: int64 [B, L] on the device)."""
    B, L = att.shape
    att = att.contiguous()
    length = torch.empty(B, device=att.device, dtype=torch.int64)
    mask = torch.empty((B, L - 1), device=att.device, dtype=torch.float32)
    call("dvlp_text_mask_len", B, L, p(att), p(length), p(mask), stream())
    return length, mask


def copy_by_kernel(dst, src):
    """dst <- src through a kernel launch rather than ``copy_`` (= hipMemcpyAsync for contiguous same-dtype tensors).  Inside a captured
    step a copy / memset becomes a memcpy / memset NODE, and those were not reliably ordered against the kernels around them when a replay
    started on an idle device (a hipMemsetAsync in the local-loss backward gave garbage gradients that way): the captured step holds kernel
    nodes only."""
    if (dst.dtype in _DT and src.dtype in _DT and not (dst.dtype == src.dtype == torch.bfloat16) and dst.is_contiguous() and src.is_contiguous()
            and dst.numel() == src.numel() and dst.numel() > 0 and dst.is_cuda):
        call("dvlp_cast", dt(src), dt(dst), src.numel(), p(src), p(dst), stream())
    else:
        dst.copy_(src)
    return dst


def cast(src, dtype, out=None):
    if out is None:
        out = torch.empty(src.shape, device=src.device, dtype=dtype)
    call("dvlp_cast", dt(src), dt(dtype), src.numel(), p(src), p(out), stream())
    return out


# ----------------------------------------------------------------------------------------------------------------
# losses
# ----------------------------------------------------------------------------------------------------------------
XATTN_ONE_STREAM = False      # per-call option of dvlp_xattn_fwd / _bwd (gate bit 1): keep the local loss' two halves on the caller's stream
                              # (bench.py sets it around its per-launch timing pass; the default lets the library fork its side stream)


def _gate_bits(gate):
    return (1 if gate else 0) | (2 if XATTN_ONE_STREAM else 0)


def xattn_fwd(C, Q, m_img, m_cap, lam, gate, need_bwd):
    """C [Bi,G,256], Q [Bj,W,256] (compute dtype), additive fp32 masks -> scores fp32 [Bi,Bj], workspace."""
    Bi, G, d = C.shape
    Bj, W, _ = Q.shape
    nbytes = call("dvlp_xattn_workspace_bytes", dt(C), Bi, Bj, G, W, int(need_bwd))
    ws = torch.empty(int(nbytes), device=C.device, dtype=torch.uint8)
    scores = torch.empty((Bi, Bj), device=C.device, dtype=torch.float32)
    call("dvlp_xattn_fwd", dt(C), Bi, Bj, G, W, d, p(C), p(Q), p(m_img), p(m_cap), float(lam), _gate_bits(gate), p(scores), p(ws),
         int(need_bwd), stream())
    return scores, ws


def xattn_bwd(C, Q, m_img, m_cap, lam, gate, dscores, ws):
    Bi, G, d = C.shape
    Bj, W, _ = Q.shape
    dC, dQ = torch.empty_like(C), torch.empty_like(Q)
    call("dvlp_xattn_bwd", dt(C), Bi, Bj, G, W, d, p(C), p(Q), p(m_img), p(m_cap), float(lam), _gate_bits(gate), p(dscores), p(ws),
         p(dC), p(dQ), stream())
    return dC, dQ


def global_local_loss(gt, go, xs, temperature, lam, use_global, use_local, stages, sim=None, dsim=None):
    """Raw call of the fused loss-head kernel.  Returns dict with whichever of sim/dsim/dgt/dgo/dxs/losses were produced."""
    ref = gt if gt is not None else (xs if xs is not None else sim)
    B = ref.shape[0]
    dev = ref.device
    if sim is None:
        sim = torch.empty((B, B), device=dev, dtype=torch.float32)
    if dsim is None:
        dsim = torch.empty((B, B), device=dev, dtype=torch.float32)
    dgt = torch.empty_like(gt) if (gt is not None and stages & 4) else None
    dgo = torch.empty_like(go) if (go is not None and stages & 4) else None
    dxs = torch.empty_like(xs) if (xs is not None and stages & 2) else None
    losses = torch.empty(3, device=dev, dtype=torch.float32) if stages & 2 else None      # (all three are written by stage 2, nothing else touches them)
    d = dt(gt) if gt is not None else F32
    call("dvlp_global_local_loss", d, B, 256, p(gt), p(go), p(xs), float(temperature), float(lam), int(use_global), int(use_local),
         int(stages), p(sim), p(dsim), p(dgt), p(dgo), p(dxs), p(losses), stream())
    return dict(sim=sim, dsim=dsim, dgt=dgt, dgo=dgo, dxs=dxs, losses=losses)


def rownorm_fwd(x):
    """x [M,256] -> (x / max(|x|, 1e-8) as fp32 [M,256], |x| fp32 [M])."""
    M = x.shape[0]
    xn = torch.empty((M, 256), device=x.device, dtype=torch.float32)
    norm = torch.empty(M, device=x.device, dtype=torch.float32)
    call("dvlp_rownorm_fwd", dt(x), M, 256, p(x), p(xn), p(norm), stream())
    return xn, norm


def rownorm_bwd(x, norm, dxn):
    dx = torch.empty_like(x)
    call("dvlp_rownorm_bwd", dt(x), x.shape[0], 256, p(x), p(norm), p(dxn.contiguous()), p(dx), stream())
    return dx


# ----------------------------------------------------------------------------------------------------------------
# region select, optimizer
# ----------------------------------------------------------------------------------------------------------------
def region_select(feats, bbox, conf, wh, R, nvalid=None, out=None):
    """feats [B,F,Nraw,2048], bbox [B,F,Nraw,4], conf [B,F,Nraw], wh [B,F,2] (fp32, on device) ->
    obj [B,F,R,2054] fp32, mask [B,F,R] fp32, order [B,F,R] int32, lens [B,F] int32.  ``out`` = (obj, mask): write there."""
    B, F, Nraw, _ = feats.shape
    dev = feats.device
    if out is not None:
        obj, mask = out
        for t, shp in ((obj, (B, F, R, 2054)), (mask, (B, F, R))):
            if tuple(t.shape) != shp or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise ValueError(f"region_select: out tensor must be contiguous fp32 {shp} on {dev}")
    else:
        obj = torch.empty((B, F, R, 2054), device=dev, dtype=torch.float32)
        mask = torch.empty((B, F, R), device=dev, dtype=torch.float32)
    order = torch.empty((B, F, R), device=dev, dtype=torch.int32)
    lens = torch.empty((B, F), device=dev, dtype=torch.int32)
    call("dvlp_region_select", B * F, F, Nraw, R, p(feats), p(bbox), p(conf), p(wh), p(nvalid), p(obj), p(mask), p(order), p(lens),
         stream())
    return obj, mask, order, lens


def adamw_step(pflat, gflat, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, shadow=None):
    call("dvlp_adamw_step", pflat.numel(), p(pflat), p(gflat), p(m), p(v), float(lr), float(beta1), float(beta2), float(eps),
         float(weight_decay), int(step), float(grad_scale), p(shadow), stream())


def adamw_step_dev(pflat, gflat, m, v, hyper, shadow=None):
    """hyper: fp32 [8] device tensor {lr, b1, b2, eps, wd, grad_scale, step, step_size}; the step counter advances on the device."""
    call("dvlp_adamw_step_dev", pflat.numel(), p(pflat), p(gflat), p(m), p(v), p(hyper), p(shadow), stream())


def adamw_prep_dev(hyper):
    """Advance the device step counter / step size once (first half of adamw_step_dev)."""
    call("dvlp_adamw_prep_dev", p(hyper), stream())


def adamw_range_dev(pflat, gflat, m, v, hyper, shadow, lo, hi):
    """Second half of adamw_step_dev on elements [lo, hi) of the flat buffers (lo a multiple of 4)."""
    call("dvlp_adamw_range_dev", hi - lo, p(pflat[lo:hi]), p(gflat[lo:hi]), p(m[lo:hi]), p(v[lo:hi]), p(hyper),
         p(shadow[lo:hi]) if shadow is not None else None, stream())


def prof_enable(on: bool):
    call("dvlp_prof_enable", int(on))


def prof_collect():
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    call("dvlp_prof_collect", ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n))
    return ms.value, fl.value, n.value
