"""Thin, allocation-explicit Python wrappers over the C ABI (include/gctplus_hip.h).

Every function takes/returns fp32 CUDA tensors, enqueues on the current torch stream and
never synchronises.  No autograd here: `engine.py` composes these into hand-written
forward/backward passes.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import math
import os
import threading
import weakref
from typing import List, Optional, Sequence

import torch

from . import _lib
from ._lib import check

EPI_BIAS, EPI_GELU_DROP, EPI_DROP_RESID = 0, 1, 2
DEPI_STORE, DEPI_ACCUM, DEPI_GELU_BWD = 0, 1, 2


def _L():
    return _lib.load()


_raw_stream, _cur_dev = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def _st():
    # the raw handle of the current stream: torch.cuda.current_stream() builds a Stream object through several
    # Python layers (8.7 us per call, a third of the host time of a training step)
    return _raw_stream(_cur_dev())


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.GctError(f"{name}: expected a CUDA (ROCm) tensor; gct_plus_amd has no CPU path")
    if t.dtype != dtype:
        raise _lib.GctError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


# ---------------------------------------------------------------------------- side stream
# Weight-gradient GEMMs are off the critical path of a backward pass (nothing downstream needs
# dW until the optimiser / all-reduce), so they are launched on a SIDE stream: their workgroups
# fill the partial last round of the critical-path dgrad GEMMs and the gaps of the small
# bandwidth kernels.  Correctness bookkeeping (all stream-ordered, no host syncs):
#   * the side stream waits for an event recorded on the main stream at launch (inputs ready);
#   * inputs are record_stream()'ed so the caching allocator cannot recycle them early;
#   * every buffer a pending wgrad READS is remembered (by storage pointer) with the wgrad's
#     completion event; any main-stream op that WRITES into an existing buffer waits on it first
#     (the in-place residual-gradient buffer `g` of engine.py is the case that matters);
#   * join_side() makes the main stream wait for everything (called at the end of each trunk's
#     backward, before autograd / all-reduce / Adam can touch the gradients).
# Measured on MI355X (bench.py, B=512): 112.5 ms/step without vs 112.7 ms with the side stream (round 1); with the
# bf16x6 kernels 49.2 without, 50.3 with, 50.7-50.9 with the side stream at low and the main stream at high priority
# (round 3: one 144 KB workgroup fills a CU, so a second stream only interleaves whole workgroups and splits the L2)
# -- no gain, so it is OFF by default (GCT_SIDE_STREAM=1 enables it; tests cover both settings).
SIDE_ENABLED = os.environ.get("GCT_SIDE_STREAM", "0") != "0"
_SIDE = {}
_PENDING = {}


def _side_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _SIDE.get(idx)
    if st is None:
        st = torch.cuda.Stream(device=idx)
        _SIDE[idx] = st
    return st


def _wait_pending(t):
    """Main stream is about to WRITE into tensor t: wait for side-stream readers of its storage."""
    if _PENDING and t is not None:
        ev = _PENDING.pop(t.untyped_storage().data_ptr(), None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)


def join_side(device=None):
    if _SIDE:
        cur = torch.cuda.current_stream()
        for st in _SIDE.values():
            cur.wait_stream(st)
        _PENDING.clear()


# ------------------------------------------------------------------------------ workspace
_WS = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Per (device, stream) scratch buffer, grown geometrically; kernels on one stream are
    ordered, so consecutive ops may reuse it."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), _st())
    buf = _WS.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        n = max(int(nbytes * 1.25) // 4 + 64, 1 << 20)
        buf = torch.empty(n, dtype=torch.float32, device=device)
        _WS[key] = buf
    return buf


# Deferred slab reductions (csrc/reduce.hip): a layer's backward pass records the reductions that end its weight-gradient
# GEMMs and Norm backward kernels and runs them as ONE launch at the end of the layer (engine.enc_layer_bwd /
# dec_layer_bwd): 106 launches of 5-10 us per step become 12.  While recording, the workspaces of those calls must stay
# intact until the flush: kept_workspace() hands out consecutive regions of a second buffer instead of the shared one.
DEFER_REDUCTIONS = os.environ.get("GCT_DEFER_REDUCTIONS", "1") != "0"
class _DeferState(threading.local):        # per thread, like the library's record of pending reductions
    def __init__(self):
        self.d = {"on": False, "off": 0, "demand": 0, "want": 0}


_DEFER_TL = _DeferState()
_ARENA = {}


def kept_workspace(nbytes: int, device) -> torch.Tensor:
    """Workspace of a call whose slab reduction may be deferred: the shared scratch buffer when nothing is being
    recorded, otherwise a fresh region of the arena (the recorded reductions are flushed first when it is full)."""
    if not _DEFER_TL.d["on"]:
        return workspace(nbytes, device)
    key = (device.index if device.index is not None else torch.cuda.current_device(), _st())
    n = (int(nbytes) + 255) // 256 * 64            # floats, 256-byte regions
    _DEFER_TL.d["demand"] += n
    buf = _ARENA.get(key)
    if buf is None or _DEFER_TL.d["off"] + n > buf.numel():
        # full (or absent): what was recorded so far is reduced now, on this stream, before anything overwrites it
        check(_L().gct_reduce_defer_flush(_st()), "gct_reduce_defer_flush")
        _DEFER_TL.d["off"] = 0
        if buf is None or n > buf.numel():
            buf = _ARENA[key] = torch.empty(max(n, _DEFER_TL.d["want"], 1 << 22), dtype=torch.float32, device=device)
    out = buf[_DEFER_TL.d["off"]:_DEFER_TL.d["off"] + n]
    _DEFER_TL.d["off"] += n
    return out


class deferred_reductions:
    """with deferred_reductions(): ... -- the slab reductions issued inside by THIS thread run as one launch at exit
    (bit-identical results: same lanes and summation order per destination).  Not while a graph is being captured, not
    with the side stream (its weight gradients are ordered by events, not by this stream), not nested."""

    def __enter__(self):
        self.mine = (DEFER_REDUCTIONS and not _DEFER_TL.d["on"] and not SIDE_ENABLED
                     and not torch.cuda.is_current_stream_capturing())
        if self.mine:
            key = (torch.cuda.current_device(), _st())
            buf = _ARENA.get(key)
            if buf is not None and buf.numel() < _DEFER_TL.d["want"]:
                del _ARENA[key]                    # grown on first use below: a whole layer fits from the second layer on
            _DEFER_TL.d.update(on=True, off=0, demand=0)
            check(_L().gct_reduce_defer_begin(), "gct_reduce_defer_begin")
        return self

    def __exit__(self, et, ev, tb):
        if self.mine:
            _DEFER_TL.d["on"] = False
            _DEFER_TL.d["want"] = max(_DEFER_TL.d["want"], _DEFER_TL.d["demand"])
            rc = _L().gct_reduce_defer_end(_st())
            if et is None:
                check(rc, "gct_reduce_defer_end")
        return False


_WS_NEED = {}


def _ws_need(fn: str, a: int, b: int, c: int) -> int:
    """Workspace size queries of the library (pure functions of the shape), remembered per shape."""
    key = (fn, a, b, c)
    v = _WS_NEED.get(key)
    if v is None:
        v = _WS_NEED[key] = int(getattr(_L(), fn)(a, b, c))
    return v


# ----------------------------------------------------------------------------------- norm
def norm_fwd(x2d, alpha, bias, eps=1e-6, out=None):
    _chk(x2d, "norm_fwd.x")
    rows, d = x2d.shape
    y = torch.empty_like(x2d) if out is None else out
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    check(_L().gct_norm_fwd(_p(x2d), _p(alpha), _p(bias), _p(y), _p(mean), _p(rstd), rows, d, eps,
                            _st()), "gct_norm_fwd")
    return y, mean, rstd


def norm_bwd(dy, x2d, alpha, mean, rstd, dalpha, dbias, dres=None, out=None, eps=1e-6, live=None, drop=None):
    """live (LiveRows): dy / dres / out are quad-compacted rows [live.Mc, d]; x2d, mean, rstd stay in the forward's
    row space -- or, when the forward itself ran on the compact rows (live.fwd), are compact too.
    drop = (buffer, p, seed, site): also write dropout_bwd(result) with that mask into buffer (the
    gradient's next consumer is the sub-layer whose output dropout used (seed, site))."""
    src_rows, d = x2d.shape
    rows = src_rows if live is None else live.Mc
    if live is not None and live.fwd:
        src_rows = 0                                   # x / mean / rstd are compact: the map only places the dropout bits
    dx = (torch.empty_like(x2d) if live is None else live.empty(d)) if out is None else out
    _wait_pending(out)
    if drop is not None:
        _wait_pending(drop[0])
    ws = kept_workspace(_L().gct_rowred_ws_bytes(rows, 2 * d), x2d.device)
    check(_L().gct_norm_bwd(_p(dy), _p(x2d), _p(alpha), _p(mean), _p(rstd), _p(dres), _p(dx),
                            _p(dalpha), _p(dbias), _p(ws), rows, d, eps,
                            None if live is None else _p(live.quad_list), src_rows,
                            None if drop is None else _p(drop[0]), 0.0 if drop is None else drop[1],
                            0 if drop is None else drop[2], 0 if drop is None else drop[3], _st()), "gct_norm_bwd")
    return dx


# ------------------------------------------------------------------------------ embedding
def embed_pe_fwd(tok, table, cond, pe2d, n_c, scale, p, seed, site, d=None):
    _chk(tok, "embed.tok", torch.int64)
    B, S = tok.shape
    if table is not None:
        vocab, d = table.shape
    else:
        vocab = 1
    out = torch.empty(B * (S + n_c), d, dtype=torch.float32, device=pe2d.device)
    check(_L().gct_embed_pe_fwd(_p(tok), _p(table), _p(cond), _p(pe2d), _p(out), B, S, n_c, d, vocab,
                                scale, p, seed, site, _st()), "gct_embed_pe_fwd")
    return out


def embed_pe_bwd(dout, tok, dtable, dcond, n_c, scale, p, seed, site, d=None):
    B, S = tok.shape
    if dtable is not None:
        vocab, d = dtable.shape
    else:
        vocab = 1
    ws = workspace(_L().gct_embed_ws_bytes(B, S, d, vocab), dout.device)
    check(_L().gct_embed_pe_bwd(_p(dout), _p(tok), _p(dtable), _p(dcond), _p(ws), B, S, n_c, d, vocab,
                                scale, p, seed, site, _st()), "gct_embed_pe_bwd")


# --------------------------------------------------------------------------------- linear
# Optional per-launch timing of the GEMM family (bench.py roofline leg): when PROFILE is a
# dict, every gct_linear_* launch is bracketed by HIP events on the launch stream and logged as
# (kernel kind, flops, start event, end event).  Off (None) in normal operation.
PROFILE = None
PROFILE_KINDS = None      # None: every GEMM kind; else a set such as {"gemm_fwd"}


class _Timed:
    def __init__(self, kind, flops, nbytes=0):
        self.kind, self.flops, self.nbytes = kind, flops, nbytes

    def __enter__(self):
        self.on = PROFILE is not None and (PROFILE_KINDS is None or self.kind in PROFILE_KINDS)
        if self.on:
            self.c0 = gemm_launch_counts()[1]
            self.k0 = _L().gct_gemm_x6_kernel_launches()
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            # launches that took the bf16x6 kernels are logged under their own key
            kind = self.kind + "[x6]" if gemm_launch_counts()[1] > self.c0 else self.kind
            PROFILE.setdefault(kind, []).append((self.flops, self.e0, self.e1))
            PROFILE["_x6_kernel_launches:" + kind] = PROFILE.get("_x6_kernel_launches:" + kind, 0) + \
                (_L().gct_gemm_x6_kernel_launches() - self.k0)
            # algorithmic HBM bytes of the calls (every operand read once, every result written once)
            PROFILE["_bytes:" + kind] = PROFILE.get("_bytes:" + kind, 0) + self.nbytes


def _seg3(ts: Sequence[Optional[torch.Tensor]]):
    ts = list(ts) + [None] * (3 - len(ts))
    return [_p(t) for t in ts]


# ---- bf16x6 GEMM mode: pre-split weight planes --------------------------------------------------
GEMM_F32, GEMM_BF16X6 = 0, 1
_PLANES = []     # [(base_ptr, end_ptr, planes tensor (int16 [3, numel]), numel)]


def gemm_set_mode(mode: int):
    check(_L().gct_gemm_set_mode(int(mode)), "gct_gemm_set_mode")


def gemm_get_mode() -> int:
    return _L().gct_gemm_get_mode()


def gemm_launch_counts():
    """(launches of the fp32-MFMA tile kernels, launches of the bf16x6 kernels) since load."""
    import ctypes
    out = (ctypes.c_int64 * 2)()
    check(_L().gct_gemm_launch_counts(ctypes.cast(out, ctypes.c_void_p)), "gct_gemm_launch_counts")
    return int(out[0]), int(out[1])


def split_planes(flat: torch.Tensor, planes: Optional[torch.Tensor] = None) -> torch.Tensor:
    """flat fp32 [numel] -> int16 [3, numel] holding the hi / mid / lo bf16 pieces of every element
    (exact: hi + mid + lo == x).  numel % 4 == 0."""
    _chk(flat, "flat")
    n = flat.numel()
    if planes is None:
        planes = torch.empty(3, n, dtype=torch.int16, device=flat.device)
    check(_L().gct_split_planes(_p(flat), n, _p(planes), planes.stride(0), _st()), "gct_split_planes")
    return planes


def _drop_planes(base: int, end: int):
    _PLANES[:] = [e for e in _PLANES if e[1] <= base or e[0] >= end]


def register_planes(flat: torch.Tensor, planes: torch.Tensor):
    """Weights that live inside `flat` are looked up in `planes` by linear_fwd / linear_dgrad.
    The registration dies with the `flat` tensor object (its address range may be handed to other
    tensors by the caching allocator afterwards) and replaces anything it overlaps."""
    base, end = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    known = any(e[0] == base and e[1] == end and e[4]() is flat for e in _PLANES)
    _drop_planes(base, end)
    _PLANES.append((base, end, planes, flat.numel(), weakref.ref(flat)))
    if not known:
        weakref.finalize(flat, _drop_planes, base, end)


def unregister_planes(flat: torch.Tensor):
    _drop_planes(flat.data_ptr(), flat.data_ptr() + flat.numel() * 4)


def _plane_ptr(ws_):
    """(pointer to plane 0 of ws_[0], plane stride) if every segment lies in one registered buffer."""
    if not _PLANES:
        return None, 0
    a = ws_[0].data_ptr()
    for base, end, planes, numel, ref in _PLANES:
        if base <= a < end:
            if ref() is None:
                return None, 0
            if all(base <= w.data_ptr() < end for w in ws_[1:]):
                return planes.data_ptr() + (a - base) // 2, planes.stride(0)
            return None, 0
    return None, 0


def linear_fwd(x2d, ws_: Sequence[torch.Tensor], bs: Sequence[Optional[torch.Tensor]],
               outs: Sequence[torch.Tensor], ldy: int, epi=EPI_BIAS, resid=None, pre=None,
               p=0.0, seed=0, site=0, splitk_ws: Optional[torch.Tensor] = None, ws: Optional[torch.Tensor] = None,
               live=None):
    """y_s = epi(x @ w_s^T + b_s); outs are (views of) pre-allocated [M, nper] blocks with
    leading dimension ldy.  splitk_ws: force the skinny-M split-K kernel with this workspace; ws: a caller-owned
    workspace (>= gct_linear_fwd_ws_bytes) for the general path (fixed address: graph capture); live (LiveRows): the
    rows are quad-compacted -- the dropout masks of the fused epilogues are drawn at the original coordinates."""
    M, K = x2d.shape
    nper = ws_[0].shape[0]
    w = _seg3(ws_)
    b = _seg3(bs)
    y = _seg3(outs)
    need = _ws_need("gct_linear_fwd_ws_bytes", M, K, nper * len(ws_))     # skinny split-K slabs or bf16x6 tail slabs
    for t in (splitk_ws, ws):
        if t is not None and t.numel() * 4 < need:
            raise _lib.GctError(f"linear_fwd: workspace of {t.numel() * 4} B < {need} B "
                                f"(gct_linear_fwd_ws_bytes({M}, {K}, {nper * len(ws_)}))")
    if splitk_ws is not None:      # skinny-M path (decode): split-K through the workspace
        check(_L().gct_linear_fwd_ws(_p(x2d), x2d.stride(0), M, K, w[0], w[1], w[2], ws_[0].stride(0),
                                     b[0], b[1], b[2], len(ws_), nper, y[0], y[1], y[2], ldy, epi,
                                     _p(resid), _p(pre), p, seed, site, _p(splitk_ws), splitk_ws.numel() * 4, _st()),
              "gct_linear_fwd_ws")
        return
    wp, pstride = _plane_ptr(ws_)
    wsb = ws if ws is not None else (workspace(need, x2d.device) if need > 256 else None)
    N = nper * len(ws_)
    nbytes = 4 * M * K + (6 if wp else 4) * N * K + 4 * N + 4 * M * N * (2 if epi in (EPI_GELU_DROP, EPI_DROP_RESID) else 1)
    with _Timed("gemm_fwd", 2.0 * M * K * N, nbytes):
        check(_L().gct_linear_fwd_p(_p(x2d), x2d.stride(0), M, K, w[0], w[1], w[2], ws_[0].stride(0),
                                    wp, pstride, b[0], b[1], b[2], len(ws_), nper, y[0], y[1], y[2],
                                    ldy, epi, _p(resid), _p(pre), p, seed, site, _p(wsb),
                                    0 if wsb is None else wsb.numel() * 4,
                                    None if live is None else _p(live.quad_list), _st()),
              "gct_linear_fwd_p")


def linear_dgrad(dys: Sequence[torch.Tensor], lddy: int, M: int, ws_: Sequence[torch.Tensor], dx,
                 depi=DEPI_STORE, pre=None, p=0.0, seed=0, site=0, live=None, pre_full=False):
    nper, K = ws_[0].shape
    d = _seg3(dys)
    w = _seg3(ws_)
    _wait_pending(dx)
    with _Timed("gemm_dgrad", 2.0 * M * K * nper * len(ws_)):
        wp, pstride = _plane_ptr(ws_)
        need = _ws_need("gct_linear_dgrad_ws_bytes", M, nper * len(ws_), K)
        wsb = workspace(need, dx.device) if need > 256 else None
        check(_L().gct_linear_dgrad_p(d[0], d[1], d[2], lddy, M, len(ws_), nper, w[0], w[1], w[2],
                                      ws_[0].stride(0), wp, pstride, K, _p(dx), dx.stride(0), depi,
                                      _p(pre), p, seed, site, _p(wsb), 0 if wsb is None else wsb.numel() * 4,
                                      None if live is None else _p(live.quad_list),
                                      pre.shape[0] if (live is not None and pre is not None and pre_full) else 0,
                                      _st()), "gct_linear_dgrad_p")


def nonzero_row_tiles(x2d: torch.Tensor):
    """(list, count): ascending indices of the 32-row tiles of x2d that hold a non-zero element, and their
    number as a device scalar (no host round trip).  Feed to linear_wgrad(kt=...) for GEMMs whose dY rows are
    zero wherever x2d's are (the decoder backward under an ignore_index loss)."""
    _chk(x2d, "x2d")
    rows, cols = x2d.shape
    nt = (rows + 31) // 32
    lst = torch.empty(max(nt, 1), dtype=torch.int32, device=x2d.device)
    cnt = torch.empty(1, dtype=torch.int32, device=x2d.device)
    flags = torch.empty(max(nt, 1), dtype=torch.uint8, device=x2d.device)
    check(_L().gct_nonzero_row_tiles(_p(x2d), x2d.stride(0), rows, cols, _p(lst), _p(cnt), _p(flags), _st()),
          "gct_nonzero_row_tiles")
    return lst, cnt


_SKIPPED = {}
_SKIPPED_MSG = ("{} decoder rows that an earlier forward skipped (loss_rows: rows that do not reach the loss) received a "
                "non-zero gradient in its backward pass: that gradient was ignored, and the optimizer steps launched since "
                "then were skipped on the device (FusedAdam's guard), so no wrong update has been applied; this error is "
                "raised at the first host read-back after the fact.  Call the model without loss_rows "
                "(forward_propagation(..., skip_ignored=False)) for losses other than the reference's ignore_index "
                "cross-entropy.")


def assert_no_skipped_row_gradients():
    """Read the counter now (one synchronisation) and raise if it is non-zero: the trainer calls this at the end of an
    epoch so that a gradient on a skipped row in the LAST step of a run is reported too."""
    n = int(skipped_row_gradients().item())
    if n:
        skipped_row_gradients().zero_()
        raise _lib.GctError(_SKIPPED_MSG.format(n))


def skipped_row_gradients():
    """Device counter (int32[1], one per device) of gradient rows that fell on decoder rows a forward had skipped."""
    dev = torch.cuda.current_device()
    t = _SKIPPED.get(dev)
    if t is None:
        t = _SKIPPED[dev] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", dev))
    return t


class LiveRows:
    """Device-side description of the rows of a decoder gradient that are not identically zero, the check that makes
    shortcuts on them exact, and the compaction map of the decoder backward (csrc/liverows.hip).
      kt          feeds linear_wgrad(kt=...): lists every tile when the check failed -- no host decision needed;
      host()      reads the counters back (ONE device->host sync);
      quad_list / cstart / n_b / Mc  (after host()): rows are compacted in aligned quads, compact row 4i+e <->
                  original row 4*quad_list[i]+e, Mc compact rows (multiple of 128)."""
    SLACK = 256        # rows behind Mc in every compact buffer: attention stages whole L-row windows of a sample
    fwd = False        # True (instance): the FORWARD ran on these compact rows too (engine.decoder_trunk_fwd)

    def __init__(self, g2d, B, T, mask_u8, lists=True):
        _chk(g2d, "g2d")
        dev = g2d.device
        M = B * T
        assert g2d.shape[0] == M
        nt, nq = (M + 31) // 32, (M + 3) // 4
        i32 = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)            # noqa: E731
        self.B, self.T, self.M, self.dev = B, T, M, dev
        self.live = torch.empty(max(M, 1), dtype=torch.uint8, device=dev)
        self.n_b, self.info = i32(B), i32(8)
        self.cstart = i32(B) if lists else None
        self.quad_list = i32(nq + 32) if lists else None
        qrank = i32(nq) if lists else None
        self.tile_list, self.tile_count = i32(nt), i32(1)
        flags = torch.empty(max(nt, 1), dtype=torch.uint8, device=dev)
        if mask_u8 is None:
            sb = sq = 0
        else:
            _chk(mask_u8, "live_rows.mask", torch.uint8)
            sb, sq = _mask_strides(mask_u8, B, T, T)
        check(_L().gct_live_rows(_p(g2d), g2d.stride(0), B, T, g2d.shape[1], _p(mask_u8), sb, sq, _p(self.live),
                                 _p(self.n_b), _p(self.info), _p(self.cstart), _p(self.quad_list), _p(qrank),
                                 _p(self.tile_list), _p(self.tile_count), _p(flags), _st()), "gct_live_rows")
        self.kt = (self.tile_list, self.tile_count)
        self._host = None
        self.Mc = None

    @classmethod
    def from_rows(cls, rows_u8, B, T, mask_u8):
        """The same maps from a given row set (rows_u8 [B, T], non-zero = live) instead of from a gradient: the decoder
        forward over the rows that reach the loss."""
        _chk(rows_u8, "live_rows.rows", torch.uint8)
        return cls(rows_u8.reshape(B * T, 1).to(torch.float32), B, T, mask_u8)

    def _fill_host(self, v):
        self._host = dict(n_live=v[0], violations=v[1], nonprefix=v[2], tiles=v[3], padded=v[4], quads=v[5], empty=v[6])
        self.Mc = v[4]

    def host(self):
        if self._host is None:
            read_back(self)                                 # ONE read-back (also of the skipped-row gradient counter)
        return self._host

    def scatter_add(self, src2d, dst2d):
        """dst2d [M, cols] rows += the compact rows src2d [Mc, cols]."""
        cols = src2d.shape[1]
        check(_L().gct_scatter_add_quads(_p(src2d), src2d.stride(0), _p(self.quad_list), self.Mc, cols, _p(dst2d),
                                         dst2d.stride(0), self.M, _st()), "gct_scatter_add_quads")
        return dst2d

    def check_grad(self, g2d):
        """The backward of a forward that ran on these rows only: count the gradient rows outside them that are not
        zero (device counter, read at the next host() of any LiveRows -- no synchronisation here)."""
        check(_L().gct_dead_rows_nonzero(_p(g2d), g2d.stride(0), self.M, g2d.shape[1], _p(self.live),
                                         _p(skipped_row_gradients()), _st()), "gct_dead_rows_nonzero")

    def empty(self, cols):
        """A compact [Mc, cols] activation (with SLACK rows of allocation behind it)."""
        return torch.empty(self.Mc + self.SLACK, cols, dtype=torch.float32, device=self.dev)[:self.Mc]

    def empty_zero_gaps(self, cols):
        """A compact [Mc, cols] buffer for a kernel that writes the rows cstart[b] .. + n_b[b] of every sample only
        (attention over compact rows): uninitialised, except that the rows in between -- padded rows that travel with a
        live quad, the padding quads at the end -- are zero, so that the GEMMs that consume all Mc rows see finite
        values there (a few rows per sample instead of a fill of the whole buffer)."""
        t = self.empty(cols)
        check(_L().gct_zero_gap_rows(_p(t), t.stride(0), cols, _p(self.cstart), _p(self.n_b), self.B, self.Mc, _st()),
              "gct_zero_gap_rows")
        return t

    def gather(self, src2d, out=None):
        """src2d [M, cols] (row stride >= cols) -> compact [Mc, cols]; padding rows are zero."""
        cols = src2d.shape[1]
        dst = self.empty(cols) if out is None else out
        check(_L().gct_gather_quads(_p(src2d), src2d.stride(0), src2d.shape[0], _p(self.quad_list), self.Mc, cols,
                                    _p(dst), dst.stride(0), _st()), "gct_gather_quads")
        return dst

    def scatter(self, src2d, out=None):
        """compact [Mc, cols] -> [M, cols], zero rows where nothing is live."""
        cols = src2d.shape[1]
        dst = torch.zeros(self.M, cols, dtype=torch.float32, device=self.dev) if out is None else out
        check(_L().gct_scatter_quads(_p(src2d), src2d.stride(0), _p(self.quad_list), self.Mc, cols, _p(dst),
                                     dst.stride(0), self.M, _st()), "gct_scatter_quads")
        return dst


HOST_BLOCKED_S = [0.0]      # host seconds spent waiting in read_back (the one synchronisation of a training step)


def read_back(*objs):
    """host() of several LiveRows / KeyRows with ONE device->host copy (one synchronisation instead of one each)."""
    objs = [o for o in objs if o is not None and o._host is None]
    if not objs:
        return
    import time
    t0 = time.perf_counter()
    v = torch.cat([o.info for o in objs] + [skipped_row_gradients()]).tolist()
    HOST_BLOCKED_S[0] += time.perf_counter() - t0
    for i, o in enumerate(objs):
        o._fill_host(v[8 * i:8 * i + 8])
    if v[-1] != 0:
        skipped_row_gradients().zero_()
        raise _lib.GctError(_SKIPPED_MSG.format(v[-1]))


class PendingReadBack:
    """read_back() in two halves: the constructor queues ONE asynchronous device->host copy of the maps' info records
    (and of the skipped-row gradient counter) behind the kernels that wrote them and records an event; finish() waits
    for that event only -- not for whatever was queued on the stream afterwards -- and fills the host side in.  The
    trainer queues a batch's maps one step ahead (Model/forward_propagation1.prefetch), so finish() finds them done."""

    def __init__(self, *objs):
        seen, self.objs = set(), []
        for o in objs:
            if o is not None and o._host is None and id(o) not in seen:
                seen.add(id(o))
                self.objs.append(o)
        self.host = self.ev = None
        if self.objs:
            dev = torch.cat([o.info for o in self.objs] + [skipped_row_gradients()])
            self.host = torch.empty(dev.numel(), dtype=dev.dtype, pin_memory=True)
            self.host.copy_(dev, non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record()

    def finish(self):
        if self.host is None:
            return
        import time
        t0 = time.perf_counter()
        self.ev.synchronize()
        HOST_BLOCKED_S[0] += time.perf_counter() - t0
        v = self.host.tolist()
        self.host = None
        for i, o in enumerate(self.objs):
            if o._host is None:
                o._fill_host(v[8 * i:8 * i + 8])
        if v[-1] != 0:
            skipped_row_gradients().zero_()
            raise _lib.GctError(_SKIPPED_MSG.format(v[-1]))


class KeyRows(LiveRows):
    """The compaction map of the KEY side of cross-attention, from a key-padding mask [B, Lk] (gct_key_rows): the
    padded rows of the encoder memory are masked keys, so their K / V projections (and dK / dV) need not exist.
    Same interface as LiveRows (gather / scatter / empty / quad_list / cstart / n_b / Mc after host())."""

    def __init__(self, mask_u8, B, Lk):
        _chk(mask_u8, "key_rows.mask", torch.uint8)
        dev = mask_u8.device
        M = B * Lk
        nq = (M + 3) // 4
        i32 = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)            # noqa: E731
        self.B, self.T, self.M, self.dev = B, Lk, M, dev
        self.live = torch.empty(max(M, 1), dtype=torch.uint8, device=dev)
        self.n_b, self.info = i32(B), i32(8)
        self.cstart, self.quad_list = i32(B), i32(nq + 32)
        qrank = i32(nq)
        check(_L().gct_key_rows(_p(mask_u8), Lk, B, Lk, _p(self.live), _p(self.n_b), _p(self.info), _p(self.cstart),
                                _p(self.quad_list), _p(qrank), _st()), "gct_key_rows")
        self._host = None
        self.Mc = None

    # (host(): LiveRows' -- the same 8-word info record, read together with whatever else is pending)


def linear_wgrad(dys: Sequence[torch.Tensor], lddy: int, x2d, dws: Sequence[torch.Tensor],
                 dbs: Sequence[Optional[torch.Tensor]], kt=None):
    M, K = x2d.shape
    nper = dws[0].shape[0]
    nseg = len(dws)
    d = _seg3(dys)
    dw = _seg3(dws)
    db = _seg3(dbs)

    def launch():
        ws = kept_workspace(_ws_need("gct_wgrad_ws_bytes", M, nseg * nper, K), x2d.device)
        with _Timed("gemm_wgrad+bias+reduce", 2.0 * M * K * nper * nseg):
            check(_L().gct_linear_wgrad_kt(d[0], d[1], d[2], lddy, M, nseg, nper, _p(x2d), x2d.stride(0),
                                           K, dw[0], dw[1], dw[2], K, db[0], db[1], db[2], _p(ws),
                                           _p(kt[0]) if kt else None, _p(kt[1]) if kt else None, _st()),
                  "gct_linear_wgrad_kt")

    if not SIDE_ENABLED or torch.cuda.is_current_stream_capturing():
        launch()
        return
    main = torch.cuda.current_stream()
    side = _side_stream(x2d.device)
    ready = torch.cuda.Event()
    ready.record(main)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        launch()
        done = torch.cuda.Event()
        done.record(side)
    for t in list(dys) + [x2d]:
        t.record_stream(side)
        _PENDING[t.untyped_storage().data_ptr()] = done


def dropout_bwd(dout2d, p, seed, site, out=None, live=None):
    rows, cols = dout2d.shape
    if live is not None:
        rows = live.Mc
    dy = torch.empty_like(dout2d) if out is None else out
    _wait_pending(out)
    check(_L().gct_dropout_bwd(_p(dout2d), _p(dy), rows, cols, p, seed, site,
                               None if live is None else _p(live.quad_list), _st()), "gct_dropout_bwd")
    return dy


# ------------------------------------------------------------------------------ attention
class MaskBits:
    """An attention mask packed for the kernels (gct_attn_mask_pack): one bit per key, 8 words per query row.
    Built once per trunk call from the reference's bool / int64 mask and shared by every layer, head and the
    backward pass.  `u8` keeps the byte form (ops.LiveRows checks the decoder's zero-row shortcut against it)."""

    def __init__(self, mask_u8, B, Lq, Lk):
        _chk(mask_u8, "attn.mask", torch.uint8)
        if Lk > 256 or Lq > 256:
            raise _lib.GctError(f"attention: sequence length {max(Lq, Lk)} > 256 unsupported")
        sb, sq = _mask_strides(mask_u8, B, Lq, Lk)
        rows = 1 if sq == 0 else Lq
        self.u8, self.B, self.Lq, self.Lk = mask_u8, B, Lq, Lk
        self.bits = torch.empty(B, rows, 8, dtype=torch.int32, device=mask_u8.device)
        self.sb, self.sq = rows * 8, (0 if sq == 0 else 8)
        # visible key tiles per (batch, query tile): lets the kernels request K before the mask rows are back
        ntr = 1 if sq == 0 else (Lq + 15) // 16
        self.tiles = torch.empty(B, ntr, dtype=torch.int32, device=mask_u8.device)
        self.t_sb, self.t_su = ntr, (0 if sq == 0 else 1)
        check(_L().gct_attn_mask_pack(_p(mask_u8), sb, sq, B, Lq, Lk, _p(self.bits), _p(self.tiles), _st()),
              "gct_attn_mask_pack")


def pack_mask(mask, B, Lq, Lk) -> Optional["MaskBits"]:
    """None | MaskBits | uint8 / bool / int64 mask of [B,Lk], [B,1,Lk] or [B,Lq,Lk] elements -> MaskBits."""
    if mask is None or isinstance(mask, MaskBits):
        if isinstance(mask, MaskBits) and (mask.B, mask.Lk) != (B, Lk):
            raise _lib.GctError(f"packed mask was built for B={mask.B}, Lk={mask.Lk}, not B={B}, Lk={Lk}")
        return mask
    return MaskBits(to_mask_u8(mask), B, Lq, Lk)


def _mb(mask, B, Lq, Lk):
    """(pointer, batch stride, row stride, owner): the caller keeps `owner` referenced until its launch is enqueued --
    a mask packed on the fly lives only in that object, and a freed block may be handed to the next torch.empty."""
    mb = pack_mask(mask, B, Lq, Lk)
    if mb is None:
        return None, 0, 0, None
    if mb.t_su and mb.Lq != Lq:
        raise _lib.GctError(f"packed mask was built for Lq={mb.Lq}, not Lq={Lq}")
    return mb.bits.data_ptr(), mb.sb, mb.sq, mb


def _tb(mb):
    """(tile words, batch stride, query-tile stride) of a packed mask."""
    return (None, 0, 0) if mb is None else (mb.tiles.data_ptr(), mb.t_sb, mb.t_su)


def attn_fwd(q, k, v, ld_q, ld_k, ld_v, mask, B, H, Lq, Lk, dk, p, seed, site, out=None,
             want_probs=False, keys=None, live=None):
    """q/k/v: tensors whose data_ptr is element (b=0,l=0,h=0,0) with row strides ld_*.
    mask: None | MaskBits | uint8 [B,Lk] / [B,1,Lk] (key padding) / [B,Lq,Lk] (packed on the fly).
    live (LiveRows): q and the output hold the compact query rows only; keys (LiveRows / KeyRows): so do k / v."""
    dev = q.device
    if out is not None:
        o = out
    elif live is not None:     # rows of a live quad outside every sample's live prefix are not written: keep them finite
        o = live.empty_zero_gaps(H * dk)
    else:
        o = torch.empty(B * Lq, H * dk, dtype=torch.float32, device=dev)
    lse = torch.empty(B * H * Lq, dtype=torch.float32, device=dev)
    probs = torch.empty(B, H, Lq, Lk, dtype=torch.float32, device=dev) if want_probs else None
    mp, sb, sq, mask_owner = _mb(mask, B, Lq, Lk)
    check(_L().gct_attn_fwd(_p(q), ld_q, _p(k), ld_k, _p(v), ld_v, mp, sb, sq, _p(o),
                            o.stride(0), _p(lse), _p(probs), B, H, Lq, Lk, dk,
                            1.0 / math.sqrt(dk), p, seed, site, None if keys is None else _p(keys.cstart),
                            None if keys is None else _p(keys.n_b), *_tb(mask_owner),
                            None if live is None else _p(live.cstart), None if live is None else _p(live.n_b),
                            _st()), "gct_attn_fwd")
    del mask_owner
    return o, lse, probs


_ATTN_BWD_WS = {}


def _attn_bwd_ws(dev, nbytes):
    """Scratch of the two-launch attention backward: one buffer per (device, stream) like `workspace`, grown on demand
    (launches on one stream are ordered, so consecutive calls may share it; two streams never share one)."""
    if nbytes <= 0:
        return None
    key = (dev, _st())
    ws = _ATTN_BWD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _ATTN_BWD_WS[key] = ws
    return ws


def attn_bwd(q, k, v, ld_q, ld_k, ld_v, mask, o, dout, lse, dq, dk_, dv, ld_dq, ld_dk, ld_dv,
             B, H, Lq, Lk, dk, p, seed, site, live=None, kv_compact=False, keys=None):
    """live (LiveRows): dout / dq are quad-compacted; kv_compact: so are dk / dv (self-attention); keys (KeyRows):
    k / v / dk / dv hold the visible keys only (cross-attention).  live.fwd: the forward ran on the compact rows, so
    q and o are compact as well (and, for self-attention, k / v: pass keys=live, kv_compact=False)."""
    ws = _attn_bwd_ws(q.device, int(_L().gct_attn_bwd_ws_bytes(B, H, Lq, Lk)))
    mp, sb, sq, mask_owner = _mb(mask, B, Lq, Lk)
    check(_L().gct_attn_bwd(_p(q), ld_q, _p(k), ld_k, _p(v), ld_v, mp, sb, sq, _p(o),
                            _p(dout), o.stride(0), _p(lse), _p(dq), ld_dq, _p(dk_), ld_dk,
                            _p(dv), ld_dv, B, H, Lq, Lk, dk, 1.0 / math.sqrt(dk), p, seed, site,
                            None if live is None else _p(live.cstart), None if live is None else _p(live.n_b),
                            int(bool(kv_compact)) | (2 if (live is not None and live.fwd) else 0),
                            None if keys is None else _p(keys.cstart),
                            None if keys is None else _p(keys.n_b), *_tb(mask_owner), _p(ws),
                            0 if ws is None else ws.numel(),
                            _st()), "gct_attn_bwd")
    del mask_owner


def _mask_strides(mask_u8, B, Lq, Lk):
    if mask_u8 is None:
        return 0, 0
    _chk(mask_u8, "attn.mask", torch.uint8)
    n = mask_u8.numel()
    if n == B * Lk:
        return Lk, 0
    if n == B * Lq * Lk:
        return Lq * Lk, Lk
    raise _lib.GctError(f"attention mask with {n} elements fits neither [B,Lk] nor [B,Lq,Lk] "
                        f"(B={B}, Lq={Lq}, Lk={Lk})")


def trg_mask_u8(target: torch.Tensor, pad_id: int) -> torch.Tensor:
    """uint8 [B,T,T] form of Model.modules.get_trg_mask(target, pad_id, use_cond2dec=False) built from the token ids in
    one launch (reference Model/modules.py:47-58 builds an int64 [B,T,T] tensor: 27 MB at B = 512).  Nonzero = the
    query may attend, exactly where the reference's mask is nonzero (including its `* pad_idx` quirk)."""
    _chk(target, "trg_mask.target", torch.int64)
    B, T = target.shape
    out = torch.empty(B, T, T, dtype=torch.uint8, device=target.device)
    check(_L().gct_trg_mask_tokens(_p(target), target.stride(0), int(pad_id), B, T, _p(out), _st()),
          "gct_trg_mask_tokens")
    return out


def to_mask_u8(mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """Reference masks are bool [B,1,Lk] (Model/modules.py:38-44) or int64 [B,T,T]
    (modules.py:47-58); the kernels take uint8 (0 = masked)."""
    if mask is None:
        return None
    if mask.dtype == torch.uint8:
        return mask.contiguous()
    return (mask != 0).to(torch.uint8).contiguous()


# -------------------------------------------------------------------------------- VAE / loss
def reparam_fwd(mu, log_var, eps, seed, site):
    z = torch.empty_like(mu)
    eps_out = torch.empty_like(mu)
    check(_L().gct_reparam_fwd(_p(mu), _p(log_var), _p(eps), _p(eps_out), _p(z), mu.numel(), seed,
                               site, _st()), "gct_reparam_fwd")
    return z, eps_out


def reparam_bwd(dz, log_var, eps, dmu_ext, dlv_ext, dmu, dlv):
    check(_L().gct_reparam_bwd(_p(dz), _p(log_var), _p(eps), _p(dmu_ext), _p(dlv_ext), _p(dmu),
                               _p(dlv), dz.numel(), _st()), "gct_reparam_bwd")


def kld_fwd(mu, log_var):
    out = torch.empty((), dtype=torch.float32, device=mu.device)
    ws = workspace(4096, mu.device)
    check(_L().gct_kld_fwd(_p(mu), _p(log_var), _p(out), _p(ws), mu.numel(), _st()), "gct_kld_fwd")
    return out


def kld_bwd(mu, log_var, gout):
    dmu, dlv = torch.empty_like(mu), torch.empty_like(log_var)
    check(_L().gct_kld_bwd(_p(mu), _p(log_var), _p(gout), _p(dmu), _p(dlv), mu.numel(), _st()),
          "gct_kld_bwd")
    return dmu, dlv


def ce_fwd(logits2d, target, pad_id):
    _chk(target, "ce.target", torch.int64)
    rows, V = logits2d.shape
    out = torch.empty((), dtype=torch.float32, device=logits2d.device)
    ws = workspace(4096, logits2d.device)
    check(_L().gct_ce_fwd(_p(logits2d), _p(target), _p(out), _p(ws), rows, V, pad_id, _st()),
          "gct_ce_fwd")
    return out


def ce_bwd(logits2d, target, gout, pad_id):
    rows, V = logits2d.shape
    dl = torch.empty_like(logits2d)
    check(_L().gct_ce_bwd(_p(logits2d), _p(target), _p(gout), _p(dl), rows, V, pad_id, _st()),
          "gct_ce_bwd")
    return dl


# ------------------------------------------------------------------------------- optimiser
def adam_step(p, g, m, v, lr, b1, b2, eps, step, gscale=1.0, guard=True):
    """guard: the update is skipped ON THE DEVICE when a gradient has landed on a decoder row that the forward skipped
    (skipped_row_gradients() != 0): the wrong gradient is never applied, and the next read-back raises."""
    check(_L().gct_adam_step_guarded(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, b1, b2, eps, step, gscale,
                                     _p(skipped_row_gradients()) if guard else None, _st()), "gct_adam_step_guarded")


# ---------------------------------------------------------------------------------- utility
def copy_rows(src, src_rpb, src_off, dst, dst_rpb, dst_off, rows, rpb, cols, accumulate=False):
    check(_L().gct_copy_rows(_p(src), src_rpb, src_off, _p(dst), dst_rpb, dst_off, rows, rpb, cols,
                             int(accumulate), _st()), "gct_copy_rows")


def small_linear_fwd(x, w, b):
    rows, K = x.shape
    N = w.shape[0]
    y = torch.empty(rows, N, dtype=torch.float32, device=x.device)
    check(_L().gct_small_linear_fwd(_p(x), _p(w), _p(b), _p(y), rows, K, N, _st()),
          "gct_small_linear_fwd")
    return y


def small_linear_bwd(dy, x, dw, db):
    rows, K = x.shape
    N = dw.shape[0]
    check(_L().gct_small_linear_bwd(_p(dy), _p(x), _p(dw), _p(db), rows, K, N, _st()),
          "gct_small_linear_bwd")


def add(a, b, out=None):
    y = torch.empty_like(a) if out is None else out
    check(_L().gct_add(_p(a), _p(b), _p(y), a.numel(), _st()), "gct_add")
    return y


# ----------------------------------------------------------------------------------- decode
def attn_decode(q, ldq, k, v, kv_row, kv_batch, valid_u8, valid_sb, out, n, H, Lc, dk, pos=None, cache_off=0,
                knew=None, vnew=None, ldn=0, klen=None):
    """klen (int32 [n], optional, fixed caches only): leading keys to look at per sample (the rest are masked)."""
    check(_L().gct_attn_decode(_p(q), ldq, _p(k), _p(v), kv_row, kv_batch, _p(valid_u8), valid_sb,
                               _p(out), out.stride(0), n, H, Lc, dk, 1.0 / math.sqrt(dk), _p(pos), cache_off,
                               _p(knew), _p(vnew), ldn, _p(klen), _st()), "gct_attn_decode")


def attn_decode_z(q, qoff, z, ckv, nc, valid_u8, out, ooff, n, H, dk, klen=None):
    """Cross-attention of one decode step over the latent rows z [n, Le, lat] (gct_attn_decode_z): q [n, ldq] holds the
    plain query in columns [0, H*dk) (read only with condition rows) and the G^T-folded query at qoff; ckv [n*nc, >= 2*H*dk]
    the shifted keys | values of the nc condition rows; out [n, ldo] gets the per-head latent context at ooff (and the
    condition rows' part in columns [0, H*dk) when nc > 0)."""
    Le, lat = z.shape[1], z.shape[2]
    check(_L().gct_attn_decode_z(_p(q), q.stride(0), qoff, _p(z), z.stride(0), lat, Le, _p(ckv),
                                 0 if ckv is None else nc * ckv.stride(0), 0 if ckv is None else ckv.stride(0), nc,
                                 _p(valid_u8), 0 if valid_u8 is None else valid_u8.stride(0), _p(klen), _p(out),
                                 out.stride(0), ooff, n, H, dk, 1.0 / math.sqrt(dk), _st()), "gct_attn_decode_z")


def select_token(logits2d, ys, pos, valid_u8, done_u8, mode, pad_id, eos_id, seed=0, probs_out=None, pos_dev=None,
                 valid_off=0, seed_dev=None):
    n, V = logits2d.shape
    check(_L().gct_select_token(_p(logits2d), V, _p(ys), ys.stride(0), pos, _p(valid_u8),
                                valid_u8.stride(0) if valid_u8 is not None else 0, _p(done_u8),
                                _p(probs_out), n, mode, pad_id, eos_id, seed, _p(pos_dev), valid_off, _p(seed_dev),
                                _st()), "gct_select_token")
