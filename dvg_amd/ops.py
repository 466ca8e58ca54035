"""Tensor-level wrappers over the C ABI (one Python function per entry point).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every
arithmetic op of the hot path is a kernel of libdvg_hip.so.  Activations travel as
(N,C,H,W) tensors with channels_last strides, i.e. NHWC in memory.
"""
from __future__ import annotations

import os

import torch

from ._lib import check, lib

ACT_NONE, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3
MODE_CONV3, MODE_CONV4S2, MODE_CONVT4S2 = 0, 1, 2


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (name, flops, bytes, ev0, ev1)

    def summary(self):
        """Per kernel name: launches, ms, EXECUTED flops, algorithmic bytes, and `alg_flops` = the direct-form FLOPs of the
        layer a launch belongs to (differs from `flops` only for the Winograd GEMMs, which execute 1/2.25 or 1/4 of them)."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, by, e0, e1, alg in self.records:
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "alg_flops": 0.0})
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += fl
            a["bytes"] += by
            a["alg_flops"] += fl if alg is None else alg
        return agg


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def _run(name, flops, nbytes, fn, *args, alg_flops=None):
    """Launch through the C ABI; with a KernelTimer installed, bracket the launch with events."""
    if _timer is None:
        check(fn(*args), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), name)
    e1.record()
    _timer.records.append((name, flops, nbytes, e0, e1, alg_flops))


def _p(t):
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor — the DVG hot path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")


def nhwc_empty(n: int, c: int, h: int, w: int, device) -> torch.Tensor:
    """(N,C,H,W)-shaped view of a fresh NHWC buffer."""
    return torch.empty((n, h, w, c), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def is_nhwc(t: torch.Tensor) -> bool:
    if t.dim() != 4:
        return False
    n, c, h, w = t.shape
    return t.stride() == (h * w * c, 1, w * c, c) or (c == 1 and t.is_contiguous()) or \
        (h == 1 and w == 1 and t.is_contiguous())


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return `t` (N,C,H,W) as an NHWC-in-memory tensor, converting with the layout kernel if needed."""
    _dev_f32(t, "to_nhwc")
    if is_nhwc(t):
        return t
    src = t if t.is_contiguous() else t.contiguous()
    n, c, h, w = src.shape
    out = nhwc_empty(n, c, h, w, t.device)
    check(lib().dvg_nchw_to_nhwc(_p(src), _p(out), n, c, h, w, _stream()), "nchw_to_nhwc")
    return out


def to_nchw(t: torch.Tensor) -> torch.Tensor:
    """Contiguous NCHW copy of an NHWC-in-memory tensor."""
    _dev_f32(t, "to_nchw")
    if t.is_contiguous():
        return t
    if not is_nhwc(t):
        return t.contiguous()
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), device=t.device, dtype=torch.float32)
    check(lib().dvg_nhwc_to_nchw(_p(t), _p(out), n, c, h, w, _stream()), "nhwc_to_nchw")
    return out


# ----------------------------------------------------------------------------------
# weight packing
# ----------------------------------------------------------------------------------
def pack_conv_weight(w: torch.Tensor) -> torch.Tensor:
    _dev_f32(w, "pack_conv_weight")
    w = w.detach().contiguous()
    co, ci, kh, kw = w.shape
    out = torch.empty((kh * kw, co, ci), device=w.device, dtype=torch.float32)
    check(lib().dvg_pack_conv_weight(_p(w), _p(out), co, ci, kh, kw, _stream()), "pack_conv_weight")
    return out


def pack_convT_weight(w: torch.Tensor) -> torch.Tensor:
    _dev_f32(w, "pack_convT_weight")
    w = w.detach().contiguous()
    ci, co, kh, kw = w.shape
    out = torch.empty((kh * kw, co, ci), device=w.device, dtype=torch.float32)
    check(lib().dvg_pack_convT_weight(_p(w), _p(out), ci, co, kh, kw, _stream()), "pack_convT_weight")
    return out


def packed_row_floats() -> int:
    """Floats per packed weight row (the 16 k-values of one output channel): 16 for the native f32-MFMA build of the library,
    24 (three planes of 16 bf16) for the default build, whose fp32 products run as exact bf16 triples on the bf16 MFMA."""
    return int(lib().dvg_packed_row_floats())


def pack_igemm_weight(w: torch.Tensor, transposed: bool = False) -> torch.Tensor:
    """Weight in the layout the implicit-GEMM kernels read; logical shape [Cin/16][taps][Cout][row], in memory
    [Cin/16][Cout/64][tap slot][64][row] (conv_igemm2.hip: pack_k16_kernel).  Cout % 64 == 0."""
    _dev_f32(w, "pack_igemm_weight")
    w = w.detach().contiguous()
    if transposed:
        ci, co, kh, kw = w.shape
    else:
        co, ci, kh, kw = w.shape
    out = torch.empty((ci // 16, kh * kw, co, packed_row_floats()), device=w.device, dtype=torch.float32)
    check(lib().dvg_pack_conv_weight_k16(_p(w), _p(out), co, ci, kh, kw, int(transposed), _stream()), "pack_k16")
    return out


def _wp_dims(wp: torch.Tensor):
    """(taps, cout, cin) of a packed igemm weight [Cin/16][taps][Cout][row]."""
    if wp.dim() != 4 or wp.shape[3] != packed_row_floats():
        raise RuntimeError(f"packed igemm weight must be [Cin/16][taps][Cout][{packed_row_floats()}], got {tuple(wp.shape)}")
    return wp.shape[1], wp.shape[2], wp.shape[0] * 16


def unpack_conv_weight(wp: torch.Tensor, kh: int, kw: int) -> torch.Tensor:
    t, co, ci = wp.shape
    out = torch.empty((co, ci, kh, kw), device=wp.device, dtype=torch.float32)
    check(lib().dvg_unpack_conv_weight(_p(wp.contiguous()), _p(out), co, ci, kh, kw, _stream()), "unpack_conv_weight")
    return out


def unpack_convT_weight(wp: torch.Tensor, kh: int, kw: int) -> torch.Tensor:
    t, co, ci = wp.shape
    out = torch.empty((ci, co, kh, kw), device=wp.device, dtype=torch.float32)
    check(lib().dvg_unpack_convT_weight(_p(wp.contiguous()), _p(out), ci, co, kh, kw, _stream()),
          "unpack_convT_weight")
    return out


# ----------------------------------------------------------------------------------
# conv blocks.  All take / return NHWC-in-memory (N,C,H,W) tensors.
# `stats=True` returns (y, stats_partial) with stats_partial [rows][2][Cout].
# ----------------------------------------------------------------------------------
def _splitk_ws(mode, n, h, w, cin, cout, out_numel, device):
    """Workspace for the split-K path of the implicit GEMM (None when the launch fills the chip on its own)."""
    s = lib().dvg_conv_splitk_v2(mode, n, h, w, cin, cout)
    if s <= 1:
        return None
    # One buffer per (stream, size), reused by every launch of that size on that stream (launches on a stream are ordered).
    # Keyed by the hipGraph capture too: a buffer allocated during a capture lives in that graph's private pool.
    stream = torch.cuda.current_stream(device).cuda_stream
    key = (device.index if device.index is not None else torch.cuda.current_device(), stream, s * out_numel,
           lib().dvg_stream_capture_id(stream))
    ws = _SPLITK_WS.get(key)
    if ws is None:
        if len(_SPLITK_WS) > 256:
            _SPLITK_WS.clear()
        ws = _SPLITK_WS[key] = torch.empty(s * out_numel, device=device, dtype=torch.float32)
    return ws


_SPLITK_WS = {}


def evict_captured_workspaces() -> None:
    """Drop the split-K workspaces that were allocated from a hipGraph's private pool (capture id != 0).  Called by
    rollout.snapshot_eager_caches right after a capture has ended: the graph replays through raw pointers and its pool
    keeps the memory, so the tensor objects are only needed WHILE the capture runs (later launches of the same size reuse
    them) - kept afterwards they would pin segments of a pool that GraphedIteration._release() / a sampler rebuild wants
    to free."""
    for k in [k for k in _SPLITK_WS if k[-1] != 0]:
        del _SPLITK_WS[k]


def _stats_buf(rows: int, cout: int, device, tile_images: int = 0):
    """tile_images > 0: the rows are per-TILE partial sums in image-major order, a tile spanning `tile_images` consecutive
    images (recorded on the tensor: fused.group_stats may then cut the rows into per-group runs without another pass);
    0: rows of a split-K finish launch, whose blocks do not respect image boundaries."""
    if rows <= 0:
        raise RuntimeError("unsupported shape for fused BN statistics")
    st = torch.empty((rows, 2, cout), device=device, dtype=torch.float32)
    st.tile_images = tile_images
    return st


def _tile_images(mode, n, h, w, cin, cout, gh, gw, ws):
    """Images per statistics row of an igemm launch (see _stats_buf): 4 on 4x4 tile grids, else 1; 0 when the launch
    splits K (its finish kernel writes the statistics)."""
    if ws is not None and lib().dvg_conv_splitk_v2(mode, n, h, w, cin, cout) != 1:
        return 0
    return 4 if (gh, gw) == (4, 4) else 1


class SharedBlocks:
    """A tensor of `blocks` consecutive blocks of `block` images each, shared by the groups of a larger batch: group g (the
    g-th run of `block` images of that batch) uses block map[g].  The time-batched decoder calls of train.py:227-231: the
    three calls of a time step share the step's skip tensors, and once the skip is frozen (i >= n_past) every later step
    shares them too.  `map_dev`: int32 device tensor [groups]; `map_host`: the same as a tuple."""
    __slots__ = ("t", "block", "map_dev", "map_host")

    def __init__(self, t, block, map_dev, map_host):
        self.t, self.block, self.map_dev, self.map_host = t, int(block), map_dev, tuple(map_host)

    @property
    def blocks(self):
        return self.t.shape[0] // self.block

    @property
    def groups(self):
        return len(self.map_host)

    def like(self, t):
        """The same sharing pattern over another tensor of blocks (e.g. a conv of this one)."""
        return SharedBlocks(t, self.block, self.map_dev, self.map_host)


_MAP_CACHE = {}


def shared_map(map_host, device):
    """int32 device tensor of a group -> block map, cached per (map, device): created eagerly (an H2D copy is not capturable),
    so the first - eager, warm-up - iteration of a training run creates the ones the captured iterations use."""
    key = (tuple(map_host), str(device))
    m = _MAP_CACHE.get(key)
    if m is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("shared_map: a new group map cannot be created while a hipGraph is being captured")
        m = _MAP_CACHE[key] = torch.tensor(list(map_host), dtype=torch.int32, device=device)
    return m


def group_sum(src, shared: "SharedBlocks"):
    """dst[b] = sum_{g: map[g] == b} src[g] (deterministic order): src (groups * block, ...) -> (blocks * block, ...), the
    adjoint of reading a SharedBlocks operand (dvg_group_sum)."""
    _dev_f32(src, "group_sum")
    g, blk = shared.groups, shared.block
    if src.shape[0] != g * blk:
        raise RuntimeError(f"group_sum: {src.shape[0]} images are not {g} groups of {blk}")
    per_image = src.numel() // src.shape[0]
    if src.stride(0) != per_image:     # images must be dense and consecutive (NCHW-contiguous or NHWC-in-memory alike)
        src = src.contiguous()
    dst = torch.empty_strided((shared.blocks * blk,) + tuple(src.shape[1:]), src.stride(), device=src.device,
                              dtype=torch.float32)
    elems = src.numel() // g
    check(lib().dvg_group_sum(_p(src), _p(shared.map_dev), _p(dst), g, shared.blocks, elems, _stream()), "group_sum")
    return dst


def _check_addend(addend, y, excluded):
    """Returns (addend tensor, map pointer, block) for the kernels; addend may be a SharedBlocks."""
    if addend is None:
        return None, None, 0
    if excluded:
        raise RuntimeError("addend excludes the pooled output")
    if isinstance(addend, SharedBlocks):
        t = addend.t
        _dev_f32(t, "addend")
        if (tuple(t.shape[1:]) != tuple(y.shape[1:]) or t.stride() != y.stride() or
                y.shape[0] != addend.groups * addend.block):
            raise RuntimeError(f"shared addend {tuple(t.shape)} x {addend.groups} groups does not match {tuple(y.shape)}")
        return t, addend.map_dev, addend.block
    _dev_f32(addend, "addend")
    if tuple(addend.shape) != tuple(y.shape) or addend.stride() != y.stride():
        raise RuntimeError(f"addend {tuple(addend.shape)} must match the output {tuple(y.shape)} (NHWC in memory)")
    return addend, None, 0


def conv3x3(x, skip, wp, scale, shift, *, upsample=False, act=ACT_LRELU, slope=0.2, pool=False, stats=False,
            addend=None):
    """`addend`: raw partial sums in the output's NHWC shape, y = act((conv + addend) * scale + shift) (v2 only)."""
    _dev_f32(x, "conv3x3.x")
    assert is_nhwc(x), "conv3x3: x must be NHWC in memory"
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0
    if skip is not None:
        _dev_f32(skip, "conv3x3.skip")
        assert is_nhwc(skip), "conv3x3: skip must be NHWC in memory"
        c2 = skip.shape[1]
        if tuple(skip.shape) != (n, c2, h, w):
            raise RuntimeError(f"conv3x3: skip shape {tuple(skip.shape)} does not match {(n, c2, h, w)}")
    taps, cout, cin = _wp_dims(wp)
    if taps != 9 or cin != c1 + c2:
        raise RuntimeError(f"conv3x3: packed weight {tuple(wp.shape)} does not match Cin={c1 + c2}")
    y = nhwc_empty(n, cout, h, w, x.device)
    yp = nhwc_empty(n, cout, h // 2, w // 2, x.device) if pool else None
    addend, amap, ablk = _check_addend(addend, y, pool)
    ws = _splitk_ws(MODE_CONV3, n, h, w, cin, cout, y.numel(), x.device)
    if stats:
        rows = lib().dvg_conv_stats_rows_v2(MODE_CONV3, n, h, w, cin, cout, int(pool), int(ws is not None))
    st = _stats_buf(rows, cout, x.device, _tile_images(MODE_CONV3, n, h, w, cin, cout, h, w, ws)) if stats else None
    fl, by = 2.0 * n * h * w * cout * 9 * cin, 4.0 * (x.numel() + (skip.numel() if c2 else 0) + n * h * w * cout +
                                                      wp.numel())
    _run("conv3x3_igemm", fl, by, lib().dvg_conv3x3_bn_act_v2, _p(x), _p(skip), _p(wp), _p(scale), _p(shift), _p(y),
         _p(yp), _p(st), n, h, w, c1, c2, cout, int(upsample), act, slope, _p(ws), 0 if ws is None else ws.numel(),
         _p(addend), _p(amap), ablk, _stream())
    out = (y, yp) if pool else y
    return (out, st) if stats else out


# ----------------------------------------------------------------------------------
# Winograd F(2x2,3x3) form of the 3x3 block (eval-mode deep layers; dvg_amd/csrc/winograd.hip)
# ----------------------------------------------------------------------------------
def winograd_weight(w: torch.Tensor, m: int = 2) -> torch.Tensor:
    """U = G g G^T of a Conv2d weight (Cout,Cin,3,3) for F(m x m, 3x3), in the k16 layout of the (m+2)^2 batched GEMMs:
    logical shape ((m+2)^2, Cin/16, 1, Cout, row), in memory [(m+2)^2][Cout/64][Cin/16][64][row]."""
    _dev_f32(w, "winograd_weight")
    w = w.detach().contiguous()
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3) or ci % 16 or m not in (2, 4):
        raise RuntimeError("winograd_weight: (Cout, Cin % 16 == 0, 3, 3) and m in (2, 4) expected")
    u = torch.empty(((m + 2) ** 2, ci // 16, 1, co, packed_row_floats()), device=w.device, dtype=torch.float32)
    check(lib().dvg_winograd_weight(_p(w), _p(u), co, ci, m, _stream()), "winograd_weight")
    return u


def winograd_ok(n, c, h, w, cout, m: int = 2) -> bool:
    """Shapes the Winograd path takes: whole 128-row GEMM tiles and 64-channel blocks."""
    return h % m == 0 and w % m == 0 and c % 64 == 0 and cout % 64 == 0 and (n * (h // m) * (w // m)) % 128 == 0


class WinoV:
    """The Winograd F(4x4,3x3) input transform V (36, T, C) of an activation (N,C,H,W) that was never materialised: what
    conv3x3_winograd(..., to_v=True) hands to the next layer's conv3x3_winograd instead of y."""
    __slots__ = ("v", "shape", "up")

    def __init__(self, v, shape, up=False):
        # up: `shape` is the UPSAMPLED map the transform was taken of (conv3x3_winograd(to_v="up")): only a layer called with
        # upsample=True may consume it
        self.v, self.shape, self.up = v, tuple(shape), bool(up)

    @property
    def device(self):
        return self.v.device

    @property
    def requires_grad(self):
        return False


def winograd_chain_ok(n, c, h, w):
    """Shapes dvg_winograd_output_input takes (the fused output -> input transform between two F(4x4,3x3) layers)."""
    return h == w and h in (8, 16, 32) and c % 64 == 0 and n > 0


def winograd_up_chain_ok(n, c, h, w):
    """Shapes dvg_winograd_output_up_input takes (last layer of a decoder block -> upsample -> first conv of the next)."""
    return h == w and h == 8 and c % 64 == 0 and n > 0


def winograd_pool_chain_ok(n, c, h, w):
    """Shapes dvg_winograd_output_pool_input takes (last layer of an encoder stage -> first layer of the next stage)."""
    return h == w and h in (16, 32) and c % 64 == 0 and n > 0


def conv3x3_winograd(x, u, scale, shift, *, act=ACT_LRELU, slope=0.2, pool=False, return_v=False, to_v=False,
                     upsample=False, addend=None, y_from=0):
    """y = act(conv3x3(x) * scale + shift) (+ pooled y) through input transform -> (m+2)^2 batched GEMMs -> output
    transform; m (2 or 4) follows from u's leading dimension (16 or 36).  x may be a WinoV (the previous layer's to_v=True
    result: no input transform); to_v=True returns the NEXT layer's input transform as a WinoV instead of y (m = 4,
    winograd_chain_ok shapes: dvg_winograd_output_input); pool=True with to_v=True returns (y, WinoV of maxpool2x2(y)): the
    last layer of an encoder stage handing over to the first layer of the next (dvg_winograd_output_pool_input).
    to_v="up": the WinoV is the input transform of nearest_up2(y) (dvg_winograd_output_up_input; the consumer is a layer called
    with upsample=True, which then skips its own input transform).
    upsample (m = 4): x is read through nearest-x2 upsampling (output 2H x 2W).  addend (m = 4, no pool): raw partial sums
    in the output's shape, y = act((conv + addend) * scale + shift) - the hoisted skip half of a decoder block's first conv.
    y_from (pool only): y is stored for the images [y_from, N) only (N - y_from images; None when y_from == N)."""
    from_v = isinstance(x, WinoV)
    if from_v:
        n, c, h, w = x.shape
        if x.up != bool(upsample):
            raise RuntimeError("conv3x3_winograd: a WinoV taken through the upsampling needs upsample=True (and only then)")
    else:
        _dev_f32(x, "conv3x3_winograd.x")
        assert is_nhwc(x), "conv3x3_winograd: x must be NHWC in memory"
        n, c, h, w = x.shape
        if upsample:
            h, w = 2 * h, 2 * w
    cout, npos = u.shape[3], u.shape[0]
    mt = {16: 2, 36: 4}.get(npos, 0)
    if mt == 0 or tuple(u.shape) != (npos, c // 16, 1, cout, packed_row_floats()) or not winograd_ok(n, c, h, w, cout, mt):
        raise RuntimeError(f"conv3x3_winograd: unsupported shape x {tuple(x.shape)} u {tuple(u.shape)}")
    if not 0 <= y_from <= n or (y_from and not pool):      # (all argument checks BEFORE the first launch: ADVICE r05)
        raise RuntimeError("conv3x3_winograd: y_from needs the pooled output and 0 <= y_from <= N")
    if (from_v or to_v or upsample or addend is not None) and (mt != 4 or return_v):
        raise RuntimeError("conv3x3_winograd: WinoV hand-over / upsample / addend need F(4x4,3x3) and no return_v")
    if addend is not None and pool:
        raise RuntimeError("conv3x3_winograd: addend excludes the pooled output")
    if to_v == "up" and (pool or addend is not None or not winograd_up_chain_ok(n, cout, h, w)):
        raise RuntimeError(f"conv3x3_winograd: to_v='up' unsupported for output {(n, cout, h, w)} pool={pool}")
    if addend is not None:
        _dev_f32(addend, "conv3x3_winograd.addend")
        if tuple(addend.shape) != (n, cout, h, w) or not is_nhwc(addend):
            raise RuntimeError(f"conv3x3_winograd: addend {tuple(addend.shape)} must be NHWC {(n, cout, h, w)}")
    if to_v and to_v != "up" and not (winograd_pool_chain_ok(n, cout, h, w) if pool else winograd_chain_ok(n, cout, h, w)):
        raise RuntimeError(f"conv3x3_winograd: to_v unsupported for output {(n, cout, h, w)} pool={pool}")
    t = n * (h // mt) * (w // mt)
    dev = x.device
    m = torch.empty((npos, t, cout), device=dev, dtype=torch.float32)
    if from_v:
        v = x.v
        if tuple(v.shape) != (npos, t, c):
            raise RuntimeError("conv3x3_winograd: WinoV does not match its shape")
    else:
        v = torch.empty((npos, t, c), device=dev, dtype=torch.float32)
        _run("winograd_input", 0.0, 4.0 * (x.numel() + v.numel()), lib().dvg_winograd_input, _p(x), _p(v), n, h, w, c, mt,
             int(upsample), _stream())
    _run("winograd_gemm", 2.0 * npos * t * c * cout, 4.0 * (v.numel() + m.numel() + u.numel()), lib().dvg_gemm_batched_k16,
         _p(v), _p(u), _p(m), npos, t // 16, 16, c, cout, _stream(), alg_flops=2.0 * n * h * w * cout * 9 * c)
    if to_v and pool:
        y = nhwc_empty(n - y_from, cout, h, w, dev) if y_from < n else None
        vn = torch.empty((npos, t // 4, cout), device=dev, dtype=torch.float32)
        _run("winograd_output_pool_input", 0.0, 4.0 * (m.numel() + (0 if y is None else y.numel()) + vn.numel()),
             lib().dvg_winograd_output_pool_input, _p(m), _p(scale), _p(shift), _p(y), _p(vn), n, h, w, cout, act, slope, y_from,
             _stream())
        return y, WinoV(vn, (n, cout, h // 2, w // 2))
    if to_v == "up":
        vn = torch.empty((npos, 4 * t, cout), device=dev, dtype=torch.float32)
        _run("winograd_output_up_input", 0.0, 4.0 * (m.numel() + vn.numel()), lib().dvg_winograd_output_up_input, _p(m), _p(scale),
             _p(shift), _p(vn), n, h, w, cout, act, slope, _stream())
        return WinoV(vn, (n, cout, 2 * h, 2 * w), up=True)
    if to_v:
        vn = torch.empty((npos, t, cout), device=dev, dtype=torch.float32)
        _run("winograd_output_input", 0.0, 4.0 * (m.numel() + vn.numel()), lib().dvg_winograd_output_input, _p(m), _p(scale),
             _p(shift), _p(vn), n, h, w, cout, act, slope, _p(addend), _stream())
        return WinoV(vn, (n, cout, h, w))
    y = nhwc_empty(n - y_from, cout, h, w, dev) if y_from < n else None
    yp = nhwc_empty(n, cout, h // 2, w // 2, dev) if pool else None
    _run("winograd_output", 0.0, 4.0 * (m.numel() + (0 if y is None else y.numel()) + (0 if yp is None else yp.numel())),
         lib().dvg_winograd_output, _p(m), _p(scale), _p(shift), _p(y), _p(yp), n, h, w, cout, act, slope, mt, _p(addend), y_from,
         _stream())
    if return_v:   # the input transform (P, T, C): the Winograd-form weight gradient's second operand (training)
        return ((y, yp) if pool else y), v
    return (y, yp) if pool else y


def conv3x3_first(x_nchw, w, scale, shift, *, act=ACT_LRELU, slope=0.2, stats=False):
    _dev_f32(x_nchw, "conv3x3_first.x")
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    n, nc, h, wd = x.shape
    w = w.detach()
    cout = w.shape[0]
    if tuple(w.shape) != (cout, nc, 3, 3) or not w.is_contiguous():
        raise RuntimeError("conv3x3_first: weight must be contiguous (Cout,nc,3,3)")
    y = nhwc_empty(n, cout, h, wd, x.device)
    st = _stats_buf(lib().dvg_conv_first_stats_rows(3, n, h, wd), cout, x.device, 1) if stats else None
    _run("conv3x3_first", 2.0 * n * h * wd * cout * 9 * nc, 4.0 * (x.numel() + n * h * wd * cout),
         lib().dvg_conv3x3_first, _p(x), _p(w), _p(scale), _p(shift), _p(y), _p(st), n, h, wd, nc, cout, act, slope,
         _stream())
    return (y, st) if stats else y


def first_pair_ok(n, nc, h, w, cout) -> bool:
    """Shapes `conv3x3_first_pair` takes: one input channel, 8 x 16 tiles, a launch that fills the chip."""
    return nc == 1 and h % 8 == 0 and w % 16 == 0 and cout % 64 == 0 and n * (h // 8) * (w // 16) * (cout // 64) >= 512


def conv3x3_first_pair(x_nchw, w0, scale0, shift0, wp1, scale1, shift1, *, act=ACT_LRELU, slope=0.2, pool=False, y_from=0):
    """vgg_layer(1, 64) -> vgg_layer(64, Cout) (+ 2x2 max-pool) of the encoder's first stage in eval mode as ONE launch
    (dvg_conv3x3_first_pair): the 64-channel activation between the two layers is never materialised.
    w0: the first layer's (64,1,3,3) weight as a contiguous (9, 64) tensor [tap][channel].
    y_from (pool only): the full-resolution output is stored for the images [y_from, N) only - y has N - y_from images, None
    when y_from == N (a rollout discards the skip tensors of every frame but the last conditioning one)."""
    _dev_f32(x_nchw, "conv3x3_first_pair.x")
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    n, nc, h, wd = x.shape
    taps, cout, cin = _wp_dims(wp1)
    w0 = w0.detach()
    if nc != 1 or tuple(w0.shape) != (9, 64) or not w0.is_contiguous() or (taps, cin) != (9, 64) or \
            not first_pair_ok(n, nc, h, wd, cout):
        raise RuntimeError(f"conv3x3_first_pair: unsupported shapes x {tuple(x.shape)} w0 {tuple(w0.shape)} wp1 {tuple(wp1.shape)}")
    if not 0 <= y_from <= n or (y_from and not pool):
        raise RuntimeError("conv3x3_first_pair: y_from needs the pooled output and 0 <= y_from <= N")
    y = nhwc_empty(n - y_from, cout, h, wd, x.device) if y_from < n else None
    yp = nhwc_empty(n, cout, h // 2, wd // 2, x.device) if pool else None
    flops = 2.0 * n * h * wd * (64 * 9 + cout * 9 * 64)
    _run("conv3x3_igemm", flops, 4.0 * (x.numel() + (0 if y is None else y.numel()) + (0 if yp is None else yp.numel()) + wp1.numel()),
         lib().dvg_conv3x3_first_pair, _p(x), _p(w0), _p(scale0), _p(shift0), _p(wp1), _p(scale1), _p(shift1), _p(y), _p(yp),
         n, h, wd, cout, act, slope, y_from, _stream(), alg_flops=flops)
    return (y, yp) if pool else y


def convT3x3_last(x, w, bias, nc, *, act=ACT_SIGMOID):
    _dev_f32(x, "convT3x3_last.x")
    assert is_nhwc(x)
    n, cin, h, wd = x.shape
    w = w.detach()
    if tuple(w.shape) != (cin, nc, 3, 3) or not w.is_contiguous():
        raise RuntimeError("convT3x3_last: weight must be contiguous (Cin,nc,3,3)")
    y = torch.empty((n, nc, h, wd), device=x.device, dtype=torch.float32)
    _run("convT3x3_last", 2.0 * n * h * wd * cin * 9 * nc, 4.0 * (x.numel() + y.numel()),
         lib().dvg_convT3x3_last, _p(x), _p(w), _p(bias), _p(y), n, h, wd, cin, nc, act, _stream())
    return y


CONV4S2_MAX_FLOATS = 1 << 31     # dvg_conv4x4s2_bn_act_v2: N * H * W * Cin must stay below (32-bit activation offsets)


def conv4x4s2(x, wp, scale, shift, *, act=ACT_LRELU, slope=0.2, stats=False):
    _dev_f32(x, "conv4x4s2.x")
    assert is_nhwc(x)
    n, cin, h, w = x.shape
    taps, cout, cin_w = _wp_dims(wp)
    if taps != 16 or cin_w != cin:
        raise RuntimeError(f"conv4x4s2: packed weight {tuple(wp.shape)} does not match Cin={cin}")
    y = nhwc_empty(n, cout, h // 2, w // 2, x.device)
    if x.numel() >= CONV4S2_MAX_FLOATS:
        # The parity-split kernel addresses the activation with 32-bit offsets (dvg_conv4x4s2_bn_act_v2 refuses N*H*W*Cin >=
        # 2^31): run the batch as several launches over runs of images (large time-batched 128 x 128 batches).
        # Statistics rows are per tile in image-major order (launches this large never split K), so the runs' rows
        # concatenate to the rows of the whole batch.
        per = (CONV4S2_MAX_FLOATS - 1) // (cin * h * w)
        per -= per % 8
        if per <= 0:
            raise RuntimeError(f"conv4x4s2: one image of {cin} x {h} x {w} exceeds the kernel's 32-bit offsets")
        sts = []
        for lo in range(0, n, per):
            r = conv4x4s2(x[lo:lo + per], wp, scale, shift, act=act, slope=slope, stats=stats)
            y[lo:lo + per].copy_(r[0] if stats else r)
            if stats:
                if r[1].tile_images == 0:
                    raise RuntimeError("conv4x4s2: a split-K run inside a batch split (unexpected at this size)")
                sts.append(r[1])
        if not stats:
            return y
        st = torch.cat(sts)
        st.tile_images = sts[0].tile_images
        return y, st
    ws = _splitk_ws(MODE_CONV4S2, n, h, w, cin, cout, y.numel(), x.device)
    if stats:
        rows = lib().dvg_conv_stats_rows_v2(MODE_CONV4S2, n, h, w, cin, cout, 0, int(ws is not None))
    st = _stats_buf(rows, cout, x.device, _tile_images(MODE_CONV4S2, n, h, w, cin, cout, h // 2, w // 2, ws)) if stats else None
    fl, by = 2.0 * n * (h // 2) * (w // 2) * cout * 16 * cin, 4.0 * (x.numel() + y.numel() + wp.numel())
    _run("conv4x4s2_igemm", fl, by, lib().dvg_conv4x4s2_bn_act_v2, _p(x), _p(wp), _p(scale), _p(shift), _p(y), _p(st),
         n, h, w, cin, cout, act, slope, _p(ws), 0 if ws is None else ws.numel(), _stream())
    return (y, st) if stats else y


def conv4x4s2_first(x_nchw, w, scale, shift, *, act=ACT_LRELU, slope=0.2, stats=False):
    _dev_f32(x_nchw, "conv4x4s2_first.x")
    x = x_nchw if x_nchw.is_contiguous() else x_nchw.contiguous()
    n, nc, h, wd = x.shape
    w = w.detach()
    cout = w.shape[0]
    if tuple(w.shape) != (cout, nc, 4, 4) or not w.is_contiguous():
        raise RuntimeError("conv4x4s2_first: weight must be contiguous (Cout,nc,4,4)")
    y = nhwc_empty(n, cout, h // 2, wd // 2, x.device)
    st = _stats_buf(lib().dvg_conv_first_stats_rows(4, n, h, wd), cout, x.device, 1) if stats else None
    _run("conv4x4s2_first", 2.0 * n * (h // 2) * (wd // 2) * cout * 16 * nc, 4.0 * (x.numel() + y.numel()),
         lib().dvg_conv4x4s2_first, _p(x), _p(w), _p(scale), _p(shift), _p(y), _p(st), n, h, wd, nc, cout, act, slope,
         _stream())
    return (y, st) if stats else y


def convT4x4s2(x, skip, wp, scale, shift, *, act=ACT_LRELU, slope=0.2, stats=False, addend=None):
    _dev_f32(x, "convT4x4s2.x")
    assert is_nhwc(x)
    n, c1, h, w = x.shape
    c2 = 0
    if skip is not None:
        assert is_nhwc(skip)
        c2 = skip.shape[1]
        if tuple(skip.shape) != (n, c2, h, w):
            raise RuntimeError("convT4x4s2: skip shape mismatch")
    taps, cout, cin = _wp_dims(wp)
    if taps != 16 or cin != c1 + c2:
        raise RuntimeError(f"convT4x4s2: packed weight {tuple(wp.shape)} does not match Cin={c1 + c2}")
    y = nhwc_empty(n, cout, 2 * h, 2 * w, x.device)
    addend, amap, ablk = _check_addend(addend, y, False)
    ws = _splitk_ws(MODE_CONVT4S2, n, h, w, cin, cout, y.numel(), x.device)
    if stats:
        rows = lib().dvg_conv_stats_rows_v2(MODE_CONVT4S2, n, h, w, cin, cout, 0, int(ws is not None))
    st = _stats_buf(rows, cout, x.device, _tile_images(MODE_CONVT4S2, n, h, w, cin, cout, h, w, ws)) if stats else None
    fl = 2.0 * n * h * w * cout * 16 * cin
    by = 4.0 * (x.numel() + (skip.numel() if c2 else 0) + y.numel() + wp.numel())
    _run("convT4x4s2_igemm", fl, by, lib().dvg_convT4x4s2_bn_act_v2, _p(x), _p(skip), _p(wp), _p(scale), _p(shift),
         _p(y), _p(st), n, h, w, c1, c2, cout, act, slope, _p(ws), 0 if ws is None else ws.numel(), _p(addend),
         _p(amap), ablk, _stream())
    return (y, st) if stats else y


def convT4x4s2_last(x, skip, w, bias, nc, *, act=ACT_TANH):
    _dev_f32(x, "convT4x4s2_last.x")
    assert is_nhwc(x)
    n, c1, h, wd = x.shape
    c2 = 0
    if skip is not None:
        assert is_nhwc(skip)
        c2 = skip.shape[1]
    w = w.detach()
    if tuple(w.shape) != (c1 + c2, nc, 4, 4) or not w.is_contiguous():
        raise RuntimeError("convT4x4s2_last: weight must be contiguous (Cin,nc,4,4)")
    y = torch.empty((n, nc, 2 * h, 2 * wd), device=x.device, dtype=torch.float32)
    _run("convT4x4s2_last", 2.0 * n * h * wd * (c1 + c2) * 16 * nc,
         4.0 * (x.numel() + (skip.numel() if c2 else 0) + y.numel()), lib().dvg_convT4x4s2_last, _p(x), _p(skip),
         _p(w), _p(bias), _p(y), n, h, wd, c1, c2, nc, act, _stream())
    return y


# ----------------------------------------------------------------------------------
# BatchNorm (train mode) helpers
# ----------------------------------------------------------------------------------
def channel_stats(u2d: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """u2d: [rows][C] contiguous -> partial stats [r][2][C]; groups: the rows are `groups` equal consecutive runs, each with
    its own partial rows (group-major: [groups * r][2][C])."""
    _dev_f32(u2d, "channel_stats")
    rows, c = u2d.shape
    if rows % groups:
        raise RuntimeError(f"channel_stats: {rows} rows do not split into {groups} groups")
    r = lib().dvg_channel_stats_rows(rows // groups)
    st = torch.empty((groups * r, 2, c), device=u2d.device, dtype=torch.float32)
    check(lib().dvg_channel_stats(_p(u2d), _p(st), rows // groups, c, groups, _stream()), "channel_stats")
    st.grouped = groups       # rows are per-group runs already (fused.group_stats)
    return st


SYNC_BN = None    # fused.set_sync_bn: (torch.distributed, group, world) while synchronised BatchNorm is on


def sync_partial_rows(partial: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """[groups * r][2][C] per-tile partial sums of this rank -> [groups][2][C] sums over ALL ranks (sync-BN): the rows of a
    group are added up in fp64 (what dvg_bn_finalize itself does with them), all-reduced in fp64 and rounded once."""
    dist, group, _ = SYNC_BN
    rows, two, c = partial.shape
    if rows % groups:
        raise RuntimeError(f"sync_partial_rows: {rows} partial rows do not split into {groups} groups")
    tot = partial.view(groups, rows // groups, two, c).sum(1, dtype=torch.float64)
    dist.all_reduce(tot, group=group)
    out = tot.to(torch.float32)
    out.grouped = groups
    return out


def bn_finalize(stats_partial, gamma, beta, running_mean, running_var, count, eps, momentum, save=False,
                num_batches_tracked=None, passes=0, groups=1, group_momenta=None):
    """`num_batches_tracked` (int64 device scalar of nn.BatchNorm2d) is advanced by `passes` inside the same launch.
    groups > 1 (time-batched training): stats_partial holds `groups` equal runs of partial rows, `count` is PER GROUP, the
    results are [groups][C] and the running statistics advance group after group (dvg_bn_running_update) with
    group_momenta = (first, middle, last); `passes` = the total over all groups."""
    rows, _, c = stats_partial.shape
    dev = stats_partial.device
    if groups == 1:
        scale = torch.empty(c, device=dev, dtype=torch.float32)
        shift = torch.empty(c, device=dev, dtype=torch.float32)
        sm = torch.empty(c, device=dev, dtype=torch.float32) if save else None
        si = torch.empty(c, device=dev, dtype=torch.float32) if save else None
        check(lib().dvg_bn_finalize(_p(stats_partial), rows, _p(gamma), _p(beta), _p(scale), _p(shift), _p(running_mean),
                                    _p(running_var), _p(sm), _p(si), c, float(count), eps, momentum,
                                    _p(num_batches_tracked), int(passes), 1, None, _stream()),
              "bn_finalize")
        return (scale, shift, sm, si) if save else (scale, shift)
    if rows % groups:
        raise RuntimeError(f"bn_finalize: {rows} partial rows do not split into {groups} groups")
    buf = torch.empty((5, groups, c), device=dev, dtype=torch.float32)
    scale, shift, sm, si, gv = buf[0], buf[1], buf[2], buf[3], buf[4]
    check(lib().dvg_bn_finalize(_p(stats_partial), rows // groups, _p(gamma), _p(beta), _p(scale), _p(shift), None, None,
                                _p(sm), _p(si), c, float(count), eps, 0.0, None, 0, groups, _p(gv), _stream()), "bn_finalize")
    if running_mean is not None:
        m0, m1, m2 = group_momenta
        check(lib().dvg_bn_running_update(_p(sm), _p(gv), groups, c, m0, m1, m2, _p(running_mean), _p(running_var),
                                          _p(num_batches_tracked), int(passes), _stream()), "bn_running_update")
    return (scale, shift, sm, si) if save else (scale, shift)


def bn_act_apply(u, scale, shift, *, act=ACT_LRELU, slope=0.2, pool=False, inplace=True):
    """u NHWC (N,C,H,W); returns y (and pooled y).  scale / shift [C], or [G][C] for G consecutive groups of N / G images."""
    assert is_nhwc(u)
    n, c, h, w = u.shape
    groups = scale.shape[0] if scale.dim() == 2 else 1
    if n % groups or not scale.is_contiguous() or not shift.is_contiguous():
        raise RuntimeError("bn_act_apply: group coefficients must be contiguous [G][C] with G dividing N")
    y = u if inplace else nhwc_empty(n, c, h, w, u.device)
    yp = nhwc_empty(n, c, h // 2, w // 2, u.device) if pool else None
    check(lib().dvg_bn_act_apply(_p(u), _p(scale), _p(shift), _p(y), _p(yp), n, h, w, c, act, slope,
                                 0 if groups == 1 else n // groups, _stream()), "bn_act_apply")
    return (y, yp) if pool else y


# ----------------------------------------------------------------------------------
# dense / recurrent
# ----------------------------------------------------------------------------------
def gemm_nt(a, w, scale, shift, *, act=ACT_NONE, slope=0.0, period=None, splitk=1, out=None, accumulate=False):
    """out[m][n] = act((sum_k a[m][k] w[n][k]) * scale[n%period] + shift[n%period]); accumulate: out += (needs `out`)."""
    _dev_f32(a, "gemm_nt.a")
    _dev_f32(w, "gemm_nt.w")
    if a.dim() != 2 or a.stride(1) != 1:
        a = a.contiguous().view(a.shape[0], -1)
    w = w if w.is_contiguous() else w.contiguous()
    m, k = a.shape
    n, kw = w.shape
    if kw != k:
        raise RuntimeError(f"gemm_nt: K mismatch {k} vs {kw}")
    if period is None:
        period = n
    if out is None:
        out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    ws = torch.empty((splitk, m, n), device=a.device, dtype=torch.float32) if splitk > 1 else None
    _run("gemm_nt", 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n), lib().dvg_gemm_nt_bias_act, _p(a), _p(w),
         _p(scale), _p(shift), _p(out), _p(ws), m, n, k, a.stride(0), out.stride(0), period, splitk, act, slope,
         int(accumulate), _stream())
    return out


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh, want_gates=False):
    for t, nm in ((x, "x"), (h, "h"), (c, "c")):
        _dev_f32(t, "lstm_cell." + nm)
    b, hid = h.shape
    x = x if x.is_contiguous() else x.contiguous()
    h = h if h.is_contiguous() else h.contiguous()
    c = c if c.is_contiguous() else c.contiguous()
    if tuple(x.shape) != (b, hid) or tuple(w_ih.shape) != (4 * hid, hid) or tuple(w_hh.shape) != (4 * hid, hid):
        raise RuntimeError("lstm_cell: shape mismatch (input size must equal hidden size)")
    h_out = torch.empty_like(h)
    c_out = torch.empty_like(c)
    gates = torch.empty((b, 4 * hid), device=h.device, dtype=torch.float32) if want_gates else None
    _run("lstm_cell", 2.0 * b * 4 * hid * 2 * hid, 4.0 * (8 * hid * hid + 5 * b * hid), lib().dvg_lstm_cell, _p(x),
         _p(h), _p(c), _p(w_ih.detach()), _p(w_hh.detach()), _p(b_ih.detach()), _p(b_hh.detach()), _p(h_out),
         _p(c_out), _p(gates), b, hid, _stream())
    return (h_out, c_out, gates) if want_gates else (h_out, c_out)


def lstm_cell_pre(pre, h, c, w_hh, h_out, c_out, gates_out):
    """One LSTMCell step whose input half `pre` = W_ih x + b_ih + b_hh (B,4H) is given (dvg_lstm_cell_pre); writes into the
    caller's h_out / c_out / gates_out (slices of the per-sequence buffers of autograd._LSTMSequence)."""
    b, hid = h.shape
    _run("lstm_cell", 2.0 * b * 4 * hid * hid, 4.0 * (4 * hid * hid + 9 * b * hid), lib().dvg_lstm_cell_pre, _p(pre), _p(h),
         _p(c), _p(w_hh), _p(h_out), _p(c_out), _p(gates_out), b, hid, _stream())


def lstm_cell_bwd(dh_a, dh_b, dc, gates, c_prev, c_new, w_hh_t, dG, dc_prev, dh_prev):
    """One BPTT step of an LSTMCell in one launch (dvg_lstm_cell_bwd): dG, dc_prev and dh_prev = dG W_hh."""
    b, hid = c_new.shape
    _run("lstm_cell_bwd", 2.0 * b * 4 * hid * hid, 4.0 * (4 * hid * hid + 12 * b * hid), lib().dvg_lstm_cell_bwd, _p(dh_a),
         _p(dh_b), _p(dc), _p(gates), _p(c_prev), _p(c_new), _p(w_hh_t), _p(dG), _p(dc_prev), _p(dh_prev), b, hid, _stream())


def lstm_cell_x(x, h, c, w_x, w_hh, bias):
    """First cell of a time step with the embedding folded in (dvg_lstm_cell_x): x (B,Kx) raw LSTM input, w_x = W_ih W_e
    zero-padded to (4H,Kxp), bias = W_ih b_e + b_ih + b_hh.  Inference path."""
    for t, nm in ((x, "x"), (h, "h"), (c, "c")):
        _dev_f32(t, "lstm_cell_x." + nm)
    b, hid = h.shape
    if x.dim() != 2 or x.stride(1) != 1:
        x = x.contiguous().view(x.shape[0], -1)
    h = h if h.is_contiguous() else h.contiguous()
    c = c if c.is_contiguous() else c.contiguous()
    kx = x.shape[1]
    if x.shape[0] != b or tuple(w_x.shape) != (4 * hid, w_x.shape[1]) or w_x.shape[1] < kx or \
            tuple(w_hh.shape) != (4 * hid, hid) or bias.numel() != 4 * hid:
        raise RuntimeError("lstm_cell_x: shape mismatch")
    h_out, c_out = torch.empty_like(h), torch.empty_like(c)
    _run("lstm_cell", 2.0 * b * 4 * hid * (kx + hid), 4.0 * (4 * hid * (kx + hid) + 5 * b * hid), lib().dvg_lstm_cell_x,
         _p(x), x.stride(0), kx, _p(h), _p(c), _p(w_x), w_x.shape[1], _p(w_hh.detach()), _p(bias), _p(h_out), _p(c_out),
         b, hid, _stream())
    return h_out, c_out


def stem_gemm(vec, w_kn, k, scale, shift, out, *, period, act=ACT_LRELU, slope=0.2):
    """out[m][n] = act((vec[m][:k] . w_kn[:k][n]) * scale[n % period] + shift[n % period]) (dvg_stem_gemm); w_kn is the
    zero-padded transposed GEMM weight (KP,N), KP in {96, 128}."""
    _dev_f32(vec, "stem_gemm.vec")
    if vec.dim() != 2 or vec.stride(1) != 1:
        vec = vec.contiguous().view(vec.shape[0], -1)
    m = vec.shape[0]
    kp, n = w_kn.shape
    if vec.shape[1] != k or tuple(out.shape) != (m, n) or out.stride(1) != 1:
        raise RuntimeError("stem_gemm: shape mismatch")
    _run("gemm_nt", 2.0 * m * n * k, 4.0 * (m * k + n * k + m * n), lib().dvg_stem_gemm, _p(vec), vec.stride(0), _p(w_kn),
         kp, _p(scale), _p(shift), _p(out), out.stride(0), m, n, k, period, act, slope, _stream())
    return out


def stem_up_winograd_input(vec, w_kn, k, scale, shift, cout, *, act=ACT_LRELU, slope=0.2):
    """Decoder stem + BN + activation, nearest-x2 upsampling and the F(4x4,3x3) input transform of the result in one launch
    (dvg_stem_up_winograd_input): a WinoV (.up) of shape (M, cout, 8, 8) for the x half of the first decoder block's concat
    conv; the 4 x 4 map is never written.  w_kn as stem_gemm takes it."""
    _dev_f32(vec, "stem_up_winograd_input.vec")
    if vec.dim() != 2 or vec.stride(1) != 1:
        vec = vec.contiguous().view(vec.shape[0], -1)
    m = vec.shape[0]
    kp, n = w_kn.shape
    if vec.shape[1] != k or n != 16 * cout or cout % 16:
        raise RuntimeError("stem_up_winograd_input: shape mismatch")
    v = torch.empty((36, 4 * m, cout), device=vec.device, dtype=torch.float32)
    _run("stem_up_winograd_input", 2.0 * m * n * k, 4.0 * (m * k + n * k + v.numel()), lib().dvg_stem_up_winograd_input, _p(vec),
         vec.stride(0), _p(w_kn), kp, _p(scale), _p(shift), _p(v), m, cout, k, act, slope, _stream())
    return WinoV(v, (m, cout, 8, 8), up=True)


# ----------------------------------------------------------------------------------
# GP
# ----------------------------------------------------------------------------------
def gp_predict(h, z, var_mean, chol_var, mean_const, outputscale, lengthscale, *, noise=None, eps=None,
               want_var=True, want_cov=False, want_kl=False, train_mode=False, jitter=1e-3, raw_hypers=False, param_period=0,
               step_group=1):
    """h [B][D] (any strides); returns dict(mean [D][B], var, sample, cov, kl).  raw_hypers: outputscale / lengthscale /
    noise are the RAW parameters, soft-plus'ed (noise: + 1e-4 floor) inside the kernel.
    param_period = P > 0: h carries D = S x P columns - S time steps side by side - and column d uses the parameters of
    latent dim d % P (the parameter tensors have P rows).  step_group = k > 1: k consecutive steps of a latent dim are one
    workgroup's problem (dvg_hip.h; train-mode outputs only) - same outputs."""
    _dev_f32(h, "gp_predict.h")
    h = h if h.is_contiguous() else h.contiguous()
    b, d = h.shape
    m = z.shape[1]
    dev = h.device
    mean = torch.empty((d, b), device=dev, dtype=torch.float32)
    var = torch.empty((d, b), device=dev, dtype=torch.float32) if want_var else None
    sample = torch.empty((d, b), device=dev, dtype=torch.float32) if eps is not None else None
    cov = torch.empty((d, b, b), device=dev, dtype=torch.float32) if want_cov else None
    kl = torch.empty((d,), device=dev, dtype=torch.float32) if want_kl else None
    if eps is not None:
        eps = eps.contiguous()
        if tuple(eps.shape) != (d, b):
            raise RuntimeError(f"gp_predict: eps must be ({d},{b})")
    args = [t.detach().contiguous().view(-1) for t in (z, var_mean, chol_var, mean_const, outputscale, lengthscale)]
    dp = param_period or d
    if d % dp or args[0].numel() != dp * m or args[2].numel() != dp * m * m or args[3].numel() != dp:
        raise RuntimeError("gp_predict: parameter shapes do not match (D,M) / the parameter period")
    nz = None if noise is None else noise.detach().contiguous().view(-1)
    _run("gp_predict", 0.0, 4.0 * (b * d + d * m * (m + 2) + 3 * d * b), lib().dvg_gp_predict, _p(h),
         *[_p(t) for t in args], _p(nz), _p(eps), _p(mean), _p(var), _p(sample), _p(cov), _p(kl), b, d, m,
         int(train_mode) | (2 if raw_hypers else 0), jitter, int(param_period), int(step_group), _stream())
    return {"mean": mean, "var": var, "sample": sample, "cov": cov, "kl": kl}


# ----------------------------------------------------------------------------------
# backward (training) wrappers
# ----------------------------------------------------------------------------------
_LOSS_W = {}     # (device, weights) -> device tensor of per-call weights


def frame_losses(pred, target, weights):
    """pred (S, K, ...) - the K decoder calls of every step -, target (S, ...) contiguous: (sums (K,), dpred like pred) with
    sums[k] = sum over steps and elements of (pred[:, k] - target)^2 and dpred = d(sum_k weights[k] sums[k]) / d pred, one pass
    (dvg_frame_losses; train.py:227-239).  `weights`: K Python floats (loss weight / elements per call)."""
    _dev_f32(pred, "frame_losses.pred")
    _dev_f32(target, "frame_losses.target")
    if not pred.is_contiguous() or not target.is_contiguous():
        raise RuntimeError("frame_losses: contiguous operands expected")
    s_, k = pred.shape[0], pred.shape[1]
    n = target[0].numel()
    if target.shape[0] != s_ or pred[0, 0].numel() != n or n % 4 or len(weights) != k or not 1 <= k <= 3:
        raise RuntimeError(f"frame_losses: pred {tuple(pred.shape)} / target {tuple(target.shape)} / {len(weights)} weights")
    key = (pred.device, tuple(float(w) for w in weights))
    w = _LOSS_W.get(key)
    if w is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("frame_losses: first call with these weights during a capture (run one eager iteration first)")
        w = _LOSS_W[key] = torch.tensor(key[1], dtype=torch.float32, device=pred.device)
    sums = torch.empty(k, device=pred.device, dtype=torch.float32)
    dpred = torch.empty_like(pred)
    partial = torch.empty(lib().dvg_frame_losses_blocks(s_ * n) * k, device=pred.device, dtype=torch.float32)
    _run("frame_losses", 3.0 * pred.numel(), 4.0 * (2 * pred.numel() + target.numel()), lib().dvg_frame_losses, _p(pred), _p(target),
         _p(sums), _p(dpred), n, s_, k, _p(w), _p(partial), _stream())
    return sums, dpred


def mse_sum_grad(a, b, scale, need_grad=True):
    """(sum (a - b)^2 as a 0-dim tensor, 2 scale (a - b) or None) for small tensors (dvg_mse_sum_grad): a closure's latent MSE
    and its gradient (train.py:188,223)."""
    _dev_f32(a, "mse_sum_grad.a")
    _dev_f32(b, "mse_sum_grad.b")
    a, b = a.contiguous(), b.contiguous()
    if a.shape != b.shape:
        raise RuntimeError("mse_sum_grad: shape mismatch")
    out = torch.empty((), device=a.device, dtype=torch.float32)
    da = torch.empty_like(a) if need_grad else None
    check(lib().dvg_mse_sum_grad(_p(a), _p(b), _p(out), _p(da), a.numel(), float(scale), _stream()), "mse_sum_grad")
    return out, da


def gp_var_norms(var: torch.Tensor) -> torch.Tensor:
    """(B,) L2 norm over the latent dims of a predictive variance (D,B): generate_frames.py:230,275's
    `np.linalg.norm(variance.cpu().numpy().transpose(), axis=1)` without the host round trip (dvg_gp_var_norms)."""
    _dev_f32(var, "gp_var_norms.var")
    var = var.contiguous()
    d, b = var.shape
    out = torch.empty(b, device=var.device, dtype=torch.float32)
    check(lib().dvg_gp_var_norms(_p(var), _p(out), d, b, _stream()), "gp_var_norms")
    return out


def gp_trigger_step(var, col, ctx, coef, flag, values, thresholds, flags, slot) -> None:
    """One decision of GPtrigger_gen's main loop on the device (dvg_gp_trigger_step; generate_frames.py:227-232,285-289): ctx
    (window floats) slides in place, flag (1 int32) = value > threshold, logs at `slot`."""
    _dev_f32(var, "gp_trigger_step.var")
    d, b = var.shape
    if not var.is_contiguous() or ctx.dtype != torch.float32 or flag.dtype != torch.int32 or flags.dtype != torch.int32:
        raise RuntimeError("gp_trigger_step: contiguous (D,B) variance, float32 window, int32 flags expected")
    if not 0 <= slot < min(values.numel(), thresholds.numel(), flags.numel()):
        raise RuntimeError("gp_trigger_step: log slot out of range")
    check(lib().dvg_gp_trigger_step(_p(var), d, b, int(col), _p(ctx), ctx.numel(), float(coef), _p(flag), _p(values),
                                    _p(thresholds), _p(flags), int(slot), _stream()), "gp_trigger_step")


def gp_trigger_replay(values, ctx0, coef):
    """(flags int32 (n,), thresholds (n,)): the decisions another batch index would take on the recorded main-loop values
    from its own initial window (dvg_gp_trigger_replay)."""
    _dev_f32(values, "gp_trigger_replay.values")
    _dev_f32(ctx0, "gp_trigger_replay.ctx0")
    values, ctx0 = values.contiguous(), ctx0.contiguous()
    n = values.numel()
    flags = torch.empty(n, dtype=torch.int32, device=values.device)
    thr = torch.empty(n, dtype=torch.float32, device=values.device)
    check(lib().dvg_gp_trigger_replay(_p(values), n, _p(ctx0), ctx0.numel(), float(coef), _p(flags), _p(thr), _stream()),
          "gp_trigger_replay")
    return flags, thr


def gp_trigger_select(flag, sample_db, h_pred, states_old, states_new):
    """(vec (B,D), [state tensors]) of a GPtrigger_gen step (dvg_gp_trigger_select): the GP sample (D,B), transposed, and the
    OLD recurrent state when the device flag is set, the LSTM output and the NEW state otherwise (generate_frames.py:289-296)."""
    import ctypes as C
    _dev_f32(sample_db, "gp_trigger_select.sample")
    _dev_f32(h_pred, "gp_trigger_select.h_pred")
    d, b = sample_db.shape
    if tuple(h_pred.shape) != (b, d) or not sample_db.is_contiguous() or not h_pred.is_contiguous():
        raise RuntimeError("gp_trigger_select: sample (D,B) and h_pred (B,D), both contiguous, expected")
    n = len(states_old)
    if n != len(states_new) or n > 8 or any(a.shape != c.shape or not a.is_contiguous() or not c.is_contiguous()
                                             or a.numel() != states_old[0].numel() for a, c in zip(states_old, states_new)):
        raise RuntimeError("gp_trigger_select: up to 8 contiguous state tensors of one size, old and new alike")
    vec = torch.empty((b, d), device=h_pred.device, dtype=torch.float32)
    outs = [torch.empty_like(a) for a in states_old]
    arr = lambda ts: (C.c_void_p * max(1, n))(*[t.data_ptr() for t in ts])   # noqa: E731
    check(lib().dvg_gp_trigger_select(_p(flag), _p(sample_db), _p(h_pred), _p(vec), d, b, n,
                                      states_old[0].numel() if n else 0, arr(states_old), arr(states_new), arr(outs), _stream()),
          "gp_trigger_select")
    return vec, outs


def transpose2d(a: torch.Tensor) -> torch.Tensor:
    """[R][C] -> contiguous [C][R] with the LDS-tiled layout kernel (a (1,R,C) 'NCHW' -> 'NHWC' pass)."""
    _dev_f32(a, "transpose2d")
    a = a if a.is_contiguous() else a.contiguous()
    r, c = a.shape
    out = torch.empty((c, r), device=a.device, dtype=torch.float32)
    check(lib().dvg_nchw_to_nhwc(_p(a), _p(out), 1, r, c, 1, _stream()), "transpose2d")
    return out


def bn_act_bwd(dy, dyp, y, u, gamma, mean, invstd, count, *, act, slope, train=True, sinks=None, du_sum=None):
    """BatchNorm(+act, +2x2 max-pool) backward.  dy/dyp/y/u NHWC-in-memory (N,C,H,W) tensors (2-D [rows][C]
    tensors are passed as (rows,C,1,1)).  Returns (du, dgamma, dbeta, dbias); with `sinks` = (g_gamma, g_beta, g_bias or
    None) the three parameter gradients are ACCUMULATED into those buffers by the finalize kernel and None is returned
    in their place.  du_sum = (tensor like du, mode): a second output, mode 1: du_sum = du, 2: du_sum += du."""
    n, c, h, w = y.shape
    pool = dyp is not None
    # groups (time-batched training): mean / invstd are [G][C], the batch is G consecutive BatchNorm batches of n / G
    # images, `count` is per group; the per-group parameter gradients are summed over the groups by dvg_colsum
    groups = mean.shape[0] if mean.dim() == 2 else 1
    if n % groups:
        raise RuntimeError("bn_act_bwd: the groups must divide the batch")
    rows = lib().dvg_bn_act_bwd_rows(n // groups, h, w, int(pool))
    dev = y.device
    partial = torch.empty((groups * rows, 2, c), device=dev, dtype=torch.float32)
    dp = torch.empty_like(u)
    _run("bn_act_bwd_reduce", 0.0, 4.0 * 4 * y.numel(), lib().dvg_bn_act_bwd_reduce, _p(dy), _p(dyp), _p(y), _p(u),
         _p(dp), _p(partial), n, h, w, c, act, slope, groups, _stream())
    coef = torch.empty((3, groups, c), device=dev, dtype=torch.float32)
    if train and SYNC_BN is not None:
        # sync-BN (mean / invstd are statistics of the GLOBAL batch): the coefficients of du need the two per-channel sums over
        # all ranks and the global count; dgamma / dbeta / dbias stay LOCAL sums below - the gradient all-reduce averages them
        # over the ranks like every other parameter gradient (each rank's dy carries its own 1 / local-batch factor).
        glob = sync_partial_rows(partial, groups)
        check(lib().dvg_bn_bwd_finalize(_p(glob), 1, _p(gamma), _p(mean), _p(invstd), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                                        None, None, None, c, float(count) * SYNC_BN[2], 1, 0, groups, _stream()),
              "bn_bwd_finalize")
        coef_keep, coef = coef, torch.empty((3, groups, c), device=dev, dtype=torch.float32)   # the local pass's: discarded
    else:
        coef_keep = None
    if groups > 1:
        pg = torch.empty((3, groups, c), device=dev, dtype=torch.float32)       # per-group dgamma, dbeta, dbias
        check(lib().dvg_bn_bwd_finalize(_p(partial), rows, _p(gamma), _p(mean), _p(invstd), _p(coef[0]), _p(coef[1]),
                                        _p(coef[2]), _p(pg[0]), _p(pg[1]), _p(pg[2]), c, float(count), int(train), 0,
                                        groups, _stream()), "bn_bwd_finalize")
        if sinks is None:
            dgamma, dbeta, dbias = colsum(pg[0]), colsum(pg[1]), colsum(pg[2])
        else:
            dgamma = dbeta = dbias = None
            colsum(pg[0], out=sinks[0], accumulate=True)
            colsum(pg[1], out=sinks[1], accumulate=True)
            if sinks[2] is not None and not train:      # train-mode BatchNorm: d(bias) == 0 identically
                colsum(pg[2], out=sinks[2], accumulate=True)
    else:
        if sinks is None:
            dgamma = torch.empty(c, device=dev, dtype=torch.float32)
            dbeta = torch.empty(c, device=dev, dtype=torch.float32)
            dbias = torch.empty(c, device=dev, dtype=torch.float32)
            outs, acc = (dgamma, dbeta, dbias), 0
        else:
            dgamma = dbeta = dbias = None
            outs, acc = sinks, 1
        check(lib().dvg_bn_bwd_finalize(_p(partial), rows, _p(gamma), _p(mean), _p(invstd), _p(coef[0]), _p(coef[1]),
                                        _p(coef[2]), _p(outs[0]), _p(outs[1]), _p(outs[2]), c, float(count), int(train),
                                        acc, 1, _stream()), "bn_bwd_finalize")
    if coef_keep is not None:
        coef = coef_keep
    sum_t, sum_mode = (None, 0) if du_sum is None else du_sum
    if sum_t is not None and (sum_t.shape != dp.shape or sum_t.stride() != dp.stride()):
        raise RuntimeError("bn_act_bwd: du_sum must have du's shape and layout")
    _run("affine3_apply", 0.0, 4.0 * 3 * y.numel(), lib().dvg_affine3_apply, _p(dp), _p(u), _p(coef[0]), _p(coef[1]),
         _p(coef[2]), _p(dp), dp.numel(), c, _p(sum_t), int(sum_mode), groups, _stream())
    return dp, dgamma, dbeta, dbias


def conv_wgrad_partial(mode, x, skip, du, *, upsample=False):
    """The K-split partial slabs (S, taps, Cout, Cin) of a dense conv's weight gradient (dvg_conv_wgrad); finish them
    with wgrad_finish / k4_to_w3."""
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0 if skip is None else skip.shape[1]
    cout = du.shape[1]
    cin = c1 + c2
    taps = 9 if mode == MODE_CONV3 else 16
    s = lib().dvg_conv_wgrad_splits(mode, n, h, w, cin, cout)
    if s <= 0:
        raise RuntimeError(f"conv_wgrad: unsupported shape N={n} H={h} W={w} Cin={cin} Cout={cout}")
    partial = torch.empty((s, taps, cout, cin), device=x.device, dtype=torch.float32)
    _run("conv_wgrad", 2.0 * du.numel() * taps * cin / (4 if mode == MODE_CONVT4S2 else 1),
         4.0 * (x.numel() + du.numel() + partial.numel()), lib().dvg_conv_wgrad, mode, _p(x), _p(skip), _p(du),
         _p(partial), n, h, w, c1, c2, cout, int(upsample), _stream())
    return partial


def conv_wgrad_partial_multi(mode, xs, skips, dus, *, upsample=False):
    """conv_wgrad_partial over several same-shape uses of ONE layer in one launch (dvg_conv_wgrad_multi): the partial
    slabs hold sum_i dOut_i (x) In_i.  xs / dus: lists of 1..8 tensors; skips: list or None."""
    import ctypes as C
    items = len(xs)
    x, du = xs[0], dus[0]
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0 if skips is None else skips[0].shape[1]
    cout, cin = du.shape[1], c1 + c2
    taps = 9 if mode == MODE_CONV3 else 16
    for i in range(items):
        if xs[i].shape != x.shape or dus[i].shape != du.shape or not is_nhwc(xs[i]) or not is_nhwc(dus[i]) or \
                (c2 and (skips[i].shape != skips[0].shape or not is_nhwc(skips[i]))):
            raise RuntimeError("conv_wgrad_partial_multi: items must share one NHWC shape")
    s = lib().dvg_conv_wgrad_splits_multi(mode, n, h, w, cin, cout, items)
    if s <= 0:
        raise RuntimeError(f"conv_wgrad: unsupported shape N={n} H={h} W={w} Cin={cin} Cout={cout} items={items}")
    partial = torch.empty((s, taps, cout, cin), device=x.device, dtype=torch.float32)
    arr = C.c_void_p * items
    px, pd = arr(*[t.data_ptr() for t in xs]), arr(*[t.data_ptr() for t in dus])
    ps = arr(*[t.data_ptr() for t in skips]) if c2 else None
    _run("conv_wgrad", items * 2.0 * du.numel() * taps * cin / (4 if mode == MODE_CONVT4S2 else 1),
         4.0 * (items * (x.numel() + du.numel()) + partial.numel()), lib().dvg_conv_wgrad_multi, mode, items, px, ps, pd,
         _p(partial), n, h, w, c1, c2, cout, int(upsample), _stream())
    return partial


def winograd_wgrad_ok(n, cin, h, w, cout):
    """Shapes the Winograd-form weight gradient takes: 3x3 / stride 1, maps up to 32x32 in whole 4x4 tiles, channel counts
    multiples of 128 on both sides."""
    return h % 4 == 0 and w % 4 == 0 and h <= 32 and w <= 32 and cin % 128 == 0 and cout % 128 == 0 and n > 0


def winograd_wgrad_partial_multi(xs, dus, vs=None):
    """The packed (1, 9, Cout, Cin) weight-gradient slab of a 3x3 conv, sum_i dOut_i (x) In_i over 1..n same-shape uses, in
    Winograd F(4x4,3x3) form (dvg_winograd_wgrad_*): interchangeable with conv_wgrad_partial_multi(MODE_CONV3, ...).
    vs: the uses' input transforms (36, T, Cin) as the forward pass computed them (conv3x3_winograd(return_v=True)); given
    for every use (and T % 64 == 0) they are read in place instead of recomputed from xs."""
    import ctypes as C
    items = len(xs)
    x, du = xs[0], dus[0]
    n, cin, h, w = x.shape
    cout = du.shape[1]
    if not winograd_wgrad_ok(n, cin, h, w, cout) or tuple(du.shape) != (n, cout, h, w):
        raise RuntimeError(f"winograd_wgrad: unsupported shape x {tuple(x.shape)} dout {tuple(du.shape)}")
    for i in range(items):
        if xs[i].shape != x.shape or dus[i].shape != du.shape or not is_nhwc(xs[i]) or not is_nhwc(dus[i]):
            raise RuntimeError("winograd_wgrad_partial_multi: items must share one NHWC shape")
        _dev_f32(xs[i], "winograd_wgrad.x")
        _dev_f32(dus[i], "winograd_wgrad.dout")
    t = n * (h // 4) * (w // 4)
    tp = (items * t + 63) // 64 * 64
    saved = vs is not None and t % 64 == 0 and all(
        v is not None and tuple(v.shape) == (36, t, cin) and v.is_contiguous() and v.dtype == torch.float32 and v.device == x.device
        for v in vs)
    alloc = torch.empty if tp == items * t else torch.zeros      # padding rows must be zero
    v = None if saved else alloc((36, tp, cin), device=x.device, dtype=torch.float32)
    dm = alloc((36, tp, cout), device=x.device, dtype=torch.float32)
    for i in range(items):
        _run("winograd_wgrad_operands", 0.0, 4.0 * 3.25 * ((0 if saved else x.numel()) + du.numel()),
             lib().dvg_winograd_wgrad_operands, None if saved else _p(xs[i]), _p(dus[i]), _p(v), _p(dm), n, h, w, cin, cout,
             tp, i * t, _stream())
    s = lib().dvg_winograd_wgrad_splits(tp, cin, cout)
    if s <= 0:
        raise RuntimeError(f"winograd_wgrad: unsupported shape tiles={tp} Cin={cin} Cout={cout}")
    part = torch.empty((s, 36, cout, cin), device=x.device, dtype=torch.float32)
    gemm_bytes = 4.0 * (36.0 * tp * cin + dm.numel() + part.numel())
    if saved:
        pv = (C.c_void_p * items)(*[t_.data_ptr() for t_ in vs])
        _run("winograd_wgrad_gemm", 2.0 * 36 * tp * cin * cout, gemm_bytes, lib().dvg_winograd_wgrad_gemm_items, _p(dm), pv,
             items, t, _p(part), cin, cout, _stream(), alg_flops=items * 2.0 * du.numel() * 9 * cin)
    else:
        _run("winograd_wgrad_gemm", 2.0 * 36 * tp * cin * cout, gemm_bytes, lib().dvg_winograd_wgrad_gemm, _p(dm), _p(v),
             _p(part), tp, cin, cout, _stream(), alg_flops=items * 2.0 * du.numel() * 9 * cin)
    packed = torch.empty((1, 9, cout, cin), device=x.device, dtype=torch.float32)
    if s > 2:
        # many thin slabs (the 128-channel layers: one output block per position, K split ~30 ways): sum them with the wide
        # slab reduction first - the transform kernel's one thread per (co, ci) would walk 36 * S loads serially
        summed = torch.empty((36, cout, cin), device=x.device, dtype=torch.float32)
        _run("reduce_partials", 0.0, 4.0 * (part.numel() + summed.numel()), lib().dvg_reduce_partials, _p(part), _p(summed),
             s, summed.numel(), _stream())
        part, s = summed, 1
    _run("winograd_wgrad_reduce", 0.0, 4.0 * (part.numel() + packed.numel()), lib().dvg_winograd_wgrad_reduce, _p(part), s,
         _p(packed), cin, cout, _stream())
    return packed


def wgrad_finish(partial, dst, kind, kh, kw, *, ctot=None, c_lo=0, beta=0.0):
    """dst = beta * dst + sum of the partial slabs (S, kh*kw, Cout, Cin), addressed as kind 0: Conv2d weight
    (Cout, Ctot, kh, kw)[:, c_lo:c_lo+Cin]; 1: ConvTranspose2d weight (Ctot, Cout, kh, kw)[c_lo:c_lo+Cin] (flipped);
    2: plain packed (kh*kw, Cout, Cin) (dvg_wgrad_finish)."""
    s, taps, cout, cin = partial.shape
    if taps != kh * kw:
        raise RuntimeError("wgrad_finish: tap count mismatch")
    if ctot is None:
        ctot = cin
    want = (cout, ctot, kh, kw) if kind == 0 else ((ctot, cout, kh, kw) if kind == 1 else (taps, cout, cin))
    if tuple(dst.shape) != want or not dst.is_contiguous():
        raise RuntimeError(f"wgrad_finish: destination {tuple(dst.shape)} must be contiguous {want}")
    check(lib().dvg_wgrad_finish(_p(partial), s, _p(dst), kind, kh, kw, cout, cin, ctot, c_lo, float(beta), _stream()),
          "wgrad_finish")
    return dst


def k4_to_w3(dk4_packed, dw, c_lo=0, beta=0.0):
    """dw (Cout, Ctot, 3, 3)[:, c_lo:c_lo+C1] = beta * dw + 2x2 window sums of dK4 (packed (16, Cout, C1)) (dvg_k4_to_w3)."""
    t, cout, c1 = dk4_packed.shape
    if t != 16 or dw.dim() != 4 or dw.shape[0] != cout or tuple(dw.shape[2:]) != (3, 3) or not dw.is_contiguous():
        raise RuntimeError("k4_to_w3: shape mismatch")
    check(lib().dvg_k4_to_w3(_p(dk4_packed), _p(dw), cout, c1, dw.shape[1], c_lo, float(beta), _stream()), "k4_to_w3")
    return dw


def conv_wgrad(mode, x, skip, du, *, upsample=False):
    """Packed weight gradient [taps][Cout][Cin] of a dense conv (see dvg_conv_wgrad)."""
    n, c1, hx, wx = x.shape
    h, w = (hx * 2, wx * 2) if upsample else (hx, wx)
    c2 = 0 if skip is None else skip.shape[1]
    cout = du.shape[1]
    cin = c1 + c2
    taps = 9 if mode == MODE_CONV3 else 16
    s = lib().dvg_conv_wgrad_splits(mode, n, h, w, cin, cout)
    if s <= 0:
        raise RuntimeError(f"conv_wgrad: unsupported shape N={n} H={h} W={w} Cin={cin} Cout={cout}")
    partial = torch.empty((s, taps, cout, cin), device=x.device, dtype=torch.float32)
    _run("conv_wgrad", 2.0 * du.numel() * taps * cin / (4 if mode == MODE_CONVT4S2 else 1),
         4.0 * (x.numel() + du.numel() + partial.numel()), lib().dvg_conv_wgrad, mode, _p(x), _p(skip), _p(du),
         _p(partial), n, h, w, c1, c2, cout, int(upsample), _stream())
    if s == 1:
        return partial[0]
    out = torch.empty((taps, cout, cin), device=x.device, dtype=torch.float32)
    check(lib().dvg_reduce_partials(_p(partial), _p(out), s, out.numel(), _stream()), "reduce_partials")
    return out


def wgrad_thin(inp_nchw, dout_nhwc, ks, out=None, beta=0.0):
    """dW (C, nc, ks, ks) of a thin layer (see dvg_wgrad_thin); `out` (contiguous, same shape): out = beta * out + dW."""
    inp = inp_nchw if inp_nchw.is_contiguous() else inp_nchw.contiguous()
    n, nc, hi, wi = inp.shape
    c = dout_nhwc.shape[1]
    assert is_nhwc(dout_nhwc)
    rows = lib().dvg_wgrad_thin_rows(ks, n, hi, wi)
    partial = torch.empty((rows, c, nc * ks * ks), device=inp.device, dtype=torch.float32)
    _run("wgrad_thin", 2.0 * dout_nhwc.numel() * nc * ks * ks, 4.0 * (inp.numel() + dout_nhwc.numel()),
         lib().dvg_wgrad_thin, _p(inp), _p(dout_nhwc), _p(partial), ks, n, hi, wi, nc, c, _stream())
    if out is None:
        out = torch.empty((c, nc, ks, ks), device=inp.device, dtype=torch.float32)
        beta = 0.0
    elif tuple(out.shape) != (c, nc, ks, ks) or not out.is_contiguous():
        raise RuntimeError("wgrad_thin: bad destination")
    n = out.numel()
    check(lib().dvg_wgrad_finish(_p(partial), rows, _p(out), 2, 1, 1, 1, n, n, 0, float(beta), _stream()),
          "wgrad_finish(thin)")
    return out


def act_bwd(dy, y, act, slope=0.0):
    dy = dy if dy.is_contiguous() else dy.contiguous()
    y = y if y.is_contiguous() else y.contiguous()
    out = torch.empty_like(y)
    check(lib().dvg_act_bwd(_p(dy), _p(y), _p(out), y.numel(), act, slope, _stream()), "act_bwd")
    return out


def upsample2x_bwd(dxu):
    assert is_nhwc(dxu)
    n, c, h2, w2 = dxu.shape
    dx = nhwc_empty(n, c, h2 // 2, w2 // 2, dxu.device)
    check(lib().dvg_upsample2x_bwd(_p(dxu), _p(dx), n, h2 // 2, w2 // 2, c, _stream()), "upsample2x_bwd")
    return dx


def colsum(a, out=None, accumulate=False):
    a = a if a.is_contiguous() else a.contiguous()
    rows, c = a.shape
    if out is None:
        out, accumulate = torch.empty(c, device=a.device, dtype=torch.float32), False
    check(lib().dvg_colsum(_p(a), _p(out), rows, c, int(accumulate), _stream()), "colsum")
    return out


GEMM_TN_MAX_ROWS = 512     # above: two transposes + the NT kernel (its K = rows spread over lanes and split-K suit long sums)


def gemm_tn(a, b, out=None, accumulate=False, colsums=(), colsum_accumulate=True):
    """out (M, N) (+)= a^T b for a (R, M), b (R, N) - a dense layer's weight gradient dY^T X over the rows of a BPTT pass - in one
    launch (dvg_gemm_tn); `colsums`: up to two (M,) buffers that receive (colsum_accumulate: are added) the column sums of `a`
    (the bias gradients).  Returns out."""
    _dev_f32(a, "gemm_tn.a")
    _dev_f32(b, "gemm_tn.b")
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise RuntimeError(f"gemm_tn: a {tuple(a.shape)} and b {tuple(b.shape)} must share their rows")
    a = a if a.stride(1) == 1 else a.contiguous()
    b = b if b.stride(1) == 1 else b.contiguous()
    r, m = a.shape
    n = b.shape[1]
    colsums = [c for c in colsums if c is not None]
    if len(colsums) > 2 or any(c.numel() != m or not c.is_contiguous() for c in colsums):
        raise RuntimeError("gemm_tn: at most two contiguous column-sum buffers of a.shape[1] entries")
    if out is None:
        out, accumulate = torch.empty((m, n), device=a.device, dtype=torch.float32), False
    elif tuple(out.shape) != (m, n) or out.stride(1) != 1:
        raise RuntimeError(f"gemm_tn: out must be ({m},{n}) with unit column stride")
    if r > GEMM_TN_MAX_ROWS:
        gemm_nt(transpose2d(a), transpose2d(b), None, None, out=out, accumulate=accumulate)
        for c in colsums:
            colsum(a, out=c, accumulate=colsum_accumulate)
        return out
    _run("gemm_tn", 2.0 * r * m * n, 4.0 * (r * m + r * n + m * n), lib().dvg_gemm_tn, _p(a), _p(b), _p(out),
         _p(colsums[0]) if colsums else None, _p(colsums[1]) if len(colsums) > 1 else None, r, m, n, a.stride(0), b.stride(0),
         out.stride(0), int(accumulate), int(colsum_accumulate), _stream())
    return out


def lstm_gates_bwd(dh, dc, gates, c_prev, c_new):
    b, hid = c_prev.shape
    dh = None if dh is None else (dh if dh.is_contiguous() else dh.contiguous())
    dc = None if dc is None else (dc if dc.is_contiguous() else dc.contiguous())
    dG = torch.empty((b, 4 * hid), device=c_prev.device, dtype=torch.float32)
    dcp = torch.empty_like(c_prev)
    check(lib().dvg_lstm_gates_bwd(_p(dh), _p(dc), _p(gates), _p(c_prev), _p(c_new), _p(dG), _p(dcp), b, hid,
                                   _stream()), "lstm_gates_bwd")
    return dG, dcp


def gp_elbo(mean, var, kl, target, raw_noise, num_data, noise_period=0):
    """VariationalELBO(combine_terms=True) with the Gaussian likelihood's expected log-probability -> (D,) (dvg_gp_elbo).
    mean, var (D,B) contiguous, kl (D,), target (D,B) with any strides, raw_noise (D,) or (D,1) - (P,) with noise_period = P:
    row d uses raw_noise[d % P]."""
    for t, n in ((mean, "mean"), (var, "var"), (kl, "kl"), (target, "target"), (raw_noise, "raw_noise")):
        _dev_f32(t, "gp_elbo." + n)
    d, b = mean.shape
    if tuple(var.shape) != (d, b) or tuple(target.shape) != (d, b) or kl.numel() != d or raw_noise.numel() != (noise_period or d) \
            or d % (noise_period or d):
        raise RuntimeError(f"gp_elbo: shapes mean {tuple(mean.shape)} var {tuple(var.shape)} target {tuple(target.shape)}")
    mean, var, kl, raw = mean.contiguous(), var.contiguous(), kl.contiguous(), raw_noise.reshape(-1).contiguous()
    out = torch.empty(d, device=mean.device, dtype=torch.float32)
    check(lib().dvg_gp_elbo(_p(mean), _p(var), _p(kl), _p(target), target.stride(0), target.stride(1), _p(raw), _p(out),
                            b, d, int(num_data), int(noise_period), _stream()), "gp_elbo")
    return out


def gp_elbo_bwd(mean, var, kl, target, raw_noise, gelbo, num_data, need_gtarget=True, noise_period=0):
    """Gradients of gp_elbo w.r.t. mean, var (D,B), kl (D,), target (D,B; None unless asked for), raw_noise (D,: one entry per
    ROW also with a noise period - the caller sums the steps)."""
    d, b = mean.shape
    mean, var, kl, raw = mean.contiguous(), var.contiguous(), kl.contiguous(), raw_noise.reshape(-1).contiguous()
    gelbo = gelbo.contiguous()
    dev = mean.device
    gmean, gvar = torch.empty((d, b), device=dev), torch.empty((d, b), device=dev)
    gkl, graw = torch.empty(d, device=dev), torch.empty(d, device=dev)
    gtarget = torch.empty((d, b), device=dev) if need_gtarget else None
    check(lib().dvg_gp_elbo_bwd(_p(mean), _p(var), _p(kl), _p(target), target.stride(0), target.stride(1), _p(raw),
                                _p(gelbo), _p(gmean), _p(gvar), _p(gkl), _p(gtarget), _p(graw), b, d, int(num_data),
                                int(noise_period), _stream()), "gp_elbo_bwd")
    return gmean, gvar, gkl, gtarget, graw


def sum_steps(tensors, steps):
    """[t.view(steps, -1).sum(0) for t in tensors] as ONE launch (dvg_sum_steps_multi; up to 8 tensors): (steps * n_k,) -> (n_k,)."""
    import ctypes as C
    if not 1 <= len(tensors) <= 8:
        raise RuntimeError("sum_steps: 1..8 tensors")
    src = [t.contiguous() for t in tensors]
    for t in src:
        _dev_f32(t, "sum_steps")
        if t.numel() % steps:
            raise RuntimeError("sum_steps: tensor size is not a multiple of the step count")
    dst = [torch.empty(t.numel() // steps, device=t.device, dtype=torch.float32) for t in src]
    k = len(src)
    check(lib().dvg_sum_steps_multi((C.c_void_p * k)(*[t.data_ptr() for t in src]), (C.c_void_p * k)(*[t.data_ptr() for t in dst]),
                                    (C.c_long * k)(*[t.numel() for t in dst]), k, int(steps), _stream()), "sum_steps")
    return dst


def gp_step_group(b, steps, period, m):
    """Steps per workgroup of a time-batched train-mode GP call (dvg_gp_step_group; 1 = one workgroup per (step, dim))."""
    return lib().dvg_gp_step_group(int(b), int(steps), int(period), int(m))


def gp_train_bwd(h, z, m, ls, c, s, ell, gmean, gvar, gkl, jitter=1e-3, param_period=0, step_group=1):
    """Gradients of the train-mode GP prediction (see dvg_gp_train_bwd); with param_period = P the parameter gradients come
    back per WORKGROUP: G x P rows, G = ceil(S / step_group) groups of the D = S x P columns of h (out["groups"] = G;
    sum_steps adds the G copies up)."""
    h = h if h.is_contiguous() else h.contiguous()
    b, d = h.shape
    mm = z.shape[1]
    dev = h.device
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)  # noqa: E731
    period = param_period or d
    k = max(int(step_group), 1)
    groups = -(-(d // period) // k)
    r = groups * period
    out = {"dh": f(b, d), "dz": f(r, mm), "dm": f(r, mm), "dls": f(r, mm, mm), "dc": f(r), "ds": f(r), "dell": f(r),
           "groups": groups}
    args = [t.detach().contiguous().view(-1) for t in (z, m, ls, c, s, ell)]
    g = [None if t is None else t.contiguous() for t in (gmean, gvar, gkl)]
    check(lib().dvg_gp_train_bwd(_p(h), *[_p(t) for t in args], *[_p(t) for t in g], _p(out["dh"]), _p(out["dz"]),
                                 _p(out["dm"]), _p(out["dls"]), _p(out["dc"]), _p(out["ds"]), _p(out["dell"]), b, d,
                                 mm, jitter, int(param_period), k, _stream()), "gp_train_bwd")
    return out


def pixel_proj(x, wm):
    """d[px][t] = sum_c x[px][c] wm[t][c] over the pixels of an NHWC-in-memory activation (dvg_pixel_proj)."""
    assert is_nhwc(x)
    n, c, h, wd = x.shape
    t = wm.shape[0]
    d = torch.empty((n * h * wd, t), device=x.device, dtype=torch.float32)
    _run("pixel_proj", 2.0 * n * h * wd * c * t, 4.0 * (x.numel() + d.numel()), lib().dvg_pixel_proj, _p(x), _p(wm),
         _p(d), n * h * wd, c, t, _stream())
    return d


_SKIP_PROJ_CACHE = {}   # id(skip) -> (weakref(skip), skip._version, id(w), w._version, d2)


def clear_skip_proj_cache():
    """Drop cached skip projections (rollout.GraphedRollout calls this around a capture: buffers allocated while
    capturing belong to the graph's pool and must not leak into eager calls, nor the reverse)."""
    _SKIP_PROJ_CACHE.clear()


def _cached_skip_proj(skip, wm_fn, w):
    """The skip tensor of a rollout is frozen after the conditioning frames (generate_frames.py:154-157), so its
    share of the last layer's projection is computed once and reused while (tensor identity, version, weight
    version) stay the same.  Inference only (callers route training through autograd)."""
    import weakref
    key = id(skip)
    ent = _SKIP_PROJ_CACHE.get(key)
    if ent is not None and ent[0]() is skip and ent[1] == skip._version and ent[2] == id(w) and ent[3] == w._version:
        return ent[4]
    d2 = pixel_proj(skip, wm_fn())
    if len(_SKIP_PROJ_CACHE) > 8:
        _SKIP_PROJ_CACHE.clear()
    _SKIP_PROJ_CACHE[key] = (weakref.ref(skip), skip._version, id(w), w._version, d2)
    return d2


def _last_wmat(wpart, t):
    return wpart.permute(2, 3, 1, 0).reshape(t, wpart.shape[0]).contiguous()          # [(kh,kw,co)][ci]


_WMAT_CACHE = {}   # (id(w), lo, hi) -> (weakref(w), version, data_ptr, matrix)


def _last_wmat_cached(w, lo, hi, t):
    """[(kh,kw,co)][ci] projection matrix of rows lo:hi of the last layer's ConvTranspose2d weight, per weight version (it
    used to be re-permuted and copied on every decoder call)."""
    import weakref
    key = (id(w), lo, hi)
    hit = _WMAT_CACHE.get(key)
    if hit is not None and hit[0]() is w and hit[1] == w._version and hit[2] == w.data_ptr():
        return hit[3]
    m = _last_wmat(w.detach()[lo:hi], t)
    if len(_WMAT_CACHE) > 64:
        _WMAT_CACHE.clear()
    _WMAT_CACHE[key] = (weakref.ref(w), w._version, w.data_ptr(), m)
    return m


def precompute_skip_proj(skip, w, ks: int) -> None:
    """The frozen skip tensor's share of the last layer's per-pixel projection, computed ahead of the first decoder call
    (rollout.condition(), second stream); convT_last_two_step then finds it in the cache."""
    wdet = w.detach()
    c1 = wdet.shape[0] - skip.shape[1]
    t = ks * ks * wdet.shape[1]
    _cached_skip_proj(skip, lambda: _last_wmat_cached(w, c1, wdet.shape[0], t), w)


def convT_last_two_step(x, skip, w, bias, nc, ks, *, act):
    """Last layer as per-pixel projection (dvg_pixel_proj: reads the activation once) + shifted sum
    (dvg_convT_gather).  x / skip NHWC-in-memory; w the original ConvTranspose2d weight (Cin,nc,ks,ks); returns
    NCHW frames."""
    assert is_nhwc(x)
    n, c1, h, wd = x.shape
    wdet = w.detach()
    t = ks * ks * nc
    d1 = pixel_proj(x, _last_wmat_cached(w, 0, c1, t))
    d2 = None
    d2_map, d2_blk = None, 0
    if isinstance(skip, SharedBlocks):      # time-batched decoder calls: the projection of the DISTINCT skip blocks only
        d2 = pixel_proj(skip.t, _last_wmat_cached(w, c1, wdet.shape[0], t))
        d2_map, d2_blk = skip.map_dev, skip.block
        if n != skip.groups * skip.block:
            raise RuntimeError("convT_last_two_step: shared skip does not match the batch")
    elif skip is not None:
        d2 = _cached_skip_proj(skip, lambda: _last_wmat_cached(w, c1, wdet.shape[0], t), w)
    s = 2 if ks == 4 else 1
    y = torch.empty((n, nc, s * h, s * wd), device=x.device, dtype=torch.float32)
    _run("convT_gather", 0.0, 4.0 * (d1.numel() * (2 if skip is not None else 1) + y.numel()), lib().dvg_convT_gather,
         _p(d1), _p(d2), _p(bias), _p(y), ks, n, h, wd, nc, act, _p(d2_map), d2_blk, _stream())
    return y


def eval_frames(gt, pred):
    """(ssim, psnr), each (B,), of a predicted NCHW frame batch against the ground truth: per-channel skimage-style
    metrics (dvg_eval_frames) averaged over channels as utils.eval_seq does (utils.py:227-232)."""
    _dev_f32(gt, "eval_frames.gt")
    _dev_f32(pred, "eval_frames.pred")
    if gt.shape != pred.shape or gt.dim() != 4:
        raise RuntimeError(f"eval_frames: shapes {tuple(gt.shape)} vs {tuple(pred.shape)}")
    gt = gt if gt.is_contiguous() else gt.contiguous()
    pred = pred if pred.is_contiguous() else pred.contiguous()
    b, c, h, w = gt.shape
    out = torch.empty((2, b, c), device=gt.device, dtype=torch.float32)
    _run("eval_frames", 0.0, 8.0 * gt.numel(), lib().dvg_eval_frames, _p(gt), _p(pred), _p(out[0]), _p(out[1]), b * c, h,
         w, _stream())
    return out[0].mean(1), out[1].mean(1)


def moving_mnist_compose(sprites, ids, pos, seq_len, image_size):
    """(T,B,1,S,S) frames from sprites (N,D,D), ids (B,ND) int32 and pos (B,ND,T,2) int32 (dvg_moving_mnist_compose)."""
    _dev_f32(sprites, "moving_mnist_compose.sprites")
    if ids.dtype != torch.int32 or pos.dtype != torch.int32 or not ids.is_cuda or not pos.is_cuda:
        raise RuntimeError("moving_mnist_compose: ids / pos must be int32 device tensors")
    ids, pos, sprites = ids.contiguous(), pos.contiguous(), sprites.contiguous()
    b, nd = ids.shape
    if tuple(pos.shape) != (b, nd, seq_len, 2):
        raise RuntimeError(f"moving_mnist_compose: pos shape {tuple(pos.shape)}")
    out = torch.empty((seq_len, b, 1, image_size, image_size), device=sprites.device, dtype=torch.float32)
    check(lib().dvg_moving_mnist_compose(_p(sprites), _p(ids), _p(pos), _p(out), sprites.shape[0], seq_len, b, nd,
                                         image_size, sprites.shape[1], _stream()), "dvg_moving_mnist_compose")
    return out
