"""Implicit-GEMM convolution wrappers (conv_igemm2.hip): weight packing, the 3x3 / 4x4 stride-2 / transposed 4x4 stride-2
blocks with fused BN + activation (+ pool, + statistics), split-K workspaces, shared skip blocks."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import ACT_LRELU, MODE_CONV3, MODE_CONV4S2, MODE_CONVT4S2, _dev_f32, _p, _run, _stream, is_nhwc, nhwc_empty


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
        runs = -(-n // per)
        per = min(per, -(-(-(-n // runs)) // 8) * 8)      # runs of about equal size (multiples of 8 images): no tiny tail launch
        sts = []
        for lo in range(0, n, per):
            r = conv4x4s2(x[lo:lo + per], wp, scale, shift, act=act, slope=slope, stats=stats)
            y[lo:lo + per].copy_(r[0] if stats else r)
            if stats:
                sts.append(r[1])
        if not stats:
            return y
        st = torch.cat(sts)
        # rows of a run that split K are its finish kernel's blocks, which do not respect image boundaries (_stats_buf): the
        # concatenation is then still the batch's partial sums, but not cuttable into per-group runs
        ti = {t.tile_images for t in sts}
        st.tile_images = ti.pop() if len(ti) == 1 else 0
        return y, st
    ws = _splitk_ws(MODE_CONV4S2, n, h, w, cin, cout, y.numel(), x.device)
    if stats:
        rows = lib().dvg_conv_stats_rows_v2(MODE_CONV4S2, n, h, w, cin, cout, 0, int(ws is not None))
    st = _stats_buf(rows, cout, x.device, _tile_images(MODE_CONV4S2, n, h, w, cin, cout, h // 2, w // 2, ws)) if stats else None
    fl, by = 2.0 * n * (h // 2) * (w // 2) * cout * 16 * cin, 4.0 * (x.numel() + y.numel() + wp.numel())
    _run("conv4x4s2_igemm", fl, by, lib().dvg_conv4x4s2_bn_act_v2, _p(x), _p(wp), _p(scale), _p(shift), _p(y), _p(st),
         n, h, w, cin, cout, act, slope, _p(ws), 0 if ws is None else ws.numel(), _stream())
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
