"""Train-mode BatchNorm wrappers: per-channel partial statistics, finalize (+ running statistics), apply, backward; the
cross-rank reduction of synchronised BatchNorm (train.py --sync_bn)."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import ACT_LRELU, _dev_f32, _p, _run, _stream, is_nhwc, nhwc_empty
from .dense import colsum


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


SYNC_BN = None    # (torch.distributed, group, world) while synchronised BatchNorm is on; re-bound only through set_sync_bn_state


def set_sync_bn_state(state) -> None:
    """fused.set_sync_bn: the (torch.distributed, process group, world size) of synchronised BatchNorm, or None."""
    global SYNC_BN
    SYNC_BN = state


def sync_bn_state():
    return SYNC_BN


def sync_partial_rows(partial: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """[groups * r][2][C] per-tile partial sums of this rank -> [groups * 2][2][C]: per group TWO fp32 rows whose sum is the
    fp64 total over all rows of ALL ranks (sync-BN).  The rows of a group are added up in fp64 (what dvg_bn_finalize /
    dvg_bn_bwd_finalize themselves do with them) and all-reduced in fp64; the total then travels as hi = fp32(total) and
    lo = fp32(total - hi), which the finalize kernels - fp64 accumulators over fp32 rows - put back together to ~1e-15.
    (One fp32 row would round the total before the variance E[x^2] - mean^2 and the backward's sum dp (u - mean) take their
    differences; the single-process path never does: its finalize kernels accumulate the fp32 partial rows in fp64.)"""
    dist, group, _ = SYNC_BN
    rows, two, c = partial.shape
    if rows % groups:
        raise RuntimeError(f"sync_partial_rows: {rows} partial rows do not split into {groups} groups")
    tot = partial.view(groups, rows // groups, two, c).sum(1, dtype=torch.float64)
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(tot, group=group)        # RCCL: on its own stream, ordered after / before the current one
    else:
        # gloo (CPU tests, the one-GPU rehearsal): an explicit, synchronous host round trip instead of gloo's CUDA-tensor path
        # (worker threads and side streams of its own) - one suspect less in profiles/r06_dp_race_bisect.txt (it was not the cause)
        host = tot.cpu()
        dist.all_reduce(host, group=group)
        tot = host.to(partial.device)
    hi = tot.to(torch.float32)
    lo = (tot - hi.to(torch.float64)).to(torch.float32)
    out = torch.stack([hi, lo], 1).reshape(groups * 2, two, c)
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
        check(lib().dvg_bn_bwd_finalize(_p(glob), 2, _p(gamma), _p(mean), _p(invstd), _p(coef[0]), _p(coef[1]), _p(coef[2]),
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
