"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the HOST-side pipeline of the reference's hot path:
the ATen op sequences of Renderer::Render (Renderer/Renderer.cpp:105-208), SHShader::Query
(Shader/SHShader.cpp:23-29), Hash3DAnchored::AnchoredQuery (Field/Hash3DAnchored.cpp:84-99), the autograd
backward chain those ops imply, the losses of ExpRunner::Train (ExpRunner.cpp:93-118) and LibTorch's Adam.
Kernels are the C oracle's (oracle/capi.py); everything element-wise is float32 numpy in the reference's op
order.  "parity unpinned" for anything that goes through the tcnn MLP restatement (see f2n_oracle.c header).
"""
import numpy as np

from . import capi

F32 = np.float32
EPS_RGB = F32(1e-3)


def _exp(x):
    return np.exp(x.astype(F32)).astype(F32)


# ---------------------------------------------------------------------------------------------------
# Field
# ---------------------------------------------------------------------------------------------------
class HashGrid:
    """State of Hash3DAnchored (Field/Hash3DAnchored.cpp:19-82) as plain arrays."""

    def __init__(self, table_f32, prim_pool, bias_pool, n_volumes, log2_table_size):
        local = ((1 << log2_table_size) >> 4) << 4
        self.table_f32 = np.ascontiguousarray(table_f32, F32)           # [pool, 2] fp32 master
        self.prim_pool = np.ascontiguousarray(prim_pool, np.int32)
        self.bias_pool = np.ascontiguousarray(bias_pool, F32)
        self.n_volumes = int(n_volumes)
        self.local_size = np.full(16, local, np.int32)
        self.local_idx = (np.arange(16) * local).astype(np.int32)        # cumsum - local_size, in HALVES
        self.scales = capi.level_scales()

    @property
    def table_h(self):  # Hash3DAnchored.cu:186: the whole table is cast to fp16 on every call
        return capi.f2h(self.table_f32.reshape(-1))


def field_fwd(grid, mlp_params, pts_warped, volume_idx, want_ctx=False, d_hidden=64, n_hidden=1):
    """AnchoredQuery: (p+1)/2 -> hash -> fp32 -> tcnn MLP (32->64->16 in the shipped configs) -> fp32 [n,16]."""
    q01 = ((np.asarray(pts_warped, F32) + F32(1.)) * F32(.5)).astype(F32)
    x_h = capi.hash_fwd(grid.table_h, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, q01,
                        volume_idx, grid.n_volumes, grid.scales)
    x = capi.h2f(x_h)
    out_h, acts = capi.mlp_fwd(mlp_params, x, d_hidden, n_hidden, want_acts=True)
    feat = capi.h2f(out_h)
    if want_ctx:
        return feat, dict(q01=q01, x=x, x_h=x_h, acts=acts, vol=np.ascontiguousarray(volume_idx, np.int32), shape=(d_hidden, n_hidden))
    return feat


def field_bwd(grid, mlp_params, ctx, dfeat, loss_scale=128.0, fp32_accumulate=True):
    """Returns (dparams fp32 unscaled, grad_table in the table's addressing: fp32 if fp32_accumulate else the
    order-dependent fp16 accumulation), both as TRUE (unscaled) gradients."""
    d_hidden, n_hidden = ctx.get("shape", (64, 1))
    dparams, _dx, dx_scaled_h = capi.mlp_bwd(mlp_params, ctx["x"], ctx["acts"], dfeat, d_hidden, n_hidden, loss_scale)
    pool_halves = grid.table_f32.size
    g = capi.hash_bwd(pool_halves, grid.prim_pool, grid.local_idx, grid.local_size, grid.bias_pool, ctx["q01"],
                      ctx["vol"], grid.n_volumes, dx_scaled_h, grid.scales, fp32_accumulate=fp32_accumulate)
    if not fp32_accumulate:
        g = capi.h2f(g)
    return dparams, (g / F32(loss_scale)).astype(F32), dx_scaled_h


# ---------------------------------------------------------------------------------------------------
# Shader
# ---------------------------------------------------------------------------------------------------
def shade_input(feat, dirs, app_emb=None, sample_emb_idx=None, degree=4):
    feat = np.asarray(feat, F32)
    shading = np.concatenate([np.ones_like(feat[:, :1]), feat[:, 1:]], 1)          # Renderer.cpp:181-182
    if app_emb is not None:
        shading = capi.scatter_add(app_emb, sample_emb_idx, shading)                # Scatter.cu:10-18
    return np.concatenate([shading, capi.sh_encode(dirs, degree)], -1).astype(F32)  # SHShader.cpp:24-25


def shade_fwd(color_params, feat, dirs, app_emb=None, sample_emb_idx=None, want_ctx=False, d_hidden=64, n_hidden=2, degree=4):
    x = shade_input(feat, dirs, app_emb, sample_emb_idx, degree)
    out_h, acts = capi.mlp_fwd(color_params, x, d_hidden, n_hidden, want_acts=True)
    o = capi.h2f(out_h)[:, :3]
    rgb = ((F32(1.) + F32(2.) * EPS_RGB) / (F32(1.) + _exp(-o)) - EPS_RGB).astype(F32)  # SHShader.cpp:27-28
    if want_ctx:
        return rgb, dict(x=x, acts=acts, o=o, shape=(d_hidden, n_hidden))
    return rgb


def shade_bwd(color_params, ctx, drgb, n_emb=0, sample_emb_idx=None, loss_scale=128.0):
    o = ctx["o"]
    with np.errstate(over="ignore", invalid="ignore"):
        e = _exp(-o)
        dsig = ((F32(1.) + F32(2.) * EPS_RGB) * e / ((F32(1.) + e) * (F32(1.) + e))).astype(F32)
    # e = +inf (o below ~-88.7): inf / inf where the derivative's limit is 0.  The reference's autograd lets the NaN through and
    # tcnn drops the step (TCNNWP.cpp:234-240); product, taped path and this restatement take the limit.
    dsig = np.where(np.isinf(e), F32(0.), dsig).astype(F32)
    do = (np.asarray(drgb, F32) * dsig).astype(F32)
    dy = np.zeros((o.shape[0], 16), F32)
    dy[:, :3] = do
    d_hidden, n_hidden = ctx.get("shape", (64, 2))
    dparams, dx, _ = capi.mlp_bwd(color_params, ctx["x"], ctx["acts"], dy, d_hidden, n_hidden, loss_scale)
    dshading = np.ascontiguousarray(dx[:, :16])
    dfeat = dshading.copy()
    dfeat[:, 0] = 0                                       # the constant-1 column has no gradient to feat
    demb = capi.scatter_add_bwd(n_emb, sample_emb_idx, dshading) if n_emb else None
    return dparams, dfeat, demb


# ---------------------------------------------------------------------------------------------------
# Renderer
# ---------------------------------------------------------------------------------------------------
def early_stop(f0, dt, se):
    """Renderer.cpp:115-137.  Returns weights, alphas, mask(int32), new bounds."""
    density = _exp(np.asarray(f0, F32) - F32(3.))
    sec = (density * np.asarray(dt, F32)).astype(F32)
    alphas = (F32(1.) - _exp(-sec)).astype(F32)
    acc = capi.flex_acc(sec, se, False)
    trans = _exp(-acc)
    weights = (trans * alphas).astype(F32)
    mask = (trans > F32(1e-4)).astype(np.int32)
    return weights, alphas, mask, capi.filter_idx_bounds(se, mask)


def compact(mask, *arrays):
    idx = np.nonzero(mask)[0]
    return [np.ascontiguousarray(a[idx]) for a in arrays]


def composite_fwd(feat, dt, t, rgb, bg, se, want_ctx=False):
    """Renderer.cpp:180-208 (training path without gradient scaling in the forward)."""
    density = _exp(np.asarray(feat, F32)[:, 0] - F32(3.))
    tt = (np.asarray(t, F32) + F32(1e-2)).astype(F32)
    sec = (density * np.asarray(dt, F32)).astype(F32)
    alphas = (F32(1.) - _exp(-sec)).astype(F32)
    acc = capi.flex_acc(sec, se, False)
    trans = _exp(-acc)
    weights = (trans * alphas).astype(F32)
    last_trans = _exp(-capi.flex_sum(sec, se))
    colors = capi.flex_sum((weights[:, None] * np.asarray(rgb, F32)).astype(F32), se)
    colors = (colors + last_trans[:, None] * np.asarray(bg, F32)).astype(F32)
    disparity = capi.flex_sum((weights / tt).astype(F32), se)
    dsum = capi.flex_sum((weights * tt).astype(F32), se)
    depth = (dsum / (F32(1.) - last_trans + F32(1e-4))).astype(F32)
    out = dict(colors=colors, disparity=disparity, depth=depth, weights=weights)
    if want_ctx:
        out["ctx"] = dict(density=density, tt=tt, sec=sec, alphas=alphas, trans=trans, last_trans=last_trans, dsum=dsum,
                          x=(np.asarray(feat, F32)[:, 0] - F32(3.)).astype(F32))
    return out


def composite_bwd(ctx, dt, rgb, bg, se, dcolors=None, ddisparity=None, ddepth=None, dweights=None, gs_progress=1.0):
    """Autograd chain of the above, op by op (FlexOps/TruncExp/GradientScaling backward kernels).
    Returns (drgb [M,3], df0 [M])."""
    n = ctx["sec"].shape[0]
    R = se.shape[0]
    z1, z3 = np.zeros(R, F32), np.zeros((R, 3), F32)
    dcolors = z3 if dcolors is None else np.asarray(dcolors, F32)
    ddisparity = z1 if ddisparity is None else np.asarray(ddisparity, F32)
    ddepth = z1 if ddepth is None else np.asarray(ddepth, F32)
    w = (ctx["trans"] * ctx["alphas"]).astype(F32)
    rgb, bg, dt = np.asarray(rgb, F32), np.asarray(bg, F32), np.asarray(dt, F32)
    d_wc = capi.flex_sum_bwd(dcolors, se, n)                       # colors = Sum(w*c) + last*bg
    dw = (d_wc * rgb).sum(-1).astype(F32)
    drgb = (d_wc * w[:, None]).astype(F32)
    d_last = (dcolors * bg).sum(-1).astype(F32)
    dw = dw + capi.flex_sum_bwd(ddisparity, se, n) / ctx["tt"]     # disparity = Sum(w / t')
    den = (F32(1.) - ctx["last_trans"] + F32(1e-4)).astype(F32)     # depth = Sum(w*t') / den
    dw = dw + capi.flex_sum_bwd((ddepth / den).astype(F32), se, n) * ctx["tt"]
    d_last = d_last + ddepth * ctx["dsum"] / (den * den)
    if dweights is not None:
        dw = dw + np.asarray(dweights, F32)
    dw = dw.astype(F32)
    dsec = capi.flex_sum_bwd((-ctx["last_trans"] * d_last).astype(F32), se, n)   # last = exp(-Sum(sec))
    dtrans = (dw * ctx["alphas"]).astype(F32)
    dalphas = (dw * ctx["trans"]).astype(F32)
    dacc = (-ctx["trans"] * dtrans).astype(F32)
    dsec = dsec + capi.flex_acc_bwd(dacc, se, False)
    dsec = (dsec + dalphas * _exp(-ctx["sec"])).astype(F32)
    ddensity = (dsec * dt).astype(F32)
    if gs_progress < 1.0:                                          # Renderer.cpp:190-195
        ddensity = capi.grad_scaling_bwd(ddensity, se, gs_progress)
        drgb = capi.grad_scaling_bwd(drgb, se, gs_progress)
    df0 = (ddensity * _exp(np.clip(ctx["x"], F32(-100.), F32(5.)))).astype(F32)   # TruncExp backward
    return drgb, df0


# ---------------------------------------------------------------------------------------------------
# Losses (ExpRunner.cpp:93-118) and their gradients w.r.t. the RenderResult tensors
# ---------------------------------------------------------------------------------------------------
def losses_and_grads(colors, gt, disparity, weights, se, edge_feats, var_loss_weight, disp_loss_weight, tv_loss_weight):
    colors, gt = np.asarray(colors, F32), np.asarray(gt, F32)
    R = colors.shape[0]
    diff = colors - gt
    ch = np.sqrt(diff * diff + F32(1e-4)).astype(F32)
    color_loss = ch.mean(dtype=np.float64)
    dcolors = (diff / ch / F32(diff.size)).astype(F32)
    disp_loss = (disparity.astype(np.float64) ** 2).mean()
    ddisp = (F32(2.) * disparity / F32(R) * F32(disp_loss_weight)).astype(F32)
    var = capi.weight_var(weights, se)
    sv = np.sqrt(var + F32(1e-2)).astype(F32)
    var_loss = sv.mean(dtype=np.float64)
    dvar = (F32(var_loss_weight) * F32(.5) / sv / F32(R)).astype(F32)
    dweights = capi.weight_var_bwd(weights, se, dvar)
    d = edge_feats[:, 0] - edge_feats[:, 1]
    tv_loss = (d.astype(np.float64) ** 2).mean()
    dedge = np.zeros_like(edge_feats)
    dedge[:, 0] = F32(tv_loss_weight) * F32(2.) * d / F32(d.size)
    dedge[:, 1] = -dedge[:, 0]
    loss = color_loss + var_loss * var_loss_weight + disp_loss * disp_loss_weight + tv_loss * tv_loss_weight
    return dict(loss=loss, color_loss=color_loss, dcolors=dcolors, ddisparity=ddisp, dweights=dweights, dedge=dedge)


# ---------------------------------------------------------------------------------------------------
# LibTorch Adam::step (torch/csrc/api/src/optim/adam.cpp of the 1.13 line, as driven by ExpRunner)
# ---------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.99, eps=1e-15, weight_decay=0.0):
    p, g, m, v = [np.asarray(a, F32).copy() for a in (p, g, m, v)]
    if weight_decay != 0:
        g = (g + F32(weight_decay) * p).astype(F32)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    m = (m * F32(beta1) + F32(1.0 - beta1) * g).astype(F32)
    v = (v * F32(beta2) + F32(1.0 - beta2) * g * g).astype(F32)
    denom = (np.sqrt(v) / F32(np.sqrt(bc2)) + F32(eps)).astype(F32)
    p = (p + F32(-(lr / bc1)) * (m / denom)).astype(F32)
    return p, m, v
