"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

CPU (numpy) restatement of the reference's hot path (ankitdhall/learning_embeddings, joint
image+label hyperbolic entailment-cone training step).  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import this module, and only as the checker / the timed
CPU baseline -- never as a fallback for the HIP path.

Parity status: PINNED.  Every function here is checked in tests/test_oracle_golden.py against
fixtures F1..F14 (+ F4b: the negative stream on config 5's 8-level 50 000-label hierarchy) under
tests/golden/, which were produced by importing the reference itself in the build container
(tests/golden/make_golden*.py).

Each function cites the reference file:line it restates (paths relative to the reference root).
Forward arithmetic is done in float32 in the reference's operation order; analytic backward
passes are evaluated in float64 from the float32 forward's branch decisions (clamp masks).
"""
import numpy as np

F32 = np.float32
EPS_CLAMP = 1e-5


# --------------------------------------------------------------------------------------------
# constants (network/oe_h.py:59,75,106-110)
# --------------------------------------------------------------------------------------------
def inner_radius(K):
    """oe_h.py:59  r_in = 2K / (1 + sqrt(1 + 4K^2)) (python float64)."""
    return 2 * K / (1 + np.sqrt(1 + 4 * K * K))


def inner_radius_h(K):
    """oe_h.py:75,106-110  atanh(clamp(float32(r_in), +-(1-1e-5))) evaluated in float32."""
    x = F32(inner_radius(K))
    x = np.clip(x, F32(-1 + 1e-5), F32(1 - 1e-5))
    return F32(F32(0.5) * (np.log(F32(1) + x) - np.log(F32(1) - x)))


def _rownorm(a):
    return np.sqrt(np.sum(a * a, axis=1, dtype=a.dtype))


# --------------------------------------------------------------------------------------------
# a2: Embedder.forward + soft_clip (oe_h.py:77-104)
# --------------------------------------------------------------------------------------------
def embedder_forward(W, idx, K, return_saved=False):
    W = np.asarray(W, dtype=F32)
    e = W[np.asarray(idx)] + F32(1e-15)                                   # :78-79
    n = _rownorm(e)                                                       # :81
    t = np.tanh(np.clip(inner_radius_h(K) + n, F32(-15), F32(15))).astype(F32)   # :83
    den = np.maximum(n, F32(1e-12))                                       # F.normalize eps
    out = (t[:, None] * (e / den[:, None])).astype(F32)
    r_in = F32(inner_radius(K))
    no = _rownorm(out)                                                    # :101
    lo = no <= r_in                                                       # :102
    hi = no >= F32(1.0)                                                   # :103
    clipped = out.copy()
    clipped[lo] = out[lo] / no[lo, None] * r_in
    clipped[hi] = out[hi] / no[hi, None] * F32(1.0 - EPS_CLAMP)
    if return_saved:
        return clipped, (e, n, t, den)
    return clipped


def embedder_backward(W, idx, gout, K):
    """Dense gradient w.r.t. the table (nn.Embedding sparse=False): the clip is invisible to
    autograd (in-place under no_grad, oe_h.py:100-103)."""
    _, (e, n, t, den) = embedder_forward(W, idx, K, True)
    e = e.astype(np.float64); n = n.astype(np.float64); t = t.astype(np.float64); den = den.astype(np.float64)
    go = np.asarray(gout, dtype=np.float64)
    arg = float(inner_radius_h(K)) + n
    tp = np.where((arg >= -15.0) & (arg <= 15.0), 1.0 - t * t, 0.0)      # d tanh(clamp(.)) / dn
    denp = (n >= 1e-12).astype(np.float64)                                # clamp_min gradient
    nsafe = np.where(n > 0, n, 1.0)
    dot = np.sum(e * go, axis=1)
    coef = (tp / den - t * denp / (den * den)) * dot / nsafe
    ge = (t / den)[:, None] * go + coef[:, None] * e
    gW = np.zeros(np.asarray(W).shape, dtype=np.float64)
    np.add.at(gW, np.asarray(idx), ge)
    return gW


# --------------------------------------------------------------------------------------------
# a3: FeatCNN18.soft_clip (oe_h.py:323-328)   out = normalize(x) * (||x|| + r_in)   (NOT a ball projection)
# --------------------------------------------------------------------------------------------
def image_soft_clip(raw, K):
    raw = np.asarray(raw, dtype=F32)
    n = _rownorm(raw)
    den = np.maximum(n, F32(1e-12))
    return ((raw / den[:, None]) * (n + F32(inner_radius(K)))[:, None]).astype(F32)


def image_soft_clip_backward(raw, gout, K):
    x = np.asarray(raw, dtype=np.float64); go = np.asarray(gout, dtype=np.float64)
    n = np.sqrt(np.sum(x * x, axis=1)); den = np.maximum(n, 1e-12)
    denp = (n >= 1e-12).astype(np.float64)
    r = inner_radius(K)
    nsafe = np.where(n > 0, n, 1.0)
    dot = np.sum(x * go, axis=1)
    coef = (1.0 / den - (n + r) * denp / (den * den)) * dot / nsafe
    return ((n + r) / den)[:, None] * go + coef[:, None] * x


def featnet_forward(inp, w, b, K):
    """oe_h.py:168-224 FeatNet.forward (fc1 -> +1e-15 -> exp0-style tanh projection -> soft_clip with 1e-6 guards)."""
    x = (np.asarray(inp, F32) @ np.asarray(w, F32).T + np.asarray(b, F32)).astype(F32) + F32(1e-15)
    n = _rownorm(x)
    t = np.tanh(np.clip(inner_radius_h(K) + n, F32(-15), F32(15))).astype(F32)
    out = (t[:, None] * (x / np.maximum(n, F32(1e-12))[:, None])).astype(F32)
    r_in = F32(inner_radius(K)); no = _rownorm(out)
    lo = no <= r_in; hi = no >= F32(1.0)
    res = out.copy()
    res[lo] = (F32(1e-6) + out[lo]) / (F32(1e-6) + no[lo, None]) * r_in   # :222
    res[hi] = out[hi] / no[hi, None] * F32(1.0 - EPS_CLAMP)
    return res


# --------------------------------------------------------------------------------------------
# a7: hyperbolic cone energy  E_operator (oe_h.py:811-833)
# --------------------------------------------------------------------------------------------
def _cone_terms(x, y, K, dt):
    x = np.asarray(x, dtype=dt).reshape(-1, np.shape(x)[-1]); y = np.asarray(y, dtype=dt).reshape(x.shape)
    K = dt(K)
    xn = _rownorm(x); yn = _rownorm(y); dist = _rownorm(x - y)            # :817-819
    s = np.sum(x * y, axis=1, dtype=dt)                                   # :821
    xn2 = xn * xn; yn2 = yn * yn
    num = s * (1 + xn2) - xn2 * (1 + yn2)
    xy = xn * yn
    rad = 1 + xy * xy - 2 * s
    with np.errstate(invalid='ignore', divide='ignore'):
        sq = np.sqrt(rad)
        den = xn * dist * sq
        a = num / den                                                     # :823
        pa = K * (1 - xn2) / xn
    lo, hi = dt(-1 + EPS_CLAMP), dt(1 - EPS_CLAMP)
    ac = np.clip(a, lo, hi); pc = np.clip(pa, lo, hi)
    theta = np.arccos(ac); psi = np.arcsin(pc)                            # :826-827
    diff = theta - psi
    return dict(x=x, y=y, xn=xn, yn=yn, dist=dist, s=s, xn2=xn2, yn2=yn2, num=num, rad=rad, den=den, a=a, pa=pa,
                ac=ac, pc=pc, diff=diff, lo=lo, hi=hi)


def cone_energy(x, y, K, dtype=F32):
    shp = np.shape(x)[:-1]
    t = _cone_terms(x, y, K, dtype)
    return np.maximum(t['diff'], dtype(0)).reshape(shp)                  # :833


def cone_energy_grad(x, y, gE, K):
    """Analytic dE/dx, dE/dy (float64) with the clamp/hinge masks taken from the float32 forward."""
    D = np.shape(x)[-1]
    f = _cone_terms(x, y, K, F32)
    d = _cone_terms(x, y, K, np.float64)
    g = np.asarray(gE, dtype=np.float64).reshape(-1)
    live = (f['diff'] >= 0).astype(np.float64)                            # clamp(min=0) passes grad where input >= 0
    a_in = ((f['a'] >= f['lo']) & (f['a'] <= f['hi'])).astype(np.float64)
    p_in = ((f['pa'] >= f['lo']) & (f['pa'] <= f['hi'])).astype(np.float64)
    xn, yn, dist, s, xn2, yn2, num, rad, den, a = (d[k] for k in ('xn', 'yn', 'dist', 's', 'xn2', 'yn2', 'num', 'rad', 'den', 'a'))
    with np.errstate(invalid='ignore', divide='ignore'):
        dth = -1.0 / np.sqrt(1 - d['ac'] ** 2) * a_in
        dps = 1.0 / np.sqrt(1 - d['pc'] ** 2) * p_in
        A_ = g * live * dth
        P_ = -g * live * dps
        da_ds = (1 + xn2) / den + a / rad
        da_dd = -a / dist
        da_dxn = (2 * xn * s - 2 * xn * (1 + yn2)) / den - a * (1 / xn + xn * yn2 / rad)
        da_dyn = -2 * xn2 * yn / den - a * xn2 * yn / rad
        dpa_dxn = -float(K) * (1 + xn2) / xn2
        g_s = A_ * da_ds; g_d = A_ * da_dd; g_xn = A_ * da_dxn + P_ * dpa_dxn; g_yn = A_ * da_dyn
        cxx = g_xn / xn + g_d / dist; cxy = g_s - g_d / dist; cyy = g_yn / yn + g_d / dist
    z = (g * live) == 0                                                   # dead pairs: exact zeros, no NaN leakage
    cxx = np.where(z, 0.0, cxx); cxy = np.where(z, 0.0, cxy); cyy = np.where(z, 0.0, cyy)
    gx = cxx[:, None] * d['x'] + cxy[:, None] * d['y']
    gy = cxy[:, None] * d['x'] + cyy[:, None] * d['y']
    shp = np.shape(x)
    return gx.reshape(shp), gy.reshape(shp)


# --------------------------------------------------------------------------------------------
# a12: Euclidean order-violation energy (order_embeddings.py:818-824)
# --------------------------------------------------------------------------------------------
def order_energy(x, y, dtype=F32):
    x = np.asarray(x, dtype); y = np.asarray(y, dtype)
    return np.sum(np.maximum(x - y, dtype(0)) ** 2, axis=-1, dtype=dtype)


def order_energy_grad(x, y, gE):
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    r = 2 * np.maximum(x - y, 0) * np.asarray(gE, np.float64)[..., None]
    return r, -r


# --------------------------------------------------------------------------------------------
# 8f rank 4: the Euclidean entailment-cone sibling (network/oe.py)
#   soft_clip  oe.py:75-80 (Embedder) / :235-240 (FeatCNN18):  normalize(x) * (||x|| + K)
#   E_operator oe.py:721-739:  theta = -<normalize(x), normalize(y - x)>,  psi = -sqrt(1 - K^2/||x||^2),  E = max(theta - psi, 0)
# --------------------------------------------------------------------------------------------
def soft_clip_add(raw, add):
    raw = np.asarray(raw, dtype=F32)
    n = _rownorm(raw)
    den = np.maximum(n, F32(1e-12))
    return ((raw / den[:, None]) * (n + F32(add))[:, None]).astype(F32)


def soft_clip_add_backward(raw, gout, add):
    x = np.asarray(raw, dtype=np.float64); go = np.asarray(gout, dtype=np.float64)
    n = np.sqrt(np.sum(x * x, axis=1)); den = np.maximum(n, 1e-12)
    denp = (n >= 1e-12).astype(np.float64)
    nsafe = np.where(n > 0, n, 1.0)
    dot = np.sum(x * go, axis=1)
    coef = (1.0 / den - (n + add) * denp / (den * den)) * dot / nsafe
    return ((n + add) / den)[:, None] * go + coef[:, None] * x


def _euc_cone_terms(x, y, K, dt):
    x = np.asarray(x, dtype=dt).reshape(-1, np.shape(x)[-1]); y = np.asarray(y, dtype=dt).reshape(x.shape)
    xn = _rownorm(x)                                                      # oe.py:727
    df = y - x
    dn = _rownorm(df)
    xh = x / np.maximum(xn, dt(1e-12))[:, None]; dh = df / np.maximum(dn, dt(1e-12))[:, None]   # F.normalize eps
    theta = -np.sum(xh * dh, axis=1, dtype=dt)                            # :735
    with np.errstate(invalid='ignore', divide='ignore'):
        psi = -np.sqrt(1 - (dt(float(K) * float(K)) / xn ** 2))           # :737
    return dict(x=x, y=y, xn=xn, dn=dn, df=df, theta=theta, psi=psi, diff=theta - psi)


def euc_cone_energy(x, y, K, dtype=F32):
    shp = np.shape(x)[:-1]
    return np.maximum(_euc_cone_terms(x, y, K, dtype)['diff'], dtype(0)).reshape(shp)   # :739


def euc_cone_energy_grad(x, y, gE, K):
    """Analytic dE/dx, dE/dy (float64); the hinge mask comes from the float32 forward."""
    f = _euc_cone_terms(x, y, K, F32)
    d = _euc_cone_terms(x, y, K, np.float64)
    g = np.asarray(gE, dtype=np.float64).reshape(-1) * (f['diff'] >= 0)
    xn, dn, psi = d['xn'], d['dn'], d['psi']
    u = np.sum(d['x'] * d['df'], axis=1)
    ok_x = xn >= 1e-12; ok_d = dn >= 1e-12
    xc = np.maximum(xn, 1e-12); dc = np.maximum(dn, 1e-12)
    with np.errstate(invalid='ignore', divide='ignore'):
        A_ = -1.0 / (xc * dc)                                             # d theta / d u
        Bx = np.where(ok_x, u / (xc * xc * dc), 0.0) - float(K) ** 2 / (xn ** 3 * psi)   # d theta/d|x| - d psi/d|x|
        C_ = np.where(ok_d, u / (xc * dc * dc), 0.0)                      # d theta / d|y - x|
        cd = C_ / dc
        cxx = g * (-2 * A_ + Bx / xn + cd); cxy = g * (A_ - cd); cyy = g * cd
    z = g == 0
    cxx = np.where(z, 0.0, cxx); cxy = np.where(z, 0.0, cxy); cyy = np.where(z, 0.0, cyy)
    gx = cxx[:, None] * d['x'] + cxy[:, None] * d['y']
    gy = cxy[:, None] * d['x'] + cyy[:, None] * d['y']
    shp = np.shape(x)
    return gx.reshape(shp), gy.reshape(shp)


# --------------------------------------------------------------------------------------------
# a8/a9: the criterion's train-mode forward + backward given the sampled negatives
#        (oe_h.py:904-967 with :835-847; negative slot layout :951-957)
# --------------------------------------------------------------------------------------------
def joint_loss_fwd_bwd(W, R, pos_from, pos_to, neg, alpha, K, weights=None, energy='hyp_cone'):
    """Nodes are integer indices: < N -> label row of W (through Embedder.forward), >= N -> image j = ix - N whose raw
    CNN output is R[j] (through FeatCNN18.soft_clip).  neg[b, p] (p<K) corrupts the `to` side of (from_b, .);
    neg[b, K+p] corrupts the `from` side of (., to_b).
    Returns loss, e_pos[B], e_neg[B,2K], gW[N,D], gR[M,D]  (float64 grads)."""
    W = np.asarray(W, F32); N = W.shape[0]
    R = np.zeros((0, W.shape[1]), F32) if R is None else np.asarray(R, F32)
    pos_from = np.asarray(pos_from); pos_to = np.asarray(pos_to); neg = np.asarray(neg)
    B, K2 = neg.shape; Kn = K2 // 2
    nf = np.empty((B, K2), dtype=np.int64); nt = np.empty((B, K2), dtype=np.int64)
    nf[:, :Kn] = pos_from[:, None]; nt[:, :Kn] = neg[:, :Kn]
    nf[:, Kn:] = neg[:, Kn:];       nt[:, Kn:] = pos_to[:, None]
    frm = np.concatenate([pos_from, nf.reshape(-1)]); to = np.concatenate([pos_to, nt.reshape(-1)])

    def embed(ix):
        out = np.zeros((len(ix), W.shape[1]), F32)
        lab = ix < N
        if lab.any():
            out[lab] = (embedder_forward(W, ix[lab], K) if energy == 'hyp_cone' else
                        soft_clip_add(W[ix[lab]], K) if energy == 'euc_cone' else W[ix[lab]])      # oe.py:65-80
        if (~lab).any():
            out[~lab] = (soft_clip_add(R[ix[~lab] - N], K) if energy == 'euc_cone' else       # oe.py:225-240
                         image_soft_clip(R[ix[~lab] - N], K))
        return out

    x = embed(frm); y = embed(to)
    if energy == 'hyp_cone':
        E = cone_energy(x, y, K)
    elif energy == 'euc_cone':
        E = euc_cone_energy(x, y, K)
    else:
        E = order_energy(x, y)
    e_pos = E[:B]; e_neg = E[B:].reshape(B, K2)
    w = np.ones(B, F32) if weights is None else np.asarray(weights, F32)
    hinge = np.maximum(F32(alpha) - e_neg, F32(0))
    loss = np.sum(w * e_pos, dtype=F32) + np.sum(w * np.sum(hinge, axis=1, dtype=F32), dtype=F32)   # :846
    gE = np.concatenate([w.astype(np.float64),
                         (-(w[:, None].astype(np.float64)) * ((F32(alpha) - e_neg) >= 0)).reshape(-1)])
    if energy == 'hyp_cone':
        gx, gy = cone_energy_grad(x, y, gE, K)
    elif energy == 'euc_cone':
        gx, gy = euc_cone_energy_grad(x, y, gE, K)
    else:
        gx, gy = order_energy_grad(x, y, gE)
    gW = np.zeros(W.shape, np.float64); gR = np.zeros(R.shape, np.float64)
    for ix, gout in ((frm, gx), (to, gy)):
        lab = ix < N
        if lab.any():
            if energy == 'hyp_cone':
                gW += embedder_backward(W, ix[lab], gout[lab], K)
            elif energy == 'euc_cone':
                np.add.at(gW, ix[lab], soft_clip_add_backward(W[ix[lab]], gout[lab], K))
            else:
                np.add.at(gW, ix[lab], gout[lab])
        if (~lab).any():
            j = ix[~lab] - N
            np.add.at(gR, j, soft_clip_add_backward(R[j], gout[~lab], K) if energy == 'euc_cone' else
                      image_soft_clip_backward(R[j], gout[~lab], K))
    return loss, e_pos, e_neg, gW, gR


# --------------------------------------------------------------------------------------------
# a10: label-table maintenance   lambda-rescale -> Adam -> soft_clip   (oe_h.py:1766-1771, :1632-1636, :1604-1617)
# --------------------------------------------------------------------------------------------
def table_clip(W, K):
    W = np.asarray(W, F32).copy()
    r_in = F32(inner_radius(K)); n = _rownorm(W)
    lo = n <= r_in; hi = n >= F32(1.0)
    W[lo] = W[lo] / n[lo, None] * r_in
    W[hi] = W[hi] / n[hi, None] * F32(1.0 - EPS_CLAMP)
    return W


def riemannian_rescale(W, g):
    """grad *= (1/lambda_x)^2, lambda_x = 2/(1-||w||)   (norm, not squared norm: oe_h.py:1636)."""
    lam = F32(2.0) / (F32(1.0) - _rownorm(np.asarray(W, F32)))
    return (np.asarray(g, F32) * ((F32(1.0) / lam) ** 2)[:, None]).astype(F32)


def adam_update(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam single-tensor arithmetic (torch 2.10 _single_tensor_adam, no amsgrad/weight-decay)."""
    p = np.asarray(p, F32); g = np.asarray(g, F32)
    m = (m + (g - m) * F32(1 - beta1)).astype(F32)                        # lerp_
    v = (v * F32(beta2) + F32(1 - beta2) * g * g).astype(F32)
    bc1 = 1 - beta1 ** step; bc2 = 1 - beta2 ** step
    denom = (np.sqrt(v) / F32(bc2 ** 0.5) + F32(eps)).astype(F32)
    p = (p + F32(-(lr / bc1)) * (m / denom)).astype(F32)
    return p, m, v


def table_step_adam(W, g, m, v, step, lr, K, beta1=0.9, beta2=0.999, eps=1e-8):
    g2 = riemannian_rescale(W, g)
    W2, m2, v2 = adam_update(W, g2, m, v, step, lr, beta1, beta2, eps)
    return table_clip(W2, K), m2, v2


def table_step_rsgd(W, g, lr, K):
    """oe_h.py:1761-1762 (+ :1638-1644 exp_map_x, :1619-1630 mob_add): true Riemannian SGD."""
    W = np.asarray(W, F32)
    g2 = riemannian_rescale(W, g)
    vv = (F32(-lr) * g2 + F32(1e-15)).astype(F32)
    nv = _rownorm(vv)
    lam = F32(2.0) / (F32(1.0) - _rownorm(W))
    t = np.tanh(np.clip(lam * nv / F32(2), F32(-15), F32(15))).astype(F32)
    second = (t[:, None] * vv / nv[:, None]).astype(F32)
    tt = second + F32(1e-6)
    dot2 = F32(2.0) * np.sum(W * tt, axis=1, dtype=F32)
    xx = np.sum(W * W, axis=1, dtype=F32); t2 = np.sum(tt * tt, axis=1, dtype=F32)
    den = F32(1.0) + dot2 + t2 * xx
    res = (((F32(1.0) + dot2 + t2) / den)[:, None] * W + ((F32(1.0) - xx) / den)[:, None] * tt).astype(F32)
    return table_clip(res, K)


# --------------------------------------------------------------------------------------------
# a13: MultiLevelCELoss (loss.py:5-38; class_weights = its `weight` argument, :16-25: CrossEntropyLoss(weight=slice, reduction='none')
# scales a sample's term by its target class's weight)
# --------------------------------------------------------------------------------------------
def multilevel_ce(logits, level_labels, levels, level_weights=None, class_weights=None):
    z = np.asarray(logits, np.float64); B = z.shape[0]
    lw = [1.0] * len(levels) if level_weights is None else list(level_weights)
    cw = None if class_weights is None else np.asarray(class_weights, np.float64)
    per = np.zeros(B); g = np.zeros_like(z); s = 0
    for l, n in enumerate(levels):
        zl = z[:, s:s + n]; zm = zl - zl.max(axis=1, keepdims=True)
        lse = np.log(np.exp(zm).sum(axis=1)); lab = np.asarray(level_labels)[:, l]
        wl = lw[l] * (cw[s + lab] if cw is not None else np.ones(B))
        per += wl * (lse - zm[np.arange(B), lab])
        p = np.exp(zm - lse[:, None]); p[np.arange(B), lab] -= 1.0
        g[:, s:s + n] = wl[:, None] * p / B
        s += n
    return per.mean(), g


# --------------------------------------------------------------------------------------------
# a6: negative sampler, dense-matrix restatement (oe_h.py:849-902) on CPython's MT19937 `random.choice`
# --------------------------------------------------------------------------------------------
class MT19937:
    """CPython `random.Random` core (Modules/_randommodule.c, CPython 3.x -- the reference targets 3.6, this image
    runs 3.10; `choice -> _randbelow -> getrandbits(k)` is unchanged between them): init_by_array seeding from the
    32-bit limbs of abs(seed), genrand_uint32, getrandbits(k<=32) = top k bits, _randbelow by rejection."""
    N, M = 624, 397

    def __init__(self, seed=0):
        self.seed(seed)

    def _init_genrand(self, s):
        mt = [0] * self.N
        mt[0] = s & 0xffffffff
        for i in range(1, self.N):
            mt[i] = (1812433253 * (mt[i - 1] ^ (mt[i - 1] >> 30)) + i) & 0xffffffff
        self.mt, self.idx = mt, self.N

    def seed(self, seed):
        seed = abs(int(seed))
        key = []
        while True:
            key.append(seed & 0xffffffff); seed >>= 32
            if seed == 0:
                break
        self._init_genrand(19650218)
        mt, N = self.mt, self.N
        i, j = 1, 0
        for _ in range(max(N, len(key))):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525)) + key[j] + j) & 0xffffffff
            i += 1; j += 1
            if i >= N: mt[0] = mt[N - 1]; i = 1
            if j >= len(key): j = 0
        for _ in range(N - 1):
            mt[i] = ((mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941)) - i) & 0xffffffff
            i += 1
            if i >= N: mt[0] = mt[N - 1]; i = 1
        mt[0] = 0x80000000

    def u32(self):
        if self.idx >= self.N:
            mt, N, M = self.mt, self.N, self.M
            for k in range(N):
                y = (mt[k] & 0x80000000) | (mt[(k + 1) % N] & 0x7fffffff)
                mt[k] = mt[(k + M) % N] ^ (y >> 1) ^ (0x9908b0df if y & 1 else 0)
            self.idx = 0
        y = self.mt[self.idx]; self.idx += 1
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680; y ^= (y << 15) & 0xefc60000; y ^= y >> 18
        return y & 0xffffffff

    def getrandbits(self, k):
        assert 0 < k <= 32
        return self.u32() >> (32 - k)

    def randbelow(self, n):
        k = int(n).bit_length()
        r = self.getrandbits(k)
        while r >= n:
            r = self.getrandbits(k)
        return r


class DenseSampler:
    """oe_h.py:849-902 with the dense bool matrix A (1 = not a TC edge, diag 0), ascending np.where candidate order,
    optional per-level window, levels_to_hide remap (:850-854).  labels_only=True gives the labels-only trainer's
    variant (order_embeddings.py:797-816: level_id % L, no image slot)."""

    def __init__(self, A, levels, n_labels=None, pick_per_level=False, seed=0, labels_only=False):
        self.A = np.asarray(A, dtype=bool)
        self.levels = list(levels)
        self.level_start = [int(sum(levels[:i])) for i in range(len(levels))]
        self.level_stop = [int(sum(levels[:i + 1])) for i in range(len(levels))]
        self.n_labels = self.level_stop[-1] if n_labels is None else n_labels
        self.pick_per_level = pick_per_level
        self.labels_only = labels_only
        self.levels_to_hide = []
        self.rng = MT19937(seed)

    def seed(self, s):
        self.rng.seed(s)

    def _row(self, ix):
        return self.A[ix, :]

    def _col(self, ix):
        return self.A[:, ix]

    def candidates(self, side, ix, level_id):
        L = len(self.levels)
        if self.labels_only:
            level_id = level_id % L
        elif len(self.levels_to_hide) > 0:
            level_id = level_id % (L - len(self.levels_to_hide) + 1)
            # the reference's own expression (oe_h.py:854): the list is in CPython's set ITERATION order -- ascending for the 4-level ETHEC,
            # not for 8 levels with fewer than five slots left (fixture F4b: [8, 1, 2, 3]); the oracle is Python, so it evaluates the same expression
            level_id = list(set(list(range(L + 1))) - set(self.levels_to_hide))[level_id]
        else:
            level_id = level_id % (L + 1)
        c = np.where(self._row(ix) == 1)[0] if side == 0 else np.where(self._col(ix) == 1)[0]
        if self.pick_per_level:
            if level_id < L:
                c = c[(c >= self.level_start[level_id]) & (c < self.level_stop[level_id])]
            elif not self.labels_only:
                c = c[c < self.n_labels] if ix >= self.n_labels else c[c >= self.n_labels]
        return c

    def draw(self, side, ix, level_id):
        """side 0: `u` fixed (row of A, corrupt the `to` end); side 1: `v` fixed (column of A)."""
        c = self.candidates(side, ix, level_id)
        if len(c) == 0:
            raise IndexError('Cannot choose from an empty sequence')
        return int(c[self.rng.randbelow(len(c))])

    def draw_batch(self, pos_from, pos_to, K):
        """oe_h.py:940-957 call order: for b: for p<K: (u fixed) then (v fixed)."""
        B = len(pos_from)
        neg = np.zeros((B, 2 * K), dtype=np.int64)
        for b in range(B):
            for p in range(K):
                neg[b, p] = self.draw(0, int(pos_from[b]), p)
                neg[b, p + K] = self.draw(1, int(pos_to[b]), p)
        return neg


class LazyDenseSampler(DenseSampler):
    """DenseSampler without the matrix in memory: row `ix` / column `ix` of A = 1 - TC - I (oe_h.py:554-561) are written out per draw from the
    node's descendant / ancestor set -- the same np.where scan over N + M bools as the reference, for hierarchies whose dense A (2.9 GB at
    config 5's 54 096 nodes) should not travel with a test.  Pinned against the reference's own run over the real dense matrix: fixture F4b."""

    def __init__(self, levels, label_edges, image_leaf=None, **kw):
        N = int(sum(levels))
        M = 0 if image_leaf is None else len(image_leaf)
        self.n = N + M
        par = [[] for _ in range(N)]
        for u, v in label_edges:
            par[int(v)].append(int(u))
        memo = {}

        def anc(v):                                          # iterative (an 8-level chain is shallow, a general DAG may not be)
            stack = [v]
            while stack:
                w = stack[-1]
                todo = [p for p in par[w] if p not in memo]
                if todo:
                    stack.extend(todo); continue
                stack.pop()
                if w not in memo:
                    s = set()
                    for p in par[w]:
                        s.add(p); s |= memo[p]
                    memo[w] = s
            return memo[v]

        self.anc = [None] * self.n
        self.desc = [[] for _ in range(self.n)]
        for v in range(N):
            self.anc[v] = np.fromiter(sorted(anc(v)), dtype=np.int64)
        for j in range(M):
            leaf = int(image_leaf[j])
            self.anc[N + j] = np.fromiter(sorted(anc(leaf) | {leaf}), dtype=np.int64)
        for v in range(self.n):
            for a in self.anc[v]:
                self.desc[int(a)].append(v)
        self.desc = [np.asarray(d, dtype=np.int64) for d in self.desc]
        super().__init__(np.zeros((0, 0), dtype=bool), levels, n_labels=N, **kw)

    def _row(self, ix):                                      # A[ix, :]: 0 at ix and at every v with a TC edge ix -> v
        r = np.ones(self.n, dtype=bool)
        r[self.desc[ix]] = 0; r[ix] = 0
        return r

    def _col(self, ix):                                      # A[:, ix]: 0 at ix and at every ancestor of ix
        c = np.ones(self.n, dtype=bool)
        c[self.anc[ix]] = 0; c[ix] = 0
        return c


def dense_negative_adjacency(n_labels, label_edges, image_leaf=None):
    """oe_h.py:506-561 restated on integer data: TC of the label DAG plus, per image j, edges from image_leaf[j] and all
    its ancestors to node n_labels + j; A = 1 - TC - I."""
    par = {}
    for u, v in label_edges:
        par.setdefault(int(v), []).append(int(u))
    memo = {}

    def anc(v):
        if v not in memo:
            s = set()
            for p in par.get(v, []):
                s.add(p); s |= anc(p)
            memo[v] = s
        return memo[v]

    M = 0 if image_leaf is None else len(image_leaf)
    n = n_labels + M
    A = np.ones((n, n), dtype=bool)
    for v in range(n_labels):
        for a in anc(v):
            A[a, v] = 0
    for j in range(M):
        leaf = int(image_leaf[j])
        A[leaf, n_labels + j] = 0
        for a in anc(leaf):
            A[a, n_labels + j] = 0
    np.fill_diagonal(A, 0)
    return A


# --------------------------------------------------------------------------------------------
# image tensors of a step (network/oe_h.py:668-677, 700-712, 1463-1471)
# --------------------------------------------------------------------------------------------
def image_batch(u8, flips=None, c_out=3):
    """What the reference's transforms make of RESIZED uint8 images `u8` [n, H, W, 3] (the output of
    cv2.imread -> ToPILImage -> Resize((224, 224)), oe_h.py:623-626 / 1464-1471; channel order as decoded):
    RandomHorizontalFlip (oe_h.py:1465, train items only: `flips[i]` is its coin) mirrors along W, then
    ToTensor = HWC uint8 -> CHW float32, `img.to(float32).div(255)` (torchvision functional.to_tensor,
    third party: pinned by tests/test_image_store_cpu.py against torch's own uint8 -> float32 division).
    Returned [n, c_out, H, W] float32; c_out = 4 appends the zero channel the f32 stem kernel consumes."""
    u8 = np.asarray(u8)
    assert u8.dtype == np.uint8 and u8.ndim == 4 and u8.shape[3] == 3
    out = np.zeros((u8.shape[0], c_out, u8.shape[1], u8.shape[2]), dtype=F32)
    for i in range(u8.shape[0]):
        a = u8[i, :, ::-1, :] if (flips is not None and flips[i]) else u8[i]
        out[i, :3] = np.transpose(a, (2, 0, 1)).astype(F32) / F32(255)
    return out
