"""
First-divergence diagnostic of the parity harness.

The HIP path's floats of the predictive differ from the reference's LAPACK-LU ones at the 1e-13 level, which can flip a
draw only if the visit's uniform lands within that distance of a CDF boundary.  When a parity test sees a differing
label this module tells the two apart: it finds the first visit at which the chains part (partial sweeps on both sides,
`bgmm_set_sweep_visits` / the C oracle's `n_visits`), rebuilds BOTH sides' `log_prob_z` of that visit -- the oracle's
through `go_probe_visit`, the device's through the method-level C-ABI (`bgmm_del_item`, `bgmm_log_post_pred`,
`bgmm_get_counts`, `bgmm_get_log_prior`), i.e. not through the window kernels whose verdict is in question -- and
reports the uniform, both CDFs' nearest boundary and the margin.  A margin of ~1e-13 is a tie; anything larger is a bug.

Test infrastructure: uses oracle/, never imported by the package.
"""
import numpy as np


def _cdf(logp):
    """The reference's normalisation (utils/utils.py:7-20 applied to crpmm.py:74-76): exp(lp - logsumexp), running sum."""
    top = logp.max()
    lse = np.log(np.exp(logp - top).sum()) + top
    p = np.exp(logp - lse)
    return p, np.cumsum(p)


def _draw(p, u):
    for j in range(len(p)):
        u = u - p[j]
        if u < 0:
            return j
    return len(p) - 1


def _weights(counts, power):
    n = counts.astype(np.float64)
    return np.log(n) if power is None else np.log(np.power(n, power))


def first_divergence(make_ctx, make_oracle, us, orders, powers, it, alpha=1.0, us_dev=None, max_bisect=24):
    """make_ctx() / make_oracle(): fresh device context / C oracle at the chain's initial state.  us[s], orders[s] (or
    None), powers[s] (or None): the inputs of sweep s.  `it`: the sweep after which the labels differed.  us_dev: the
    device's uniforms when they are meant to differ from the oracle's (the self-test of this diagnostic).
    Returns a dict (and a printable `text`)."""
    us_dev = us if us_dev is None else us_dev
    N = len(us[it])
    order = np.arange(N) if orders[it] is None else np.asarray(orders[it])

    def replay(n_vis):
        ctx, o = make_ctx(), make_oracle()
        for s in range(it):
            ctx.sweep(us_dev[s], orders[s], powers[s])
            o.sweep(us[s], orders[s], powers[s])
        if n_vis > 0:
            ctx.set_sweep_visits(n_vis)
            ctx.sweep(us_dev[it], orders[it], powers[it])
            o.sweep(us[it], orders[it], powers[it], n_visits=n_vis)
        return ctx, o

    # the first visit whose FINAL label differs is the candidate; confirm with partial sweeps, bisect if it is not
    ctx, o = replay(N)
    zd, zo = ctx.assignments(), o.z
    ctx.close()
    diff = np.nonzero(zd[order] != zo[order])[0]
    if diff.size == 0:
        return {"text": "no label differs after sweep %d" % it, "visit": None}
    lo, hi = 0, int(diff[0]) + 1            # invariant: equal after `lo` visits, different after `hi`
    ctx, o = replay(int(diff[0]))
    same = np.array_equal(ctx.assignments(), o.z)
    ctx.close()
    if same:
        lo = int(diff[0])
    steps = 0
    while hi - lo > 1 and steps < max_bisect:
        mid = (lo + hi) // 2
        ctx, o = replay(mid)
        if np.array_equal(ctx.assignments(), o.z):
            lo = mid
        else:
            hi = mid
        ctx.close()
        steps += 1
    p = lo                                   # visits 0 .. p-1 agree; visit p is where the chains part
    i = int(order[p])
    ctx, o = replay(p)
    k_old = int(o.z[i])
    lp_o = o.probe_visit(i, powers[it])
    if ctx.assignments()[i] >= 0:
        ctx.del_item(i)
    cnt = ctx.counts()
    lp_d = np.empty(len(cnt) + 1)
    lp_d[:-1] = _weights(cnt, powers[it]) + ctx.log_post_pred(i)
    lp_d[-1] = np.log(alpha) + ctx.log_prior()[i]
    ctx.close()
    u_o, u_d = float(us[it][p]), float(us_dev[it][p])
    p_o, c_o = _cdf(lp_o)
    out = {"sweep": it, "visit": p, "point": i, "u": u_o, "u_device": u_d, "home": k_old, "K": len(lp_o) - 1,
           "oracle_draw": _draw(p_o, u_o), "oracle_margin": float(np.abs(c_o - u_o).min()),
           "oracle_boundary": int(np.abs(c_o - u_o).argmin())}
    if len(lp_d) == len(lp_o):
        p_d, c_d = _cdf(lp_d)
        out.update(device_draw=_draw(p_d, u_d), device_margin=float(np.abs(c_d - u_d).min()),
                   max_abs_dlogp=float(np.abs(lp_d - lp_o)[np.isfinite(lp_o)].max()),
                   max_abs_dcdf=float(np.abs(c_d - c_o).max()))
    else:
        out.update(device_draw=None, device_margin=None, max_abs_dlogp=None, max_abs_dcdf=None,
                   note="K differs at the visit: device %d, oracle %d" % (len(lp_d) - 1, len(lp_o) - 1))
    tie = out["oracle_margin"] < 1e-11 and (out["max_abs_dcdf"] is None or out["max_abs_dcdf"] < 1e-11)
    out["verdict"] = "tie at the CDF boundary (rounding-level)" if tie else "NOT a tie: the scores or the state differ"
    out["text"] = ("first divergence: sweep %(sweep)d visit %(visit)d (point %(point)d, home %(home)d, K %(K)d): u = %(u).17g; "
                   "oracle draws %(oracle_draw)s, CDF margin %(oracle_margin).3e at boundary %(oracle_boundary)d; device "
                   "(method-level C-ABI) draws %(device_draw)s, margin %(device_margin)s, max |dlogp| %(max_abs_dlogp)s, "
                   "max |dCDF| %(max_abs_dcdf)s -> %(verdict)s") % out
    return out


def assert_same_labels(z_dev, z_ref, what, explain=None):
    """np.array_equal with a useful failure: how many labels differ, the first index, and -- when `explain` (a
    zero-argument callable returning first_divergence's dict) is given -- the CDF margin at the first divergence."""
    bad = np.nonzero(np.asarray(z_dev) != np.asarray(z_ref))[0]
    if bad.size == 0:
        return
    msg = "%s: %d labels differ, first at i=%d" % (what, bad.size, bad[0])
    if explain is not None:
        try:
            msg += "\n" + explain()["text"]
        except Exception as e:      # the diagnostic must never mask the failure it explains
            msg += "\n(first-divergence diagnostic failed: %r)" % (e,)
    raise AssertionError(msg)
