"""CPU restatement of the reference's Adam loop.  TEST INFRASTRUCTURE ONLY.

Follows pyvbmc/vbmc/minimize_adam.py:62-146 statement by statement (same update
order, same in-place first update of the caller's x0, same stopping rule through
numpy.polyfit, same return tuple).  Pinned by tests/golden/adam.npz, which
oracle/make_golden.py produces by running the reference's own minimize_adam.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
import numpy as np


def minimize_adam(f, x0, lb=None, ub=None, tol_fun=0.001, max_iter=10000, master_min=0.001,
                  master_max=0.1, master_decay=200, use_early_stopping=True):
    eps_guard = np.sqrt(np.spacing(1))          # :62
    b1, b2, batch = 0.9, 0.999, 20              # :63-65
    tol_x, tol_x_max, tol_fun_max = 0.001, 0.1, tol_fun * 100  # :66-68
    nvars = np.size(x0)
    lo = np.full(nvars, -np.inf) if lb is None else lb
    hi = np.full(nvars, np.inf) if ub is None else ub
    mom1, mom2 = 0, 0
    xs = np.zeros((nvars, max_iter))
    ys = np.full(max_iter, np.nan)
    x = x0
    it = 0
    for it in range(max_iter):
        ys[it], g = f(x)                                          # :87
        mom1 = b1 * mom1 + (1 - b1) * g                           # :89
        mom2 = b2 * mom2 + (1 - b2) * g**2                        # :90
        mh = mom1 / (1 - b1 ** (it + 1))                          # :91
        vh = mom2 / (1 - b2 ** (it + 1))                          # :92
        lr = master_min + (master_max - master_min) * np.exp(-(it + 1) / master_decay)  # :94-96
        x -= lr * mh / (np.sqrt(vh) + eps_guard)                  # :97 (in place)
        x = np.minimum(hi, np.maximum(lo, x))                     # :98
        xs[:, it] = x
        end_of_batch = (it + 1) % batch == 0
        if use_early_stopping and end_of_batch and it + 1 >= 2 * batch:   # :103
            t = np.linspace(-(batch - 1) / 2, (batch - 1) / 2, batch)
            p, V = np.polyfit(t, ys[it - batch + 1 : it + 1], 1, cov=True)
            slope = p[0]
            se = np.sqrt(V[0, 0] + tol_fun**2)
            se_max = np.sqrt(V[0, 0] + tol_fun_max**2)
            recent = np.mean(xs[:, it - batch + 1 : it + 1], axis=1)
            before = np.mean(xs[:, it - 2 * batch + 1 : it + 1 - batch], axis=1)
            dx = np.sqrt(np.sum((recent - before) ** 2 / batch, axis=0))
            if (dx < tol_x and np.abs(slope) < se_max) or (np.abs(slope) < se and dx < tol_x_max):
                break
    x = np.mean(xs[:, it - batch + 1 : it + 1], axis=1)          # :142
    y = np.mean(ys[it - batch + 1 : it + 1])
    return x, y, xs[:, : it + 1], ys[: it + 1], it + 1
