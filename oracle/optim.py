"""Oracle (test infrastructure): Keras 2.2.4 optimizer update rules in numpy fp32.

Optimizer names/knobs come from ``segmentation_pipeline/schemas/segmentation.raml:77-89``
(``optimizer: SGD|Adam|RMSprop|Nadam``, ``lr``, ``clipnorm``, ``clipvalue``).  The update
formulas restate keras/optimizers.py of Keras 2.2.4 (un-vendored: PARITY UNPINNED); they
differ from torch.optim (epsilon placement, momentum form).
"""
import numpy as np

f32 = np.float32


def clip_grads(grads, clipnorm=None, clipvalue=None):
    """Keras ``Optimizer.get_gradients``: global-norm clip, then value clip."""
    if clipnorm is not None and clipnorm > 0:
        norm = np.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads.values()))
        if norm > clipnorm:
            grads = {k: (g * f32(clipnorm / norm)).astype(f32) for k, g in grads.items()}
    if clipvalue is not None and clipvalue > 0:
        grads = {k: np.clip(g, -clipvalue, clipvalue).astype(f32) for k, g in grads.items()}
    return grads


class Adam:
    def __init__(self, lr=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7, decay=0.0):
        self.lr, self.b1, self.b2, self.eps, self.decay = lr, beta_1, beta_2, epsilon, decay
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads):
        lr = self.lr
        if self.decay > 0:
            lr = lr * (1.0 / (1.0 + self.decay * self.t))
        self.t += 1
        t = self.t
        lr_t = f32(lr * (np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)))
        for k, g in grads.items():
            m = self.m.get(k, np.zeros_like(g))
            v = self.v.get(k, np.zeros_like(g))
            m = (f32(self.b1) * m + f32(1.0 - self.b1) * g).astype(f32)
            v = (f32(self.b2) * v + f32(1.0 - self.b2) * g * g).astype(f32)
            params[k] = (params[k] - lr_t * m / (np.sqrt(v) + f32(self.eps))).astype(f32)
            self.m[k], self.v[k] = m, v
        return params


class SGD:
    def __init__(self, lr=0.01, momentum=0.0, decay=0.0, nesterov=False):
        self.lr, self.mu, self.decay, self.nesterov = lr, momentum, decay, nesterov
        self.t = 0
        self.vel = {}

    def step(self, params, grads):
        lr = self.lr
        if self.decay > 0:
            lr = lr * (1.0 / (1.0 + self.decay * self.t))
        self.t += 1
        for k, g in grads.items():
            v = self.vel.get(k, np.zeros_like(g))
            v = (f32(self.mu) * v - f32(lr) * g).astype(f32)
            if self.nesterov:
                params[k] = (params[k] + f32(self.mu) * v - f32(lr) * g).astype(f32)
            else:
                params[k] = (params[k] + v).astype(f32)
            self.vel[k] = v
        return params


class RMSprop:
    def __init__(self, lr=1e-3, rho=0.9, epsilon=1e-7, decay=0.0):
        self.lr, self.rho, self.eps, self.decay = lr, rho, epsilon, decay
        self.t = 0
        self.a = {}

    def step(self, params, grads):
        lr = self.lr
        if self.decay > 0:
            lr = lr * (1.0 / (1.0 + self.decay * self.t))
        self.t += 1
        for k, g in grads.items():
            a = self.a.get(k, np.zeros_like(g))
            a = (f32(self.rho) * a + f32(1.0 - self.rho) * g * g).astype(f32)
            params[k] = (params[k] - f32(lr) * g / (np.sqrt(a) + f32(self.eps))).astype(f32)
            self.a[k] = a
        return params


class Nadam:
    """keras/optimizers.py (2.2.4) Nadam: Nesterov Adam with the momentum schedule mu_t = beta_1 (1 - 0.5 * 0.96^(t * decay))."""

    def __init__(self, lr=0.002, beta_1=0.9, beta_2=0.999, epsilon=1e-7, schedule_decay=0.004):
        self.lr, self.b1, self.b2, self.eps, self.sd = lr, beta_1, beta_2, epsilon, schedule_decay
        self.t = 0
        self.m_schedule = 1.0
        self.m, self.v = {}, {}

    def step(self, params, grads):
        self.t += 1
        t = self.t
        mu_t = self.b1 * (1.0 - 0.5 * 0.96 ** (t * self.sd))
        mu_t1 = self.b1 * (1.0 - 0.5 * 0.96 ** ((t + 1) * self.sd))
        ms_new = self.m_schedule * mu_t
        ms_next = ms_new * mu_t1
        self.m_schedule = ms_new
        for k, g in grads.items():
            m = self.m.get(k, np.zeros_like(g))
            v = self.v.get(k, np.zeros_like(g))
            g_prime = g / f32(1.0 - ms_new)
            m = (f32(self.b1) * m + f32(1.0 - self.b1) * g).astype(f32)
            m_prime = m / f32(1.0 - ms_next)
            v = (f32(self.b2) * v + f32(1.0 - self.b2) * g * g).astype(f32)
            v_prime = v / f32(1.0 - self.b2 ** t)
            m_bar = f32(1.0 - mu_t) * g_prime + f32(mu_t1) * m_prime
            params[k] = (params[k] - f32(self.lr) * m_bar / (np.sqrt(v_prime) + f32(self.eps))).astype(f32)
            self.m[k], self.v[k] = m, v
        return params


OPTIMIZERS = {"adam": Adam, "sgd": SGD, "rmsprop": RMSprop, "nadam": Nadam}


def make(name, lr=None, **kw):
    cls = OPTIMIZERS[name.lower()]
    if lr is not None:
        kw["lr"] = lr
    return cls(**kw)
