"""NumPy restatement of the TF1 training-step arithmetic the reference builds around the hot path
(models/sparse_graph_model.py:227-260): optimizer.compute_gradients -> per-variable tf.clip_by_norm ->
optimizer.apply_gradients, for the three optimizers the reference can select.

TEST INFRASTRUCTURE; PARITY UNPINNED (oracle/__init__.py): the update rules are TensorFlow 1.13 internals
([TF-internal], recalled from tensorflow/python/ops/clip_ops.py, training/adam.py, training/rmsprop.py,
training/gradient_descent.py and the kernels in core/kernels/training_ops.cc), not code under /root/reference.
All arithmetic runs in float32 like the TF kernels; scalars (lr, betas) are rounded to float32 first, as TF casts
the Python hyper-parameters to the variable dtype.
"""
import numpy as np

F = np.float32


def clip_by_norm(t, clip_norm):
    """tf.clip_by_norm(t, clip_norm) [TF-internal clip_ops.py]: t * clip_norm / max(||t||_2, clip_norm), evaluated as
    (t * clip_norm) / max(l2norm, clip_norm); an all-zero tensor stays zero (the where() guard around sqrt)."""
    t = np.asarray(t, F)
    l2sum = np.sum(t * t, dtype=F)
    l2norm = np.sqrt(l2sum) if l2sum > 0 else l2sum
    return (t * F(clip_norm)) / np.maximum(l2norm, F(clip_norm))


class GradientDescent:
    """tf.train.GradientDescentOptimizer: var -= lr * g  (ApplyGradientDescent)."""

    def __init__(self, variables, learning_rate):
        self.vars = [np.array(v, F) for v in variables]
        self.lr = F(learning_rate)

    def apply_gradients(self, grads, lr_scale=1.0):
        lr = F(self.lr * F(lr_scale))
        for v, g in zip(self.vars, grads):
            if g is not None:
                v -= lr * np.asarray(g, F)


class RMSProp:
    """tf.train.RMSPropOptimizer(lr, decay, momentum, epsilon=1e-10), not centered (ApplyRMSProp):
        ms  <- ms + (g*g - ms) * (1 - decay)            (slot "rms", initialised to ONES)
        mom <- mom * momentum + lr * g / sqrt(ms + epsilon)   (slot "momentum", zeros)
        var <- var - mom"""

    def __init__(self, variables, learning_rate, decay=0.9, momentum=0.0, epsilon=1e-10):
        self.vars = [np.array(v, F) for v in variables]
        self.lr, self.decay, self.momentum, self.eps = F(learning_rate), F(decay), F(momentum), F(epsilon)
        self.ms = [np.ones_like(v) for v in self.vars]
        self.mom = [np.zeros_like(v) for v in self.vars]

    def apply_gradients(self, grads, lr_scale=1.0):
        lr = F(self.lr * F(lr_scale))
        for v, ms, mom, g in zip(self.vars, self.ms, self.mom, grads):
            if g is None:
                continue
            g = np.asarray(g, F)
            ms += (g * g - ms) * (F(1) - self.decay)
            mom[...] = mom * self.momentum + lr * g / np.sqrt(ms + self.eps)
            v -= mom


class Adam:
    """tf.train.AdamOptimizer(lr, beta1=0.9, beta2=0.999, epsilon=1e-8) (ApplyAdam):
        lr_t <- lr * sqrt(1 - beta2^t) / (1 - beta1^t)        (beta powers are non-trainable variables, start at beta)
        m    <- m + (g - m) * (1 - beta1)
        v    <- v + (g*g - v) * (1 - beta2)
        var  <- var - lr_t * m / (sqrt(v) + epsilon)          (epsilon OUTSIDE the bias correction: 'epsilon hat')"""

    def __init__(self, variables, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.vars = [np.array(v, F) for v in variables]
        self.lr, self.b1, self.b2, self.eps = F(learning_rate), F(beta1), F(beta2), F(epsilon)
        self.m = [np.zeros_like(v) for v in self.vars]
        self.v = [np.zeros_like(v) for v in self.vars]
        self.beta1_power, self.beta2_power = F(beta1), F(beta2)

    def apply_gradients(self, grads, lr_scale=1.0):
        lr = F(self.lr * F(lr_scale))
        lr_t = F(lr * np.sqrt(F(1) - self.beta2_power) / (F(1) - self.beta1_power))
        for var, m, v, g in zip(self.vars, self.m, self.v, grads):
            if g is None:
                continue
            g = np.asarray(g, F)
            m += (g - m) * (F(1) - self.b1)
            v += (g * g - v) * (F(1) - self.b2)
            var -= lr_t * m / (np.sqrt(v) + self.eps)
        self.beta1_power = F(self.beta1_power * self.b1)      # finish(): updated once per apply_gradients
        self.beta2_power = F(self.beta2_power * self.b2)


def make_optimizer(name, variables, learning_rate, decay=0.98, momentum=0.85):
    """models/sparse_graph_model.py:239-249 (the reference passes params['learning_rate_decay'] as RMSProp's decay)."""
    name = name.lower()
    if name == 'sgd':
        return GradientDescent(variables, learning_rate)
    if name == 'rmsprop':
        return RMSProp(variables, learning_rate, decay=decay, momentum=momentum)
    if name == 'adam':
        return Adam(variables, learning_rate)
    raise Exception('Unknown optimizer "%s".' % name)


def train_step(opt, grads, clamp_gradient_norm, lr_scale=1.0):
    """:251-260: per-variable clip_by_norm, then apply_gradients."""
    clipped = [None if g is None else clip_by_norm(g, clamp_gradient_norm) for g in grads]
    opt.apply_gradients(clipped, lr_scale)
    return clipped
