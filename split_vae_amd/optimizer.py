"""Host-side mirror of tf.keras.optimizers.Adam as used at vae/main.py:65-69 (Keras defaults
beta_1=.9, beta_2=.999, epsilon=1e-7; epsilon OUTSIDE the bias correction [TF-2.0 semantics]).
The update itself is the HIP kernel sv_adam_step over the model's flat fp32 buffers."""
import torch

from . import ops


class ExponentialDecay:
    """tf.optimizers.schedules.ExponentialDecay (vae/main.py:67): lr * rate^(step/decay_steps)."""

    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        self.initial_learning_rate = initial_learning_rate
        self.decay_steps = decay_steps
        self.decay_rate = decay_rate
        self.staircase = staircase

    def __call__(self, step):
        p = step / self.decay_steps
        if self.staircase:
            p = step // self.decay_steps
        return self.initial_learning_rate * (self.decay_rate ** p)


class Adam:
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate = learning_rate
        self.beta_1, self.beta_2, self.epsilon = beta_1, beta_2, epsilon
        self.iterations = 0
        self._slots = {}      # data_ptr -> (m, v)

    def lr(self):
        lr = self.learning_rate
        return float(lr(self.iterations)) if callable(lr) else float(lr)

    def slots(self, var):
        key = (var.data_ptr(), var.numel())
        if key not in self._slots:
            self._slots[key] = (torch.zeros_like(var), torch.zeros_like(var))
        return self._slots[key]

    def apply_flat(self, params, grads, grad_scale=1.0):
        """One launch over a flat fp32 parameter buffer (the trainer's path)."""
        m, v = self.slots(params)
        lr = self.lr()
        self.iterations += 1
        ops.adam_step(params, grads, m, v, self.iterations, lr, self.beta_1, self.beta_2, self.epsilon, grad_scale)

    def apply_gradients(self, grads_and_vars):
        """Keras surface (vae/trainer.py:138): list of (grad, var) device tensors; vars updated in place."""
        lr = self.lr()
        self.iterations += 1
        for g, var in grads_and_vars:
            if var.numel() % 4 or var.data_ptr() % 16 or g.data_ptr() % 16:
                raise ValueError("variables must be 16-byte aligned fp32 buffers with numel % 4 == 0")
            m, v = self.slots(var)
            ops.adam_step(var, g, m, v, self.iterations, lr, self.beta_1, self.beta_2, self.epsilon, 1.0)
