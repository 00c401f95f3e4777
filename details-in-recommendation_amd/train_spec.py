"""train_spec.py -- the reference DCN's training step around the HIP forward/backward path.

Mirrors models/DeepCrossNetwork/DeepCrossNetwork.py (reference):
  _get_train_op_fn                 :264-290   lr = _learning_rate_decay(params); optimizer(**optimizer_spec, learning_rate=lr);
                                              every gradient tensor clipped on its own with tf.clip_by_norm(g, 100.0)
  _learning_rate_decay / _no_decay :422-458   learning_rate_spec = {'learning_rate': ..., 'decay_method': name, **kwargs}
  _DECAY_METHOD_NAME               :26-31     the six tf.train decay schedules ([TF-upstream] formulas restated below)
  l2_reg                           :181-183   loss += l2_reg * sum of the deep kernels' l2 losses
Reference use: models/DeepCrossNetwork/train.py:111-125 (Adam, epsilon 1e-4, cosine_decay over 3000 steps, alpha 0.5).
Host-side scalar logic only; the gradients themselves come from autograd.py's HIP backward kernels.
"""
import math
import os

import torch

CLIP_NORM = 100.0   # DeepCrossNetwork.py:283


def exponential_decay(learning_rate, global_step, decay_steps, decay_rate, staircase=False, **_):
    p = global_step / decay_steps
    if staircase:
        p = math.floor(p)
    return learning_rate * decay_rate ** p


def piecewise_constant(x=None, boundaries=(), values=(), global_step=None, **_):
    x = global_step if x is None else x
    if len(values) != len(boundaries) + 1:
        raise ValueError("The length of boundaries should be 1 less than the length of values")
    for b, v in zip(boundaries, values):
        if x <= b:
            return v
    return values[-1]


def polynomial_decay(learning_rate, global_step, decay_steps, end_learning_rate=0.0001, power=1.0, cycle=False, **_):
    step = global_step
    if cycle:
        mult = 1.0 if step == 0 else math.ceil(step / decay_steps)
        decay_steps = decay_steps * mult
    else:
        step = min(step, decay_steps)
    return (learning_rate - end_learning_rate) * (1 - step / decay_steps) ** power + end_learning_rate


def cosine_decay(learning_rate, global_step, decay_steps, alpha=0.0, **_):
    step = min(global_step, decay_steps)
    cosine = 0.5 * (1 + math.cos(math.pi * step / decay_steps))
    return learning_rate * ((1 - alpha) * cosine + alpha)


def cosine_decay_restarts(learning_rate, global_step, first_decay_steps, t_mul=2.0, m_mul=1.0, alpha=0.0, **_):
    cf = global_step / first_decay_steps
    if t_mul == 1.0:
        i = math.floor(cf)
        cf -= i
    else:
        i = math.floor(math.log(1.0 - cf * (1.0 - t_mul)) / math.log(t_mul))
        cf = (cf - (1.0 - t_mul ** i) / (1.0 - t_mul)) / t_mul ** i
    cosine = 0.5 * (m_mul ** i) * (1 + math.cos(math.pi * cf))
    return learning_rate * ((1 - alpha) * cosine + alpha)


def noisy_linear_cosine_decay(learning_rate, global_step, decay_steps, initial_variance=1.0, variance_decay=0.55,
                              num_periods=0.5, alpha=0.0, beta=0.001, generator=None, **_):
    step = min(global_step, decay_steps)
    std = math.sqrt(initial_variance / (1 + step) ** variance_decay)
    noise = float(torch.randn((), generator=generator)) * std
    linear = 1.0 - step / decay_steps + noise
    cosine = 0.5 * (1 + math.cos(math.pi * 2.0 * num_periods * step / decay_steps))
    return learning_rate * ((alpha + linear) * cosine + beta)


DECAY_METHODS = {   # _DECAY_METHOD_NAME, DeepCrossNetwork.py:26-31
    "exponential_decay": exponential_decay, "piecewise_constant": piecewise_constant, "polynomial_decay": polynomial_decay,
    "cosine_decay": cosine_decay, "cosine_decay_restarts": cosine_decay_restarts,
    "noisy_linear_cosine_decay": noisy_linear_cosine_decay,
}


def learning_rate_decay(learning_rate_spec, global_step):
    """_learning_rate_decay(params), DeepCrossNetwork.py:422-447 (same error texts)."""
    if not isinstance(learning_rate_spec, dict):
        raise ValueError("learning_rate_spec must be a dict.")
    spec = dict(learning_rate_spec)
    spec["global_step"] = global_step
    name = spec.pop("decay_method", None)
    if name is None:                                    # _no_decay, :450-456
        if "learning_rate" not in spec:
            raise KeyError("learning rate must be provided in no_decay.")
        return spec["learning_rate"]
    if name not in DECAY_METHODS:
        raise ValueError("Unsupported learning rate name: {}. Supported names are: {}".format(name, tuple(sorted(DECAY_METHODS))))
    try:
        return DECAY_METHODS[name](**spec)
    except TypeError:
        raise TypeError("{} argument are not correct. Check again and note that global_step is omitted.".format(name))


def _weights_of(features, weight_column, like):
    """_get_weights_and_check_match_logits: a feature key, a numeric_column-like object with `.key`, or None (weight 1)."""
    if weight_column is None:
        return None
    key = weight_column if isinstance(weight_column, str) else getattr(weight_column, "key", None)
    if key is None or key not in features:
        raise ValueError("weight_column %r is not a key of features" % (weight_column,))
    w = features[key].to(device=like.device, dtype=torch.float32).reshape(-1, 1)
    if w.shape[0] != like.shape[0]:
        raise ValueError("weights shape must be [batch_size, 1]; given %s for logits %s" % (tuple(w.shape), tuple(like.shape)))
    return w


def weighted_sigmoid_cross_entropy(logits, labels, weights=None, reduction="mean"):
    """The losses of the three model_fns: unweighted = tf.nn.sigmoid_cross_entropy_with_logits(labels, logits), then
    [TF-upstream] tf.losses.compute_weighted_loss: 'mean' = sum(w * loss) / sum(w broadcast to the losses)
    (DeepCrossNetwork.py:209-225, ESMM.py:150-175), 'sum' = sum(w * loss) (the canned binary head's default
    loss_reduction=SUM used by DeepFM, deepFM.py:72,107-117), 'sum_over_batch_size' = sum(w * loss) / number of losses.
    -> (weighted_loss scalar, unweighted_loss [B,1])"""
    labels = labels.to(dtype=torch.float32).reshape(logits.shape)
    unweighted = torch.nn.functional.binary_cross_entropy_with_logits(logits, labels, reduction="none")
    w = torch.ones_like(unweighted) if weights is None else weights.to(unweighted.dtype).expand_as(unweighted)
    total = (unweighted * w).sum()
    if reduction in ("sum", "weighted_sum"):
        return total, unweighted
    if reduction in ("mean", "weighted_mean"):
        den = w.sum()
        return torch.where(den > 0, total / den.clamp_min(1e-30), torch.zeros_like(total)), unweighted
    if reduction == "sum_over_batch_size":
        return total / unweighted.numel(), unweighted
    raise ValueError("unknown loss reduction %r" % (reduction,))


def weighted_softmax_cross_entropy(logits, labels, weights=None, reduction="sum"):
    """The multi-class canned head's loss (deepFM.py:112-117, [TF-upstream] _multi_class_head_with_softmax_cross_entropy_loss):
    unweighted = sparse_softmax_cross_entropy(labels [B] or [B,1] class ids, logits [B, n_classes]) as [B,1], then
    compute_weighted_loss with the head's loss_reduction (default SUM).  -> (weighted_loss scalar, unweighted_loss [B,1])"""
    y = labels.reshape(-1).to(torch.int64)
    unweighted = torch.nn.functional.cross_entropy(logits, y, reduction="none").unsqueeze(1)
    w = torch.ones_like(unweighted) if weights is None else weights.to(unweighted.dtype).reshape(-1, 1).expand_as(unweighted)
    total = (unweighted * w).sum()
    if reduction in ("sum", "weighted_sum"):
        return total, unweighted
    if reduction in ("mean", "weighted_mean"):
        den = w.sum()
        return torch.where(den > 0, total / den.clamp_min(1e-30), torch.zeros_like(total)), unweighted
    if reduction == "sum_over_batch_size":
        return total / unweighted.numel(), unweighted
    raise ValueError("unknown loss reduction %r" % (reduction,))


def clip_by_norm_(grad, clip_norm=CLIP_NORM):
    """tf.clip_by_norm on one tensor, in place: g * clip / max(||g||_2, clip).  Sparse gradients: over their values."""
    if grad is None:
        return None
    vals = grad.coalesce().values() if grad.is_sparse else grad
    norm = torch.linalg.vector_norm(vals.float())
    scale = clip_norm / torch.clamp(norm, min=clip_norm)
    if grad.is_sparse:
        g = grad.coalesce()
        return torch.sparse_coo_tensor(g.indices(), g.values() * scale, g.shape)
    return grad.mul_(scale)


_OPTIMIZERS = {"Adam": torch.optim.Adam, "Adagrad": torch.optim.Adagrad, "SGD": torch.optim.SGD, "RMSProp": torch.optim.RMSprop}
_SPEC_KEYS = {"epsilon": "eps", "beta1": None, "beta2": None}   # tf.train.AdamOptimizer names -> torch names


class TrainStep:
    """train_op of the reference DCN (_get_train_op_fn): one call = backward + per-tensor clip + optimizer step with the
    decayed learning rate, global_step += 1.  `optimizer` is a torch optimizer class or one of the names above;
    `optimizer_spec` uses the tf.train names (epsilon, beta1, beta2)."""

    def __init__(self, model, optimizer="Adam", optimizer_spec=None, learning_rate_spec=None, l2_reg=None, l2_params=None):
        if not isinstance(optimizer_spec or {}, dict):
            raise ValueError("optimizer_spec must be a dict.")
        spec = dict(optimizer_spec or {})
        spec.pop("learning_rate", None)                               # :275-279
        self.learning_rate_spec = dict(learning_rate_spec or {"learning_rate": 0.001})
        self.global_step = 0
        kw = {}
        betas = [0.9, 0.999]
        for k, v in spec.items():
            if k == "beta1":
                betas[0] = v
            elif k == "beta2":
                betas[1] = v
            else:
                kw[_SPEC_KEYS.get(k, k) or k] = v
        cls = _OPTIMIZERS[optimizer] if isinstance(optimizer, str) else optimizer
        self._tf_adam_eps = None
        if cls is torch.optim.Adam:
            kw["betas"] = tuple(betas)
            # tf.train.AdamOptimizer applies epsilon AFTER folding the bias corrections into the step size ("epsilon hat"):
            #   var -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps_tf)
            # torch.optim.Adam divides by sqrt(v) / sqrt(1 - b2^t) + eps, i.e. the same update with
            # eps_torch = eps_tf / sqrt(1 - b2^t) (31.6x larger at t = 1 for b2 = 0.999); __call__ sets it every step.
            self._tf_adam_eps = float(kw.get("eps", 1e-8))
            self._beta2 = float(betas[1])
        self.params = [p for p in model.parameters() if p.requires_grad]
        # The embedding tables under Adam: the HIP update (ops.SparseAdam: sorted touched rows + one streaming pass over every row of w, m,
        # v -- TF's sparse Adam moves all rows every step) instead of a scatter into a dense gradient, torch's multi-tensor Adam over
        # 1.7 GB of tables and ~10 small torch kernels per table (DIR_TRAIN_HIP_ADAM=0 keeps that path).
        self.sparse_adam = []
        il = getattr(model, "input_layer", None)
        if cls is torch.optim.Adam and il is not None and hasattr(il, "fused_sparse_adam") and os.environ.get("DIR_TRAIN_HIP_ADAM", "1") == "1" \
                and not kw.get("amsgrad") and not kw.get("weight_decay"):
            self.sparse_adam, owned = il.fused_sparse_adam(betas[0], betas[1], self._tf_adam_eps, CLIP_NORM)
            owned = {id(p) for p in owned}
            self.params = [p for p in self.params if id(p) not in owned]
        self._beta1 = float(betas[0])
        lr0 = learning_rate_decay(self.learning_rate_spec, 0)
        self.optimizer = None
        if cls is torch.optim.Adam and self.params and all(p.is_cuda for p in self.params) \
                and os.environ.get("DIR_TRAIN_FUSED_OPTIMIZER", "1") == "1" and "fused" not in kw:
            try:      # one multi-tensor launch family per step instead of ~10 elementwise passes over every variable
                self.optimizer = cls(self.params, lr=lr0, fused=True, **kw)
            except (RuntimeError, TypeError):
                self.optimizer = None
        if self.optimizer is None:
            self.optimizer = cls(self.params, lr=lr0, **kw)
        self.l2_reg, self.l2_params = l2_reg, list(l2_params or [])
        self._dense_grad = {}                                         # table -> persistent all-zero dense gradient buffer
        self._torch_t = 0                                             # optimizer.step() calls so far == torch's per-parameter step counters

    def _sync_adam_step(self):
        """tf.train.AdamOptimizer's beta powers are functions of ONE global step; torch.optim.Adam keeps a step counter per parameter,
        created at 0 the first time that parameter has a gradient.  When the two disagree -- global_step restored from a checkpoint
        into a fresh optimizer, or set by hand -- every state's counter is moved to global_step (and missing states are created the
        way torch creates them), so that the step about to be taken uses t = global_step + 1 in both bias corrections and in the
        epsilon emulation above."""
        if self._tf_adam_eps is None or self._torch_t == self.global_step:
            return
        for group in self.optimizer.param_groups:
            on_dev = bool(group.get("fused")) or bool(group.get("capturable"))
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.optimizer.state[p]
                if "step" not in st:
                    st["step"] = (torch.zeros((), dtype=torch.float32, device=p.device) if on_dev else torch.tensor(0.0, dtype=torch.float32))
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group.get("amsgrad"):
                        st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"].fill_(float(self.global_step))
        self._torch_t = self.global_step

    def __call__(self, loss):
        if self.l2_reg:                                               # :181-183 (tf.nn.l2_loss = sum(w^2)/2)
            loss = loss + self.l2_reg * sum(0.5 * (w * w).sum() for w in self.l2_params)
        lr = learning_rate_decay(self.learning_rate_spec, self.global_step)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
            if self._tf_adam_eps is not None:
                group["eps"] = self._tf_adam_eps / math.sqrt(1.0 - self._beta2 ** (self.global_step + 1))
        t = self.global_step + 1
        for opt in self.sparse_adam:        # tf.train.AdamOptimizer: lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t), epsilon un-corrected
            opt.lr_t = lr * math.sqrt(1.0 - self._beta2 ** t) / (1.0 - self._beta1 ** t)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        # Tables owned by the HIP Adam whose gradient did NOT go through the sink this step (multi-hot / weighted columns: their backward
        # returns sparse .grad): the same step on the same m / v, by torch ops -- never left without an optimiser, never accumulating.
        for opt in self.sparse_adam:
            for f, p in enumerate(getattr(opt, "owned", ())):
                if p.grad is not None:
                    opt.step_table_grad(f, p.grad)
                    p.grad = None
        touched, dense = [], []
        for p in self.params:
            g = p.grad
            if g is None:
                continue
            if g.is_sparse and g.dim() == 2 and p.is_cuda:
                # The embedding tables' IndexedSlices-style gradients.  The reference's optimizers update EVERY row of such a
                # variable ([TF-upstream] Adam _apply_sparse decays all of m and v and steps all of var), i.e. a dense step with
                # zeros outside the looked-up rows: scatter-add the rows into a persistent all-zero buffer (duplicates add up,
                # which is also what the per-tensor norm is taken over), clip the touched rows in place, step, re-zero the rows.
                idx, vals = g._indices()[0], g._values()
                buf = self._dense_grad.get(p)
                if buf is None:
                    buf = self._dense_grad[p] = torch.zeros_like(p)
                vals = vals.to(buf.dtype)
                buf.index_add_(0, idx, vals)
                rows = buf.index_select(0, idx)                      # read before any write: duplicate ids copy the same row
                # ||dense gradient||^2 = sum over the touched rows r of ||S_r||^2 = sum over the ENTRIES i of <vals_i, S_row(i)>
                # (S_r = the sum of the entries of row r): a pass over the B looked-up rows instead of the whole table (64 MB each)
                norm = (vals * rows).sum().clamp_min(0).sqrt()
                scale = CLIP_NORM / torch.clamp(norm, min=CLIP_NORM)
                buf.index_copy_(0, idx, rows * scale)
                p.grad = buf
                touched.append((buf, idx))
            elif g.is_sparse or not g.is_cuda or g.dtype != torch.float32:
                g = clip_by_norm_(g)
                p.grad = g.to_dense() if g.is_sparse else g           # torch's Adam/Adagrad here take dense gradients
            else:
                dense.append(g)
        if dense:       # tf.clip_by_norm per tensor, all dense tensors in four launches (one tensor at a time: five launches EACH)
            scales = CLIP_NORM / torch.clamp(torch.stack(torch._foreach_norm(dense)), min=CLIP_NORM)
            torch._foreach_mul_(dense, list(scales.unbind()))
        self._sync_adam_step()
        self.optimizer.step()
        self._torch_t += 1
        for buf, idx in touched:
            buf.index_fill_(0, idx, 0.0)
        self.global_step += 1
        return loss.detach(), lr
