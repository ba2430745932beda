"""Outer (meta) parameter updates restated in numpy fp32 (test infrastructure).

The reference does these on the host in numpy between `K.batch_get_value`
(model_zoo/maml.py:194) and `SetVarOp.__call__` (utils/tool.py:36-45).  Each
function names the reference lines it follows; every arithmetic step is one
fp32 rounding (numpy keeps fp32 when an fp32 array meets a python scalar), in
the reference's order, so the HIP kernels can be compared bit-for-bit.
Pinned by tests/golden/outer_goldens.npz (generated from the reference's own
methods by tests/golden/make_outer_goldens.py).
"""
import numpy as np

F32 = np.float32


def dn_update(theta, new, meta_lr):
    """model_zoo/domain_negotiation.py:118-123 == model_zoo/reptile.py:127-132:
    old += (new - old) * meta_lr."""
    theta += ((new - theta) * F32(meta_lr)).astype(F32)
    return theta


def reptile_accumulate(acc, new, old):
    """model_zoo/reptile.py:134-137: acc += new - old."""
    acc += (new - old).astype(F32)
    return acc


def reptile_apply(theta, acc, meta_lr):
    """model_zoo/reptile.py:139-142: old += acc * meta_lr; acc = 0."""
    theta += (acc * F32(meta_lr)).astype(F32)
    acc[...] = 0
    return theta


def merge(theta, phi, method="plus"):
    """model_zoo/specific_base_model.py:164-172."""
    if method == "plus":
        return (theta + phi).astype(F32)
    elif method == "times":
        return (theta * phi).astype(F32)
    raise ValueError(method)


def mamdr_update(update, new, old, meta_lr):
    """model_zoo/mamdr.py:173-180: update += (new - old) * meta_lr, with
    old = merged weights (DR) or `update` itself (DN)."""
    update += ((new - old) * F32(meta_lr)).astype(F32)
    return update


def mamdr_domain_weights(new, merged):
    """model_zoo/mamdr.py:168-171: phi = new - merged."""
    return (new - merged).astype(F32)


def mamdr_accumulate(acc, new, old, shared, method="plus", train_step=1):
    """model_zoo/mamdr.py:182-191 (batch variant)."""
    if method == "plus":
        acc += ((new - old) / F32(train_step)).astype(F32)
    elif method == "times":
        acc += ((new - old) * shared / F32(train_step)).astype(F32)
    return acc


def mamdr_apply_grads(old, grads, sample_num, meta_lr):
    """model_zoo/mamdr.py:193-196: old += grads / sample_num * meta_lr; grads = 0."""
    old += (grads / F32(sample_num) * F32(meta_lr)).astype(F32)
    grads[...] = 0
    return old


def pcgrad_project(final, aux):
    """model_zoo/pcgrad.py:152-160 with `final_grads is current_grads`, as its train loop calls it
    (pcgrad.py:107-124).  Per tensor and per slice along the last axis (rows of a kernel / table, the whole
    vector of a bias): d = <cur, aux>; where d > 0 the auxiliary gradient loses d / ||cur|| times cur (the
    reference divides by the norm, not its square, and projects where the dot product is POSITIVE); the result
    is added to the running gradient.  fp32 throughout; numpy's own reductions (pairwise summation) are part of
    the pinned arithmetic.  Mutates both lists in place, returns `final`."""
    for k in range(len(final)):
        cur, a = final[k], aux[k]
        flat_cur = cur.reshape(-1, cur.shape[-1])
        flat_aux = a.reshape(-1, a.shape[-1])
        dots = np.sum(flat_cur * flat_aux, axis=-1)
        hit = dots > 0
        if hit.any():
            norms = np.sqrt(np.add.reduce(flat_cur[hit] * flat_cur[hit], axis=-1))
            flat_aux[hit] -= (dots[hit] / norms)[:, None] * flat_cur[hit]
        flat_cur += flat_aux
    return final


def moving_average_update(unbiased, biased, value, momentum, local_step):
    """`K.moving_average_update(ag, g, 0.999)` of average_meta_grad == "moving_mean" (model_zoo/maml.py:219-220,
    mldg.py:222-223, pcgrad.py:229-230), in place on float32 arrays; returns the new local_step.

    TensorFlow is a dependency the reference tree does not hold (requirements: tensorflow 1.12); this restates
    its published code -- parity unpinned.  tensorflow/python/keras/backend.py (r1.12) moving_average_update:
        moving_averages.assign_moving_average(x, value, momentum, zero_debias=True)
    tensorflow/python/training/moving_averages.py (r1.12) assign_moving_average / _zero_debias:
        decay = convert_to_tensor(1.0 - decay)            # python double, then a float32 tensor
        biased       -= (biased - value) * decay          # hidden variable "biased", zeros at creation
        local_step   += 1                                 # hidden variable "local_step"
        unbiased_var -= unbiased_var - biased / (1 - pow(1.0 - decay, local_step))
    The hidden variables belong to the accumulator variable and are NOT touched by `clear_grads`
    (maml.py:203): the average runs over every meta batch since the start of training."""
    decay = F32(1.0 - momentum)
    biased -= (biased - value) * decay
    local_step = int(local_step) + 1
    denom = F32(1.0) - np.power(F32(1.0) - decay, F32(local_step), dtype=F32)
    unbiased -= unbiased - biased / denom
    return local_step
