"""Meta-loop control flow restated on the oracle model (test infrastructure).

Follows the loop structure of model_zoo/mamdr.py:41-108 (MAMDR = DN + DR),
model_zoo/domain_negotiation.py:37-88, model_zoo/reptile.py:40-99 and the
alternate training of model_zoo/DeepCTR/deepctr.py:63-93.  The reference drives
these with the unseeded python `random` module (mamdr.py:46,68;
domain_negotiation.py:42), so the domain order and the DR support-domain
samples are INJECTED through `plan`, and the per-pass shuffle through
`perm_fn(domain) -> int32 permutation`.  Each function returns a trace of
(phase, domain, n_steps) tuples so the host's control flow can be compared.
"""
import numpy as np

from . import outer

F32 = np.float32


def _pass(model, data, perm_fn, d, batch_size, trace, phase, max_steps=0, accumulate_into=None, window=None):
    perm = perm_fn(d) if window is None else perm_fn(d, window)
    losses = model.train_pass(data[d], perm, batch_size, max_steps, accumulate_into)
    trace.append((phase, d, len(losses)))
    return losses


def alternate_epoch(model, data, seq, perm_fn, batch_size):
    """deepctr.py:70-78: one full pass per domain in shuffled order."""
    trace = []
    for d in seq:
        _pass(model, data, perm_fn, d, batch_size, trace, "alt")
    return trace


def dn_epoch(model, theta, data, seq, perm_fn, batch_size, meta_lr, meta_train_step=0, target=-1):
    """domain_negotiation.py:49-93: theta set once, sequential passes without reset (the target domain, if any,
    last and uncapped: :44-45,67), then theta += beta (theta~ - theta), model := theta, and with a target
    domain one more full pass of the model over it (:89-93)."""
    trace = []
    model.set_flat(theta)
    for d in seq:
        _pass(model, data, perm_fn, d, batch_size, trace, "dn", meta_train_step)
    if target >= 0:
        _pass(model, data, perm_fn, target, batch_size, trace, "dn")
    outer.dn_update(theta, model.get_flat(), meta_lr)
    model.set_flat(theta)
    if target >= 0:
        _pass(model, data, perm_fn, target, batch_size, trace, "target")
    return trace


def reptile_epoch(model, theta, data, seq, perm_fn, batch_size, meta_lr, batch_variant=False,
                  meta_train_step=0, target=-1):
    """reptile.py:45-102 (target domain: skipped in the sequence :47-48, one step on it after every
    domain's pass :82-85, a full pass at the end of the epoch :98-102)."""
    trace = []
    acc = np.zeros_like(theta)
    for d in seq:
        if target >= 0 and d == target:
            continue
        model.set_flat(theta)
        _pass(model, data, perm_fn, d, batch_size, trace, "reptile", meta_train_step)
        if target >= 0:
            _pass(model, data, perm_fn, target, batch_size, trace, "target_step", 1)
        if batch_variant:
            outer.reptile_accumulate(acc, model.get_flat(), theta)
        else:
            outer.dn_update(theta, model.get_flat(), meta_lr)
    if batch_variant:
        outer.reptile_apply(theta, acc, meta_lr)
    model.set_flat(theta)
    if target >= 0:
        _pass(model, data, perm_fn, target, batch_size, trace, "target")
    return trace


def mldg_epoch(model, theta, outer, acc, data, seq, perm_fn, batch_size, meta_lr, batch_variant=False,
               meta_train_step=0, grad_scale=1.0, windows=None, meta_domain=-1):
    """model_zoo/mldg.py:62-125: accumulate at theta over the meta-train split, outer-Adam step of the LIVE
    model with the accumulator kept, accumulate at the moved weights over the meta-val split, reset to theta,
    outer-Adam step -> new theta."""
    trace = []

    def outer_step():
        outer.apply(theta, acc, meta_lr, grad_scale)
        acc[...] = 0

    for d in seq:
        wt, wm = windows[d] if windows else (None, None)
        model.set_flat(theta)
        _pass(model, data, perm_fn, d, batch_size, trace, "mldg_train", meta_train_step, accumulate_into=acc, window=wt)
        live = model.get_flat()
        outer.apply(live, acc, meta_lr, grad_scale)
        model.set_flat(live)
        dm = d
        if meta_domain >= 0:       # train.target_domain (mldg.py:339-341)
            dm, wm = meta_domain, None
        _pass(model, data, perm_fn, dm, batch_size, trace, "mldg_meta", meta_train_step, accumulate_into=acc, window=wm)
        if not batch_variant:
            outer_step()
    if batch_variant:
        outer_step()
    model.set_flat(theta)
    return trace


def tensor_views(model, flat):
    """per-tensor views of a flat vector in the shapes the Keras variables have (the 1-d linear tables and the
    uncertainty scalars are [n, 1] there) -- what numpy's axis=-1 reductions see in the reference."""
    views, o = [], 0
    for name in model.names:
        p = model.params[name]
        shape = (-1, 1) if name in ("lin_user", "lin_item", "lin_domain", "log_var") else p.shape
        views.append(flat[o:o + p.size].reshape(shape))
        o += p.size
    return views


def pcgrad_epoch(model, outer_opt, data, seq, aux_plan, perm_fn, batch_size, meta_lr, meta_train_step=0,
                 grad_scale=1.0, windows=None):
    """model_zoo/pcgrad.py:62-124: no reset between domains; gradient of the query domain's pass, each sampled
    auxiliary domain's pass-gradient projected onto it (outer.pcgrad_project), one outer-Adam step of the model."""
    trace = []
    for d in seq:
        cur = np.zeros_like(model.get_flat())
        _pass(model, data, perm_fn, d, batch_size, trace, "pcgrad_query", meta_train_step, accumulate_into=cur,
              window=windows[d][0] if windows else None)
        for a in aux_plan[d]:
            aux = np.zeros_like(cur)
            _pass(model, data, perm_fn, a, batch_size, trace, "pcgrad_aux", 0, accumulate_into=aux,
                  window=windows[a][0] if windows else None)
            outer.pcgrad_project(tensor_views(model, cur), tensor_views(model, aux))
        live = model.get_flat()
        outer_opt.apply(live, cur, meta_lr, grad_scale)
        model.set_flat(live)
    return trace


def maml_epoch(model, theta, outer, acc, data, seq, perm_fn, batch_size, meta_lr, batch_variant=False,
               meta_train_step=0, grad_scale=1.0, windows=None, meta_domain=-1):
    """first-order MAML, model_zoo/maml.py:62-116 with meta_split "train-train": per domain reset
    to theta, inner Adam pass, meta pass that only accumulates gradients at the adapted weights,
    then (per domain, or once per epoch for "batch" names) the outer Adam step on theta."""
    trace = []

    def outer_step():
        outer.apply(theta, acc, meta_lr, grad_scale)
        acc[...] = 0

    for d in seq:
        wt, wm = windows[d] if windows else (None, None)
        model.set_flat(theta)
        _pass(model, data, perm_fn, d, batch_size, trace, "maml_train", meta_train_step, window=wt)
        dm = d
        if meta_domain >= 0:       # train.target_domain (maml.py:336-338): the meta pass runs over the target domain
            dm, wm = meta_domain, None
        _pass(model, data, perm_fn, dm, batch_size, trace, "maml_meta", meta_train_step, accumulate_into=acc, window=wm)
        if not batch_variant:
            outer_step()
    if batch_variant:
        outer_step()
    model.set_flat(theta)
    return trace


def mamdr_epoch(model, theta, phis, data, plan, perm_fn, batch_size, meta_lr, merged_method="plus",
                domain_regulation_step=0, batch_variant=False, sample_num=None, finetune_every_epoch=False):
    """mamdr.py:44-108.  plan = {"seq": [...], "dr": [(query, [support...]), ...]}
    (support list already contains the query domain when add_query_domain)."""
    trace = []
    # --- DN phase (mamdr.py:48-57)
    model.set_flat(theta)
    for d in plan["seq"]:
        _pass(model, data, perm_fn, d, batch_size, trace, "dn")
    outer.mamdr_update(theta, model.get_flat(), theta, meta_lr)
    # --- DR phase (mamdr.py:59-108)
    for query, support in plan["dr"]:
        merged = outer.merge(theta, phis[query], merged_method)
        acc = np.zeros_like(theta)
        for j in support:
            model.set_flat(merged)
            _pass(model, data, perm_fn, j, batch_size, trace, "dr_support")
            _pass(model, data, perm_fn, query, batch_size, trace, "dr_query", domain_regulation_step)
            if batch_variant:
                outer.mamdr_accumulate(acc, model.get_flat(), merged, theta, merged_method)
            else:
                outer.mamdr_update(phis[query], model.get_flat(), merged, meta_lr)
                merged = outer.merge(theta, phis[query], merged_method)
        if batch_variant:
            outer.mamdr_apply_grads(phis[query], acc, sample_num, meta_lr)
        if finetune_every_epoch:          # mamdr.py:110-143: full pass of the merged model, phi := theta~ - merged
            merged = outer.merge(theta, phis[query], merged_method)
            model.set_flat(merged)
            _pass(model, data, perm_fn, query, batch_size, trace, "dr_finetune")
            phis[query][...] = outer.mamdr_domain_weights(model.get_flat(), merged)
    return trace


def finetune_domains(model, data, start_weights, perm_fn, batch_size, epochs, patience, lr, auc_fn, domains=None,
                     set_start=None, epoch_hook=None):
    """The finetune stage (`*_finetune` model names): model_zoo/base_model.py:41-109 (every domain restarts from the
    SAME saved weights, plain SGD with `learning_rate`: base_model.py:66-71) and, for MAMDR,
    model_zoo/specific_base_model.py:99-162 (restart from best theta (+|*) best phi_d, SGD lr 0.001: :118-125).
    Per domain: `model.fit(train.repeat(), steps_per_epoch=n_step, epochs=E, validation_data=val)` = per epoch one
    shuffled pass over the train split (final partial batch kept) + one evaluation of the val split, under
      callbacks.EarlyStopping(monitor='val_AUC', patience, mode='max', min_delta=1e-4)
          -> improvement iff val_AUC - 1e-4 > best (best starts at -inf); else wait += 1, stop when wait >= patience
      callbacks.ModelCheckpoint(save_best_only=True, mode='max')
          -> weights saved iff val_AUC > best saved val_AUC (no min_delta)
    then `load_weights(chk_path)` and `evaluate(test)` (base_model.py:85-96).
    data = {"train": {d: cols}, "val": ..., "test": ...}; start_weights(d) -> flat vector; set_start (optional)
    assigns it (default model.set_flat).  Returns {d: dict(epochs, best_epoch, val_auc list, test_loss, test_auc)}
    and the trace of (phase, domain, n_steps)."""
    out, trace = {}, []
    model.use_sgd = True
    model.lr = lr
    for d in (sorted(data["train"]) if domains is None else domains):
        (set_start or model.set_flat)(start_weights(d))
        es_best, wait, ck_best, best_w, best_epoch, vals = -np.inf, 0, -np.inf, None, -1, []
        n_epochs = 0
        for epoch in range(epochs):
            _pass(model, data["train"], perm_fn, d, batch_size, trace, "finetune")
            if epoch_hook is not None:                 # tests: the weights after every epoch's pass
                epoch_hook(d, epoch, model)
            _, preds = model.evaluate(data["val"][d], batch_size)
            val_auc = float(auc_fn(data["val"][d]["label"], preds, batch_size))
            vals.append(val_auc)
            n_epochs = epoch + 1
            if val_auc > ck_best:
                ck_best, best_w, best_epoch = val_auc, model.get_flat().copy(), epoch
            if val_auc - 1e-4 > es_best:
                es_best, wait = val_auc, 0
            else:
                wait += 1
                if wait >= patience:
                    break
        model.set_flat(best_w)
        t_loss, t_preds = model.evaluate(data["test"][d], batch_size)
        out[d] = {"epochs": n_epochs, "best_epoch": best_epoch, "val_auc": vals, "test_loss": float(t_loss),
                  "test_auc": float(auc_fn(data["test"][d]["label"], t_preds, batch_size))}
    model.use_sgd = False
    return out, trace


def evaluate_domains(model, data_split, batch_size, weights_for_domain, auc_fn):
    """base_model.py:111-144 / specific_base_model.py:64-97: per-domain
    (loss, AUC); averages are plain means over domains (base_model.py:138-139)."""
    domain_loss, domain_auc = {}, {}
    for d in sorted(data_split.keys()):
        w = weights_for_domain(d)
        if w is not None:
            model.set_flat(w)
        loss, preds = model.evaluate(data_split[d], batch_size)
        domain_loss[d] = float(loss)
        domain_auc[d] = float(auc_fn(data_split[d]["label"], preds, batch_size))
    avg_loss = sum(domain_loss.values()) / len(domain_loss)
    avg_auc = sum(domain_auc.values()) / len(domain_auc)
    return avg_loss, avg_auc, domain_loss, domain_auc
