"""Synthetic Taobao-/Amazon-shaped click logs.

The real datasets are not in the reference tree and cannot be downloaded, so
throughput and parity are measured on generated logs with the published shapes
(slides p.24 Table I): number of domains / users / items and train/val/test
totals; long-tailed domain sizes (Zipf s=1 over domains, each >= one batch);
per-domain positive rate r/(1+r) with r = round(U[0.2,0.5], 2)
(dataset/Taobao/split.py:110-112,50); every domain draws users and items from
its own subset; labels are Bernoulli draws from a planted model over the
"pretrained" tables, sigma(s (<a_d,u> + <b_d,i>) + s/2 <u,i> + c_d) with a_d, b_d a
shared direction plus a domain-specific one, so that AUC is well above 0.5 and
domains are related but not identical.
Columns match the reference's csv header uid,pid,domain,label (split.py:21).
"""
import numpy as np

# slides p.24 Table I
SHAPES = {
    "taobao10": dict(name="Taobao", split="split_by_theme_10", n_domain=10, n_user=23778, n_item=6932,
                     n_train=92137, n_val=37645, n_test=43502, pretrained=True),
    "taobao20": dict(name="Taobao", split="split_by_theme_20", n_domain=20, n_user=58190, n_item=16319,
                     n_train=243592, n_val=96591, n_test=106500, pretrained=True),
    "taobao30": dict(name="Taobao", split="split_by_theme_30", n_domain=30, n_user=99143, n_item=29945,
                     n_train=394805, n_val=151369, n_test=179252, pretrained=True),
    "amazon6": dict(name="Amazon", split="split_by_category_6", n_domain=6, n_user=445789, n_item=172653,
                    n_train=9968333, n_val=3372666, n_test=3585877, pretrained=False),
    "amazon13": dict(name="Amazon", split="split_by_category_13", n_domain=13, n_user=502222, n_item=215403,
                     n_train=11999607, n_val=4100756, n_test=4339523, pretrained=False),
}


def _domain_sizes(total, n_domain, min_size, rs):
    w = 1.0 / np.arange(1, n_domain + 1)
    w = w[rs.permutation(n_domain)]
    sizes = np.maximum(np.floor(w / w.sum() * total).astype(np.int64), min_size)
    # give the rounding remainder (or take the excess) to/from the largest domain
    sizes[np.argmax(sizes)] += total - sizes.sum()
    if sizes.min() < 1:
        raise ValueError("total %d too small for %d domains" % (total, n_domain))
    return sizes


def generate(shape="taobao10", batch_size=1024, seed=123, scale=1.0, emb_dim=128, signal=10.0, row_scale=1.0,
             splits=("train", "val", "test"), threads=None, hot=None):
    """returns dict(tables, data, info).  scale < 1 shrinks users/items/rows (tests);
    row_scale < 1 shrinks only the number of rows (full-size tables, shorter epochs).
    splits: which splits to draw (bench.py binds the train split only; skipping the others changes the random
    stream of the later domains, i.e. gives a different sample of the same distribution).
    threads: worker threads for the planted model's logits (default: up to 16 of the visible cores).
    hot: None or dict(users=K, items=K, share=s): a share s of every domain's rows draws its user / item from the
    first K ids of the domain's subsets (a separate random stream: the other rows are the ones hot=None gives) --
    short runs over FULL-SIZE tables in which part of the rows repeats often enough to be learnt while the rest of
    the tables is touched rarely or never (the trained-model parity tests at the Amazon table sizes)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    if threads is None:
        try:
            threads = min(16, len(os.sched_getaffinity(0)))
        except Exception:
            threads = min(16, os.cpu_count() or 1)
    pool = ThreadPoolExecutor(threads) if threads > 1 else None
    try:
        return _generate(shape, batch_size, seed, scale, emb_dim, signal, row_scale, tuple(splits), pool, hot)
    finally:
        if pool is not None:
            pool.shutdown()


def _generate(shape, batch_size, seed, scale, emb_dim, signal, row_scale, splits, pool, hot=None):
    spec = dict(SHAPES[shape]) if isinstance(shape, str) else dict(shape)
    rs = np.random.RandomState(seed)
    D = spec["n_domain"]
    n_user = max(int(spec["n_user"] * scale), 4 * D)
    n_item = max(int(spec["n_item"] * scale), 4 * D)
    user_emb = (rs.standard_normal((n_user, emb_dim)) * 0.1).astype(np.float32)
    item_emb = (rs.standard_normal((n_item, emb_dim)) * 0.1).astype(np.float32)
    dir_shared = rs.standard_normal((2, emb_dim)) / np.sqrt(emb_dim)
    min_size = max(1, int(batch_size * min(1.0, scale))) if scale < 1.0 else batch_size
    sizes = {}
    for split in ("train", "val", "test"):
        ms = min_size if split == "train" else max(1, min_size // 4)
        total = max(int(spec["n_" + split] * scale * row_scale), D * ms)
        sizes[split] = _domain_sizes(total, D, ms, np.random.RandomState(seed + 1))
    data = {"train": {}, "val": {}, "test": {}}
    info = {"n_user": n_user, "n_item": n_item}
    totals = {"train": 0, "val": 0, "test": 0}
    for d in range(D):
        r = round(rs.uniform(0.2, 0.5), 2)
        rate = r / (1.0 + r)
        bias = np.log(rate / (1.0 - rate))
        # per-domain subsets (overlapping across domains)
        n_u_d = max(8, int(n_user * min(1.0, 3.0 / D)))
        n_i_d = max(8, int(n_item * min(1.0, 3.0 / D)))
        users = rs.choice(n_user, n_u_d, replace=False)
        items = rs.choice(n_item, n_i_d, replace=False)
        info[d] = {"ctr_ratio": r}
        dir_d = dir_shared + 0.7 * rs.standard_normal((2, emb_dim)) / np.sqrt(emb_dim)
        for split in splits:
            n = int(sizes[split][d])
            uid = users[rs.randint(0, n_u_d, n)].astype(np.int32)
            pid = items[rs.randint(0, n_i_d, n)].astype(np.int32)
            if hot:
                hrs = np.random.RandomState((seed * 1009 + d * 7 + ("train", "val", "test").index(split)) % (2 ** 31))
                is_hot = hrs.uniform(size=n) < hot["share"]
                uid = np.where(is_hot, users[hrs.randint(0, min(hot["users"], n_u_d), n)], uid).astype(np.int32)
                pid = np.where(is_hot, items[hrs.randint(0, min(hot["items"], n_i_d), n)], pid).astype(np.int32)
            logit = np.empty(n, np.float64)

            def chunk(c0, uid=uid, pid=pid, logit=logit, dir_d=dir_d, bias=bias, n=n):
                sl = slice(c0, min(n, c0 + (1 << 18)))       # chunked: the gathered rows are 1 KiB per sample
                ue, ie = user_emb[uid[sl]].astype(np.float64), item_emb[pid[sl]].astype(np.float64)
                logit[sl] = signal * (ue @ dir_d[0] + ie @ dir_d[1]) + 0.5 * signal * np.einsum("ij,ij->i", ue, ie) + bias
            # (chunks are independent and each is computed by the same calls whatever the thread count: the labels
            # do not depend on it; numpy's gathers / contractions release the GIL)
            starts = range(0, n, 1 << 18)
            if pool is not None and len(starts) > 1:
                list(pool.map(chunk, starts))
            else:
                for c0 in starts:
                    chunk(c0)
            label = (rs.uniform(size=n) < 1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
            data[split][d] = {"uid": uid, "pid": pid, "domain": np.full(n, d, np.int32), "label": label}
            info[d]["n_" + split] = n
            totals[split] += n
    info["total_train"], info["total_val"], info["total_test"] = totals["train"], totals["val"], totals["test"]
    tables = {"user_emb": user_emb, "item_emb": item_emb}
    return {"spec": spec, "tables": tables, "data": data, "info": info, "n_user": n_user, "n_item": n_item,
            "n_domain": D, "batch_size": batch_size}
