"""Deterministic random streams shared by the oracle and the HIP path (test infrastructure).

TF's RNG streams (dropout masks, tf.data shuffle buffer) cannot be reproduced
(SURVEY.md section 7 hard part 3; run.py:26 seeds TF only), so the build
*defines* counter-based streams and both sides implement them in exact integer
arithmetic:

* dropout keep-mask: one 32-bit hash per (seed, step, layer, row, col)
  -- mirrors mamdr_amd/csrc/mamdr_device.h `mamdr_dropout_u32`.
* shuffle buffer: tf.data `shuffle(buffer_size)` semantics
  (utils/dataset.py:27-37: shuffle_buffer_size=10000, reshuffled every pass),
  driven by splitmix64 -- mirrors mamdr_amd/csrc/mamdr_host.cpp
  `mamdr_shuffle_perm`.
"""
import numpy as np

_M32 = np.uint32(0xFFFFFFFF)
GOLDEN = 0x9E3779B9
C1 = 0x85EBCA6B
C2 = 0xC2B2AE35


def fmix32(h):
    """murmur3 finaliser on uint32 arrays (wrapping arithmetic)."""
    h = np.asarray(h, dtype=np.uint32).copy()
    with np.errstate(over="ignore"):
        h ^= h >> np.uint32(16)
        h *= np.uint32(C1)
        h ^= h >> np.uint32(13)
        h *= np.uint32(C2)
        h ^= h >> np.uint32(16)
    return h


def dropout_layer_key(seed, step, layer):
    """Wave-uniform key for one (seed, step, layer)."""
    with np.errstate(over="ignore"):
        k0 = fmix32(np.uint32(seed & 0xFFFFFFFF) + np.uint32(GOLDEN) * np.uint32((step + 1) & 0xFFFFFFFF))
        k1 = fmix32(k0 ^ (np.uint32(C1) * np.uint32(layer + 1)))
    return np.uint32(k1)


def dropout_u32(seed, step, layer, n_rows, n_cols):
    """uint32 [n_rows, n_cols]: element (r, c) of the hash stream."""
    key = dropout_layer_key(seed, step, layer)
    e = (np.arange(n_rows, dtype=np.uint32)[:, None] * np.uint32(n_cols)
         + np.arange(n_cols, dtype=np.uint32)[None, :])
    with np.errstate(over="ignore"):
        return fmix32(key + np.uint32(GOLDEN) * e)


def dropout_threshold(rate):
    """keep iff u >= threshold; P(keep) = 1 - rate."""
    t = int(float(rate) * 4294967296.0)
    return np.uint32(min(max(t, 0), 0xFFFFFFFF))


def dropout_mask(seed, step, layer, n_rows, n_cols, rate):
    """float32 {0,1} keep mask."""
    if rate <= 0.0:
        return np.ones((n_rows, n_cols), np.float32)
    return (dropout_u32(seed, step, layer, n_rows, n_cols) >= dropout_threshold(rate)).astype(np.float32)


def splitmix64(state):
    """returns (new_state, output) with python ints mod 2^64."""
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z = z ^ (z >> 31)
    return state, z


def shuffle_perm(n, buffer_size, seed):
    """tf.data shuffle-buffer order of range(n) (pure-python loop: small n only).

    Fill the buffer with the first min(n, buffer_size) elements; each output is a
    uniformly chosen slot, refilled with the next input element, or, once the
    input is exhausted, with the buffer's last element.
    """
    out = np.empty(n, dtype=np.int32)
    buf = list(range(min(n, max(1, buffer_size))))
    nxt = len(buf)
    state = seed & 0xFFFFFFFFFFFFFFFF
    for i in range(n):
        state, z = splitmix64(state)
        j = ((z >> 32) * len(buf)) >> 32
        out[i] = buf[j]
        if nxt < n:
            buf[j] = nxt
            nxt += 1
        else:
            buf[j] = buf[-1]
            buf.pop()
    return out
