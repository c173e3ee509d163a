"""Shared helpers for the parity tests."""
import numpy as np

from mplan2vdl_amd import datagen


def lineitem(names, n, seed=datagen.SEED, row0=0):
    return {name: datagen.generate(datagen.LINEITEM[name], row0, n, seed) for name in names}


def oracle_run(text, cols):
    import oracle

    o = oracle.Oracle()
    for k, v in cols.items():
        o.add_column(k, v)
    try:
        return o.run(text)["results"]
    finally:
        o.close()


def engine_with(cols, device=0):
    import mplan2vdl_amd as m

    e = m.Engine(device=device)
    for k, v in cols.items():
        e.upload(k, v)
    return e


def prog(*lines):
    return "\n".join(lines) + "\n"


def rand_cols(rng, n, spec):
    """spec: {name: (dtype, lo, hi)} -> uniform random integer columns."""
    return {k: rng.integers(lo, hi + 1, size=n, dtype=np.int64).astype(dt) for k, (dt, lo, hi) in spec.items()}
