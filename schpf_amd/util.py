"""The one helper of the reference's schpf/util.py that the CAVI loop uses."""
import numpy as np

__all__ = ["minibatch_ix_generator"]


def minibatch_ix_generator(ncells, batchsize):
    """Endless stream of cell-index batches: one shuffle of 0..ncells-1 (NumPy global RNG, drawn
    at the first `next`), walked cyclically in strides of `batchsize`; a stride that runs off the
    end wraps around to the front (reference util.py:218-231)."""
    assert ncells >= batchsize
    order = np.arange(ncells)
    np.random.shuffle(order)
    start = 0
    while True:
        end = start + batchsize
        if end > ncells:
            end %= ncells
            batch = np.hstack([order[start:], order[:end]])
        else:
            batch = order[start:end]
        start = end % ncells
        yield batch
