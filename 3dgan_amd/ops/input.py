"""Per-replica input sharding (reference: ops/input.py:11-25 batch_slice)."""


def batch_slice(x, batch_size, slice_index, name=None):
    """Rows [slice_index*batch_size, (slice_index+1)*batch_size) of a [B*n, ...] batch:
    tower i of the reference == rank i here."""
    return x[slice_index * batch_size:(slice_index + 1) * batch_size]
