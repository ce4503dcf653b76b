"""Host-side sharding plans: how one party's local work is split over the GPUs of a node.

Nothing here touches a GPU.  Two kinds of work shard naturally (no exchange inside any kernel):

* matmul row-blocks: rank g owns rows [row0, row0+rows) of X (both shares), Y is replicated; the
  row-major mask words of the block are pops [row0*c, (row0+rows)*c) of both streams, so a block only
  needs its row offset.  The blocks of Z0 (the message to the previous party) are gathered with one
  all-gather.
* element ranges: rank g owns elements [i0, i0+n) of every vector; kernels take (n_total, i0, n)
  because buffer offsets in SecurennEngine depend on the whole vector's size and pop-stream sections
  are laid out per whole vector.
"""


def split_range(total, parts, part, granule=1):
    """Contiguous, balanced split of range(total) into `parts`, in multiples of `granule`; returns (first, count) of
    `part`.  The arithmetic is the library's own planner (mpcx_plan_split), the one the C++ engines and the group
    entry points use, so that every layer cuts a vector in the same places."""
    import ctypes

    from .lib import load_library
    first, count = ctypes.c_uint64(), ctypes.c_uint64()
    rc = load_library().mpcx_plan_split(ctypes.c_uint64(total), ctypes.c_int(parts), ctypes.c_int(part),
                                        ctypes.c_uint64(granule), ctypes.byref(first), ctypes.byref(count))
    if rc:
        raise ValueError(f"split_range({total}, {parts}, {part}, {granule}): bad arguments")
    return first.value, count.value


def matmul_row_block(r_total, world, rank):
    """(row0, rows) of the X / Z row-block owned by `rank`."""
    return split_range(r_total, world, rank)


def aby3_matmul_pop_plan(r_total, l, c):
    """Pop-stream offsets (per shared stream, relative to the engine's position before the calls) of
    share_matrix(X), share_matrix(Y), matrix_mul on fresh engines: each call consumes one word per
    element of its output from both of the party's shared streams."""
    return {"share_x": 0, "share_y": r_total * l, "matmul": r_total * l + l * c,
            "total": r_total * l + l * c + r_total * c}
