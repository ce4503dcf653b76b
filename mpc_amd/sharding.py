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
    `part`.  The first parts take one granule more when the granules do not divide evenly; the last part ends at
    `total`.  Plain arithmetic, no library and no GPU: the library's planner (mpcx_plan_split, used by the C++ engines
    and the group entry points) cuts in the same places, which a GPU-tier test checks over a grid of arguments."""
    granule = max(int(granule), 1)
    if parts < 1 or not 0 <= part < parts or total < 0:
        raise ValueError(f"split_range({total}, {parts}, {part}, {granule}): bad arguments")
    units = -(-total // granule)
    base, rem = divmod(units, parts)
    u0 = part * base + min(part, rem)
    un = base + (1 if part < rem else 0)
    first, end = min(u0 * granule, total), min((u0 + un) * granule, total)
    return first, end - first


def matmul_row_block(r_total, world, rank):
    """(row0, rows) of the X / Z row-block owned by `rank`."""
    return split_range(r_total, world, rank)


def aby3_matmul_pop_plan(r_total, l, c):
    """Pop-stream offsets (per shared stream, relative to the engine's position before the calls) of
    share_matrix(X), share_matrix(Y), matrix_mul on fresh engines: each call consumes one word per
    element of its output from both of the party's shared streams."""
    return {"share_x": 0, "share_y": r_total * l, "matmul": r_total * l + l * c,
            "total": r_total * l + l * c + r_total * c}
