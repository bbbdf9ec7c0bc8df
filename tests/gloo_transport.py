"""TEST INFRASTRUCTURE: the transport interface of xmhw_amd.sharded (rank, size, allgather_i64,
allgather_u8, agree, gather_columns, gather_rows) on a torch.distributed gloo group with host
arrays, so that the sharding logic runs on CPU with 2 or 3 ranks.  Device buffers handed over by a
real HIP stage are copied to the host first (two ranks may share one GPU in the tests)."""
import numpy as np
import torch
import torch.distributed as dist

from xmhw_amd.exception import XmhwException


class GlooTransport:
    def __init__(self, group=None):
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)

    def allgather_i64(self, value):
        self.int_collectives = getattr(self, "int_collectives", 0) + 1
        mine = torch.tensor([int(value)], dtype=torch.int64)
        parts = [torch.zeros_like(mine) for _ in range(self.size)]
        dist.all_gather(parts, mine, group=self.group)
        return np.array([int(p.item()) for p in parts], dtype=np.int64)

    def allgather_u8(self, arr):
        self.mask_collectives = getattr(self, "mask_collectives", 0) + 1
        arr = np.ascontiguousarray(arr, dtype=np.uint8)
        counts = self.allgather_i64(arr.shape[0])
        width = int(max(int(counts.max()), 1))
        mine = torch.zeros(width, dtype=torch.uint8)
        mine[: arr.shape[0]] = torch.from_numpy(arr)
        parts = [torch.zeros_like(mine) for _ in range(self.size)]
        dist.all_gather(parts, mine, group=self.group)
        return [parts[r][: int(counts[r])].numpy().copy() for r in range(self.size)]

    def agree(self, error=None, value=0):
        vals = self.allgather_i64(-1 if error is not None else int(value))
        if (vals < 0).any():
            if error is not None:
                raise error
            raise XmhwException(f"sharded run aborted: rank(s) {[int(r) for r in np.nonzero(vals < 0)[0]]} failed")
        return vals

    def gather_columns(self, block, rows, dst=0, counts=None):
        self.bulk_collectives = getattr(self, "bulk_collectives", 0) + 1
        if hasattr(block, "to_array"):           # a DeviceBuffer from the HIP stage
            cols = (block.nbytes // (8 * rows) if rows else 0) if counts is None else int(counts[self.rank])
            block = block.to_array((rows, cols), np.float64)
        block = np.ascontiguousarray(block, dtype=np.float64).reshape(rows, -1)
        if counts is None:
            counts = self.allgather_i64(block.shape[1])
        counts = np.asarray(counts, dtype=np.int64)
        assert int(counts[self.rank]) == block.shape[1]
        width = int(max(int(counts.max()), 1))
        pad = torch.zeros((rows, width), dtype=torch.float64)
        pad[:, : block.shape[1]] = torch.from_numpy(block)
        out = [torch.empty_like(pad) for _ in range(self.size)] if self.rank == dst else None
        dist.gather(pad, out, dst=dst, group=self.group)
        if self.rank != dst:
            return None
        return np.concatenate([out[r][:, : int(counts[r])].numpy() for r in range(self.size)], axis=1)

    def gather_rows(self, table, dst=0):
        table = np.ascontiguousarray(table, dtype=np.float64)
        flat = self.gather_columns(table.reshape(1, -1), 1, dst)
        return None if flat is None else flat.reshape(-1, table.shape[1])
