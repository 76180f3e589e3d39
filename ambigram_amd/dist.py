"""Sample sharding and the single end-of-batch exchange.

Independent samples shard embarrassingly: rank r takes samples r, r+W, r+2W, ... and runs them as its own batch with no
data-path collective.  The only exchange is the result gather: path lengths (int32 per unit, all_gather) followed by the
concatenated int32 paths (gather to rank 0, padded to the largest rank).  On GPUs this is RCCL over xGMI (backend
"nccl"); the CPU tests run the same code on gloo.
"""
import torch
import torch.distributed as dist


def shard(n_samples, rank, world):
    """Round-robin sample indices of `rank` (config C3: 1024 samples -> 128 per GPU at 8)."""
    return list(range(rank, n_samples, world))


class PathExchange:
    """Buffers + collectives of the end-of-batch gather.  `n_units` must be the same on every rank (pad with empty
    units otherwise); `cell_cap` is agreed with one all_reduce(MAX) at setup."""

    def __init__(self, n_units, local_cells, device, world=None, rank=None):
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        cap = torch.tensor([max(int(local_cells), 1)], dtype=torch.int64, device=device)
        if self.world > 1:
            dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        self.cell_cap = int(cap.item())
        self.n_units = n_units
        self.lengths = torch.zeros(n_units, dtype=torch.int32, device=device)
        self.cells = torch.zeros(self.cell_cap, dtype=torch.int32, device=device)
        self.total = torch.zeros(1, dtype=torch.int64, device=device)
        self.lengths_all = torch.zeros(n_units * self.world, dtype=torch.int32, device=device) if self.world > 1 else self.lengths
        self.gather_list = [torch.zeros_like(self.cells) for _ in range(self.world)] if (self.world > 1 and self.rank == 0) else None

    def exchange(self):
        """Call after the batch packed its paths into self.lengths / self.cells."""
        if self.world > 1:
            dist.all_gather_into_tensor(self.lengths_all, self.lengths)
            dist.gather(self.cells, self.gather_list, dst=0)

    def collect(self):
        """Rank 0: list (per rank) of lists (per unit) of int paths."""
        if self.rank != 0:
            return None
        out = []
        lens = self.lengths_all.cpu().view(self.world, self.n_units)
        for r in range(self.world):
            cells = (self.gather_list[r] if self.world > 1 else self.cells).cpu()
            paths, off = [], 0
            for u in range(self.n_units):
                n = int(lens[r, u])
                paths.append(cells[off:off + n].tolist())
                off += n
            out.append(paths)
        return out
