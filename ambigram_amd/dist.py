"""Sample sharding and the single end-of-batch exchange.

Independent samples shard embarrassingly: rank r takes samples r, r+W, r+2W, ... and runs them as its own batch with no
data-path collective.  The only exchange is the result gather to rank 0.  On GPUs this is RCCL over xGMI (backend
"nccl"); the CPU tests run the same code on gloo.

Two forms of the payload:
* `RunExchange` (what bench.py uses): the paths travel in run-length form.  A BFB path is a few dozen runs of
  consecutive segments on one strand (`3+4+5+`: start 3, length 3) for thousands of cells, so a rank sends ~0.5 KB per
  256-segment sample instead of ~54 KB; the per-unit cell / run counts go out with one all_gather, the runs with one
  gather, and rank 0 expands every rank's runs into cells in its own HBM (`ambi_expand_runs`, one wavefront per run, HBM
  write speed).  With 4096 samples per rank that is 1.8 MB instead of 220 MB per xGMI link.
* `PathExchange`: the expanded int32 cells themselves (all_gather of the lengths + gather of the cells).
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


class RunExchange:
    """End-of-batch gather in run-length form.  `lib` is the engine library (its `ambi_expand_runs` runs on the memory
    the tensors live in: HIP on the GPU, the host simulation in the CPU tests).  Ranks may hold DIFFERENT numbers of
    units (`n_units` is this rank's): the unit, run and cell capacities are agreed with one all_reduce(MAX) at setup and
    every rank's buffers are that large; unused unit slots carry length 0, unused run slots length 0.

    Streams: on the GPU every step (pack, the two collectives, expand and the prefix sums feeding it) is issued on ONE
    stream -- `stream` (a raw hipStream_t) if given, else torch's current stream -- by making it torch's current stream
    for the duration of the call, so the steps are ordered without host synchronisation."""

    def __init__(self, lib, n_units, local_runs, local_cells, device, world=None, rank=None, force_collectives=False):
        """force_collectives: issue the collectives even in a one-rank group (lets a 1-GPU box exercise the RCCL calls)."""
        self.lib = lib
        self.world = world if world is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.rank = rank if rank is not None else (dist.get_rank() if dist.is_initialized() else 0)
        self.collective = self.world > 1 or (force_collectives and dist.is_initialized())
        cap = torch.tensor([max(int(local_runs), 1), max(int(local_cells), 1), max(int(n_units), 1)], dtype=torch.int64, device=device)
        if self.collective:
            dist.all_reduce(cap, op=dist.ReduceOp.MAX)
        self.run_cap, self.cell_cap, self.unit_cap = int(cap[0].item()), int(cap[1].item()), int(cap[2].item())
        self.n_units = n_units
        U = self.unit_cap
        self.counts = torch.zeros(2 * U, dtype=torch.int32, device=device)            # [cells per unit | runs per unit]
        self.runs = torch.zeros(2 * self.run_cap, dtype=torch.int32, device=device)   # [start values | lengths]
        self.totals = torch.zeros(2, dtype=torch.int64, device=device)                # {runs, cells} of this rank
        multi = self.collective
        self.counts_all = torch.zeros(2 * U * self.world, dtype=torch.int32, device=device) if multi else self.counts
        self.gather_list = [torch.zeros_like(self.runs) for _ in range(self.world)] if (multi and self.rank == 0) else None
        # rank 0: the expanded paths of every rank, and the run offsets the expansion reads (kept alive until the next call)
        self.cells_all = [torch.zeros(self.cell_cap, dtype=torch.int32, device=device) for _ in range(self.world)] if self.rank == 0 else None
        self.offs = [None] * self.world

    class _on:
        """torch's current stream := the given raw stream (GPU tensors only)"""
        def __init__(self, tensor, stream):
            self.ctx = None
            if tensor.is_cuda and stream:
                self.ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(stream)))
        def __enter__(self):
            if self.ctx is not None:
                self.ctx.__enter__()
        def __exit__(self, *a):
            if self.ctx is not None:
                self.ctx.__exit__(*a)

    @staticmethod
    def _raw(tensor, stream):
        """the raw stream handle the C ABI gets: the given one, else torch's current stream on the GPU"""
        if stream:
            return int(stream)
        return torch.cuda.current_stream().cuda_stream if tensor.is_cuda else 0

    @staticmethod
    def probe(batch, n_units, device, which=1, stream=None):
        """(runs, cells) the final paths of `batch` need: a counting pass with zero capacity (nothing is written)."""
        counts = torch.zeros(2 * max(n_units, 1), dtype=torch.int32, device=device)
        dummy = torch.zeros(2, dtype=torch.int32, device=device)
        totals = torch.zeros(2, dtype=torch.int64, device=device)
        with RunExchange._on(counts, stream):
            batch.pack_runs(which, counts.data_ptr(), counts[n_units:].data_ptr(), dummy.data_ptr(), dummy[1:].data_ptr(), 0, totals.data_ptr(),
                            RunExchange._raw(counts, stream))
            if totals.is_cuda:
                torch.cuda.current_stream().synchronize()
        return int(totals[0].item()), int(totals[1].item())

    @property
    def lengths(self):
        return self.counts[: self.n_units]

    def pack(self, batch, which=1, stream=None):
        """This rank's final paths -> run-length form in self.counts / self.runs (device side)."""
        with RunExchange._on(self.counts, stream):
            batch.pack_runs(which, self.counts.data_ptr(), self.counts[self.unit_cap:].data_ptr(), self.runs.data_ptr(),
                            self.runs[self.run_cap:].data_ptr(), self.run_cap, self.totals.data_ptr(), RunExchange._raw(self.counts, stream))

    def exchange(self, stream=None):
        if self.collective:
            with RunExchange._on(self.counts, stream):
                dist.all_gather_into_tensor(self.counts_all, self.counts)
                dist.gather(self.runs, self.gather_list, dst=0)

    def expand(self, stream=None):
        """Rank 0: every rank's runs -> cells (self.cells_all[r]); no host synchronisation (unused run slots have length 0)."""
        if self.rank != 0:
            return
        import ctypes as C
        with RunExchange._on(self.runs, stream):
            raw = RunExchange._raw(self.runs, stream)
            for r in range(self.world):
                runs = self.gather_list[r] if self.collective else self.runs
                lens = runs[self.run_cap:]
                self.offs[r] = torch.cumsum(lens, 0, dtype=torch.int64) - lens      # kept: the kernel reads it asynchronously
                rc = self.lib.ambi_expand_runs(C.c_void_p(runs.data_ptr()), C.c_void_p(lens.data_ptr()), C.c_void_p(self.offs[r].data_ptr()), self.run_cap,
                                               C.c_void_p(self.cells_all[r].data_ptr()), self.cell_cap, C.c_void_p(raw))
                if rc != 0:
                    raise RuntimeError("ambi_expand_runs failed: %d" % rc)

    def collect(self):
        """Rank 0, after expand(): list (per rank) of lists (per unit SLOT, unit_cap of them; slots a rank does not use
        are empty paths) of int paths."""
        if self.rank != 0:
            return None
        out = []
        lens = self.counts_all.cpu().view(self.world, 2, self.unit_cap)[:, 0, :]
        for r in range(self.world):
            cells = self.cells_all[r].cpu()
            paths, off = [], 0
            for u in range(self.unit_cap):
                n = int(lens[r, u])
                paths.append(cells[off:off + n].tolist())
                off += n
            out.append(paths)
        return out


class _DevBytes:
    """a device address as a CUDA-array-interface object (so that torch can wrap the engine's own allocation)"""
    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def all_mode_merge(batch, device, stream=None, force=False):
    """`--all` with the orders of the batch's units dealt over the ranks (Batch.all_set_shard before run): merges the ranks'
    validity bitmaps and undefined-order flags with ONE all-reduce (bytes, MAX: the contributions are disjoint and zero
    elsewhere) and lets the engine recount.  Every rank ends with the complete lists."""
    import ctypes as C
    ptr, nbytes = batch.all_device()
    if nbytes > 0 and dist.is_initialized() and (dist.get_world_size() > 1 or force):   # force: one-rank group (exercises the RCCL call)
        if str(device).startswith("cuda"):
            ctx = torch.cuda.stream(torch.cuda.ExternalStream(int(stream))) if stream else None
            if ctx is not None:
                ctx.__enter__()
            try:
                t = torch.as_tensor(_DevBytes(ptr, nbytes), device=device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                torch.cuda.current_stream().synchronize()
            finally:
                if ctx is not None:
                    ctx.__exit__(None, None, None)
        else:
            buf = (C.c_uint8 * nbytes).from_address(ptr)
            t = torch.frombuffer(buf, dtype=torch.uint8)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    batch.all_finish()
