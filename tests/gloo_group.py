"""torch.distributed (gloo) behind the small group interface of nbmf_mm_amd._rendezvous -- TESTS ONLY: the
world-size-2 gloo tests drive the product's multi-GPU host logic through it; the product itself meets through
the standard-library group and never imports PyTorch."""
import numpy as np


class GlooGroup:
    def __init__(self, rank, world, port):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        self.rank, self.world = rank, world

    def all_gather(self, obj):
        table = [None] * self.world
        self._dist.all_gather_object(table, obj)
        return table

    def broadcast(self, obj, src=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        self._dist.barrier()

    def all_reduce(self, arr, op="sum"):
        ops = {"sum": self._dist.ReduceOp.SUM, "min": self._dist.ReduceOp.MIN, "max": self._dist.ReduceOp.MAX}
        self._dist.all_reduce(self._torch.from_numpy(arr), op=ops[op])
        return arr

    def agree(self, ok):
        flag = np.array([1 if ok else 0], dtype=np.int32)
        return bool(self.all_reduce(flag, "min")[0])

    def max_float(self, x):
        return float(self.all_reduce(np.array([float(x)]), "max")[0])

    def close(self):
        self._dist.destroy_process_group()


def make_group(kind, rank, world, port):
    """kind "gloo": torch.distributed; kind "stdlib": the product's own rendezvous over TCP loopback."""
    if kind == "gloo":
        return GlooGroup(rank, world, port)
    from nbmf_mm_amd import _rendezvous
    return _rendezvous.Group(rank, world, ("tcp", "127.0.0.1", port), secret=b"tests-%d" % port)
