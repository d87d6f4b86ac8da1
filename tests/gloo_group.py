"""torch.distributed (gloo) behind the interface of catch_amd.netstore.TcpGroup: what the CPU tests drive the
product's multi-rank helpers with (world size 2, gloo).  Test infrastructure: the product imports no torch and
rendezvouses over catch_amd.netstore only (it lived in the product tree, behind CATCHHIP_RENDEZVOUS=gloo, until round 5)."""


class GlooGroup:
    """rank / size / barrier / allgather / broadcast / allreduce over torch.distributed."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.size = dist.get_rank(), dist.get_world_size()

    def allgather(self, obj):
        out = [None] * self.size
        self.dist.all_gather_object(out, obj)
        return out

    def broadcast(self, obj, src=0):
        box = [obj if self.rank == src else None]
        self.dist.broadcast_object_list(box, src=src)
        return box[0]

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, arr, op="sum"):
        import numpy as np
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return t.numpy()

    def close(self):
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
