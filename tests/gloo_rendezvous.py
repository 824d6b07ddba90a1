"""Test helper: the `launch.Rendezvous` interface over a torch.distributed gloo group (CPU), so that the sharded
samplers are also covered on the transport the driver's launcher provides.  Not part of the package."""
import os
import struct


class GlooRendezvous:
    def __init__(self, rank, world, port):
        import torch.distributed as dist
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        self.dist, self.rank, self.world = dist, rank, world

    def allgather(self, payload):
        out = [None] * self.world
        self.dist.all_gather_object(out, bytes(payload))
        return out

    def bcast(self, payload, src=0):
        return self.allgather(payload if self.rank == src else b"")[src]

    def barrier(self):
        self.dist.barrier()

    def max(self, value):
        return max(struct.unpack("<d", b)[0] for b in self.allgather(struct.pack("<d", float(value))))

    def exchange_id(self, uid):
        return self.bcast(uid or b"", src=0)

    def close(self):
        self.dist.destroy_process_group()
