"""torch.distributed plumbing of the multi-GPU path (one process per GPU, RCCL over xGMI on the GPU box, gloo in
the CPU tests).  The path shards by genome (sketching) and by query row block (distances); its ONE exchange step
is an all-gather of every rank's packed sketches so that each rank can index all references.

Everything here is static-shape tensor code on the caller's device: no size ever visits the host, so the step
stays free of host synchronisation.
"""
import torch


class SketchGather:
    """All-gather fixed-size padded sketch units and compact them into one CSR on the device.

    every rank contributes   off_l  int64[G+1]  exclusive prefix of its G sketch sizes
                             ids_l  int32[cap]  its ids, only the first off_l[G] are meaningful
    and receives             roff   int64[world*G+1], rids int32[world*cap] (first roff[-1] meaningful):
    genome r*G+g of the global numbering is genome g of rank r.
    """

    def __init__(self, world, G, cap, device):
        self.world, self.G, self.cap = world, G, cap
        self.off_all = torch.zeros(world * (G + 1), dtype=torch.int64, device=device)
        self.ids_all = torch.zeros(world * cap, dtype=torch.int32, device=device)
        self.roff = torch.zeros(world * G + 1, dtype=torch.int64, device=device)
        self.rids = torch.zeros(world * cap, dtype=torch.int32, device=device)
        self._j = torch.arange(world * cap, device=device, dtype=torch.int64)

    def __call__(self, off_l, ids_l, group=None):
        import torch.distributed as dist
        dist.all_gather_into_tensor(self.off_all, off_l, group=group)
        dist.all_gather_into_tensor(self.ids_all, ids_l, group=group)
        return self.compact()

    def compact(self):
        w, G, cap = self.world, self.G, self.cap
        o = self.off_all.view(w, G + 1)
        sizes = (o[:, 1:] - o[:, :-1]).reshape(-1)
        self.roff[0] = 0
        self.roff[1:] = torch.cumsum(sizes, 0)
        tot = o[:, G]                                   # ids held by each rank
        ends = torch.cumsum(tot, 0)
        starts = ends - tot
        rk = torch.searchsorted(ends, self._j, right=True).clamp_(max=w - 1)
        src = (self._j - starts[rk] + rk * cap).clamp_(min=0, max=w * cap - 1)
        torch.index_select(self.ids_all, 0, src, out=self.rids)
        return self.roff, self.rids


def query_block(rank, G):
    """rows of the global all-pairs matrix this rank computes: its own genomes"""
    return rank * G, (rank + 1) * G
