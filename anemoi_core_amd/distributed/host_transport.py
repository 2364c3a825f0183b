"""DEBUG transport (never selected by default): lets several ranks share ONE GPU (RCCL refuses two ranks per device) by
carrying the communication calls of the hot path over a gloo group through host memory.  The kernels, the packing, the halo
plans and the segmented hipGraph capture are the product's; only the wire is swapped.  Used by the multi-rank GPU tests and
by ``bench.py`` under ``ANEMOI_BENCH_TRANSPORT=host`` to exercise the N > 1 code path on a 1-GPU box; the product wire is
RCCL (``distributed/primitives.py``)."""
import torch
import torch.distributed as dist


def install():
    from . import primitives as P
    from ..utils.segments import collective

    def a2a(recv, send, recv_counts, send_counts, group):
        def fn():
            s = send.reshape(send.shape[0], -1).view(torch.uint8).cpu()
            r = torch.empty((recv.shape[0], s.shape[1]), dtype=torch.uint8)
            dist.all_to_all_single(r, s, output_split_sizes=recv_counts, input_split_sizes=send_counts, group=group)
            recv.reshape(recv.shape[0], -1).view(torch.uint8).copy_(r)
        collective(fn)

    def allgather(out, inp, group):
        def fn():
            s = inp.reshape(inp.shape[0], -1).view(torch.uint8).cpu()
            r = torch.empty((out.shape[0], s.shape[1]), dtype=torch.uint8)
            dist.all_gather_into_tensor(r, s, group=group)
            out.reshape(out.shape[0], -1).view(torch.uint8).copy_(r)
        collective(fn)

    def allreduce(x, group):
        def fn():
            h = x.detach().float().cpu()
            dist.all_reduce(h, group=group)
            x.copy_(h.to(x.dtype))
        collective(fn)

    P._all_to_all_single = a2a
    P._all_gather_into_tensor = allgather
    P._all_reduce_sum = allreduce
