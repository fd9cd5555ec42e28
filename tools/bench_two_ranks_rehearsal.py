"""Rehearsal of bench.py's multi-rank path on ONE GPU.  Without a torchrun environment this script calls bench.py's own
launcher (`launch_ranks`: a `python -m torch.distributed.run` child with two ranks) with itself as the rank program; as
a rank it puts both processes on cuda:0 and carries rendezvous and collectives over gloo through host memory (RCCL
refuses two ranks on one device).  Everything else - the launcher, the environment plumbing, the reducer, the timed loop,
the per-rank times gathered to rank 0, the event-timed steps with collectives inside, rank 0 reporting while rank 1
waits - is bench.py's own code.  The throughput it prints means nothing.
usage: bench_two_ranks_rehearsal.py [--ranks N] [bench.py arguments other than --gpus]   (N ranks on the one GPU, default 2)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist
import bench


def rank_main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import image_generator_amd.trainer as trainer

    def init_gloo():
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return rank, world, 0
    trainer.init_distributed = init_gloo
    real_ar, real_ag = dist.all_reduce, dist.all_gather

    def all_reduce_through_host(t, op=dist.ReduceOp.SUM, group=None):
        if not t.is_cuda:
            return real_ar(t, op=op, group=group)
        c = t.float().cpu()
        real_ar(c, op=op, group=group)
        t.copy_(c)

    def all_gather_through_host(outs, t, group=None):
        if not t.is_cuda:
            return real_ag(outs, t, group=group)
        c = [o.cpu() for o in outs]
        real_ag(c, t.cpu(), group=group)
        for o, v in zip(outs, c):
            o.copy_(v)
    dist.all_reduce, dist.all_gather = all_reduce_through_host, all_gather_through_host
    bench.main()


if __name__ == "__main__":
    if "RANK" in os.environ:
        rank_main()
    else:
        rest, ranks = sys.argv[1:], 2
        if "--ranks" in rest:
            i = rest.index("--ranks")
            ranks = int(rest[i + 1])
            rest = rest[:i] + rest[i + 2:]
        argv = ["--gpus", str(ranks)] + rest
        # the launcher's own device count runs too; the ranks share the one GPU here, so one device is enough
        rc = bench.launch_ranks(ranks, argv, entry=os.path.abspath(__file__), need_devices=1)
        print("rehearsal finished, exit status", rc)
        sys.exit(rc)
