"""Rehearsal of bench.py's multi-rank control flow on ONE GPU: two processes, both on cuda:0, rendezvous and collectives
over gloo through host memory (RCCL refuses two ranks on one device).  Everything else - the reducer, the timed loop,
the max-over-ranks, the event-timed steps with collectives inside, rank 0 reporting while rank 1 waits - is bench.py's
own code.  The throughput it prints means nothing.  usage: bench_two_ranks_rehearsal.py [bench.py arguments]"""
import os, sys, socket
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, torch.distributed as dist, torch.multiprocessing as mp


def worker(rank, world, port, argv):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import image_generator_amd.trainer as trainer

    def init_gloo():
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        return rank, world, 0
    trainer.init_distributed = init_gloo
    real = dist.all_reduce

    def through_host(t, op=dist.ReduceOp.SUM, group=None):
        if not t.is_cuda:
            return real(t, op=op, group=group)
        c = t.float().cpu()
        real(c, op=op, group=group)
        t.copy_(c)
    dist.all_reduce = through_host
    import bench
    sys.argv = ["bench.py", "--gpus", str(world)] + argv
    bench.main()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(2, port, sys.argv[1:]), nprocs=2, join=True)
    print("rehearsal finished")
