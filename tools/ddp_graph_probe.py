"""One-rank RCCL group: eager data-parallel steps, then capture + replays, with phase markers on stderr (diagnostic for
the watchdog / capture interplay).   python tools/ddp_graph_probe.py [drain_seconds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "intro-tc-vae_amd"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ["ITCV_DDP_GRAPH"] = "1"
import torch
import torch.distributed as dist


def mark(s):
    print(f"[probe {time.time() % 1000:8.3f}] {s}", file=sys.stderr, flush=True)


def main():
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    import models
    from hipvae import ddp
    from solvers.intro_tc import IntroTCSovler
    dev = torch.device("cuda:0")

    class DS:
        def __len__(self):
            return 1000

    ddp.init(sync_bn=True, force=True)
    model = models.SoftIntroVAE(arch="conv", cdim=3, zdim=10, channels=(8, 16, 32), image_size=32).to(dev).train()
    solver = IntroTCSovler(DS(), model, 8, torch.optim.Adam(model.encoder.parameters(), lr=2e-4),
                           torch.optim.Adam(model.decoder.parameters(), lr=2e-4), "mse", 1.0, 1.0, 256.0, 1e-8,
                           dev, False, None, clip=100.0)
    solver.enable_graph()
    xs = [torch.rand(8, 3, 32, 32).to(dev) for _ in range(8)]
    for i, x in enumerate(xs):
        mark(f"step {i} begin")
        solver.train_step(x, i)
        mark(f"step {i} end (graph: {getattr(solver, '_graph', None) is not None})")
    time.sleep(0.5)
    mark("shutdown")
    ddp.shutdown()
    dist.destroy_process_group()
    mark("done")


if __name__ == "__main__":
    main()
