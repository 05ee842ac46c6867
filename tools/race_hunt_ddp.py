"""Developer tool: the two-ranks-on-one-GPU DDP step of tests/test_gpu_ddp.py, repeated in fresh process pairs in deterministic mode with a
shared tuning table: rank 0's reduced gradient must be bit-identical to the run with every launch on one stream (ZSG_SIDE_STREAM=0).
Prints which parameters differ.  usage (GPU box): python tools/race_hunt_ddp.py [trials]"""
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    import torch
    import torch.distributed as dist
    from oracle import zsg_oracle as O
    from zsgnet_pytorch_amd import config, dist as zdist, loss, mdl, optim
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    cfg = config.get_cfg(resnet_arch="resnet18")
    net = mdl.get_default_net(9, cfg)
    net.load_state_dict(O.seeded_state_dict("resnet18", 40 + rank))
    net.to("cuda").train()
    ddp = zdist.DistributedDataParallel(net, device_ids=[0], broadcast_buffers=True, bucket_mb=4.0)
    r, s = config.ratios_scales(cfg)
    lf = loss.get_default_loss(r, s, cfg)
    opt = optim.FusedAdam(net, lr=1e-3)
    bt = {k: v.cuda() for k, v in O.synthetic_batch(2, 96, 96, seed=70 + rank).items()}
    bt["h0"], bt["c0"] = torch.zeros(2, 2, 128), torch.zeros(2, 2, 128)
    opt.zero_grad()
    lf(ddp(bt), bt)["loss"].backward()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"g": net.store.grad.clone().cpu(), "names": list(net._param_names),
                    "ents": {n: (net.store.entries[n].offset, net.store.entries[n].size) for n in net._param_names}}, out)
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 4 and sys.argv[1] == "--worker":
        return worker(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    import torch
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    tmp = tempfile.mkdtemp()
    env = dict(os.environ, ZSG_DETERMINISTIC="1", ZSG_TUNE_CACHE=os.path.join(tmp, "tune.json"))

    def run(tag, extra):
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        out = os.path.join(tmp, tag + ".pt")
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(port), out], env=dict(env, **extra),
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for r in range(2)]
        for p in ps:
            p.wait(timeout=300)
        return torch.load(out)
    run("tune", {"ZSG_SIDE_STREAM": "0"})
    ref = run("ref", {"ZSG_SIDE_STREAM": "0"})
    ref2 = run("ref2", {"ZSG_SIDE_STREAM": "0"})
    print("serial run reproducible:", bool(torch.equal(ref["g"], ref2["g"])))
    bad = 0
    for t in range(trials):
        d = run(f"t{t}", {})
        if not torch.equal(d["g"], ref["g"]):
            bad += 1
            worst = []
            for n in d["names"]:
                o, sz = d["ents"][n]
                a, b = d["g"][o:o + sz], ref["g"][o:o + sz]
                e = float((a - b).abs().max())
                if e > 0:
                    worst.append((e / (float(b.abs().max()) + 1e-30), n))
            worst.sort(reverse=True)
            rel = float((d["g"].double() - ref["g"].double()).norm() / ref["g"].double().norm())
            print(f"trial {t}: rel {rel:.3g}; {len(worst)} parameters differ; worst: " + ", ".join(f"{n} {e:.2e}" for e, n in worst[:8]))
    print(f"{bad} of {trials} trials differ from the single-stream step")


main()
