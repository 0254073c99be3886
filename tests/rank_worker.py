"""One rank of the multi-process sharding rehearsal (tests/test_gpu_parity.py::test_two_ranks_share_one_batch):
decodes shard.partition(costs, world)[rank] of ONE mixed batch on the device it is told to use and writes
{global image index: sha256 of the RGB bytes} to a JSON file.  Rendezvous over gloo on 127.0.0.1."""
import hashlib
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def mixed_batch():
    """64 images of mixed size / sampling / restart interval (BASELINE configs 4 and 5 in miniature).  Deterministic."""
    import hjdtools as T
    specs = []
    sizes = [(640, 480), (320, 240), (1000, 700), (250, 187), (1280, 720), (96, 80), (512, 512), (777, 333)]
    samp = [(2, 2), (2, 1), (1, 1), (1, 2)]
    for g in range(64):
        w, h = sizes[g % 8]
        hf, vf = samp[(g // 8) % 4]
        rst = [0, 0, 7, 0][g % 4] if g % 16 >= 8 else 0
        specs.append((w, h, 4000 + g, 85 - (g % 3) * 10, hf, vf, rst, 7.0))
    return [T.synth_jpeg(*s) for s in specs]


def main():
    rank, world, device, out = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(sys.argv[1]), sys.argv[2]
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hjd = importlib.import_module("hls-jpeg-decoder_amd")
    shard = importlib.import_module("hls-jpeg-decoder_amd.shard")
    jpgs = mixed_batch()
    parts = shard.partition([len(j) for j in jpgs], world)  # cost = compressed bytes
    mine = parts[rank]
    dec = hjd.Decoder(device)
    st, outs = dec.decode([jpgs[g] for g in mine])
    res = {}
    for g, s, o in zip(mine, st, outs):
        res[str(g)] = hashlib.sha256(o.tobytes()).hexdigest() if s == 0 else "status %d" % s
    dec.close()
    # step agreement exactly as bench.py does it: MAX of a time, SUM of the units
    t, u = shard.agree(1.0 + rank, float(len(mine)))
    assert t == float(world) and u == float(len(jpgs)), (t, u)
    dist.barrier()
    json.dump(res, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
