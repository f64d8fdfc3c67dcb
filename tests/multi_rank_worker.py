"""One rank of a tile-sharded G-BRE run on the device (a script, not a test: tests/test_multi_rank_gpu.py starts it
under torch.distributed.run).  Every rank gathers the 4x4-pixel tiles t with t % world == rank of a small frame for
a few SPPM iterations (photon map replicated), turns its accumulators into its partial film (gvpm_download_film_dev)
and the films are summed across ranks -- through torch.distributed (`--collective torch`), or through the library's own
RCCL path gvpm_comm_init / gvpm_allreduce_film (`--collective gvpm`).  Rank 0 saves the summed film."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--scene", default="fogroom")
ap.add_argument("--frame", type=int, default=96)
ap.add_argument("--photons", type=int, default=60000)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--scale", type=float, default=3.0)
ap.add_argument("--backend", default="gloo")
ap.add_argument("--collective", default="torch", choices=["torch", "gvpm"])
ap.add_argument("--single-device", action="store_true")
args = ap.parse_args()

rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from gvpm_amd import hip  # noqa: E402
from gvpm_amd.host import SynthScene  # noqa: E402

torch.cuda.set_device(local_rank)
if world > 1:
    dist.init_process_group(args.backend)
sc = SynthScene(args.scene, args.frame, args.frame)
p = sc.params()
p.initial_scale_volume = args.scale
ctx = hip.Context(p, device=local_rank)
ctx.upload_scene(*sc.triangles())
ctx.upload_medium(sc.medium())
for it in range(1, args.steps + 1):
    ph, nb = sc.shoot_photons(it, args.photons)
    rays = sc.camera_beams_interleaved(it, world, rank) if world > 1 else sc.camera_beams(it)
    ctx.upload_photons(ph)
    ctx.upload_camera_beams(rays)
    ctx.gather(it, nb)
film = torch.zeros(args.frame * args.frame * 9, dtype=torch.float32, device="cuda")
ctx.download_film_dev(args.steps, film.data_ptr())
ctx.synchronize()
if args.collective == "gvpm":
    # the C ABI's own collective: ncclUniqueId made on rank 0, handed round by the host (here: torch.distributed)
    ident = [hip.comm_unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(ident, src=0)
    ctx.comm_init(ident[0], rank, world)
    ctx.allreduce_film(film.data_ptr())
    ctx.synchronize()
elif world > 1:
    dist.all_reduce(film)
st = ctx.stats()
if rank == 0:
    np.save(args.out, film.cpu().numpy().reshape(3, args.frame, args.frame, 3))
    print("evaluations", st["evaluations"], flush=True)
ctx.close()
if world > 1:
    dist.destroy_process_group()
