"""The N > 1 path with the ENGINE as the photon loop, on a box with ONE GPU: two processes, both on device 0, each runs its own history range
(pmc_history_range) through the HIP engine; the detector arrays are summed onto rank 0 over gloo (RCCL refuses two ranks on one device:
the collective is the stand-in here, everything in front of it is the product).  Rank 0 must then hold the frames of the undivided
single-rank segment -- the counterpart of ProcessManager::sumToRoot behind FluxRecorder::flush (SKIRT/mpi/ProcessManager.cpp:223-255,
SKIRT/core/FluxRecorder.cpp:487-493; the chunk server the static split replaces: SKIRT/core/MultiHybridParallel.cpp:26-104).
And `python bench.py --gpus 2`, launched PLAINLY (no torchrun in front), starts its own ranks and prints the JSON line."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, ski

pytestmark = pytest.mark.gpu

N = 60001
SEED = 11


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from skirt9_amd.engine import Engine, history_range
    from skirt9_amd.host import Simulation
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sim = Simulation(ski("cfg2small.ski"), num_packets=N).setup()  # full replica of the scene on every rank
    eng = Engine(sim.scene, 0)                                     # both ranks on device 0
    first, count = history_range(N, rank, world)
    eng.run_primary(first, count, SEED)
    mine = eng.download()
    histories = eng.counters()["histories"]
    eng.close()
    np.save(os.path.join(outdir, f"before{rank}.npy"), mine)
    t = torch.from_numpy(mine.copy())
    dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(outdir, "sum.npy"), t.numpy())
    h = torch.tensor([histories], dtype=torch.int64)
    dist.all_reduce(h)
    assert int(h.item()) == N
    dist.barrier()
    dist.destroy_process_group()


def test_two_engine_ranks_on_one_device_equal_one(tmp_path):
    import torch.multiprocessing as mp
    from skirt9_amd.engine import Engine, history_range
    from skirt9_amd.host import Simulation
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    both = np.load(tmp_path / "sum.npy")
    parts = [np.load(tmp_path / f"before{r}.npy") for r in range(2)]
    sim = Simulation(ski("cfg2small.ski"), num_packets=N).setup()
    eng = Engine(sim.scene, 0)
    eng.run_primary(0, N, SEED)
    single = eng.download()
    eng.close()
    lay = sim.layout(0)
    # integer counts (histories per wavelength bin that reached the SED of the first instrument): every history exactly once
    assert both[lay.wsed_offset] == single[lay.wsed_offset]
    for r in range(2):
        assert parts[r][lay.wsed_offset] > 0          # both ranks did detect something of their own range
    assert parts[0][lay.wsed_offset] + parts[1][lay.wsed_offset] == single[lay.wsed_offset]
    assert history_range(N, 0, 2)[1] + history_range(N, 1, 2)[1] == N
    assert abs(both.sum() - single.sum()) <= 1e-9 * np.abs(single).sum()
    bad = np.abs(both - single) > 1e-6 * np.abs(single) + 1e-12 * np.abs(single).max()
    assert bad.sum() <= 1e-3 * both.size, int(bad.sum())


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher in front: the ranks are started by bench.py itself; on a one-GPU box they share
    device 0 (BENCH_SHARE_DEVICE=1) and the line says that the exchange ran over gloo (nccl_ranks 0)"""
    env = dict(os.environ, BENCH_SHARE_DEVICE="1", PMC_NUM_SLOTS=str(1 << 20))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--packets", "4e5",
                          "--no-cpu-baseline", "--no-secondary", "--no-breakdown"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["packets_per_step"] == 800000
    assert line["reduce_check"]["rank0_after_reduce"] == line["reduce_check"]["sum_over_ranks_before"]
    assert [r["rank"] for r in line["per_rank"]] == [0, 1] and all(r["packets_per_step"] == 400000 for r in line["per_rank"])
