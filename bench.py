#!/usr/bin/env python3
"""bench.py -- photon packets/s of the MI355X primary-emission engine on BASELINE.json's configs[1]:
Sersic source, ~10^6-cell octree dust grid (953 688 cells, tests/ski/cfg2.ski), monochromatic, peel-off to one
FullInstrument 512^2 with component + statistics recording, 10^8 packets per step and GPU.

  python bench.py --gpus N --steps K --warmup W [--packets P]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one segment: every rank runs P histories of its own index range through the HIP engine and the
detector frames of all ranks are summed onto rank 0 with ONE RCCL reduce -- the engine library's own
pmc_reduce_frames over the communicator it built with pmc_comm_init_rank (the counterpart of
ProcessManager::sumToRoot at FluxRecorder.cpp:487-493); torch.distributed is the control plane only (hands the
communicator id to the ranks, barrier, maximum of the ranks' times).  Inputs (grid, densities, tables) are resident in HBM
before the timed region.  Prints one JSON line (rank 0).

roofline: HBM-bound walk.  achieved = algorithmic bytes of one step (V*20 + U*8 with V = cell visits and U = detector
updates, both COUNTED by the kernels) / GPU time of the step's segment, measured with HIP events on the
engine's stream.  The walk kernel launches of the two slot groups overlap with each other and with the transition /
launch kernels on separate streams, so the segment time (not the sum of the per-launch durations, which is reported
as walk_kernel_ms_sum) is the time the dominant kernel has to move those bytes.
cpu_baseline: the unmodified reference (oracle/_ref, built by oracle/Makefile.ref) run on the host cores of this box
on a bounded number of packets of the same ski file; falls back to the scalar CPU oracle if the binary is absent.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SKI = os.path.join(ROOT, "tests", "ski", "cfg2.ski")
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
VORONOI_KEPT_NEIGHBOURS = 0.50  # fraction of a Voronoi cell's neighbours that can be the exit for ONE direction (n . k > 0): what a walk
                                # has to read at least.  The peel-off walks read exactly those (observer tables, round 5), the propagation
                                # walks the 60 % their cone's mask keeps: the smaller figure prices every visit
# random dependent gathers per second of the whole chip with every lane on a trajectory of its own (profiles/microbench/
# true_gather_mi355x.txt; the 2.1e11 of rounds 2-3 came from lanes that merged, profiles/r04_bridge.md): a 32 MB table without any
# locality, and a table that fits the 4 MB L2 of an XCD -- the walk kernels' lane-step rates lie between the two by their locality
GATHER_NO_LOCALITY = 0.665e11
GATHER_L2_RESIDENT = 2.5e11


def pmc_traffic(packets_per_step):
    """(bytes of the walk kernels, bytes of all kernels, source file) -- memory-side bytes of one step, from the newest committed PMC summary
    profiles/r*_pmc_hbm.csv: FETCH_SIZE + WRITE_SIZE (KiB, separate rocprofv3 --pmc passes over one step of 2e7
    packets of this workload; tools/run_profile_set.sh, tools/pmc_hbm_summary.py), scaled by the packet count -- the FALLBACK when the
    counter passes of the run itself (pmc_in_run) are not available.  What the counters count (calibrated in round 6 on known access
    patterns, profiles/microbench/fetch_calibration_mi355x.txt): a 16-byte gather that misses L2 is ONE 64-byte request to the fabric and
    FETCH_SIZE counts 64 bytes for it; a wide coalesced read goes out as 128-byte requests that FETCH_SIZE counts at 64 (half the bytes:
    MI355X_MICROARCH.md); a coalesced write is counted in full, a scattered 8-byte store as one 32-byte request.  The walk kernels' reads are
    gathers, so their figure stands as counted; Infinity-Cache hits are included."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.csv")),
                   key=lambda f: [int(x) for x in re.findall(r"[0-9]+", os.path.basename(f))])
    if not files:
        return None, None, None
    kib = allkib = 0.0
    for row in csv.DictReader(open(files[-1])):
        allkib += float(row["sum_KiB_per_step_of_2e7_packets"])
        if row["kernel"] in ("walkKernel", "walkPeelKernel", "walkPropKernel", "voroPeelKernel", "voroPropKernel"):
            kib += float(row["sum_KiB_per_step_of_2e7_packets"])
    scale = 1024.0 / 2e7 * packets_per_step
    return (kib * scale if kib else None), (allkib * scale if allkib else None), os.path.relpath(files[-1], ROOT)


def pmc_l2(kernel):
    """L2 requests, hits, misses and VALU instructions of `kernel` per step of 2e7 packets of the headline workload, from the newest
    committed counter pass profiles/r*_pmc_l2.csv (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VALU SQ_WAVES, a run of its
    own: tools/run_profile_set.sh, tools/pmc_l2_summary.py); (dict, source file) or (None, None)"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_l2.csv")),
                   key=lambda f: [int(x) for x in re.findall(r"[0-9]+", os.path.basename(f))])
    if not files:
        return None, None
    for row in csv.DictReader(open(files[-1])):
        if kernel.startswith(row["kernel"]):
            return {k.replace("_per_step_of_2e7_packets", ""): float(v) for k, v in row.items() if k.endswith("_per_step_of_2e7_packets")}, \
                os.path.relpath(files[-1], ROOT)
    return None, None


KERNEL_RE = (r"(voroPropKernel|voroPeelKernel|walkPeelKernel|walkPropKernel|walkKernel|transitionKernel|launchKernel|cycleStartKernel|endedScanKernel|"
             r"statMergeKernel|statReduceKernel|peelSortCountKernel|peelSortOffsetsKernel|rfHistKernel|rfScanKernel|rfScatterKernel|rfReduceKernel)")
WALK_KERNELS = ("walkKernel", "walkPeelKernel", "walkPropKernel", "voroPeelKernel", "voroPropKernel")
PMC_PASS_PACKETS = 20000000
# what one unit of FETCH_SIZE / WRITE_SIZE stands for, calibrated on this box class with a KNOWN number of lines from beyond L2
# (tools/fetch_calibration.sh, profiles/microbench/fetch_calibration_mi355x.txt; profiles/README.md "FETCH_SIZE calibration")
FETCH_UNIT_BYTES = 1024.0


def pmc_in_run(scene_file, passthrough):
    """Hardware counters of THIS build on THIS box, behind the timed region: one step of PMC_PASS_PACKETS packets of the same workload is run
    again in a child process under `rocprofv3 --pmc` -- three passes (FETCH_SIZE; WRITE_SIZE; the L2 set), each a run of its own as
    MI355X_MICROARCH.md prescribes (no trace domain next to --pmc) -- on the scene the timed run has saved (no second set-up).  Returns
    {kernel: {counter: sum over its launches}} or None when rocprofv3 is not on PATH or a pass fails (the line then falls back to the
    committed summaries and says so)."""
    import csv
    import glob
    import shutil
    prof = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not prof:
        return None
    tot = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    for counters in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum SQ_INSTS_VALU SQ_WAVES"):
        out = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        cmd = [prof, "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__),
               "--steps", "1", "--warmup", "0", "--packets", str(PMC_PASS_PACKETS), "--no-cpu-baseline", "--no-secondary", "--no-breakdown",
               "--no-counters", "--scene-file", scene_file] + passthrough
        try:
            subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, cwd="/tmp", env=env)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if not files:
                raise RuntimeError("no counter_collection.csv")
            for f in files:
                for row in csv.DictReader(open(f)):
                    m = re.search(KERNEL_RE, row["Kernel_Name"])
                    if m:
                        k = tot.setdefault(m.group(1), {})
                        k[row["Counter_Name"]] = k.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
        except Exception as exc:  # noqa: BLE001 - evidence, not the measurement
            sys.stderr.write(f"[bench] counter pass '{counters}' failed: {exc}\n")
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    return tot


def cpu_baseline(ski_path=SKI, input_dir=None, packets_per_core=100000):
    """photon packets/s of the CPU path on this box's host cores, on a bounded sample (about 10-30 s)"""
    cores = min(os.cpu_count() or 1, 24)  # the reference caps a process at 24 threads (ParallelFactory.cpp:43-50)
    ref = os.path.join(ROOT, "oracle", "_ref", "release", "SKIRT", "main", "skirt_ref")
    if os.path.exists(ref):
        n = packets_per_core * cores
        with tempfile.TemporaryDirectory() as tmp:
            text = open(ski_path).read().replace('numPackets="1e5"', f'numPackets="{n}"')
            ski = os.path.join(tmp, "cfg2cpu.ski")
            open(ski, "w").write(text)
            if input_dir:  # input files named in the ski file are read from the working directory
                for f in os.listdir(input_dir):
                    os.symlink(os.path.join(input_dir, f), os.path.join(tmp, f))
            try:
                subprocess.run([ref, "run", ski, "-t", str(cores), "-o", tmp], check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, timeout=900, cwd=tmp)
                log = open(os.path.join(tmp, "cfg2cpu_log.txt")).read()
                m = re.search(r"Finished primary emission in ([0-9.]+) s", log)
                if m and float(m.group(1)) > 0:
                    return {"value": n / float(m.group(1)), "unit": "photon packets/s", "cores": cores, "kind": "reference",
                            "sample": f"unmodified SKIRT 9 (oracle/_ref), {os.path.basename(ski_path)} scene, {n} packets, -t {cores}; "
                                      f"'Finished primary emission' {m.group(1)} s"}
            except Exception as exc:  # noqa: BLE001 - the baseline is informational
                sys.stderr.write(f"[bench] reference baseline failed: {exc}\n")
    # scalar port on one core
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    from skirt9_amd.host import Simulation
    sim = Simulation(ski_path, num_packets=20000).setup()
    t0 = time.time()
    O.run_primary(sim, 0, 20000, O.RNG_PHILOX, seed=1)
    dt = time.time() - t0
    return {"value": 20000 / dt, "unit": "photon packets/s", "cores": 1, "kind": "port",
            "sample": f"oracle/life_cycle.cpp, {os.path.basename(ski_path)} scene, 20000 packets, 1 thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--packets", type=float, default=1e8, help="photon packets per step and GPU")
    ap.add_argument("--ski", default=SKI)
    ap.add_argument("--source", choices=["sersic", "uniform"], default="sersic",
                    help="sersic: cfg2.ski as is (the headline workload); uniform: the same scene with the Sersic source "
                         "replaced by a UniformBoxGeometry source of +-10 x +-10 x +-1 kpc (north_star's second source)")
    ap.add_argument("--config", type=int, choices=[2, 3, 4, 5], default=2,
                    help="2: BASELINE configs[1] (the headline workload, tests/ski/cfg2.ski); 3: BASELINE configs[2], the same scene "
                         "panchromatic with a 50-bin wavelength grid (tests/ski/cfg3.ski); 4: BASELINE configs[3], the same "
                         "Sersic source in dust imported from 10^6 smoothed particles (tests/ski/cfg4.ski; the particle file "
                         "is regenerated by tools/make_sph.py); 5: BASELINE configs[4], the config-3 scene on a Voronoi grid of 10^5 sites "
                         "(tests/ski/cfg5.ski; sites regenerated by tools/make_sites.py), three instruments")
    ap.add_argument("--store-radiation-field", action="store_true",
                    help="run the same workload with RadiationFieldOptions storeRadiationField=true (the RF flavour of the "
                         "walk kernel: one exp, one lnmean and one f64 atomic more per path segment); not the headline number")
    ap.add_argument("--explicit-absorption", action="store_true",
                    help="run the same workload with PhotonPacketOptions explicitAbsorption=true (the EA flavour of the propagation kernels); not the headline number")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --packets is the TOTAL per step, split over the ranks by history range (e.g. "
                         "--config 4 --packets 1e9 --gpus 8 = BASELINE configs[3]); default: weak scaling, --packets per GPU")
    ap.add_argument("--sites", type=float, default=1e5, help="--config 5: number of Voronoi sites (BASELINE configs[4]: 1e5)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second measurement of the default run (the same octree with a uniform-box source)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--secondary", action="store_true", help="N > 1: also measure the uniform-box source (default at N = 1)")
    ap.add_argument("--no-breakdown", action="store_true", help="skip the per-kernel pass (one slot group, kernels in series) behind the timed region")
    ap.add_argument("--no-counters", action="store_true",
                    help="skip the hardware-counter passes behind the timed region (one step of 2e7 packets re-run under rocprofv3 --pmc: FETCH_SIZE, "
                         "WRITE_SIZE, L2 requests / hits / misses; N = 1 only)")
    ap.add_argument("--scene-file", default=None, help="(internal: the counter passes) load the scene from this file (skh_scene_save) instead of setting it up")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from skirt9_amd.engine import Engine, tuning_from_environment
    from skirt9_amd.host import SceneFile, Simulation

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.gpus != world:
        if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
            # launched plainly (`python bench.py --gpus N`): start the N ranks here -- one process per GPU under torch.distributed.run, the
            # form the driver uses -- and hand their output through; the JSON line of rank 0 stays the last line on stdout
            if torch.cuda.device_count() < args.gpus and os.environ.get("BENCH_SHARE_DEVICE") != "1":
                raise SystemExit(f"--gpus {args.gpus}: this box has {torch.cuda.device_count()} device(s) (BENCH_SHARE_DEVICE=1 runs all ranks "
                                 "on device 0 with the exchange over gloo: a flow check, not a measurement)")
            import socket
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            raise SystemExit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
                                              "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:], env=env))
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}")
    # testing aid for a 1-GPU box: BENCH_SHARE_DEVICE=1 runs all ranks on device 0 and exchanges over gloo (RCCL refuses
    # two ranks on one device); the history split, the per-step reduce and the timing protocol are the same
    share = world > 1 and os.environ.get("BENCH_SHARE_DEVICE") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # the data path of N > 1: the PRODUCT's RCCL communicator (include/pmc.h pmc_comm_*), one rank per process.  Rank 0
    # draws the id, the control plane hands it to the others.
    comm = None
    nccl_ranks = 1
    # (BENCH_FORCE_COMM=1: also with one rank -- a one-rank communicator walks the same calls on a one-GPU box)
    if (world > 1 and not share) or (world == 1 and os.environ.get("BENCH_FORCE_COMM") == "1"):
        from skirt9_amd.engine import Communicator
        ident = torch.zeros(128, dtype=torch.uint8, device=f"cuda:{local_rank}")
        if rank == 0:
            ident.copy_(torch.frombuffer(bytearray(Communicator.unique_id()), dtype=torch.uint8))
        if world > 1:
            dist.broadcast(ident, src=0)
        comm = Communicator.rank(local_rank, world, rank, bytes(ident.cpu().numpy().tobytes()))
        nccl_ranks, my_rank = comm.size()
        if nccl_ranks != world or my_rank != rank:
            raise SystemExit(f"RCCL communicator reports {nccl_ranks} ranks / rank {my_rank}, expected {world} / {rank}")

    P = int(args.packets)
    total_per_step = P if args.strong else P * world
    # (N > 1: rank 0 alone sets the scene up, with all host cores, and the others load the file it saves: measure() below)
    if args.config == 3:
        if args.ski != SKI or args.source != "sersic":
            raise SystemExit("--config 3 selects its own ski file and source")
        args.ski = os.path.join(ROOT, "tests", "ski", "cfg3.ski")
    if args.config == 5:
        if args.ski != SKI or args.source != "sersic":
            raise SystemExit("--config 5 selects its own ski file and source")
        args.ski = os.path.join(ROOT, "tests", "ski", "cfg5.ski")
        sitedir = tempfile.mkdtemp(prefix=f"bench_sites_r{rank}_")
        if (rank == 0 or world == 1) and not args.scene_file:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_sites.py"), "--n", str(int(args.sites)), "--seed", "1",
                                   os.path.join(sitedir, "cfg5_sites.txt")])
        os.environ["SKH_INPUT_PATH"] = sitedir
    if args.config == 4:
        if args.ski != SKI or args.source != "sersic":
            raise SystemExit("--config 4 selects its own ski file and source")
        args.ski = os.path.join(ROOT, "tests", "ski", "cfg4.ski")
        sphdir = tempfile.mkdtemp(prefix=f"bench_sph_r{rank}_")
        if (rank == 0 or world == 1) and not args.scene_file:
            subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_sph.py"), "--n", "1000000", "--seed", "1",
                                   os.path.join(sphdir, "cfg4_sph.txt")])
        os.environ["SKH_INPUT_PATH"] = sphdir
    # every rank sets up the same scene (replica of grid, densities, tables); numPackets = packets of one step over
    # all ranks, so that the per-packet luminosity is that of the whole segment
    ski_path = args.ski
    if args.source == "uniform":
        text = open(args.ski).read()
        new = ('<UniformBoxGeometry minX="-10000 pc" maxX="10000 pc" minY="-10000 pc" maxY="10000 pc" '
               'minZ="-1000 pc" maxZ="1000 pc"/>')
        text, nsub = re.subn(r"<SersicGeometry[^>]*/>", new, text, count=1)
        if nsub != 1:
            raise SystemExit("--source uniform: no SersicGeometry source in the ski file")
        ski_path = os.path.join(tempfile.mkdtemp(prefix=f"bench_r{rank}_"), "cfg2u.ski")
        open(ski_path, "w").write(text)
    if args.store_radiation_field:
        text = open(ski_path).read()
        if 'storeRadiationField="false"' not in text:
            raise SystemExit("--store-radiation-field: the ski file has no storeRadiationField=\"false\" to switch")
        rf_ski = os.path.join(tempfile.mkdtemp(prefix=f"bench_rf_r{rank}_"), os.path.basename(ski_path))
        if '<RadiationFieldOptions storeRadiationField="false"/>' in text:
            # (a ski file without a radiation field wavelength grid: the instruments' range in twenty bins)
            text = text.replace('<RadiationFieldOptions storeRadiationField="false"/>',
                                '<RadiationFieldOptions storeRadiationField="true"><radiationFieldWLG type="DisjointWavelengthGrid"><LogWavelengthGrid '
                                'minWavelength="0.1 micron" maxWavelength="10 micron" numWavelengths="20"/></radiationFieldWLG></RadiationFieldOptions>')
        open(rf_ski, "w").write(text.replace('storeRadiationField="false"', 'storeRadiationField="true"'))
        ski_path = rf_ski
    if args.explicit_absorption:
        text = open(ski_path).read()
        if 'explicitAbsorption="false"' not in text:
            raise SystemExit("--explicit-absorption: the ski file has no explicitAbsorption=\"false\" to switch")
        ea_ski = os.path.join(tempfile.mkdtemp(prefix=f"bench_ea_r{rank}_"), os.path.basename(ski_path))
        open(ea_ski, "w").write(text.replace('explicitAbsorption="false"', 'explicitAbsorption="true"'))
        ski_path = ea_ski
    from skirt9_amd.engine import history_range
    from skirt9_amd.host import scene_head

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(path, steps, warmup, save_scene_to=None):
        """sets the scene of ski file `path` up on this rank's GPU and times `steps` segments; returns the numbers of the
        JSON line (rank 0) -- value, roofline inputs, counters"""
        # Every rank needs a replica of the scene.  ONE rank sets it up (all host cores) and saves it as one file in /dev/shm
        # (skh_scene_save); the others load that file -- no second tree construction, no second density sampling.
        sim = None
        cache = None
        t_setup = time.perf_counter()
        if args.scene_file:
            sim = SceneFile(args.scene_file)
        elif rank == 0:
            sim = Simulation(path, num_packets=total_per_step)
            if args.config == 4:
                sim.use_device_sampler(local_rank)  # setup-time density sampling of the particle medium on this rank's GPU
            sim.setup()
            if save_scene_to:
                sim.save_scene(save_scene_to)
            if world > 1:
                # (a private file under an unpredictable name, readable by this user only; the loader checks sizes, offsets and a
                # checksum before it trusts a byte of it)
                fd, cache = tempfile.mkstemp(prefix="pmc_scene_", suffix=".bin", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
                os.close(fd)
                os.chmod(cache, 0o600)
                sim.save_scene(cache)
        if world > 1:
            names = [cache]
            dist.broadcast_object_list(names, src=0)
            cache = names[0]
            if rank != 0:
                sim = SceneFile(cache)
            dist.barrier()
            if rank == 0:
                os.unlink(cache)
        tuning_from_environment()  # (PMC_* variables other than the library's three settings: tuning switches for measurement runs)
        eng = Engine(sim.scene, local_rank)
        setup_s = time.perf_counter() - t_setup   # (rank 0: the whole setup; the others: waiting for it + loading the scene file)
        frames = torch.zeros(sim.frame_size, dtype=torch.float64, device=f"cuda:{local_rank}")
        eng.bind_frames(frames.data_ptr(), frames.numel())
        reduce_ms = []
        rf = None
        if sim.radiation_field_size:
            # the radiation field table in a torch tensor, so that the ranks can sum it onto ALL ranks over RCCL
            # (MediumSystem::communicateRadiationField, MediumSystem.cpp:1304-1313)
            rf = torch.zeros(sim.radiation_field_size, dtype=torch.float64, device=f"cuda:{local_rank}")
            eng.bind_radiation_field(rf.data_ptr(), rf.numel())
        seed = sim.seed

        def step(index):
            # static split of the segment's history range over the ranks (pmc_history_range); a fresh range
            # of history indices per step.  The engine runs on its own streams: torch's work on the bound tensors (zero_,
            # the previous reduce) must be complete before the segment starts.
            torch.cuda.synchronize()
            first, count = history_range(total_per_step, rank, world)
            eng.run_primary(index * total_per_step + first, count, seed)
            eng.sync()
            if comm is not None:
                # ONE ncclReduce of the bound frames onto rank 0 on the engine's stream; the other ranks' frames are cleared
                # (pmc_reduce_frames returns when the reduce is complete)
                t_red = time.perf_counter()
                eng.reduce_frames(comm.handles[0], 0)
                if rf is not None:
                    eng.allreduce_radiation_field(comm.handles[0])
                    rf.zero_()      # (a benchmark step is a whole segment: what follows would consume the field here)
                reduce_ms.append(1e3 * (time.perf_counter() - t_red))   # (includes waiting for the slowest rank's segment)
            elif world > 1:
                # (BENCH_SHARE_DEVICE: all ranks on one device, exchange over gloo -- a flow check, RCCL refuses it)
                dist.all_reduce(frames, op=dist.ReduceOp.SUM)
                if rank != 0:
                    frames.zero_()
                if rf is not None:
                    dist.all_reduce(rf, op=dist.ReduceOp.SUM)
                    rf.zero_()

        for w in range(warmup):
            step(w)
        frames.zero_()
        eng.reset_counters()
        fence()
        t0 = time.perf_counter()
        timings = []
        for k in range(steps):
            step(warmup + k)
            t = eng.last_timing()
            t["walk_ms"] = eng.last_kernel_ms()  # HIP events around every walk-kernel launch of the segment, summed
            t.update(eng.last_walk_timing())     # octree: the peel-off and the propagation kernels apart
            timings.append(t)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        counters = eng.counters()
        walk_work = eng.walk_work()
        # outside the timed region: one more (small) segment whose reduce is checked -- what rank 0 holds afterwards is the
        # sum of what the ranks held before it.  wsed[0] (histories per wavelength bin that reached the first instrument's
        # SED) is a table of integer counts, so the comparison is exact.
        reduce_check = None
        if world > 1 or comm is not None:
            frames.zero_()
            torch.cuda.synchronize()
            n_check = 200000 * world
            first, count = history_range(n_check, rank, world)
            eng.run_primary((warmup + steps) * total_per_step + first, count, seed)
            eng.sync()
            lay = sim.layout(0)
            part = frames[lay.wsed_offset:lay.wsed_offset + lay.num_lambda] if lay.wsed_offset >= 0 else frames
            before = part.sum().reshape(1).clone()
            if world > 1:
                dist.all_reduce(before, op=dist.ReduceOp.SUM)
            if comm is not None:
                eng.reduce_frames(comm.handles[0], 0)
            else:
                dist.all_reduce(frames, op=dist.ReduceOp.SUM)
            after = float(part.sum().item())
            want = float(before.item())
            exact = lay.wsed_offset >= 0
            failed = torch.tensor([1.0 if (rank == 0 and (after != want if exact else abs(after - want) > 1e-9 * abs(want))) else 0.0],
                                  dtype=torch.float64, device=f"cuda:{local_rank}")
            if world > 1:
                dist.all_reduce(failed, op=dist.ReduceOp.MAX)   # (every rank leaves together: none is left waiting in a collective)
            if float(failed.item()) != 0.0:
                raise SystemExit(f"reduce check failed: rank 0 holds {after!r} after the reduce, the ranks held {want!r} before it")
            reduce_check = {"quantity": "sum of wsed[0] of the first instrument (integer counts)" if exact else "sum of the frames",
                            "histories": n_check, "rank0_after_reduce": after, "sum_over_ranks_before": want}
        grid = scene_head(sim).grid
        # behind the timed region: the kernels of this build one after the other -- ONE slot group, peel-off and propagation kernel in
        # series, HIP events per kernel kind -- on a fifth of a step, so that the line can name its dominant kernel with that kernel's
        # own time (the overlapped segment cannot: its kernels share the device)
        breakdown = None
        if rank == 0 and grid.kind == 2 and not args.no_breakdown:
            eng.close()
            # (one slot group: a library setting read from the environment at pmc_create; kernels in series: a tuning switch)
            from skirt9_amd import engine as _engine_module
            keep = {k: os.environ.get(k) for k in ("PMC_NUM_GROUPS",)}
            os.environ["PMC_NUM_GROUPS"] = "1"
            _engine_module.set_tuning("PMC_SERIAL_WALKS", "1")
            try:
                one = Engine(sim.scene, local_rank)
                one.bind_frames(frames.data_ptr(), frames.numel())
                nb = max(1000000, history_range(total_per_step, rank, world)[1] // 5)
                one.run_primary((warmup + steps + 2) * total_per_step, nb // 10, seed)
                one.sync()
                one.reset_counters()
                one.run_primary((warmup + steps + 3) * total_per_step, nb, seed)
                one.sync()
                t, k, w, c = one.last_timing(), one.last_walk_timing(), one.walk_work(), one.counters()
                breakdown = {"packets": nb, "segment_ms": t["total_ms"], "peel_ms": k["peel_ms"], "prop_ms": k["prop_ms"],
                             "transition_side_ms": t["transition_ms"], "peel_lane_steps": w["peel_lane_steps"], "prop_lane_steps": w["prop_lane_steps"],
                             "peel_lanes_in_use": w["peel_lane_steps"] / (64.0 * max(1, w["peel_wave_steps"])),
                             "prop_lanes_in_use": w["prop_lane_steps"] / (64.0 * max(1, w["prop_wave_steps"])),
                             "rewalk_visits": c["rewalk_visits"], "cell_visits": c["cell_visits"]}
                one.close()
            finally:
                _engine_module.set_tuning("PMC_SERIAL_WALKS", os.environ.get("PMC_SERIAL_WALKS"))
                for k, v in keep.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            frames.zero_()
        result = {"elapsed": elapsed, "timings": timings, "counters": counters, "cells": int(grid.num_cells), "steps": steps,
                  "breakdown": breakdown, "setup_s": setup_s, "reduce_ms": (sum(reduce_ms[warmup:]) / max(1, len(reduce_ms[warmup:]))) if reduce_ms else None,
                  "reduce_check": reduce_check, "walk_work": walk_work,
                  "packets_this_rank": history_range(total_per_step, rank, world)[1],
                  "vnbr_mean": (grid.vnbr_start[grid.num_cells] / grid.num_cells) if args.config == 5 else None}
        eng.close()
        del frames, rf
        return result

    def mean_ms_of(m):
        return sum(t["total_ms"] for t in m["timings"]) / len(m["timings"])

    def roofline_of(m):
        """algorithmic bytes of one step of this rank (V * bytes per visit + U * 8, V and U COUNTED by the kernels) over the
        GPU time of the step's segment (HIP events on the engine's stream)"""
        launches = max(1, m["steps"])
        V = m["counters"]["cell_visits"] / launches
        U = m["counters"]["detector_updates"] / launches
        bytes_per_visit = 20.0
        if args.config == 5:
            # Voronoi: a visit reads the cell's own record (site + density, 32 B) and, for each of its neighbours, the
            # neighbour index (4 B) and the neighbour's site (24 B) -- VoronoiMeshSnapshot.cpp:1096-1150; the mean
            # neighbour count is taken over the cells of the mesh (15.2 for tests/ski/cfg5.ski)
            # the walk skips the neighbours that cannot be the exit (n . k <= 0: half of them; the observer tables hold exactly the other
            # half, the cone masks of the propagation walks keep 60 %): only the bytes of the neighbours every
            # walk has to read count, so that `frac` cannot exceed what is read
            bytes_per_visit = 32.0 + 28.0 * m["vnbr_mean"] * VORONOI_KEPT_NEIGHBOURS
        n = max(1, m["packets_this_rank"])
        bytes_per_launch = bytes_per_visit * V + 8.0 * U
        mean_ms = sum(t["total_ms"] for t in m["timings"]) / len(m["timings"])
        achieved = bytes_per_launch / (mean_ms * 1e-3) / 1e9
        extra = {}
        w = m["walk_work"]
        if w["prop_wave_steps"] and w["peel_wave_steps"]:
            # octree: secondary yardsticks.  A lane-step is one cell visit by one lane (first pass, second pass and peel-off walks
            # alike): one dependent gather of a line of the hot cell table.  The chip's rate of such gathers depends on their locality
            # alone (profiles/r04_bridge.md): GATHER_NO_LOCALITY for a table of this size without any, GATHER_L2_RESIDENT when every
            # request hits L2.  The per-kernel figures divide each kernel's own work by its own span (the spans of the two kernels
            # overlap on two streams, and with the other slot groups' kernels).
            peel_ms = sum(t["peel_ms"] for t in m["timings"]) / len(m["timings"])
            prop_ms = sum(t["prop_ms"] for t in m["timings"]) / len(m["timings"])
            peel_ls, prop_ls = w["peel_lane_steps"] / launches, w["prop_lane_steps"] / launches
            extra = {"lane_steps_per_s": (peel_ls + prop_ls) / (mean_ms_of(m) * 1e-3),
                     "gather_rate_no_locality": GATHER_NO_LOCALITY, "gather_rate_l2_resident": GATHER_L2_RESIDENT,
                     "gather_rates_source": "profiles/microbench/true_gather_mi355x.txt (private trajectories; profiles/r04_bridge.md)",
                     "lane_steps_per_packet": (peel_ls + prop_ls) / n,
                     "prop": {"span_ms": prop_ms, "lane_steps_per_s": prop_ls / (prop_ms * 1e-3), "frac": 20.0 * prop_ls / (prop_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "lanes_in_use": w["prop_lane_steps"] / (64.0 * w["prop_wave_steps"])},
                     "peel": {"span_ms": peel_ms, "lane_steps_per_s": peel_ls / (peel_ms * 1e-3), "frac": 20.0 * peel_ls / (peel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "lanes_in_use": w["peel_lane_steps"] / (64.0 * w["peel_wave_steps"])}}
        b = m.get("breakdown")
        if b:
            # the dominant kernel by its OWN time: the kernels of one slot group in series (measured behind the timed region on
            # b["packets"] packets), scaled to one step of this rank; algorithmic bytes = 20 B x the cell visits of the reference's
            # paths that kernel walks (the propagation kernel's second-pass re-walk is work of this engine, not of the algorithm)
            scale = n / b["packets"]
            prop_visits = b["prop_lane_steps"] - b["rewalk_visits"]
            kernels = {"walkPropKernel": (b["prop_ms"], 20.0 * prop_visits), "walkPeelKernel2": (b["peel_ms"], 20.0 * b["peel_lane_steps"])}
            name = max(kernels, key=lambda k: kernels[k][0])
            ms, nbytes = kernels[name]
            extra["dominant_kernel"] = {"name": name, "serial_ms_per_step": ms * scale, "algorithmic_bytes_per_step": nbytes * scale,
                                        "achieved": nbytes / (ms * 1e-3) / 1e9, "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                        "lane_steps_per_s": (b["prop_lane_steps"] if name == "walkPropKernel" else b["peel_lane_steps"]) / (ms * 1e-3),
                                        "measured": f"one slot group, kernels in series, {b['packets']} packets behind the timed region of this run"}
            # line-rate evidence for that kernel: its L2 requests / hits / misses from the committed counter pass (2e7 packets of the
            # headline workload), per lane-step of this run's count, and its miss rate against the chip's rate of random lines from
            # beyond L2 (GATHER_NO_LOCALITY: profiles/microbench/true_gather_mi355x.txt) with the kernel's own serial time
            if counters_in_run or (args.config == 2 and args.source == "sersic" and not args.store_radiation_field and not args.explicit_absorption):
                l2, l2_source = None, None
                in_run = False
                for kname, v in (counters_in_run or {}).items():
                    if name.startswith(kname) and v.get("TCC_REQ_sum"):
                        l2, l2_source, in_run = v, f"this run (rocprofv3 --pmc pass over {PMC_PASS_PACKETS} packets behind the timed region)", True
                if l2 is None and args.config == 2 and args.source == "sersic" and not args.store_radiation_field and not args.explicit_absorption:
                    l2, l2_source = pmc_l2(name)
                if l2 and l2.get("TCC_REQ_sum"):
                    lane_steps_2e7 = (b["prop_lane_steps"] if name == "walkPropKernel" else b["peel_lane_steps"]) * (2e7 / b["packets"])
                    seconds_2e7 = ms * 1e-3 * (2e7 / b["packets"])
                    extra["dominant_kernel"].update({
                        "l2_requests_per_lane_step": l2["TCC_REQ_sum"] / lane_steps_2e7,
                        "l2_miss_frac": l2["TCC_MISS_sum"] / l2["TCC_REQ_sum"],
                        "lines_per_visit": l2["TCC_MISS_sum"] / lane_steps_2e7,
                        "line_rate_frac": l2["TCC_MISS_sum"] / seconds_2e7 / GATHER_NO_LOCALITY,
                        "valu_instructions_per_lane_step": l2["SQ_INSTS_VALU"] / lane_steps_2e7,
                        "l2_counters_measured_in_run": in_run, "l2_counters_source": l2_source})
            extra["serial_kernel_ms_per_step"] = {"walkPropKernel": b["prop_ms"] * scale, "walkPeelKernel2": b["peel_ms"] * scale,
                                                  "launch + transition + cycle start + scan": b["transition_side_ms"] * scale,
                                                  "segment": b["segment_ms"] * scale}
        return {**extra, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "kernel_ms": mean_ms, "segment_ms": mean_ms,
                "walk_kernel_ms_sum": sum(t["walk_ms"] for t in m["timings"]) / len(m["timings"]),
                "transition_kernel_ms": sum(t["transition_ms"] for t in m["timings"]) / len(m["timings"]),
                "generations": sum(t["generations"] for t in m["timings"]) / len(m["timings"]),
                "cell_visits_per_packet": V / n, "detector_updates_per_packet": U / n,
                "rewalk_visits_per_packet": m["counters"]["rewalk_visits"] / launches / n,
                "algorithmic_bytes_per_packet": bytes_per_launch / n, "algorithmic_bytes_per_visit": bytes_per_visit}

    # behind the timed region (N = 1): hardware counters of this build on this box -- one step of 2e7 packets re-run under rocprofv3 --pmc on
    # the scene that the timed run saves
    counter_scene = None
    if rank == 0 and world == 1 and not args.no_counters and not args.scene_file:
        fd, counter_scene = tempfile.mkstemp(prefix="pmc_scene_", suffix=".bin", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        os.close(fd)
        os.chmod(counter_scene, 0o600)
    main_run = measure(ski_path, args.steps, args.warmup, save_scene_to=counter_scene)
    counters_in_run = None
    if counter_scene:
        try:
            passthrough = ["--config", str(args.config)] if args.config != 2 else []
            if args.config == 5:
                passthrough += ["--sites", str(args.sites)]
            counters_in_run = pmc_in_run(counter_scene, passthrough)
        finally:
            os.unlink(counter_scene)
    # the second source north_star names, measured in the same run on the same octree (default workload only): three steps
    secondary = None
    if args.config == 2 and args.source == "sersic" and not args.store_radiation_field and not args.explicit_absorption and args.ski == SKI and not args.no_secondary \
            and (world == 1 or args.secondary):
        text = open(SKI).read()
        new = ('<UniformBoxGeometry minX="-10000 pc" maxX="10000 pc" minY="-10000 pc" maxY="10000 pc" '
               'minZ="-1000 pc" maxZ="1000 pc"/>')
        text, nsub = re.subn(r"<SersicGeometry[^>]*/>", new, text, count=1)
        if nsub == 1:
            upath = os.path.join(tempfile.mkdtemp(prefix=f"bench_u_r{rank}_"), "cfg2u.ski")
            open(upath, "w").write(text)
            secondary = measure(upath, 3, 1)

    if rank == 0:
        roof = roofline_of(main_run)
        if counters_in_run and any("FETCH_SIZE" in v for v in counters_in_run.values()):
            # measured behind the timed region of THIS run: FETCH_SIZE + WRITE_SIZE of one step of PMC_PASS_PACKETS packets, scaled by the
            # packet count (the unit and what a miss moves: FETCH_UNIT_BYTES, profiles/README.md "FETCH_SIZE calibration")
            scale = FETCH_UNIT_BYTES / PMC_PASS_PACKETS * main_run["packets_this_rank"]
            per_kernel = {k: (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * scale for k, v in counters_in_run.items()}
            roof["traffic"] = sum(b for k, b in per_kernel.items() if k in WALK_KERNELS)
            roof["traffic_all_kernels"] = sum(per_kernel.values())
            roof["traffic_measured_in_run"] = True
            roof["traffic_per_kernel"] = {k: {"fetch": v.get("FETCH_SIZE", 0.0) * scale, "write": v.get("WRITE_SIZE", 0.0) * scale}
                                          for k, v in sorted(counters_in_run.items()) if "FETCH_SIZE" in v}
            roof["traffic_over_algorithmic"] = roof["traffic"] / (roof["algorithmic_bytes_per_packet"] * main_run["packets_this_rank"])
            # the other kernels (transition side, sorts, logs) read in wide coalesced streams, which FETCH_SIZE counts at HALF their bytes
            # (fetch_calibration_mi355x.txt: 0.500 per byte): their bytes with that correction, next to the figure as counted
            roof["traffic_other_kernels_fetch_doubled"] = sum((2.0 * v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * scale
                                                              for k, v in counters_in_run.items() if k not in WALK_KERNELS)
            roof["traffic_source"] = (f"this run: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE passes (one each) over one step of {PMC_PASS_PACKETS} packets of this "
                                      "workload behind the timed region, scaled by the packet count; `traffic` = the walk kernels, `traffic_all_kernels` = "
                                      "every kernel of the step; memory-side (fabric) requests of the L2s: Infinity-Cache hits included")
        else:
            traffic, traffic_all, traffic_source = pmc_traffic(main_run["packets_this_rank"]) if (args.ski == SKI and args.source == "sersic" and args.config == 2) else (None, None, None)
            roof["traffic"] = traffic
            roof["traffic_all_kernels"] = traffic_all
            roof["traffic_measured_in_run"] = False
            roof["traffic_source"] = (traffic_source + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/run_profile_set.sh) over a step of 2e7 packets, "
                                      "scaled by the packet count; `traffic` = the walk kernels, `traffic_all_kernels` = every kernel of the step") if traffic_source else None
        if counters_in_run:
            # every walk kernel's L2 picture from the same passes: requests, the share that missed, and the 64-byte requests to the fabric (a gather
            # that misses is ONE such request: profiles/microbench/fetch_calibration_mi355x.txt) -- `frac` prices algorithmic bytes against the HBM
            # peak whether the kernel's lines come from HBM, from the Infinity Cache or (l2_miss_frac small) from L2
            roof["walk_kernels_l2"] = {k: {"l2_requests": v["TCC_REQ_sum"] * main_run["packets_this_rank"] / PMC_PASS_PACKETS,
                                           "l2_miss_frac": v.get("TCC_MISS_sum", 0.0) / v["TCC_REQ_sum"],
                                           "valu_instructions": v.get("SQ_INSTS_VALU", 0.0) * main_run["packets_this_rank"] / PMC_PASS_PACKETS}
                                       for k, v in sorted(counters_in_run.items()) if k in WALK_KERNELS and v.get("TCC_REQ_sum")}
        if args.config == 5:
            roof["note"] = ("Voronoi: `frac` prices the bytes every walk has to read (32 B per visit + 28 B per neighbour entry that can be the exit) against the HBM peak; it is NOT "
                            "an HBM utilisation: the peel-off walks (60 % of the walk time) take their lines from L2 in tile order -- see walk_kernels_l2.*.l2_miss_frac -- and are "
                            "bound by the L1's access rate and the VALU; the propagation walks run at the chip's rate of lines from beyond L2")
        roof["kernel"] = ("voroPropKernel + voroPeelKernel (Voronoi)" if args.config == 5 else "walkPeelKernel + walkPropKernel (octree)") + \
                         ": all launches of one step, overlapped on the slot groups' streams (denominator: segment_ms)"
        value = total_per_step * args.steps / main_run["elapsed"]
        out = {
            "metric": "photon packets/s (whole node), 10^6-cell octree",
            "value": value,
            "unit": "photon packets/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * main_run["elapsed"] / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: " + ("Sersic" if args.source == "sersic" else "uniform-box")
                                    + f" source, {main_run['cells']}-cell PolicyTreeSpatialGrid octree (exp-disk dust, tau_z=1)"
                                    if args.config == 2 else
                                    f"BASELINE configs[2]: Sersic source, {main_run['cells']}-cell octree, panchromatic 0.1-10 micron, 50-bin "
                                    "wavelength grid, tabulated dust mix (2102-point opacity table)"
                                    if args.config == 3 else
                                    f"BASELINE configs[4]: Sersic source, VoronoiMeshSpatialGrid of {main_run['cells']} sites (tools/make_sites.py "
                                    f"--n {int(args.sites)} --seed 1), panchromatic 0.1-10 micron, 20 bins, THREE FullInstruments 256^2"
                                    if args.config == 5 else
                                    "BASELINE configs[3]: Sersic source, dust imported from 10^6 smoothed particles "
                                    f"(tools/make_sph.py --n 1000000 --seed 1), {main_run['cells']}-cell density-policy octree")
                                   + (", 0.55 micron" if args.config not in (3, 5) else "") + ", forced scattering, peel-off"
                                   + (" to one FullInstrument 512^2" if args.config != 5 else "")
                                   + " (components + statistics), " + os.path.relpath(args.ski, ROOT),
                       "packets_per_step": total_per_step,
                       "packets_per_step_per_gpu": total_per_step / world,
                       "cells": main_run["cells"],
                       "store_radiation_field": bool(args.store_radiation_field),
                       "explicit_absorption": bool(args.explicit_absorption),
                       "parallelism": f"history-range x{world}"},
            "roofline": roof,
        }
        if world > 1 or comm is not None:
            # what summed the ranks' detector arrays, and what RCCL itself reports about the communicator
            out["reduce"] = ("pmc_reduce_frames: one ncclReduce (f64, sum) per step on the engine's stream, communicator built by the "
                             "engine library (pmc_comm_unique_id / pmc_comm_init_rank)") if comm is not None else \
                            "torch.distributed all_reduce over gloo (BENCH_SHARE_DEVICE flow check: all ranks on one device)"
            out["nccl_ranks"] = nccl_ranks if comm is not None else 0
            out["reduce_check"] = main_run["reduce_check"]
        if secondary is not None:
            sroof = roofline_of(secondary)
            out["secondary"] = {"workload": "the same octree with a UniformBoxGeometry source of +-10 x +-10 x +-1 kpc (north_star's second "
                                            "source), 3 steps after 1 warm-up, same packet count per step",
                                "value": total_per_step * 3 / secondary["elapsed"], "unit": "photon packets/s",
                                "ms_per_step": 1e3 * secondary["elapsed"] / 3,
                                "roofline": {k: sroof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "segment_ms",
                                                                   "cell_visits_per_packet", "detector_updates_per_packet",
                                                                   "algorithmic_bytes_per_packet")}}
        if not args.no_cpu_baseline:
            # (N > 1: on rank 0's host cores behind the timed region, while the other ranks wait at the final barrier)
            out["cpu_baseline"] = cpu_baseline(args.ski if args.source == "sersic" else ski_path, os.environ.get("SKH_INPUT_PATH"))
        else:
            out["cpu_baseline"] = None
    else:
        out = None
    if world > 1:
        # what every rank measured by itself: its packets, the mean GPU time of its segments, the time its reduce call took (it includes
        # waiting for the slowest rank), and the time until its engine existed (rank 0: the scene set-up; the others: waiting for it and
        # loading the scene file)
        mine = {"rank": rank, "packets_per_step": main_run["packets_this_rank"], "segment_ms": mean_ms_of(main_run), "reduce_ms": main_run["reduce_ms"],
                "setup_s": main_run["setup_s"]}
        every = [None] * world
        dist.all_gather_object(every, mine)
        if rank == 0:
            out["per_rank"] = every
    if rank == 0:
        line = json.dumps(out)
    else:
        line = None
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    # the JSON line is the LAST thing on stdout: whatever the communication libraries have printed through C stdio (version
    # banners) is flushed first
    import ctypes
    ctypes.CDLL(None).fflush(None)
    if line is not None:
        print(line, flush=True)


if __name__ == "__main__":
    main()
