#!/usr/bin/env python3
"""Tuning aid (GPU box): one scene setup, then one timed segment per engine variant.
usage: tools/sweep.py [--packets 2e7] [--ski tests/ski/cfg2.ski] VARIANT...
  VARIANT = lib[,KEY=VALUE...]   lib: file name under skirt9_amd/lib (or "default"); KEY=VALUE: environment of pmc_create
  e.g.  default,PMC_NUM_GROUPS=1  libpmc_r24.so,PMC_WALK_BLOCKS_PER_CU=4
Prints packets/s, segment / walk / transition ms and the generation count of each variant."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--packets", type=float, default=2e7)
    ap.add_argument("--ski", default=os.path.join(ROOT, "tests", "ski", "cfg2.ski"))
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    import torch
    from skirt9_amd import engine
    from skirt9_amd.host import Simulation
    P = int(args.packets)
    sim = Simulation(args.ski, num_packets=P).setup()
    frames = torch.zeros(sim.frame_size, dtype=torch.float64, device="cuda:0")
    keys = set()
    for v in args.variants:
        parts = v.split(",")
        for k in keys:
            os.environ.pop(k, None)
        os.environ.pop("PMC_LIBRARY", None)
        if parts[0] != "default":
            os.environ["PMC_LIBRARY"] = os.path.join(ROOT, "skirt9_amd", "lib", parts[0])
        for kv in parts[1:]:
            k, val = kv.split("=")
            os.environ[k] = val
            keys.add(k)
        engine._lib = None
        # (PMC_NUM_SLOTS / PMC_NUM_GROUPS / PMC_STAT_POOL_BLOCKS are library settings read from the environment; every other PMC_*
        # pair is a tuning switch, include/pmc_tuning.h)
        engine.clear_tuning()
        engine.tuning_from_environment()
        eng = engine.Engine(sim.scene, 0)
        frames.zero_()
        eng.bind_frames(frames.data_ptr(), frames.numel())
        eng.run_primary(0, P // 10, sim.seed)  # warm-up
        eng.sync()
        best = None
        eng.reset_counters()
        for rep in range(2):
            eng.run_primary((rep + 1) * P, P, sim.seed)
            eng.sync()
            t = eng.last_timing()
            try:
                t["kernels"] = eng.last_walk_timing()
            except Exception:  # noqa: BLE001
                t["kernels"] = None
            if best is None or t["total_ms"] < best["total_ms"]:
                best = t
        print(f"{v:60s} pkt/s {P / best['total_ms'] * 1e3:.3e} seg {best['total_ms']:.1f} walk {best['walk_ms']:.1f} "
              f"trans {best['transition_ms']:.1f} gen {best['generations']}  sum {frames.sum().item():.9e}", flush=True)
        try:
            # octree: the two walk kernels apart (spans overlap unless PMC_SERIAL_WALKS), lane-steps counted over both timed segments
            w, k = eng.walk_work(), best["kernels"]
            if w["prop_wave_steps"]:
                print(f"    peel {k['peel_ms']:.1f} ms, {w['peel_lane_steps'] / 2 / P:.1f} lane-steps/packet, "
                      f"{w['peel_lane_steps'] / 2 / k['peel_ms'] / 1e8:.3f}e11 lane-steps/s, lanes {w['peel_lane_steps'] / 0.64 / w['peel_wave_steps']:.1f} %;  "
                      f"prop {k['prop_ms']:.1f} ms, {w['prop_lane_steps'] / 2 / P:.1f} lane-steps/packet, "
                      f"{w['prop_lane_steps'] / 2 / k['prop_ms'] / 1e8:.3f}e11 lane-steps/s, lanes {w['prop_lane_steps'] / 0.64 / w['prop_wave_steps']:.1f} %",
                      flush=True)
        except Exception as exc:  # noqa: BLE001 - variants built before the ABI had these entries
            print(f"    (no per-kernel figures: {exc})", flush=True)
        if os.environ.get("PMC_PROFILE_DUMP"):
            eng.counters()  # a profiling build prints its in-kernel timers to stderr
        eng.close()


if __name__ == "__main__":
    main()
