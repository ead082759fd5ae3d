#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracer core.

Workload (BASELINE.json configs[1], "C2"): box room + one kd-tree icosphere of 81,920 triangles, simple path
tracing depth 6, 1920x1080.  One step = one pass of the hot path over one batch: `--spp-per-step` samples of
every pixel on every rank (weak scaling: per-GPU work is fixed; the default K=32 x 8 spp = the config's 256 spp
per GPU).  Ranks render interleaved sample indices of the SAME frame and the fp32 framebuffer is all-reduced
once (RCCL) inside the timed region.  value = pixel-samples of all ranks / max-over-ranks wall time.

Extra objects on the JSON line:
  roofline     dominant kernel k_extend (kd closest hit): algorithmic bytes (SURVEY 8d formula, per-ray means from
               the in-kernel traversal counters of an untimed step of the same workload) / summed launch time (HIP
               events on the library's stream) vs 8 TB/s HBM.
  cpu_baseline the CPU oracle ("port") rendering a bounded pixel-subsampled slice of the same workload.
  closest_hit  (N=1 only) Mrays/s of mipt_trace_closest_device on the 1,310,720-triangle icosphere (C4 mesh) for a
               coherent and an incoherent batch of 2^24 rays, with the same roofline accounting.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def bytes_per_ray(nodes, refs, tris, hit_frac):
    """32 (ray in) + 16 (hit out) + 8*N_node + 4*N_ref + 36*N_tri + 36*[hit]."""
    return 32.0 + 16.0 + 8.0 * nodes + 4.0 * refs + 36.0 * tris + 36.0 * hit_frac


def cpu_threads():
    return max(1, min(16, os.cpu_count() or 1))


def closest_hit_bench(torch, G, Scene, FLAG_COUNTERS, n_rays, with_cpu):
    from oracle.binding import OracleScene
    out = {"mesh": "icosphere L8, 1310720 triangles, radius 2.5", "rays_per_batch": n_rays}
    desc = G.scene_env_sphere(8)
    desc.planes.clear()  # the mesh of C4 alone (SURVEY 8d micro-benchmark)
    t0 = time.time()
    s = desc.apply(Scene())
    s.commit(devices=(torch.cuda.current_device(),))
    out["kd_build_s"] = round(s.build_seconds(), 3)
    out["commit_s"] = round(time.time() - t0, 3)
    out["arena_MB"] = round(s.device_bytes() / 1e6, 1)
    batches = {
        "coherent": G.camera_rays(desc.camera, 3840, 2160, n_rays),
        "incoherent": G.incoherent_rays(n_rays, (-2.5,) * 3, (2.5,) * 3),
    }
    orc = None
    if with_cpu:
        orc = desc.apply(OracleScene(threads=cpu_threads()))
        orc.commit()
    for name, (ro, rd) in batches.items():
        d_o = torch.from_numpy(ro).cuda()
        d_d = torch.from_numpy(rd).cuda()
        d_p = torch.empty(n_rays, dtype=torch.int32, device="cuda")
        d_t = torch.empty(n_rays, dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_rays, d_p.data_ptr(), d_t.data_ptr())  # warm-up
        ms_ext, ms_all = [], []
        for _ in range(5):
            st = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_rays, d_p.data_ptr(), d_t.data_ptr())
            ms_ext.append(st["ms_extend"])
            ms_all.append(st["ms_total"])
        n_cpu = min(n_rays, 1 << 20)
        gp = d_p[:n_cpu].cpu().numpy()  # results of the timed (production) kernel, before the counting run overwrites them
        gt = d_t[:n_cpu].cpu().numpy()
        n_ctr = min(n_rays, 1 << 22)
        sc = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_ctr, d_p.data_ptr(), d_t.data_ptr(), counters=True)
        hit_frac = sc["ctr_hits"] / n_ctr
        means = {k: sc["ctr_" + k] / n_ctr for k in ("nodes", "pops", "refs", "tris")}
        bpr = bytes_per_ray(means["nodes"], means["refs"], means["tris"], hit_frac)
        ms = float(np.median(ms_ext))
        rec = {
            "Mrays_per_s": round(n_rays / ms / 1e3, 1),
            "Mrays_per_s_incl_ray_setup": round(n_rays / float(np.median(ms_all)) / 1e3, 1),
            "ms_extend": round(ms, 3),
            "per_ray": {k: round(v, 2) for k, v in means.items()}, "hit_frac": round(hit_frac, 4),
            "culled_candidate_frac": round(sc["ctr_culled"] / max(1, sc["ctr_tris"]), 4),
            "bytes_per_ray": round(bpr, 1),
            "reference_layout_bytes_per_ray": round(80 * means["nodes"] + 72 * means["tris"], 1),
            "roofline": {"bound": "hbm", "achieved": round(bpr * n_rays / (ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(bpr * n_rays / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None},
        }
        if orc is not None:
            t0 = time.time()
            op, ot, oc = orc.trace_closest(ro[:n_cpu], rd[:n_cpu])
            dt = time.time() - t0
            rec["cpu_oracle"] = {"Mrays_per_s": round(n_cpu / dt / 1e6, 3), "cores": cpu_threads(), "rays": n_cpu,
                                 "parity_prim_mismatches": int((gp != op).sum()),
                                 "parity_t_bit_mismatches": int((gt.view(np.uint32) != ot.view(np.uint32)).sum())}
        out[name] = rec
        del d_o, d_d, d_p, d_t
    s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--level", type=int, default=6, help="icosphere subdivision level (6 = 81,920 triangles)")
    ap.add_argument("--closest-rays", type=int, default=1 << 24)
    ap.add_argument("--no-closest", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from cuda_path_tracer_amd import generators as G
    from cuda_path_tracer_amd._lib import FLAG_COUNTERS
    from cuda_path_tracer_amd.parallel import rank_samples
    from cuda_path_tracer_amd.scene import Scene, make_config

    W, H, K, Wm, S = args.width, args.height, args.steps, args.warmup, args.spp_per_step
    desc = G.scene_icosphere_room(args.level)
    scene = desc.apply(Scene())
    scene.commit(devices=(local_rank,))
    spp_total = K * S * world  # normalisation constant of the timed render (kernel.cu:343)

    def step(i, accum, flags=0):
        b, e, stride = rank_samples(i * S * world, (i + 1) * S * world, rank, world)
        cfg = make_config(W, H, spp=spp_total, cam_depth=6, light_depth=0, sample_begin=b, sample_end=e,
                          sample_stride=stride, flags=flags)
        return scene.render_device(cfg, accum.data_ptr())

    scratch = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    accum = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for i in range(Wm):
        step(K + i, scratch)
    ctr = step(K + Wm, scratch, flags=FLAG_COUNTERS)  # untimed: per-ray traversal means of this workload
    if world > 1:
        dist.all_reduce(scratch)  # warm the communicator
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tot = {"rays_total": 0, "rays_extended": 0, "ms_extend": 0.0, "ms_total": 0.0, "extend_launches": 0, "samples": 0,
           "rays_camera": 0, "rays_shadow": 0}
    for i in range(K):
        st = step(i, accum)
        for k in tot:
            tot[k] += st[k]
    if world > 1:
        dist.all_reduce(accum)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    img_mean = float(accum.mean().item())

    if rank == 0:
        samples_all = W * H * S * K * world
        # per-ray means over the rays k_extend traced (rays that miss every mesh root box are answered by the stage that
        # generated them and never reach the kernel; they are not counted in its roofline)
        rays = max(1, ctr["rays_extended"])
        means = {k: ctr["ctr_" + k] / rays for k in ("nodes", "pops", "refs", "tris")}
        hit_frac = ctr["ctr_hits"] / rays
        bpr = bytes_per_ray(means["nodes"], means["refs"], means["tris"], hit_frac)
        ext_s = tot["ms_extend"] * 1e-3
        achieved = bpr * tot["rays_extended"] / ext_s / 1e9
        line = {
            "metric": "samples_per_sec", "value": round(samples_all / elapsed, 1), "unit": "pixel-samples/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: box room + icosphere L%d (%d tris), simple PT depth 6, %dx%d, %d spp per GPU"
                                   % (args.level, 20 * 4 ** args.level, W, H, K * S),
                       "spp_per_step": S, "integrator": "simple", "cam_depth": 6, "parallelism": "spp-interleave x%d + all-reduce" % world},
            "rays_per_sample": round(tot["rays_total"] / max(1, tot["samples"]), 3),
            "rays_through_extend_frac": round(tot["rays_extended"] / max(1, tot["rays_total"]), 4),
            "Mrays_per_s_rank0": round(tot["rays_total"] / elapsed / 1e6, 1),
            "extend_share_of_gpu_time": round(tot["ms_extend"] / max(1e-9, tot["ms_total"]), 3),
            "image_mean": round(img_mean, 6),
            "roofline": {"bound": "hbm", "kernel": "k_extend", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                         "bytes_per_ray": round(bpr, 1),
                         "reference_layout_bytes_per_ray": round(80 * means["nodes"] + 72 * means["tris"], 1),
                         "per_ray": {k: round(v, 2) for k, v in means.items()},
                         "launches": tot["extend_launches"],
                         "avg_launch_ms": round(tot["ms_extend"] / max(1, tot["extend_launches"]), 4),
                         "rays_per_launch": round(tot["rays_extended"] / max(1, tot["extend_launches"])),
                         "Mrays_per_s_in_kernel": round(tot["rays_extended"] / ext_s / 1e6, 1)},
        }
        # HBM bytes per k_extend launch: PMC counters cannot be read inside this process, so the figure is taken from the
        # committed summary of `scripts/pmc_bench.sh` (the same workload under separate rocprofv3 --pmc passes), if present
        pmc_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r1_pmc_bench_c2.json")
        if os.path.exists(pmc_path):
            try:
                tb = json.load(open(pmc_path))["traffic_bytes_per_launch"]
                line["roofline"]["traffic"] = round(tb["read_raw"] + tb["write"])
                line["roofline"]["traffic_unit"] = "bytes per launch (TCC FETCH_SIZE + WRITE_SIZE, raw)"
                line["roofline"]["traffic_source"] = "profiles/r1_pmc_bench_c2.json"
                line["roofline"]["algorithmic_bytes_per_launch"] = round(bpr * tot["rays_extended"] / max(1, tot["extend_launches"]))
                pm = json.load(open(pmc_path))
                # what actually bounds the kernel: VALU issue.  busy = 4 cycles per wave64 VALU instruction
                # over the kernel's cycles on 1024 SIMDs (GRBM_GUI_ACTIVE is summed over the 8 XCDs)
                line["roofline"]["valu_busy_frac"] = round(4.0 * pm["SQ_ACTIVE_INST_VALU"]["mean_per_launch"]
                                                           / (pm["GRBM_GUI_ACTIVE"]["mean_per_launch"] / 8.0 * 1024.0), 3)
                line["roofline"]["valu_lane_use"] = round(pm["valu_lane_utilisation"], 3)
            except (KeyError, ValueError):
                pass
        if not args.no_cpu_baseline and world == 1:
            from oracle.binding import OracleScene
            orc = desc.apply(OracleScene(threads=cpu_threads()))
            orc.commit()
            pstride = 61
            n_px = (W * H + pstride - 1) // pstride
            done, t_cpu, orays = 0, 0.0, 0
            octr = {"nodes": 0, "refs": 0, "tris": 0, "rays": 0}
            while t_cpu < args.cpu_seconds and done < S * K:
                cfg = make_config(W, H, spp=spp_total, cam_depth=6, light_depth=0, sample_begin=done, sample_end=done + 1)
                t1 = time.perf_counter()
                _, oc = orc.render(cfg, pixel_stride=pstride)
                t_cpu += time.perf_counter() - t1
                done += 1
                for k in octr:
                    octr[k] += oc[k]
            line["cpu_baseline"] = {
                "value": round(n_px * done / t_cpu, 1), "unit": "pixel-samples/s", "cores": cpu_threads(), "kind": "port",
                "sample": "every %dth pixel of the same %dx%d frame, %d spp, CPU oracle (g++ -O2 -fopenmp -ffp-contract=off)"
                          % (pstride, W, H, done),
                "seconds": round(t_cpu, 2),
                "per_ray": {k: round(octr[k] / max(1, octr["rays"]), 2) for k in ("nodes", "refs", "tris")},
            }
        if not args.no_closest and world == 1:
            del scratch
            line["closest_hit"] = closest_hit_bench(torch, G, Scene, FLAG_COUNTERS, args.closest_rays, not args.no_cpu_baseline)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
