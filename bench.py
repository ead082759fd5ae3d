#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracer core.

Workload (BASELINE.json configs[1], "C2"): box room + one kd-tree icosphere of 81,920 triangles, simple path
tracing depth 6, 1920x1080.  One step = one pass of the hot path over one batch: `--spp-per-step` samples of
every pixel on every rank (weak scaling: per-GPU work is fixed; the default 4 steps x 64 spp = the config's 256 spp
per GPU).  Ranks render interleaved sample indices of the SAME frame and the fp32 framebuffer is all-reduced once
(RCCL) inside the timed region.  value = pixel-samples of all ranks / max-over-ranks wall time.

Launch: `python bench.py --gpus N` starts N rank processes itself (before anything touches a GPU) when it is not
already running under torchrun (`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`); it
refuses to run with fewer than N visible devices and never reports n_gpus != N.

Extra objects on the JSON line:
  roofline     dominant kernel k_extend (kd closest hit), three views that each are a fraction of a real limit:
               `hbm`   memory-side bytes (L2 misses x 128 B, reads and writes; rocprofv3 --pmc passes run by this script in
                       child processes before the timed part; calibration: profiles/r2_gather_calibration.txt) / launch
                       time (HIP events) vs 8 TB/s;  `valu`  VALU busy fraction and lane use from the same passes;
               `algorithmic` what the production kernel reads by its own layout (counted by k_extend<kCountWork> in an
                       untimed step) -- memory-side / algorithmic < 1 is what the caches absorbed.
               roofline.bound names what binds (VALU issue busy vs memory side / the 6.3 TB/s a copy reaches); for a
               VALU-bound kernel roofline.frac = busy x lane use (useful lane throughput / peak); `reference_model` keeps SURVEY 8d's
               reference-traversal byte model as a throughput normalisation (it is not a fraction of anything).
  cpu_baseline the CPU oracle ("port") rendering a bounded pixel-subsampled slice of the same workload.
  closest_hit  (N=1 only) Mrays/s of mipt_trace_closest_device on the 1,310,720-triangle icosphere (C4 mesh) for a
               coherent and an incoherent batch of 2^24 rays, with the same three roofline views each.
  strong       (N>1 only) the same frame at FIXED total spp (what one GPU renders in the weak run) split over N ranks.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_MEASURED_COPY_GBS = 6300.0  # same guide: ~6.3 TB/s achievable by a streaming copy
LINE_BYTES = 128               # one L2 miss fills one 128-byte line, for gathers as for streams (profiles/r2_gather_calibration.txt)

PMC_PASSES = (
    ("FETCH_SIZE", "TCC_MISS_sum"),
    ("WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"),
    ("SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "GRBM_GUI_ACTIVE"),
)


def bytes_per_ray(nodes, refs, tris, hit_frac):
    """SURVEY.md 8d (reference traversal): 32 (ray in) + 16 (hit out) + 8*N_node + 4*N_ref + 36*N_tri + 36*[hit]."""
    return 32.0 + 16.0 + 8.0 * nodes + 4.0 * refs + 36.0 * tris + 36.0 * hit_frac


def product_bytes_per_ray(st, rays, origin_bytes=16.0, queue_bytes=4.0):
    """What k_extend<kProd> reads and writes per ray by its own layout: per node step ONE 64-byte step
    record (both children's cull boxes and their 8-byte nodes); per stack pop / mesh root one 8-byte node + one 32-byte box; 32 bytes per streamed
    candidate-sphere pair; 16 bytes per marked candidate's plane + 48 for the vertices of those that reach the edge
    tests; ray in (16 direction [+ 16 origin] [+ 4 queue index] + 8 parked sphere/plane answer) and 8 hit out."""
    rays = max(1, rays)
    per = {k: st["work_" + k] / rays for k in ("steps", "pops", "sphere_pairs", "planes", "verts", "leaves")}
    b = (64.0 * per["steps"] + 40.0 * per["pops"] + 32.0 * per["sphere_pairs"] + 16.0 * per["planes"] + 48.0 * per["verts"]
         + 16.0 + origin_bytes + queue_bytes + 8.0 + 8.0)
    return b, per


def cpu_threads():
    return max(1, min(16, os.cpu_count() or 1))


# ----------------------------------------------------------------------------------------------- PMC passes
def collect_pmc(args, log):
    """rocprofv3 --pmc passes over scripts/pmc_worker.py (the same C2 step and closest-hit batches), each in a child
    process of its own, BEFORE this process touches the GPU.  Returns {part: {counter: mean per k_extend<0> launch}}
    or (None, reason)."""
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    out = tempfile.mkdtemp(prefix="mipt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    worker = os.path.join(ROOT, "scripts", "pmc_worker.py")
    sums = {}
    manifest = None
    for i, ctrs in enumerate(PMC_PASSES):
        mf = os.path.join(out, "manifest%d.json" % i)
        cmd = ["timeout", "-s", "KILL", str(int(args.pmc_timeout)), "rocprofv3", "--pmc", *ctrs, "--output-format", "csv",
               "-d", os.path.join(out, "p%d" % i), "--", sys.executable, worker, mf, "--spp", str(args.spp_per_step),
               "--width", str(args.width), "--height", str(args.height), "--level", str(args.level)]
        if args.no_closest:
            cmd.append("--no-closest")
        t0 = time.time()
        with open(os.path.join(out, "p%d.log" % i), "w") as lf:
            rc = subprocess.call(cmd, cwd="/tmp", env=env, stdout=lf, stderr=subprocess.STDOUT)
        log("pmc pass %d (%s): rc %d, %.1f s" % (i, " ".join(ctrs), rc, time.time() - t0))
        if rc != 0 or not os.path.exists(mf):
            return None, "pmc pass %d failed (rc %d, see %s)" % (i, rc, out)
        manifest = json.load(open(mf))["parts"]
        rows, shade_rows = [], []
        for f in glob.glob(os.path.join(out, "p%d" % i, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row["Kernel_Name"]
                if "k_extend<0>" in name or "k_extend<(int)0>" in name:
                    rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
                elif "k_shade" in name:  # only the C2 part of the worker shades
                    shade_rows.append((int(row["Dispatch_Id"]), row["Counter_Name"], float(row["Counter_Value"])))
        if shade_rows:
            disp = {}
            for d, c, v in shade_rows:
                disp.setdefault(d, {})[c] = disp.get(d, {}).get(c, 0.0) + v
            dst = sums.setdefault("c2_shade", {"launches": len(disp)})
            for c in ctrs:
                dst[c] = sum(x.get(c, 0.0) for x in disp.values()) / max(1, len(disp))
        by_disp = {}
        for d, c, v in rows:
            by_disp.setdefault(d, {})[c] = by_disp.get(d, {}).get(c, 0.0) + v
        order = sorted(by_disp)
        if len(order) != sum(p["launches"] for p in manifest):
            return None, "pmc pass %d: %d k_extend dispatches, manifest says %d" % (i, len(order), sum(p["launches"] for p in manifest))
        at = 0
        for p in manifest:
            sel = order[at:at + p["launches"]]
            at += p["launches"]
            dst = sums.setdefault(p["name"], {"launches": p["launches"], "rays": p["rays"]})
            for c in ctrs:
                dst[c] = sum(by_disp[d].get(c, 0.0) for d in sel) / max(1, len(sel))
    shutil.rmtree(out, ignore_errors=True)
    return sums, "live: rocprofv3 --pmc passes over scripts/pmc_worker.py in this run"


def roofline_views(pmc, launch_ms, rays_per_launch, alg_bpr, ref_bpr, extra=None):
    """The three views of one kernel launch (see the module docstring).  pmc: per-launch counter means or None."""
    sec = launch_ms * 1e-3
    alg_bytes = alg_bpr * rays_per_launch
    r = {"kernel": "k_extend", "avg_launch_ms": round(launch_ms, 4), "rays_per_launch": int(round(rays_per_launch)),
         "Mrays_per_s_in_kernel": round(rays_per_launch / sec / 1e6, 1),
         "algorithmic": {"bytes_per_ray": round(alg_bpr, 1), "bytes_per_launch": int(round(alg_bytes)),
                         "GB_per_s": round(alg_bytes / sec / 1e9, 1),
                         "note": "product-layout bytes the kernel asks for (k_extend<kCountWork>); served by L2 / Infinity Cache / HBM"},
         "reference_model": {"bytes_per_ray": round(ref_bpr, 1), "GB_per_s": round(ref_bpr * rays_per_launch / sec / 1e9, 1),
                             "x_hbm_peak": round(ref_bpr * rays_per_launch / sec / 1e9 / HBM_PEAK_GBS, 3),
                             "note": "SURVEY 8d byte model of the REFERENCE's traversal over the same rays: a throughput "
                                     "normalisation, not a fraction of a limit (the product skips ~90 % of those candidates)"}}
    if extra:
        r.update(extra)
    if not pmc:
        r.update({"bound": "hbm", "achieved": round(alg_bytes / sec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": None, "traffic": None, "note": "no PMC counters in this run: memory-side traffic unknown, frac withheld"})
        return r
    # the PMC passes ran a launch of the same shape; scale their per-launch counters to this launch by ray count
    scale = rays_per_launch / max(1.0, pmc["rays"] / max(1, pmc["launches"]))
    # TCC_MISS counts read AND write misses (streaming writes all miss): on k_shade, which writes as much as it reads,
    # FETCH_SIZE x 2 + WRITE_SIZE = 19.2 GB and TCC_MISS x 128 B = 19.4 GB per launch (profiles/r2_pmc_shading_kernels.txt)
    # -- so the lines out of L2 are the whole memory-side traffic; WRITE_SIZE is reported beside it, not added to it.
    write_bytes = pmc["WRITE_SIZE"] * 1024.0 * scale
    traffic = pmc["TCC_MISS_sum"] * LINE_BYTES * scale
    hbm_gbs = traffic / sec / 1e9
    busy = 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    lane = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
    r["hbm"] = {"achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                "frac_of_measured_copy_peak": round(hbm_gbs / HBM_MEASURED_COPY_GBS, 4),
                "bytes_per_ray": round(traffic / rays_per_launch, 1),
                "l2_hit_rate": round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4),
                "fetch_size_x2_bytes": int(round(pmc["FETCH_SIZE"] * 1024.0 * 2.0 * scale)), "write_size_bytes": int(round(write_bytes)),
                "note": "L2 misses (reads and writes) x 128 B per launch / HIP-event launch time; Infinity Cache hits are included "
                        "(the counters sit on the L2's memory side)"}
    r["valu"] = {"busy_frac": round(busy, 4), "lane_use": round(lane, 4), "useful_lane_frac": round(busy * lane, 4),
                 "note": "busy = 4 x SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lane use = SQ_THREAD_CYCLES_VALU / "
                         "(64 x SQ_ACTIVE_INST_VALU)"}
    r["algorithmic"]["memory_side_over_algorithmic"] = round(traffic / max(1.0, alg_bytes), 4)
    r["traffic"] = int(round(traffic))
    r["traffic_unit"] = "bytes per launch (TCC_MISS_sum x 128 B: read and write misses of L2)"
    # which resource binds: the VALU issue slots (busy) or the memory side (against the ~6.3 TB/s a copy achieves).  The
    # fraction reported for a VALU-bound kernel is USEFUL lane throughput / peak lane throughput = busy x lane use: an
    # issue slot spent on a wave with most lanes masked off is not progress.
    # (memory-side bytes above the HBM peak can only have come from the Infinity Cache: then HBM is not what binds)
    if busy >= hbm_gbs / HBM_MEASURED_COPY_GBS or hbm_gbs > HBM_PEAK_GBS:
        r.update({"bound": "valu", "achieved": round(busy * lane, 4), "peak": 1.0,
                  "unit": "fraction of peak VALU lane throughput (issue-slot busy x lanes active)", "frac": round(busy * lane, 4),
                  "issue_busy_frac": round(busy, 4)})
    else:
        r.update({"bound": "memory side (L2 miss lines, Infinity Cache included)", "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS,
                  "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                  "frac_of_measured_copy_peak": round(hbm_gbs / HBM_MEASURED_COPY_GBS, 4),
                  "bound_note": "lines out of L2 per second against the 8 TB/s HBM peak and the 6.3 TB/s a copy reaches; part of them "
                                "is served by the 256 MB Infinity Cache (the counters sit on the L2's memory side), so this is the "
                                "memory-side ceiling, not an HBM fraction"})
    return r


def shade_roofline(pmc, launch_ms, vertices_per_launch):
    """k_shade (camera-vertex shading: surface, direct connection, roulette, BSDF sample, queue appends): the second
    largest kernel of a render.  Memory-side bytes, VALU issue and lane use from the same PMC passes; time from HIP events."""
    r = {"kernel": "k_shade", "avg_launch_ms": round(launch_ms, 4), "shaded_vertices_per_launch": int(round(vertices_per_launch))}
    if not pmc or launch_ms <= 0:
        r.update({"bound": "hbm", "frac": None, "traffic": None, "note": "no PMC counters in this run"})
        return r
    sec = launch_ms * 1e-3
    traffic = pmc["TCC_MISS_sum"] * LINE_BYTES
    gbs = traffic / sec / 1e9
    busy = 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (pmc["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    lane = pmc["SQ_THREAD_CYCLES_VALU"] / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
    r.update({"hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                      "frac_of_measured_copy_peak": round(gbs / HBM_MEASURED_COPY_GBS, 4),
                      "bytes_per_shaded_vertex": round(traffic / max(1.0, vertices_per_launch), 1),
                      "l2_hit_rate": round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc["TCC_MISS_sum"]), 4),
                      "fetch_size_x2_bytes": int(round(pmc["FETCH_SIZE"] * 1024.0 * 2.0)), "write_size_bytes": int(round(pmc["WRITE_SIZE"] * 1024.0))},
              "valu": {"busy_frac": round(busy, 4), "lane_use": round(lane, 4), "useful_lane_frac": round(busy * lane, 4),
                       "insts_per_shaded_vertex": round(pmc["SQ_INSTS_VALU"] * 64.0 * lane / max(1.0, vertices_per_launch), 1)},
              "traffic": int(round(traffic))})
    if busy >= gbs / HBM_MEASURED_COPY_GBS:
        r.update({"bound": "valu", "achieved": round(busy * lane, 4), "peak": 1.0, "unit": "fraction of peak VALU lane throughput", "frac": round(busy * lane, 4)})
    else:
        r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)})
    return r


# ----------------------------------------------------------------------------------------------- the other BASELINE configurations
def env_faces_1024(G):
    rng = np.random.Generator(np.random.Philox(key=G.SEED))
    base = rng.random((6, 64, 64, 3), dtype=np.float32)
    return np.ascontiguousarray(np.repeat(np.repeat(base, 16, axis=1), 16, axis=2)) * np.float32(2.0)


def other_config(name, G, cube):
    """(scene description, render keywords, text) of BASELINE.json configs C1, C3, C4, C5 at a bounded sample count (samples are
    independent, so samples/s does not depend on spp)."""
    if name == "C1":
        return G.scene_default(cube), dict(width=640, height=480, spp=16, cam_depth=6, light_depth=0), "default scene, simple d6, 640x480, 16 spp"
    if name == "C3":
        return G.scene_default(cube), dict(width=1024, height=1024, spp=200, cam_depth=3, light_depth=3), "default scene, bidirectional 3+3, 1024x1024, 200 spp"
    if name == "C4":
        return (G.scene_env_sphere(8, env_f32=env_faces_1024(G)), dict(width=3840, height=2160, spp=16, cam_depth=6, light_depth=0, use_envmap=True),
                "env map + icosphere L8 (1,310,720 tris), simple d6, 4K, 16 of 1024 spp")
    if name == "C4L":  # not a BASELINE config: C4 with the cube map ALSO sampled as a light (use_envmap == 2, build-defined)
        return (G.scene_env_sphere(8, env_f32=env_faces_1024(G)), dict(width=3840, height=2160, spp=16, cam_depth=6, light_depth=0, use_envmap=2),
                "C4 with the env map sampled as a light (use_envmap = 2), 16 spp")
    if name == "C5":
        return (G.scene_multi_sphere(8), dict(width=3840, height=2160, spp=4, cam_depth=5, light_depth=5),
                "8 x icosphere L8 (10,485,760 tris), bidirectional 5+5, 4K, 4 of 4096 spp")
    raise KeyError(name)


def run_other_config(name, G, Scene, make_config, device=0):
    import tempfile as _tf
    cube = G.write_cube_obj(os.path.join(_tf.mkdtemp(), "cube.obj"))
    desc, kw, text = other_config(name, G, cube)
    t0 = time.time()
    s = desc.apply(Scene())
    s.commit((device,))
    t_commit = time.time() - t0
    s.render(make_config(**{**kw, "sample_end": 1}))  # warm-up: one sample per pixel
    t0 = time.time()
    img, st = s.render(make_config(**kw))
    wall = time.time() - t0
    out = {"config": name, "workload": text, "samples_per_sec": round(st["samples"] / (st["ms_total"] * 1e-3), 1),
           "Mrays_per_s": round(st["rays_total"] / (st["ms_total"] * 1e-3) / 1e6, 1),
           "rays_per_sample": round(st["rays_total"] / st["samples"], 2), "ms_gpu": round(st["ms_total"], 1),
           "extend_share": round(st["ms_extend"] / st["ms_total"], 3), "shade_share": round(st["ms_shade"] / st["ms_total"], 3),
           "wall_s_incl_readback": round(wall, 3),
           "commit_s": round(t_commit, 2), "kd_build_s": round(s.kd_seconds(), 2), "pack_s": round(s.build_seconds() - s.kd_seconds(), 2),
           "arena_MB": round(s.device_bytes() / 1e6, 1),
           "image_mean_finite_pixels": float(img[np.isfinite(img).all(axis=2)].mean()),
           "nonfinite_pixels": int((~np.isfinite(img).all(axis=2)).sum()), "nonfinite_samples": int(st["nonfinite_samples"])}
    s.close()
    return out


# ----------------------------------------------------------------------------------------------- closest hit
def closest_hit_bench(torch, G, Scene, n_rays, with_cpu, pmc):
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
        s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_rays, d_p.data_ptr(), d_t.data_ptr())  # warm-up
        ms_ext, ms_all = [], []
        for _ in range(5):
            st = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_rays, d_p.data_ptr(), d_t.data_ptr())
            ms_ext.append(st["ms_extend"])
            ms_all.append(st["ms_total"])
        n_cpu = min(n_rays, 1 << 20)
        gp = d_p[:n_cpu].cpu().numpy()  # results of the timed (production) kernel, before the counting runs overwrite them
        gt = d_t[:n_cpu].cpu().numpy()
        n_ctr = min(n_rays, 1 << 22)
        sc = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_ctr, d_p.data_ptr(), d_t.data_ptr(), counters=True)
        sw = s.trace_closest_device(d_o.data_ptr(), d_d.data_ptr(), n_ctr, d_p.data_ptr(), d_t.data_ptr(), work_counters=True)
        hit_frac = sc["ctr_hits"] / n_ctr
        means = {k: sc["ctr_" + k] / n_ctr for k in ("nodes", "pops", "refs", "tris")}
        ref_bpr = bytes_per_ray(means["nodes"], means["refs"], means["tris"], hit_frac)
        alg_bpr, per = product_bytes_per_ray(sw, n_ctr, origin_bytes=16.0, queue_bytes=0.0)
        ms = float(np.median(ms_ext))
        rec = {
            "Mrays_per_s": round(n_rays / ms / 1e3, 1),
            "Mrays_per_s_incl_ray_setup": round(n_rays / float(np.median(ms_all)) / 1e3, 1),
            "ms_extend": round(ms, 3),
            "reference_walk_per_ray": {k: round(v, 2) for k, v in means.items()}, "hit_frac": round(hit_frac, 4),
            "culled_candidate_frac": round(sc["ctr_culled"] / max(1, sc["ctr_tris"]), 4),
            "product_walk_per_ray": {k: round(v, 2) for k, v in per.items()},
            "roofline": roofline_views(pmc.get(name) if pmc else None, ms, n_rays, alg_bpr, ref_bpr),
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


# ----------------------------------------------------------------------------------------------- launching ranks
def launch_ranks(n, argv, share_gpu=False):
    """`python bench.py --gpus N` outside torchrun: start N rank processes (fresh interpreters; this parent never touches
    a GPU) and exit with the worst of their codes.  Rank 0 prints the JSON line on our stdout."""
    import torch  # device_count() does not initialise the GPU
    have = torch.cuda.device_count()
    if share_gpu and have >= 1:
        have = n  # rehearsal: every rank uses device 0 (gloo only; RCCL refuses two ranks on one GPU)
    if have < n:
        sys.stderr.write("bench.py: --gpus %d asked for, %d visible device(s): refusing to run a smaller job under that label\n" % (n, have))
        return 3
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    worst = 0
    try:
        while procs:
            for p in list(procs):
                rc = p.poll()
                if rc is None:
                    continue
                procs.remove(p)
                if rc != 0:
                    worst = worst or rc
                    for q in procs:  # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            p.kill()
    return worst


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
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline.frac is then withheld)")
    ap.add_argument("--no-strong", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short runs of BASELINE configs C1, C3, C4, C5 (N=1 only)")
    ap.add_argument("--pmc-timeout", type=float, default=150.0)
    ap.add_argument("--save-pmc", default=None, help="write the per-launch counter means of the PMC passes to this JSON file")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo: collectives staged through the host)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal of the N > 1 path on a one-GPU box: every rank renders on device 0 (needs --backend gloo; "
                         "the line is labelled, its value is not a scaling number)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        if args.share_gpu and args.backend == "nccl":
            raise SystemExit("bench.py: --share-gpu needs --backend gloo (RCCL refuses two ranks on one GPU)")
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:], args.share_gpu))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (torchrun --nproc-per-node %d)"
                         % (args.gpus, world, args.gpus))

    def log(msg):
        if rank == 0:
            sys.stderr.write("[bench] %s\n" % msg)
            sys.stderr.flush()

    # PMC passes first: child processes own the GPU while this one has not initialised it yet
    pmc, pmc_source = None, "skipped (--no-pmc or N > 1)"
    if world == 1 and not args.no_pmc:
        pmc, pmc_source = collect_pmc(args, log)
        if pmc is not None and args.save_pmc:
            json.dump({"parts": pmc, "passes": [list(p) for p in PMC_PASSES], "note": "mean per k_extend<kProd> launch; rocprofv3 --pmc, one pass per list"},
                      open(args.save_pmc, "w"), indent=1)
        if pmc is None:
            log("PMC collection failed: " + pmc_source)
            fallback = os.path.join(ROOT, "profiles", "r3_pmc_bench.json")
            if os.path.exists(fallback):
                pmc = json.load(open(fallback))["parts"]
                pmc_source = "profiles/r3_pmc_bench.json (committed; live collection failed: %s)" % pmc_source
    elif world > 1:
        # a scaling run collects no counters (N ranks under a profiler each): its roofline object carries the committed N = 1
        # counters, labelled as such, scaled to this run's launch shape and timed by this run's own HIP events
        committed = os.path.join(ROOT, "profiles", "r3_pmc_bench.json")
        if os.path.exists(committed):
            pmc = json.load(open(committed))["parts"]
            pmc_source = "profiles/r3_pmc_bench.json: counters of the committed N = 1 run (a run at N > 1 collects none); launch times are this run's"

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no device %d" % (rank, local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from cuda_path_tracer_amd import generators as G
    from cuda_path_tracer_amd._lib import FLAG_COUNTERS, FLAG_WORK_COUNTERS
    from cuda_path_tracer_amd.parallel import render_distributed, share_scene
    from cuda_path_tracer_amd.scene import Scene, make_config

    W, H, K, Wm, S = args.width, args.height, args.steps, args.warmup, args.spp_per_step

    def build():
        sc = G.scene_icosphere_room(args.level).apply(Scene())
        sc.commit(devices=())  # kd build + pack on the host, once per job
        return sc

    t_build = time.perf_counter()
    scene = share_scene(build, Scene, rank, world, device=dev if args.backend == "nccl" else None)
    scene.commit(devices=(local_rank,))  # packed already (built or imported): this only uploads the arena
    t_build = time.perf_counter() - t_build

    def renderer(spp_total, flags=0):
        def render_into(b, e, stride, accum):
            cfg = make_config(W, H, spp=spp_total, cam_depth=6, light_depth=0, sample_begin=b, sample_end=e, sample_stride=stride,
                              flags=flags)
            return scene.render_device(cfg, accum.data_ptr())
        return render_into

    def host_staged_all_reduce(t, op=None):  # gloo: reduce a host copy, write it back (rehearsals; RCCL reduces in place)
        h = t.cpu()
        dist.all_reduce(h) if op is None else dist.all_reduce(h, op=op)
        t.copy_(h)

    reduce_fn = dist.all_reduce if args.backend == "nccl" else host_staged_all_reduce
    all_reduce = reduce_fn if world > 1 else None
    scratch = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    accum = torch.zeros(W * H * 3, dtype=torch.float32, device="cuda")
    spp_total = K * S * world  # normalisation constant of the timed render (kernel.cu:343)

    # untimed: warm-up steps (incl. the collective), then the two counting steps of this workload on rank 0's share
    if Wm > 0:
        render_distributed(renderer(spp_total), scratch, 0, Wm * S * world, rank, world, all_reduce, samples_per_step=S * world)
    ctr = render_distributed(renderer(spp_total, FLAG_COUNTERS), scratch, 0, S * world, rank, world, all_reduce)[0]
    wrk = render_distributed(renderer(spp_total, FLAG_WORK_COUNTERS), scratch, 0, S * world, rank, world, all_reduce)[0]

    def timed(total_samples, samples_per_step, target):
        """barrier + sync | K steps + one all-reduce | sync + barrier; max over ranks."""
        target.zero_()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        stats = render_distributed(renderer(total_samples), target, 0, total_samples, rank, world, all_reduce,
                                   samples_per_step=samples_per_step)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
            reduce_fn(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, stats

    elapsed, steps = timed(spp_total, S * world, accum)
    assert len(steps) == K
    tot = {k: sum(st[k] for st in steps) for k in ("rays_total", "rays_extended", "ms_extend", "ms_total", "extend_launches",
                                                    "samples", "rays_camera", "rays_shadow", "nonfinite_samples", "ms_shade", "shade_launches")}
    img_mean = float(accum.mean().item())
    strong = None
    if world > 1 and not args.no_strong:
        # strong scaling: the frame ONE GPU renders in the weak run (K*S spp in all), split over the ranks
        dt, sst = timed(K * S, S, scratch)
        strong = {"value": round(W * H * K * S / dt, 1), "unit": "pixel-samples/s", "total_spp": K * S, "steps": len(sst),
                  "ms_total": round(dt * 1e3, 3), "scaling": "strong",
                  "note": "fixed total work: %d spp of the %dx%d frame over %d ranks (%d spp per rank per step)" % (K * S, W, H, world, max(1, S // world))}

    if rank == 0:
        samples_all = W * H * S * K * world
        rays = max(1, ctr["rays_extended"])
        means = {k: ctr["ctr_" + k] / rays for k in ("nodes", "pops", "refs", "tris")}
        ref_bpr = bytes_per_ray(means["nodes"], means["refs"], means["tris"], ctr["ctr_hits"] / rays)
        # (every launch of a render goes through a queue: 4 bytes; origins are charged for every ray although primary and
        #  direct-connection launches share theirs and read none -- an upper bound that is off by < 1 %)
        alg_bpr, per = product_bytes_per_ray(wrk, max(1, wrk["rays_extended"]), origin_bytes=16.0, queue_bytes=4.0)
        launches = max(1, tot["extend_launches"])
        line = {
            "metric": "samples_per_sec", "value": round(samples_all / elapsed, 1), "unit": "pixel-samples/s",
            "n_gpus": world, "steps": K, "warmup": Wm, "ms_per_step": round(elapsed / K * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            **({"rehearsal": "all %d ranks shared ONE GPU over gloo: checks the N > 1 code path, not scaling" % world} if args.share_gpu else {}),
            "config": {"workload": "C2: box room + icosphere L%d (%d tris), simple PT depth 6, %dx%d, %d spp per GPU"
                                   % (args.level, 20 * 4 ** args.level, W, H, K * S),
                       "spp_per_step": S, "integrator": "simple", "cam_depth": 6, "parallelism": "spp-interleave x%d + all-reduce" % world,
                       "scene_build": "rank 0 builds and packs, the arena is broadcast and imported by the others" if world > 1 else "local"},
            "rays_per_sample": round(tot["rays_total"] / max(1, tot["samples"]), 3),
            "rays_per_sample_note": "rays the product traced; the reference bound is 12 (exact elisions: shadow rays whose "
                                    "contribution is provably zero are not made, DESIGN.md 2 item 6)",
            "rays_through_extend_frac": round(tot["rays_extended"] / max(1, tot["rays_total"]), 4),
            "Mrays_per_s_rank0": round(tot["rays_total"] / elapsed / 1e6, 1),
            "extend_share_of_gpu_time": round(tot["ms_extend"] / max(1e-9, tot["ms_total"]), 3),
            "shade_share_of_gpu_time": round(tot["ms_shade"] / max(1e-9, tot["ms_total"]), 3),
            "image_mean": round(img_mean, 6), "nonfinite_samples": tot["nonfinite_samples"],
            "scene_seconds": round(t_build, 3), "commit_s": round(t_build, 3),
            "commit_note": "kd build + pack + upload of the C2 scene (81,920 triangles; rank 0 builds and packs it on the HOST so that the packed arena can be broadcast to the other ranks, then every rank uploads it; includes the first HIP call of the process); commits that build and pack on the GPU are in closest_hit.commit_s and other_configs[].commit_s",
            "roofline": roofline_views(pmc.get("c2") if pmc else None, tot["ms_extend"] / launches, tot["rays_extended"] / launches,
                                       alg_bpr, ref_bpr,
                                       {"launches": tot["extend_launches"], "pmc_source": pmc_source,
                                        "reference_walk_per_ray": {k: round(v, 2) for k, v in means.items()},
                                        "product_walk_per_ray": {k: round(v, 2) for k, v in per.items()}}),
        }
        line["roofline_k_shade"] = shade_roofline(pmc.get("c2_shade") if pmc else None, tot["ms_shade"] / max(1, tot["shade_launches"]),
                                                  tot["rays_camera"] / max(1, tot["shade_launches"]))
        if strong is not None:
            line["strong"] = strong
        if not args.no_cpu_baseline and world == 1:
            from oracle.binding import OracleScene
            orc = G.scene_icosphere_room(args.level).apply(OracleScene(threads=cpu_threads()))
            orc.commit()
            pstride = 61
            n_px = (W * H + pstride - 1) // pstride
            done, t_cpu = 0, 0.0
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
        if world > 1:
            committed = os.path.join(ROOT, "profiles", "r3_bench_default.json")
            if os.path.exists(committed):
                try:
                    cb = json.load(open(committed)).get("cpu_baseline")
                except ValueError:
                    cb = None
                if cb:
                    line["cpu_baseline"] = dict(cb, source="profiles/r3_bench_default.json: timed in the committed N = 1 run (the CPU leg runs on rank 0 at N = 1 only)")
        if not args.no_closest and world == 1:
            del scratch
            line["closest_hit"] = closest_hit_bench(torch, G, Scene, args.closest_rays, not args.no_cpu_baseline, pmc)
        if not args.no_other_configs and world == 1:
            scene.close()
            del accum
            torch.cuda.empty_cache()
            line["other_configs"] = []
            for name in ("C1", "C3", "C4", "C5"):
                t1 = time.time()
                line["other_configs"].append(run_other_config(name, G, Scene, make_config, local_rank))
                log("%s: %.1f s" % (name, time.time() - t1))
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
